#!/bin/bash
# Round 5, call 2: the K3 suite and the new randomDDMin suite on the device with the parent filter + device-resident queue, then
# where the DPOR records stand (timing split, A/B against the host queue and against probing every pair), and the bench line.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_random_ddmin_gpu.py tests/test_comm_gpu.py -m gpu -x -q --durations=8 > gpurun_out/r05_gpu_tests_call2.log 2>&1
grep -E 'passed|failed|error' gpurun_out/r05_gpu_tests_call2.log | tail -3
export DEMI_EXPERIMENT=1
for v in "" "DEMI_K3_NO_PARENT_FILTER=1" "DEMI_DPOR_HOST_QUEUE=1" "DEMI_DPOR_HOST_QUEUE=1 DEMI_K3_NO_PARENT_FILTER=1"; do
  echo "== config5 [$v]"
  env $v DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload config5 --no-cpu-baseline 2> gpurun_out/r05_c5_$(echo $v | tr ' =' '__').err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  %.4g/s  %.3f s  kernels %.1f ms  launches %d  d2h %.1f MB  digest %s' % (d['value'], d['seconds'], d['kernel_ms_total'], d['launches'], d['pcie_bytes']['d2h']/1e6, d['sequence_digest']))"
  grep "racing pairs\|dpor loop" gpurun_out/r05_c5_$(echo $v | tr ' =' '__').err | tail -2
done
for v in "" "DEMI_DPOR_HOST_QUEUE=1 DEMI_K3_NO_PARENT_FILTER=1"; do
  echo "== config3 [$v]"
  env $v timeout 300 python bench.py --workload dpor --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for k,r in d['orders'].items(): print('  %s %.4g/s %.4f s kernels %.1f ms launches %d digest %s' % (k, r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['sequence_digest']))"
done
echo "== ddmin record"
timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps(d.get('random_ddmin_R100'), indent=1)[:1800])"
