#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_comm_gpu.py tests/test_k1_gpu.py -x -q --timeout 600 -k "reference or bench_py or carried or golden or resident" 2>&1 | tail -12
DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --no-cpu-baseline 2> gpurun_out/r3_k3.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['orders'].items(): print(k, round(v['value']), 'sec %.3f' % v['seconds'], 'il', v['interleavings'], 'exec', v['executed_on_device'], 'launches', v['launches'], 'kernel_ms %.1f' % v['kernel_ms_total'], 'd2h', v['d2h_bytes'], 'h2d', v['h2d_bytes'], v['sequence_digest'])
"
grep -E "dpor|reference" gpurun_out/r3_k3.err | tail -8
