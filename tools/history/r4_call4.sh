#!/bin/bash
# Round 4, fourth GPU call: the re-binned K1 against the plain kernel (one process), the K1 suite with the re-binned kernel on,
# the K2 suites with the window walk, config 4's DDMin end to end.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python tools/r4_k1_rebin_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_k1_rebin_ab.txt
DEMI_EXPERIMENT=1 DEMI_K1_REBIN=1 timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_invariant_gpu.py -m gpu -q --timeout 800 -x > gpurun_out/r04_call4_k1_rebin_tests.log 2>&1; tail -3 gpurun_out/r04_call4_k1_rebin_tests.log
timeout 900 python -m pytest tests/test_k2_gpu.py tests/test_k3_gpu.py -m gpu -q --timeout 800 > gpurun_out/r04_call4_k2k3_tests.log 2>&1; tail -3 gpurun_out/r04_call4_k2k3_tests.log
timeout 600 python bench.py --workload ddmin > gpurun_out/r04_ddmin_c.json 2> gpurun_out/r04_ddmin_c.err; tail -3 gpurun_out/r04_ddmin_c.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_ddmin_c.json').read().strip().splitlines()[-1])
print('ddmin value %.4g' % d['value'], 'frontiers', {k: round(v['kernel_us'], 1) for k, v in d['frontiers'].items()})
print('e2e', d['ddmin_end_to_end'])
print('cpu e2e', d['cpu_baseline']['ddmin_end_to_end'], d['cpu_baseline']['bit_identical_to_gpu'])
PY
