#!/bin/bash
# Round 4, third GPU call: the multi-rank tests with the rebuilt library, the K2 suites with the wave-per-candidate mode, the
# ddmin record (config 4 end to end), and K1's convergence ceiling.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_comm_gpu.py tests/test_k2_gpu.py tests/test_blocked_actors_gpu.py -m gpu -q --timeout 900 > gpurun_out/r04_call3_tests.log 2>&1; tail -5 gpurun_out/r04_call3_tests.log
timeout 600 python bench.py --workload ddmin > gpurun_out/r04_ddmin_b.json 2> gpurun_out/r04_ddmin_b.err; tail -3 gpurun_out/r04_ddmin_b.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_ddmin_b.json').read().strip().splitlines()[-1])
print('ddmin value %.4g' % d['value'], 'frontiers', {k: round(v['kernel_us'], 1) for k, v in d['frontiers'].items()})
print('e2e', d['ddmin_end_to_end'])
print('cpu e2e', d['cpu_baseline']['ddmin_end_to_end'], d['cpu_baseline']['bit_identical_to_gpu'])
PY
timeout 600 python tools/r4_k1_ceiling.py 2>&1 | tail -9 | tee gpurun_out/r04_k1_ceiling.txt
