#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_k2_gpu.py tests/test_wide_gpu.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|rror" | tail -5
DEMI_DDMIN_TIMING=1 timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2> gpurun_out/r3_ddmin2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'floor', d['launch_floor'])
print({k: (round(v['kernel_us']), round(v['wall_us'])) for k, v in d['frontiers'].items()})
print(d['ddmin_end_to_end'])
print(d['ddmin_end_to_end_python_mirror'])
"
grep "\[ddmin\]" gpurun_out/r3_ddmin2.err | tail -4
bash tools/k2_phases.sh | grep phases
