#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# native DDMin (demi_ddmin): parity on the GPU, then the ddmin record
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_k2_gpu.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|Error|error" | tail -5
DEMI_DDMIN_TIMING=1 timeout 600 python bench.py --workload ddmin 2> gpurun_out/r3_ddmin.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k: d[k] for k in ('value', 'launch_floor', 'frontiers', 'ddmin_end_to_end', 'ddmin_end_to_end_python_mirror')}, indent=1))
print(json.dumps(d['cpu_baseline']['ddmin_end_to_end'], indent=1))
"
grep "\[ddmin\]" gpurun_out/r3_ddmin.err | sort | uniq -c | sort -rn | head -8
