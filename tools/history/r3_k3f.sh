#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_wide_gpu.py tests/test_comm_gpu.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|Error" | tail -4
timeout 300 python bench.py --workload dpor --no-cpu-baseline 2> gpurun_out/r3_k3f.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['orders'].items(): print(k, round(v['value']), 'sec %.4f' % v['seconds'], 'il', v['interleavings'], 'launches', v['launches'], 'kernel_ms %.1f' % v['kernel_ms_total'], 'd2h', v['d2h_bytes'], v['sequence_digest'])
"
bash tools/k3_phases.sh | tail -3
