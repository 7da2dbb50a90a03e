#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# after the racing-pair rewrite: parity, then the dpor record in both orders, the phase split, residency variants
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_wide_gpu.py tests/test_comm_gpu.py -x -q --timeout 600 2>&1 | tail -4
run() {
  timeout 300 python bench.py --workload dpor $2 --no-cpu-baseline 2> gpurun_out/r3_k3e.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['orders'].items(): print('$1', k, round(v['value']), 'sec %.4f' % v['seconds'], 'il', v['interleavings'], 'launches', v['launches'], 'kernel_ms %.1f' % v['kernel_ms_total'], 'd2h', v['d2h_bytes'], v['sequence_digest'])
"
  grep "k3 launch" gpurun_out/r3_k3e.err | sort | uniq -c | sort -rn | head -1
}
export DEMI_K3_VERBOSE=1
run default
DEMI_JIT_K3_HOT=8 DEMI_K3_WAVES=2 run hot8w2 "--dpor-order rounds"
DEMI_JIT_K3_HOT=16 DEMI_K3_WAVES=2 run hot16w2 "--dpor-order rounds"
DEMI_JIT_K3_HOT=32 DEMI_K3_WAVES=4 run hot32w4 "--dpor-order rounds"
DEMI_JIT_K3_HOT=12 DEMI_K3_WAVES=2 DEMI_K3_LANES_PER_WAVE=32 run hot12w2l32 "--dpor-order rounds"
unset DEMI_K3_VERBOSE
bash tools/k3_phases.sh | tail -4
