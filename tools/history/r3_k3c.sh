#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# where k3_dpor's time goes: lanes per wave, and the interleavings without the pair analysis (ROUNDS order, config 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() {
  timeout 300 python bench.py --workload dpor --dpor-order rounds --no-cpu-baseline 2> gpurun_out/r3_k3c.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d['orders'].items(): print('$1', k, round(v['value']), 'sec %.4f' % v['seconds'], 'il', v['interleavings'], 'launches', v['launches'], 'kernel_ms %.1f' % v['kernel_ms_total'])
"
  grep "k3 launch" gpurun_out/r3_k3c.err | sort | uniq -c | sort -rn | head -4
}
DEMI_K3_VERBOSE=1 run default
for l in 4 16 32; do DEMI_K3_LANES_PER_WAVE=$l run lanes$l; done
DEMI_JIT_DEFINES="DEMI_K3_NO_PAIRS=1" run nopairs
