#!/bin/bash
# scratch-layout A/B for K1 + the GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'])"
}
run blocked A=1
run lanemajor DEMI_JIT_DEFINES=DEMI_SPILL_LANE_MAJOR=1
run blocked_wg5 DEMI_K1_MAX_WG_PER_CU=5
run blocked_wg4 DEMI_K1_MAX_WG_PER_CU=4
run blocked_hot8 DEMI_JIT_K1_HOT=8
run blocked2 A=1
timeout 1500 python -m pytest tests -m gpu -x -q --timeout 900 2>&1 | tail -8
