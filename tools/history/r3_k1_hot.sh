#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Round 3: with the effect queue cut to the SEND slots there is LDS for a window of pending slots at full occupancy: sweep
# the window (DEMI_JIT_K1_HOT) against the resident workgroups per CU.  Parity of the mixed LDS / HBM paths first.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
echo "== parity with a 15-slot window"
DEMI_JIT_K1_HOT=15 timeout 600 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py -x -q --timeout 300 2>&1 | tail -2
run() {  # name, env...
  name=$1; shift
  env "$@" DEMI_K1_VERBOSE=1 timeout 200 python bench.py --steps 30 --warmup 5 --cpu-sample 131072 --no-secondary 2>gpurun_out/r3_hot_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'], 'same', d['cpu_baseline']['bit_identical_to_gpu'])"
  grep "k1 launch" gpurun_out/r3_hot_$name.err | tail -1 | cut -c1-90
}
run hot0_auto A=1
run hot0_wg6 DEMI_K1_MAX_WG_PER_CU=6
run hot0_wg5 DEMI_K1_MAX_WG_PER_CU=5
run hot6_auto DEMI_JIT_K1_HOT=6
run hot9_auto DEMI_JIT_K1_HOT=9
run hot12_auto DEMI_JIT_K1_HOT=12
run hot15_auto DEMI_JIT_K1_HOT=15
run hot19_auto DEMI_JIT_K1_HOT=19
run hot23_auto DEMI_JIT_K1_HOT=23
run hot28_auto DEMI_JIT_K1_HOT=28
run hot36_auto DEMI_JIT_K1_HOT=36
