#!/bin/bash
# K1 parity on the planes layout, then knob runs (each in its own process, several times: see tools/README.md)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py -x -q --timeout 300 2>&1 | tail -3
run() {  # name, flags
  name=$1; shift
  DEMI_JIT_FLAGS="$*" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'], 'clock', round(d['roofline']['probe']['shader_clock_ghz'],3))"
}
run default1; run default2
DEMI_JIT_K1_HOT=4 run hot4_1; DEMI_JIT_K1_HOT=4 run hot4_2
DEMI_JIT_K1_HOT=2 run hot2_1
DEMI_JIT_K1_WAVES_PER_EU=7 run w7_1; DEMI_JIT_K1_WAVES_PER_EU=7 run w7_2
DEMI_K1_MAX_WG_PER_CU=5 run wg5
