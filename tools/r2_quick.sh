#!/bin/bash
# Wide tables on the GPU (parity with the oracle), K1 parity of the 8-bit layout after the word_t change, bench line twice
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wide_gpu.py -x -q --timeout 600 2>&1 | grep -vE "^RCCL|^HIP|^ROCm|^Hostname|^Librccl" | tail -25
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3
run() {  # name, flags
  name=$1; shift
  DEMI_JIT_FLAGS="$*" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'], 'clock', round(d['roofline']['probe']['shader_clock_ghz'],3))"
}
run default1; run default2; run O3_1 -O3; run O3_2 -O3
