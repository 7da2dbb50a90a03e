#!/bin/bash
# Optimisation level of the specialised K1 (each twice, own process), and the wide raft on the bench trace
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, bench args..., -- flags
  name=$1; shift
  timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary $BARGS 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'], 'clock', round(d['roofline']['probe']['shader_clock_ghz'],3), d['config'].get('wide_register_window'))"
}
for f in -Oz -O2 -O1; do DEMI_JIT_FLAGS=$f run ${f}_1; DEMI_JIT_FLAGS=$f run ${f}_2; done
DEMI_K1_VERBOSE=1 BARGS="--wide-term0 1000" run wide1; BARGS="--wide-term0 1000" run wide2
grep -m1 "k1 launch" gpurun_out/r2_q_wide1.err
