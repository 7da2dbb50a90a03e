#!/bin/bash
# round-2 K1 experiment: bench lines of the specialised kernel's variants (+ parity of the ones that change memory ordering)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 60 --warmup 30 --no-cpu-baseline 2>gpurun_out/r2_sweep_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'])"
}
run base A=1
run nofence DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1"
run nofence_w7 DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" DEMI_JIT_K1_WAVES_PER_EU=7
run nofence_w8 DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" DEMI_JIT_K1_WAVES_PER_EU=8
run nofence_split DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" DEMI_JIT_FX_SPLIT=1
run nofence_w7_split DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" DEMI_JIT_K1_WAVES_PER_EU=7 DEMI_JIT_FX_SPLIT=1
DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" timeout 600 python -m pytest tests/test_k1_gpu.py -x -q --timeout 300 2>&1 | tail -2
DEMI_JIT_DEFINES="DEMI_K1_FLUSH_NOFENCE=1" DEMI_JIT_K1_WAVES_PER_EU=7 timeout 600 python -m pytest tests/test_k1_gpu.py -x -q --timeout 300 -k "parity_all_capacities or limits_matrix or full_size" 2>&1 | tail -2
