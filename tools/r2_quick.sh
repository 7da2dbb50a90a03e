#!/bin/bash
# Code-layout / scheduling options of the specialised K1 on top of -Os (one process each)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, flags...
  name=$1; shift
  DEMI_JIT_FLAGS="$*" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_q_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$name', 'kernel_ms', round(r['kernel_ms'],3), 'value %.4g' % d['value'])" || tail -2 gpurun_out/r2_q_$name.err
}
run base
run wavepri -mllvm -amdgpu-set-wave-priority
run noloopalign -mllvm -amdgpu-disable-loop-alignment
run exttsp -mllvm -enable-ext-tsp-block-placement
run align5 -mllvm -align-all-nofallthru-blocks=5
run align6 -mllvm -align-all-nofallthru-blocks=6
run notaildup -mllvm -disable-tail-duplicate
run noplacement -mllvm -disable-block-placement
run bias100 -mllvm -amdgpu-schedule-metric-bias=100
run maxilp -mllvm -amdgpu-sched-strategy=max-ilp
run maxmem -mllvm -amdgpu-sched-strategy=max-memory-clause
run nobranchfold -mllvm -disable-branch-fold
