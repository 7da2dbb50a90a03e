#!/bin/bash
# -Os for the specialised K2 / K3 as well? (K1 gained 8 % from it)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
for v in O3 Os O2; do
  DEMI_JIT_FLAGS=-$v timeout 300 python bench.py --workload ddmin --no-cpu-baseline 2>gpurun_out/r2_q_dd_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ddmin $v', {k: round(v['kernel_us']) for k,v in d['frontiers'].items()}, 'value %.4g' % d['value'])"
  DEMI_JIT_FLAGS=-$v timeout 300 python bench.py --workload dpor --no-cpu-baseline 2>gpurun_out/r2_q_dp_$v.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dpor $v', {k: (round(v['value']), round(v['kernel_ms_total'],1)) for k,v in d['orders'].items()})"
done
