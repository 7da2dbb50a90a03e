#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# round-2 K1 experiment: does a scratch working set that fits the L2 (fewer resident workgroups, some slots in LDS) pay?
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>gpurun_out/r2_sweep_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'])"
}
run base A=1
for wg in 3 4 5; do run wg$wg DEMI_K1_MAX_WG_PER_CU=$wg; done
for hot in 8 16 24; do run hot$hot DEMI_JIT_K1_HOT=$hot; for wg in 3 4; do run hot${hot}_wg$wg DEMI_JIT_K1_HOT=$hot DEMI_K1_MAX_WG_PER_CU=$wg; done; done
