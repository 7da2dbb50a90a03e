#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Round 3: where the specialised K1 stands after the effect schedule: counters (issue vs wait), and whether the scratch
# working set is now what bounds it (fewer resident workgroups, LDS-resident pending slots).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>gpurun_out/r3_sweep_$name.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'kernel_ms', round(d['roofline']['kernel_ms'],3), 'value %.4g' % d['value'])"
}
run base A=1
for wg in 3 4 5; do run wg$wg DEMI_K1_MAX_WG_PER_CU=$wg; done
for hot in 4 8 12 16; do run hot$hot DEMI_JIT_K1_HOT=$hot; done
run hot8_wg4 DEMI_JIT_K1_HOT=8 DEMI_K1_MAX_WG_PER_CU=4
run hot16_wg3 DEMI_JIT_K1_HOT=16 DEMI_K1_MAX_WG_PER_CU=3
DEMI_K1_VERBOSE=1 timeout 120 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-prewarm 2>&1 | grep "k1 launch" | tail -1
timeout 900 bash tools/profile_r3.sh > gpurun_out/r03_profile.log 2>&1; tail -3 gpurun_out/r03_profile.log
cat gpurun_out/r03_k1_counters.json | head -30
