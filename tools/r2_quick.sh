#!/bin/bash
# The evidence run with the traced process on the same compiler as the bench, then the bench lines; and K1 under the system's
# ROCm 7.2 compiler (what a host without PyTorch gets) at three optimisation levels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 1500 bash tools/profile_r2.sh all > gpurun_out/r02_profile.log 2>&1; tail -3 gpurun_out/r02_profile.log
grep -h "the same traced run\|timed ones" gpurun_out/r02_k1.txt
cp gpurun_out/r02_k1_counters.json profiles/k1_counters.json
bash tools/r2_extra_lines.sh
for f in Os O3 O2; do
  LD_PRELOAD=/opt/rocm/lib/libamd_comgr.so.3 DEMI_JIT_FLAGS=-$f timeout 300 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('llvm22 $f', 'kernel_ms', round(r['kernel_ms'],3), 'value %.4g' % d['value'], r['kernel_code_id'])"
done
