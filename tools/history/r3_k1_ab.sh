#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Round 3, K1 A/B in one call: parity of the current build (K1 suites), then the bench line's kernel_ms for each prebuilt
# variant under variants/*.so (base = round 2's kernel), the current build with the effect schedule switched off, the
# current build compiled by /opt/rocm's LLVM 22, and the phase split.  Writes gpurun_out/r3_ab_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
cp demi_amd/libdemi_gpu.so /tmp/libdemi_gpu.so.cur
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms_per_step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'code', d['roofline'].get('kernel_code_id'), 'cpu_same', (d.get('cpu_baseline') or {}).get('bit_identical_to_gpu'))
except Exception as e: print('$1', 'FAILED', e)"; }
echo "== parity (current build)"
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py tests/test_wide_gpu.py -x -q --timeout 600 2>&1 | tail -4
echo "== bench variants"
for v in variants/*.so; do
  cp $v demi_amd/libdemi_gpu.so
  timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>gpurun_out/r3_ab_$(basename $v .so).err | tee gpurun_out/r3_ab_$(basename $v .so).json | line $(basename $v .so)
done
cp /tmp/libdemi_gpu.so.cur demi_amd/libdemi_gpu.so
timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary 2>gpurun_out/r3_ab_cur.err | tee gpurun_out/r3_ab_cur.json | line cur_with_cpu_check
DEMI_JIT_NO_FX_SCHED=1 timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3_ab_cur_nosched.json | line cur_no_fx_sched
LD_PRELOAD=/opt/rocm/lib/libamd_comgr.so.3 timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | tee gpurun_out/r3_ab_cur_llvm22.json | line cur_llvm22
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --no-specialize 2>/dev/null | tee gpurun_out/r3_ab_cur_interp.json | line cur_interpreter
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --strategy fifo 2>/dev/null | tee gpurun_out/r3_ab_cur_fifo.json | line cur_fifo
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline --wide-term0 1000 2>/dev/null | tee gpurun_out/r3_ab_cur_wide.json | line cur_wide
echo "== phases"
timeout 600 bash tools/k1_phases.sh 2>&1 | tail -4
cp /tmp/libdemi_gpu.so.cur demi_amd/libdemi_gpu.so
