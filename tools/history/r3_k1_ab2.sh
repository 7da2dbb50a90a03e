#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Round 3, second K1 A/B: prebuilt variants (variants/*.so), then the current build under JIT defines (last-word register,
# service iterations), each with the bit-for-bit check of 2^18 schedules against the oracle; phases of two of them.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
cp demi_amd/libdemi_gpu.so /tmp/libdemi_gpu.so.cur
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'kernel_ms %.4f' % d['roofline']['kernel_ms'], 'ms_per_step %.4f' % d['ms_per_step'], 'value %.4g' % d['value'], 'code', d['roofline'].get('kernel_code_id'), 'cpu_same', (d.get('cpu_baseline') or {}).get('bit_identical_to_gpu'))
except Exception as e: print('$1', 'FAILED', e)"; }
echo "== parity (current build, K1 suites)"
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_blocked_actors_gpu.py tests/test_wide_gpu.py -x -q --timeout 600 2>&1 | tail -3
echo "== prebuilt variants"
for v in variants/*.so; do
  cp $v demi_amd/libdemi_gpu.so
  timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --cpu-sample 262144 2>gpurun_out/r3_ab2_$(basename $v .so).err | tee gpurun_out/r3_ab2_$(basename $v .so).json | line $(basename $v .so)
done
cp /tmp/libdemi_gpu.so.cur demi_amd/libdemi_gpu.so
echo "== current build under JIT defines"
for D in "" "DEMI_K1_NO_LASTW=1" "DEMI_K1_SVC_PERIOD=2" "DEMI_K1_SVC_PERIOD=4" "DEMI_K1_SVC_PERIOD=8" "DEMI_K1_SVC_PERIOD=4;DEMI_K1_SVC_LANES=8" "DEMI_K1_SVC_PERIOD=4;DEMI_K1_SVC_LANES=32" "DEMI_K1_SVC_PERIOD=8;DEMI_K1_SVC_LANES=24" "DEMI_K1_SVC_PERIOD=16;DEMI_K1_SVC_LANES=24"; do
  DEMI_JIT_DEFINES="$D" timeout 300 python bench.py --steps 40 --warmup 5 --no-secondary --cpu-sample 262144 2>/tmp/err.txt | tee "gpurun_out/r3_ab2_cur_$(echo $D | tr ';=' '__').json" | line "cur[$D]"
done
echo "== phases"
PHASES_JIT_ONLY=1 timeout 300 bash tools/k1_phases.sh 2>&1 | tail -2
DEMI_JIT_DEFINES="DEMI_K1_SVC_PERIOD=4" PHASES_JIT_ONLY=1 timeout 300 bash tools/k1_phases.sh 2>&1 | tail -2
cp /tmp/libdemi_gpu.so.cur demi_amd/libdemi_gpu.so
