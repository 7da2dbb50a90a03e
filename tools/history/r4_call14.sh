#!/bin/bash
# Round 4, call 14: the K3 key plane - config 3 (both orders) and config 5, LDS-slot variants, launch durations by size, the K3 tests.
export DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
for hot in ${HOTS:-default 16 24 32 40}; do
  for wl in dpor config5; do
    if [ $hot = default ]; then E="DEMI_X=1"; else E="DEMI_JIT_K3_HOT=$hot"; fi
    env $E DEMI_K3_VERBOSE=1 timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r04_key_${hot}_$wl.json 2> gpurun_out/r04_key_${hot}_$wl.err
    python - <<PY
import json, re
d = json.loads(open('gpurun_out/r04_key_${hot}_$wl.json').read().strip().splitlines()[-1])
sh = sorted(set(re.findall(r"per_cu=\d+ lds=\d+ hot=\d+", open('gpurun_out/r04_key_${hot}_$wl.err').read())))
if '$wl' == 'dpor':
    o = d['orders']
    print('hot %-7s config3 rounds %.4g/s (%.2f ms, kernels %.2f ms, %s)  reference %.4g/s (%.2f ms, kernels %.2f ms, %s)  %s' % ('$hot', o['rounds']['value'], 1e3 * o['rounds']['seconds'], o['rounds']['kernel_ms_total'], o['rounds']['sequence_digest'][:8], o['reference_order']['value'], 1e3 * o['reference_order']['seconds'], o['reference_order']['kernel_ms_total'], o['reference_order']['sequence_digest'][:8], sh))
else:
    print('hot %-7s config5 %.4g/s (%.3f s, kernels %.1f ms, %s)  %s' % ('$hot', d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest'][:8], sh))
PY
  done
done
bash tools/k3_phases.sh 2>&1 | grep 'k3 phases' | head -4
} 2>&1 | tee gpurun_out/r04_k3_key_plane.txt
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_k3_gpu.py tests/test_blocked_actors_gpu.py tests/test_payloads_gpu.py tests/test_zz_array_gpu.py tests/test_wide_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r04_k3_key_tests.log
