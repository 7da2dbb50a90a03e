#!/bin/bash
# Round 4, call 9: K3 launch shape (lanes per wave as few as fill the resident waves, instead of powers of two from 8) and the
# prefix head requested one step ahead - A/B on config 3 (both orders) and config 5, then the K3 parity tests on the device.
export DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  for wl in dpor config5; do
    env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r04_k3ab_${name}_$wl.json 2> gpurun_out/r04_k3ab_${name}_$wl.err
    python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r04_k3ab_${name}_$wl.json').read().strip().splitlines()[-1])
    if '$wl' == 'dpor':
        o = d['orders']
        print('%-22s dpor    rounds %.4g/s (%.2f ms, kernels %.2f ms, digest %s)   reference %.4g/s (%.2f ms, kernels %.2f ms, digest %s)' % ('$name', o['rounds']['value'], 1e3 * o['rounds']['seconds'], o['rounds']['kernel_ms_total'], o['rounds']['sequence_digest'], o['reference_order']['value'], 1e3 * o['reference_order']['seconds'], o['reference_order']['kernel_ms_total'], o['reference_order']['sequence_digest']))
    else:
        print('%-22s config5 %.4g/s (%.3f s, kernels %.1f ms, digest %s)' % ('$name', d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest']))
except Exception as ex:
    print('$name $wl failed:', ex, open('gpurun_out/r04_k3ab_${name}_$wl.err').read()[-500:])
PY
  done
}
{
run old_pow2_noprefetch DEMI_K3_LANES_POW2=1 DEMI_JIT_DEFINES=DEMI_K3_NO_PREFETCH=1
run pow2_prefetch DEMI_K3_LANES_POW2=1
run fine_noprefetch DEMI_JIT_DEFINES=DEMI_K3_NO_PREFETCH=1
run fine_prefetch DEMI_X=1
run fine_min2 DEMI_K3_MIN_LANES=2
run fine_min8 DEMI_K3_MIN_LANES=8
} 2>&1 | tee gpurun_out/r04_k3_shape_ab.txt
timeout 600 python -m pytest tests/test_k3_gpu.py tests/test_blocked_actors_gpu.py -m gpu -x -q --durations=5 2>&1 | tail -12 | tee gpurun_out/r04_k3_tests.log
