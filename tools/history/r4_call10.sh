#!/bin/bash
# Round 4, call 10: the racing-pair analysis as a kernel of its own (k3_analyze: one wave per finished trace) - config 3 in both
# orders and config 5, lanes-per-wave variants, the launch shapes, then the K3 / DPOR parity tests on the device.
export DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
run() {  # name, env...
  local name=$1; shift
  for wl in dpor config5; do
    env "$@" timeout 300 python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r04_k3split_${name}_$wl.json 2> gpurun_out/r04_k3split_${name}_$wl.err
    python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r04_k3split_${name}_$wl.json').read().strip().splitlines()[-1])
    if '$wl' == 'dpor':
        o = d['orders']
        print('%-14s dpor    rounds %.4g/s (%.2f ms, kernels %.2f ms, digest %s)   reference %.4g/s (%.2f ms, kernels %.2f ms, digest %s)' % ('$name', o['rounds']['value'], 1e3 * o['rounds']['seconds'], o['rounds']['kernel_ms_total'], o['rounds']['sequence_digest'], o['reference_order']['value'], 1e3 * o['reference_order']['seconds'], o['reference_order']['kernel_ms_total'], o['reference_order']['sequence_digest']))
    else:
        print('%-14s config5 %.4g/s (%.3f s, kernels %.1f ms, digest %s)' % ('$name', d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest']))
except Exception as ex:
    print('$name $wl failed:', ex, open('gpurun_out/r04_k3split_${name}_$wl.err').read()[-500:])
PY
  done
}
{
run default DEMI_X=1
run min2 DEMI_K3_MIN_LANES=2
run min1 DEMI_K3_MIN_LANES=1
run min8 DEMI_K3_MIN_LANES=8
run pow2 DEMI_K3_LANES_POW2=1
DEMI_K3_VERBOSE=1 timeout 300 python bench.py --workload dpor --no-cpu-baseline --dpor-order rounds 2>&1 >/dev/null | grep 'k3 launch' | sort | uniq -c | sort -rn | head -12
} 2>&1 | tee gpurun_out/r04_k3_split_ab.txt
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_blocked_actors_gpu.py tests/test_comm_gpu.py tests/test_payloads_gpu.py tests/test_zz_array_gpu.py -m gpu -x -q --durations=8 2>&1 | tail -16 | tee gpurun_out/r04_k3_split_tests.log
