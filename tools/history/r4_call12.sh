#!/bin/bash
# Round 4, call 12: two knob sweeps.  (a) how many interleavings of the commit's queue front one record fetch of the
# REFERENCE-order exploration covers (DEMI_DPOR_FETCH_WIDTH, default 128: config 3 makes 514 fetches of ~45 us each);
# (b) LDS-resident pending slots of the compiled k3_dpor (DEMI_JIT_K3_HOT) on config 3 and on config 5 (8 actors), now that the
# racing-pair analysis no longer takes 9 KB of LDS per wave.
export DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
{
for w in 128 256 512 1024 4096; do
  DEMI_DPOR_TIMING=1 DEMI_DPOR_FETCH_WIDTH=$w timeout 300 python bench.py --workload dpor --no-cpu-baseline --dpor-order reference_order > gpurun_out/r04_fw_$w.json 2> gpurun_out/r04_fw_$w.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r04_fw_$w.json').read().strip().splitlines()[-1])['orders']['reference_order']
print('fetch width %5d: %.4g/s  %.2f ms  kernels %.2f ms  fetches %d  d2h %.1f MB  digest %s' % ($w, d['value'], 1e3 * d['seconds'], d['kernel_ms_total'], d['record_fetches'], d['d2h_bytes'] / 1e6, d['sequence_digest']))
PY
  grep 'dpor loop' gpurun_out/r04_fw_$w.err | tail -1
done
for hot in 12 24 40 64; do
  for wl in dpor config5; do
    DEMI_JIT_K3_HOT=$hot timeout 300 python bench.py --workload $wl --no-cpu-baseline --dpor-order rounds > gpurun_out/r04_hot_${hot}_$wl.json 2> gpurun_out/r04_hot_${hot}_$wl.err
    python - <<PY
import json
d = json.loads(open('gpurun_out/r04_hot_${hot}_$wl.json').read().strip().splitlines()[-1])
if '$wl' == 'dpor':
    o = d['orders']['rounds']; print('hot %2d config3 rounds %.4g/s  %.2f ms  kernels %.2f ms' % ($hot, o['value'], 1e3 * o['seconds'], o['kernel_ms_total']))
else:
    print('hot %2d config5 %.4g/s  %.3f s  kernels %.1f ms  digest %s' % ($hot, d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest']))
PY
  done
done
} 2>&1 | tee gpurun_out/r04_fetch_width_and_hot.txt
