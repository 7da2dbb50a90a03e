#!/bin/bash
# Round 5, call 16: the reference order as kept (two-launch record fetch with the request in the kernel's arguments and a short
# prefix loop, a launch's copies through one pinned area and one synchronisation, the host loop with its lines requested ahead).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1 TMPDIR=/tmp
timeout 900 python -m pytest tests/test_k3_gpu.py -m gpu -x -q -k "reference or golden or resident or queue or transliteration" 2>&1 | tail -2
for v in "" "DEMI_DPOR_FETCH_WIDTH=160"; do
  echo "== reference order [$v]"
  for rep in 1 2 3; do
  env $v DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline 2> gpurun_out/r05_ref16.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['orders']['reference_order']; print('  %.4g/s %.4f s kernels %.1f ms launches %d fetches %d d2h %.1f MB digest %s' % (r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['record_fetches'], r['d2h_bytes']/1e6, r['sequence_digest']))"
  done
  grep "dpor loop\|dpor reference" gpurun_out/r05_ref16.err | tail -2
done
cd /tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof16 -- python $R/bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline > /dev/null 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof16/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print("  %-60s calls %6s total %10s ns avg %9s ns" % (r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"]))
PY
