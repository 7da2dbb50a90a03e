#!/bin/bash
# Round 5, call 3: where the reference order's 55 ms go (timing split), per fetch width; per-kernel times of config 5 and config 3.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1 TMPDIR=/tmp
for w in 128 64 256 512; do
  echo "== reference order, fetch width $w"
  DEMI_DPOR_FETCH_WIDTH=$w DEMI_DPOR_TIMING=1 timeout 300 python bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline 2> gpurun_out/r05_ref_$w.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['orders']['reference_order']; print('  %.4g/s %.4f s kernels %.1f ms launches %d fetches %d d2h %.1f MB' % (r['value'], r['seconds'], r['kernel_ms_total'], r['launches'], r['record_fetches'], r['d2h_bytes']/1e6))"
  grep "dpor loop\|reference order" gpurun_out/r05_ref_$w.err | tail -2
done
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
for w in config5 dpor; do
  rm -rf /tmp/p_$w
  timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d /tmp/p_$w -o k -- python $R/bench.py --workload $w --no-cpu-baseline > /dev/null 2>&1
  python - <<PY
import glob, sqlite3
dbs = glob.glob("/tmp/p_$w/*.db") + glob.glob("/tmp/p_$w/*/*.db")
cur = sqlite3.connect(dbs[0]).cursor()
print("== $w: kernel-trace stats")
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 16"):
    print("%-70s %6d %12.0f us %10.1f us %6.2f" % (r[0][:70], r[1], r[2], r[3], r[4]))
PY
done
