#!/bin/bash
# kernel-trace stats of config 5 in ROUNDS of $1 (default 65536): which kernels a wide round's time goes to
export TMPDIR=/tmp
W=${1:-65536}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/profb
rm -rf $P; mkdir -p $OUT $P
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d $P/s -o k3 -- python $R/tools/r6_batch_sweep.py $W > $OUT/r06_batch_prof_$W.log 2>&1
python - <<PY
import glob, sqlite3
dbs = glob.glob("$P/s/*.db")
cur = sqlite3.connect(dbs[0]).cursor()
lines = ["# tools/r6_batch_sweep.py $W (config 5, ROUNDS of $W, 4 explorations): rocprofv3 --kernel-trace --stats (ns)",
         "%-70s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct")]
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    lines.append("%-70s %8d %14.0f %12.0f %7.2f" % (r[0][:70], r[1], r[2] * 1000, r[3] * 1000, r[4]))
open("$OUT/r06_config5_w$W.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:16]))
PY
