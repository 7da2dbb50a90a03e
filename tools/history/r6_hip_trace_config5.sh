#!/bin/bash
# where the host's time goes in the reference order (config 3): HIP API trace + kernel trace, stats only
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/profh5
rm -rf $P; mkdir -p $OUT $P
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
timeout 600 rocprofv3 --preload $COMGR --hip-trace --kernel-trace --stats -d $P/s -o ref -- python $R/bench.py --workload config5 --dpor-order rounds --no-cpu-baseline > $OUT/r06_hip_trace_config5.log 2>&1
python - <<PY
import glob, sqlite3
dbs = glob.glob("$P/s/*.db")
con = sqlite3.connect(dbs[0]); cur = con.cursor()
names = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
print([n for n in names if 'top' in n or 'stat' in n.lower()][:20])
lines = []
for v in [n for n in names if n.startswith('top')]:
    try:
        lines.append("## " + v)
        for r in list(cur.execute("select * from %s" % v))[:25]:
            lines.append("  " + " | ".join(str(x)[:60] for x in r))
    except Exception as e:
        lines.append("  (%s)" % e)
open("$OUT/r06_hip_trace_config5.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:70]))
PY
