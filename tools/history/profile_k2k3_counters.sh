#!/bin/bash
# Issue counters of the K2 (bench.py --workload ddmin) and K3 (--workload dpor) kernels: one --pmc pass each, condensed into
# gpurun_out/r02_<workload>_counters.txt (instructions per kernel, active lanes per VALU instruction).  The traced process is
# given PyTorch's comgr so that the tables are specialised by the same compiler as in an untraced run.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/prof23
rm -rf $P; mkdir -p $OUT $P
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
for w in ddmin dpor; do
  rocprofv3 --preload $COMGR --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $P/$w -o c -- python $R/bench.py --workload $w --no-cpu-baseline > $OUT/r02_pmc_$w.log 2>&1
done
python - <<'PY'
import glob, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
for w in ("ddmin", "dpor"):
    dbs = glob.glob("/tmp/prof23/%s/*.db" % w)
    if not dbs:
        print(w, "no database"); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection "
         "where kernel_name like '%demi%' group by kernel_name, counter_name")
    rows = {}
    for kn, cn, cnt, avg, tot in cur.execute(q):
        rows.setdefault(kn, {})[cn] = (cnt, avg, tot)
    lines = ["# python bench.py --workload %s --no-cpu-baseline under rocprofv3 --pmc: per kernel, dispatches, average and total per counter" % w]
    for kn, cs in rows.items():
        lines.append(kn[:100])
        for cn, (cnt, avg, tot) in sorted(cs.items()):
            lines.append("    %-24s %6d dispatches  avg %16.1f  total %18.1f" % (cn, cnt, avg, tot))
        if "SQ_INSTS_VALU" in cs and "SQ_THREAD_CYCLES_VALU" in cs and cs["SQ_INSTS_VALU"][2]:
            lines.append("    active lanes per VALU instruction: %.1f" % (cs["SQ_THREAD_CYCLES_VALU"][2] / cs["SQ_INSTS_VALU"][2]))
    open(os.path.join(out, "r02_%s_counters.txt" % w), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:30]))
PY
