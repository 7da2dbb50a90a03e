#!/bin/bash
# Round 3: evidence for the secondary records, reproducible from profiles/:
#   r03_ddmin.txt / r03_dpor.txt          kernel-trace stats incl. a per-launch-shape table (256 ... 2^20 candidates apart)
#   r03_ddmin_counters.txt, r03_dpor_counters.txt   one --pmc pass each (instructions, active lanes per VALU instruction)
#   r03_dpor_counters.json                fabric bytes per ROUNDS exploration (FETCH_SIZE x 2048 + WRITE_SIZE x 1024, the
#                                          calibration of tools/calib_counters.py) = the `traffic` of the dpor roofline
# Every rocprofv3 call has its own timeout (a pass that asks for too many counters aborts and then hangs in finalisation).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/prof23
rm -rf $P; mkdir -p $OUT $P
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
PRE="--preload $COMGR"
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_ddmin -o k2 -- python $R/bench.py --workload ddmin --no-cpu-baseline > $OUT/r03_prof_stats_ddmin.log 2>&1
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_dpor -o k3 -- python $R/bench.py --workload dpor --no-cpu-baseline > $OUT/r03_prof_stats_dpor.log 2>&1
for w in ddmin dpor; do
  timeout 300 rocprofv3 $PRE --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $P/$w -o c -- python $R/bench.py --workload $w --no-cpu-baseline --dpor-order rounds > $OUT/r03_pmc_$w.log 2>&1
done
timeout 300 rocprofv3 $PRE --pmc FETCH_SIZE -d $P/dpor_fetch -o c -- python $R/bench.py --workload dpor --no-cpu-baseline --dpor-order rounds > $OUT/r03_pmc_dpor_fetch.log 2>&1
timeout 300 rocprofv3 $PRE --pmc WRITE_SIZE -d $P/dpor_write -o c -- python $R/bench.py --workload dpor --no-cpu-baseline --dpor-order rounds > $OUT/r03_pmc_dpor_write.log 2>&1
python $R/tools/summarize_prof.py r03x $P $OUT > /dev/null 2>&1
mv $OUT/r03x_ddmin.txt $OUT/r03_ddmin.txt 2>/dev/null; mv $OUT/r03x_dpor.txt $OUT/r03_dpor.txt 2>/dev/null; rm -f $OUT/r03x_k1.txt
python - <<'PY'
import glob, json, os, sqlite3
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
    lines = ["# python bench.py --workload %s --no-cpu-baseline%s under rocprofv3 --pmc: per kernel, dispatches, average and total per counter" % (w, " --dpor-order rounds" if w == "dpor" else "")]
    for kn, cs in rows.items():
        lines.append(kn[:100])
        for cn, (cnt, avg, tot) in sorted(cs.items()):
            lines.append("    %-24s %6d dispatches  avg %16.1f  total %18.1f" % (cn, cnt, avg, tot))
        if "SQ_INSTS_VALU" in cs and "SQ_THREAD_CYCLES_VALU" in cs and cs["SQ_INSTS_VALU"][2]:
            lines.append("    active lanes per VALU instruction: %.1f" % (cs["SQ_THREAD_CYCLES_VALU"][2] / cs["SQ_INSTS_VALU"][2]))
    open(os.path.join(out, "r03_%s_counters.txt" % w), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))
tot = {}
for d, cn in (("dpor_fetch", "FETCH_SIZE"), ("dpor_write", "WRITE_SIZE")):
    dbs = glob.glob("/tmp/prof23/%s/*.db" % d)
    if dbs:
        cur = sqlite3.connect(dbs[0]).cursor()
        tot[cn] = list(cur.execute("select sum(value) from counters_collection where kernel_name like '%demi%' and counter_name = ?", (cn,)))[0][0]
if len(tot) == 2 and all(v is not None for v in tot.values()):
    # bench.py --workload dpor runs the exploration twice per order (one untimed, one timed): two explorations in the profile
    fabric = (tot["FETCH_SIZE"] * 2048.0 + tot["WRITE_SIZE"] * 1024.0) / 2.0
    json.dump({"fabric_bytes_per_exploration": fabric, "FETCH_SIZE_total": tot["FETCH_SIZE"], "WRITE_SIZE_total": tot["WRITE_SIZE"],
               "explorations_profiled": 2, "bytes_per_unit": {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0},
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over python bench.py --workload dpor --dpor-order rounds, all demi kernels"},
              open(os.path.join(out, "r03_dpor_counters.json"), "w"), indent=1)
    print("dpor fabric bytes per exploration: %.3e" % fabric)
PY
