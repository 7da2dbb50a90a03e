#!/bin/bash
# Round 6 (TAG=r06 by default): evidence for the secondary records, reproducible from profiles/:
#   <tag>_ddmin.txt / <tag>_dpor.txt / <tag>_config5.txt   kernel-trace stats incl. a per-launch-shape table
#   <tag>_ddmin_counters.txt, <tag>_dpor_counters.txt    one --pmc pass each (instructions, active lanes per VALU instruction)
#   <tag>_dpor_counters.json, <tag>_config5_counters.json  fabric bytes per exploration (FETCH_SIZE x 2048 + WRITE_SIZE x 1024, the
#                                          calibration of tools/calib_counters.py) = the `traffic` of those records' rooflines
#   <tag>_ddmin_counters.json               the same per 2^20-candidate launch of k2_replay_fp_hbm (the six largest dispatches)
#   <tag>_{ddmin,dpor,config5}_insts.json    instruction / wave-cycle counters in the form bench.py's issue model reads
# Every rocprofv3 call has its own timeout (a pass that asks for too many counters aborts and then hangs in finalisation).
export TMPDIR=/tmp
export TAG=${TAG:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
P=/tmp/prof23
rm -rf $P; mkdir -p $OUT $P
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
PRE="--preload $COMGR"
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_ddmin -o k2 -- python $R/bench.py --workload ddmin --no-cpu-baseline > $OUT/${TAG}_prof_stats_ddmin.log 2>&1
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_dpor -o k3 -- python $R/bench.py --workload dpor --no-cpu-baseline > $OUT/${TAG}_prof_stats_dpor.log 2>&1
timeout 300 rocprofv3 $PRE --kernel-trace --stats -d $P/prof_stats_config5 -o k3 -- python $R/bench.py --workload config5 --no-cpu-baseline --dpor-order rounds > $OUT/${TAG}_prof_stats_config5.log 2>&1
for w in ddmin dpor config5; do
  timeout 300 rocprofv3 $PRE --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE -d $P/$w -o c -- python $R/bench.py --workload $w --no-cpu-baseline --dpor-order rounds > $OUT/${TAG}_pmc_$w.log 2>&1
  timeout 300 rocprofv3 $PRE --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY -d $P/${w}_sq2 -o c -- python $R/bench.py --workload $w --no-cpu-baseline --dpor-order rounds > $OUT/${TAG}_pmc_${w}_sq2.log 2>&1
done
for w in dpor ddmin config5; do
  timeout 300 rocprofv3 $PRE --pmc FETCH_SIZE -d $P/${w}_fetch -o c -- python $R/bench.py --workload $w --no-cpu-baseline --dpor-order rounds > $OUT/${TAG}_pmc_${w}_fetch.log 2>&1
  timeout 300 rocprofv3 $PRE --pmc WRITE_SIZE -d $P/${w}_write -o c -- python $R/bench.py --workload $w --no-cpu-baseline --dpor-order rounds > $OUT/${TAG}_pmc_${w}_write.log 2>&1
done
python $R/tools/summarize_prof.py ${TAG}x $P $OUT > /dev/null 2>&1
mv $OUT/${TAG}x_ddmin.txt $OUT/${TAG}_ddmin.txt 2>/dev/null; mv $OUT/${TAG}x_dpor.txt $OUT/${TAG}_dpor.txt 2>/dev/null; rm -f $OUT/${TAG}x_k1.txt
python - <<'PY'
import glob, json, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
TAG = os.environ.get("TAG", "r06")
def db(d):
    dbs = glob.glob("/tmp/prof23/%s/*.db" % d)
    return sqlite3.connect(dbs[0]).cursor() if dbs else None
cur = db("prof_stats_config5")
if cur:
    lines = ["# python bench.py --workload config5 --no-cpu-baseline: rocprofv3 --kernel-trace --stats (durations in ns)",
             "%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct")]
    for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-90s %8d %14.0f %12.0f %7.2f" % (r[0][:90], r[1], r[2] * 1000, r[3] * 1000, r[4]))
    open(os.path.join(out, TAG + "_config5.txt"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
# the instruction counters in the form bench.py's issue model reads (<tag>_<workload>_insts.json): totals per exploration (the DPOR
# workloads: the traced bench runs the exploration twice) / per 2^20-candidate launch (K2: its six largest dispatches), with what
# identifies the run they belong to (the bench line of the traced process: digest, interleavings, kernel_ms)
def bench_line(w, suffix=""):
    try:
        for line in open(os.path.join(out, "%s_pmc_%s%s.log" % (TAG, w, suffix))):
            if line.startswith('{"metric"'):
                return json.loads(line)
    except (OSError, ValueError):
        pass
    return None
for w in ("ddmin", "dpor", "config5"):
    tot = {}
    for d in (w, w + "_sq2"):
        cur = db(d)
        if not cur:
            continue
        if w == "ddmin":
            # per counter: the six largest dispatches of the throughput kernel
            for (cn,) in list(cur.execute("select distinct counter_name from counters_collection where kernel_name like '%k2_replay%'")):
                vals = [r[0] for r in cur.execute("select sum(value) from counters_collection where kernel_name like '%k2_replay%' and counter_name = ? "
                                                  "group by dispatch_id order by sum(value) desc limit 6", (cn,))]
                if vals:
                    tot[cn] = sum(vals) / len(vals)
        else:
            for cn, v in cur.execute("select counter_name, sum(value) from counters_collection where kernel_name like '%demi%' group by counter_name"):
                tot[cn] = v / 2.0
    line = bench_line(w)
    if tot and line:
        rec = {"counters": tot, "unit": "per 2^20-candidate launch (mean of the six largest k2_replay dispatches)" if w == "ddmin" else "per exploration (all demi kernels; the traced run explores twice)",
               "traced_run": {"kernel_ms": line.get("roofline", {}).get("kernel_ms"), "value": line.get("value"),
                              "sequence_digest": (line.get("orders", {}).get("rounds", {}) or line).get("sequence_digest"),
                              "interleavings": (line.get("orders", {}).get("rounds", {}) or line).get("interleavings"),
                              "still_violating": line.get("still_violating")},
               "source": "rocprofv3 --pmc (two passes) over python bench.py --workload %s --no-cpu-baseline --dpor-order rounds" % w}
        json.dump(rec, open(os.path.join(out, "%s_%s_insts.json" % (TAG, w)), "w"), indent=1)
        print(w, "instruction counters:", {k: "%.4g" % v for k, v in tot.items()})
for w in ("ddmin", "dpor"):
    cur = db(w)
    if not cur:
        print(w, "no database"); continue
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
    open(os.path.join(out, "%s_%s_counters.txt" % (TAG, w)), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:40]))
# fabric bytes: per exploration for the DPOR workloads (bench.py runs the exploration twice: one untimed, one timed), per
# 2^20-candidate launch for K2 (the six largest dispatches of the throughput kernel: 1 warm-up + 5 timed)
for w, runs in (("dpor", 2), ("config5", 2)):
    tot = {}
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        cur = db("%s_%s" % (w, "fetch" if cn == "FETCH_SIZE" else "write"))
        if cur:
            tot[cn] = list(cur.execute("select sum(value) from counters_collection where kernel_name like '%demi%' and counter_name = ?", (cn,)))[0][0]
    if len(tot) == 2 and all(v is not None for v in tot.values()):
        fabric = (tot["FETCH_SIZE"] * 2048.0 + tot["WRITE_SIZE"] * 1024.0) / runs
        json.dump({"fabric_bytes_per_exploration": fabric, "FETCH_SIZE_total": tot["FETCH_SIZE"], "WRITE_SIZE_total": tot["WRITE_SIZE"],
                   "explorations_profiled": runs, "bytes_per_unit": {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0},
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over python bench.py --workload %s%s, all demi kernels" % (w, " --dpor-order rounds" if w == "dpor" else "")},
                  open(os.path.join(out, "%s_%s_counters.json" % (TAG, w)), "w"), indent=1)
        print("%s fabric bytes per exploration: %.3e" % (w, fabric))
big = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    cur = db("ddmin_%s" % ("fetch" if cn == "FETCH_SIZE" else "write"))
    if cur:
        vals = [r[0] for r in cur.execute("select sum(value) from counters_collection where kernel_name like '%k2_replay%' and counter_name = ? "
                                          "group by dispatch_id order by sum(value) desc limit 6", (cn,))]
        if vals:
            big[cn] = vals
if len(big) == 2:
    f = sum(big["FETCH_SIZE"]) / len(big["FETCH_SIZE"]); wv = sum(big["WRITE_SIZE"]) / len(big["WRITE_SIZE"])
    json.dump({"fabric_bytes_per_launch": f * 2048.0 + wv * 1024.0, "FETCH_SIZE_per_launch": f, "WRITE_SIZE_per_launch": wv,
               "launches_averaged": len(big["FETCH_SIZE"]), "bytes_per_unit": {"FETCH_SIZE": 2048.0, "WRITE_SIZE": 1024.0},
               "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over python bench.py --workload ddmin: the six largest K2 dispatches (the 2^20-candidate launches)"},
              open(os.path.join(out, TAG + "_ddmin_counters.json"), "w"), indent=1)
    print("ddmin fabric bytes per 2^20-candidate launch: %.3e" % (f * 2048.0 + wv * 1024.0))
PY
