#!/bin/bash
# Round 4, call 15: the last K3 step (one residency test per produce / swap-remove / delivery), the K3 suite, the default bench line,
# the kernel-trace stats of config 3 and config 5.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
mkdir -p $OUT
timeout 600 python -m pytest tests/test_k3_gpu.py tests/test_blocked_actors_gpu.py tests/test_payloads_gpu.py tests/test_zz_array_gpu.py tests/test_wide_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 600 python bench.py > $OUT/r04_bench_final.json 2> $OUT/r04_bench_final.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
s = d['secondary']
print('fuzz %.4g/s %.3f ms kernel %.3f ms' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
print('dpor rounds %.4g/s (%.2f ms, kernels %.2f)  reference %.4g/s (%.2f ms, kernels %.2f)' % (s['dpor']['orders']['rounds']['value'], 1e3 * s['dpor']['orders']['rounds']['seconds'], s['dpor']['orders']['rounds']['kernel_ms_total'], s['dpor']['orders']['reference_order']['value'], 1e3 * s['dpor']['orders']['reference_order']['seconds'], s['dpor']['orders']['reference_order']['kernel_ms_total']))
print('ddmin %.4g replays/s  e2e %.3f ms' % (s['ddmin']['value'], 1e3 * s['ddmin']['ddmin_end_to_end']['seconds']))
print('config5 %.4g/s %.3f s kernels %.1f ms' % (s['config5']['value'], s['config5']['seconds'], s['config5']['kernel_ms_total']))
PY
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
P=/tmp/prof23; rm -rf $P; mkdir -p $P
timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d $P/prof_stats_dpor -o k3 -- python $R/bench.py --workload dpor --no-cpu-baseline > $OUT/r04_prof_stats_dpor.log 2>&1
timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d $P/prof_stats_config5 -o k3 -- python $R/bench.py --workload config5 --no-cpu-baseline > $OUT/r04_prof_stats_config5.log 2>&1
python $R/tools/summarize_prof.py r04x $P $OUT > /dev/null 2>&1
mv $OUT/r04x_dpor.txt $OUT/r04_dpor.txt 2>/dev/null; rm -f $OUT/r04x_k1.txt $OUT/r04x_ddmin.txt
python - <<'PY'
import glob, os, sqlite3
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out")
dbs = glob.glob("/tmp/prof23/prof_stats_config5/*.db")
if dbs:
    cur = sqlite3.connect(dbs[0]).cursor()
    lines = ["# python bench.py --workload config5 --no-cpu-baseline: rocprofv3 --kernel-trace --stats (durations in ns)",
             "%-90s %8s %14s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "pct")]
    for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("%-90s %8d %14.0f %12.0f %7.2f" % (r[0][:90], r[1], r[2] * 1000, r[3] * 1000, r[4]))
    open(os.path.join(out, "r04_config5.txt"), "w").write("\n".join(lines) + "\n")
PY
head -6 $OUT/r04_dpor.txt; head -6 $OUT/r04_config5.txt
