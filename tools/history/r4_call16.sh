#!/bin/bash
# Round 4, call 16: the explored-pair entry with key + state in one 32-byte sector (config 5's pair kernels moved a 64-byte line per
# probe): config 3 / config 5 lines, the kernel traces, the K3 + comm tests.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
mkdir -p $OUT
for wl in dpor config5; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $OUT/r04_pe_$wl.json 2> $OUT/r04_pe_$wl.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r04_pe_$wl.json').read().strip().splitlines()[-1])
if '$wl' == 'dpor':
    o = d['orders']
    print('config3 rounds %.4g/s (%.2f ms, kernels %.2f ms, %s)  reference %.4g/s (%.2f ms, kernels %.2f ms, %s)' % (o['rounds']['value'], 1e3 * o['rounds']['seconds'], o['rounds']['kernel_ms_total'], o['rounds']['sequence_digest'][:8], o['reference_order']['value'], 1e3 * o['reference_order']['seconds'], o['reference_order']['kernel_ms_total'], o['reference_order']['sequence_digest'][:8]))
else:
    print('config5 %.4g/s (%.3f s, kernels %.1f ms, %s)' % (d['value'], d['seconds'], d['kernel_ms_total'], d['sequence_digest'][:8]))
PY
done
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
head -7 $OUT/r04_dpor.txt | tail -5; head -6 $OUT/r04_config5.txt | tail -4
cd $R
timeout 600 python -m pytest tests/test_k3_gpu.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -2
