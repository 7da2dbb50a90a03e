#!/bin/bash
# Round 5, call 21: bench.py with two launches in flight (two contexts, a stream each) - the fuzz line, the kernel's duration by
# events against rocprofv3's, and the two bench tests that run it with two ranks.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
export TMPDIR=/tmp
for L in 2 1; do
timeout 300 python bench.py --no-secondary --no-cpu-baseline --launches-in-flight $L 2> gpurun_out/b21.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('in flight $L: %.4g/s %.3f ms per step; kernel_ms %.3f alone %.3f; one at a time %s; stale=%s issue=%s' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['kernel_ms_alone'], (d.get('one_launch_at_a_time') or {}).get('value'), r['counters_stale'], (r.get('issue_model') or {}).get('issue_frac_straight_line')))" || tail -5 gpurun_out/b21.err
done
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
cd /tmp && timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats --output-format csv -d /tmp/p21 -o k1 -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $R/gpurun_out/p21.log 2>&1
cd $R && python - <<'PY'
import csv, glob
for f in glob.glob("/tmp/p21/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print("  %-70s calls %6s avg %9s ns" % (r["Name"][:70], r["Calls"], r["AverageNs"]))
PY
timeout 600 python -m pytest tests/test_comm_gpu.py -m gpu -x -q -k "bench_py" 2>&1 | tail -3
