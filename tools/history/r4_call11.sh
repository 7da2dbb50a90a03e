#!/bin/bash
# Round 4, call 11: is the spill part of K3's latency chain?  k3_dpor launch durations by size (rocprofv3 kernel trace of config 3,
# ROUNDS order) for several numbers of LDS-resident pending slots (DEMI_JIT_K3_HOT; the rest of a pending set is in HBM scratch).
export DEMI_EXPERIMENT=1
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
for hot in 12 24 40 64; do
  P=/tmp/prof_hot_$hot; rm -rf $P
  DEMI_JIT_K3_HOT=$hot DEMI_K3_VERBOSE=1 timeout 300 rocprofv3 --preload $COMGR --kernel-trace --stats -d $P -o k3 -- python $R/bench.py --workload dpor --no-cpu-baseline --dpor-order rounds > $OUT/r04_k3hot_$hot.log 2>&1
  python - <<PY
import glob, sqlite3, re
dbs = glob.glob("$P/*.db")
print("== hot $hot", sorted(set(re.findall(r"per_cu=\d+ lds=\d+ hot=\d+ waves=\d+", open("$OUT/r04_k3hot_$hot.log").read()))))
if dbs:
    cur = sqlite3.connect(dbs[0]).cursor()
    rows = list(cur.execute("select name, grid_x, count(*), avg(duration), min(duration) from kernels where name like '%k3_%' group by name, grid_x order by name, grid_x"))
    for r in rows:
        print("   %-28s grid %7d  launches %3d  avg %9.0f ns  min %9.0f ns" % (r[0].split('(')[0].replace('demi::', ''), r[1], r[2], r[3], r[4]))
import json
for l in open("$OUT/r04_k3hot_$hot.log"):
    if l.startswith('{"metric"'):
        o = json.loads(l)["orders"]["rounds"]; print("   rounds %.4g/s  %.2f ms  kernels %.2f ms" % (o["value"], 1e3 * o["seconds"], o["kernel_ms_total"]))
PY
done 2>&1 | tee $OUT/r04_k3_hot_sweep.txt
