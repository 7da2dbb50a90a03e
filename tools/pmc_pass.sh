#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# One rocprofv3 --pmc pass over the headline bench; prints the per-dispatch averages of the K1 kernel.
#   bash tools/pmc_pass.sh <tag> COUNTER [COUNTER ...]         (env is passed through: DEMI_JIT_DEFINES etc.)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
P=/tmp/pmc_$tag
rm -rf $P; mkdir -p $P $R/gpurun_out
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
cd /tmp
rocprofv3 --preload $COMGR --pmc "$@" -d $P -o k1 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary > $R/gpurun_out/pmc_$tag.log 2>&1
python - <<PY
import glob, sqlite3
dbs = glob.glob("$P/*.db") + glob.glob("$P/*/*.db")
if not dbs:
    print("$tag: no database (see gpurun_out/pmc_$tag.log)")
else:
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k1_random_explore%' group by counter_name")
    for cn, cnt, avg in cur.execute(q):
        print("$tag %-36s %18.1f  (%d dispatches)" % (cn, avg, cnt))
PY
