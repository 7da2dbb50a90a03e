#!/bin/bash
# Round 5, call 11: what bounds k3_pairs_insert on config 5 - two SQ counter passes (each its own run, with a timeout).
export TMPDIR=/tmp DEMI_EXPERIMENT=1
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
COMGR=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib', 'libamd_comgr.so'))")
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_REQ_sum" ; do
  i=$((i+1))
  rm -rf /tmp/pm_$i
  timeout 240 rocprofv3 --preload $COMGR --pmc $ctrs -d /tmp/pm_$i -o c -- python $R/bench.py --workload config5 --no-cpu-baseline > /tmp/pm_$i.log 2>&1
  python - <<PY
import glob, sqlite3
dbs = glob.glob("/tmp/pm_$i/*.db") + glob.glob("/tmp/pm_$i/*/*.db")
if not dbs:
    print("pass $i: no database"); print(open("/tmp/pm_$i.log").read()[-600:])
else:
    cur = sqlite3.connect(dbs[0]).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%k3_pairs_insert%' or kernel_name like '%k3_pairs_decide_q%' or kernel_name like '%k3_analyze%' group by kernel_name, counter_name")
    for kn, cn, cnt, avg in cur.execute(q):
        print("%-28s %-24s %6d dispatches  avg %16.1f" % (kn.split('(')[0][-28:], cn, cnt, avg))
PY
done
