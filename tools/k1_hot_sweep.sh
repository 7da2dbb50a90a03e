#!/bin/bash
# Experiment: resident workgroups per CU x LDS-resident slots of the specialised K1: bench line + HBM write traffic
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for cfg in "0 6" "0 5" "0 4" "0 3" "4 5"; do
  set -- $cfg
  export DEMI_JIT_K1_HOT=$1 DEMI_K1_MAX_WG_PER_CU=$2
  timeout 120 python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hot $1 wg/cu $2', d['roofline']['kernel_ms'], d['value'])"
  rm -rf /tmp/pw; timeout 120 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE -d /tmp/pw -o k1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - <<'PY'
import sqlite3, glob
db = glob.glob('/tmp/pw/*.db')
if db:
    cur = sqlite3.connect(db[0]).cursor()
    for r in cur.execute("select counter_name, avg(value) from counters_collection where kernel_name like '%k1_random%' group by counter_name"):
        print('   ', r[0], round(r[1] / 1024.0, 1), 'MB per launch')
PY
done
