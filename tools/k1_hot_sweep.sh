#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Experiment: resident workgroups per CU x LDS-resident slots of the specialised K1 (bench line per setting)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for cfg in "0 6" "0 5" "0 4" "0 3" "4 5"; do
  set -- $cfg
  export DEMI_JIT_K1_HOT=$1 DEMI_K1_MAX_WG_PER_CU=$2
  timeout 120 python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hot $1 wg/cu $2', d['roofline']['kernel_ms'], d['value'])"
done
