#!/bin/bash
# Experiment: LDS-resident pending slots x requested waves per SIMD of the specialised K1 (bench line per setting)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "0 0" "0 6" "2 0" "2 6" "5 6"; do
  set -- $cfg
  export DEMI_JIT_K1_HOT=$1
  if [ "$2" != "0" ]; then export DEMI_JIT_K1_WAVES_PER_EU=$2; else unset DEMI_JIT_K1_WAVES_PER_EU; fi
  timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hot $1 waves_per_eu $2', d['roofline']['kernel_ms'], d['value'])"
done
