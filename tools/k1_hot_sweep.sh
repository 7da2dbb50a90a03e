#!/bin/bash
# LDS-resident pending slots of the specialised K1: parity at the extremes, then the bench line for each setting
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for h in 8 16; do
  DEMI_JIT_K1_HOT=$h timeout 200 python -m pytest tests/test_k1_gpu.py -x -q --timeout 90 -k "raft5_parity_all_capacities or srcdst_fifo_parity_raft5 or fault_heavy" 2>&1 | tail -1
done
for h in 8 12 16 20 24; do
  DEMI_JIT_K1_HOT=$h timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hot $h', d['roofline']['kernel_ms'], d['value'])"
done
DEMI_JIT_K1_HOT=16 timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --strategy fifo 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fifo hot 16', d['roofline']['kernel_ms'], d['value'])"
