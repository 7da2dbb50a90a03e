#!/bin/bash
# k3_pairs_insert cut short (DEMI_K3_INSERT_PROBE = 1 index build, 2 + filter, 3 + table loads) beside the real launch, config 5 in ROUNDS of $1
W=${1:-65536}
for s in 1 2 3; do
  DEMI_EXPERIMENT=1 DEMI_K3_INSERT_PROBE=$s python tools/r6_batch_sweep.py $W 2>&1 | grep "insert probe" | sed -n '10,18p;$p'
done
