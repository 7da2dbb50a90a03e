#!/bin/bash
# config 5 in ROUNDS of $W: the insert kernel's variants (knobs), each as the sweep tool's line
for W in 16384 65536; do
echo "== default, width $W"; python tools/r6_batch_sweep.py $W 2>/dev/null | tail -1
for K in "$@"; do
echo "== $K, width $W"; env DEMI_EXPERIMENT=1 $K python tools/r6_batch_sweep.py $W 2>/dev/null | tail -1
done
done
