#!/bin/bash
# reference order on config 3: fewest lanes per wave a small launch may use (DEMI_K3_MIN_LANES; the default was 4)
for L in 4 2 1; do
  echo "== DEMI_K3_MIN_LANES=$L"
  DEMI_EXPERIMENT=1 DEMI_K3_MIN_LANES=$L timeout 300 python bench.py --workload dpor --dpor-order reference_order --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['orders']['reference_order']
print(r['value'], r['seconds'], r['launches'], r['kernel_ms_total'], r['sequence_digest'])"
done
