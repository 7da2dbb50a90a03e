#!/bin/bash
# randomDDMin (config 4, R = 100): lanes per wave of its frontier launches
for K in "DEMI_K1_NO_SPREAD=1" "DEMI_K1_LANES_PER_WAVE=1" "DEMI_K1_LANES_PER_WAVE=2" "DEMI_K1_LANES_PER_WAVE=4" "DEMI_K1_LANES_PER_WAVE=8" "DEMI_K1_LANES_PER_WAVE=16" "DEMI_K1_VERBOSE=0"; do
  echo -n "== $K  "
  env DEMI_EXPERIMENT=1 $K timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('random_ddmin_R100',{})
print(r.get('seconds'), r.get('executions_per_s'), r.get('consulted_digest'), r.get('mcs_len'))"
done
