#!/bin/bash
# K1's SPREAD variant (few lanes per wave when a launch does not fill the chip) against the plain launch: config 1 (100 schedules per
# call) and config 4's randomDDMin (frontiers of candidates x 100 executions)
for K in "DEMI_K1_NO_SPREAD=1" "DEMI_K1_VERBOSE=0"; do
  echo "== $K"
  env DEMI_EXPERIMENT=1 $K python - <<'PY'
import json, sys, time
sys.path.insert(0, ".")
import bench
r = bench.bench_config1(0, cpu_baseline=False)
print("config1", r["value"], r["seconds_per_100_schedules"], r["violations"])
PY
  env DEMI_EXPERIMENT=1 $K timeout 600 python bench.py --workload ddmin --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('random_ddmin_R100',{})
print('random_ddmin', json.dumps({k:v for k,v in r.items() if not isinstance(v,(dict,list))})[:900])"
done
