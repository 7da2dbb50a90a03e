#!/bin/bash
# Round 6, call 10: K1's actor states in an HBM scratch (DEMI_ST_HBM) against LDS, A/B on the tables it is meant for: the big
# tables, the wide raft5, the raft with a real log of 15 entries.  GPU tests of those tables first (states in HBM by default).
mkdir -p gpurun_out
export DEMI_EXPERIMENT=1
timeout 1200 python -m pytest tests/test_big_gpu.py tests/test_wide_gpu.py tests/test_zz_array_gpu.py tests/test_payloads_gpu.py -m gpu -x -q > gpurun_out/r06_call10_tests.log 2>&1
tail -2 gpurun_out/r06_call10_tests.log
for hbm in 0 1; do
  DEMI_JIT_K1_ST_HBM=$hbm timeout 600 python bench.py --workload big --no-cpu-baseline > gpurun_out/r06_big_sthbm$hbm.json 2> gpurun_out/r06_big_sthbm$hbm.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r06_big_sthbm$hbm.json"))
print("st_hbm=$hbm", {k: (round(v["value"]), v.get("ms_per_step")) for k, v in d["workloads"].items()})
PY
  for v in "--wide-term0 1000" "--log-cap 15" "--log-cap 15 --real-fields"; do
    tag=$(echo $v | tr -d ' -')
    DEMI_JIT_K1_ST_HBM=$hbm timeout 300 python bench.py $v --no-secondary --no-cpu-baseline > gpurun_out/r06_sthbm${hbm}_$tag.json 2>> gpurun_out/r06_big_sthbm$hbm.err
    python -c "
import json; d=json.load(open('gpurun_out/r06_sthbm${hbm}_$tag.json')); print('st_hbm=$hbm', '$v', 'value %.4g ms_per_step %.3f alone %s' % (d['value'], d['ms_per_step'], d['roofline'].get('kernel_ms_alone')))"
  done
done
