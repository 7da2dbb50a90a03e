#!/bin/bash
# Round 6, call 11: the tree with external Sends' payload areas, pruneConcurrentEvents on big tables - the whole GPU suite, smoke,
# the driver's bench line.
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call11_build.log 2>&1
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r06_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
tail -1 gpurun_out/r06_smoke.log
timeout 900 python bench.py > gpurun_out/r06_bench_1gpu.json 2> gpurun_out/r06_bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_1gpu.json"))
print("value %.4g ms_per_step %.3f kernel_ms %.3f alone %.3f frac %.3g traffic %s stale %s id %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["counters_stale"], d["roofline"].get("kernel_code_id")))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("error"))
PY
