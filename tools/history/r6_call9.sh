#!/bin/bash
# Round 6, call 9: the BIG layout (tables of 9..16 actors) on the device for the first time - its GPU tests first, then the whole GPU
# suite, smoke, the driver's bench line, the big-table record, and K1's profile re-taken (the 8-actor K1 compiles to the same 2 804
# instructions under another register allocation: a new code id).
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call9_build.log 2>&1
timeout 900 python -m pytest tests/test_big_gpu.py -m gpu -x -q > gpurun_out/r06_big_tests.log 2>&1
tail -3 gpurun_out/r06_big_tests.log
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r06_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
tail -1 gpurun_out/r06_smoke.log
timeout 600 python bench.py --workload big > gpurun_out/r06_bench_big_tables.json 2> gpurun_out/r06_bench_big_tables.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_big_tables.json"))
for k, v in d["workloads"].items():
    print(k, "%.4g %s" % (v["value"], v["unit"]), v.get("ms_per_step"), v.get("violations"), v.get("capacity_aborts"), v.get("cpu_baseline"))
PY
timeout 1500 bash tools/profile_r6.sh > gpurun_out/r06_profile_k1.log 2>&1
tail -3 gpurun_out/r06_profile_k1.log
head -12 gpurun_out/r06_k1.txt
cp gpurun_out/k1_counters.json profiles/k1_counters.json 2>/dev/null
timeout 900 python bench.py > gpurun_out/r06_bench_1gpu.json 2> gpurun_out/r06_bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_1gpu.json"))
print("value %.4g ms_per_step %.3f kernel_ms %.3f alone %.3f frac %.3g traffic %s stale %s id %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["counters_stale"], d["roofline"].get("kernel_code_id")))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("error"))
PY
