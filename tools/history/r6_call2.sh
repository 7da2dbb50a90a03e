#!/bin/bash
# Round 6, call 2: the library's own overlap (two K1 launches in flight in ONE demi_ctx; demi_random_explore_submit / _wait):
# the whole GPU suite, the driver's bench line, the one-stream line beside it, then K1's profile (tools/profile_r6.sh).
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call2_build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests_call2.log 2>&1
tail -5 gpurun_out/r06_gpu_tests_call2.log
timeout 900 python bench.py > gpurun_out/r06_bench_call2.json 2> gpurun_out/r06_bench_call2.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_call2.json"))
print("value %.4g ms_per_step %.3f kernel_ms %.3f alone %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d.get("one_launch_at_a_time", {}).get("ms_per_step")))
print("pcie", json.dumps(d.get("pcie_inclusive"))[:900])
print("code id", d["roofline"].get("kernel_code_id"), "stale", d["roofline"].get("counters_stale"), "cpu same", d.get("cpu_baseline", {}).get("bit_identical_to_gpu"))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("error"))
PY
timeout 300 python bench.py --launches-in-flight 1 --no-secondary --no-cpu-baseline > gpurun_out/r06_bench_one_stream_call2.json 2>> gpurun_out/r06_bench_call2.err
python -c "import json; d=json.load(open('gpurun_out/r06_bench_one_stream_call2.json')); print('one stream: value %.4g ms_per_step %.3f' % (d['value'], d['ms_per_step']))"
timeout 1500 bash tools/profile_r6.sh > gpurun_out/r06_profile_k1.log 2>&1
tail -3 gpurun_out/r06_profile_k1.log
head -30 gpurun_out/r06_k1.txt
