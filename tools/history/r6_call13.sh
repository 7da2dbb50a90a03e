#!/bin/bash
# Round 6, call 13 (run twice: after the K3 round work, and again - the evidence run - after the reference order's launch reordering, the pinned staging and the callers' output arrays):
# ROUNDS of 65 536 (config 5) / 32 768 (config 3): GPU suite, smoke, the secondary records' profiles re-taken (new digests), the bench line.
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call13_build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r06_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
tail -1 gpurun_out/r06_smoke.log
timeout 2400 bash tools/profile_r6_k2k3.sh > gpurun_out/r06_profile_k2k3.log 2>&1
tail -3 gpurun_out/r06_profile_k2k3.log
for w in dpor config5 ddmin; do cp gpurun_out/r06_${w}_insts.json profiles/ 2>/dev/null; cp gpurun_out/r06_${w}_counters.json profiles/ 2>/dev/null; done
timeout 900 python bench.py > gpurun_out/r06_bench_1gpu.json 2> gpurun_out/r06_bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_1gpu.json"))
print("value %.4g ms_per_step %.3f kernel_ms %.3f alone %.3f frac %.3g traffic %s stale %s id %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["counters_stale"], d["roofline"].get("kernel_code_id")))
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("seconds"), v.get("error"), (v.get("roofline") or {}).get("issue_model") is not None)
PY
