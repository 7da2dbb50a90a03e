#!/bin/bash
# Round 6, call 4: the submit / wait pipeline with two calls submitted ahead (tools/history/r6_pipeline_ab.py), the K1 tests around it,
# the driver's bench line with the issue models of the K2 / K3 records.
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call4_build.log 2>&1
timeout 600 python tools/history/r6_pipeline_ab.py > gpurun_out/r06_call4_pipeline_ab.txt 2>&1
cat gpurun_out/r06_call4_pipeline_ab.txt
timeout 900 python -m pytest tests/test_k1_gpu.py tests/test_srcdst_fifo_gpu.py -m gpu -x -q > gpurun_out/r06_call4_tests.log 2>&1
tail -3 gpurun_out/r06_call4_tests.log
timeout 900 python bench.py > gpurun_out/r06_bench_call4.json 2> gpurun_out/r06_bench_call4.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_call4.json"))
print("value %.4g ms_per_step %.3f" % (d["value"], d["ms_per_step"]))
print("pcie", json.dumps(d.get("pcie_inclusive"))[:1200])
for k, v in d.get("secondary", {}).items():
    print(k, v.get("value"), v.get("error"), json.dumps((v.get("roofline") or {}).get("issue_model"))[:700])
PY
