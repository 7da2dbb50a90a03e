#!/bin/bash
# K3 parity with the one-probe host map + one-shard commit book, and the dpor record with the loop's timing split
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_k3_gpu.py tests/test_comm_gpu.py -x -q --timeout 600 2>&1 | grep -E "passed|failed|error" | tail -3
DEMI_DPOR_TIMING=1 timeout 600 python bench.py --workload dpor > gpurun_out/r2_q_dpor.json 2> gpurun_out/r2_q_dpor.err
grep -i "dpor\|timing" gpurun_out/r2_q_dpor.err | tail -8
python - <<'PY'
import json
d = json.load(open("gpurun_out/r2_q_dpor.json"))
for k, v in d["orders"].items():
    print(k, "%.4g /s" % v["value"], "%.3f s" % v["seconds"], v["interleavings"], "launches", v["launches"], "kernel_ms", round(v["kernel_ms_total"], 1))
for k, v in d["cpu_baseline"]["orders"].items():
    print("cpu", k, "%.4g /s" % v["value"], "%.3f s" % v["seconds"])
PY
