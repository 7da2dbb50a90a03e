#!/bin/bash
# Round 6, call 7 (the evidence run): the whole GPU suite, smoke, the driver's bench line, the submit / wait pipeline in a process
# without torch streams (what a JVM host is).
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call7_build.log 2>&1
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gpu_tests.log 2>&1
grep -E "passed|failed|error" gpurun_out/r06_gpu_tests.log | tail -3
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
tail -1 gpurun_out/r06_smoke.log
timeout 900 python bench.py > gpurun_out/r06_bench_1gpu.json 2> gpurun_out/r06_bench_1gpu.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_bench_1gpu.json"))
print("value %.4g ms_per_step %.3f kernel_ms %.3f alone %.3f frac %.3g traffic %s stale %s" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["kernel_ms_alone"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["counters_stale"]))
print("one launch at a time", d.get("one_launch_at_a_time", {}).get("value"), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["bit_identical_to_gpu"])
print("pcie", json.dumps(d.get("pcie_inclusive"))[:1400])
s = d["secondary"]
print("dpor", s["dpor"].get("value"), s["dpor"].get("violations"), {k: (v["value"], v["seconds"]) for k, v in s["dpor"]["orders"].items()}, s["dpor"].get("round5_workload"))
print("config5", s["config5"].get("value"), s["config5"].get("violations"), s["config5"].get("reference_order", {}).get("value"), s["config5"].get("reference_order", {}).get("seconds"))
print("ddmin", s["ddmin"].get("value"), s["ddmin"]["ddmin_end_to_end"]["seconds"], json.dumps(s["ddmin"].get("random_ddmin_R100"))[:600])
print("config1", s["config1"].get("value"))
PY
timeout 600 python - > gpurun_out/r06_pipeline_no_torch.txt 2>&1 <<'PY'
import sys, time, json
sys.path.insert(0, ".")
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
model, events, limits = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
n, K = 1 << 20, 40
hv = [np.ones(n, dtype=T.VERDICT_DTYPE) for _ in range(2)]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    ctx.random_explore_flagged(n, limits, T.V_VIOLATION, seed_base=SEED_BASE)
def piped(with_verdicts, ahead):
    def f(k):
        tk = [ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + 1) * n, want_verdicts=with_verdicts) for j in range(min(ahead, k))]
        for j in range(k):
            if j + ahead < k:
                tk.append(ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + ahead + 1) * n, want_verdicts=with_verdicts))
            ctx.random_explore_wait(tk[j], out=hv[j & 1] if with_verdicts else None)
    return f
def sync_verdicts(k):
    for i in range(k):
        ctx.random_explore(n, limits, seed_base=SEED_BASE + (i + 1) * n)
res = {}
for name, fn in (("flagged_sync_call", lambda k: [ctx.random_explore_flagged(n, limits, T.V_VIOLATION, seed_base=SEED_BASE + (i + 1) * n) for i in range(k)]),
                 ("submit_wait_flagged_1_ahead", piped(False, 1)), ("submit_wait_flagged_2_ahead", piped(False, 2)),
                 ("submit_wait_verdicts_1_ahead", piped(True, 1)), ("submit_wait_verdicts_2_ahead", piped(True, 2))):
    fn(4)
    t = time.perf_counter(); fn(K); res[name] = (time.perf_counter() - t) / K * 1e3
print("# a process that never creates a torch stream (what a JVM host is): ms per 2^20 schedules of config 2's step, 40 steps each")
print(json.dumps({"ms_per_2^20_schedules": res, "schedules_per_s": {k: n / (v * 1e-3) for k, v in res.items()}}, indent=1))
PY
cat gpurun_out/r06_pipeline_no_torch.txt
