#!/bin/bash
# Round 6, call 6: do the library's two streams share a hardware queue with each other when torch's streams took the others first?
# The same A/B with GPU_MAX_HW_QUEUES=8, and a process that never creates a torch stream (submit / wait only).
mkdir -p gpurun_out
python __graft_entry__.py > gpurun_out/r06_call6_build.log 2>&1
echo "== GPU_MAX_HW_QUEUES=8" > gpurun_out/r06_call6_pipeline_ab.txt
GPU_MAX_HW_QUEUES=8 timeout 900 python tools/history/r6_pipeline_ab.py >> gpurun_out/r06_call6_pipeline_ab.txt 2>&1
echo "== no torch streams in the process (submit / wait only)" >> gpurun_out/r06_call6_pipeline_ab.txt
timeout 600 python - >> gpurun_out/r06_call6_pipeline_ab.txt 2>&1 <<'PY'
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
        tk = [ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + 1) * n) for j in range(min(ahead, k))]
        for j in range(k):
            if j + ahead < k:
                tk.append(ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + ahead + 1) * n))
            ctx.random_explore_wait(tk[j], out=hv[j & 1] if with_verdicts else None)
    return f
res = {}
for name, fn in (("flagged_sync_call", lambda k: [ctx.random_explore_flagged(n, limits, T.V_VIOLATION, seed_base=SEED_BASE + (i + 1) * n) for i in range(k)]),
                 ("submit_wait_flagged_1_ahead", piped(False, 1)), ("submit_wait_flagged_2_ahead", piped(False, 2)),
                 ("submit_wait_verdicts_1_ahead", piped(True, 1)), ("submit_wait_verdicts_2_ahead", piped(True, 2))):
    fn(4)
    t = time.perf_counter(); fn(K); res[name] = (time.perf_counter() - t) / K * 1e3
print(json.dumps({"ms_per_2^20_schedules": res}, indent=1))
PY
cat gpurun_out/r06_call6_pipeline_ab.txt
