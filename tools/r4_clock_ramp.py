"""Do the short secondary workloads (config 3's 17 ms exploration, config 4's 1.4 ms DDMin) run at the clock a long job sees?
Repeats each back to back and prints how the time per call moves while the shader clock ramps (DVFS, about a second of load)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config3, raft5_config4

model, ev, depth = raft5_config3()
par = T.DporParams(depth, 0, 0, 0, 64, 4096)
for order in (T.DPOR_ORDER_ROUNDS, T.DPOR_ORDER_REFERENCE):
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct()); ctx.model_specialize(); ctx.dpor_load(ev)
    srch = T.DporSearch(16384, 1 << 17, 0, 1, order)
    ctx.dpor_explore(par, srch)
    time.sleep(2.0)                      # an idle device
    t0 = time.perf_counter(); out = []
    for i in range(120 if order == T.DPOR_ORDER_ROUNDS else 40):
        t = time.perf_counter(); v = ctx.dpor_explore(par, srch)[0]; dt = time.perf_counter() - t
        out.append((time.perf_counter() - t0, dt, len(v)))
    print("dpor order", order, " ".join("%.2fs:%.2fms" % (a, 1e3 * b) for a, b, _ in out[::8]), " best %.2f ms = %.4g/s" % (1e3 * min(b for _, b, _ in out), out[0][2] / min(b for _, b, _ in out)), flush=True)
    ctx.close()

model, events, lim = raft5_config4()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
ctx.replay_load(used, rec)
p = T.DdminParams(0, 256, 1, 1)
ctx.ddmin(target, p)
time.sleep(2.0)
t0 = time.perf_counter(); out = []
for k in range(3000):
    t = time.perf_counter(); ctx.ddmin(target, p); dt = time.perf_counter() - t
    out.append((time.perf_counter() - t0, dt))
print("ddmin e2e", " ".join("%.2fs:%.3fms" % (a, 1e3 * b) for a, b in out[::200]), " best %.3f ms" % (1e3 * min(b for _, b in out)), flush=True)
ctx.close()
