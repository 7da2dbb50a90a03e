"""Round 4: config 4's DDMin end to end (demi_ddmin: RunnerUtils.stsSchedDDMin in one library call) by counter mode of K2
(DEMI_K2_MODE: wave = one candidate per wave with the cooperative look-ahead, lds = lock step with LDS counter planes, hbm) and by
speculation budget (candidates per launch): best of 7, the MCS and the consultation count each time."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["DEMI_EXPERIMENT"] = "1"
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config4
model, events, lim = raft5_config4()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events)
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.model_specialize()
ctx.replay_load(used, rec)
target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
ref = None
for mode in ("lds", "wave", "hbm"):
    os.environ["DEMI_K2_MODE"] = mode
    for budget in (256, 1024, 4096, 16384, 65536):
        par = T.DdminParams(0, budget, 1, 1)
        r = ctx.ddmin(target, par)
        best = 1e9
        for _ in range(7):
            t = time.perf_counter(); r = ctx.ddmin(target, par); best = min(best, time.perf_counter() - t)
        mcs, st = r[0], r[-1]
        if ref is None: ref = list(mcs)
        print("mode %-4s budget %6d: e2e %.3f ms, launches %d, replays %d, consultations %d, mcs %d %s" % (
            mode, budget, best * 1e3, st.launches, st.replays, st.consultations, len(mcs), "" if list(mcs) == ref else "!!! MCS differs"), flush=True)
