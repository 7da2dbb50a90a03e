import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config4
from demi_amd.minification import stsSchedDDMin
from demi_amd.schedulers import EventTrace, STSScheduler, SchedulerConfig, ViolationFingerprint
model, events, lim = raft5_config4()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events)
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.close()
fp = ViolationFingerprint(vv.fingerprint)
for spec in (False, True):
    for depth in (4, 6, 8):
        sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used), p_max=128, specialize=spec)
        stsSchedDDMin(sts, used, fp, speculative_depth=depth)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            mcs, dd, _ = stsSchedDDMin(sts, used, fp, speculative_depth=depth)
            best = min(best, time.perf_counter() - t)
        print("specialize", spec, "depth", depth, "e2e ms %.2f" % (best * 1e3), "launches", len(dd.batches), "consultations", len(dd.consulted), "replays", int(dd.speculative_replays), "mcs", len(mcs), flush=True)
        sts.shutdown()
