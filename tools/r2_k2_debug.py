import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.minification import events_to_mask
from oracle import oracle_py as O
model, events, lim = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events)
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.replay_load(used, rec)
target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
full = np.array([events_to_mask(range(len(used)))], dtype=np.uint64)
print("orig verdict", hex(vv.flags), vv.fingerprint, vv.hash)
print("replay_batch full mask :", ctx.replay_batch(full, target))
print("oracle                 :", O.sts_replay_batch(model, used, rec, full, target))
print("removal NO_SKIP        :", ctx.replay_removal_batch([0xFFFFFFFF], target))
print("removal NO_SKIP x3     :", ctx.replay_removal_batch([0xFFFFFFFF] * 3, target))
print("removal NO_SKIP + mask :", ctx.replay_removal_batch([0xFFFFFFFF], target, masks=full))
from demi_amd.internal_minimization import deliveries
from demi_amd.schedulers import EventTrace
dl = [i for i, _, _ in deliveries(EventTrace(rec, used))]
for skips in ([0xFFFFFFFF] * 61, dl[:3] + [0xFFFFFFFF], dl + [0xFFFFFFFF]):
    skips = np.array(skips, dtype=np.uint32)
    g = ctx.replay_removal_batch(skips, target)
    c = O.sts_removal_batch(model, used, rec, skips, target, n_threads=8)
    bad = np.nonzero(g != c)[0]
    print("n", len(skips), "differ", len(bad), "first gpu", g[bad[0]] if len(bad) else None, "cpu", c[bad[0]] if len(bad) else None, "last gpu", g[-1], "cpu", c[-1])
