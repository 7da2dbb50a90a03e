import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
from oracle import oracle_py as O
model, events, lim = raft5_config2()
for n in (2, 3, 4, 5, 8, 17, 61):
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct()); ctx.trace_load(events)
    v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    ctx.replay_load(used, rec)
    target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
    g = ctx.replay_removal_batch([0xFFFFFFFF] * n, target)
    print("n", n, "deliveries per lane", [int(x) >> 16 for x in g["flags"]])
    ctx.close()
