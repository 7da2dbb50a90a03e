#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# the same phase split for a THROUGHPUT launch of K2: 2^20 random candidate masks of config 4's execution, 64 per wave
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/k2phw.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
rng = np.random.default_rng(0)
for n in (1 << 16, 1 << 20):
    keep = rng.random((n, len(used))) < 0.7
    masks = np.zeros((n, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    out = ctx.replay_batch(masks, target)
    print(n, int((out["flags"] & T.V_VIOLATION).sum()))
PY
DEMI_K2_PHASES=1 DEMI_JIT_DEFINES="DEMI_K2_PHASES=1" DEMI_K2_VERBOSE=1 timeout 300 python /tmp/k2phw.py 2>&1 | grep -E "k2 phases|k2 launch|^[0-9]" | tail -8
