#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# per-phase s_memtime split of k2_replay's lock-step loop on the native DDMin of config 4 (diagnostic build; proportions only)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/k2ph.py <<'PY'
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
print(ctx.ddmin(target, T.DdminParams(0, 1024, 1, 1))[2:])
PY
DEMI_K2_PHASES=1 DEMI_JIT_DEFINES="DEMI_K2_PHASES=1" DEMI_K2_VERBOSE=1 timeout 300 python /tmp/k2ph.py 2>&1 | grep -E "k2 phases|k2 launch|\(\[" | tail -8
