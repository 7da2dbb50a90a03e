#!/bin/bash
export DEMI_EXPERIMENT=1     # the library reads its experiment / diagnostic variables only with this set (csrc/knobs.hpp)
# Where the native DPOR loop's wall time goes (BASELINE config 3), for two round sizes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
DEMI_DPOR_TIMING=1 python - <<'PY'
import time
from demi_amd.apps import raft5_config3
from demi_amd.dpor import DPORwHeuristics
from demi_amd.schedulers import SchedulerConfig
model, ev, depth = raft5_config3()
for nb in (2048, 16384):
    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=nb)
    d.explore_native(ev, max_interleavings=64)
    t = time.perf_counter(); r = d.explore_native(ev, max_interleavings=1 << 17); print(nb, len(r.interleavings), round(time.perf_counter() - t, 3))
    d.shutdown()
PY
