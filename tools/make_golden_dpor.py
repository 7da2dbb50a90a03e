#!/usr/bin/env python
"""Golden record of BASELINE config 3 explored in DPORwHeuristics' own order: the CPU oracle at batch = 1 (one backtrack
point per step, the reference's loop, ~1 minute) -> tests/golden/dpor_config3_reference_order.json.  The GPU suite checks
that the REFERENCE order of demi_dpor_explore (device speculation + sequential commit) reproduces this sequence."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import raft5_config3  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

model, ev, depth = raft5_config3()
par = T.DporParams(depth, 0, 0, 0, 64, 4096)
v, plen, rounds, vt, st, secs = O.dpor_explore(model, ev, par, T.DporSearch(1, 1 << 17, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
viol = v[(v["flags"] & T.V_VIOLATION) != 0]
rec = {"generator": "tools/make_golden_dpor.py (oracle, batch = 1)", "interleavings": int(len(v)), "exhausted": bool(st.exhausted),
       "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(), "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
       "violations": int(len(viol)), "distinct_schedules": int(len(set(v["hash"].tolist())))}
with open(os.path.join(ROOT, "tests", "golden", "dpor_config3_reference_order.json"), "w") as f:
    json.dump(rec, f, indent=1)
print(rec)
