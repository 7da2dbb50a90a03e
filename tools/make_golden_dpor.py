#!/usr/bin/env python
"""Golden records of BASELINE config 3 explored in DPORwHeuristics' own order: the CPU oracle at batch = 1 (one backtrack
point per step, the reference's loop).  The GPU suite checks that the REFERENCE order of demi_dpor_explore (device
speculation + sequential commit) reproduces these sequences.
  (no argument)  apps.raft5_config3, the workload of rounds 1-5 (no violating interleaving; ~1 minute)
                 -> tests/golden/dpor_config3_reference_order.json
  bug            apps.raft5_dpor_config3, the workload bench.py times from round 6 on (finds the seeded bug; ~25 minutes)
                 -> tests/golden/dpor_config3_bug_reference_order.json (also the SHA-256 of the violating executions' hashes,
                 sorted: the found-violation SET, which is what the ROUNDS order is compared with)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import raft5_config3, raft5_dpor_config3  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

bug = len(sys.argv) > 1 and sys.argv[1] == "bug"
if bug:
    model, ev, par = raft5_dpor_config3()
    out, cap = "dpor_config3_bug_reference_order.json", 1 << 20
else:
    model, ev, depth = raft5_config3()
    par = T.DporParams(depth, 0, 0, 0, 64, 4096)
    out, cap = "dpor_config3_reference_order.json", 1 << 17
t0 = time.perf_counter()
v, plen, rounds, vt, st, secs = O.dpor_explore(model, ev, par, T.DporSearch(1, cap, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
viol = v[(v["flags"] & T.V_VIOLATION) != 0]
rec = {"generator": "tools/make_golden_dpor.py%s (oracle, batch = 1, %.0f s)" % (" bug" if bug else "", time.perf_counter() - t0),
       "interleavings": int(len(v)), "exhausted": bool(st.exhausted),
       "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(), "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
       "violations": int(len(viol)), "distinct_schedules": int(len(set(v["hash"].tolist())))}
if bug:
    vh = np.unique(viol["hash"])
    rec.update({"model": model.name, "first_violation": int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0]) if len(viol) else None,
                "distinct_violating_schedules": int(len(vh)), "sha256_sorted_violating_hashes": hashlib.sha256(vh.tobytes()).hexdigest(),
                "sha256_first_65536_verdicts": hashlib.sha256(v[:65536].tobytes()).hexdigest()})
with open(os.path.join(ROOT, "tests", "golden", out), "w") as f:
    json.dump(rec, f, indent=1)
print(rec)
