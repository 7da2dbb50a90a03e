#!/usr/bin/env python
"""BASELINE config 3 in DPORwHeuristics' own order, explored by the literal Python transliteration of the Scala scheduler
(tests/test_dpor_scheduler_transliteration_cpu.py ScalaDPORwHeuristics: scheduling half and dpor(), its own dependency graph and
ExploredTacker - it shares neither the oracle's interleavings nor the product's bookkeeping, only the actors' row interpreter),
held against tests/golden/dpor_config3_reference_order.json (the C oracle under the product's batch = 1 loop).  About 40 ms per
interleaving, 60 332 of them: the better part of an hour on one core, so this is a tool and not a test; the suite holds the
first few hundred interleavings of the same comparison (tests/test_dpor_scheduler_transliteration_cpu.py).  Writes
tests/golden/dpor_config3_transliteration.json."""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import raft5_config3  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.test_dpor_scheduler_transliteration_cpu import ScalaDPORwHeuristics  # noqa: E402

if len(sys.argv) > 2 and sys.argv[1] == "--config5":
    # config 5's three-job pipeline, its first N interleavings, against the C oracle under the product's one-at-a-time loop
    # (what tests/test_k3_gpu.py::test_config5_pipeline_in_reference_order_is_the_one_at_a_time_sequence holds the GPU against)
    from demi_amd.apps import shuffle8_config5_large
    model, ev, depth, _budget = shuffle8_config5_large()
    cap = int(sys.argv[2])
    sc = ScalaDPORwHeuristics(O, model, ev, depth_bound=depth, max_messages=0)
    t0 = time.perf_counter()
    exhausted = sc.run(cap)
    seconds = time.perf_counter() - t0
    v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
    plen = np.array(sc.next_trace_lens, dtype=np.uint32)
    one = O.dpor_explore(model, ev, T.DporParams(depth, 0, 0, 0, 64, 4096), T.DporSearch(1, cap, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    same = len(one[0]) == len(v) and bool((one[0] == v).all()) and bool((one[1] == plen).all())
    rec = {"generator": "tools/check_golden_dpor_transliteration.py --config5 %d (ScalaDPORwHeuristics, one core, %.0f s)" % (cap, seconds),
           "interleavings": int(len(v)), "exhausted": bool(exhausted), "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(),
           "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
           "equals_the_oracles_one_at_a_time_exploration": same}
    print(rec)
    with open(os.path.join(ROOT, "tests", "golden", "dpor_config5_transliteration.json"), "w") as f:
        json.dump(rec, f, indent=1)
    sys.exit(0 if same else 1)

model, ev, depth = raft5_config3()
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
sc = ScalaDPORwHeuristics(O, model, ev, depth_bound=depth, max_messages=0)
t0 = time.perf_counter()
exhausted = sc.run(cap)
seconds = time.perf_counter() - t0
v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
plen = np.array(sc.next_trace_lens, dtype=np.uint32)
rec = {"generator": "tools/check_golden_dpor_transliteration.py (ScalaDPORwHeuristics, one core, %.0f s)" % seconds,
       "interleavings": int(len(v)), "exhausted": bool(exhausted),
       "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(), "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
       "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()), "distinct_schedules": int(len(set(v["hash"].tolist())))}
gold = json.load(open(os.path.join(ROOT, "tests", "golden", "dpor_config3_reference_order.json")))
same = all(rec[k] == gold[k] for k in ("interleavings", "exhausted", "sha256_verdicts", "sha256_prefix_lens", "violations", "distinct_schedules"))
rec["equals_dpor_config3_reference_order_json"] = same
print(rec)
if cap >= gold["interleavings"]:
    with open(os.path.join(ROOT, "tests", "golden", "dpor_config3_transliteration.json"), "w") as f:
        json.dump(rec, f, indent=1)
sys.exit(0 if same or cap < gold["interleavings"] else 1)
