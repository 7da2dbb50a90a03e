#!/usr/bin/env python
"""The DPOR workloads in DPORwHeuristics' own order, explored by the literal Python transliteration of the Scala scheduler
(tests/test_dpor_scheduler_transliteration_cpu.py ScalaDPORwHeuristics: scheduling half and dpor(), its own dependency graph and
ExploredTacker - it shares neither the oracle's interleavings nor the product's bookkeeping, only the actors' row interpreter),
held against the C oracle under the product's batch = 1 loop.  Tens of milliseconds per interleaving on one core, so this is a
tool and not a test; the suite holds the first few hundred interleavings of the same comparisons.

  [CAP]                  config 3 of rounds 1-5 (apps.raft5_config3; 60 332 interleavings, most of an hour) against
                         tests/golden/dpor_config3_reference_order.json -> tests/golden/dpor_config3_transliteration.json
  --bug [CAP]            config 3 as timed from round 6 on (apps.raft5_dpor_config3: finds the seeded bug; hours) against
                         tests/golden/dpor_config3_bug_reference_order.json -> tests/golden/dpor_config3_bug_transliteration.json.
                         Every 16 384 interleavings the running SHA-256 goes to PROGRESS (default /tmp/dpor_tl_progress.json) so
                         that an interrupted run still leaves a checked prefix.
  --config5 N            the first N interleavings of the three-job pipeline of rounds 4-5 (apps.shuffle8_config5_large) against
                         the oracle's one-at-a-time exploration -> tests/golden/dpor_config5_transliteration.json
  --config5-bug N        the same for the pipeline timed from round 6 on (apps.shuffle8_dpor_config5)
                         -> tests/golden/dpor_config5_bug_transliteration.json"""
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
from tests.test_dpor_scheduler_transliteration_cpu import ScalaDPORwHeuristics  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

if len(sys.argv) > 2 and sys.argv[1] in ("--config5", "--config5-bug"):
    # the pipeline's first N interleavings against the C oracle under the product's one-at-a-time loop (what
    # tests/test_k3_gpu.py holds the GPU's REFERENCE order against)
    bug = sys.argv[1] == "--config5-bug"
    if bug:
        from demi_amd.apps import shuffle8_dpor_config5
        model, ev, par, _budget = shuffle8_dpor_config5()
    else:
        from demi_amd.apps import shuffle8_config5_large
        model, ev, depth, _budget = shuffle8_config5_large()
        par = T.DporParams(depth, 0, 0, 0, 64, 4096)
    cap = int(sys.argv[2])
    sc = ScalaDPORwHeuristics(O, model, ev, depth_bound=par.depth_bound, max_messages=0,
                              prioritizePendingUponDivergence=bool(par.prioritize_pending))
    t0 = time.perf_counter()
    exhausted = sc.run(cap)
    seconds = time.perf_counter() - t0
    v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
    plen = np.array(sc.next_trace_lens, dtype=np.uint32)
    one = O.dpor_explore(model, ev, par, T.DporSearch(1, cap, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    same = len(one[0]) == len(v) and bool((one[0] == v).all()) and bool((one[1] == plen).all())
    rec = {"generator": "tools/check_golden_dpor_transliteration.py %s %d (ScalaDPORwHeuristics, one core, %.0f s)" % (sys.argv[1], cap, seconds),
           "model": model.name, "interleavings": int(len(v)), "exhausted": bool(exhausted),
           "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(), "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
           "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()),
           "equals_the_oracles_one_at_a_time_exploration": same}
    print(rec)
    with open(os.path.join(GOLD, "dpor_config5_bug_transliteration.json" if bug else "dpor_config5_transliteration.json"), "w") as f:
        json.dump(rec, f, indent=1)
    sys.exit(0 if same else 1)

args = sys.argv[1:]
bug = bool(args) and args[0] == "--bug"
if bug:
    args = args[1:]
    model, ev, par = raft5_dpor_config3()
    gold_name, out_name = "dpor_config3_bug_reference_order.json", "dpor_config3_bug_transliteration.json"
else:
    model, ev, depth = raft5_config3()
    par = T.DporParams(depth, 0, 0, 0, 64, 4096)
    gold_name, out_name = "dpor_config3_reference_order.json", "dpor_config3_transliteration.json"
cap = int(args[0]) if args else 1 << 20
progress = os.environ.get("PROGRESS", "/tmp/dpor_tl_progress.json")
# lean: one backtrack-queue entry per flipped pair (the literal queue of this workload would hold ~6 x 10^8 replay lists);
# tests/test_dpor_scheduler_transliteration_cpu.py::test_lean_queue_is_the_literal_queue holds the two queues against each other
sc = ScalaDPORwHeuristics(O, model, ev, depth_bound=par.depth_bound, max_messages=0,
                          prioritizePendingUponDivergence=bool(par.prioritize_pending), lean=bug)
t0 = time.perf_counter()
exhausted = False
while not exhausted and len(sc.verdicts) < cap:           # run() continues an exploration: 16 384 interleavings at a time
    exhausted = sc.run(min(cap, len(sc.verdicts) + 16384))
    v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
    with open(progress, "w") as f:
        json.dump({"interleavings": len(v), "seconds": time.perf_counter() - t0, "exhausted": bool(exhausted),
                   "sha256_verdicts_so_far": hashlib.sha256(v.tobytes()).hexdigest(),
                   "sha256_first_65536_verdicts": hashlib.sha256(v[:65536].tobytes()).hexdigest() if len(v) >= 65536 else None,
                   "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum())}, f)
seconds = time.perf_counter() - t0
v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
plen = np.array(sc.next_trace_lens, dtype=np.uint32)
rec = {"generator": "tools/check_golden_dpor_transliteration.py%s (ScalaDPORwHeuristics, one core, %.0f s)" % (" --bug" if bug else "", seconds),
       "interleavings": int(len(v)), "exhausted": bool(exhausted),
       "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(), "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
       "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()), "distinct_schedules": int(len(set(v["hash"].tolist())))}
gold = json.load(open(os.path.join(GOLD, gold_name)))
same = all(rec[k] == gold[k] for k in ("interleavings", "exhausted", "sha256_verdicts", "sha256_prefix_lens", "violations", "distinct_schedules"))
rec["equals_" + gold_name.replace(".", "_")] = same
print(rec)
if cap >= gold["interleavings"]:
    with open(os.path.join(GOLD, out_name), "w") as f:
        json.dump(rec, f, indent=1)
sys.exit(0 if same or cap < gold["interleavings"] else 1)
