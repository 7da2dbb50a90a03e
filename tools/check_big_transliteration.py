#!/usr/bin/env python
"""tests/golden/big_tables_transliteration.json: the workloads with more than 8 actors (the BIG layout of include/demi_gpu.h) as
the literal Python transliterations of the Scala schedulers execute them - ScalaRandomScheduler
(tests/test_random_scheduler_transliteration_cpu.py) over the first N schedules of the fuzz steps of apps.raft11_config2 and
apps.shuffle12_config5, ScalaDPORwHeuristics (tests/test_dpor_scheduler_transliteration_cpu.py, lean queue) over the 12-actor
job's exploration in DPORwHeuristics' own order until its queue is empty - each held against the C oracle (the oracle's batch;
the product's one-at-a-time loop around the oracle's interleavings).  One core, tens of minutes: a tool, not a test; the
suites hold the first interleavings / executions of the same comparisons, and tests/test_big_gpu.py holds the device against this
record.  Usage: python tools/check_big_transliteration.py [N_FUZZ [DPOR_CAP]]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import SEED_BASE, raft11_config2, shuffle12_config5  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.test_dpor_scheduler_transliteration_cpu import ScalaDPORwHeuristics  # noqa: E402
from tests.test_random_scheduler_transliteration_cpu import ScalaRandomScheduler  # noqa: E402

n_fuzz = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
out = {"generator": "tools/check_big_transliteration.py %d %d" % (n_fuzz, cap), "seed_base": SEED_BASE}
ok = True
m, ev, lim = raft11_config2()
m2, dev, fev, lim2, par = shuffle12_config5()
for name, model, events, limits in (("raft11", m, ev, lim), ("shuffle12", m2, fev, lim2)):
    t0 = time.perf_counter()
    v = np.zeros(n_fuzz, dtype=T.VERDICT_DTYPE)
    for i in range(n_fuzz):
        s = ScalaRandomScheduler(O, model, events, SEED_BASE + i, limits.max_messages, limits.invariant_check_interval)
        s.execute()
        v[i] = s.verdict()
    oc = O.random_explore(model, events, n_fuzz, seed_base=SEED_BASE, limits=limits, n_threads=os.cpu_count())
    assert not (oc["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)).any()       # (capacities of the restatement: none hit on these workloads)
    same = bool((v == oc).all())
    ok &= same
    out[name] = {"model": model.name, "schedules": n_fuzz, "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(),
                 "violating_executions": int(((v["flags"] & T.V_VIOLATION) != 0).sum()), "equals_the_oracle": same,
                 "seconds": round(time.perf_counter() - t0, 1)}
    print(name, out[name], flush=True)
t0 = time.perf_counter()
sc = ScalaDPORwHeuristics(O, m2, dev, depth_bound=par.depth_bound, max_messages=0, prioritizePendingUponDivergence=bool(par.prioritize_pending), lean=True)
exhausted = sc.run(cap)
v = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
plen = np.array(sc.next_trace_lens, dtype=np.uint32)
one = O.dpor_explore(m2, dev, par, T.DporSearch(1, cap, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
same = len(one[0]) == len(v) and bool((one[0] == v).all()) and bool((one[1] == plen).all()) and bool(one[4].exhausted) == bool(exhausted)
ok &= same
out["shuffle12_dpor_reference_order"] = {"model": m2.name, "interleavings": int(len(v)), "exhausted": bool(exhausted),
                                         "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()),
                                         "first_violation": int(np.nonzero((v["flags"] & T.V_VIOLATION) != 0)[0][0]) if ((v["flags"] & T.V_VIOLATION) != 0).any() else -1,
                                         "sha256_verdicts": hashlib.sha256(v.tobytes()).hexdigest(),
                                         "sha256_prefix_lens": hashlib.sha256(plen.tobytes()).hexdigest(),
                                         "equals_the_oracles_one_at_a_time_exploration": same, "seconds": round(time.perf_counter() - t0, 1)}
print(out["shuffle12_dpor_reference_order"], flush=True)
with open(os.path.join(ROOT, "tests", "golden", "big_tables_transliteration.json"), "w") as f:
    json.dump(out, f, indent=1)
sys.exit(0 if ok else 1)
