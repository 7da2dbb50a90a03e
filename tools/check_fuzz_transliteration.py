#!/usr/bin/env python
"""The bench's fuzz workload (BASELINE config 2: raft5, the frozen 50-event trace, seeds SEED_BASE + i) executed by the literal
Python transliteration of the Scala RandomScheduler (tests/test_random_scheduler_transliteration_cpu.py ScalaRandomScheduler: its
own pendingEvents / RandomizedHashSet / java.util.Random, ExternalEventInjector and EventOrchestrator state; only the actors'
row interpreter is shared) for the first N schedules: verdict for verdict against the C oracle, and the SHA-256 of the N
verdicts into tests/golden/fuzz_config2_transliteration.json - which the CPU suite holds the oracle against, and the GPU
suite the device's verdicts of the same schedules.  About 4 ms per schedule and core; N = 2^20 - the bench's whole fixed-seed step - by default, over all cores.
Usage: python tools/check_fuzz_transliteration.py [N [processes]]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import SEED_BASE, raft5_config2  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.test_random_scheduler_transliteration_cpu import ScalaRandomScheduler  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)
model, events, lim = raft5_config2()


def chunk(lo_hi):
    lo, hi = lo_hi
    out = np.zeros(hi - lo, dtype=T.VERDICT_DTYPE)
    for i in range(lo, hi):
        s = ScalaRandomScheduler(O, model, events, SEED_BASE + i, lim.max_messages, lim.invariant_check_interval)
        s.execute()
        out[i - lo] = s.verdict()
    return lo, out


if __name__ == "__main__":
    import multiprocessing as mp
    want = O.random_explore(model, events, n, seed_base=SEED_BASE, limits=lim)
    cap = int(((want["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)) != 0).sum())
    assert cap == 0, "a capacity flag of the restatement in this workload (%d schedules): not a behaviour of the reference" % cap
    got = np.zeros(n, dtype=T.VERDICT_DTYPE)
    step = 4096
    t0 = time.perf_counter()
    with mp.Pool(procs) as pool:
        for lo, out in pool.imap_unordered(chunk, [(lo, min(n, lo + step)) for lo in range(0, n, step)]):
            got[lo:lo + len(out)] = out
    seconds = time.perf_counter() - t0
    same = bool((got == want).all())
    prefixes = [p for p in (1 << 14, 1 << 17, 1 << 20) if p <= n]
    rec = {"generator": "tools/check_fuzz_transliteration.py %d (ScalaRandomScheduler, %d processes, %.0f s)" % (n, procs, seconds),
           "schedules": n, "seed_base": int(SEED_BASE),
           "violating_executions": int(((got["flags"] & T.V_VIOLATION) != 0).sum()),
           "sha256_verdicts_of_the_first": {str(p): hashlib.sha256(got[:p].tobytes()).hexdigest() for p in prefixes},
           "equals_the_oracle": same}
    print(rec)
    if not same:
        bad = np.nonzero(got != want)[0]
        print("first differing schedule:", int(bad[0]), got[bad[0]], want[bad[0]])
        sys.exit(1)
    with open(os.path.join(ROOT, "tests", "golden", "fuzz_config2_transliteration.json"), "w") as f:
        json.dump(rec, f, indent=1)
