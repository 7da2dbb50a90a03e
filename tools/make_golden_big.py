#!/usr/bin/env python3
"""tests/golden/big_tables.json: the oracle's results on the workloads with more than 8 actors (the BIG layout of
include/demi_gpu.h) - apps.raft11_config2, apps.shuffle12_config5 - that the CPU suite re-checks the oracle against and the GPU
suite holds the kernels against (tests/test_big_gpu.py).  The oracle's BIG layout itself is pinned by the literal transliterations
of the Scala schedulers (tests/test_random_scheduler_transliteration_cpu.py, tests/test_dpor_scheduler_transliteration_cpu.py).
Run from the repo root: python tools/make_golden_big.py"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demi_amd import types as T                                            # noqa: E402
from demi_amd.apps import SEED_BASE, raft11_config2, shuffle12_config5     # noqa: E402
from oracle import oracle_py                                               # noqa: E402

PREFIX = 4096


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    out = {"generator": "tools/make_golden_big.py (oracle/demi_oracle.c)", "seed_base": SEED_BASE}
    m, ev, lim = raft11_config2()
    m2, dev, fev, lim2, par = shuffle12_config5()
    for name, model, events, limits in (("raft11", m, ev, lim), ("shuffle12", m2, fev, lim2)):
        v = oracle_py.random_explore(model, events, PREFIX, seed_base=SEED_BASE, limits=limits, n_threads=os.cpu_count())
        viol = v[(v["flags"] & T.V_VIOLATION) != 0]
        kinds = set(int(x) >> 30 for x in viol["fingerprint"])
        assert len(kinds) == 1
        out[name] = {"model": model.name, "n_actors": model.n_actors, "fuzz_prefix": PREFIX, "sha256_fuzz_verdicts": sha(v),
                     "violating_executions": int(len(viol)), "fingerprint_kind": kinds.pop(),
                     "distinct_fingerprints": sorted(set(int(x) for x in viol["fingerprint"]))[:16]}
    r = oracle_py.dpor_explore(m2, dev, par, T.DporSearch(4096, 1 << 17, 0, 1, T.DPOR_ORDER_ROUNDS), os.cpu_count())
    assert int(r[4].exhausted) == 1
    out["shuffle12"]["dpor_rounds_batch_4096"] = {"interleavings": int(len(r[0])), "violating": int(r[4].violations),
                                                  "first_violation": int(r[4].first_violation), "sha256_verdicts": sha(r[0]),
                                                  "sha256_prefix_lengths": sha(r[1])}
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "big_tables.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
