#!/usr/bin/env python
"""Generates the frozen fixtures under tests/golden/ (run once, in the build container).

The reference has no golden vectors (it has no tests at all), so these pin (a) java.util.Random
known answers — the JDK javadoc LCG; the first five rows are the values quoted in SURVEY.md §8c,
the rest are produced by demi_amd.fuzzer.JavaRandom, the Python port of the documented algorithm —
and (b) our own frozen inputs and the CPU oracle's verdicts on them (regression pins, NOT JVM
outputs: parity versus the JVM reference is unpinned).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from demi_amd import types as T  # noqa: E402
from demi_amd.apps import SEED_BASE, raft3_config1, raft5_config2  # noqa: E402
from demi_amd.fuzzer import JavaRandom, array_to_events  # noqa: E402
from demi_amd.model import save_model  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)

kat = {"source": "JDK java.util.Random javadoc algorithm; rows marked survey are quoted in SURVEY.md 8c",
       "next_int": [{"seed": 0, "value": -1155484576, "survey": True}, {"seed": 42, "value": -1170105035, "survey": True}],
       "next_int_bound": [{"seed": 42, "bound": 10, "values": [0, 3, 8, 4, 0, 5, 5, 8, 9, 3], "survey": True},
                          {"seed": 0, "bound": 5, "values": [0, 3, 4, 2, 0, 3, 1, 1, 4, 4], "survey": True},
                          {"seed": 12345, "bound": 7, "values": [5, 2, 4, 6, 2, 4, 2, 4, 6, 1], "survey": True}]}
for seed, bound in [(1, 3), (2, 64), (3, 63), (0x5EED0000, 37), (0xDE31, 128), (7, 1), (99, 2), (2 ** 47 + 5, 100)]:
    r = JavaRandom(seed)
    kat["next_int_bound"].append({"seed": seed, "bound": bound, "values": [r.next_int(bound) for _ in range(16)]})
r = JavaRandom(0xDE31)
kat["next_double"] = {"seed": 0xDE31, "values": [r.next_double() for _ in range(8)]}
with open(os.path.join(G, "jrandom_kat.json"), "w") as f:
    json.dump(kat, f, indent=1)

for name, cfg, n in (("raft5_config2", raft5_config2, 4096), ("raft3_config1", raft3_config1, 100)):
    model, events, limits = cfg()
    save_model(model, os.path.join(G, name + "_model.json"))
    with open(os.path.join(G, name + "_trace.json"), "w") as f:
        json.dump({"events": array_to_events(events),
                   "limits": [limits.max_messages, limits.invariant_check_interval, limits.p_max, 0, 0, 0],
                   "seed_base": SEED_BASE}, f)
    v = O.random_explore(model, events, n, seed_base=SEED_BASE, limits=limits)
    np.save(os.path.join(G, name + "_verdicts.npy"), v)
    print(name, "violations", int((v["flags"] & T.V_VIOLATION).sum()), "of", n)
