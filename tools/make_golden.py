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

# ---------------------------------------------------------------------------------------------
# Regression pins for the other paths (again: the CPU oracle's outputs on frozen inputs, not JVM outputs)
import hashlib  # noqa: E402

from demi_amd.fuzzer import events_to_array, send, start  # noqa: E402
from demi_amd import model as M  # noqa: E402
from demi_amd.dpor import DPORwHeuristics  # noqa: E402
from demi_amd.internal_minimization import deliveries  # noqa: E402
from demi_amd.minification import events_to_mask  # noqa: E402
from demi_amd.schedulers import EventTrace, SchedulerConfig  # noqa: E402

model, events, limits = raft5_config2()
# K1 with the SrcDstFIFO strategy
fifo = T.Limits(limits.max_messages, limits.invariant_check_interval, limits.p_max, 0, 0, 0, T.STRATEGY_SRC_DST_FIFO)
np.save(os.path.join(G, "raft5_config2_fifo_verdicts.npy"), O.random_explore(model, events, 1024, seed_base=SEED_BASE, limits=fifo))
# K2: the first violating execution of the frozen workload, subsequence masks, removal candidates, executed-trace marks
v = np.load(os.path.join(G, "raft5_config2_verdicts.npy"))
i0 = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec, _ = O.random_execute(model, events, SEED_BASE + i0, limits)
used = events[:T.verdict_trace_idx(vv.flags)]
rng = np.random.default_rng(2024)
masks = np.zeros((96, 4), dtype=np.uint64)
for r in range(96):
    keep = np.arange(len(used)) if r == 0 else np.nonzero(rng.random(len(used)) < rng.choice([0.3, 0.6, 0.9]))[0]
    masks[r] = events_to_mask(keep)
target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
dl = np.array([i for i, _, _ in deliveries(EventTrace(rec, used))], dtype=np.uint32)
skips = np.concatenate([dl[::3], [0xFFFFFFFF]]).astype(np.uint32)
kept = np.stack([O.sts_removal_kept(model, used, rec, int(s), target)[1] for s in skips[:4]])
np.savez(os.path.join(G, "raft5_config2_replay.npz"), index=i0, fingerprint=vv.fingerprint, rec=rec, used=used, masks=masks,
         mask_verdicts=O.sts_replay_batch(model, used, rec, masks, target), skips=skips,
         skip_verdicts=O.sts_removal_batch(model, used, rec, skips, target), kept=kept)
# K3: prefixes of a bounded exploration of raft3, per-interleaving outputs as verdicts + digests
m3 = M.raft_model(3)
ev3 = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
launched = []


def backend(m, e, prefixes, params, shared=None):
    launched.extend(prefixes)
    return O.dpor_batch(m, e, prefixes, params, shared)


DPORwHeuristics(SchedulerConfig(model=m3), depth_bound=30, stopIfViolationFound=False, batch=16, backend=backend).explore(
    ev3, max_interleavings=64)
par = T.DporParams(30, 0, 0, 0, 64, 4096, 0)
dv, dt, dp = O.dpor_batch(m3, ev3, launched, par)
stride = max(len(p) for p in launched)
pf = np.zeros((len(launched), stride), dtype=T.DPOR_TRACE_DTYPE)
for k, p in enumerate(launched):
    pf[k, :len(p)] = p
np.savez(os.path.join(G, "raft3_dpor.npz"), externals=ev3, prefixes=pf, prefix_len=np.array([len(p) for p in launched]),
         verdicts=dv, trace_len=np.array([len(t) for t in dt]), n_pairs=np.array([len(p) for p in dp]),
         trace_sha=np.array([hashlib.sha256(np.ascontiguousarray(t).tobytes()).hexdigest() for t in dt]),
         pairs_sha=np.array([hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() for p in dp]))
print("extra fixtures written")

# ---------------------------------------------------------------------------------------------
# Tables with arrays (DEMI_MODEL_ARRAY): the raft with a real log on the bench trace, the replicated log with its hole - the
# models as JSON (rows, flags) and the oracle's verdicts, both strategies
from demi_amd.fuzzer import wait_quiescence  # noqa: E402

_, events, limits = raft5_config2()
arr = {"raft5_log8": (M.raft_model(5, log_cap=8), events, limits, 2048),
       # DEMI_MODEL_PAYLOADS(5): the same raft with akka-raft's field sets on the wire (AppendEntries with five fields)
       "raft5_log8_fields": (M.raft_model(5, log_cap=8, real_fields=True), events, limits, 2048),
       "replog4_6": (M.replog_model(4, 6, True, False),
                     events_to_array([start(a) for a in range(4)] + [send(0 if i % 3 else i % 4, M.RL_PUT, 20 + i, 0) for i in range(6)]),
                     T.Limits(400, 7, 64, 0, 0, 0), 1024)}
for name, (m, ev, lim, n) in arr.items():
    save_model(m, os.path.join(G, name + "_model.json"))
    out = {}
    for sname, strat in (("random", T.STRATEGY_FULLY_RANDOM), ("fifo", T.STRATEGY_SRC_DST_FIFO)):
        l2 = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 0, 0, 0, strat)
        out[sname] = O.random_explore(m, ev, n, seed_base=SEED_BASE, limits=l2)
    np.savez(os.path.join(G, name + "_verdicts.npz"), events=ev, limits=np.array([lim.max_messages, lim.invariant_check_interval, lim.p_max]), **out)
    print(name, {k: int((v["flags"] & T.V_VIOLATION).sum()) for k, v in out.items()}, "of", n)
