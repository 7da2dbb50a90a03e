#!/usr/bin/env python
"""The UNMODIFIED kernel sources (demi_amd/csrc, compiled with g++ on top of the wave64 emulator of tests/emu: TEST
INFRASTRUCTURE) over the FULL records of the timed workloads - what rounds 4-5 checked "once by hand", as a command:

  k1        config 2, all 2^20 schedules of the fixed-seed step, interpreter and compiled table
            -> must equal tests/golden/fuzz_config2_transliteration.json (the Scala RandomScheduler's transliteration)
  k2        config 4, all 2^20 candidate subsequences -> tests/golden/replay_config4_transliteration.json
  k3        config 3 of rounds 1-5 in the REFERENCE order (60 332 interleavings) -> tests/golden/dpor_config3_reference_order.json
  k3bug     config 3 as timed from round 6 on, REFERENCE order (258 025) -> tests/golden/dpor_config3_bug_reference_order.json,
            and ROUNDS order (297 396) -> the oracle's exploration in rounds
  k3c5      config 5 (round 6), REFERENCE order, first 6 000 -> tests/golden/dpor_config5_bug_transliteration.json
  big       tables of more than 8 actors: raft11 / shuffle12 fuzz prefixes, shuffle12's DPOR exhausted in both orders
            -> tests/golden/big_tables.json, big_tables_transliteration.json

Usage: python tools/emu_full_workloads.py [k1 k2 k3 k3bug k3c5 ...]   (default: all; W64_THREADS = host threads the emulator's
workgroups run on, default: all cores).  Writes tests/golden/emu_full_workloads.json (one entry per workload: SHA-256s, whether
they equal the record, seconds); the CPU suite runs short slices of the same (tests/test_emu_suite_cpu.py).  No GPU, no oracle
in the measured path: the oracle is only asked for the ROUNDS exploration of k3bug, which has no committed record."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DEMI_EMU"] = "1"
os.environ.setdefault("DEMI_EXPERIMENT", "1")
os.environ.setdefault("W64_THREADS", str(os.cpu_count() or 1))

import numpy as np  # noqa: E402

from tests.emu import build as emu_build  # noqa: E402

emu = emu_build.build()
os.environ["DEMI_NO_TORCH"] = "1"
os.environ["DEMI_HIPRTC_LIB"] = emu["hiprtc"]
from demi_amd import _native, types as T  # noqa: E402

_native.LIB_PATH = emu["lib"]
from demi_amd.apps import SEED_BASE, raft5_config2, raft5_config3, raft5_config4, raft5_dpor_config3, shuffle8_dpor_config5  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "emu_full_workloads.json")


def sha(a, dtype=None):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=dtype).tobytes()).hexdigest()


def gold(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def k1():
    rec = gold("fuzz_config2_transliteration.json")
    model, events, lim = raft5_config2()
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    n, out = 1 << 20, {}
    for name in ("interpreter", "compiled_table"):
        if name == "compiled_table":
            ctx.model_specialize()
        t = time.perf_counter()
        v = ctx.random_explore(n, lim, seed_base=SEED_BASE)
        out[name] = {"sha256_verdicts": sha(v), "seconds": time.perf_counter() - t,
                     "violating_executions": int(((v["flags"] & T.V_VIOLATION) != 0).sum())}
        out[name]["equals_the_record"] = out[name]["sha256_verdicts"] == rec["sha256_verdicts_of_the_first"][str(n)]
    ctx.close()
    return {"workload": "config 2: 2^20 schedules, demi_random_explore", "record": "fuzz_config2_transliteration.json", **out}


def k2():
    rec_t = gold("replay_config4_transliteration.json")
    model, events, lim = raft5_config4()
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    # the failing execution: the first violating schedule of the 200-event trace, recorded by the (emulated) recording kernel
    v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(int(vv.flags))]
    n = rec_t["candidates"]
    keep = np.random.default_rng(0).random((n, len(used))) < 0.7
    masks = np.zeros((n, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    ctx.replay_load(used, rec)
    t = time.perf_counter()
    got = ctx.replay_batch(masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0))
    dt = time.perf_counter() - t
    ctx.close()
    return {"workload": "config 4: 2^20 candidate subsequences, demi_replay_batch", "record": "replay_config4_transliteration.json",
            "sha256_masks": sha(masks), "sha256_verdicts": sha(got), "seconds": dt,
            "still_violating": int(((got["flags"] & T.V_VIOLATION) != 0).sum()),
            "equals_the_record": sha(masks) == rec_t["sha256_masks_of_all"] and sha(got) == rec_t["sha256_verdicts_of_all"]}


def _dpor(model, ev, par, order, budget, batch):
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    t = time.perf_counter()
    v, plen, rounds, _vt, st = ctx.dpor_explore(par, T.DporSearch(batch, budget, 0, 1, order))
    dt = time.perf_counter() - t
    ctx.close()
    return v, plen, st, dt


def k3():
    g = gold("dpor_config3_reference_order.json")
    model, ev, depth = raft5_config3()
    v, plen, st, dt = _dpor(model, ev, T.DporParams(depth, 0, 0, 0, 64, 4096), T.DPOR_ORDER_REFERENCE, 1 << 17, 2048)
    return {"workload": "config 3 of rounds 1-5, REFERENCE order, speculation 2 048 wide", "record": "dpor_config3_reference_order.json",
            "interleavings": len(v), "executed": int(st.executed), "sha256_verdicts": sha(v, T.VERDICT_DTYPE), "sha256_prefix_lens": sha(plen, np.uint32),
            "seconds": dt, "equals_the_record": sha(v, T.VERDICT_DTYPE) == g["sha256_verdicts"] and sha(plen, np.uint32) == g["sha256_prefix_lens"]}


def k3bug():
    g = gold("dpor_config3_bug_reference_order.json")
    model, ev, par = raft5_dpor_config3()
    v, plen, st, dt = _dpor(model, ev, par, T.DPOR_ORDER_REFERENCE, 1 << 20, 2048)
    out = {"workload": "config 3 (round 6: finds the seeded bug)", "record": "dpor_config3_bug_reference_order.json",
           "reference_order": {"interleavings": len(v), "executed": int(st.executed), "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()),
                               "sha256_verdicts": sha(v, T.VERDICT_DTYPE), "sha256_prefix_lens": sha(plen, np.uint32), "seconds": dt,
                               "equals_the_record": sha(v, T.VERDICT_DTYPE) == g["sha256_verdicts"] and sha(plen, np.uint32) == g["sha256_prefix_lens"]}}
    # ROUNDS of 16 384 on ONE host thread: pair_slot() gives a claimed-but-unpublished table entry 4 096 looks before it reports the
    # table full - on the device the claiming lane publishes within the same iteration; an OS thread of the emulator can be
    # descheduled for milliseconds in between, and with rounds this wide that does happen (seen with W64_THREADS=5)
    threads = os.environ["W64_THREADS"]
    os.environ["W64_THREADS"] = "1"
    v, plen, st, dt = _dpor(model, ev, par, T.DPOR_ORDER_ROUNDS, 1 << 19, 16384)
    os.environ["W64_THREADS"] = threads
    from oracle import oracle_py as O          # (the checker: ROUNDS order has no committed record)
    cpu = O.dpor_explore(model, ev, par, T.DporSearch(16384, 1 << 19, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=os.cpu_count() or 1)
    out["rounds"] = {"interleavings": len(v), "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()), "sha256_verdicts": sha(v, T.VERDICT_DTYPE),
                     "seconds": dt, "equals_the_oracles_exploration_in_rounds": len(cpu[0]) == len(v) and bool((cpu[0] == v).all()) and bool((cpu[1] == plen).all())}
    return out


def k3c5():
    g = gold("dpor_config5_bug_transliteration.json")
    model, ev, par, _budget = shuffle8_dpor_config5()
    v, plen, st, dt = _dpor(model, ev, par, T.DPOR_ORDER_REFERENCE, g["interleavings"], 1024)
    return {"workload": "config 5 (round 6), REFERENCE order, first %d interleavings" % g["interleavings"], "record": "dpor_config5_bug_transliteration.json",
            "interleavings": len(v), "executed": int(st.executed), "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()),
            "sha256_verdicts": sha(v, T.VERDICT_DTYPE), "sha256_prefix_lens": sha(plen, np.uint32), "seconds": dt,
            "equals_the_record": sha(v, T.VERDICT_DTYPE) == g["sha256_verdicts"] and sha(plen, np.uint32) == g["sha256_prefix_lens"]}


def big():
    """Tables of more than 8 actors (the BIG layout): K1 over the fuzz prefixes of raft11 / shuffle12, K3 over the 12-actor job's
    exploration to exhaustion in both orders -> tests/golden/big_tables.json (the oracle's) and big_tables_transliteration.json
    (the Scala schedulers' transliterations')."""
    from demi_amd.apps import raft11_config2, shuffle12_config5
    g = gold("big_tables.json")
    try:
        tl = gold("big_tables_transliteration.json")
    except OSError:
        tl = None
    out = {"workload": "raft11 / shuffle12 fuzz prefixes, shuffle12 DPOR exhausted in both orders", "record": "big_tables.json, big_tables_transliteration.json"}
    ok = True
    m, ev, lim = raft11_config2()
    m2, dev, fev, lim2, par = shuffle12_config5()
    t0 = time.perf_counter()
    for name, model, events, limits in (("raft11", m, ev, lim), ("shuffle12", m2, fev, lim2)):
        ctx = _native.Context(0)
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        v = ctx.random_explore(g[name]["fuzz_prefix"], limits, seed_base=SEED_BASE)
        ctx.close()
        out[name + "_sha256_fuzz_verdicts"] = sha(v, T.VERDICT_DTYPE)
        same = out[name + "_sha256_fuzz_verdicts"] == g[name]["sha256_fuzz_verdicts"]
        if tl and tl[name]["schedules"] == g[name]["fuzz_prefix"]:
            same = same and out[name + "_sha256_fuzz_verdicts"] == tl[name]["sha256_verdicts"]
        ok &= same
    d = g["shuffle12"]["dpor_rounds_batch_4096"]
    v, plen, st, dt = _dpor(m2, dev, par, T.DPOR_ORDER_ROUNDS, 1 << 17, 4096)
    out["dpor_rounds"] = {"interleavings": len(v), "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()), "sha256_verdicts": sha(v, T.VERDICT_DTYPE)}
    ok &= len(v) == d["interleavings"] and out["dpor_rounds"]["sha256_verdicts"] == d["sha256_verdicts"] and sha(plen, np.uint32) == d["sha256_prefix_lengths"]
    if tl:
        r = tl["shuffle12_dpor_reference_order"]
        v, plen, st, dt = _dpor(m2, dev, par, T.DPOR_ORDER_REFERENCE, r["interleavings"] + 64, 512)
        out["dpor_reference_order"] = {"interleavings": len(v), "executed": int(st.executed), "violations": int(((v["flags"] & T.V_VIOLATION) != 0).sum()),
                                       "sha256_verdicts": sha(v, T.VERDICT_DTYPE)}
        ok &= len(v) == r["interleavings"] and out["dpor_reference_order"]["sha256_verdicts"] == r["sha256_verdicts"] and sha(plen, np.uint32) == r["sha256_prefix_lens"]
    out["seconds"] = time.perf_counter() - t0
    out["equals_the_record"] = bool(ok)
    return out


WORK = {"k1": k1, "k2": k2, "k3": k3, "k3bug": k3bug, "k3c5": k3c5, "big": big}

if __name__ == "__main__":
    which = sys.argv[1:] or list(WORK)
    try:
        with open(OUT) as f:
            res = json.load(f)
    except (OSError, ValueError):
        res = {}
    res["generator"] = "tools/emu_full_workloads.py (demi_amd/csrc compiled with g++ over tests/emu, W64_THREADS=%s)" % os.environ["W64_THREADS"]
    for w in which:
        t = time.perf_counter()
        res[w] = WORK[w]()
        res[w]["wall_seconds"] = time.perf_counter() - t
        print(w, json.dumps(res[w]), flush=True)
        with open(OUT, "w") as f:
            json.dump(res, f, indent=1)
