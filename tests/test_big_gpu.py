"""GPU parity suite for tables of MORE THAN 8 ACTORS - the BIG layout of include/demi_gpu.h (9 .. 16 actors: 4-bit receiver and
5-bit sender fields in the message word, deadLetters = 31, 16-bit actor masks in the fingerprints, 16 x 16 partition / reach
matrices, 64 timer bits; a wide, compiled table).  The reference puts no bound on actor names (ExternalEvents.scala:62-91,
EventOrchestrator.scala:203-217, 345-351); rounds 1-5 stopped at 8.  Every kernel through the C ABI against the oracle, whose
own BIG layout is pinned by the literal transliterations of the Scala schedulers (tests/test_random_scheduler_transliteration_cpu.py
::test_tables_of_more_than_eight_actors..., tests/test_dpor_scheduler_transliteration_cpu.py cases shuffle12 / raft11) and by
tests/golden/big_tables.json.  Workloads: apps.raft11_config2 (11 raft nodes, fuzz trace with kills and partitions),
apps.shuffle12_config5 (driver, two coordinators, nine workers)."""
import hashlib
import json
import os

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft11_config2, raft11_dpor, shuffle12_config5

pytestmark = pytest.mark.gpu

EMU = os.environ.get("DEMI_EMU") == "1"
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_tables.json")


def _fuzz_workloads():
    m, ev, lim = raft11_config2()
    yield "raft11", m, ev, lim
    m, _dev, fev, lim, _par = shuffle12_config5()
    yield "shuffle12", m, fev, lim


def _ctx(model, events=None):
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    if events is not None:
        ctx.trace_load(events)
    ctx.model_specialize()
    assert ctx.is_specialized()
    return ctx


def _limits(lim, **kw):
    l = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 0, 0, 0)
    for k, v in kw.items():
        setattr(l, k, v)
    return l


def test_random_scheduler_on_tables_of_more_than_eight_actors(oracle):
    """K1, FullyRandom: every verdict field of n schedules = the oracle's; the golden record's prefix; the violating executions
    carry the BIG fingerprint layout (kind << 30 | key << 16 | 16-bit actor mask)."""
    with open(GOLDEN) as f:
        gold = json.load(f)
    n = 1024 if EMU else 1 << 16
    for name, m, ev, lim in _fuzz_workloads():
        ctx = _ctx(m, ev)
        try:
            g = ctx.random_explore(n, lim, seed_base=SEED_BASE)
            c = oracle.random_explore(m, ev, n, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
            assert (g == c).all(), name
            viol = g[(g["flags"] & T.V_VIOLATION) != 0]
            assert len(viol) >= (1 if EMU else 100) and not (g["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)).any()
            assert ((viol["fingerprint"] >> 30) == gold[name]["fingerprint_kind"]).all()
            k = gold[name]["fuzz_prefix"]
            if n >= k:
                assert hashlib.sha256(np.ascontiguousarray(g[:k]).tobytes()).hexdigest() == gold[name]["sha256_fuzz_verdicts"]
        finally:
            ctx.close()


def test_recording_fifo_and_carried_variants(oracle):
    """The other K1 variants a wide table gets compiled: the recording kernel (every recorded event incl. deadLetters = 31 as the
    sender of externals and timers), SrcDstFIFO (16 x 16 pairs: srcDsts scanned instead of a 64-bit pair mask) and the
    carried-generator mode."""
    n = 256 if EMU else 8192
    for name, m, ev, lim in _fuzz_workloads():
        ctx = _ctx(m, ev)
        try:
            for seed in (0, 1, 5, 17):
                v, rec = ctx.random_get_trace(SEED_BASE + seed, lim)
                ov, orec, _st = oracle.random_execute(m, ev, SEED_BASE + seed, lim)
                assert (v.flags, v.fingerprint, v.hash) == (ov.flags, ov.fingerprint, ov.hash) and len(rec) == len(orec) and (rec == orec).all()
                snd = rec["snd"][(rec["kind"] == T.REC_MSG_EVENT)]
                assert int(snd.max()) == T.DEADLETTERS_BIG and ((snd < m.n_actors) | (snd == T.DEADLETTERS_BIG)).all()
            lf = _limits(lim, strategy=1)
            g = ctx.random_explore(n, lf, seed_base=SEED_BASE)
            c = oracle.random_explore(m, ev, n, seed_base=SEED_BASE, limits=lf, n_threads=os.cpu_count())
            assert (g == c).all(), name + " SrcDstFIFO"
            v, rec = ctx.random_get_trace(SEED_BASE + 3, lf)
            ov, orec, _st = oracle.random_execute(m, ev, SEED_BASE + 3, lf)
            assert (v.flags, v.hash) == (ov.flags, ov.hash) and len(rec) == len(orec) and (rec == orec).all()
            lc = _limits(lim, executions_per_instance=4)
            g = ctx.random_explore(n, lc, seed_base=SEED_BASE)
            c = oracle.random_explore(m, ev, n, seed_base=SEED_BASE, limits=lc, n_threads=os.cpu_count())
            assert (g == c).all(), name + " carried generators"
        finally:
            ctx.close()


def test_replay_and_ddmin(oracle):
    """K2 (the scanning kernel a wide table replays with): candidates of a violating execution's externals under the three
    filterKnownAbsents settings, then demi_ddmin end to end against the same loop around the oracle."""
    rng = np.random.default_rng(7)
    n = 300 if EMU else 20000
    for name, m, ev, lim in _fuzz_workloads():
        ctx = _ctx(m, ev)
        try:
            l0 = _limits(lim, invariant_check_interval=0)
            v = oracle.random_explore(m, ev, 4000, seed_base=SEED_BASE, limits=l0, n_threads=os.cpu_count())
            idx = int(np.nonzero((v["flags"] & T.V_VIOLATION) != 0)[0][0])
            vd, rec = ctx.random_get_trace(SEED_BASE + idx, l0)
            assert vd.flags & T.V_VIOLATION
            lr = _limits(l0, looking_for_valid=1, looking_for=int(vd.fingerprint))
            masks = rng.integers(0, 2**63, size=(n, 4), dtype=np.uint64) | (rng.integers(0, 2, size=(n, 4), dtype=np.uint64) << np.uint64(63))
            masks[0] = 0xFFFFFFFFFFFFFFFF
            masks[1:n // 4] |= rng.integers(0, 2**63, size=(n // 4 - 1, 4), dtype=np.uint64)       # denser candidates
            ctx.replay_load(ev, rec)
            for fk in (0, 1, 2):
                lr.filter_known_absents = fk
                g = ctx.replay_batch(masks, lr)
                c = oracle.sts_replay_batch(m, ev, rec, masks, lr, n_threads=os.cpu_count())
                assert (g == c).all(), (name, fk)
                assert g[0]["flags"] & T.V_VIOLATION
            lr.filter_known_absents = 0
            if name == "raft11":        # (sub-traces of the 11-node cluster run into the pending capacity of 128: demi_ddmin refuses those)
                continue
            mcs, consulted, batches, st = ctx.ddmin(lr)
            omcs, oconsulted, obatches, ost = oracle.ddmin(m, ev, rec, lr, n_threads=os.cpu_count())
            assert mcs == omcs and consulted == oconsulted and 0 < len(mcs) < len(ev)
        finally:
            ctx.close()


def test_candidate_frontier_of_the_random_scheduler(oracle):
    """K1's MULTI variant (demi_random_explore_candidates, what demi_random_ddmin launches): a workgroup per candidate subsequence,
    verdict by verdict what the plain kernel gives for trace_load(candidate)."""
    m, _dev, ev, lim, _par = shuffle12_config5()
    ctx = _ctx(m, ev)
    try:
        rng = np.random.default_rng(3)
        n_ev = len(ev)
        masks = np.zeros((6, 4), dtype=np.uint64)
        masks[0, 0] = (1 << n_ev) - 1
        for i in range(1, 6):
            masks[i, 0] = int(rng.integers(0, 1 << n_ev)) | 0xFFF | (1 << 12)             # every Start, the Submit, some Speculates
        execs = 32 if EMU else 256
        gv, gf = ctx.random_explore_candidates(masks, execs, lim, seed_base=SEED_BASE)
        for i in range(len(masks)):
            sub = ev[[j for j in range(n_ev) if (int(masks[i, 0]) >> j) & 1]]
            c = oracle.random_explore(m, sub, execs, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
            assert (gv[i] == c).all(), i
            assert bool(gf[i] & 1) == bool((c["flags"] & T.V_VIOLATION).any())
    finally:
        ctx.close()


def test_dpor_in_both_orders(oracle):
    """K3 with the device-resident queue, the checkpoints (104-byte record headers: 64 timer bits) and k3_analyze_big (4-bit
    receiver field in the trace entries): ROUNDS order against the oracle's exploration in rounds, the reference's order against
    the oracle one backtrack point at a time.  On the GPU the 12-actor shuffle job is explored until its queue is empty:
    33 529 interleavings, 1 836 violating (tests/golden/big_tables.json)."""
    with open(GOLDEN) as f:
        gold = json.load(f)
    m, dev, _fev, _lim, par = shuffle12_config5()
    cases = [("shuffle12", m, dev, par)]
    m2, ev2, par2 = raft11_dpor()
    cases.append(("raft11", m2, ev2, par2))
    for name, model, ev, p in cases:
        for order, budget, batch in ((T.DPOR_ORDER_ROUNDS, 1500 if EMU else 1 << 16, 128 if EMU else 4096),
                                     (T.DPOR_ORDER_REFERENCE, 300 if EMU else 3000, 32 if EMU else 256)):
            if name == "raft11" and not EMU:
                budget = min(budget, 20000)
            ctx = _native.Context(0)
            try:
                ctx.model_load(model.to_struct())
                ctx.model_specialize()
                ctx.dpor_load(ev)
                g = ctx.dpor_explore(p, T.DporSearch(batch, budget, 0, 1, order))
            finally:
                ctx.close()
            if order == T.DPOR_ORDER_ROUNDS:
                c = oracle.dpor_explore(model, ev, p, T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS), os.cpu_count())
            else:
                c = oracle.dpor_explore(model, ev, p, T.DporSearch(1, budget, 0, 1, T.DPOR_ORDER_ROUNDS), 1)
            assert len(g[0]) == len(c[0]) and (g[0] == c[0]).all() and (g[1] == c[1]).all(), (name, order)
            assert not (g[0]["flags"] & 0xFC).any()
            if name == "shuffle12" and order == T.DPOR_ORDER_ROUNDS and not EMU:
                d = gold["shuffle12"]["dpor_rounds_batch_4096"]
                assert int(g[4].exhausted) == 1 and len(g[0]) == d["interleavings"] and int(g[4].violations) == d["violating"]
                assert hashlib.sha256(np.ascontiguousarray(g[0]).tobytes()).hexdigest() == d["sha256_verdicts"]


def test_reference_order_exhausted_is_the_scala_transliterations_record():
    """tests/golden/big_tables_transliteration.json (tools/check_big_transliteration.py): the literal transliteration of
    DPORwHeuristics explored the 12-actor job in the reference's own order until its queue was empty - 28 767 interleavings, 1 566
    violating - and ScalaRandomScheduler executed the first 4 096 schedules of both fuzz steps.  The device returns those bytes."""
    with open(os.path.join(os.path.dirname(GOLDEN), "big_tables_transliteration.json")) as f:
        tl = json.load(f)
    r = tl["shuffle12_dpor_reference_order"]
    assert r["exhausted"] and r["equals_the_oracles_one_at_a_time_exploration"] and tl["raft11"]["equals_the_oracle"] and tl["shuffle12"]["equals_the_oracle"]
    for name, m, ev, lim in _fuzz_workloads():
        ctx = _ctx(m, ev)
        try:
            k = 512 if EMU else tl[name]["schedules"]
            g = ctx.random_explore(k, lim, seed_base=SEED_BASE)
            if not EMU:
                assert hashlib.sha256(np.ascontiguousarray(g).tobytes()).hexdigest() == tl[name]["sha256_verdicts"]
                assert int(((g["flags"] & T.V_VIOLATION) != 0).sum()) == tl[name]["violating_executions"]
        finally:
            ctx.close()
    if EMU:
        return                  # (the whole exploration under the emulator: tools/emu_full_workloads.py big)
    m, dev, _fev, _lim, par = shuffle12_config5()
    ctx = _native.Context(0)
    try:
        ctx.model_load(m.to_struct())
        ctx.model_specialize()
        ctx.dpor_load(dev)
        g = ctx.dpor_explore(par, T.DporSearch(512, r["interleavings"] + 64, 0, 1, T.DPOR_ORDER_REFERENCE))
    finally:
        ctx.close()
    assert int(g[4].exhausted) == 1 and len(g[0]) == r["interleavings"] and int(g[4].violations) == r["violations"]
    assert hashlib.sha256(np.ascontiguousarray(g[0]).tobytes()).hexdigest() == r["sha256_verdicts"]
    assert hashlib.sha256(np.ascontiguousarray(g[1], dtype=np.uint32).tobytes()).hexdigest() == r["sha256_prefix_lens"]
    assert int(g[4].first_violation) == r["first_violation"]


def test_what_a_big_table_is_refused():
    """More than 8 actors need DEMI_MODEL_WIDE; more than 16 are refused."""
    ctx = _native.Context(0)
    try:
        m = M.raft_model(9)
        assert m.wide
        m.wide = False
        m.init_state = m.init_state[::2]
        with pytest.raises(_native.DemiError, match="DEMI_MODEL_WIDE"):
            ctx.model_load(m.to_struct())
        m17 = M.raft_model(11)
        m17.n_actors = 17
        with pytest.raises(_native.DemiError, match="n_actors"):
            ctx.model_load(m17.to_struct())
    finally:
        ctx.close()


def test_provenance_of_violations_on_a_big_table():
    """ProvenanceTracker.pruneConcurrentEvents (k_provenance_big: 4-bit receiver fields, up to 16 affected actors) over violating
    executions of the 12-actor job - recorded by K1, lowered by DepTracker.getInitialTrace's mirror (64-bit words of the BIG
    layout in the node keys) - against the host class (the shape of the reference's own)."""
    from demi_amd.incremental_ddmin import dpor_initial_trace
    from demi_amd.provenance import ProvenanceTracker, pruneConcurrentEventsBatch
    from demi_amd.schedulers import EventTrace
    m, _dev, fev, lim, _par = shuffle12_config5()
    ctx = _ctx(m, fev)
    try:
        v = ctx.random_explore(600 if EMU else 6000, lim, seed_base=SEED_BASE)
        traces, affected = [], []
        for i in np.nonzero(v["flags"] & T.V_VIOLATION)[0]:
            vv, rec = ctx.random_get_trace(SEED_BASE + int(i), lim)
            it = dpor_initial_trace(EventTrace(rec, fev[:T.verdict_trace_idx(vv.flags)]), m)
            if len(it) <= T.DPOR_MAX_TRACE:
                traces.append(it)
                affected.append(T.fingerprint_actors(vv.fingerprint))
            if len(traces) >= (10 if EMU else 150):
                break
        assert len(traces) >= 5 and max(max(a) for a in affected) >= T.MAX_ACTORS      # workers 8 .. 11 are among the violating actors
        got = pruneConcurrentEventsBatch(ctx, traces, affected)
        kept_some = 0
        for tr, aff, k in zip(traces, affected, got):
            w = tr[ProvenanceTracker(tr, big=True).pruneConcurrentEvents(aff)]
            assert len(k) == len(w) and (k == w).all()
            kept_some += int(0 < len(k) < len(tr))
        assert kept_some >= len(traces) // 2
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [2, 5])
def test_random_tables_of_sixteen_actors(oracle, seed):
    """Random wide tables at the layout's limit - 16 actors in three classes, every op, SENDs to computed targets (an id above 15
    addresses nobody), timers, RND, 16-bit external payloads - under traces with kills, partitions between any two of the sixteen
    and quiescence markers: K1 (both strategies, recorded traces), K2 over a recorded execution, K3."""
    from demi_amd.fuzzer import events_to_array, kill, partition, send, start, unpartition, wait_quiescence
    from .test_jit_cpu import _random_handler_wide
    rng = np.random.default_rng(1000 + seed)
    A = 16
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for cls in range(3):
        for name, _ in MSGS:
            if rng.integers(6):
                h[(cls, name)] = _random_handler_wide(rng, int(rng.integers(3, 24)), len(MSGS))
    model = M.build_model("rand_big%d" % seed, A, MSGS, h, [[int(x) for x in rng.integers(0, 65536, 8)] for _ in range(A)],
                          (T.INV_AT_MOST_ONE, 1, int(rng.integers(0, 4)), 2), actor_class=[int(x) for x in rng.integers(0, 3, A)], n_classes=3, wide=True)
    ev = [start(a) for a in range(A)]
    for i in range(44):
        k = int(rng.integers(0, 12))
        if k == 0 and ev[-1][0] != T.EV_WAIT_QUIESCENCE:
            ev.append(wait_quiescence())
        elif k == 1:
            ev.append(partition(int(rng.integers(0, A)), int(rng.integers(0, A))))
        elif k == 2:
            ev.append(unpartition(int(rng.integers(0, A)), int(rng.integers(0, A))))
        elif k == 3:
            ev.append(kill(int(rng.integers(0, A))) if rng.integers(2) else start(int(rng.integers(0, A))))
        else:
            ev.append(send(int(rng.integers(0, A)), 0, int(rng.integers(0, 65536)), int(rng.integers(0, 65536))))
    events = events_to_array(ev)
    lim = T.Limits(160, 9, 96, 0, 0, 0)
    n = 384 if EMU else 6000
    ctx = _ctx(model, events)
    try:
        for strategy in (0, 1):
            l2 = _limits(lim, strategy=strategy)
            g = ctx.random_explore(n, l2, seed_base=77)
            c = oracle.random_explore(model, events, n, seed_base=77, limits=l2, n_threads=os.cpu_count())
            assert (g == c).all(), strategy
            assert len(np.unique(g["hash"])) > n // 8
        rec = None
        for s in (77, 78, 90):
            v, rec = ctx.random_get_trace(s, lim)
            ov, orec, _st = oracle.random_execute(model, events, s, lim)
            assert (v.flags, v.fingerprint, v.hash) == (ov.flags, ov.fingerprint, ov.hash) and len(rec) == len(orec) and (rec == orec).all()
        # K2 over the last recorded execution
        masks = rng.integers(0, 2**63, size=(200, 4), dtype=np.uint64) | rng.integers(0, 2**63, size=(200, 4), dtype=np.uint64)
        masks[0] = 0xFFFFFFFFFFFFFFFF
        lr = _limits(lim, invariant_check_interval=0, looking_for_valid=1, looking_for=int(v.fingerprint))
        ctx.replay_load(events, rec)
        for fk in (0, 2):
            lr.filter_known_absents = fk
            gr = ctx.replay_batch(masks, lr)
            cr = oracle.sts_replay_batch(model, events, rec, masks, lr, n_threads=os.cpu_count())
            assert (gr == cr).all(), fk
        # K3: Start / Send / WaitQuiescence only
        dev = events_to_array([e for e in ev if e[0] in (T.EV_START, T.EV_SEND, T.EV_WAIT_QUIESCENCE)][:A + 7])
        par = T.DporParams(24, 120, 0, 0, 96, 4096, int(seed & 1))
        srch = T.DporSearch(64, 300 if EMU else 3000, 0, 1, T.DPOR_ORDER_ROUNDS)
        ctx.dpor_load(dev)
        gd = ctx.dpor_explore(par, srch)
        cd = oracle.dpor_explore(model, dev, par, srch, os.cpu_count())
        assert len(gd[0]) == len(cd[0]) and (gd[0] == cd[0]).all() and (gd[1] == cd[1]).all()
    finally:
        ctx.close()
