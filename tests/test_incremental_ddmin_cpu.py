"""CPU suite, part 7: DDMin over DPOR with a growing edit-distance bound (IncrementalDDMin, ResumableDPOR,
ArvindDistanceOrdering, prioritizePendingUponDivergence), with the oracle standing in for the K3 kernel."""
import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.dpor import ArvindDistanceOrdering, DPORwHeuristics, DefaultBacktrackOrdering, StopImmediatelyOrdering
from demi_amd.fuzzer import events_to_array, kill, partition, send, start, wait_quiescence
from demi_amd.incremental_ddmin import (IncrementalDDMin, ResumableDPOR, convertToDPORTrace, dpor_initial_trace,
                                        editDistanceDporDDMin)
from demi_amd.minification import EventDagView, UnmodifiedEventDag
from demi_amd.model import Asm, build_model
from demi_amd.schedulers import EventTrace, SchedulerConfig, ViolationFingerprint

from .test_dpor_cpu import two_writers_model


def _execution(oracle, model, ev, want_violation=True, lim=None):
    lim = lim or T.Limits(0, 0, 64, 0, 0, 0)
    for seed in range(200):
        v, rec, _ = oracle.random_execute(model, ev, seed, lim)
        if bool(v.flags & T.V_VIOLATION) == want_violation:
            return v, EventTrace(rec, ev)
    raise AssertionError("no execution with the wanted outcome")


def test_initial_trace_of_a_random_execution_replays_under_dpor(oracle):
    """The deliveries of a RandomScheduler execution, re-identified by causal-path keys, are exactly the nodes DPOR
    generates: replaying them as nextTrace reproduces the same delivery sequence and the same verdict hash."""
    model = two_writers_model()
    ev = events_to_array([start(0), start(1), start(2), send(1, 0), send(2, 0), send(1, 0)])
    v, trace = _execution(oracle, model, ev)
    init = dpor_initial_trace(trace)
    assert init[0]["kind"] == 0 and int(init[0]["key"]) == T.DPOR_ROOT_KEY
    assert len(init) == 1 + T.verdict_deliveries(v.flags)
    out = oracle.dpor_batch(model, ev, [init], T.DporParams(0, 0, 0, 0, 64, 4096, 0))
    dv, dt = out[0][0], out[1][0]
    assert (dt["key"] == init["key"]).all() and (dt["word"] == init["word"]).all()
    assert (dt["parent"] == init["parent"]).all() and (dt["depth"] == init["depth"]).all()
    assert int(dv["hash"]) == v.hash and int(dv["fingerprint"]) == v.fingerprint
    # raft (no repeating timer before a leader exists): same property on a longer execution
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    v, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(12, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace)
    dt = oracle.dpor_batch(model, ev, [init], T.DporParams(0, len(init), 0, 0, 64, 4096, 0))[1][0]
    n = min(len(dt), len(init))
    assert n >= 8 and (dt["key"][:8] == init["key"][:8]).all()
    # a wide table with five payload fields: the keys chain the 64-bit message words (header | payload area << 16)
    model = M.raft_model(3, log_cap=4, real_fields=True)
    v, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(12, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace, model)
    dt = oracle.dpor_batch(model, ev, [init], T.DporParams(0, len(init), 0, 0, 64, 4096, 0))[1][0]
    n = min(len(dt), len(init))
    assert n >= 8 and (dt["key"][:8] == init["key"][:8]).all() and (dt["word"][:8] == init["word"][:8]).all()
    assert (trace.events["p0"] != 0).any() and (dpor_initial_trace(trace)["key"] != init["key"]).any()     # (the 32-bit layout gives other keys)


def test_prioritize_pending_upon_divergence(oracle):
    """Expected heads that are not pending are popped until one is (getNextMatchingMessage); without the option
    only one head is tried and the scheduler then diverges to the first pending message."""
    model = two_writers_model()
    ev = events_to_array([start(0), start(1), start(2), send(1, 0), send(2, 0)])
    base = oracle.dpor_batch(model, ev, [np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)], T.DporParams(0, 0, 0, 0, 64, 4096, 0))[1][0]
    deliveries = base[base["kind"] == 1]
    go2 = deliveries[[int(w) >> 5 & 7 == 2 and int(w) & 31 == 0 for w in deliveries["word"]]][0]
    bogus = np.zeros(2, dtype=T.DPOR_TRACE_DTYPE)
    bogus["kind"] = 1; bogus["key"] = [12345, 67890]; bogus["word"] = [int(go2["word"])] * 2
    prefix = np.concatenate([base[:1], bogus, np.array([go2])])
    with_p = oracle.dpor_batch(model, ev, [prefix], T.DporParams(0, 0, 0, 0, 64, 4096, 1))[1][0]
    without = oracle.dpor_batch(model, ev, [prefix], T.DporParams(0, 0, 0, 0, 64, 4096, 0))[1][0]
    assert int(with_p[1]["key"]) == int(go2["key"])                 # skipped both absent heads, then matched
    assert int(without[1]["key"]) == int(base[1]["key"]) != int(go2["key"])   # diverged to the pinned first pending


def test_arvind_distance_is_the_literal_path_computation():
    def tr(rows):      # (key, parent)
        a = np.zeros(len(rows), dtype=T.DPOR_TRACE_DTYPE)
        a["key"] = [r[0] for r in rows]; a["parent"] = [r[1] for r in rows]; a["kind"] = [0] + [1] * (len(rows) - 1)
        return a

    orig = tr([(1, 0), (10, 0), (20, 0), (30, 1), (40, 2)])
    h = ArvindDistanceOrdering()
    h.init(None, orig)
    # the original order itself: path(later=3 (key 30, parent 1), earlier=2, branch 0) = [1, 10, 30] ++ [10, 30] ++ [30, 20]
    # inversions: (30 before 10): 1, (30,20): three 30s before the final 20 -> 3  => 4
    assert h.arvindDistance(orig, 0, 3, 2) == 4
    # unknown events cost 1 each and do not take part in the inversion count
    other = tr([(1, 0), (10, 0), (99, 0), (30, 1)])
    assert h.arvindDistance(other, 0, 3, 2) == h.arvindDistance(tr([(1, 0), (10, 0), (20, 0), (30, 1)]), 0, 3, 2) + 1 - 3
    assert h.priority(orig, 0, 3, 2) == (4, 0) and h.getDistance((4, 0)) == 4
    assert DefaultBacktrackOrdering().getDistance((7,)) == 0 and StopImmediatelyOrdering().getDistance((7,)) > 1 << 30


def test_distance_cap_and_resumption(oracle):
    """setMaxDistance(0) runs the initial trace only (every queued point is >= 0 away); raising the cap resumes from
    the queue without re-running the initial trace; the farthest point is dequeued first."""
    model = two_writers_model()
    ev = events_to_array([start(0), start(1), start(2), send(1, 0), send(2, 0), send(1, 0)])
    _, trace = _execution(oracle, model, ev, want_violation=False)
    init = dpor_initial_trace(trace)
    h = ArvindDistanceOrdering()
    d = DPORwHeuristics(SchedulerConfig(model=model), prioritizePendingUponDivergence=True, backtrackHeuristic=h,
                        stopIfViolationFound=False, batch=4, backend=oracle.dpor_batch)
    d.setInitialTrace(init)
    h.init(d, init)
    d.setMaxDistance(0)
    r0 = d.explore(ev)
    assert len(r0.interleavings) == 1 and (r0.interleavings[0].trace["key"] == init["key"]).all()
    queued = len(d.backTrack)
    assert queued > 0 and not r0.exhausted
    head_distance = -d.backTrack[0][0][0]
    assert head_distance == max(-e[0][0] for e in d.backTrack)
    d.setMaxDistance(head_distance)              # still capped: the head is exactly at the bound
    assert len(d.explore(ev).interleavings) == 0
    d.setMaxDistance(1 << 20)
    r2 = d.explore(ev)
    assert len(r2.interleavings) >= 1 and r2.interleavings[0].prefix_len > 0      # resumed, not restarted
    assert r2.exhausted


def test_convert_to_dpor_trace_drops_faults():
    ev = events_to_array([start(0), kill(0), start(0), send(0, 0), partition(0, 1), wait_quiescence(), send(0, 0)])
    assert [int(k) for k in convertToDPORTrace(ev)["kind"]] == [T.EV_START, T.EV_START, T.EV_SEND, T.EV_SEND]
    assert len(convertToDPORTrace(ev, ignoreQuiescence=False)) == 5


def test_edit_distance_dpor_ddmin_end_to_end(oracle):
    """two writers + noise: the violation (actor 0's last writer is 1) needs Start(0), Start(1) and one Go to actor 1;
    the second writer, the duplicate Go and the idle actor are pruned."""
    MSGS = [("Go", T.MSG_EXTERNAL), ("Write", T.MSG_INTERNAL)]
    hnd = {(0, "Go"): Asm().mov(M.T0, 0).send(1, M.T0, M.ME, 0),
           (0, "Write"): Asm().mov(M.F[0], M.P0).add(M.F[1], M.F[1], 1)}
    model = build_model("race4", 4, MSGS, hnd, [[0] * 8] * 4, (T.INV_NEVER, 0, 1, 0))
    ev = events_to_array([start(0), start(1), start(2), start(3), send(2, 0), send(1, 0), send(3, 0), send(2, 0)])
    v, trace = _execution(oracle, model, ev)
    fp = ViolationFingerprint(v.fingerprint)
    mcs, ddmin, verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=2, maxMaxDistance=8,
                                                    batch=8, backend=oracle.dpor_batch)
    assert verified is not None
    kinds = [(int(ev[i]["kind"]), int(ev[i]["a"])) for i in mcs]
    assert (T.EV_START, 0) in kinds and (T.EV_START, 1) in kinds and (T.EV_SEND, 1) in kinds
    assert len(mcs) < len(ev) and len(mcs) <= 4
    assert ddmin.distances[0][0] == 0 and ddmin._stats.total_replays > 0
    # one DPOR instance per consulted subsequence, reused across distance rounds
    assert len(ddmin.oracle.subseqToDPOR) >= len({c for c, _ in ddmin.ddmin.consulted})


def test_edit_distance_dpor_ddmin_on_a_table_with_payload_fields(oracle):
    """The same search on a wide table whose messages carry four fields (DEMI_MODEL_PAYLOADS(4)): the DPOR initial trace chains
    64-bit message words, the writer's identity travels in the third field (LDP / PSET), the violation needs the same three
    externals."""
    MSGS = [("Go", T.MSG_EXTERNAL), ("Write", T.MSG_INTERNAL)]
    hnd = {(0, "Go"): Asm().mov(M.T0, 0).ldi16(M.T1, 0x123).send(1, M.T0, M.T1, 7, M.ME, 0xEE),
           (0, "Write"): Asm().ldp(M.F[0], 2).ldp(M.F[2], 3).add(M.F[1], M.F[1], 1)}
    model = build_model("race4p", 4, MSGS, hnd, [[0] * 8] * 4, (T.INV_NEVER, 0, 1, 0), wide=True, payloads=4)
    assert oracle.model_validate(model)[0] == 0
    ev = events_to_array([start(0), start(1), start(2), start(3), send(2, 0), send(1, 0), send(3, 0), send(2, 0)])
    v, trace = _execution(oracle, model, ev)
    writes = trace.events[(trace.events["kind"] == T.REC_MSG_EVENT) & (trace.events["msg_type"] == 1)]
    assert len(writes) and all(T.payload_fields(T.rec_area(e), 4)[:2] == [0x123, 7] and T.payload_fields(T.rec_area(e), 4)[3] == 0xEE for e in writes)
    fp = ViolationFingerprint(v.fingerprint)
    mcs, ddmin, verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=2, maxMaxDistance=8,
                                                    batch=8, backend=oracle.dpor_batch)
    assert verified is not None
    kinds = [(int(ev[i]["kind"]), int(ev[i]["a"])) for i in mcs]
    assert (T.EV_START, 0) in kinds and (T.EV_START, 1) in kinds and (T.EV_SEND, 1) in kinds
    assert len(mcs) < len(ev) and len(mcs) <= 4 and ddmin.distances[0][0] == 0


@pytest.mark.parametrize("which", ["two_writers", "raft3"])
def test_native_ordered_exploration_equals_the_python_mirror(oracle, which):
    """demi_dpor_explore's plain loop for ArvindDistanceOrdering / setMaxDistance / setInitialTrace (dpor_host.hpp
    explore_rounds_ordered, here around the oracle's interleavings) against the Python mirror's explore() of a fresh
    DPORwHeuristics: the same rounds, verdicts (incl. the delivery hashes), prefix lengths and exhaustion - with the Arvind
    ordering uncapped and capped at several distances, with the default ordering under a cap, with and without an initial
    trace, one at a time and in rounds."""
    from oracle import oracle_py
    if which == "two_writers":
        model = two_writers_model()
        ev = events_to_array([start(0), start(1), start(2), send(1, 0), send(2, 0), send(1, 0)])
        depth, budget = 0, 400
    else:
        model = M.raft_model(3)
        ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
        depth, budget = 30, 300
    _, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(60, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace)
    par = T.DporParams(depth, 0, 0, 0, 64, 4096, 1)
    seen = 0
    for batch in (1, 8):
        for arvind, cap, with_init in ((True, None, True), (True, 3, True), (True, 0, True), (True, 6, False), (False, 5, True),
                                       (False, None, True), (True, None, False)):
            h = ArvindDistanceOrdering() if arvind else DefaultBacktrackOrdering()
            d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth or None, prioritizePendingUponDivergence=True,
                                backtrackHeuristic=h, stopIfViolationFound=False, batch=batch, backend=oracle.dpor_batch)
            if with_init:
                d.setInitialTrace(init)
            h.init(d, init)
            if cap is not None:
                d.setMaxDistance(cap)
            rp = d.explore(ev, max_interleavings=budget)
            srch = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS, 0, T.DPOR_ORDERING_ARVIND if arvind else T.DPOR_ORDERING_DEFAULT, cap)
            nv, npl, nr, _, st = oracle_py.dpor_explore_ordered(model, ev, par, srch, original_trace=init if arvind else None,
                                                                initial_trace=init if with_init else None, n_threads=2)
            assert [int(x) for x in nr] == rp.rounds, (batch, arvind, cap, with_init)
            assert len(nv) == len(rp.interleavings)
            assert all(nv[k] == il.verdict and int(npl[k]) == il.prefix_len for k, il in enumerate(rp.interleavings))
            assert int(st.queue_len) == len(d.backTrack)
            if len(nv) < budget:
                assert bool(st.exhausted) == (rp.exhausted or (not d.backTrack and rp.aborted > 0))
            seen += len(nv)
    assert seen > (15 if which == "two_writers" else 500)


@pytest.mark.parametrize("which", ["two_writers", "raft3"])
def test_native_ordered_exploration_resumes_like_the_mirror(oracle, which):
    """ResumableDPOR's pattern on one instance: explore under a small distance cap, raise the cap, explore again, ... - a later
    call continues from the queue the earlier one left (one dequeued point first, DPORwHeuristics.scala:1219-1220), keeps the
    explored pairs, and restarts from the initial trace only when the queue is empty.  The native loop with
    demi_dpor_search.resume against the Python mirror, call by call."""
    from oracle import oracle_py
    if which == "two_writers":
        model = two_writers_model()
        ev = events_to_array([start(0), start(1), start(2), send(1, 0), send(2, 0), send(1, 0)])
        depth = 0
    else:
        model = M.raft_model(3)
        ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
        depth = 30
    _, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(60, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace)
    par = T.DporParams(depth, 0, 0, 0, 64, 4096, 1)
    for batch in (1, 4):
        h = ArvindDistanceOrdering()
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth or None, prioritizePendingUponDivergence=True,
                            backtrackHeuristic=h, stopIfViolationFound=False, batch=batch, backend=oracle.dpor_batch)
        d.setInitialTrace(init)
        h.init(d, init)
        state = oracle_py.OrderedState()
        total = 0
        for call, (cap, budget) in enumerate(((0, 50), (2, 50), (2, 50), (5, 40), (9, 60), (1 << 20, 150), (1 << 20, 150))):
            d.setMaxDistance(cap)
            rp = d.explore(ev, max_interleavings=budget)
            srch = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS, 0, T.DPOR_ORDERING_ARVIND, cap, resume=1)
            nv, npl, nr, _, st = oracle_py.dpor_explore_ordered(model, ev, par, srch, original_trace=init, initial_trace=init,
                                                                n_threads=2, state=state)
            assert [int(x) for x in nr] == rp.rounds, (batch, call, cap)
            assert len(nv) == len(rp.interleavings) and int(st.queue_len) == len(d.backTrack)
            assert all(nv[k] == il.verdict and int(npl[k]) == il.prefix_len for k, il in enumerate(rp.interleavings))
            total += len(nv)
        assert total >= (3 if which == "two_writers" else 200)


def _race4():
    MSGS = [("Go", T.MSG_EXTERNAL), ("Write", T.MSG_INTERNAL)]
    hnd = {(0, "Go"): Asm().mov(M.T0, 0).send(1, M.T0, M.ME, 0),
           (0, "Write"): Asm().mov(M.F[0], M.P0).add(M.F[1], M.F[1], 1)}
    model = build_model("race4", 4, MSGS, hnd, [[0] * 8] * 4, (T.INV_NEVER, 0, 1, 0))
    ev = events_to_array([start(0), start(1), start(2), start(3), send(2, 0), send(1, 0), send(3, 0), send(2, 0)])
    return model, ev


@pytest.mark.parametrize("which", ["race4", "race4_faults", "raft3"])
def test_native_edit_distance_dpor_ddmin_equals_the_python_mirror(oracle, which):
    """demi_edit_distance_dpor_ddmin's host loop (csrc/incddmin_host.hpp: IncrementalDDMin over ResumableDPOR, here around the
    oracle's interleavings) against the Python mirror's editDistanceDporDDMin over the same oracle: the same MCS, the same
    consultations - subsequence, verdict and distance cap, pass after pass -, the same (cap, MCS size) per pass, the same merged
    replay count, the same answer of verify_mcs; also with a small budget per internal exploration call (a consultation is then
    several resumed calls) and one-at-a-time launches."""
    from oracle import oracle_py
    if which.startswith("race4"):
        model, ev = _race4()
        if which == "race4_faults":      # faults and quiescence markers in the externals: not part of the minimization
            ev = events_to_array([start(0), start(1), start(2), start(3), partition(2, 3), send(2, 0), wait_quiescence(), send(1, 0),
                                  kill(3), send(2, 0)])
        want, stop_at, lim = True, 2, None
    else:
        model = M.raft_model(3)
        ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)] + [send(0, M.M_CLIENT)])
        want, stop_at, lim = True, 3, T.Limits(40, 0, 64, 0, 0, 0)
    v, trace = _execution(oracle, model, ev, want_violation=want, lim=lim)
    fp = ViolationFingerprint(v.fingerprint)
    init = dpor_initial_trace(trace)
    for batch, budget in ((8, 1 << 16), (1, 1 << 16), (4, 5)):
        mcs, dd, verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=stop_at, maxMaxDistance=8, batch=batch,
                                                     backend=oracle.dpor_batch)
        par = T.DporParams(0, len(init), 1, fp.code, 64, 4096, 1)
        ip = T.IncDdminParams(max_max_distance=8, stop_at_size=stop_at, check_unmodified=0, ignore_quiescence=1, verify_mcs=1, batch=batch,
                              budget=budget)
        n_mcs, n_cons, n_pass, n_vt, st = oracle_py.edit_distance_dpor_ddmin(model, ev, init, par, ip)
        assert tuple(n_mcs) == tuple(mcs), (which, batch, budget)
        assert n_pass == dd.distances
        assert [(tuple(c), p, d) for c, p, d in n_cons] == [(tuple(c), p, d) for c, p, d in dd.consulted_all]
        assert int(st.replays) == dd._stats.total_replays and int(st.consultations) == len(dd.consulted_all)
        assert int(st.instances) == len(dd.oracle.subseqToDPOR)
        if len(mcs) < len([i for i in range(len(ev)) if int(ev[i]["kind"]) in (T.EV_START, T.EV_SEND)]):
            assert (int(st.verified) == 1) == (verified is not None)
            if verified is not None:
                assert n_vt is not None and len(n_vt) == len(verified) and (n_vt["key"] == verified["key"]).all()
        else:
            assert int(st.verified) == -1 and verified is None
    assert len(mcs) < len(ev)
