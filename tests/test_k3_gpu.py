"""GPU parity suite for K3 (DPORwHeuristics interleavings + racing pairs): per-interleaving
outputs (verdict, trace, pair list) bit-exact vs the oracle, and whole explorations identical."""
import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import raft5_config3
from demi_amd.dpor import DPORwHeuristics
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence
from demi_amd.schedulers import SchedulerConfig, ViolationFingerprint

pytestmark = pytest.mark.gpu


def same_batch(g, c):
    gv, gt, gp = g
    cv, ct, cp = c
    assert (gv == cv).all(), (gv[gv != cv][:3], cv[gv != cv][:3])
    for a, b in zip(gt, ct):
        assert len(a) == len(b) and (a == b).all()
    for a, b in zip(gp, cp):
        assert len(a) == len(b) and (a == b).all()


launched_shared = []      # the shared lengths of the prefixes of the last collect_prefixes call


def collect_prefixes(oracle, model, ev, depth, batch, budget):
    """Run an oracle-backed exploration and keep every prefix it launched."""
    launched = []
    launched_shared.clear()

    def backend(m, e, prefixes, params, shared=None):
        launched.extend(prefixes)
        launched_shared.extend(shared if shared is not None else [0] * len(prefixes))
        return oracle.dpor_batch(m, e, prefixes, params, shared)

    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=batch, backend=backend)
    res = d.explore(ev, max_interleavings=budget)
    return launched, res, d


@pytest.mark.parametrize("cfg", ["raft3", "raft5", "quiescence"])
def test_per_interleaving_outputs_match_the_oracle(gpu_ctx, oracle, cfg):
    if cfg == "raft3":
        model = M.raft_model(3)
        ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    elif cfg == "raft5":
        model, ev, _ = raft5_config3()
    else:
        model = M.raft_model(3)
        ev = events_to_array([start(0), start(1), send(0, M.M_BOOTSTRAP), send(1, M.M_BOOTSTRAP), wait_quiescence(), start(2),
                              send(2, M.M_BOOTSTRAP), send(0, M.M_CLIENT, 1), wait_quiescence(), send(1, M.M_CLIENT, 2)])
    prefixes, res, _ = collect_prefixes(oracle, model, ev, 30, 32, 160)
    assert len(prefixes) >= 128
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.dpor_load(ev)
    for par in (T.DporParams(30, 0, 0, 0, 64, 4096), T.DporParams(12, 0, 0, 0, 64, 4096), T.DporParams(30, 40, 0, 0, 64, 64),
                T.DporParams(30, 0, 0, 0, 3, 4096), T.DporParams(30, 0, 1, 0x1000103, 64, 4096)):
        g = gpu_ctx.dpor_batch(prefixes, par)
        c = oracle.dpor_batch(model, ev, prefixes, par)
        same_batch(g, c)
    # prioritizePendingUponDivergence: prefixes with expected heads that are never pending (keys of other prefixes'
    # events spliced in), with and without the option
    rng = np.random.default_rng(5)
    noisy = []
    for k, pfx in enumerate(prefixes[:96]):
        other = prefixes[(k * 7 + 3) % len(prefixes)]
        if len(pfx) < 2 or len(other) < 2:
            noisy.append(pfx)
            continue
        cut = int(rng.integers(1, len(pfx)))
        junk = other[rng.integers(1, len(other), size=int(rng.integers(1, 4)))]
        noisy.append(np.concatenate([pfx[:cut], junk, pfx[cut:]]))
    for prio in (0, 1):
        par = T.DporParams(30, 0, 0, 0, 64, 4096, prio)
        same_batch(gpu_ctx.dpor_batch(noisy, par), oracle.dpor_batch(model, ev, noisy, par))
    # the kernel compiled for this model's table (demi_model_specialize; K3 is compiled at its first launch)
    gpu_ctx.model_specialize()
    for par in (T.DporParams(30, 0, 0, 0, 64, 4096), T.DporParams(30, 40, 0, 0, 64, 64)):
        same_batch(gpu_ctx.dpor_batch(prefixes, par), oracle.dpor_batch(model, ev, prefixes, par))
    gpu_ctx.model_specialize(False)
    # capacity verdicts appear and agree
    tiny = gpu_ctx.dpor_batch(prefixes[:64], T.DporParams(30, 0, 0, 0, 3, 4096))[0]
    assert (tiny["flags"] & T.V_PENDING_OVF).any()
    unbounded = gpu_ctx.dpor_batch(prefixes[:8], T.DporParams(0, 0, 0, 0, 64, 16))
    same_batch(unbounded, oracle.dpor_batch(model, ev, prefixes[:8], T.DporParams(0, 0, 0, 0, 64, 16)))


@pytest.mark.parametrize("batch", [1, 64])
def test_whole_exploration_identical_gpu_vs_oracle_backend(oracle, batch):
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(2)])
    budget = 100 if batch == 1 else 800
    dg = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=batch)
    rg = dg.explore(ev, max_interleavings=budget)
    dc = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=batch, backend=oracle.dpor_batch)
    rc = dc.explore(ev, max_interleavings=budget)
    assert rg.rounds == rc.rounds and rg.exhausted == rc.exhausted and rg.violations == rc.violations
    assert len(rg.interleavings) == len(rc.interleavings)
    for a, b in zip(rg.interleavings, rc.interleavings):
        assert a.verdict == b.verdict and (a.trace == b.trace).all() and a.prefix_len == b.prefix_len
    dg.shutdown()


def test_config3_bounded_exploration_raft5(oracle):
    """BASELINE config 3 (depth 30, Start x 5 + Bootstrap x 5), a budgeted slice of it: every explored
    interleaving is a distinct backtrack point, and sampled interleavings replay identically on the oracle."""
    model, ev, depth = raft5_config3()
    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=512)
    res = d.explore(ev, max_interleavings=1500)
    assert len(res.interleavings) == 1500 and not res.exhausted and len(res.rounds) <= 5
    assert len(res.schedule_hashes()) > 500
    for k in (0, 1, 700, 1499):
        il = res.interleavings[k]
        v, t, _ = oracle.dpor_batch(model, ev, [il.trace], T.DporParams(depth, 0, 0, 0, 64, 4096))
        assert v[0] == il.verdict and (t[0] == il.trace).all()
    d.shutdown()


def test_config5_shuffle8_bounded_dpor(oracle):
    """BASELINE config 5 (single GPU leg): bounded exhaustive DPOR over the 8-actor, 3-class shuffle job
    runs to an empty backtrack queue; the exploration is identical with the oracle backend."""
    from demi_amd.apps import shuffle8_config5
    model, ev, _, _ = shuffle8_config5()
    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=256)
    res = d.explore(ev, max_interleavings=4000)
    dc = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=256, backend=oracle.dpor_batch)
    rc = dc.explore(ev, max_interleavings=4000)
    assert res.exhausted and rc.exhausted and res.rounds == rc.rounds and res.violations == rc.violations
    assert len(res.interleavings) > 1000
    assert all(a.verdict == b.verdict and (a.trace == b.trace).all() for a, b in zip(res.interleavings, rc.interleavings))
    d.shutdown()


@pytest.mark.parametrize("batch,budget", [(1, 150), (64, 3000), (1024, 20000)])
def test_native_exploration_loop_equals_the_python_loop(batch, budget):
    """demi_dpor_explore (queue + explored set in C++) walks exactly the Python mirror's exploration."""
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(2)])
    if batch == 1024:
        model, ev, _ = raft5_config3(n_sends=3)
        budget = 4000
    dp = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=batch)
    rp = dp.explore(ev, max_interleavings=budget)
    dn = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=batch)
    rn = dn.explore_native(ev, max_interleavings=budget)
    assert rn.rounds == rp.rounds and rn.exhausted == rp.exhausted and rn.violations == rp.violations
    assert len(rn.interleavings) == len(rp.interleavings)
    assert all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rn.interleavings, rp.interleavings))
    dp.shutdown(); dn.shutdown()


def test_specialised_native_exploration_is_the_same_exploration():
    model, ev, depth = raft5_config3(n_sends=3)
    runs = []
    for spec in (False, True):
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=512, specialize=spec)
        r = d.explore_native(ev, max_interleavings=5000)
        assert d._ctx.is_specialized() == spec
        runs.append(r)
        d.shutdown()
    a, b = runs
    assert a.rounds == b.rounds and a.exhausted == b.exhausted and len(a.interleavings) == len(b.interleavings)
    assert all(x.verdict == y.verdict and x.prefix_len == y.prefix_len for x, y in zip(a.interleavings, b.interleavings))


def test_edit_distance_dpor_ddmin_on_the_gpu(oracle):
    """DDMin over DPOR with a growing edit-distance bound (IncrementalDDMin + ResumableDPOR + ArvindDistanceOrdering):
    every oracle consultation is a bounded K3 exploration; same MCS and consultation sequence as the same host
    logic over the CPU oracle."""
    from demi_amd.incremental_ddmin import editDistanceDporDDMin
    from demi_amd.schedulers import EventTrace
    from .test_incremental_ddmin_cpu import _execution
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)] +
                         [send(0, M.M_CLIENT, 1), send(1, M.M_CLIENT, 2)])
    v, trace = _execution(oracle, model, ev, lim=T.Limits(60, 0, 64, 0, 0, 0))
    fp = ViolationFingerprint(v.fingerprint, model.fp_match_mask)
    runs = []
    for backend, native in ((None, False), (oracle.dpor_batch, False), (None, True)):
        mcs, ddmin, verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=2,
                                                        maxMaxDistance=4, batch=64, backend=backend, native=native)
        runs.append((mcs, ddmin.ddmin.consulted, ddmin.distances, verified is not None, ddmin._stats.total_replays))
    assert runs[0] == runs[1]
    # ... and with every DPOR consultation inside the library (ArvindDistanceOrdering, the cap, the initial trace and the
    # resumable state of a subsequence consulted again at a larger distance): the same minimization
    assert runs[2] == runs[0]
    assert runs[0][3] and len(runs[0][0]) < len(ev)
    # ... and with the loop itself inside the library too (demi_edit_distance_dpor_ddmin: IncrementalDDMin over ResumableDPOR
    # in one call), interpreted and with the compiled table: the same MCS, consultations with their caps, passes, replays
    mcs, dd, verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=2, maxMaxDistance=4, batch=64,
                                                 backend=oracle.dpor_batch)
    for spec in (False, True):
        n_mcs, n_dd, n_verified, _ = editDistanceDporDDMin(SchedulerConfig(model=model), trace, fp, stopAtSize=2, maxMaxDistance=4, batch=64,
                                                           native_loop=True, specialize=spec)
        assert tuple(n_mcs) == tuple(mcs) and n_dd.distances == dd.distances
        assert n_dd.consulted_all == [(tuple(c), p, d) for c, p, d in dd.consulted_all]
        assert n_dd._stats.total_replays == dd._stats.total_replays
        assert (n_verified is not None) == (verified is not None)
        if verified is not None:
            assert len(n_verified) == len(verified) and (n_verified["key"] == verified["key"]).all()


def test_native_arvind_ordering_and_distance_cap_on_the_gpu(oracle):
    """demi_dpor_explore with ArvindDistanceOrdering / setMaxDistance / setInitialTrace (the plain loop of the reference around
    K3 launches, dpor_host.hpp explore_rounds_ordered) against the Python mirror's explore() over the CPU oracle: the same
    rounds, verdicts and prefix lengths."""
    from demi_amd.dpor import ArvindDistanceOrdering, DefaultBacktrackOrdering
    from demi_amd.incremental_ddmin import dpor_initial_trace
    from .test_incremental_ddmin_cpu import _execution
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    _, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(60, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace)
    seen = 0
    for batch, arvind, cap, with_init in ((1, True, None, True), (16, True, 4, True), (16, True, None, False), (16, False, 5, True),
                                          (64, True, None, True)):
        runs = []
        for backend in (oracle.dpor_batch, None):
            h = ArvindDistanceOrdering() if arvind else DefaultBacktrackOrdering()
            d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, prioritizePendingUponDivergence=True, backtrackHeuristic=h,
                                stopIfViolationFound=False, batch=batch, backend=backend, specialize=backend is None and batch == 64)
            if with_init:
                d.setInitialTrace(init)
            h.init(d, init)
            if cap is not None:
                d.setMaxDistance(cap)
            runs.append(d.explore(ev, max_interleavings=400) if backend is not None else d.explore_native(ev, max_interleavings=400))
            if backend is None:
                d.shutdown()
        rp, rn = runs
        assert rn.rounds == rp.rounds and len(rn.interleavings) == len(rp.interleavings), (batch, arvind, cap, with_init)
        assert all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rn.interleavings, rp.interleavings))
        seen += len(rn.interleavings)
    assert seen > 300


def test_native_test_continues_past_its_budget_instead_of_reporting_no_violation(oracle):
    """TestOracle.test of the native DPORwHeuristics: one library call explores at most native_budget interleavings; a call that
    stopped BECAUSE of the budget is continued from the queue the library kept (an ordered, resumable search) until that queue
    is empty - as one call with a sufficient budget ends (the two explore slightly different sets: a resumed call starts with
    one dequeued point, so the rounds are cut differently, and the explored-pair heuristic depends on the order of absorption) -
    and refused where nothing can be resumed (never read as "the subsequence does not reproduce")."""
    from demi_amd.dpor import ArvindDistanceOrdering
    from demi_amd.incremental_ddmin import dpor_initial_trace
    from demi_amd.schedulers import MinimizationStats, ViolationFingerprint
    from .test_incremental_ddmin_cpu import _execution
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    _, trace = _execution(oracle, model, ev, want_violation=False, lim=T.Limits(60, 0, 64, 0, 0, 0))
    init = dpor_initial_trace(trace)
    nothing = ViolationFingerprint(0x7FFFFFF1)        # a fingerprint no interleaving produces: test() explores to the end
    counts = []
    for budget in (1 << 16, 37):
        h = ArvindDistanceOrdering()
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, prioritizePendingUponDivergence=True, backtrackHeuristic=h,
                            stopIfViolationFound=True, batch=16, native=True)
        d.setInitialTrace(init)
        h.init(d, init)
        d.native_budget = budget
        st = MinimizationStats()
        assert d.test(ev, nothing, st) is None
        counts.append(st.total_replays)
        assert bool(d.last_native_stats.exhausted) and int(d.last_native_stats.queue_len) == 0      # explored to the end, not to a budget
        d.shutdown()
    assert counts[0] > 1000 and counts[1] > 1000 and abs(counts[0] - counts[1]) < counts[0] // 5, counts
    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=True, batch=16, native=True)
    d.native_budget = 37
    with pytest.raises(RuntimeError, match="native_budget"):
        d.test(ev, nothing, MinimizationStats())
    d.shutdown()


def test_dpor_golden_fixture_on_gpu(gpu_ctx):
    import hashlib
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raft3_dpor.npz"))
    prefixes = [z["prefixes"][k, :int(n)] for k, n in enumerate(z["prefix_len"])]
    gpu_ctx.model_load(M.raft_model(3).to_struct())
    gpu_ctx.dpor_load(z["externals"])
    dv, dt, dp = gpu_ctx.dpor_batch(prefixes, T.DporParams(30, 0, 0, 0, 64, 4096, 0))
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    assert (dv == z["verdicts"]).all()
    assert [sha(t) for t in dt] == list(z["trace_sha"]) and [sha(p) for p in dp] == list(z["pairs_sha"])


# ------------------------------------------------------------------ shared-prefix pair filter, reference order
def test_shared_prefix_pairs_are_not_reported_again(gpu_ctx, oracle):
    """demi_dpor_batch(shared_len): racing pairs whose later event lies in the take() part of the next trace are left
    out, identically on the kernel and on the oracle, and they are exactly the pairs the full list has there."""
    model, ev, _ = raft5_config3()
    prefixes, res, _ = collect_prefixes(oracle, model, ev, 30, 64, 400)
    shared = list(launched_shared)
    assert len(prefixes) == len(shared) and max(shared) > 20
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.dpor_load(ev)
    par = T.DporParams(30, 0, 0, 0, 64, 4096)
    g = gpu_ctx.dpor_batch(prefixes, par, shared)
    c = oracle.dpor_batch(model, ev, prefixes, par, shared)
    same_batch(g, c)
    full = gpu_ctx.dpor_batch(prefixes, par)
    for k in range(len(prefixes)):
        want = full[2][k][full[2][k]["later"] >= shared[k]]
        assert len(want) == len(g[2][k]) and (want == g[2][k]).all()
        assert (g[0][k] == full[0][k]) and (g[1][k] == full[1][k]).all()       # verdict and trace do not depend on it


@pytest.mark.parametrize("jit", [False, True])
def test_reference_order_on_the_gpu_is_the_batch1_sequence(oracle, jit):
    """demi_dpor_explore with DEMI_DPOR_ORDER_REFERENCE: the device speculates 256 wide, the committed sequence is the one
    the CPU oracle produces one backtrack point at a time (DPORwHeuristics' own loop); with a violating set that depends
    on the order (writers model) and with raft3."""
    from tests.test_dpor_cpu import native_explore, writers_model
    cases = [(M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)]), 30),
             (writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), 0)]
    for model, ev, depth in cases:
        one = native_explore(model, ev, T.DporParams(depth, 0, 0, 0, 64, 4096), 1, 20000)
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth or None, stopIfViolationFound=False, batch=256, specialize=jit)
        res = d.explore_native(ev, max_interleavings=20000, reference_order=True)
        got = np.array([il.verdict for il in res.interleavings], dtype=T.VERDICT_DTYPE)
        assert len(got) == len(one[0]) and (got == one[0]).all() and res.exhausted
        assert [il.prefix_len for il in res.interleavings] == [int(x) for x in one[1]]
        st = d.last_native_stats
        assert st.executed >= len(got) and (len(got) < 100 or st.launches < len(got) / 4)
        d.shutdown()


def test_config3_reference_order_matches_the_golden_batch1_record():
    """BASELINE config 3 to exhaustion in the reference's order on the GPU: the verdict sequence (60 332 interleavings)
    hashes to the record the CPU oracle produced at batch = 1 (tools/make_golden_dpor.py)."""
    import hashlib, json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dpor_config3_reference_order.json")) as f:
        want = json.load(f)
    model, ev, depth = raft5_config3()
    d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=4096, specialize=True)
    res = d.explore_native(ev, max_interleavings=1 << 17, reference_order=True)
    got = np.array([il.verdict for il in res.interleavings], dtype=T.VERDICT_DTYPE)
    plen = np.array([il.prefix_len for il in res.interleavings], dtype=np.uint32)
    assert len(got) == want["interleavings"] and res.exhausted == want["exhausted"]
    assert hashlib.sha256(got.tobytes()).hexdigest() == want["sha256_verdicts"]
    assert hashlib.sha256(plen.tobytes()).hexdigest() == want["sha256_prefix_lens"]
    assert len(res.violations) == want["violations"] and len(res.schedule_hashes()) == want["distinct_schedules"]
    d.shutdown()


def test_device_resident_bookkeeping_equals_the_host_bookkeeping_config3(monkeypatch):
    """BASELINE config 3 to exhaustion in ROUNDS order: explored set + enqueue decision on the device (the default) against
    the path that fetches every racing pair and absorbs it on the host (DEMI_DPOR_HOST_BOOKKEEPING): the same rounds and
    verdicts, a fraction of the bytes over PCIe."""
    model, ev, depth = raft5_config3()
    out = {}
    for host in (False, True):
        if host:
            monkeypatch.setenv("DEMI_DPOR_HOST_BOOKKEEPING", "1")
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth, stopIfViolationFound=False, batch=2048, specialize=True)
        res = d.explore_native(ev, max_interleavings=1 << 17)
        st = d.last_native_stats
        out[host] = (np.array([il.verdict for il in res.interleavings], dtype=T.VERDICT_DTYPE), res.rounds, res.exhausted,
                     int(st.d2h_bytes) + int(st.h2d_bytes))
        d.shutdown()
    assert len(out[False][0]) == len(out[True][0]) == 62902 and (out[False][0] == out[True][0]).all()
    assert out[False][1] == out[True][1] and out[False][2] and out[True][2]
    assert out[False][3] * 20 < out[True][3]


def test_reference_order_resident_equals_the_host_commit(monkeypatch):
    """REFERENCE order, both implementations on the GPU: the round-3 path - traces, the speculation's bookkeeping and the commit's
    pair filter on the device, the commit fed with 24-byte records (explore_reference_resident) - against the round-2 path that
    fetched every trace and every racing pair (DEMI_DPOR_HOST_BOOKKEEPING): the same committed sequence on a budgeted config 3,
    a fraction of the bytes over PCIe, and the first violating trace of the writers model."""
    from tests.test_dpor_cpu import writers_model
    model, ev, depth = raft5_config3()

    def run(model, ev, depth, budget, host, stop=False):
        if host:
            monkeypatch.setenv("DEMI_DPOR_HOST_BOOKKEEPING", "1")
        else:
            monkeypatch.delenv("DEMI_DPOR_HOST_BOOKKEEPING", raising=False)
        d = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=depth or None, stopIfViolationFound=stop, batch=1024, specialize=True)
        res = d.explore_native(ev, max_interleavings=budget, reference_order=True)
        got = np.array([il.verdict for il in res.interleavings], dtype=T.VERDICT_DTYPE)
        plen = [il.prefix_len for il in res.interleavings]
        st = d.last_native_stats
        out = (got, plen, int(st.d2h_bytes), int(st.executed), res, int(st.fetches))
        d.shutdown()
        return out
    new = run(model, ev, depth, 12000, host=False)
    old = run(model, ev, depth, 12000, host=True)
    assert len(new[0]) == 12000 and (new[0] == old[0]).all() and new[1] == old[1]
    assert new[2] * 20 < old[2], (new[2], old[2])                     # PCIe: records the commit can still use, fetched when it
    assert 0 < new[5] < 12000 and old[5] == 0                          # gets there, instead of traces + every pair
    for width in ("1", "4096"):                                        # whatever the fetch covers, the commit is the same
        monkeypatch.setenv("DEMI_DPOR_FETCH_WIDTH", width)
        w = run(model, ev, depth, 3000, host=False)
        assert (w[0] == old[0][:3000]).all() and w[1] == old[1][:3000]
    monkeypatch.delenv("DEMI_DPOR_FETCH_WIDTH")
    wm = writers_model(4)
    wev = events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)])
    a, b = run(wm, wev, 0, 5000, host=False, stop=True), run(wm, wev, 0, 5000, host=True, stop=True)
    assert (a[0] == b[0]).all() and a[1] == b[1] and len(a[4].violations) == len(b[4].violations) == 1
    ta, tb = a[4].interleavings[a[4].violations[0]].trace, b[4].interleavings[b[4].violations[0]].trace
    assert len(ta) == len(tb) and (np.asarray(ta) == np.asarray(tb)).all()


def _seq_digest(verdicts):
    """bench.py's order-sensitive digest of a verdict sequence (the `sequence_digest` of its dpor / config5 records)."""
    idx = np.arange(1, len(verdicts) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (verdicts["hash"] ^ (verdicts["flags"].astype(np.uint64) << np.uint64(32)) ^ verdicts["fingerprint"].astype(np.uint64)) * \
            (idx * np.uint64(0x9E3779B97F4A7C15) | np.uint64(1))
    return int(np.bitwise_xor.reduce(x)) if len(verdicts) else 0


def test_config5_pipeline_exploration_against_the_oracle(oracle):
    """BASELINE config 5 as bench.py times it (apps.shuffle8_config5_large: pipeline of three shuffle jobs, depth 40, budget
    2^20, ROUNDS of 16 384), the WHOLE exploration held against the CPU oracle in two ways:
      (1) its first 2^17 interleavings verdict for verdict against the oracle's exploration (same order, same prefixes);
      (2) more than 6 000 interleavings from everywhere in the 2^20 - every round's first and last member, the whole tail of the
          last round, the rest drawn at random - re-executed ONE BY ONE by the oracle (orc_dpor_execute) from the next trace the
          exploration started them from (demi_dpor_explored): the verdict the exploration returned, the trace it left in the arena
          and the racing pairs of that prefix must be the oracle's.  No host bookkeeping is shared in (2): the oracle sees a
          prefix and nothing else."""
    import os
    from demi_amd import _native
    from demi_amd.apps import shuffle8_config5_large
    emu = os.environ.get("DEMI_EMU") == "1"
    model, ev, depth, budget = shuffle8_config5_large()
    batch, head, n_random = 16384, 1 << 17, 6000
    if emu:
        budget, batch, head, n_random = 2500, 256, 800, 40
    par = T.DporParams(depth, 0, 0, 0, 64, 4096)
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    verdicts, plen, rounds, _vt, st = ctx.dpor_explore(par, T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS))
    n = len(verdicts)
    assert n == budget and not st.exhausted and int(rounds.sum()) == n
    if not emu:
        assert "%016x" % _seq_digest(verdicts) == "a8b790d52998a190"          # the sequence bench.py's config5 record reports
    # (1) the head of the exploration
    v, pl, _r, _t, _s, _secs = oracle.dpor_explore(model, ev, par, T.DporSearch(batch, head, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=os.cpu_count() or 1)
    assert len(v) == head and (v == verdicts[:head]).all() and (pl == plen[:head]).all()
    # (2) samples from the whole of it
    starts = np.concatenate([[0], np.cumsum(rounds)[:-1]]).astype(np.int64)
    ends = (np.cumsum(rounds) - 1).astype(np.int64)
    last_lo = int(starts[-1])
    rng = np.random.default_rng(20260922)
    pick = set(int(x) for x in starts) | set(int(x) for x in ends) | set(range(max(last_lo, n - 400), n)) | \
        set(int(x) for x in rng.integers(0, n, n_random))
    pick = sorted(pick)
    assert emu or (len(pick) >= 6000 and sum(1 for i in pick if i >= head) >= 5000 and sum(1 for i in pick if i >= last_lo) >= 300)
    prefixes, shared, traces = [], [], []
    for i in pick:
        nt, sh, tr = ctx.dpor_explored(i)
        assert len(nt) == plen[i] and (i == 0) == (len(nt) == 0)
        ov, otr, _opr = oracle.dpor_batch(model, ev, [nt], par, shared=[sh])
        assert ov[0] == verdicts[i], i
        assert len(otr[0]) == len(tr) and (otr[0] == tr).all(), i
        prefixes.append(nt); shared.append(sh); traces.append(tr)
    # the racing pairs of those prefixes (the exploration consumes them on the device: the same kernels on the same prefixes)
    for lo in range(0, len(pick), 512):
        gv, gt, gp = ctx.dpor_batch(prefixes[lo:lo + 512], par, shared[lo:lo + 512])
        cv, ct, cp = oracle.dpor_batch(model, ev, prefixes[lo:lo + 512], par, shared[lo:lo + 512])
        same_batch((gv, gt, gp), (cv, ct, cp))
        assert all((gv[k] == verdicts[pick[lo + k]]) and (gt[k] == traces[lo + k]).all() for k in range(len(gv)))
    ctx.close()


def test_one_job_shuffle_in_reference_order_equals_the_scala_transliteration(oracle):
    """A bookkeeping check that shares NOTHING with the product's host loops: the literal transliteration of DPORwHeuristics
    (tests/test_dpor_scheduler_transliteration_cpu.py - its own id counter, dependency graph, backtrack PriorityQueue and
    ExploredTacker, Python containers) explores the one-job shuffle8 table (config 5's application, 1 399 interleavings in the
    reference's own order) to exhaustion, and the GPU exploration in DEMI_DPOR_ORDER_REFERENCE must return the same sequence:
    every verdict (the hash covers every delivery and final state), every next-trace length, exhaustion."""
    import os
    from demi_amd import _native
    from demi_amd.apps import shuffle8_config5
    from tests.test_dpor_scheduler_transliteration_cpu import ScalaDPORwHeuristics
    emu = os.environ.get("DEMI_EMU") == "1"
    model, ev, _f, _l = shuffle8_config5()
    cap = 120 if emu else 4000
    sc = ScalaDPORwHeuristics(oracle, model, ev, depth_bound=40, max_messages=0)
    exhausted = sc.run(cap)
    want = np.array(sc.verdicts, dtype=T.VERDICT_DTYPE)
    assert emu or (exhausted and len(want) == 1399)
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    par = T.DporParams(40, 0, 0, 0, 64, 4096)
    for batch in (256, 1):          # the device speculating 256 wide, and the plain one-at-a-time loop
        got, plen, _rounds, _vt, st = ctx.dpor_explore(par, T.DporSearch(batch, cap, 0, 1, T.DPOR_ORDER_REFERENCE))
        assert len(got) == len(want) and bool(st.exhausted) == exhausted
        assert (got == want).all() and [int(x) for x in plen] == sc.next_trace_lens
        if batch == 1:
            break
        if emu:
            break
    # and what the exploration says it started each interleaving from is what the transliteration's getNext() built
    nt, sh, tr = ctx.dpor_explored(len(want) - 1)
    assert len(nt) == sc.next_trace_lens[-1]
    ctx.close()


def test_reference_order_with_the_record_fetch_in_flight(monkeypatch):
    """DEMI_DPOR_PREFETCH=1: the commit's record fetch for the next window of its queue front is issued while this window is
    absorbed (dpor_host.hpp issue() / land()); records fetched early are filtered under an older table, the committed sequence is
    the same - raft5 (config 3, a budgeted slice) and the writers model, whose violating set depends on the order."""
    from tests.test_dpor_cpu import writers_model
    from demi_amd import _native
    model3, ev3, depth3 = raft5_config3()
    cases = [(model3, ev3, depth3, 6000), (writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), 0, 20000)]
    for model, ev, depth, budget in cases:
        ctx = _native.Context(0)
        ctx.model_load(model.to_struct())
        ctx.dpor_load(ev)
        par, srch = T.DporParams(depth, 0, 0, 0, 64, 4096), T.DporSearch(256, budget, 0, 1, T.DPOR_ORDER_REFERENCE)
        monkeypatch.delenv("DEMI_DPOR_PREFETCH", raising=False)
        one = ctx.dpor_explore(par, srch)
        monkeypatch.setenv("DEMI_DPOR_PREFETCH", "1")
        two = ctx.dpor_explore(par, srch)
        monkeypatch.delenv("DEMI_DPOR_PREFETCH")
        assert len(one[0]) == len(two[0]) and (one[0] == two[0]).all() and (one[1] == two[1]).all()
        assert one[4].exhausted == two[4].exhausted and one[4].queue_len == two[4].queue_len
        assert two[4].d2h_bytes >= one[4].d2h_bytes        # (records fetched a window early: never fewer)
        ctx.close()


@pytest.mark.parametrize("cfg", ["config3", "config5", "config3_bug", "config5_bug"])
def test_device_queue_and_parent_filter_leave_the_exploration_unchanged(monkeypatch, cfg):
    """Round 5's two changes of the ROUNDS path - the backtrack queue on the device (k3_queue.hpp) and the parent filter in
    k3_pairs_insert - against round 4's loop (DEMI_DPOR_HOST_QUEUE: live points and kills to a host queue) and against probing
    every reported pair (DEMI_K3_NO_PARENT_FILTER): the same rounds, verdicts, prefix lengths, queue and backtrack-point count."""
    import os
    from demi_amd import _native
    from demi_amd.apps import shuffle8_config5_large
    emu = os.environ.get("DEMI_EMU") == "1"
    par = None
    if cfg == "config3":
        model, ev, depth = raft5_config3()
        budget, batch = (3000, 256) if emu else (1 << 17, 16384)
    elif cfg == "config5":
        model, ev, depth, _b = shuffle8_config5_large()
        budget, batch = (2000, 256) if emu else (100000, 16384)
    elif cfg == "config3_bug":          # round 6's workloads: prioritizePendingUponDivergence, violating interleavings in the rounds
        from demi_amd.apps import raft5_dpor_config3
        model, ev, par = raft5_dpor_config3()
        budget, batch = (3000, 256) if emu else (100000, 16384)
    else:
        from demi_amd.apps import shuffle8_dpor_config5
        model, ev, par, _b = shuffle8_dpor_config5()
        budget, batch = (2000, 256) if emu else (100000, 16384)
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    par = par or T.DporParams(depth, 0, 0, 0, 64, 4096)
    srch = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
    runs = {}
    for name, env in (("default", {}), ("host_queue", {"DEMI_DPOR_HOST_QUEUE": "1"}), ("no_filter", {"DEMI_K3_NO_PARENT_FILTER": "1"}),
                      ("round4", {"DEMI_DPOR_HOST_QUEUE": "1", "DEMI_K3_NO_PARENT_FILTER": "1"})):
        for k in ("DEMI_DPOR_HOST_QUEUE", "DEMI_K3_NO_PARENT_FILTER"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        runs[name] = ctx.dpor_explore(par, srch)
    for k in ("DEMI_DPOR_HOST_QUEUE", "DEMI_K3_NO_PARENT_FILTER"):
        monkeypatch.delenv(k, raising=False)
    d = runs["default"]
    assert len(d[0]) == budget or d[4].exhausted
    if cfg == "config5_bug" and not emu:
        assert (d[0]["flags"] & T.V_VIOLATION).any()
    for name, r in runs.items():
        assert len(r[0]) == len(d[0]) and (r[0] == d[0]).all() and (r[1] == d[1]).all() and (r[2] == d[2]).all(), name
        assert r[4].queue_len == d[4].queue_len and r[4].backtrack_points == d[4].backtrack_points and r[4].exhausted == d[4].exhausted, name
    # what no longer crosses PCIe: the host queue is fed 40-byte points and 16-byte kills, the device queue 256 run lengths per round
    assert d[4].d2h_bytes < runs["host_queue"][4].d2h_bytes
    ctx.close()


@pytest.mark.parametrize("case", ["late_start_two_periods", "two_periods", "writers", "config3", "config5", "config3_bug", "config5_bug"])
def test_checkpointed_interleavings_are_the_same_interleavings(oracle, monkeypatch, case):
    """k3_dpor starts an interleaving from a record of its parent's state at or below the branch point (K3Snap) instead of
    re-executing the shared prefix.  Against the same exploration without records (DEMI_K3_NO_CHECKPOINT) and against the CPU
    oracle's exploration: every verdict (the hash covers every delivery and every final state), every next-trace length, the
    queue; and the traces left in the arena entry by entry.  `late_start_two_periods` has actors that are started in a later
    quiescent period - messages to them are discarded without a trace entry, the step after which an interleaving must stop
    leaving records (so early here that the case runs without any) - and a quiescence marker in the middle of the trace;
    `two_periods` has the marker and no isolated actor: records on both sides of it."""
    import os
    from demi_amd import _native
    from demi_amd.apps import shuffle8_config5_large
    from tests.test_dpor_cpu import writers_model
    emu = os.environ.get("DEMI_EMU") == "1"
    if case == "late_start_two_periods":
        model, depth = M.raft_model(3), 40
        ev = events_to_array([start(0), start(1), send(0, M.M_BOOTSTRAP), send(1, M.M_BOOTSTRAP), wait_quiescence(), start(2),
                              send(2, M.M_BOOTSTRAP), send(0, M.M_BOOTSTRAP)])
        budget, batch = (1500, 128) if emu else (20000, 1024)
    elif case == "two_periods":
        model, depth = M.raft_model(3), 36
        ev = events_to_array([start(a) for a in range(3)] + [send(0, M.M_BOOTSTRAP), wait_quiescence(), send(1, M.M_BOOTSTRAP), send(2, M.M_BOOTSTRAP)])
        budget, batch = (1500, 128) if emu else (20000, 1024)
    elif case == "writers":
        model, depth = writers_model(4), 0
        ev = events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)])
        budget, batch = (2500, 128) if emu else (20000, 512)
    elif case == "config3":
        model, ev, depth = raft5_config3()
        budget, batch = (1500, 256) if emu else (1 << 17, 16384)
    elif case == "config5":
        model, ev, depth, _b = shuffle8_config5_large()
        budget, batch = (1200, 256) if emu else (60000, 16384)
    elif case == "config3_bug":         # (prioritizePendingUponDivergence: expected heads that are skipped, before and after a record)
        from demi_amd.apps import raft5_dpor_config3
        model, ev, par3 = raft5_dpor_config3()
        budget, batch = (1500, 256) if emu else (60000, 16384)
    else:
        from demi_amd.apps import shuffle8_dpor_config5
        model, ev, par3, _b = shuffle8_dpor_config5()
        budget, batch = (1200, 256) if emu else (60000, 16384)
    par = par3 if case.endswith("_bug") else T.DporParams(depth, 0, 0, 0, 64, 4096)
    srch = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    monkeypatch.delenv("DEMI_K3_NO_CHECKPOINT", raising=False)
    with_records = ctx.dpor_explore(par, srch)
    n = len(with_records[0])
    rng = np.random.default_rng(3)
    probe = sorted(set([0, n - 1] + [int(x) for x in rng.integers(0, n, 40)]))
    traces = [ctx.dpor_explored(i) for i in probe]
    monkeypatch.setenv("DEMI_K3_NO_CHECKPOINT", "1")
    without = ctx.dpor_explore(par, srch)
    plain = [ctx.dpor_explored(i) for i in probe]
    monkeypatch.delenv("DEMI_K3_NO_CHECKPOINT")
    assert len(without[0]) == n and (with_records[0] == without[0]).all() and (with_records[1] == without[1]).all()
    assert (with_records[2] == without[2]).all() and with_records[4].queue_len == without[4].queue_len
    assert with_records[4].exhausted == without[4].exhausted and with_records[4].backtrack_points == without[4].backtrack_points
    for (nt_a, sh_a, tr_a), (nt_b, sh_b, tr_b) in zip(traces, plain):
        assert sh_a == sh_b and len(nt_a) == len(nt_b) and (nt_a == nt_b).all() and len(tr_a) == len(tr_b) and (tr_a == tr_b).all()
    # ... and the oracle's exploration (its interleavings run from scratch, by construction)
    cpu = oracle.dpor_explore(model, ev, par, T.DporSearch(batch, min(budget, 6000 if not emu else budget), 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=os.cpu_count() or 1)
    m = len(cpu[0])
    assert (cpu[0] == with_records[0][:m]).all() and (cpu[1] == with_records[1][:m]).all()
    ctx.close()


@pytest.mark.parametrize("max_pairs,stop,batch", [(48, 0, 64), (4096, 1, 128), (4096, 0, 1), (300, 0, 7)])
def test_device_queue_edge_cases_against_the_host_bookkeeping(monkeypatch, max_pairs, stop, batch):
    """The device-resident path (queue, parent filter, checkpoints) where its special cases live: racing-pair lists that overflow
    max_pairs (DEMI_V_PAIRS_OVF ends the parent filter's invariant for the descendants), stopIfViolationFound, rounds of one
    and of an odd size, a budget that is not a multiple of the batch - against the path that fetches every racing pair and does
    dpor()'s bookkeeping on host threads (DEMI_DPOR_HOST_BOOKKEEPING), on a model with violations (writers) and on raft3."""
    from demi_amd import _native
    from tests.test_dpor_cpu import writers_model
    cases = [(writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), 0, 1801),
             (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)]), 30, 1203)]
    for model, ev, depth, budget in cases:
        if batch == 1:
            budget = 160
        ctx = _native.Context(0)
        ctx.model_load(model.to_struct())
        ctx.dpor_load(ev)
        par, srch = T.DporParams(depth, 0, 0, 0, 64, max_pairs), T.DporSearch(batch, budget, stop, 1, T.DPOR_ORDER_ROUNDS)
        monkeypatch.delenv("DEMI_DPOR_HOST_BOOKKEEPING", raising=False)
        dev = ctx.dpor_explore(par, srch)
        monkeypatch.setenv("DEMI_DPOR_HOST_BOOKKEEPING", "1")
        host = ctx.dpor_explore(par, srch)
        monkeypatch.delenv("DEMI_DPOR_HOST_BOOKKEEPING")
        assert len(dev[0]) == len(host[0]) and (dev[0] == host[0]).all() and (dev[1] == host[1]).all() and (dev[2] == host[2]).all()
        assert dev[4].exhausted == host[4].exhausted and dev[4].violations == host[4].violations and dev[4].first_violation == host[4].first_violation
        assert len(dev[3]) == len(host[3]) and (dev[3] == host[3]).all()                 # the first violating trace
        if max_pairs == 48 and depth == 30:                       # (raft3: hundreds of racing pairs per interleaving)
            assert (dev[0]["flags"] & T.V_PAIRS_OVF).any()
        if stop:
            assert (dev[4].violations >= 1) == bool((dev[0]["flags"] & T.V_VIOLATION).any())
        ctx.close()


@pytest.mark.gpu
def test_a_round_with_more_points_than_the_staging_area_is_sorted_again(monkeypatch):
    """The device-resident queue stages a round's backtrack points in an area sized for the largest round so far (2^21 points
    to begin with); a round that produces more - config 5 past 6 million interleavings leaves 2.3 million in one round - is
    sorted again into a larger area, from the decision the round has already made.  Here the area starts at two points
    (DEMI_K3_POINTS_CAP), so it is outgrown many times over: the exploration is the one with the default area."""
    from demi_amd import _native
    from tests.test_dpor_cpu import writers_model
    cases = [(writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), 0, 900, 64),
             (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)]), 30, 700, 37)]
    monkeypatch.setenv("DEMI_EXPERIMENT", "1")
    for model, ev, depth, budget, batch in cases:
        par, srch = T.DporParams(depth, 0, 0, 0, 64, 4096), T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
        res = []
        for cap in (None, "2"):
            if cap is None:
                monkeypatch.delenv("DEMI_K3_POINTS_CAP", raising=False)
            else:
                monkeypatch.setenv("DEMI_K3_POINTS_CAP", cap)
            ctx = _native.Context(0)                     # (the area belongs to the context: a fresh one per setting)
            ctx.model_load(model.to_struct())
            ctx.dpor_load(ev)
            res.append(ctx.dpor_explore(par, srch))
            ctx.close()
        a, b = res
        assert len(a[0]) == len(b[0]) > 5 and (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
        assert a[4].backtrack_points == b[4].backtrack_points > 2 and a[4].queue_len == b[4].queue_len and a[4].exhausted == b[4].exhausted


@pytest.mark.gpu
def test_config5_pipeline_in_reference_order_is_the_one_at_a_time_sequence(oracle):
    """The three-job shuffle pipeline of config 5 (8 actors, 3 classes, ~1 800 racing pairs per interleaving) in the REFERENCE
    order with a budget: the device-resident commit (speculation in rounds, record fetches, the commit filter's two passes)
    returns the verdicts and prefix lengths of the oracle's exploration one backtrack point at a time, for two widths of the
    speculation - a larger table than config 3's and a shape the commit was not tuned on."""
    import os
    from demi_amd import _native
    from demi_amd.apps import shuffle8_config5_large
    emu = os.environ.get("DEMI_EMU") == "1"
    model, ev, depth, _budget = shuffle8_config5_large()
    budget = 300 if emu else 6000
    par = T.DporParams(depth, 0, 0, 0, 64, 4096)
    one = oracle.dpor_explore(model, ev, par, T.DporSearch(1, budget, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    assert len(one[0]) == budget
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(ev)
    import hashlib
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dpor_config5_transliteration.json")) as f:
        translit = json.load(f)          # (the same 6 000 interleavings by the Python transliteration of the Scala scheduler)
    for batch in ((64,) if emu else (1024, 16384)):
        v, pl, _r, _vt, st = ctx.dpor_explore(par, T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_REFERENCE))
        assert len(v) == budget and (v == one[0]).all() and (pl == one[1]).all(), batch
        assert not st.exhausted and int(st.fetches) >= 1 and int(st.executed) >= budget
        if budget == translit["interleavings"]:
            assert hashlib.sha256(np.ascontiguousarray(v, dtype=T.VERDICT_DTYPE).tobytes()).hexdigest() == translit["sha256_verdicts"]
            assert hashlib.sha256(np.ascontiguousarray(pl, dtype=np.uint32).tobytes()).hexdigest() == translit["sha256_prefix_lens"]
    ctx.close()


@pytest.mark.gpu
def test_reference_order_when_the_speculations_table_is_full(oracle, monkeypatch):
    """In the REFERENCE order the device's explored-pair table only steers the speculation (which interleavings to run before the
    commit asks for them); the commit has its own.  A budgeted exploration can outrun a table sized for the budget - config 5
    with a budget of 6 000 did -: the speculation then stops, its table's kernels are left out of the launches, and the commit
    goes on from its own queue's front.  Here the table has 64 entries (DEMI_K3_TABLE_ENTRIES): the committed sequence is the
    one-at-a-time exploration's all the same."""
    from demi_amd import _native
    model = M.raft_model(3)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    par, budget = T.DporParams(30, 0, 0, 0, 64, 4096), 900
    one = oracle.dpor_explore(model, ev, par, T.DporSearch(1, budget, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    monkeypatch.setenv("DEMI_EXPERIMENT", "1")
    monkeypatch.setenv("DEMI_K3_TABLE_ENTRIES", "64")
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.dpor_load(ev)
    for batch in (16, 256):
        v, pl, _r, _vt, st = ctx.dpor_explore(par, T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_REFERENCE))
        assert len(v) == len(one[0]) and (v == one[0]).all() and (pl == one[1]).all(), batch
        assert bool(st.exhausted) == bool(one[4].exhausted)
    # (the ROUNDS order NEEDS that table: there a full one ends the exploration with an error)
    with pytest.raises(_native.DemiError, match="table full"):
        ctx.dpor_explore(par, T.DporSearch(64, budget, 0, 1, T.DPOR_ORDER_ROUNDS))
    ctx.close()


@pytest.mark.gpu
def test_the_device_queues_pool_is_compacted_without_changing_the_exploration(monkeypatch):
    """The device-resident backtrack queue used to keep every point ever emitted (24 B each, advisor's finding of round 5): once
    more than half of the pool is dequeued points its live runs now move to the front of a fresh pool.  With the threshold at 64
    points (DEMI_K3_POOL_COMPACT_MIN) that happens again and again: the exploration - verdicts, next-trace lengths, rounds, queue - is
    the one without any compaction, on the config-3 workload that finds the seeded bug, in small and in large rounds."""
    from demi_amd import _native
    from demi_amd.apps import raft5_dpor_config3
    emu = __import__("os").environ.get("DEMI_EMU") == "1"
    m3, e3, p3 = raft5_dpor_config3()
    cases = [(m3, e3, p3, 1500, 8), (m3, e3, p3, 3000 if emu else 40000, 32 if emu else 2048)]
    monkeypatch.setenv("DEMI_EXPERIMENT", "1")
    for model, ev, par, budget, batch in cases:
        srch = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
        res = []
        for thr in ("1000000000", "64"):
            monkeypatch.setenv("DEMI_K3_POOL_COMPACT_MIN", thr)
            ctx = _native.Context(0)
            ctx.model_load(model.to_struct())
            ctx.model_specialize()
            ctx.dpor_load(ev)
            res.append(ctx.dpor_explore(par, srch))
            ctx.close()
        monkeypatch.delenv("DEMI_K3_POOL_COMPACT_MIN")
        a, b = res
        assert int(a[4].fetches) == 0 and int(b[4].fetches) >= 2              # (compactions are counted in stats.fetches in this order)
        assert len(a[0]) == len(b[0]) > 5 and (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
        assert a[4].backtrack_points == b[4].backtrack_points and a[4].queue_len == b[4].queue_len and a[4].exhausted == b[4].exhausted
