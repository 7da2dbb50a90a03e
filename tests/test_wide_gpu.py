"""GPU parity suite for DEMI_MODEL_WIDE tables (16 x u16 register window, include/demi_gpu.h): the kernel compiled from
the table against the oracle's wide row interpreter, through the C ABI.  Bit-exact bar as for the 8-bit window: the
16-byte verdict incl. the hash over every delivered 64-bit message word and both final state words of every actor."""
import os

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence

from .test_k1_gpu import assert_same

pytestmark = pytest.mark.gpu


def wide_both(ctx, oracle, model, events, n, lim, seed_base=SEED_BASE):
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    ctx.model_specialize()
    assert ctx.is_specialized()
    g = ctx.random_explore(n, lim, seed_base=seed_base)
    c = oracle.random_explore(model, events, n, seed_base=seed_base, limits=lim, n_threads=os.cpu_count())
    return g, c


@pytest.mark.parametrize("p_max", [32, 64])
def test_wide_raft5_parity_on_the_bench_trace(oracle, p_max):
    """The bench workload's trace under a raft whose terms start at 1000 and whose logs start at 300 entries."""
    _, events, lim = raft5_config2()
    lim.p_max = p_max
    model = M.raft_model(5, term0=1000, loglen0=300)
    assert model.wide
    ctx = _native.Context(0)
    try:
        g, c = wide_both(ctx, oracle, model, events, 50000, lim)
        assert_same(g, c)
        viol = g[(g["flags"] & T.V_VIOLATION) != 0]
        assert len(viol) > 100 and (((viol["fingerprint"] >> 8) & 0xFFFF) > 1000).all()     # two leaders in a term above 1000
        assert len(np.unique(g["hash"])) > 40000
        if p_max == 32:
            assert (g["flags"] & T.V_PENDING_OVF).sum() > 0
        # same protocol as the 8-bit model: same schedules, same delivery counts, other hashes
        narrow = M.raft_model(5)
        ctx.model_load(narrow.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        gn = ctx.random_explore(50000, lim, seed_base=SEED_BASE)
        ok = ((g["flags"] | gn["flags"]) & (T.V_PENDING_OVF | T.V_QUEUE_OVF)) == 0
        assert (T.verdict_deliveries(g["flags"][ok]) == T.verdict_deliveries(gn["flags"][ok])).all()
        assert (g["hash"][ok] != gn["hash"][ok]).all()
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [2, 3, 7])
def test_wide_random_tables_parity(oracle, seed):
    """Random wide tables (every op, MOVHI / 16-bit constants, timers, RND, 16-bit external payloads, quiescence markers)."""
    from .test_jit_cpu import _random_handler_wide
    rng = np.random.default_rng(seed)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for cls in range(2):
        for name, _ in MSGS:
            if rng.integers(5):
                h[(cls, name)] = _random_handler_wide(rng, int(rng.integers(3, 30)), len(MSGS))
    model = M.build_model("rand_wide%d" % seed, 5, MSGS, h, [[int(x) for x in rng.integers(0, 65536, 8)] for _ in range(5)],
                          (T.INV_NEVER, 0, 40000, 0), actor_class=[0, 1, 0, 1, 1], n_classes=2, wide=True)
    ev = [start(a) for a in range(5)]
    for i in range(40):
        ev.append(wait_quiescence() if rng.integers(0, 7) == 0 and ev[-1][0] != T.EV_WAIT_QUIESCENCE
                  else send(int(rng.integers(0, 5)), 0, int(rng.integers(0, 65536)), int(rng.integers(0, 65536))))
    ctx = _native.Context(0)
    try:
        g, c = wide_both(ctx, oracle, model, events_to_array(ev), 4000, T.Limits(150, 9, 64, 0, 0, 0))
        assert_same(g, c)
        assert len(np.unique(g["hash"])) > 500
    finally:
        ctx.close()


def test_wide_dpor_parity(oracle):
    """DPORwHeuristics over a wide table (K3 compiled with 64-bit message words): per-interleaving verdicts, traces and racing
    pairs against the oracle, the whole exploration with the oracle as backend, the native loop in both orders.  A trace entry
    reports the low half of the message word (include/demi_gpu.h); node keys hash the whole word."""
    from demi_amd.dpor import DPORwHeuristics
    from demi_amd.schedulers import SchedulerConfig
    from .test_k3_gpu import collect_prefixes, same_batch
    model = M.raft_model(3, term0=1000, loglen0=300)
    assert model.wide
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    prefixes, res, _ = collect_prefixes(oracle, model, ev, 30, 32, 160)
    assert len(prefixes) >= 128
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.dpor_load(ev)
        with pytest.raises(_native.DemiError, match="compiled table"):
            ctx.dpor_batch(prefixes[:4], T.DporParams(30, 0, 0, 0, 64, 4096))
        ctx.model_specialize()
        for par in (T.DporParams(30, 0, 0, 0, 64, 4096), T.DporParams(12, 0, 0, 0, 64, 4096), T.DporParams(30, 40, 0, 0, 64, 64),
                    T.DporParams(30, 0, 0, 0, 3, 4096)):
            same_batch(ctx.dpor_batch(prefixes, par), oracle.dpor_batch(model, ev, prefixes, par))
    finally:
        ctx.close()
    # same protocol as the 8-bit model: the same tree of interleavings, other keys and hashes
    narrow = collect_prefixes(oracle, M.raft_model(3), ev, 30, 32, 160)[1]
    assert res.rounds == narrow.rounds and [i.prefix_len for i in res.interleavings] == [i.prefix_len for i in narrow.interleavings]
    assert res.schedule_hashes().isdisjoint(narrow.schedule_hashes())
    # whole explorations: Python loop on the GPU vs the oracle as backend, then the native loop in both orders
    dg = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=64, specialize=True)
    rg = dg.explore(ev, max_interleavings=800)
    dc = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=64, backend=oracle.dpor_batch)
    rc = dc.explore(ev, max_interleavings=800)
    assert rg.rounds == rc.rounds and len(rg.interleavings) == len(rc.interleavings) == 800
    assert all(a.verdict == b.verdict and (a.trace == b.trace).all() for a, b in zip(rg.interleavings, rc.interleavings))
    dn = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=64, specialize=True)
    rn = dn.explore_native(ev, max_interleavings=800)
    assert rn.rounds == rg.rounds and all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rn.interleavings, rg.interleavings))
    d1 = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=1, backend=oracle.dpor_batch)
    r1 = d1.explore(ev, max_interleavings=300)
    dr = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=30, stopIfViolationFound=False, batch=256, specialize=True)
    rr = dr.explore_native(ev, max_interleavings=300, reference_order=True)
    assert len(rr.interleavings) == 300 and all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rr.interleavings, r1.interleavings))
    dg.shutdown(); dn.shutdown(); dr.shutdown()
    # violations found by the exploration (the invariant reads 16-bit fields): writers whose ids start at 1000
    from demi_amd.model import Asm, build_model
    k = 4
    MSGS = [("Go", T.MSG_EXTERNAL), ("Write", T.MSG_INTERNAL)]
    h = {(0, "Go"): Asm().ldi16(M.T0, 1000).add(M.T0, M.T0, M.ME).mov(M.T1, 0).send(1, M.T1, M.T0, 0),
         (0, "Write"): Asm().mov(M.F[0], M.P0).add(M.F[1], M.F[1], 1)}
    wm = build_model("wide_writers", k + 1, MSGS, h, [[0] * 8] * (k + 1), (T.INV_NEVER, 0, 1001, 0), wide=True)
    wev = events_to_array([start(a) for a in range(k + 1)] + [send(a, 0) for a in range(1, k + 1)])
    dw = DPORwHeuristics(SchedulerConfig(model=wm), stopIfViolationFound=False, batch=16, specialize=True)
    rw = dw.explore(wev, max_interleavings=400)
    do = DPORwHeuristics(SchedulerConfig(model=wm), stopIfViolationFound=False, batch=16, backend=oracle.dpor_batch)
    ro = do.explore(wev, max_interleavings=400)
    assert rw.rounds == ro.rounds and rw.violations == ro.violations and 0 < len(rw.violations) < len(rw.interleavings)
    assert all(a.verdict == b.verdict and (a.trace == b.trace).all() for a, b in zip(rw.interleavings, ro.interleavings))
    dw.shutdown()


def test_wide_record_replay_minimize(oracle):
    """The rest of the pipeline on a wide table: the recorded EventTrace of a violating execution (16-bit payloads in
    demi_rec_event), STSScheduler replays of candidate subsequences, removal candidates and kept marks (K2 compiled with 64-bit
    message words, the scanning variant), DDMin and the internal minimization end to end - everything against the oracle."""
    from demi_amd.internal_minimization import STSSchedMinimizer
    from demi_amd.minification import stsSchedDDMin
    from demi_amd.schedulers import EventTrace, STSScheduler, SchedulerConfig, ViolationFingerprint
    from tests.test_minification_cpu import OracleSTS
    from .test_k2_gpu import random_masks
    _, events, lim = raft5_config2()
    model = M.raft_model(5, term0=1000, loglen0=300)
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
        hits = np.nonzero(v["flags"] & T.V_VIOLATION)[0]
        rng = np.random.default_rng(11)
        for k in (int(hits[0]), int(hits[3])):
            vv, rec = ctx.random_get_trace(SEED_BASE + k, lim)
            cv, crec, _ = oracle.random_execute(model, events, SEED_BASE + k, lim, record=True)
            assert vv.flags == cv.flags == v["flags"][k] and vv.hash == cv.hash == v["hash"][k] and vv.fingerprint == cv.fingerprint
            assert len(rec) == len(crec) and (rec == crec).all() and int(rec["p0"].max()) >= 1000
            used = events[:T.verdict_trace_idx(vv.flags)]
            target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
            ctx.replay_load(used, rec)
            masks = random_masks(rng, len(used), 1500)
            g = ctx.replay_batch(masks, target)
            c = oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count())
            assert_same(g, c)
            assert g[0]["flags"] & T.V_VIOLATION and not g[0]["flags"] & T.V_DIVERGED and int(g[0]["hash"]) == vv.hash
            assert (g["flags"] & T.V_VIOLATION).sum() > 1 and (g["flags"] & T.V_DIVERGED).sum() > 100
            for fka in (T.FILTER_ABSENTS_LITERAL, T.FILTER_ABSENTS_CORRECTED):
                tl = T.Limits(0, 0, 64, 1, vv.fingerprint, 0, 0, fka)
                assert_same(ctx.replay_batch(masks[:300], tl), oracle.sts_replay_batch(model, used, rec, masks[:300], tl))
            skips = np.nonzero(rec["kind"] == T.REC_MSG_EVENT)[0].astype(np.uint32)
            assert_same(ctx.replay_removal_batch(skips, target), oracle.sts_removal_batch(model, used, rec, skips, target))
            for sk in (int(skips[0]), int(skips[len(skips) // 2]), 0xFFFFFFFF):
                gv, gk = ctx.replay_get_kept(len(rec), sk, target)
                ov, ok = oracle.sts_removal_kept(model, used, rec, sk, target)
                assert gv.flags == ov.flags and gv.hash == ov.hash and (gk == ok).all()
    finally:
        ctx.close()
    # DDMin, then the removal of internal deliveries, on the GPU and with the oracle as the replay backend
    fp = ViolationFingerprint(vv.fingerprint)
    sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used))
    mcs_g, d_g, ver_g = stsSchedDDMin(sts, used, fp, speculative_depth=3)
    mcs_c, d_c, ver_c = stsSchedDDMin(OracleSTS(oracle, model, used, rec, vv.fingerprint), used, fp, speculative_depth=0)
    assert mcs_g == mcs_c and d_g.consulted == d_c.consulted and ver_g is not None and 0 < len(mcs_g) < len(used)
    verified = sts.executed_trace(mcs_g, fp)
    sts.shutdown()
    # ... and the removal of internal deliveries from the verified MCS trace: same final trace and replay count as the
    # one-replay-at-a-time loop over the oracle
    from demi_amd import internal_minimization as IM
    from .test_internal_min_cpu import OracleRemoval
    cfg = SchedulerConfig(model=model)
    stats, out = IM.minimizeInternals(cfg, verified.original_externals, verified, fp,
                                      removalStrategyCtor=lambda: IM.LeftToRightOneAtATime(verified, model))
    rstats, rout = IM.STSSchedMinimizer(verified.original_externals, verified, fp, IM.LeftToRightOneAtATime(verified, model),
                                        OracleRemoval(oracle, model), max_batch=1).minimize()
    assert (out.events == rout.events).all() and stats.total_replays == rstats.total_replays
    assert IM.countMsgEvents(out) < IM.countMsgEvents(verified)
    # the native DDMin on the same execution
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct()); ctx.model_specialize(); ctx.replay_load(used, rec)
        mcs_n, cons_n, _, st = ctx.ddmin(T.Limits(0, 0, 64, 1, vv.fingerprint, 0), T.DdminParams(0, 1024, 1, 1))
        assert tuple(mcs_n) == tuple(mcs_c) and cons_n == [(tuple(c), p) for c, p in d_c.consulted] and st.verified == 1
    finally:
        ctx.close()
    # DPOR over the wide raft5 of config 3: a budgeted exploration natively in both orders equals the oracle-backed loop
    from demi_amd.apps import raft5_config3
    from demi_amd.dpor import DPORwHeuristics
    _, dev, depth = raft5_config3()
    d1 = DPORwHeuristics(cfg, depth_bound=depth, stopIfViolationFound=False, batch=1, backend=oracle.dpor_batch)
    r1 = d1.explore(dev, max_interleavings=120)
    dr = DPORwHeuristics(cfg, depth_bound=depth, stopIfViolationFound=False, batch=512, specialize=True)
    rr = dr.explore_native(dev, max_interleavings=120, reference_order=True)
    assert len(rr.interleavings) == 120 and all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rr.interleavings, r1.interleavings))
    db = DPORwHeuristics(cfg, depth_bound=depth, stopIfViolationFound=False, batch=256, backend=oracle.dpor_batch)
    rb = db.explore(dev, max_interleavings=1500)
    dn = DPORwHeuristics(cfg, depth_bound=depth, stopIfViolationFound=False, batch=256, specialize=True)
    rn = dn.explore_native(dev, max_interleavings=1500)
    assert rn.rounds == rb.rounds and all(a.verdict == b.verdict and a.prefix_len == b.prefix_len for a, b in zip(rn.interleavings, rb.interleavings))
    dr.shutdown(); dn.shutdown()


def test_wide_srcdst_fifo_and_carried_generator(oracle):
    """The two refusals round 3's first half still listed for a wide table: `SrcDstFIFO` (RandomScheduler.scala:702-909) and
    the carried-generator mode (`executions_per_instance`, :248-269, 575-595).  Every variant of K1 - recording or not, either
    strategy, independent or chained executions - is compiled for the table (demi_gpu.hip jk_k1): verdicts and recorded traces
    against the oracle."""
    _, events, lim0 = raft5_config2()
    model = M.raft_model(5, term0=1000, loglen0=300)
    ctx = _native.Context(0)
    try:
        for p_max in (64, 24):
            fl = T.Limits(lim0.max_messages, lim0.invariant_check_interval, p_max, 0, 0, 0, T.STRATEGY_SRC_DST_FIFO)
            g, c = wide_both(ctx, oracle, model, events, 20000, fl)
            assert_same(g, c)
            assert (g["flags"] & T.V_VIOLATION).sum() > 20
            if p_max == 24:
                assert (g["flags"] & T.V_PENDING_OVF).sum() > 0
        # SrcDstFIFO is another schedule distribution than FullyRandom
        full = ctx.random_explore(2000, lim0, seed_base=SEED_BASE)
        assert (full["hash"] != g["hash"][:2000]).mean() > 0.9
        # recorded traces under SrcDstFIFO
        fl = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, T.STRATEGY_SRC_DST_FIFO)
        g = ctx.random_explore(300, fl, seed_base=SEED_BASE)
        hits = np.nonzero(g["flags"] & T.V_VIOLATION)[0]
        for i in list(hits[:3]) + [0, 1]:
            gv, grec = ctx.random_get_trace(SEED_BASE + int(i), fl)
            cv, crec, _ = oracle.random_execute(model, events, SEED_BASE + int(i), fl)
            assert (int(gv.flags), int(gv.fingerprint), int(gv.hash)) == (int(cv.flags), int(cv.fingerprint), int(cv.hash))
            assert gv.hash == g[i]["hash"] and len(grec) == len(crec) and (grec == crec).all()
            assert (grec["p0"] > 255).any()                   # 16-bit payloads in the records (terms above 1000)
        # carried generators: chains of k executions per instance, both strategies
        k, n = 5, 6003
        for strategy in (T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO):
            lim = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, strategy, 0, k)
            g, c = wide_both(ctx, oracle, model, events, n, lim)
            assert_same(g, c)
            viol = np.nonzero(g["flags"] & T.V_VIOLATION)[0]
            not_run = (g["flags"] == 0) & (g["hash"] == 0)
            assert len(viol) > 5 and not_run.sum() > 0
            lim1 = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, strategy)
            ind = ctx.random_explore(2 * k, lim1, seed_base=SEED_BASE)
            assert g[0] == ind[0] and g[k] == ind[1] and g[1] != ind[1]
            for e in (0, 3):
                v, rec, ran = ctx.random_get_trace_carried(SEED_BASE + 2, e, lim)
                vo, reco, rano = oracle.random_execute_carried(model, events, SEED_BASE + 2, e, lim)
                assert ran == rano and (int(v.flags), int(v.fingerprint), int(v.hash)) == (int(vo.flags), int(vo.fingerprint), int(vo.hash))
                assert len(rec) == len(reco) and (rec == reco).all()
    finally:
        ctx.close()


def test_wide_model_rules_at_the_boundary(oracle):
    """What is left of round 2's list of refusals: a wide table has no interpreter (it runs as compiled code or not at all),
    and 16-bit payloads belong to wide models only."""
    model = M.raft_model(3, term0=1000)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    lim = T.Limits(100, 10, 64, 0, 0, 0)
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(ev)
        with pytest.raises(_native.DemiError, match="compiled table"):
            ctx.random_explore(16, lim, seed_base=1)                    # no interpreter for the wide window
        fifo = T.Limits(100, 10, 64, 0, 0, 0)
        fifo.strategy = T.STRATEGY_SRC_DST_FIFO
        with pytest.raises(_native.DemiError, match="compiled table"):
            ctx.random_explore(16, fifo, seed_base=1)
        ctx.model_specialize()
        assert_same(ctx.random_explore(64, lim, seed_base=1), oracle.random_explore(model, ev, 64, seed_base=1, limits=lim))
        assert_same(ctx.random_explore(64, fifo, seed_base=1), oracle.random_explore(model, ev, 64, seed_base=1, limits=fifo))
        # 16-bit payloads belong to wide models only
        ctx.model_load(M.raft_model(3).to_struct())
        with pytest.raises(_native.DemiError, match="16-bit"):
            ctx.trace_load(events_to_array([start(0), send(0, M.M_BOOTSTRAP, 300, 0)]))
    finally:
        ctx.close()
