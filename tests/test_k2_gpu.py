"""GPU parity suite for K2 (STSScheduler replay of candidate subsequences) and DDMin end to end."""
import os

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2, raft5_config4
from demi_amd.fuzzer import FuzzerWeights, events_to_array, raft_trace
from demi_amd.minification import (DDMin, EventDagView, SpeculativeDDMin, UnmodifiedEventDag, events_to_mask, stsSchedDDMin)
from demi_amd.schedulers import EventTrace, MinimizationStats, STSScheduler, SchedulerConfig, ViolationFingerprint

pytestmark = pytest.mark.gpu


def record(gpu_ctx, model, events, lim, want_violation=True, skip=0, n=4000):
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    v = gpu_ctx.random_explore(n, lim, seed_base=SEED_BASE)
    hits = np.nonzero(v["flags"] & T.V_VIOLATION)[0] if want_violation else np.arange(n)
    i = int(hits[skip])
    vv, rec = gpu_ctx.random_get_trace(SEED_BASE + i, lim)
    return vv, rec, events[:T.verdict_trace_idx(vv.flags)]


def random_masks(rng, n_ext, n):
    masks = np.zeros((n, 4), dtype=np.uint64)
    for r in range(n):
        p = rng.choice([0.1, 0.5, 0.8, 0.95])
        keep = np.nonzero(rng.random(n_ext) < p)[0]
        if r == 0:
            keep = np.arange(n_ext)
        if r == 1:
            keep = np.arange(0)
        masks[r] = events_to_mask(keep)
    return masks


def assert_same(g, c):
    if not (g == c).all():
        bad = np.nonzero(g != c)[0]
        raise AssertionError("%d of %d verdicts differ; first at %d: gpu=%s cpu=%s" % (len(bad), len(g), bad[0], g[bad[0]], c[bad[0]]))


def test_replay_parity_random_subsequences_raft5(gpu_ctx, oracle):
    model, events, lim = raft5_config2()
    rng = np.random.default_rng(2)
    for skip in range(4):
        vv, rec, used = record(gpu_ctx, model, events, lim, skip=skip)
        masks = random_masks(rng, len(used), 3000)
        target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
        gpu_ctx.replay_load(used, rec)
        g = gpu_ctx.replay_batch(masks, target)
        c = oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count())
        assert_same(g, c)
        # the unmodified trace replays the recorded schedule exactly
        assert g[0]["flags"] & T.V_VIOLATION and not g[0]["flags"] & T.V_DIVERGED and int(g[0]["hash"]) == vv.hash
        assert T.verdict_deliveries(int(g[1]["flags"])) == 0
        assert (g["flags"] & T.V_VIOLATION).sum() > 1 and (g["flags"] & T.V_DIVERGED).sum() > 100


def test_replay_parity_fault_heavy_traces_and_capacities(gpu_ctx, oracle):
    """Kills / partitions (paired atoms, name-based Spawn/Kill matching), non-violating originals,
    small pending capacity (overflow verdicts), populate_all."""
    model = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.12, send=0.35, wait_quiescence=0.13, partition=0.25, unpartition=0.15)
    rng = np.random.default_rng(3)
    for seed in (1, 2, 3, 4):
        events = events_to_array(raft_trace(5, 90, seed, w, exact=False))
        lim = T.Limits(400, 10, 128, 0, 0, 0)
        vv, rec, used = record(gpu_ctx, model, events, lim, want_violation=False, skip=seed, n=64)
        masks = random_masks(rng, len(used), 1500)
        for p_max, pa in ((128, 0), (16, 0), (64, 1)):
            target = T.Limits(0, 0, p_max, 1, vv.fingerprint if vv.fingerprint else 0x1000103, pa)
            gpu_ctx.replay_load(used, rec)
            g = gpu_ctx.replay_batch(masks, target)
            c = oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count())
            assert_same(g, c)
            if p_max == 16:
                assert (g["flags"] & T.V_PENDING_OVF).any()


def test_replay_api_errors(gpu_ctx):
    from demi_amd._native import DemiError
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())          # resets any loaded replay
    with pytest.raises(DemiError) as e:
        gpu_ctx.replay_batch(np.zeros((1, 4), dtype=np.uint64), T.Limits(0, 0, 64, 1, 5, 0))
    assert e.value.code == T.ERR_NO_TRACE
    vv, rec, used = record(gpu_ctx, model, events, lim)
    gpu_ctx.replay_load(used, rec)
    with pytest.raises(DemiError) as e:
        gpu_ctx.replay_batch(np.zeros((1, 4), dtype=np.uint64), T.Limits(0, 0, 64, 0, 0, 0))   # no target fingerprint
    assert e.value.code == T.ERR_INVALID_ARG
    bad = rec.copy()
    bad[0]["kind"] = 99
    with pytest.raises(DemiError) as e:
        gpu_ctx.replay_load(used, bad)
    assert e.value.code == T.ERR_INVALID_TRACE


def test_ddmin_end_to_end_matches_the_oracle_backed_run(gpu_ctx, oracle):
    from tests.test_minification_cpu import OracleSTS
    model, events, lim = raft5_config2()
    for skip in (0, 3):
        vv, rec, used = record(gpu_ctx, model, events, lim, skip=skip)
        fp = ViolationFingerprint(vv.fingerprint)
        sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used))
        stats = MinimizationStats()
        mcs_g, d_g, ver_g = stsSchedDDMin(sts, used, fp, speculative_depth=3, stats=stats)
        mcs_s, d_s, _ = stsSchedDDMin(sts, used, fp, speculative_depth=0)
        mcs_c, d_c, ver_c = stsSchedDDMin(OracleSTS(oracle, model, used, rec, vv.fingerprint), used, fp, speculative_depth=0)
        assert mcs_g == mcs_s == mcs_c and d_g.consulted == d_s.consulted == d_c.consulted
        assert ver_g is not None and stats.total_replays == len(d_c.consulted)
        assert len(d_g.batches) < len(d_c.consulted)          # fewer launches than sequential oracle calls
        sts.shutdown()


def test_native_ddmin_equals_the_python_mirror_on_the_gpu(gpu_ctx, oracle):
    """demi_ddmin (the whole stsSchedDDMin in one call: atoms, ddmin2, the speculative frontier, K2 launches) against the Python
    mirror over the same GPU oracle and against the mirror over the CPU oracle: the same MCS and consultations; with a launch
    budget the search needs a handful of launches."""
    from tests.test_minification_cpu import OracleSTS
    for cfg, skip in ((raft5_config2, 0), (raft5_config2, 3), (raft5_config4, 0)):
        model, events, lim = cfg()
        vv, rec, used = record(gpu_ctx, model, events, lim, skip=skip)
        fp = ViolationFingerprint(vv.fingerprint)
        target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
        o = OracleSTS(oracle, model, used, rec, vv.fingerprint)
        o._v = lambda subs, m=model, u=used, r=rec: oracle.sts_replay_batch(
            m, u, r, np.array([events_to_mask(s) for s in subs], dtype=np.uint64).reshape(-1, 4), target)
        mcs_c, d_c, _ = stsSchedDDMin(o, used, fp, speculative_depth=0)
        gpu_ctx.replay_load(used, rec)
        for par in (T.DdminParams(3, 0, 1, 1), T.DdminParams(0, 1024, 1, 1), T.DdminParams(0, 0, 1, 1)):
            mcs_n, cons_n, batches_n, st = gpu_ctx.ddmin(target, par)
            assert tuple(mcs_n) == tuple(mcs_c) and cons_n == [(tuple(c), p) for c, p in d_c.consulted]
            assert st.verified == 1 and st.consultations == len(d_c.consulted) and st.launches == len(batches_n)
            if par.depth == 0:
                assert st.launches <= 4 and st.launches < st.consultations
        with pytest.raises(Exception, match="does not trigger"):
            gpu_ctx.ddmin(T.Limits(0, 0, 128, 1, 0x7777, 0), T.DdminParams(0, 0, 1, 1))


def test_config4_ddmin_200_external_events(gpu_ctx, oracle):
    """BASELINE config 4 (single GPU leg): 200 external events, DDMin over the STSSched oracle with a
    speculative frontier; every consulted verdict is checked against the oracle."""
    model, events, lim = raft5_config4()
    assert len(events) == 200
    vv, rec, used = record(gpu_ctx, model, events, lim, n=4000)
    assert len(used) == 200 and len(rec) <= T.MAX_REC_EVENTS
    fp = ViolationFingerprint(vv.fingerprint)
    sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used), p_max=128)
    mcs, d, ver = stsSchedDDMin(sts, used, fp, speculative_depth=4)
    assert ver is not None and len(mcs) < len(used) // 4
    cands = [c for c, _ in d.consulted]
    masks = np.array([events_to_mask(c) for c in cands], dtype=np.uint64).reshape(-1, 4)
    c = oracle.sts_replay_batch(model, used, rec, masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0), n_threads=os.cpu_count())
    assert [not bool(f & T.V_VIOLATION) for f in c["flags"]] == [p for _, p in d.consulted]
    # 1-minimality over atoms (what ddmin guarantees): dropping any one atom of the MCS stops reproducing
    dag = UnmodifiedEventDag(used)
    atoms = EventDagView(dag, mcs).get_atomic_events()
    singles = [tuple(e for e in mcs if e not in a) for a in atoms]
    res = sts.test_batch([s for s in singles if s], fp)
    assert not any(res)
    sts.shutdown()


def test_fuzz_validate_minimize_roundtrip_through_an_experiment_dir(gpu_ctx, tmp_path):
    """RunnerUtils.fuzz -> validate_replay -> save -> load -> stsSchedDDMin -> verify_mcs, all on the GPU."""
    from demi_amd.schedulers import RandomScheduler, ReplayException, ReplayScheduler
    from demi_amd.serialization import load_experiment, save_experiment
    model, events, lim = raft5_config2()
    sched = RandomScheduler(SchedulerConfig(model=model), max_executions=2000, invariant_check_interval=30, seed_base=SEED_BASE)
    sched.setMaxMessages(200)
    trace, fp = sched.explore(events)
    sched.shutdown()
    # a found violation is only kept if a strict replay reproduces it (RunnerUtils.scala:101-128)
    rs = ReplayScheduler(SchedulerConfig(model=model))
    v = rs.replay(trace, fp)
    assert int(v["flags"]) & T.V_VIOLATION
    # a tampered trace (one delivery that never happened) is rejected
    bad = EventTrace(trace.events.copy(), trace.original_externals)
    k = int(np.nonzero(bad.events["kind"] == T.REC_MSG_EVENT)[0][3])
    bad.events[k]["p0"] ^= 0x55
    with pytest.raises(ReplayException):
        rs.replay(bad, fp)
    rs.shutdown()
    save_experiment(str(tmp_path / "exp"), model, trace, fp, limits=lim, seed=SEED_BASE)
    m2, t2, fp2, meta, mcs0 = load_experiment(str(tmp_path / "exp"))
    assert m2.to_json() == model.to_json() and (t2.events == trace.events).all() and fp2 == fp and mcs0 is None
    sts = STSScheduler(SchedulerConfig(model=m2), t2)
    mcs, d, ver = stsSchedDDMin(sts, t2.original_externals, fp2)
    assert ver is not None and 0 < len(mcs) < len(t2.original_externals)
    save_experiment(str(tmp_path / "exp"), m2, t2, fp2, mcs=mcs)
    assert list(load_experiment(str(tmp_path / "exp"))[4]) == list(mcs)
    sts.shutdown()


# ----------------------------------------------------------------------------------------------
# internal-event minimization: K2 with one removed delivery per candidate
NO_SKIP = 0xFFFFFFFF


def test_removal_batch_and_kept_parity(gpu_ctx, oracle):
    """Every delivery of the recorded execution removed in turn (with and without pruned externals), small
    pending capacity included: verdicts and executed-trace marks bit-identical to the oracle's."""
    from demi_amd.internal_minimization import deliveries
    rng = np.random.default_rng(11)
    for cfg, skip in ((raft5_config2, 0), (raft5_config2, 3), (raft5_config4, 0)):
        model, events, lim = cfg()
        vv, rec, used = record(gpu_ctx, model, events, lim, skip=skip)
        gpu_ctx.replay_load(used, rec)
        dl = [i for i, _, _ in deliveries(EventTrace(rec, used))]
        skips = np.array(dl + [NO_SKIP], dtype=np.uint32)
        for p_max in (64, 12):
            target = T.Limits(0, 0, p_max, 1, vv.fingerprint, 0)
            g = gpu_ctx.replay_removal_batch(skips, target)
            c = oracle.sts_removal_batch(model, used, rec, skips, target, n_threads=os.cpu_count())
            assert_same(g, c)
        target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
        full = gpu_ctx.replay_batch(np.array([events_to_mask(range(len(used)))], dtype=np.uint64), target)[0]
        assert gpu_ctx.replay_removal_batch([NO_SKIP], target)[0] == full
        # removal combined with pruned externals
        masks = random_masks(rng, len(used), 512)
        sk = rng.choice(skips, size=512)
        g = gpu_ctx.replay_removal_batch(sk, target, masks=masks)
        c = oracle.sts_removal_batch(model, used, rec, sk, target, masks=masks, n_threads=os.cpu_count())
        assert_same(g, c)
        assert (g["flags"] & T.V_VIOLATION).sum() > 0
        # executed-trace marks
        for s, m in [(NO_SKIP, None)] + [(int(sk[i]), masks[i]) for i in range(0, 512, 37)] + [(d, None) for d in dl[:8]]:
            gv, gk = gpu_ctx.replay_get_kept(len(rec), s, target, mask=m)
            cv, ck = oracle.sts_removal_kept(model, used, rec, s, target, mask=m)
            assert gv.flags == cv.flags and gv.hash == cv.hash and gv.fingerprint == cv.fingerprint
            assert (gk == ck).all()
    # a removal candidate must be a delivery of the loaded trace
    from demi_amd._native import DemiError
    not_delivery = int(np.nonzero(rec["kind"] != T.REC_MSG_EVENT)[0][0])
    for bad in (not_delivery, len(rec) + 5):
        with pytest.raises(DemiError):
            gpu_ctx.replay_removal_batch([bad], target)


@pytest.mark.parametrize("strategy", ["LeftToRightOneAtATime", "SrcDstFIFORemoval"])
def test_minimize_internals_end_to_end(gpu_ctx, oracle, strategy):
    """fuzz -> DDMin -> verified MCS -> internal minimization, all replays on the GPU; identical (final trace,
    replay count, sizes) to the same host loop driven by the CPU oracle, sequentially."""
    from demi_amd import internal_minimization as IM
    from .test_internal_min_cpu import OracleRemoval
    cls = getattr(IM, strategy)
    model, events, lim = raft5_config4()
    cfg = SchedulerConfig(model=model)
    vv, rec, used = record(gpu_ctx, model, events, lim)
    fp = ViolationFingerprint(vv.fingerprint)
    sts = STSScheduler(cfg, EventTrace(rec, used))
    mcs, ddmin, ver = stsSchedDDMin(sts, used, fp)
    verified = sts.executed_trace(mcs, fp)
    sts.shutdown()
    assert ver is not None and verified is not None and len(mcs) < len(used)
    # the verified MCS trace is what the oracle computes for the same projection
    cv, ck = oracle.sts_removal_kept(model, used, rec, NO_SKIP, T.Limits(0, 0, 64, 1, fp.code, 0),
                                     mask=np.array(events_to_mask(mcs), dtype=np.uint64))
    want = IM.executed_trace(EventTrace(rec, used), ck, subseq=mcs)
    assert (verified.events == want.events).all() and (verified.original_externals == want.original_externals).all()
    stats, out = IM.minimizeInternals(cfg, verified.original_externals, verified, fp,
                                      removalStrategyCtor=lambda: cls(verified, model))
    ref = IM.STSSchedMinimizer(verified.original_externals, verified, fp, cls(verified, model),
                               OracleRemoval(oracle, model), max_batch=1)
    rstats, rout = ref.minimize()
    assert (out.events == rout.events).all() and stats.total_replays == rstats.total_replays
    assert IM.countMsgEvents(out) < IM.countMsgEvents(verified)
    # the minimized schedule replays strictly (nothing absent) to the violation
    gpu_ctx.replay_load(out.original_externals, out.events)
    v = gpu_ctx.replay_removal_batch([NO_SKIP], T.Limits(0, 0, 64, 1, fp.code, 0))[0]
    assert v["flags"] & T.V_VIOLATION and not v["flags"] & T.V_DIVERGED
    assert T.verdict_deliveries(int(v["flags"])) == IM.countMsgEvents(out)


def test_specialised_replay_kernel_is_bit_identical(gpu_ctx, oracle):
    """K2 compiled for the model's table (demi_model_specialize, compiled at the first replay launch) against the
    table interpreter and the oracle: subsequence masks, removal candidates, executed-trace marks."""
    from demi_amd.internal_minimization import deliveries
    rng = np.random.default_rng(21)
    for cfg in (raft5_config2, raft5_config4):
        model, events, lim = cfg()
        vv, rec, used = record(gpu_ctx, model, events, lim)
        gpu_ctx.replay_load(used, rec)
        masks = random_masks(rng, len(used), 4000)
        dl = np.array([i for i, _, _ in deliveries(EventTrace(rec, used))] + [NO_SKIP], dtype=np.uint32)
        sk = rng.choice(dl, size=4000)
        target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
        plain = gpu_ctx.replay_batch(masks, target)
        rem = gpu_ctx.replay_removal_batch(sk, target, masks=masks)
        kept = [gpu_ctx.replay_get_kept(len(rec), int(sk[i]), target, mask=masks[i])[1] for i in range(0, 4000, 401)]
        gpu_ctx.model_specialize()
        assert gpu_ctx.is_specialized()
        assert_same(gpu_ctx.replay_batch(masks, target), plain)
        assert_same(gpu_ctx.replay_removal_batch(sk, target, masks=masks), rem)
        for j, i in enumerate(range(0, 4000, 401)):
            assert (gpu_ctx.replay_get_kept(len(rec), int(sk[i]), target, mask=masks[i])[1] == kept[j]).all()
        gpu_ctx.model_specialize(False)
        assert_same(plain, oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count()))


def test_random_ddmin_with_the_random_scheduler_as_oracle(gpu_ctx):
    """RunnerUtils.randomDDMin: every DDMin consultation is a K1 launch of R random interleavings of the candidate."""
    from demi_amd.minification import randomDDMin
    model, events, lim = raft5_config2()
    vv, rec, used = record(gpu_ctx, model, events, lim)
    fp = ViolationFingerprint(vv.fingerprint, model.fp_match_mask)
    stats = MinimizationStats()
    mcs, ddmin, verified = randomDDMin(SchedulerConfig(model=model), EventTrace(rec, used), fp, max_executions=256,
                                       seed_base=SEED_BASE, stats=stats)
    assert verified is not None and 0 < len(mcs) < len(used)
    assert stats.total_replays == 256 * len(ddmin.consulted)
    # the MCS is 1-minimal with respect to this (randomised, seeded) oracle: the consultations that removed an atom passed
    assert any(passes for _, passes in ddmin.consulted) and any(not passes for _, passes in ddmin.consulted)
    # deterministic: same seeds, same answer
    mcs2, _, _ = randomDDMin(SchedulerConfig(model=model), EventTrace(rec, used), fp, max_executions=256, seed_base=SEED_BASE)
    assert mcs2 == mcs


def test_replay_golden_fixture_on_gpu(gpu_ctx):
    """The committed fixture (tools/make_golden.py): recorded execution, masks, removal candidates and executed-trace
    marks with the verdicts the oracle gave them."""
    model, events, lim = raft5_config2()
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "raft5_config2_replay.npz"))
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    vv, rec = gpu_ctx.random_get_trace(SEED_BASE + int(z["index"]), lim)
    assert vv.fingerprint == int(z["fingerprint"]) and (rec == T.rec_events(z["rec"])).all()
    target = T.Limits(0, 0, 64, 1, int(z["fingerprint"]), 0)
    gpu_ctx.replay_load(z["used"], z["rec"])
    assert_same(gpu_ctx.replay_batch(z["masks"], target), z["mask_verdicts"])
    assert_same(gpu_ctx.replay_removal_batch(z["skips"], target), z["skip_verdicts"])
    for k in range(len(z["kept"])):
        assert (gpu_ctx.replay_get_kept(len(z["rec"]), int(z["skips"][k]), target)[1] == z["kept"][k]).all()


def test_fuzz_then_the_gamut_end_to_end():
    """RunnerUtils.fuzz -> validate by strict replay -> provenance -> stsSchedDDMin -> minimizeInternals, every
    execution and replay on the GPU."""
    from demi_amd.fuzzer import FuzzerWeights
    from demi_amd.runner_utils import fuzz, run_the_gamut
    from demi_amd.schedulers import ReplayScheduler
    from demi_amd.internal_minimization import countMsgEvents
    model = M.raft_model(5)
    cfg = SchedulerConfig(model=model)
    w = FuzzerWeights()
    gen = lambda i: events_to_array(raft_trace(5, 50, 0xF022 + i, w, exact=False))
    res = fuzz(gen, cfg, validate_replay=lambda: ReplayScheduler(cfg), maxMessages=200, executions_per_test=2048, max_tests=8,
               provenance_device=0)
    assert res is not None
    trace, violation, initial, filtered = res
    assert len(initial) == 1 + countMsgEvents(trace) and 0 < len(filtered) <= len(initial)
    from demi_amd.provenance import pruneConcurrentEvents
    assert (filtered == pruneConcurrentEvents(initial, violation.affectedNodes())).all()      # device kernel == host class
    out = run_the_gamut(cfg, trace, violation)
    assert out["verified_mcs"] is not None and len(out["mcs"]) < out["original_externals"]
    assert out["minimized_deliveries"] <= countMsgEvents(out["verified_mcs"]) <= out["original_deliveries"]
    assert out["ddmin_replays"] > 0 and out["intmin_replays"] > 0


@pytest.mark.parametrize("k2_mode", ["auto", "wave", "lds", "hbm", "scan"])
def test_filter_known_absents_parity(gpu_ctx, oracle, k2_mode, monkeypatch):
    """SchedulerConfig.filterKnownAbsents (EventTrace.filterKnownAbsentInternals as the last stage of the projection),
    as the reference computes it and corrected: verdicts, removal candidates and executed-trace marks against the
    oracle on fault-heavy traces, switching between the two lowerings of one loaded execution, in every kernel variant."""
    from demi_amd.internal_minimization import deliveries
    if k2_mode == "scan":
        monkeypatch.setenv("DEMI_K2_SCAN", "1")
    elif k2_mode != "auto":
        monkeypatch.setenv("DEMI_K2_MODE", k2_mode)          # (one candidate per wave with the look-ahead / [word][lane] counters in LDS / in HBM)
    model = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.12, send=0.35, wait_quiescence=0.13, partition=0.25, unpartition=0.15)
    rng = np.random.default_rng(5)
    differs = 0
    for seed in (1, 2, 3):
        events = events_to_array(raft_trace(5, 90, seed, w, exact=False))
        lim = T.Limits(400, 10, 128, 0, 0, 0)
        vv, rec, used = record(gpu_ctx, model, events, lim, want_violation=False, skip=seed, n=64)
        masks = random_masks(rng, len(used), 1200)
        fpc = vv.fingerprint if vv.fingerprint else 0x1000103
        gpu_ctx.replay_load(used, rec)
        dl = np.array([i for i, _, _ in deliveries(EventTrace(rec, used))] + [NO_SKIP], dtype=np.uint32)
        sk = rng.choice(dl, size=len(masks))
        res = {}
        for mode in (T.FILTER_ABSENTS_LITERAL, T.FILTER_ABSENTS_OFF, T.FILTER_ABSENTS_CORRECTED, T.FILTER_ABSENTS_LITERAL):
            target = T.Limits(0, 0, 128, 1, fpc, 0, 0, mode)
            g = gpu_ctx.replay_batch(masks, target)
            assert_same(g, oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count()))
            res[mode] = g
            g = gpu_ctx.replay_removal_batch(sk, target, masks=masks)
            assert_same(g, oracle.sts_removal_batch(model, used, rec, sk, target, masks=masks, n_threads=os.cpu_count()))
            for i in range(0, len(masks), 173):
                gv, gk = gpu_ctx.replay_get_kept(len(rec), int(sk[i]), target, mask=masks[i])
                cv, ck = oracle.sts_removal_kept(model, used, rec, int(sk[i]), target, mask=masks[i])
                assert gv.flags == cv.flags and gv.hash == cv.hash and (gk == ck).all()
        differs += int((res[T.FILTER_ABSENTS_LITERAL] != res[T.FILTER_ABSENTS_OFF]).sum())
        differs += int((res[T.FILTER_ABSENTS_LITERAL] != res[T.FILTER_ABSENTS_CORRECTED]).sum())
        if k2_mode == "auto" and seed == 1:      # the kernel compiled from the table as well
            gpu_ctx.model_specialize()
            for mode in (T.FILTER_ABSENTS_LITERAL, T.FILTER_ABSENTS_CORRECTED):
                assert_same(gpu_ctx.replay_batch(masks, T.Limits(0, 0, 128, 1, fpc, 0, 0, mode)), res[mode])
            gpu_ctx.model_specialize(False)
    assert differs > 0
    with pytest.raises(Exception):
        gpu_ctx.replay_batch(masks, T.Limits(0, 0, 128, 1, fpc, 0, 0, 3))


@pytest.mark.gpu
def test_bench_candidates_against_the_sts_transliterations_record(gpu_ctx):
    """The bench's replay workload (config 4: the 200-event failing execution, candidates drawn with default_rng(0) at 0.7 per
    event), all 2^20 candidates: K2's verdicts are the ones the transliteration of the Scala STSScheduler produced
    (tools/check_replay_transliteration.py; tests/golden/replay_config4_transliteration.json - the CPU suite holds the C oracle
    against the same record)."""
    import hashlib
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay_config4_transliteration.json")) as f:
        rec_t = json.load(f)
    model, events, lim = raft5_config4()
    vv, rec, used = record(gpu_ctx, model, events, lim, n=4000)
    assert len(used) == rec_t["externals"] and len(rec) == rec_t["recorded_events"]
    emu = os.environ.get("DEMI_EMU") == "1"
    n = 1024 if emu else rec_t["candidates"]          # (the emulator: the first candidates only, against the oracle's bytes for them)
    keep = np.random.default_rng(0).random((n, len(used))) < 0.7
    masks = np.zeros((n, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    gpu_ctx.replay_load(used, rec)
    got = gpu_ctx.replay_batch(masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0))
    if not emu:
        f = rec_t["first"]
        assert hashlib.sha256(masks[:f].tobytes()).hexdigest() == rec_t["sha256_masks"]
        assert hashlib.sha256(np.ascontiguousarray(got[:f]).tobytes()).hexdigest() == rec_t["sha256_verdicts"]
        assert hashlib.sha256(masks.tobytes()).hexdigest() == rec_t["sha256_masks_of_all"]
        assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == rec_t["sha256_verdicts_of_all"]
        assert int(((got["flags"] & T.V_VIOLATION) != 0).sum()) == rec_t["still_violating"]
    else:
        from oracle import oracle_py as O
        want = O.sts_replay_batch(model, used, rec, masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0), n_threads=os.cpu_count() or 1)
        assert (got == want).all()


@pytest.mark.gpu
def test_replay_launches_of_one_ctx_on_two_streams(oracle):
    """demi_replay_batch_dev alternates between two sets of K2's per-launch scratch (work counter, spill, word counters in HBM): launches
    that alternate between two streams overlap and must still each replay exactly their own candidates - against the oracle, for the
    scanning kernel's table and for one with word counters, with a (re)load in between that has to wait for both."""
    import ctypes as C
    import torch
    from demi_amd import _native
    model, events, lim = raft5_config4(120)
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
        i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
        vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
        used = events[:T.verdict_trace_idx(int(vv.flags))]
        target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
        rng = np.random.default_rng(7)
        n, k = 1 << 16, 6
        masks = [random_masks(rng, len(used), n) for _ in range(k)]
        for specialise in (False, True):
            ctx.replay_load(used, rec)
            if specialise:
                ctx.model_specialize()
            want = [oracle.sts_replay_batch(model, used, rec, m, target, n_threads=os.cpu_count()) for m in masks[:2]]
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            d_masks = [torch.from_numpy(m.view(np.int64)).cuda() for m in masks]
            outs = [torch.zeros((n, 2), dtype=torch.int64, device="cuda") for _ in range(k)]
            torch.cuda.synchronize()
            for j in range(k):
                ctx.replay_batch_dev(d_masks[j].data_ptr(), n, target, outs[j].data_ptr(), stream=C.c_void_p(streams[j % 2].cuda_stream))
            ctx.replay_load(used, rec)                 # must wait for the launches in flight on both scratch sets
            torch.cuda.synchronize()
            seq = [ctx.replay_batch(m, target) for m in masks]
            for j in range(k):
                got = outs[j].cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)
                assert (got == seq[j]).all(), (specialise, j)
                if j < 2:
                    assert (got == want[j]).all(), (specialise, j)
    finally:
        ctx.close()
