"""GPU suite: crashed actors (DEMI_OP_CRASH) and Util.find_non_blocked_message on every kernel flavour, against the oracle
(whose restatement is pinned to a literal transliteration of the Scala in tests/test_blocked_actors_cpu.py)."""
import os

import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.fuzzer import events_to_array, kill, send, start, wait_quiescence
from demi_amd.model import Asm, build_model
from tests.test_k1_gpu import assert_same, both

pytestmark = pytest.mark.gpu


def crashy_model():
    """4 actors pass Work around a ring and arm a repeating timer; an actor's receive throws on its third Work (it is
    blocked until it is Start()ed again), so executions carry pending messages for blocked actors through many draws."""
    msgs = [("Go", T.MSG_EXTERNAL), ("Work", T.MSG_INTERNAL), ("Tick", T.MSG_TIMER), ("Note", T.MSG_EXTERNAL)]
    go = Asm().add(M.T0, M.ME, 1).and_(M.T0, M.T0, 3).mov(M.T1, 1).send(1, M.T0, M.T1, 0).trep(2)
    work = (Asm().add(M.F[0], M.F[0], 1).if_eq(M.F[0], 3, "ok").crash().label("ok")
            .if_lt(M.P0, 9, "end").add(M.T0, M.ME, 1).and_(M.T0, M.T0, 3).add(M.T1, M.P0, 1).send(1, M.T0, M.T1, 0)
            .if_eq(M.P0, 2, "end").bcast(1, M.T1, 7).label("end"))
    tick = Asm().add(M.F[1], M.F[1], 1).if_ge(M.F[1], 4, "end").tcancel(2).label("end")
    note = Asm().mov(M.F[2], M.P0)
    h = {(0, "Go"): go, (0, "Work"): work, (0, "Tick"): tick, (0, "Note"): note}
    return build_model("crashy", 4, msgs, h, [[0] * 8] * 4, (T.INV_AT_MOST_ONE, 0, 3, 2))


def crashy_trace():
    return events_to_array([start(a) for a in range(4)] + [send(0, 0), send(2, 0), send(1, 3, 9), send(3, 3, 4), wait_quiescence(),
                                                          start(1), send(1, 0), send(1, 3, 5), kill(2), wait_quiescence(), start(2),
                                                          send(2, 3, 6), send(0, 0)])


def test_model_really_crashes(oracle):
    model, ev = crashy_model(), crashy_trace()
    v, rec, states = oracle.random_execute(model, ev, 1234567, T.Limits(300, 0, 64, 0, 0, 0))
    assert max((int(s) & 0xFF) for s in states) >= 3           # some actor reached its third Work, i.e. crashed


@pytest.mark.parametrize("strategy", [T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO])
def test_k1_parity_with_crashed_actors(gpu_ctx, oracle, strategy):
    model, ev = crashy_model(), crashy_trace()
    for p_max, maxm, interval in ((64, 300, 0), (64, 60, 7), (16, 300, 0)):
        lim = T.Limits(maxm, interval, p_max, 0, 0, 0, strategy)
        g, c = both(gpu_ctx, oracle, model, ev, 8192, lim, seed_base=0x0BADC0DE12345, jit=True)
        assert_same(g, c)
        assert len(set(g["hash"].tolist())) > 1000
    # the recorded execution (REC kernel) equals the oracle's record
    lim = T.Limits(300, 0, 64, 0, 0, 0, strategy)
    for seed in (0x0BADC0DE12345 + k for k in (0, 17, 4000)):
        gv, grec = gpu_ctx.random_get_trace(seed, lim)
        cv, crec, _ = oracle.random_execute(model, ev, seed, lim)
        assert gv.flags == cv.flags and gv.hash == cv.hash and len(grec) == len(crec) and (grec == crec).all()


def test_k2_parity_with_crashed_actors(gpu_ctx, oracle, monkeypatch):
    model, ev = crashy_model(), crashy_trace()
    lim = T.Limits(300, 0, 64, 0, 0, 0)
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(ev)
    vv, rec = gpu_ctx.random_get_trace(0x0BADC0DE12345 + 3, lim)
    used = ev[:T.verdict_trace_idx(vv.flags)]
    rng = np.random.default_rng(3)
    masks = np.zeros((2048, 4), dtype=np.uint64)
    masks[:, 0] = rng.integers(0, 1 << len(used), size=2048, dtype=np.uint64)
    masks[0, 0] = (1 << len(used)) - 1
    target = T.Limits(0, 0, 64, 1, 0x7FFFFFFF, 0)
    c = oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count())
    for mode in ("wave", "lds", "hbm", None):
        if mode:
            monkeypatch.setenv("DEMI_K2_MODE", mode)
        else:
            monkeypatch.delenv("DEMI_K2_MODE")
        gpu_ctx.replay_load(used, rec)
        assert_same(gpu_ctx.replay_batch(masks, target), c)
    assert (c["flags"] & T.V_DIVERGED).any() and T.verdict_deliveries(int(c["flags"][0])) == T.verdict_deliveries(vv.flags)


def test_k3_parity_with_crashed_actors(gpu_ctx, oracle):
    from demi_amd.dpor import DPORwHeuristics
    from demi_amd.schedulers import SchedulerConfig
    model = crashy_model()
    ev = events_to_array([start(a) for a in range(4)] + [send(0, 0), send(2, 0), send(1, 0), send(3, 0), send(1, 3, 9)])
    for jit in (False, True):
        dg = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=24, stopIfViolationFound=False, batch=64, specialize=jit)
        rg = dg.explore(ev, max_interleavings=600)
        dc = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=24, stopIfViolationFound=False, batch=64, backend=oracle.dpor_batch)
        rc = dc.explore(ev, max_interleavings=600)
        assert rg.rounds == rc.rounds and len(rg.interleavings) == len(rc.interleavings) >= 20
        for a, b in zip(rg.interleavings, rc.interleavings):
            assert a.verdict == b.verdict and (a.trace == b.trace).all()
        dg.shutdown()


def jittery_model():
    """Raft-like randomized timeouts: an actor draws its next timeout budget from the application's generator
    (Instrumenter().seededRandom) and acts when the budget runs out - the schedule decides who draws which number."""
    msgs = [("Go", T.MSG_EXTERNAL), ("Ping", T.MSG_INTERNAL), ("Tick", T.MSG_TIMER)]
    go = Asm().rnd(M.F[0], 7).add(M.F[0], M.F[0], 1).trep(2)
    tick = (Asm().sub(M.F[0], M.F[0], 1).if_eq(M.F[0], 0, "end").rnd(M.F[0], M.F[2]).add(M.F[0], M.F[0], 1).add(M.F[1], M.F[1], 1)
            .mov(M.T1, 0).bcast(1, M.F[1], 3).if_ge(M.F[1], 4, "end").tcancel(2).label("end"))
    ping = Asm().add(M.F[3], M.F[3], 1).rnd(M.T0, 200).max(M.F[4], M.F[4], M.T0).add(M.F[2], M.F[2], 1)
    h = {(0, "Go"): go, (0, "Ping"): ping, (0, "Tick"): tick}
    return build_model("jittery", 4, msgs, h, [[0, 0, 5, 0, 0, 0, 0, 0]] * 4, (T.INV_AT_MOST_ONE, 1, 4, 4))


def test_application_randomness_parity_on_every_kernel(gpu_ctx, oracle, monkeypatch):
    model = jittery_model()
    ev = events_to_array([start(a) for a in range(4)] + [send(a, 0) for a in range(4)] + [wait_quiescence(), send(1, 0), send(2, 0)])
    for strategy in (T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO):
        lim = T.Limits(150, 11, 64, 0, 0, 0, strategy)
        g, c = both(gpu_ctx, oracle, model, ev, 8192, lim, seed_base=0x7E57AB1E0000, jit=True)
        assert_same(g, c)
        assert len(set(g["hash"].tolist())) > 2000
    lim = T.Limits(150, 0, 64, 0, 0, 0)
    gv, grec = gpu_ctx.random_get_trace(0x7E57AB1E0000 + 5, lim)
    cv, crec, _ = oracle.random_execute(model, ev, 0x7E57AB1E0000 + 5, lim)
    assert gv.hash == cv.hash and (grec == crec).all()
    # K2: replays of candidate subsequences draw the same numbers in the same deliveries
    used = ev[:T.verdict_trace_idx(gv.flags)]
    rng = np.random.default_rng(5)
    masks = np.zeros((1024, 4), dtype=np.uint64)
    masks[:, 0] = rng.integers(0, 1 << len(used), size=1024, dtype=np.uint64)
    masks[0, 0] = (1 << len(used)) - 1
    target = T.Limits(0, 0, 64, 1, 0x7FFFFFFF, 0)
    want = oracle.sts_replay_batch(model, used, grec, masks, target, n_threads=os.cpu_count())
    for mode in ("wave", "lds", "hbm"):
        monkeypatch.setenv("DEMI_K2_MODE", mode)
        gpu_ctx.replay_load(used, grec)
        assert_same(gpu_ctx.replay_batch(masks, target), want)
    monkeypatch.delenv("DEMI_K2_MODE")
    assert int(want[0]["hash"]) == gv.hash
    # K3
    from demi_amd.dpor import DPORwHeuristics
    from demi_amd.schedulers import SchedulerConfig
    ev3 = events_to_array([start(a) for a in range(4)] + [send(0, 0), send(1, 0)])
    dg = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=20, stopIfViolationFound=False, batch=64, specialize=True)
    rg = dg.explore(ev3, max_interleavings=300)
    dc = DPORwHeuristics(SchedulerConfig(model=model), depth_bound=20, stopIfViolationFound=False, batch=64, backend=oracle.dpor_batch)
    rc = dc.explore(ev3, max_interleavings=300)
    assert rg.rounds == rc.rounds and all(a.verdict == b.verdict for a, b in zip(rg.interleavings, rc.interleavings))
    dg.shutdown()
