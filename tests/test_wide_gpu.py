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


def test_wide_models_are_refused_where_they_cannot_run(oracle):
    model = M.raft_model(3, term0=1000)
    ev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
    lim = T.Limits(100, 10, 64, 0, 0, 0)
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(ev)
        with pytest.raises(_native.DemiError, match="compiled table"):
            ctx.random_explore(16, lim, seed_base=1)                    # no interpreter for the wide window
        ctx.model_specialize()
        assert_same(ctx.random_explore(64, lim, seed_base=1), oracle.random_explore(model, ev, 64, seed_base=1, limits=lim))
        with pytest.raises(_native.DemiError):
            ctx.random_get_trace(1, lim)                                # no recorded-trace format
        fifo = T.Limits(100, 10, 64, 0, 0, 0)
        fifo.strategy = T.STRATEGY_SRC_DST_FIFO
        with pytest.raises(_native.DemiError):
            ctx.random_explore(16, fifo, seed_base=1)
        with pytest.raises(_native.DemiError):
            ctx.dpor_load(ev)
        with pytest.raises(_native.DemiError):
            ctx.replay_load(ev, np.zeros(0, dtype=T.REC_EVENT_DTYPE))
        # 16-bit payloads belong to wide models only
        ctx.model_load(M.raft_model(3).to_struct())
        with pytest.raises(_native.DemiError, match="16-bit"):
            ctx.trace_load(events_to_array([start(0), send(0, M.M_BOOTSTRAP, 300, 0)]))
    finally:
        ctx.close()
