"""GPU parity suite for DEMI_INV_PROGRAM invariants (the per-actor predicate / key as a row program, include/demi_gpu.h): K1 with
the table interpreted and compiled, the recording kernel, K2 replays and K3 interleavings, all against the oracle, and a
program that restates raft's descriptor against the descriptor itself."""
import os

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence
from demi_amd.model import Asm

from .test_k1_gpu import assert_same

pytestmark = pytest.mark.gpu


def leader_program():
    """"a leader, keyed by its term": raft's descriptor (AT_MOST_ONE, ROLE == LEADER, key TERM) as rows"""
    return Asm().if_eq(M.ROLE, M.LEADER, "no").mov(M.T0, 1).mov(M.T1, M.TERM).label("no").halt()


@pytest.mark.parametrize("wide", [False, True])
def test_program_restating_the_descriptor_gives_the_same_verdicts(oracle, wide):
    _, events, lim = raft5_config2()
    kw = dict(term0=1000, loglen0=300) if wide else {}
    desc, prog = M.raft_model(5, **kw), M.raft_model(5, invariant=(T.INV_AT_MOST_ONE, leader_program()), **kw)
    assert prog.inv_kind == T.INV_AT_MOST_ONE | T.INV_PROGRAM and prog.wide == wide
    ctx = _native.Context(0)
    try:
        out = {}
        for name, m in (("desc", desc), ("prog", prog)):
            for jit in ((True,) if wide else (False, True)):
                ctx.model_load(m.to_struct())
                ctx.trace_load(events)
                if jit:
                    ctx.model_specialize()
                out[name, jit] = ctx.random_explore(20000, lim, seed_base=SEED_BASE)
        ref = oracle.random_explore(prog, events, 20000, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
        for k, v in out.items():
            assert_same(v, ref)
        assert (ref["flags"] & T.V_VIOLATION).sum() > 50
        # interval checks and lookingFor go through the same program
        lim2 = T.Limits(lim.max_messages, 3, lim.p_max, 1, int(ref[(ref["flags"] & T.V_VIOLATION) != 0]["fingerprint"][0]), 0)
        ctx.model_load(prog.to_struct()); ctx.trace_load(events); ctx.model_specialize()
        assert_same(ctx.random_explore(5000, lim2, seed_base=SEED_BASE), oracle.random_explore(prog, events, 5000, seed_base=SEED_BASE, limits=lim2, n_threads=os.cpu_count()))
    finally:
        ctx.close()


def test_invariant_relating_two_actors_on_raft5(oracle):
    """DEMI_OP_PEER: an invariant that relates two actors by more than an equal key - "a leader whose term is below the term of
    some created actor" (a stale leader, NEVER) - on the bench workload: K1 interpreted and compiled, with interval checks, against
    the oracle (whose PEER is pinned to a plain-Python evaluation in the CPU suite).  An actor's hit depends on the other actors'
    states here, so K1 rebuilds its hit mask at every check instead of updating the receiver's bit per delivery."""
    _, events, lim = raft5_config2()
    stale = Asm().if_eq(M.ROLE, M.LEADER, "no")
    for j in range(5):
        stale.mov(M.T2, j).peer(M.T1, M.T2, M.PEER_CREATED).if_ne(M.T1, 0, "n%d" % j).peer(M.T1, M.T2, M.TERM)
        stale.if_gt(M.T1, M.TERM, "n%d" % j).mov(M.T0, 1).label("n%d" % j)
    stale.label("no").mov(M.T1, 0).halt()
    model = M.raft_model(5, invariant=(T.INV_NEVER, stale))
    ctx = _native.Context(0)
    try:
        for lim_ in (lim, T.Limits(lim.max_messages, 7, lim.p_max, 0, 0, 0)):
            ref = oracle.random_explore(model, events, 20000, seed_base=SEED_BASE, limits=lim_, n_threads=os.cpu_count())
            for jit in (False, True):
                ctx.model_load(model.to_struct()); ctx.trace_load(events)
                if jit:
                    ctx.model_specialize()
                assert_same(ctx.random_explore(20000, lim_, seed_base=SEED_BASE), ref)
            assert 20 < (ref["flags"] & T.V_VIOLATION).sum() < 20000
    finally:
        ctx.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_program_invariants_through_every_kernel(oracle, seed):
    """Random tables with a random invariant program under each combining kind: K1 (interpreter and compiled), a recorded
    violating execution, its replays (K2) and DPOR interleavings (K3) - verdicts equal the oracle's.  Seeds 5 and 6: programs
    with DEMI_OP_PEER rows (other actors' fields)."""
    from tests.test_jit_cpu import _random_handler
    from tests.test_oracle_cpu import _random_pure_program
    from tests.test_k2_gpu import random_masks
    rng = np.random.default_rng(100 + seed)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for name, _ in MSGS:
        h[(0, name)] = _random_handler(rng, int(rng.integers(4, 16)), len(MSGS), few_effects=(name != "E"))
    kind = [T.INV_AT_MOST_ONE, T.INV_NEVER, T.INV_AGREE, T.INV_AT_MOST_ONE, T.INV_AT_MOST_ONE, T.INV_AGREE][seed - 1]
    model = M.build_model("rinv%d" % seed, 4, MSGS, h, [[int(x) for x in rng.integers(0, 4, 8)] for _ in range(4)],
                          (kind, _random_pure_program(rng, int(rng.integers(4, 20)), peers=seed >= 5)))
    ev = [start(a) for a in range(4)]
    for i in range(10):
        ev.append(wait_quiescence() if i == 5 else send(int(rng.integers(0, 4)), 0, int(rng.integers(0, 256)), int(rng.integers(0, 256))))
    ev = events_to_array(ev)
    lim = T.Limits(120, 4, 64, 0, 0, 0)
    ctx = _native.Context(0)
    try:
        ref = oracle.random_explore(model, ev, 6000, seed_base=7, limits=lim, n_threads=os.cpu_count())
        for jit in (False, True):
            ctx.model_load(model.to_struct()); ctx.trace_load(ev)
            if jit:
                ctx.model_specialize()
            assert_same(ctx.random_explore(6000, lim, seed_base=7), ref)
        hits = np.nonzero((ref["flags"] & T.V_VIOLATION) != 0)[0]
        ok = np.nonzero((ref["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)) == 0)[0]
        k = int(hits[0]) if len(hits) else int(ok[0])
        vv, rec = ctx.random_get_trace(7 + k, lim)
        cv, crec, _ = oracle.random_execute(model, ev, 7 + k, lim, record=True)
        assert vv.flags == cv.flags and vv.fingerprint == cv.fingerprint and vv.hash == cv.hash and (rec == crec).all()
        used = ev[:T.verdict_trace_idx(vv.flags)]
        target = T.Limits(0, 0, 64, 1, vv.fingerprint if vv.fingerprint else 0x2000001, 0)
        masks = random_masks(rng, len(used), 400)
        for jit in (False, True):
            ctx.model_load(model.to_struct()); ctx.trace_load(ev)
            if jit:
                ctx.model_specialize()
            ctx.replay_load(used, rec)
            assert_same(ctx.replay_batch(masks, target), oracle.sts_replay_batch(model, used, rec, masks, target))
        # K3: the prefixes of a small oracle-backed exploration, replayed on the GPU (Start / Send externals only)
        from demi_amd.dpor import DPORwHeuristics
        from demi_amd.schedulers import SchedulerConfig
        from tests.test_k3_gpu import collect_prefixes, same_batch
        dev = ev
        prefixes, res, _ = collect_prefixes(oracle, model, dev, 14, 16, 64)
        par = T.DporParams(14, 0, 0, 0, 64, 4096)
        for jit in (False, True):
            ctx.model_load(model.to_struct())
            if jit:
                ctx.model_specialize()
            ctx.dpor_load(dev)
            same_batch(ctx.dpor_batch(prefixes, par), oracle.dpor_batch(model, dev, prefixes, par))
    finally:
        ctx.close()
