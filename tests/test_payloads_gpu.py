"""GPU parity suite for DEMI_MODEL_PAYLOADS tables (messages with three to six payload fields in the 48-bit payload area of the
64-bit message word, rows LDP / PSET, include/demi_gpu.h): the kernels compiled from the table against the oracle, through
the C ABI - the RandomScheduler kernel in every variant, recorded traces (demi_rec_event.p_hi), STSScheduler replays and the
native DDMin, DPOR.  The raft here carries akka-raft's own field sets - AppendEntries(term, prevLogIndex, prevLogTerm, entry,
leaderCommit), RequestVote(term, candidateId, lastLogTerm, lastLogIndex) - and is pinned to the protocol written out in
tests/test_oracle_cpu.py.  Bit-exact bar as everywhere."""
import os

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence

from .test_k1_gpu import assert_same
from .test_zz_array_gpu import both

pytestmark = pytest.mark.gpu


def test_raft_with_akka_raft_field_sets_through_the_kernels(oracle):
    """raft_model(log_cap = 8, real_fields = True) on the bench workload's trace: K1 in both strategies and with a carried
    generator, the recorded trace of a violating execution (every field of every message, through p0 / p1 / p_hi), K2 replays
    of candidate subsequences of it, the native DDMin against the Python loop over the oracle, K3 on the three-node cluster."""
    from demi_amd.minification import stsSchedDDMin
    from demi_amd.schedulers import ViolationFingerprint
    from tests.test_minification_cpu import OracleSTS
    from .test_k2_gpu import random_masks
    from .test_k3_gpu import collect_prefixes, same_batch
    _, events, lim = raft5_config2()
    model = M.raft_model(5, log_cap=8, real_fields=True)
    assert model.wide and model.array_len == 8 and model.payloads == 5
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        with pytest.raises(_native.DemiError, match="compiled table"):
            ctx.random_explore(16, lim, seed_base=1)                    # no interpreter for such a table
        for strategy in (T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO):
            l2 = T.Limits(lim.max_messages, lim.invariant_check_interval, 64, 0, 0, 0, strategy)
            g, c = both(ctx, oracle, model, events, 12000, l2)
            assert_same(g, c)
            assert (g["flags"] & T.V_VIOLATION).sum() > 20 and len(np.unique(g["hash"])) > 10000
        g, c = both(ctx, oracle, model, events, 2001, T.Limits(lim.max_messages, lim.invariant_check_interval, 64, 0, 0, 0, 0, 0, 4))
        assert_same(g, c)                                               # chained executions of one scheduler instance
        g, c = both(ctx, oracle, model, events, 3000, lim)
        k = int(np.nonzero(g["flags"] & T.V_VIOLATION)[0][0])
        vv, rec = ctx.random_get_trace(SEED_BASE + k, lim)
        cv, crec, states = oracle.random_execute(model, events, SEED_BASE + k, lim)
        assert vv.hash == cv.hash == g[k]["hash"] and (rec == crec).all()
        # an execution in which a leader replicates entries: the fields past the second one are on the wire (p_hi)
        for seed in range(SEED_BASE, SEED_BASE + 400):
            cv2, crec2, _ = oracle.random_execute(model, events, seed, lim)
            if crec2["p_hi"].any():
                break
        assert crec2["p_hi"].any()
        gv2, grec2 = ctx.random_get_trace(seed, lim)
        assert gv2.hash == cv2.hash and (grec2 == crec2).all()
        sends = grec2[(grec2["kind"] == T.REC_MSG_SEND) & (grec2["msg_type"] == M.M_APPEND_ENTRIES)]
        assert len(sends)
        for e in sends:
            term, prev_index, prev_term, entry_term, commit = T.payload_fields(T.rec_area(e), 5)
            assert 1 <= term < 20 and prev_index <= 8 and prev_term <= term and entry_term <= term and commit <= 8
        used = events[:T.verdict_trace_idx(vv.flags)]
        target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
        ctx.replay_load(used, rec)
        masks = random_masks(np.random.default_rng(1), len(used), 800)
        gr = ctx.replay_batch(masks, target)
        assert_same(gr, oracle.sts_replay_batch(model, used, rec, masks, target, n_threads=os.cpu_count()))
        assert gr[0]["flags"] & T.V_VIOLATION and not gr[0]["flags"] & T.V_DIVERGED and int(gr[0]["hash"]) == vv.hash
        skips = np.nonzero(rec["kind"] == T.REC_MSG_EVENT)[0].astype(np.uint32)
        assert_same(ctx.replay_removal_batch(skips, target), oracle.sts_removal_batch(model, used, rec, skips, target))
        fp = ViolationFingerprint(vv.fingerprint)
        mcs_c, d_c, _ = stsSchedDDMin(OracleSTS(oracle, model, used, rec, vv.fingerprint), used, fp, speculative_depth=0)
        mcs_n, cons_n, _, st = ctx.ddmin(target, T.DdminParams(0, 1024, 1, 1))
        assert tuple(mcs_n) == tuple(mcs_c) and cons_n == [(tuple(c_), p) for c_, p in d_c.consulted] and st.verified == 1
        assert 0 < len(mcs_n) < len(used)
        m3 = M.raft_model(3, log_cap=4, real_fields=True)
        dev = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)] + [send(a, M.M_CLIENT) for a in range(3)])
        prefixes, _, _ = collect_prefixes(oracle, m3, dev, 30, 32, 120)
        ctx.model_load(m3.to_struct())
        ctx.dpor_load(dev)
        ctx.model_specialize()
        par = T.DporParams(30, 0, 0, 0, 64, 4096)
        same_batch(ctx.dpor_batch(prefixes, par), oracle.dpor_batch(m3, dev, prefixes, par))
    finally:
        ctx.close()


@pytest.mark.parametrize("npay", [3, 4, 6])
def test_random_payload_tables_parity(oracle, npay):
    """Random wide tables whose rows read every payload field (LDP) and stage the further fields of what they send (PSET): two
    actor classes, timers, RND, quiescence markers; 16-bit external payloads truncated to the model's field width."""
    from .test_jit_cpu import _random_handler_payloads
    rng = np.random.default_rng(60 + npay)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for cls in range(2):
        for name, _ in MSGS:
            if rng.integers(5):
                h[(cls, name)] = _random_handler_payloads(rng, int(rng.integers(3, 30)), len(MSGS), npay)
    model = M.build_model("rand_pay%d" % npay, 5, MSGS, h, [[int(x) for x in rng.integers(0, 65536, 8)] for _ in range(5)],
                          (T.INV_NEVER, 0, 77, 0), actor_class=[0, 1, 0, 1, 1], n_classes=2, wide=True, payloads=npay)
    ev = [start(a) for a in range(5)]
    for i in range(40):
        ev.append(wait_quiescence() if rng.integers(0, 7) == 0 and ev[-1][0] != T.EV_WAIT_QUIESCENCE
                  else send(int(rng.integers(0, 5)), 0, int(rng.integers(0, 65536)), int(rng.integers(0, 65536))))
    ctx = _native.Context(0)
    try:
        for strategy in (T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO):
            g, c = both(ctx, oracle, model, events_to_array(ev), 4000, T.Limits(150, 9, 64, 0, 0, 0, strategy))
            assert_same(g, c)
            assert len(np.unique(g["hash"])) > 300
        lim = T.Limits(150, 9, 64, 0, 0, 0)
        for i in (0, 5):
            gv, grec = ctx.random_get_trace(SEED_BASE + i, lim)
            cv, crec, _ = oracle.random_execute(model, events_to_array(ev), SEED_BASE + i, lim)
            assert (int(gv.flags), int(gv.fingerprint), int(gv.hash)) == (int(cv.flags), int(cv.fingerprint), int(cv.hash))
            assert len(grec) == len(crec) and (grec == crec).all()
    finally:
        ctx.close()


def _ledger_model(n_actors=4):
    """A table whose EXTERNAL message has five fields: Deposit(account, amount, fee, memo, flags) at a teller, who books the
    fields into its state and forwards Post(account, amount, fee, memo, flags) to the next teller; a Post is booked and
    acknowledged.  Invariant: nobody has booked the memo 0x1A5 (a sticky flag in F5) - reachable only when the external Send's
    fourth field crosses the boundary."""
    from demi_amd.model import Asm, F, P0, P1, ME, SRC, T0, T1, T2, T3, build_model
    MSGS = [("Deposit", T.MSG_EXTERNAL), ("Post", T.MSG_INTERNAL), ("Ack", T.MSG_INTERNAL)]
    h = {}
    for name in ("Deposit", "Post"):
        a = Asm()
        a.ldp(T0, 2).ldp(T1, 3).ldp(T2, 4)
        a.add(F[0], F[0], P1).add(F[1], F[1], T0).xor(F[2], F[2], T1).or_(F[3], F[3], T2).add(F[4], F[4], 1)
        a.ldi16(T3, 0x1A5).if_eq(T1, T3, "m").mov(F[5], 1).label("m")
        if name == "Deposit":       # forward to the next teller, every field as it came
            a.add(T3, ME, 1).if_ge(T3, n_actors, "w").mov(T3, 0).label("w")
            a.send(1, T3, P0, P1, T0, T1, T2)
        else:
            a.send(2, SRC, F[4], 0)
        h[(0, name)] = a
    h[(0, "Ack")] = Asm().add(F[6], F[6], P0)
    return build_model("ledger%d" % n_actors, n_actors, MSGS, h, [[0] * 8] * n_actors, (T.INV_NEVER, 5, 1, 0), wide=True, payloads=5)


def test_external_sends_with_all_their_fields(oracle):
    """demi_ext_payload_areas (round 6): Send(name, messageCtor) of a message with five fields - demi_ext_event has room for two.
    The areas staged before the load reach K1 (plain, recording, candidate frontiers), the replay of the recorded execution (K2:
    the recorded MsgSend carries the area) and K3, each against the oracle given the same areas; without them the other fields
    are 0 and the memo never arrives."""
    model = _ledger_model()
    assert model.payloads == 5 and T.payload_bits(5) == 9
    rng = np.random.default_rng(11)
    ev, areas = [start(a) for a in range(4)], [0] * 4
    for i in range(10):
        acct, amount = int(rng.integers(0, 512)), int(rng.integers(0, 512))
        fields = [acct, amount, int(rng.integers(0, 512)), 0x1A5 if i == 6 else int(rng.integers(0, 0x1A0)), int(rng.integers(0, 512))]
        ev.append(send(int(rng.integers(0, 4)), 0, acct, amount))
        areas.append(T.pay_area(fields, 5))
        if i == 4:
            ev.append(wait_quiescence()); areas.append(0)
    events = events_to_array(ev)
    areas = np.array(areas, dtype=np.uint64)
    lim = T.Limits(200, 0, 64, 0, 0, 0)
    n = 256 if os.environ.get("DEMI_EMU") == "1" else 4096
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.model_specialize()
        # ---- K1 with the areas: the memo arrives in every execution; without: never
        ctx.trace_load(events, areas)
        g = ctx.random_explore(n, lim, seed_base=SEED_BASE)
        oracle.set_ext_areas(areas)
        c = oracle.random_explore(model, events, n, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
        assert_same(g, c)
        assert ((g["flags"] & T.V_VIOLATION) != 0).all() and len(np.unique(g["hash"])) > n // 4
        vv, rec = ctx.random_get_trace(SEED_BASE + 3, lim)
        cv, crec, _st = oracle.random_execute(model, events, SEED_BASE + 3, lim)
        assert vv.hash == cv.hash and (rec == crec).all()
        sends = rec[(rec["kind"] == T.REC_MSG_SEND) & ((rec["flags"] & 1) != 0)]
        assert sorted(T.rec_area(e) for e in sends) == sorted(int(a) for a, e in zip(areas, events) if e["kind"] == T.EV_SEND)
        # candidate frontiers (what demi_random_ddmin launches): the areas follow the events of every candidate
        masks = np.zeros((3, 4), dtype=np.uint64)
        masks[0, 0] = (1 << len(events)) - 1
        masks[1, 0] = ((1 << len(events)) - 1) & ~(1 << 11)                 # without the Send that carries the memo (event 11)
        masks[2, 0] = 0xF | (1 << 11) | (1 << 12)
        assert int(areas[11]) >> 27 & 0x1FF == 0x1A5
        gv, gf = ctx.random_explore_candidates(masks, 64, lim, seed_base=SEED_BASE)
        for i in range(3):
            keep = [j for j in range(len(events)) if (int(masks[i, 0]) >> j) & 1]
            oracle.set_ext_areas(areas[keep])
            cc = oracle.random_explore(model, events[keep], 64, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
            assert (gv[i] == cc).all(), i
        assert gf[0] & 1 and not gf[1] & 1 and gf[2] & 1
        # ---- K2: the replay of the recorded execution enqueues the recorded areas
        oracle.set_ext_areas(areas)
        lr = T.Limits(200, 0, 64, 1, int(vv.fingerprint), 0)
        ctx.replay_load(events, rec)
        km = np.array([[0xFFFFFFFFFFFFFFFF] * 4, [0xFFFFFFFFFFFFFFFF & ~(1 << 11)] + [0xFFFFFFFFFFFFFFFF] * 3], dtype=np.uint64)
        gr = ctx.replay_batch(km, lr)
        cr = oracle.sts_replay_batch(model, events, rec, km, lr, n_threads=1)
        assert (gr == cr).all() and gr[0]["flags"] & T.V_VIOLATION and int(gr[0]["hash"]) == int(vv.hash) and not gr[1]["flags"] & T.V_VIOLATION
        # ---- K3 (Start / Send only): the same areas
        keep = [j for j in range(len(events)) if events[j]["kind"] != T.EV_WAIT_QUIESCENCE][:9]
        dev, dar = events[keep], areas[keep]
        par = T.DporParams(30, 0, 0, 0, 64, 4096, 1)
        srch = T.DporSearch(64, 400, 0, 1, T.DPOR_ORDER_ROUNDS)
        ctx.dpor_load(dev, dar)
        gd = ctx.dpor_explore(par, srch)
        oracle.set_ext_areas(dar)
        cd = oracle.dpor_explore(model, dev, par, srch, os.cpu_count())
        assert len(gd[0]) == len(cd[0]) > 20 and (gd[0] == cd[0]).all() and (gd[1] == cd[1]).all()
        # ---- without areas: P0 / P1 only, the memo never arrives; a staged array of the wrong length is refused
        oracle.set_ext_areas(None)
        ctx.trace_load(events)
        g0 = ctx.random_explore(n, lim, seed_base=SEED_BASE)
        c0 = oracle.random_explore(model, events, n, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
        assert_same(g0, c0)
        assert not (g0["flags"] & T.V_VIOLATION).any()
        ctx.ext_payload_areas(areas[:5])
        with pytest.raises(_native.DemiError, match="staged"):
            ctx.trace_load(events)
        ctx.trace_load(events)                                                # (the refused load consumed the staging)
    finally:
        oracle.set_ext_areas(None)
        ctx.close()
