"""CPU suite, part 5: internal-event minimization (removal strategies, STSSchedMinimizer with and without
speculation) over the oracle's restatement of the STS replay with one removed delivery."""
from collections import Counter

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence
from demi_amd.internal_minimization import (LeftToRightOneAtATime, SrcDstFIFORemoval, STSSchedMinimizer, countMsgEvents,
                                            deliveries, executed_trace, getFingerprintedDeliveries)
from demi_amd.minification import events_to_mask, stsSchedDDMin
from demi_amd.schedulers import EventTrace, MinimizationStats, ViolationFingerprint

from .test_minification_cpu import OracleSTS, _violating_execution

NO_SKIP = 0xFFFFFFFF


class OracleRemoval:
    """The oracle standing in for StsRemovalOracle (same interface), CPU tests only."""

    def __init__(self, oracle, model):
        self.o, self.model = oracle, model
        self.launches = 0

    def _lim(self, fp):
        return T.Limits(0, 0, 64, 1, fp.code, 0)

    def test_removals(self, trace, skips, violation):
        self.launches += 1
        v = self.o.sts_removal_batch(self.model, trace.original_externals, trace.events, skips, self._lim(violation))
        return [bool(f & T.V_VIOLATION) for f in v["flags"]]

    def executed(self, trace, skip, violation):
        v, kept = self.o.sts_removal_kept(self.model, trace.original_externals, trace.events, skip, self._lim(violation))
        if not (v.flags & T.V_VIOLATION):
            return None
        return executed_trace(trace, kept)


def _verified_mcs(oracle, model, events, lim, skip=0):
    """fuzz -> DDMin -> verified MCS trace re-based on the MCS (what RunnerUtils.stsSchedDDMin returns)."""
    vv, rec, used = _violating_execution(oracle, model, events, lim, skip)
    fp = ViolationFingerprint(vv.fingerprint)
    mcs, _, ver = stsSchedDDMin(OracleSTS(oracle, model, used, rec, vv.fingerprint), used, fp, speculative_depth=2)
    assert ver is not None
    mask = np.array(events_to_mask(mcs), dtype=np.uint64)
    v, kept = oracle.sts_removal_kept(model, used, rec, NO_SKIP, T.Limits(0, 0, 64, 1, fp.code, 0), mask=mask)
    assert v.flags & T.V_VIOLATION
    trace = executed_trace(EventTrace(rec, used), kept, subseq=mcs)
    return trace, fp


def test_removal_batch_without_a_removal_is_the_plain_replay(oracle):
    model, events, lim = raft5_config2()
    vv, rec, used = _violating_execution(oracle, model, events, lim)
    target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
    rng = np.random.default_rng(5)
    masks = rng.integers(0, 1 << 63, size=(16, 4), dtype=np.uint64)
    masks[0] = ~np.uint64(0)
    plain = oracle.sts_replay_batch(model, used, rec, masks, target)
    none = oracle.sts_removal_batch(model, used, rec, np.full(16, NO_SKIP, dtype=np.uint32), target, masks=masks)
    assert (plain == none).all()
    # masks == None keeps every external
    allk = oracle.sts_removal_batch(model, used, rec, [NO_SKIP], target)
    assert allk[0] == plain[0]


def test_kept_marks_are_the_executed_trace(oracle):
    """Full replay: every Spawn, external MsgSend and MsgEvent is kept, and so is the internal / timer MsgSend of every
    delivered message (the trace test() returns pairs each delivery with its send: DepTracker and the DPOR initial
    trace need that); sends of messages that were never delivered and quiescence records are not.  The executed trace
    replays to the same verdict (hash over deliveries and final states)."""
    model, events, lim = raft5_config2()
    vv, rec, used = _violating_execution(oracle, model, events, lim)
    target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
    v, kept = oracle.sts_removal_kept(model, used, rec, NO_SKIP, target)
    assert v.flags & T.V_VIOLATION and v.hash == vv.hash
    kinds = rec["kind"]
    core = (kinds <= T.REC_UNPARTITION) | (kinds == T.REC_MSG_EVENT) | ((kinds == T.REC_MSG_SEND) & ((rec["flags"] & 1) == 1))
    delivered_ids = set(rec["id"][kinds == T.REC_MSG_EVENT].tolist())
    internal_send = (kinds == T.REC_MSG_SEND) & ((rec["flags"] & 1) == 0)
    expect = core | (internal_send & np.isin(rec["id"], list(delivered_ids)))
    assert (kept.astype(bool) == expect).all() and (internal_send & expect).any() and (internal_send & ~expect).any()
    # every delivery of the executed trace has its send in it
    tr0 = executed_trace(EventTrace(rec, used), kept).events
    assert set(tr0["id"][tr0["kind"] == T.REC_MSG_EVENT].tolist()) <= set(tr0["id"][tr0["kind"] == T.REC_MSG_SEND].tolist())
    tr = executed_trace(EventTrace(rec, used), kept)
    v2, kept2 = oracle.sts_removal_kept(model, tr.original_externals, tr.events, NO_SKIP, target)
    assert v2.flags == v.flags and v2.hash == v.hash and kept2.all()
    # removing one delivery: it is not delivered, is not counted as "ignored" by itself, and whatever it caused
    # is now absent (those expected deliveries are ignored -> DIVERGED) or the run is simply one shorter
    dl = deliveries(EventTrace(rec, used))
    for idx, key, _ in dl[:12]:
        v3, kept3 = oracle.sts_removal_kept(model, used, rec, idx, target)
        assert not kept3[idx]
        assert T.verdict_deliveries(v3.flags) <= T.verdict_deliveries(v.flags) - 1
        absent = int(core.sum()) - 1 - int(kept3[core].sum())
        assert bool(v3.flags & T.V_DIVERGED) == (absent > 0)
        b = oracle.sts_removal_batch(model, used, rec, [idx], target)[0]
        assert int(b["flags"]) == v3.flags and int(b["hash"]) == v3.hash


def test_executed_traces_feed_the_dpor_initial_trace(oracle):
    """A trace returned by an STSSched replay (DDMin's verified MCS, every lastFailingTrace of internal minimization) is
    consumed by later stages that pair each delivery with its send (DepTracker / editDistanceDporDDMin): every MsgEvent of
    it must find its MsgSend."""
    from demi_amd.incremental_ddmin import dpor_initial_trace
    model, events, lim = raft5_config2()
    trace, fp = _verified_mcs(oracle, model, events, lim)
    it = dpor_initial_trace(trace)
    assert len(it) == 1 + int((trace.events["kind"] == T.REC_MSG_EVENT).sum())
    # ... and after a removal (internal minimization's candidate that still fails)
    dl = deliveries(trace)
    target = T.Limits(0, 0, 64, 1, fp.code, 0)
    for idx, _, _ in dl[:6]:
        v, kept = oracle.sts_removal_kept(model, trace.original_externals, trace.events, idx, target)
        sub = executed_trace(trace, kept)
        assert len(dpor_initial_trace(sub)) == 1 + int((sub.events["kind"] == T.REC_MSG_EVENT).sum())


def test_rebased_verified_mcs_replays_identically(oracle):
    model, events, lim = raft5_config2()
    trace, fp = _verified_mcs(oracle, model, events, lim)
    assert len(trace.original_externals) < 50
    assert not (trace.original_externals["kind"] == T.EV_WAIT_QUIESCENCE).any()
    ext_sends = trace.events[(trace.events["kind"] == T.REC_MSG_SEND) & ((trace.events["flags"] & 1) == 1)]
    assert len(ext_sends) > 0
    assert (trace.original_externals["kind"][ext_sends["ext_idx"]] == T.EV_SEND).all()
    v, kept = oracle.sts_removal_kept(model, trace.original_externals, trace.events, NO_SKIP, T.Limits(0, 0, 64, 1, fp.code, 0))
    assert v.flags & T.V_VIOLATION and not (v.flags & T.V_DIVERGED) and kept.all()


def _literal_get_next_trace(strategy, trace, alreadyRemoved):
    """OneAtATimeStrategy.getNextTrace transliterated with the full flatMap (no early return), LeftToRight filter."""
    keys = Counter(alreadyRemoved)
    found = [None]

    def check(key):
        keys[key] += 1
        if found[0] is not None:
            return True
        if keys[key] > strategy[key]:
            found[0] = key
            strategy[key] += 1
            return False
        return True

    out = []
    for i, e in enumerate(trace.events):
        if e["kind"] == T.REC_MSG_EVENT:
            key = (int(e["snd"]), int(e["rcv"]), (int(e["msg_type"]), int(e["p0"]), int(e["p1"])))
            if not check(key):
                continue
        out.append(i)
    return out if found[0] is not None else None


def test_left_to_right_proposals_match_the_literal_transliteration(oracle):
    model, events, lim = raft5_config2()
    trace, fp = _verified_mcs(oracle, model, events, lim)
    s = LeftToRightOneAtATime(trace, model)
    tried = Counter(s.triedIgnoring)
    assert s.unignorable == sum(1 for _, k, _ in deliveries(trace) if model.msg_class[k[2][0]] == T.MSG_EXTERNAL)
    n = 0
    while True:
        kept_idx = _literal_get_next_trace(tried, trace, Counter())
        nxt = s.getNextTrace(trace, Counter(), False)
        if kept_idx is None:
            assert nxt is None
            break
        assert (nxt.events == trace.events[kept_idx]).all()
        n += 1
    assert n == countMsgEvents(trace) - s.unignorable


@pytest.mark.parametrize("strategy_cls", [LeftToRightOneAtATime, SrcDstFIFORemoval])
@pytest.mark.parametrize("skip", [0, 1, 2])
def test_speculative_minimizer_equals_the_sequential_loop(oracle, strategy_cls, skip):
    model, events, lim = raft5_config2()
    trace, fp = _verified_mcs(oracle, model, events, lim, skip)
    mcs = trace.original_externals
    seq_or, spec_or = OracleRemoval(oracle, model), OracleRemoval(oracle, model)
    seq = STSSchedMinimizer(mcs, trace, fp, strategy_cls(trace, model), seq_or, max_batch=1)
    s1, t1 = seq.minimize()
    spec = STSSchedMinimizer(mcs, trace, fp, strategy_cls(trace, model), spec_or)
    s2, t2 = spec.minimize()
    mid = STSSchedMinimizer(mcs, trace, fp, strategy_cls(trace, model), OracleRemoval(oracle, model), max_batch=7)
    s3, t3 = mid.minimize()
    assert (t1.events == t2.events).all() and (t1.events == t3.events).all()
    assert s1.total_replays == s2.total_replays == s3.total_replays
    assert seq.internal_sizes == spec.internal_sizes == mid.internal_sizes
    assert spec_or.launches < seq_or.launches or s1.total_replays <= 1
    # the minimized schedule still triggers the violation, follows exactly (nothing absent), and is no longer
    v, kept = oracle.sts_removal_kept(model, mcs, t1.events, NO_SKIP, T.Limits(0, 0, 64, 1, fp.code, 0))
    assert v.flags & T.V_VIOLATION and not (v.flags & T.V_DIVERGED) and kept.all()
    assert countMsgEvents(t1) <= countMsgEvents(trace)
    # externals' deliveries are never removed
    ext = [k for k in getFingerprintedDeliveries(trace) if model.msg_class[k[2][0]] == T.MSG_EXTERNAL]
    left = Counter(getFingerprintedDeliveries(t1))
    for k, c in Counter(ext).items():
        assert left[k] >= c or countMsgEvents(t1) < countMsgEvents(trace)


def test_left_to_right_result_is_one_pass_minimal_on_a_chain(oracle):
    """Two independent Kick->Ping chains; the violation needs only the Pings of one of them, so the other chain's
    Ping delivery is removable, while the needed ones are not."""
    MSGS = [("Kick", T.MSG_EXTERNAL), ("Ping", T.MSG_INTERNAL), ("Noise", T.MSG_INTERNAL)]
    h = {(0, "Kick"): M.Asm().mov(M.T0, 1).mov(M.T1, 2).if_eq(M.ME, 0, "x").send(1, M.T0, M.P0, 0).send(2, M.T1, M.P0, 0).label("x"),
         (0, "Ping"): M.Asm().add(M.F[1], M.F[1], 1),
         (0, "Noise"): M.Asm().add(M.F[2], M.F[2], 1)}
    model = M.build_model("chain", 3, MSGS, h, [[0] * 8] * 3, (T.INV_NEVER, 1, 2, 0))     # actor saw 2 Pings
    ev = events_to_array([start(0), start(1), start(2), send(0, 0, 1), send(0, 0, 2), wait_quiescence()])
    lim = T.Limits(0, 0, 64, 0, 0, 0)
    vv, rec, _ = oracle.random_execute(model, ev, 3, lim)
    assert vv.flags & T.V_VIOLATION
    fp = ViolationFingerprint(vv.fingerprint)
    mcs = [0, 1, 2, 3, 4]
    v, kept = oracle.sts_removal_kept(model, ev, rec, NO_SKIP, T.Limits(0, 0, 64, 1, fp.code, 0),
                                      mask=np.array(events_to_mask(mcs), dtype=np.uint64))
    trace = executed_trace(EventTrace(rec, ev), kept, subseq=mcs)
    before = Counter(k[2][0] for k in getFingerprintedDeliveries(trace))
    assert before == Counter({0: 2, 1: 2, 2: 2})
    stats, out = STSSchedMinimizer(trace.original_externals, trace, fp, LeftToRightOneAtATime(trace, model),
                                   OracleRemoval(oracle, model)).minimize()
    after = Counter(k[2][0] for k in getFingerprintedDeliveries(out))
    assert after == Counter({0: 2, 1: 2})            # both Noise deliveries pruned, Kicks (external) and Pings stay
    assert stats.total_replays == 4                   # 2 Pings tried and kept, 2 Noises tried and dropped


class _ScalaSrcDstFIFORemoval:
    """OneAtATimeRemoval.scala:141-251 transliterated with the reference's own data structures (HashMap of Vectors,
    MultiSets as Counters, flatMap over the whole trace), to cross-check demi_amd's index-returning version."""

    def __init__(self, verified, model):
        self.verified = verified
        self.tried = Counter()
        for e in verified.events:
            if e["kind"] == T.REC_MSG_EVENT and model.msg_class[int(e["msg_type"])] == T.MSG_EXTERNAL:
                self.tried[self._key(e)] += 1
        self.srcDstToMessages = {}
        for e in verified.events:
            if e["kind"] == T.REC_MSG_EVENT and int(e["snd"]) != T.DEADLETTERS:
                self.srcDstToMessages.setdefault((int(e["snd"]), int(e["rcv"])), []).append(self._key(e)[2])
        self.previouslyChosenSrcDst = None
        self.srcDstToCurrentIdx = {}
        self._reset()

    @staticmethod
    def _key(e):
        return (int(e["snd"]), int(e["rcv"]), (int(e["msg_type"]), int(e["p0"]), int(e["p1"])))

    def _reset(self):
        for k in self.srcDstToMessages:
            self.srcDstToCurrentIdx[k] = -1

    def choiceFilter(self, snd, rcv, fp):
        if (snd, rcv) in self.srcDstToMessages:
            self.srcDstToCurrentIdx[(snd, rcv)] += 1
            idx = self.srcDstToCurrentIdx[(snd, rcv)]
            lst = self.srcDstToMessages[(snd, rcv)]
            if idx == len(lst) - 1:
                self.srcDstToMessages[(snd, rcv)] = lst[:-1]
                if not self.srcDstToMessages[(snd, rcv)]:
                    del self.srcDstToMessages[(snd, rcv)]
                self.previouslyChosenSrcDst = (snd, rcv)
                return True
        self.previouslyChosenSrcDst = None
        return snd == T.DEADLETTERS

    def getNextTrace(self, trace, alreadyRemoved, violationTriggeredLastRun):
        if not violationTriggeredLastRun and self.previouslyChosenSrcDst is not None:
            self.srcDstToMessages.pop(self.previouslyChosenSrcDst, None)
        if violationTriggeredLastRun:
            self.srcDstToMessages.clear()
            copy = Counter(alreadyRemoved)
            for e in self.verified.events[::-1]:
                if e["kind"] != T.REC_MSG_EVENT or int(e["snd"]) == T.DEADLETTERS:
                    continue
                t = self._key(e)
                if copy[t] > 0:
                    copy[t] -= 1
                else:
                    self.srcDstToMessages[(t[0], t[1])] = [t[2]] + self.srcDstToMessages.get((t[0], t[1]), [])
        self._reset()
        keysThisIteration = Counter(alreadyRemoved)
        found = [False]

        def checkDelivery(key):
            keysThisIteration[key] += 1
            if found[0]:
                return True
            if keysThisIteration[key] > self.tried[key] and self.choiceFilter(*key):
                found[0] = True
                self.tried[key] += 1
                return False
            return True

        kept = [i for i, e in enumerate(trace.events) if e["kind"] != T.REC_MSG_EVENT or checkDelivery(self._key(e))]
        return kept if found[0] else None


@pytest.mark.parametrize("skip", [0, 1])
def test_srcdst_fifo_removal_matches_the_scala_transliteration(oracle, skip):
    """Drive both versions through a whole minimization (successes and failures as the oracle decides them) and
    compare every proposed trace."""
    model, events, lim = raft5_config2()
    trace, fp = _verified_mcs(oracle, model, events, lim, skip)
    ours, ref = SrcDstFIFORemoval(trace, model), _ScalaSrcDstFIFORemoval(trace, model)
    orc = OracleRemoval(oracle, model)
    last, pruned, triggered, steps = EventTrace(trace.events, trace.original_externals), Counter(), False, 0
    while True:
        kept = ref.getNextTrace(last, pruned, triggered)
        nxt = ours.getNextTrace(last, pruned, triggered)
        if kept is None:
            assert nxt is None
            break
        assert (nxt.events == last.events[kept]).all()
        removed = [i for i in range(len(last.events)) if i not in set(kept)]
        assert len(removed) == 1
        executed = orc.executed(last, removed[0], fp)
        triggered = executed is not None
        if triggered:
            pruned += Counter(getFingerprintedDeliveries(last)) - Counter(getFingerprintedDeliveries(executed))
            last = EventTrace(executed.events, trace.original_externals)
        steps += 1
    assert steps > 10
