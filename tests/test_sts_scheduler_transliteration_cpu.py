"""CPU suite: STSScheduler.test (no peek) transliterated from the Scala, against the oracle (K2's restatement).

  STSScheduler.test / advanceReplay / messagePending / event_produced / schedule_new_message / notify_timer_cancel /
    enqueue_timer                          (schedulers/STSScheduler.scala:199-310, 392-402, 405-559, 561-623, 643-776, 828-869)
with pendingEvents as the reference has it - HashMap[(snd, rcv), HashMap[fingerprint, Queue]] - on top of the
ExternalEventInjector / EventOrchestrator / Instrumenter pieces of tests/test_random_scheduler_transliteration_cpu.py and
the trace projection of tests/test_minification_cpu.py (subsequenceIntersection + filterSends [+ filterKnownAbsentInternals]).
STSScheduler.test replays the projected original trace: an expected MsgEvent is delivered iff it is pending (and its
receiver not blocked), else ignored; the verdict is the invariant at the end matched against the target fingerprint."""
import ctypes as C
from collections import OrderedDict, deque

import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import FuzzerWeights, events_to_array, raft_trace
from demi_amd.minification import events_to_mask

from .test_minification_cpu import _scala_filter_known_absent_internals, _scala_subsequence_intersection
from .test_random_scheduler_transliteration_cpu import DEAD, MASK64, ScalaRandomScheduler


class ScalaSTSScheduler(ScalaRandomScheduler):
    def __init__(self, oracle, model, externals, rec, subseq, filterKnownAbsents=0):
        super().__init__(oracle, model, externals[:0], 0, 0, 0)
        self.rec = rec
        # populateActorSystem(original_trace.getEvents flatMap { case SpawnEvent(_, props, name, _) => ... }) (:227-236)
        for e in rec:
            if int(e["kind"]) == T.REC_SPAWN:
                self.actorToActorRef.add(int(e["rcv"]))
                self.inaccessible.add(int(e["rcv"]))
        self.exists = sum(1 << a for a in self.actorToActorRef)
        # original_trace.subsequenceIntersection(subseq, filterKnownAbsents) (:237-240); recomputeExternalMsgSends re-creates the
        # Sends' messages, which are plain values here
        proj = _scala_subsequence_intersection(rec, externals, subseq, model)
        if filterKnownAbsents:
            proj = _scala_filter_known_absent_internals(rec, proj, corrected=(filterKnownAbsents == T.FILTER_ABSENTS_CORRECTED))
        self.trace = [rec[i] for i in proj]
        self.pendingEvents = OrderedDict()            # (snd, rcv) -> {fingerprint -> Queue}
        self.ignored = 0

    # ------------------------------------------------------------------ event_produced (:561-623)
    def tell(self, snd, rcv, msg):
        uniq = self.next_uniq
        self.next_uniq += 1
        if self.enqueuedExternalMessages[msg] > 0 or not self.crosses_partition(snd, rcv):
            self.pendingEvents.setdefault((snd, rcv), OrderedDict()).setdefault(msg, deque()).append(uniq)

    def messagePending(self, sender, receiver, msg):
        self.send_external_messages()
        q = self.pendingEvents.get((sender, receiver), {}).get(msg)
        if q is not None:
            return receiver not in self.blockedActors
        return False

    def advanceReplay(self):
        while not self.trace_finished():
            e = self.trace[self.traceIdx]
            kind = int(e["kind"])
            snd = DEAD if int(e["snd"]) == T.DEADLETTERS else int(e["snd"])
            rcv = int(e["rcv"])
            msg = (int(e["msg_type"]), int(e["p0"]), int(e["p1"]))
            if kind == T.REC_SPAWN:                    # trigger_start
                self.inaccessible.discard(rcv)
                self.killed.discard(rcv)
                self.blockedActors.discard(rcv)
            elif kind == T.REC_KILL:
                self.killed.add(rcv)
                self.inaccessible.add(rcv)
            elif kind == T.REC_PARTITION:
                self.partitioned.add((int(e["snd"]), rcv))
            elif kind == T.REC_UNPARTITION:
                self.partitioned.discard((int(e["snd"]), rcv))
            elif kind == T.REC_MSG_SEND:
                if int(e["flags"]) & 1:                # EventTypes.isExternal(m)
                    self.enqueue_message(None, rcv, msg)
            elif kind == T.REC_MSG_EVENT:
                if self.messagePending(snd, rcv, msg):
                    break                              # "Yay, it's already enabled."
                self.ignored += 1                      # "Ignoring message"
            self.traceIdx += 1

    def schedule_new_message(self):
        self.send_external_messages()
        self.advanceReplay()
        self.send_external_messages()
        if self.trace_finished():
            return None
        e = self.trace[self.traceIdx]
        snd = DEAD if int(e["snd"]) == T.DEADLETTERS else int(e["snd"])
        rcv, msg = int(e["rcv"]), (int(e["msg_type"]), int(e["p0"]), int(e["p1"]))
        inner = self.pendingEvents[(snd, rcv)]
        q = inner[msg]
        q.popleft()
        if not q:
            del inner[msg]
            if not inner:
                del self.pendingEvents[(snd, rcv)]
        self.traceIdx += 1
        self.messagesScheduledSoFar += 1
        return (snd, rcv, msg, 0)

    def notify_timer_cancel(self, rcv, msg):
        if self.handle_timer_cancel(rcv, msg):
            return
        inner = self.pendingEvents.get((DEAD, rcv))
        if inner is not None and msg in inner:
            inner[msg].popleft()                       # queue.dequeueFirst(t => message == msg)
            if not inner[msg]:
                del inner[msg]
                if not inner:
                    del self.pendingEvents[(DEAD, rcv)]

    def enqueue_timer(self, receiver, msg):
        self.handle_timer(receiver, msg)

    def test(self, looking_for, match_mask):
        self.advanceReplay()
        while True:
            nxt = self.schedule_new_message()
            if nxt is None:
                break
            self.dispatch_new_message(nxt[0], nxt[1], nxt[2])
        assert self.trace_finished()
        fp = self.test_invariant()
        found = looking_for if fp and ((fp ^ looking_for) & match_mask) == 0 else 0
        h = 0xCBF29CE484222325
        for snd, rcv, mtype, p0, p1 in self.deliveries:
            h = ((h ^ (mtype | (rcv << 5) | (snd << 8) | (p0 << 16) | (p1 << 24))) * 0x100000001B3) & MASK64
        for a in range(self.model.n_actors):
            for w in self.state[a]:
                h = ((h ^ w) * 0x100000001B3) & MASK64
        flags = (T.V_VIOLATION if found else 0) | (T.V_DIVERGED if self.ignored else 0) | min(self.messagesScheduledSoFar, 0xFFFF) << 16
        return flags, found, h


def _check(oracle, model, used, rec, fpc, subseqs, filter_mode=0):
    checked = diverged = 0
    lim = T.Limits(0, 0, 128, 1, fpc, 0, 0, filter_mode)
    for subseq in subseqs:
        mask = np.array(events_to_mask(subseq), dtype=np.uint64)
        v, kept = oracle.sts_removal_kept(model, used, rec, 0xFFFFFFFF, lim, mask=mask)
        if v.flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF):
            continue
        s = ScalaSTSScheduler(oracle, model, used, rec, subseq, filter_mode)
        got = s.test(fpc, model.fp_match_mask)
        want_deliveries = [(int(e["snd"]), int(e["rcv"]), int(e["msg_type"]), int(e["p0"]), int(e["p1"]))
                           for e, k in zip(rec, kept) if k and int(e["kind"]) == T.REC_MSG_EVENT]
        assert s.deliveries == want_deliveries
        assert got == (int(v.flags), int(v.fingerprint), int(v.hash))
        checked += 1
        diverged += int(bool(v.flags & T.V_DIVERGED))
    return checked, diverged


def test_replays_of_a_violating_raft5_execution_equal_the_scala_transliteration(oracle):
    model, events, lim = raft5_config2()
    v = oracle.random_explore(model, events, 200, seed_base=SEED_BASE, limits=lim)
    i0 = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + i0, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    rng = np.random.default_rng(4)
    noq = [i for i in range(len(used)) if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE]
    subseqs = [noq] + [[i for i in noq if rng.random() < p] for p in (0.3, 0.5, 0.7, 0.9, 0.95) for _ in range(8)]
    checked, diverged = _check(oracle, model, used, rec, vv.fingerprint, subseqs)
    assert checked == len(subseqs) and 0 < diverged < checked
    # the unmodified trace replays the recorded execution: same deliveries, same hash, the violation
    full = ScalaSTSScheduler(oracle, model, used, rec, noq).test(vv.fingerprint, model.fp_match_mask)
    assert full[0] & T.V_VIOLATION and not full[0] & T.V_DIVERGED and full[2] == vv.hash


@pytest.mark.parametrize("filter_mode", [0, T.FILTER_ABSENTS_LITERAL, T.FILTER_ABSENTS_CORRECTED])
def test_fault_heavy_replays_equal_the_scala_transliteration(oracle, filter_mode):
    model = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.12, send=0.4, wait_quiescence=0.13, partition=0.2, unpartition=0.15)
    rng = np.random.default_rng(23)
    total = 0
    for seed in (1, 2, 3):
        events = events_to_array(raft_trace(5, 70, seed, w, exact=False))
        vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + seed, T.Limits(300, 10, 128, 0, 0, 0))
        used = events[:T.verdict_trace_idx(vv.flags)]
        noq = [i for i in range(len(used)) if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE]
        subseqs = [noq] + [[i for i in noq if rng.random() < p] for p in (0.4, 0.7, 0.95) for _ in range(6)]
        c, _ = _check(oracle, model, used, rec, vv.fingerprint if vv.fingerprint else 0x1000103, subseqs, filter_mode)
        total += c
    assert total >= 50


def test_crashing_application_replays_equal_the_scala_transliteration(oracle):
    from .test_blocked_actors_gpu import crashy_model, crashy_trace
    model, events = crashy_model(), crashy_trace()
    vv, rec, _ = oracle.random_execute(model, events, 424242, T.Limits(300, 0, 64, 0, 0, 0))
    used = events[:T.verdict_trace_idx(vv.flags)]
    rng = np.random.default_rng(6)
    noq = [i for i in range(len(used)) if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE]
    subseqs = [noq] + [[i for i in noq if rng.random() < p] for p in (0.5, 0.8) for _ in range(10)]
    c, _ = _check(oracle, model, used, rec, vv.fingerprint if vv.fingerprint else 0x1000103, subseqs)
    assert c >= 15


def test_replays_on_a_table_with_arrays_equal_the_scala_transliteration(oracle):
    """DEMI_MODEL_ARRAY: candidate subsequences of a violating execution of the replicated-log protocol (the hole)."""
    from demi_amd import model as M
    from demi_amd.fuzzer import events_to_array, send, start
    model = M.replog_model(4, 6, True, False)
    events = events_to_array([start(a) for a in range(4)] + [send(0 if i % 3 else i % 4, M.RL_PUT, 30 + i, 0) for i in range(7)])
    lim = T.Limits(300, 0, 128, 0, 0, 0)
    v = oracle.random_explore(model, events, 100, seed_base=500, limits=lim)
    i0 = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec, _ = oracle.random_execute(model, events, 500 + i0, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    rng = np.random.default_rng(9)
    allx = list(range(len(used)))
    subseqs = [allx] + [[i for i in allx if rng.random() < p] for p in (0.4, 0.6, 0.8, 0.9) for _ in range(8)]
    checked, diverged = _check(oracle, model, used, rec, vv.fingerprint, subseqs)
    assert checked == len(subseqs) and 0 < diverged < checked


def test_the_bench_candidates_by_the_transliteration_are_the_oracles(oracle):
    """tests/golden/replay_config4_transliteration.json: ALL 2^20 candidate subsequences of the bench's replay workload (config 4)
    as ScalaSTSScheduler above replayed them (tools/check_replay_transliteration.py, eight processes, a quarter of an hour).  The
    C oracle gives the same bytes - re-computed here for the first 2^16 -; the GPU suite holds K2's verdicts of all 2^20 against the
    same record
    (test_bench_candidates_against_the_sts_transliterations_record)."""
    import hashlib
    import json
    import os
    from demi_amd.apps import raft5_config4
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "replay_config4_transliteration.json")) as f:
        rec_t = json.load(f)
    assert "ScalaSTSScheduler" in rec_t["generator"] and rec_t["equals_the_oracle"] is True
    assert rec_t["candidates"] == 1 << 20 and rec_t["first"] == 1 << 16      # (all of the bench's candidates; re-computed here: the first 2^16)
    model, events, lim = raft5_config4()
    v = oracle.random_explore(model, events, 4000, seed_base=SEED_BASE, limits=lim)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    assert len(used) == rec_t["externals"] and len(rec) == rec_t["recorded_events"]
    n = rec_t["first"]
    keep = np.random.default_rng(0).random((n, len(used))) < 0.7
    masks = np.zeros((n, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    assert hashlib.sha256(masks.tobytes()).hexdigest() == rec_t["sha256_masks"]
    got = oracle.sts_replay_batch(model, used, rec, masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0), n_threads=os.cpu_count() or 1)
    assert hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest() == rec_t["sha256_verdicts"]
    assert int(((got["flags"] & T.V_VIOLATION) != 0).sum()) == rec_t["still_violating_of_the_first"]
