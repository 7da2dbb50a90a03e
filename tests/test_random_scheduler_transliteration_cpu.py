"""CPU suite: one RandomScheduler execution transliterated from the Scala, against the oracle (K1's restatement).

What is transliterated, with the reference's own data structures (queues of tuples, hash sets, the multiset of enqueued
external messages, RandomizedHashSet with its swap-with-last removal):
  RandomScheduler.explore / event_produced / schedule_new_message / updateRepeatingTimer / notify_quiescence /
    notify_timer_cancel / enqueue_timer                        (schedulers/RandomScheduler.scala:234-272, 282-321, 352-485, 487-500, 525-559)
  FullyRandom (+=, remove, removeRandomElement)                (:631-684)
  ExternalEventInjector.enqueue_message / handle_timer / send_external_messages / execute_trace / advanceTrace /
    handle_event_produced / handle_event_consumed / handle_quiescence / handle_timer_cancel
                                                               (schedulers/ExternalEventInjector.scala:250-297, 299-365, 382-441, 492-512, 529-580, 601-610)
  EventOrchestrator.inject_until_quiescence / trigger_* / isolate / unisolate / crosses_partition
                                                               (schedulers/EventOrchestrator.scala:132-189, 192-241, 314-330, 345-351)
  Instrumenter: the dispatch step with the repeating-timer retrigger, registerCancellable / handleTick / removeCancellable /
    cancelTimer / actorCrashed / seededRandom                  (Instrumenter.scala:159-168, 184-199, 212-229, 1008-1016, 1145-1200)
  Util.find_non_blocked_message, RandomizedHashSet            (schedulers/Util.scala:110-185, 470-489; literal classes of
                                                                tests/test_blocked_actors_cpu.py)
The actors' `receive` is the table (oracle's row interpreter, one call per delivery): what is pinned here is the scheduler.
Pinned as in the oracle: a repeating timer's retrigger runs before the receive it races with."""
import ctypes as C
from collections import Counter

import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.fuzzer import FuzzerWeights, JavaRandom, events_to_array, raft_trace, send, start, wait_quiescence

from .test_blocked_actors_cpu import RandomizedHashSet, find_non_blocked_message

MASK64 = (1 << 64) - 1
DEAD = "deadLetters"


class _Effect(C.Structure):      # orc_effect (oracle/demi_oracle.h)
    _fields_ = [("kind", C.c_uint8), ("target", C.c_uint8), ("msg_type", C.c_uint8), ("p0", C.c_uint16), ("p1", C.c_uint16),
                ("area", C.c_uint64)]


class FullyRandom:
    """RandomScheduler.scala:631-684 over the literal RandomizedHashSet; an element is (snd, rcv, msg, uniq id)."""

    def __init__(self, seed):
        self.pendingEvents = RandomizedHashSet(seed)

    def add(self, e):
        self.pendingEvents.insert(e)

    def isEmpty(self):
        return self.pendingEvents.isEmpty()

    def insert(self, e):                      # Growable.+= (find_non_blocked_message's `collection ++= blocked`)
        self.pendingEvents.insert(e)

    def remove(self, snd, rcv, msg):
        arr = self.pendingEvents.arr
        for i, e in enumerate(arr):
            if snd == e[0] and rcv == e[1] and msg == e[2]:
                arr[i] = arr[-1]              # RandomizedHashSet.remove: A[i] = d, drop the last cell
                arr.pop()
                return e
        return None

    def removeRandomElement(self):
        # userDefinedFilter is the default (always true): the rejection loop never runs
        return self.pendingEvents.removeRandomElement()


class ScalaRandomScheduler:
    def __init__(self, oracle, model, trace, seed, maxMessages, invariant_check_interval, strategy=None):
        """strategy: the FullyRandom of an earlier execution of the same scheduler instance (reset_all_state keeps the object
        and calls pendingEvents.clear(), RandomScheduler.scala:584: its Random is not reseeded)."""
        self.oracle, self.model, self.ms = oracle, model, model.to_struct()
        self.trace = [tuple(int(x) for x in (e["kind"], e["a"], e["b"], e["msg_type"], e["p0"], e["p1"])) for e in trace]
        self.maxMessages = maxMessages if maxMessages else (1 << 31) - 1
        self.invariant_check_interval = invariant_check_interval
        A = model.n_actors
        # ---- EventOrchestrator
        self.traceIdx = 0
        self.partitioned, self.inaccessible, self.killed = set(), set(), set()
        self.actorToActorRef = set()
        # ---- ExternalEventInjector
        self.enqueuedExternalMessages = Counter()
        self.messagesToSend = []                      # (senderOpt, receiver, msg)
        # ---- RandomScheduler
        self.pendingEvents = strategy if strategy is not None else FullyRandom(seed)
        self.justScheduledTimers, self.timersToResend = set(), []
        self.violationFound = None
        self.messagesScheduledSoFar = 0
        self.finished_early = False
        # ---- Instrumenter
        self.timerToCancellable, self.ongoingCancellableTasks, self.registeredCancellableTasks = {}, set(), set()
        self.next_cancellable = 0
        self.blockedActors = set()
        self.seededRandom = C.c_uint64((0 ^ 0x5DEECE66D) & ((1 << 48) - 1))       # new Random(0)
        # ---- the application
        # (an actor's state: its field word, then - DEMI_MODEL_ARRAY - the words of its array, empty at the start)
        # (a wide table - 16-bit fields - has two field words per actor; a table with more than 8 actors is wide and names
        # deadLetters 31 in the register window and in the recorded trace: the BIG layout of include/demi_gpu.h)
        self.stw = getattr(model, "state_words", 1)
        assert getattr(model, "payloads", 2) == 2       # (messages are (type, p0, p1) here: tables of two payload fields)
        self.wide = bool(getattr(model, "wide", False))
        self.big = A > T.MAX_ACTORS
        self.dl = T.DEADLETTERS_BIG if self.big else T.DEADLETTERS
        fw = 2 if self.wide else 1
        self.state = [[int(w) for w in model.init_state[a * fw:(a + 1) * fw]] + [0] * (self.stw - fw) for a in range(A)]
        self.deliveries = []
        self.next_uniq = 1
        # populateActorSystem: every actor that is ever Start()ed is created and isolated (:371-378, 397-406)
        for kind, a, *_ in self.trace:
            if kind == T.EV_START:
                self.actorToActorRef.add(a)
                self.inaccessible.add(a)
        self.exists = sum(1 << a for a in self.actorToActorRef)

    # ------------------------------------------------------------------ EventOrchestrator
    def trace_finished(self):
        return self.traceIdx >= len(self.trace)

    def crosses_partition(self, snd, rcv):
        if snd == rcv and snd not in self.killed:
            return False
        return ((snd, rcv) in self.partitioned or (rcv, snd) in self.partitioned or rcv in self.inaccessible
                or snd in self.inaccessible)

    def inject_until_quiescence(self):
        loop = True
        while loop and not self.trace_finished():
            kind, a, b, mtype, p0, p1 = self.trace[self.traceIdx]
            if kind == T.EV_START:                    # trigger_start: unisolate_node + blockedActors -= name
                self.inaccessible.discard(a)
                self.killed.discard(a)
                self.blockedActors.discard(a)
            elif kind == T.EV_KILL:                   # trigger_kill
                self.killed.add(a)
                self.inaccessible.add(a)
            elif kind == T.EV_SEND:
                self.enqueue_message(None, a, (mtype, p0, p1))
            elif kind == T.EV_PARTITION:
                self.partitioned.add((a, b))
            elif kind == T.EV_UNPARTITION:
                self.partitioned.discard((a, b))
            elif kind == T.EV_WAIT_QUIESCENCE:
                loop = False
            self.traceIdx += 1                        # trace_advanced()

    # ------------------------------------------------------------------ ExternalEventInjector
    def enqueue_message(self, sender, receiver, msg):
        if receiver not in self.actorToActorRef:      # "Unknown message receiver"
            return
        self.enqueuedExternalMessages[msg] += 1
        self.messagesToSend.append((sender, receiver, msg))

    def handle_timer(self, receiver, msg):
        if receiver in self.actorToActorRef:
            self.messagesToSend.append((None, receiver, msg))

    def send_external_messages(self):
        for senderOpt, receiver, msg in self.messagesToSend:
            # Instrumenter().receiverIsAlive(receiver): the actor exists
            if receiver in self.actorToActorRef:
                self.tell(DEAD if senderOpt is None else senderOpt, receiver, msg)        # receiver ! msg
        self.messagesToSend = []

    def handle_timer_cancel(self, rcv, msg):
        for i, (s, r, m) in enumerate(self.messagesToSend):     # dequeueFirst
            if r == rcv and m == msg:
                del self.messagesToSend[i]
                return True
        return False

    # ------------------------------------------------------------------ RandomScheduler
    def tell(self, snd, rcv, msg):
        """`rcv ! msg` under the instrumentation: event_produced (:282-321)."""
        uniq = self.next_uniq
        self.next_uniq += 1
        if self.enqueuedExternalMessages[msg] > 0:                  # handle_event_produced -> ExternalMessage
            self.pendingEvents.add((snd, rcv, msg, uniq))
        else:                                                        # InternalMessage (a timer when snd == deadLetters)
            if not self.crosses_partition(snd, rcv):
                self.pendingEvents.add((snd, rcv, msg, uniq))

    def test_invariant(self):
        states = (C.c_uint64 * (self.model.n_actors * self.stw))(*[w for st in self.state for w in st])
        return int(self.oracle.lib().orc_invariant(C.byref(self.ms), states, self.exists))

    def isTimer(self, rcv, msg):
        return (rcv, msg) in self.timerToCancellable

    def updateRepeatingTimer(self, rcv, msg):
        if self.isTimer(rcv, msg):
            self.justScheduledTimers.add((rcv, msg))
        else:
            for r, t in self.timersToResend:
                self.handle_timer(r, t)
            self.timersToResend = []
            self.justScheduledTimers.clear()

    def schedule_new_message(self):
        if self.violationFound:
            return None
        if self.messagesScheduledSoFar > self.maxMessages:
            self.traceIdx = len(self.trace)                          # event_orchestrator.finish_early
            self.finished_early = True
            return None
        if (self.invariant_check_interval > 0 and self.messagesScheduledSoFar % self.invariant_check_interval == 0
                and 0 != self.messagesScheduledSoFar):               # lastCheckpoint (= 0) != messagesScheduledSoFar
            self.violationFound = self.test_invariant() or None
            if self.violationFound:
                return None
        self.send_external_messages()
        toSchedule = find_non_blocked_message(self.blockedActors, self.pendingEvents, lambda e: e[1])
        if toSchedule is None:
            return None
        self.messagesScheduledSoFar += 1
        snd, rcv, msg, _uniq = toSchedule
        self.updateRepeatingTimer(rcv, msg)
        return toSchedule

    def notify_timer_cancel(self, rcv, msg):
        if self.handle_timer_cancel(rcv, msg):
            return
        self.pendingEvents.remove(DEAD, rcv, msg)

    def enqueue_timer(self, receiver, msg):
        if (receiver, msg) in self.justScheduledTimers:
            self.timersToResend.append((receiver, msg))
            return
        self.handle_timer(receiver, msg)

    # ------------------------------------------------------------------ Instrumenter
    def registerCancellable(self, ongoingTimer, receiver, msg):
        if (receiver, msg) in self.timerToCancellable:              # "Non-unique timer"
            return
        c = self.next_cancellable
        self.next_cancellable += 1
        self.registeredCancellableTasks.add(c)
        if ongoingTimer:
            self.ongoingCancellableTasks.add(c)
        self.timerToCancellable[(receiver, msg)] = c
        self.handleTick(receiver, msg, c)

    def removeCancellable(self, c):
        self.registeredCancellableTasks.discard(c)
        for k, v in list(self.timerToCancellable.items()):
            if v == c:
                del self.timerToCancellable[k]

    def handleTick(self, receiver, msg, c):
        assert c in self.registeredCancellableTasks
        self.enqueue_timer(receiver, msg)
        if c not in self.ongoingCancellableTasks:
            self.removeCancellable(c)

    def cancelTimer(self, rcv, msg):
        c = self.timerToCancellable.get((rcv, msg))
        if c is not None:
            self.ongoingCancellableTasks.discard(c)
            self.removeCancellable(c)
        self.notify_timer_cancel(rcv, msg)

    def dispatch_new_message(self, snd, rcv, msg):
        mtype, p0, p1 = msg
        self.deliveries.append((self.dl if snd == DEAD else snd, rcv, mtype, p0, p1))
        if self.enqueuedExternalMessages[msg] > 0:                  # handle_event_consumed
            self.enqueuedExternalMessages[msg] -= 1
        # "Check if it was a repeating timer. If so, retrigger it" (pinned before the receive)
        c = self.timerToCancellable.get((rcv, msg))
        if c is not None and c in self.ongoingCancellableTasks:
            self.handleTick(rcv, msg, c)
        # the actor's receive
        st = (C.c_uint64 * self.stw)(*self.state[rcv])
        fx = (_Effect * 64)()
        n = self.oracle.lib().orc_vm_run(C.byref(self.ms), rcv, st, mtype, self.dl if snd == DEAD else snd, p0, p1,
                                         self.exists, fx, 64, C.byref(self.seededRandom))
        assert n >= 0
        self.state[rcv] = [int(w) for w in st]
        for e in fx[:n]:
            if e.kind == 0:
                self.tell(rcv, int(e.target), (int(e.msg_type), int(e.p0), int(e.p1)))
            elif e.kind in (1, 2):                                  # scheduleOnce / schedule: the timer message has no payload
                self.registerCancellable(e.kind == 2, rcv, (int(e.msg_type), 0, 0))
            elif e.kind == 3:
                self.cancelTimer(rcv, (int(e.msg_type), 0, 0))
            elif e.kind == 4:                                       # actorCrashed
                self.blockedActors.add(rcv)

    # ------------------------------------------------------------------ explore(): one execution
    def execute(self):
        while True:
            self.inject_until_quiescence()                          # advanceTrace
            while True:                                             # start_dispatch ... after every receive
                nxt = self.schedule_new_message()
                if nxt is None:
                    break
                self.dispatch_new_message(nxt[0], nxt[1], nxt[2])
            # notify_quiescence
            if self.violationFound:
                break
            if not self.trace_finished():                           # handle_quiescence: events += Quiescence; advanceTrace()
                continue
            break
        if self.messagesScheduledSoFar <= self.maxMessages and not self.violationFound:
            self.violationFound = self.test_invariant() or None     # checkIfBugFound
        return self.violationFound

    def verdict(self):
        h = 0xCBF29CE484222325
        for snd, rcv, mtype, p0, p1 in self.deliveries:
            if self.wide:       # the 64-bit word: sender at bit 8, or bit 9 behind the 4-bit receiver of the BIG layout
                w = mtype | (rcv << 5) | (snd << (9 if self.big else 8)) | (p0 << 16) | (p1 << 32)
            else:
                w = mtype | (rcv << 5) | (snd << 8) | (p0 << 16) | (p1 << 24)
            h = ((h ^ w) * 0x100000001B3) & MASK64
        for a in range(self.model.n_actors):
            for w in self.state[a]:
                h = ((h ^ w) * 0x100000001B3) & MASK64
        flags = (T.V_VIOLATION if self.violationFound else 0) | (T.V_MAXMSG if self.finished_early else 0)
        flags |= (self.traceIdx & 0xFF) << 8 | min(self.messagesScheduledSoFar, 0xFFFF) << 16
        return flags, int(self.violationFound or 0), h


def _compare(oracle, model, events, seeds, max_messages, interval, p_max=128):
    lim = T.Limits(max_messages, interval, p_max, 0, 0, 0)
    checked = violations = 0
    for seed in seeds:
        v, rec, _states = oracle.random_execute(model, events, seed, lim)
        if v.flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF):
            continue                                    # a capacity of the restatement, not a behaviour of the reference
        s = ScalaRandomScheduler(oracle, model, events, seed, max_messages, interval)
        s.execute()
        got = [(int(e["snd"]), int(e["rcv"]), int(e["msg_type"]), int(e["p0"]), int(e["p1"])) for e in rec if e["kind"] == T.REC_MSG_EVENT]
        assert got == s.deliveries, "seed %d: delivery sequences differ at %d" % (
            seed, next((i for i, (x, y) in enumerate(zip(got, s.deliveries)) if x != y), min(len(got), len(s.deliveries))))
        assert (int(v.flags), int(v.fingerprint), int(v.hash)) == s.verdict(), "seed %d" % seed
        checked += 1
        violations += int(bool(v.flags & T.V_VIOLATION))
    return checked, violations


def test_raft5_fuzz_executions_equal_the_scala_transliteration(oracle):
    from demi_amd.apps import SEED_BASE, raft5_config2
    model, events, lim = raft5_config2()
    checked, violations = _compare(oracle, model, events, [SEED_BASE + i for i in range(70)], lim.max_messages, lim.invariant_check_interval)
    assert checked == 70 and violations >= 1            # (index 61 is the first violating execution of the frozen workload)
    checked, _ = _compare(oracle, model, events, [7, 8, 9], 0, 0)              # unbounded, invariant only at the end
    assert checked == 3
    checked, _ = _compare(oracle, model, events, [3, 4, 5, 6], 25, 7)          # maxMessages cuts the execution short
    assert checked == 4


def test_kills_partitions_and_restarts_equal_the_scala_transliteration(oracle):
    model = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.12, send=0.35, wait_quiescence=0.13, partition=0.25, unpartition=0.15)
    total = violations = 0
    for tseed in (1, 2, 3, 4, 5, 6):
        events = events_to_array(raft_trace(5, 60, tseed, w, exact=False))
        c, v = _compare(oracle, model, events, [1000 * tseed + i * 7919 for i in range(12)], 300, 10)
        total += c
        violations += v
    assert total >= 60


def test_crashing_and_randomised_applications_equal_the_scala_transliteration(oracle):
    from .test_blocked_actors_gpu import crashy_model, crashy_trace, jittery_model
    c, _ = _compare(oracle, crashy_model(), crashy_trace(), [11 + 104729 * i for i in range(40)], 300, 0, p_max=64)
    assert c >= 30
    ev = events_to_array([start(a) for a in range(4)] + [send(a, 0) for a in range(4)] + [wait_quiescence(), send(1, 0), send(2, 0)])
    c, _ = _compare(oracle, jittery_model(), ev, [0x7E57AB1E0000 + 31 * i for i in range(40)], 150, 11, p_max=64)
    assert c >= 30


def test_carried_generator_instances_equal_the_scala_explore_loop(oracle):
    """explore() with max_executions = k on ONE scheduler instance (RandomScheduler.scala:248-269): reset_all_state (:575-595)
    clears the pending set - RandomizedHashSet.clear, the restatement's true clear - but keeps the FullyRandom and its
    java.util.Random, so execution e + 1 continues the generator where e stopped; lookingFor is None from the second
    execution on (:586); the loop returns at the first violating execution.  The oracle's executions_per_instance mode,
    execution by execution, against the transliteration driven through that loop."""
    from demi_amd.apps import SEED_BASE, raft5_config2
    model, events, lim = raft5_config2()
    k, n_inst = 12, 10
    limc = T.Limits(lim.max_messages, lim.invariant_check_interval, 128, 0, 0, 0, 0, 0, k)
    got = oracle.random_explore(model, events, k * n_inst, seed_base=SEED_BASE, limits=limc)
    stopped_early = 0
    for j in range(n_inst):
        strategy = None
        for e in range(k):
            s = ScalaRandomScheduler(oracle, model, events, SEED_BASE + j, lim.max_messages, lim.invariant_check_interval, strategy=strategy)
            s.execute()
            v = got[j * k + e]
            assert (int(v["flags"]), int(v["fingerprint"]), int(v["hash"])) == s.verdict(), (j, e)
            # reset_all_state: pendingEvents.clear() (the array empties, the generator stays)
            strategy = s.pendingEvents
            del strategy.pendingEvents.arr[:]
            if s.violationFound:
                rest = got[j * k + e + 1:(j + 1) * k]
                assert not rest["flags"].any() and not rest["hash"].any(), "executions behind the first violation are not run"
                stopped_early += e + 1 < k
                break
    assert stopped_early >= 1, "the sample holds an instance that stops at a violation"
    # the chain differs from independent executions: execution 1 of instance 0 is NOT the execution seeded SEED_BASE + 1
    ind = oracle.random_explore(model, events, 2, seed_base=SEED_BASE, limits=T.Limits(lim.max_messages, lim.invariant_check_interval, 128, 0, 0, 0))
    assert int(got[1]["hash"]) != int(ind[1]["hash"]) and int(got[0]["hash"]) == int(ind[0]["hash"])
    # the recorded trace of a chained execution: same verdict as the batch, deliveries counted in its flags
    v, rec, ran = oracle.random_execute_carried(model, events, SEED_BASE, 3, limc)
    first_stop = next((e for e in range(4) if got[e]["flags"] & T.V_VIOLATION), 3)
    assert ran == first_stop and int(v.hash) == int(got[first_stop]["hash"])
    assert int((rec["kind"] == T.REC_MSG_EVENT).sum()) == T.verdict_deliveries(int(v.flags))


def test_a_table_with_arrays_equals_the_scala_transliteration(oracle):
    """DEMI_MODEL_ARRAY: whole executions of the replicated-log protocol (rows LDX / STX, an invariant program that reads the
    log) - the scheduler is the transliteration above, the handlers the oracle's rows, the state an actor's field word and its
    array words; verdict and hash (every state word) against the oracle's own loop."""
    from demi_amd import model as M
    from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence
    for buggy in (True, False):
        model = M.replog_model(4, 6, buggy, False)
        ev = [start(a) for a in range(4)]
        for i in range(7):
            ev.append(send(0 if i % 3 else i % 4, M.RL_PUT, 30 + i, 0))
            if i == 3:
                ev.append(wait_quiescence())
        events = events_to_array(ev)
        checked, violations = _compare(oracle, model, events, list(range(100, 140)), 300, 5)
        assert checked == 40 and (violations > 5) == buggy


def test_tables_of_more_than_eight_actors_equal_the_scala_transliteration(oracle):
    """The BIG layout (include/demi_gpu.h: 9 .. 16 actors, 4-bit receiver / 5-bit sender fields, deadLetters 31, 16-bit actor
    masks in the fingerprints): whole executions of the 11-node raft table (fuzz trace with kills and partitions) and of the
    12-actor shuffle job against the transliteration - delivery by delivery, verdict and hash."""
    from demi_amd.apps import SEED_BASE, raft11_config2, shuffle12_config5
    model, events, lim = raft11_config2()
    checked, violations = _compare(oracle, model, events, [SEED_BASE + i for i in range(60)], lim.max_messages, lim.invariant_check_interval)
    assert checked == 60
    model, _dev, events, lim, _par = shuffle12_config5()
    checked, violations = _compare(oracle, model, events, [SEED_BASE + i for i in range(40)], lim.max_messages, lim.invariant_check_interval)
    assert checked == 40 and violations >= 3
    w = FuzzerWeights(kill=0.12, send=0.35, wait_quiescence=0.13, partition=0.25, unpartition=0.15)
    model = M.raft_model(9, election_budget=[1, 1, 1, 1, 0, 0, 0, 0, 0])
    total = 0
    for tseed in (1, 2, 3):
        events = events_to_array(raft_trace(9, 60, tseed, w, exact=False))
        c, _ = _compare(oracle, model, events, [1000 * tseed + i * 7919 for i in range(10)], 600, 10)
        total += c
    assert total >= 25


def test_the_whole_bench_step_by_the_transliteration_is_the_oracles(oracle):
    """tests/golden/fuzz_config2_transliteration.json: ALL 2^20 schedules of the bench's fixed-seed step (config 2: raft5, the
    frozen trace, seeds SEED_BASE + i) as ScalaRandomScheduler above executed them (tools/check_fuzz_transliteration.py, eight
    processes, minutes).  The C oracle gives the same bytes - here for every recorded prefix - and the GPU suite holds the device's
    2^20 verdicts against the same record (test_full_size_properties_1m)."""
    import hashlib
    import json
    import os
    from demi_amd.apps import SEED_BASE, raft5_config2
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_config2_transliteration.json")) as f:
        rec = json.load(f)
    assert "ScalaRandomScheduler" in rec["generator"] and rec["equals_the_oracle"] is True and rec["seed_base"] == SEED_BASE
    assert rec["schedules"] == 1 << 20 and set(rec["sha256_verdicts_of_the_first"]) == {"16384", "131072", "1048576"}
    model, events, lim = raft5_config2()
    v = oracle.random_explore(model, events, rec["schedules"], seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count() or 1)
    for p, sha in rec["sha256_verdicts_of_the_first"].items():
        assert hashlib.sha256(np.ascontiguousarray(v[:int(p)]).tobytes()).hexdigest() == sha, p
    assert int(((v["flags"] & T.V_VIOLATION) != 0).sum()) == rec["violating_executions"]
