"""CPU suite: a whole DPORwHeuristics exploration transliterated from the Scala - the scheduling half as well as dpor() -
against the product's exploration (native bookkeeping around the oracle's interleavings, one backtrack point at a time).

  DPORwHeuristics.start_trace / runExternal / maybeAddGraphNode / getMessage / event_produced / schedule_new_message
    (getNextTraceMessage, getMatchingMessage, getPendingEvent, the isolated-actor discard, awaitQuiescenceUpdate) /
    notify_quiescence / notify_timer_cancel / dpor / getNext
                       (schedulers/DPORwHeuristics.scala:101-106, 256-310, 336-372, 421-648, 684-721, 773-847, 855-942, 961-984, 994-1185)
  ExploredTacker       (schedulers/AuxilaryTypes.scala:209-246)
  timers under DPOR:   Scheduler.enqueue_timer = enqueue_message = `!` at once (schedulers/Scheduler.scala:73-74,
                       DPORwHeuristics.scala:946-956) with the Instrumenter's registerCancellable / handleTick / retrigger
This file does not use the oracle's interleavings at all: Unique ids come from a counter, the dependency graph is a map
child -> parent with the children of every node, `getMessage` looks a produced message up among the parent's children.
Only the actors' `receive` is shared (the row interpreter).  Pinned as in the product: pendingEvents is iterated in
ascending (snd, rcv) order with the scheduler's own queue last (the reference iterates a Scala HashMap), PriorityQueue ties
pop in creation order, the repeating-timer retrigger runs before the receive."""
import ctypes as C
import heapq

import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence

from .test_dpor_cpu import PAR, native_explore, writers_model
from .test_random_scheduler_transliteration_cpu import MASK64, _Effect

DEAD, SCHEDULER = 15, 99          # "deadLetters", "__SCHEDULER__" as sender / receiver names


class Unique:
    __slots__ = ("event", "id")

    def __init__(self, event, uid):
        self.event, self.id = event, uid      # event: ("msg", snd, rcv, m) | ("wq",) | ("root",)


class ScalaDPORwHeuristics:
    def __init__(self, oracle, model, externals, depth_bound=0, max_messages=0, trackHistory=True, prioritizePendingUponDivergence=False,
                 lean=False):
        self.prioritizePendingUponDivergence = prioritizePendingUponDivergence
        # lean (tools/check_golden_dpor_transliteration.py --bug: hundreds of thousands of interleavings, ~2 000 racing pairs each):
        # the backtrack queue holds ONE entry per flipped pair instead of one per report - see dpor() for why that is the same
        # exploration; test_lean_queue_is_the_literal_queue holds the two against each other
        self.lean, self.best, self.exploredFlat = lean and trackHistory, {}, set()
        self.oracle, self.model, self.ms = oracle, model, model.to_struct()
        self.DEAD = T.DEADLETTERS_BIG if model.n_actors > T.MAX_ACTORS else DEAD
        self.should_bound, self.stop_at_depth = bool(depth_bound), depth_bound
        self.should_cap_messages, self.max_messages = bool(max_messages), max_messages
        self.trackHistory = trackHistory
        self.next_id = 1
        self._root = Unique(("root",), 0)
        self.parentOf, self.childrenOf, self.quiescentPeriod = {}, {self._root: []}, {}
        self.backTrack, self.seq, self.exploredStack = [], 0, {}
        self.nextTrace, self.origNextTraceSize = [], 0
        # the external events; a WaitQuiescence becomes Unique(w, id = w._id)
        self.externalEventList = []
        for i, e in enumerate(externals):
            kind = int(e["kind"])
            if kind == T.EV_WAIT_QUIESCENCE:
                self.externalEventList.append(Unique(("wq",), 100000 + i))
            else:
                self.externalEventList.append((kind, int(e["a"]), (int(e["msg_type"]), int(e["p0"]), int(e["p1"]))))
        self.verdicts, self.next_trace_lens = [], []

    # ------------------------------------------------------------------ depGraph
    def addGraphNode(self, u):
        self.childrenOf.setdefault(u, [])
        self.quiescentPeriod[u] = self.currentQuiescentPeriod

    def addEdge(self, child, parent):
        if self.parentOf.get(child) is not parent:
            self.parentOf[child] = parent
            self.childrenOf[parent].append(child)

    def pathToRoot(self, u):
        path = [u]
        while u in self.parentOf:
            u = self.parentOf[u]
            path.append(u)
        return path

    def setParentEvent(self, u):
        self.parentEvent = u
        self.currentDepth = (len(self.pathToRoot(u)) - 1) + 1            # getPathLength(event) + 1

    def maybeAddGraphNode(self, u):
        for x in self.childrenOf[self.currentRoot]:
            if x.event[0] == "wq" and x.id == u.id:
                return x
        self.addGraphNode(u)
        self.addEdge(u, self.currentRoot)
        return u

    def getMessage(self, snd, rcv, msg):
        for x in self.childrenOf[self.parentEvent]:
            if x.event[0] == "msg" and x.event[1:] == (snd, rcv, msg):
                return x
        u = Unique(("msg", snd, rcv, msg), self.next_id)
        self.next_id += 1
        return u

    # ------------------------------------------------------------------ one interleaving
    def start_trace(self):
        A = self.model.n_actors
        self.pendingEvents = {}
        self.currentTrace = []
        self.currentQuiescentPeriod = 0
        self.awaitingQuiescence, self.nextQuiescentPeriod, self.quiescentMarker = False, 0, None
        self.externalEventIdx = 0
        self.messagesScheduledSoFar = 0
        self.currentRoot = self._root
        self.addGraphNode(self._root)
        self.setParentEvent(self._root)
        self.currentTrace.append(self._root)
        self.isolatedActors = set(range(A))                       # maybeStartActors: isolatedActors ++= actorNames
        self.blockedActors = set()
        self.timerToCancellable, self.ongoing, self.registered, self.next_c = {}, set(), set(), 0
        self.seededRandom = C.c_uint64((0 ^ 0x5DEECE66D) & ((1 << 48) - 1))
        # (an actor's state: its field word, then - DEMI_MODEL_ARRAY - the words of its array, empty at the start)
        # (a wide table has two field words per actor; a table with more than 8 actors - the BIG layout of include/demi_gpu.h, a
        # wide table - names deadLetters 31: it still sorts behind every actor in the pinned queue order)
        self.stw = getattr(self.model, "state_words", 1)
        assert getattr(self.model, "payloads", 2) == 2       # (messages are (type, p0, p1) here: tables of two payload fields)
        self.wide = bool(getattr(self.model, "wide", False))
        self.big = A > T.MAX_ACTORS
        fw = 2 if self.wide else 1
        self.state = [[int(w) for w in self.model.init_state[a * fw:(a + 1) * fw]] + [0] * (self.stw - fw) for a in range(A)]
        self.deliveries = []
        self.aborted = False
        self.runExternal()

    def runExternal(self):
        wait = False
        while self.externalEventIdx < len(self.externalEventList) and not wait:
            event = self.externalEventList[self.externalEventIdx]
            if isinstance(event, Unique):
                self.pendingEvents.setdefault((SCHEDULER, SCHEDULER), []).append(event)
                self.maybeAddGraphNode(event)
                wait = True
            elif event[0] == T.EV_START:
                self.isolatedActors.discard(event[1])
            elif event[0] == T.EV_SEND:
                self.event_produced(self.DEAD, event[1], event[2])       # actorMappings(rcv) ! msgCtor()
            else:
                raise Exception("unsuported external event")
            self.externalEventIdx += 1

    def event_produced(self, snd, rcv, msg):
        unique = self.getMessage(snd, rcv, msg)
        if not self.should_bound or self.currentDepth < self.stop_at_depth:
            self.pendingEvents.setdefault((snd, rcv), []).append(unique)
        self.addGraphNode(unique)
        self.addEdge(unique, self.parentEvent)

    def _queues_in_pinned_order(self):
        return sorted(self.pendingEvents.items(), key=lambda kv: kv[0])      # (snd, rcv) ascending; SCHEDULER = 99 is last

    def getPendingEvent(self):
        for k, v in self._queues_in_pinned_order():
            if k[1] not in self.blockedActors and v:
                return v.pop(0)
        return None

    def getNextTraceMessage(self):
        while self.nextTrace:
            u = self.nextTrace.pop(0)
            if u.id == 0:                         # "All system messages need to ignored" - the root has id 0
                continue
            return u
        return None

    def getMatchingMessage(self):
        u = self.getNextTraceMessage()
        if u is None:
            return None
        if u.event[0] == "msg":
            _, snd, rcv, _m = u.event
            if rcv in self.blockedActors:
                return None
            q = self.pendingEvents.get((snd, rcv))
        else:
            q = self.pendingEvents.get((SCHEDULER, SCHEDULER))
        if q is None:
            return None
        for i, other in enumerate(q):             # dequeueFirst(equivalentTo(u, _)): same receiver and same id
            if other.id == u.id:
                return q.pop(i)
        return None

    def schedule_new_message(self):
        while True:
            self.messagesScheduledSoFar += 1
            if self.should_cap_messages and self.messagesScheduledSoFar > self.max_messages:
                return None
            if not self.awaitingQuiescence:
                if self.prioritizePendingUponDivergence:           # getNextMatchingMessage (:537-550): pop heads until one matches
                    result = None
                    while self.nextTrace and result is None:
                        result = self.getMatchingMessage()
                else:
                    result = self.getMatchingMessage()
                if result is None:
                    result = self.getPendingEvent()
            else:
                result = self.getPendingEvent()
            if result is None:
                return None
            if result.event[0] == "msg":
                _, snd, rcv, _m = result.event
                if snd in self.isolatedActors or rcv in self.isolatedActors:
                    if snd == rcv:
                        raise RuntimeError("self message without prior messages!")
                    continue                       # return schedule_new_message(blockedActors)
                self.currentTrace.append(result)
                self.setParentEvent(result)
                return result
            # awaitQuiescenceUpdate
            self.awaitingQuiescence = True
            self.nextQuiescentPeriod = result.id
            self.quiescentMarker = result

    # ------------------------------------------------------------------ Instrumenter (timers are `!` at once under DPOR)
    def registerCancellable(self, ongoingTimer, receiver, msg):
        if (receiver, msg) in self.timerToCancellable:
            return
        c = self.next_c
        self.next_c += 1
        self.registered.add(c)
        if ongoingTimer:
            self.ongoing.add(c)
        self.timerToCancellable[(receiver, msg)] = c
        self.handleTick(receiver, msg, c)

    def handleTick(self, receiver, msg, c):
        self.event_produced(self.DEAD, receiver, msg)                   # scheduler.enqueue_timer -> enqueue_message -> `!`
        if c not in self.ongoing:
            self.registered.discard(c)
            del self.timerToCancellable[(receiver, msg)]

    def cancelTimer(self, rcv, msg):
        c = self.timerToCancellable.pop((rcv, msg), None)
        if c is not None:
            self.ongoing.discard(c)
            self.registered.discard(c)
        q = self.pendingEvents.get((self.DEAD, rcv))                    # notify_timer_cancel (:961-984)
        if q is not None:
            for i, u in enumerate(q):
                if u.event[3] == msg:
                    q.pop(i)
                    break

    def dispatch(self, u):
        _, snd, rcv, msg = u.event
        mtype, p0, p1 = msg
        if self.wide:
            self.deliveries.append(mtype | (rcv << 5) | (snd << (9 if self.big else 8)) | (p0 << 16) | (p1 << 32))
        else:
            self.deliveries.append(mtype | (rcv << 5) | (snd << 8) | (p0 << 16) | (p1 << 24))
        c = self.timerToCancellable.get((rcv, msg))
        if c is not None and c in self.ongoing:                    # "Check if it was a repeating timer. If so, retrigger it"
            self.handleTick(rcv, msg, c)
        st = (C.c_uint64 * self.stw)(*self.state[rcv])
        fx = (_Effect * 64)()
        n = self.oracle.lib().orc_vm_run(C.byref(self.ms), rcv, st, mtype, snd, p0, p1, (1 << self.model.n_actors) - 1,
                                         fx, 64, C.byref(self.seededRandom))
        assert n >= 0
        self.state[rcv] = [int(w) for w in st]
        for e in fx[:n]:
            if e.kind == 0:
                self.event_produced(rcv, int(e.target), (int(e.msg_type), int(e.p0), int(e.p1)))
            elif e.kind in (1, 2):
                self.registerCancellable(e.kind == 2, rcv, (int(e.msg_type), 0, 0))
            elif e.kind == 3:
                self.cancelTimer(rcv, (int(e.msg_type), 0, 0))
            elif e.kind == 4:
                self.blockedActors.add(rcv)

    # ------------------------------------------------------------------ ExploredTacker, dpor(), getNext()
    def setExplored(self, index, pair):
        if self.lean:
            self.exploredFlat.add(pair)      # (the union of the index sets is all isExplored ever asks for)
        else:
            self.exploredStack.setdefault(index, set()).add(pair)

    def isExplored(self, pair):
        if self.lean:
            return pair in self.exploredFlat
        return any(pair in s for s in self.exploredStack.values())

    def dpor(self, trace):
        def isCoEnabeled(earlier, later):
            if earlier.event[0] != "msg" or later.event[0] != "msg":     # WaitQuiescence never; the root's receiver is "null"
                return False
            if earlier.event[2] != later.event[2]:
                return False
            if self.quiescentPeriod[earlier] != self.quiescentPeriod[later]:
                return False
            return earlier not in self.pathToRoot(later)

        for laterI in range(len(trace)):
            later = trace[laterI]
            for earlierI in range(laterI):
                earlier = trace[earlierI]
                if not isCoEnabeled(earlier, later):
                    continue
                laterPath = list(reversed(self.pathToRoot(later)))
                earlierPath = set(self.pathToRoot(earlier))
                commonPrefix = [x for x in laterPath if x in earlierPath]
                branchI = trace.index(commonPrefix[-1])
                needToReplay = [x for x in trace[branchI + 1:laterI + 1] if x.id != earlier.id]
                assert branchI < laterI
                if self.trackHistory:
                    self.setExplored(branchI, (earlier, later))
                if self.lean:
                    # Memory only, not behaviour.  (1) exploredStack never shrinks (trimExplored is commented out in the Scala),
                    # so a point whose pair is explored NOW would be skipped whenever it is popped: not stored.  (2) of several
                    # stored points for one pair the queue pops the one with the largest branchI (ties: the oldest) first and
                    # marks the pair, so the others would be skipped: only that one is kept (the superseded heap entry stays
                    # behind and is recognised by its sequence number).  The replay list is built at the pop from the trace.
                    pair = (later, earlier)
                    if not self.isExplored(pair):
                        cur = self.best.get(pair)
                        if cur is None or branchI > cur[0]:
                            self.best[pair] = (branchI, self.seq)
                            heapq.heappush(self.backTrack, (-branchI, self.seq, pair, (trace, branchI, laterI, earlier.id)))
                    self.seq += 1
                    continue
                heapq.heappush(self.backTrack, (-branchI, self.seq, (later, earlier), needToReplay))
                self.seq += 1
        while True:                                                    # getNext
            if not self.backTrack:
                return None
            negI, _s, (e1, e2), replayThis = heapq.heappop(self.backTrack)
            if self.lean:
                if self.best.get((e1, e2), (None, None))[1] != _s:
                    continue                                           # superseded by a deeper point for the same pair
                del self.best[(e1, e2)]
                tr, b, li, eid = replayThis
                replayThis = [x for x in tr[b + 1:li + 1] if x.id != eid]
            if self.trackHistory and self.isExplored((e1, e2)):
                continue
            maxIndex = -negI
            if self.trackHistory:
                self.setExplored(maxIndex, (e1, e2))
            return trace[:maxIndex + 1] + replayThis

    # ------------------------------------------------------------------ run(): the exploration
    def run(self, max_interleavings):
        while len(self.verdicts) < max_interleavings:
            self.next_trace_lens.append(len(self.nextTrace))
            self.start_trace()
            while True:
                nxt = self.schedule_new_message()
                if nxt is not None:
                    self.dispatch(nxt)
                    continue
                # notify_quiescence
                if self.awaitingQuiescence:
                    self.awaitingQuiescence = False
                    self.currentQuiescentPeriod = self.nextQuiescentPeriod
                    self.nextQuiescentPeriod = 0
                    marker = self.maybeAddGraphNode(self.quiescentMarker)
                    self.currentTrace.append(marker)
                    self.currentRoot = marker
                    self.setParentEvent(marker)
                    self.quiescentMarker = None
                    self.runExternal()
                    continue
                break
            # checkInvariant + the verdict of this interleaving
            states = (C.c_uint64 * (self.model.n_actors * self.stw))(*[w for st_ in self.state for w in st_])
            fp = int(self.oracle.lib().orc_invariant(C.byref(self.ms), states, (1 << self.model.n_actors) - 1))
            h = 0xCBF29CE484222325
            for w in self.deliveries:
                h = ((h ^ w) * 0x100000001B3) & MASK64
            for a in range(self.model.n_actors):
                for w in self.state[a]:
                    h = ((h ^ w) * 0x100000001B3) & MASK64
            capped = self.should_cap_messages and self.messagesScheduledSoFar > self.max_messages
            flags = (T.V_VIOLATION if fp else 0) | (T.V_MAXMSG if capped else 0) | min(len(self.deliveries), 0xFFFF) << 16
            self.verdicts.append((flags, fp, h))
            nxt = self.dpor(list(self.currentTrace))
            if nxt is None:
                return True
            self.nextTrace = nxt
        return False


def _config3(cap):
    from demi_amd.apps import raft5_config3
    model, ev, depth = raft5_config3()
    return model, ev, depth, 0, cap


def _config5(cap):
    from demi_amd.apps import shuffle8_config5_large
    model, ev, depth, _budget = shuffle8_config5_large()
    return model, ev, depth, 0, cap


def _config3_bug(cap):
    from demi_amd.apps import raft5_dpor_config3
    model, ev, par = raft5_dpor_config3()
    return model, ev, int(par.depth_bound), 0, cap, True


def _config5_bug(cap):
    from demi_amd.apps import shuffle8_dpor_config5
    model, ev, par, _budget = shuffle8_dpor_config5()
    return model, ev, int(par.depth_bound), 0, cap, True


def _big(which, cap):
    from demi_amd.apps import raft11_dpor, shuffle12_config5
    if which == "raft11":
        model, ev, par = raft11_dpor()
    else:
        model, ev, _fev, _lim, par = shuffle12_config5()
    return model, ev, int(par.depth_bound), 0, cap, True, int(par.p_max)


CASES = {
    # round 6: the DPOR workloads bench.py times (prioritizePendingUponDivergence; they find the seeded bugs), their first
    # interleavings (the whole of config 3 - hundreds of thousands - is tools/check_golden_dpor_transliteration.py --bug)
    "raft5_dpor_config3_first_250": lambda: _config3_bug(250),
    "shuffle8_dpor_config5_first_250": lambda: _config5_bug(250),
    # prioritizePendingUponDivergence where a flip decides the verdict: two campaigning nodes of three, exhausted
    "raft3_two_campaigners_prioritize": lambda: (M.raft_model(3, election_budget=(1, 1, 0)), events_to_array([start(a) for a in range(3)] +
                                                 [send(a, M.M_BOOTSTRAP) for a in range(3)]), 12, 0, 20000, True),   # (1 552 interleavings, 28 violating)
    # more than 8 actors (the BIG layout of include/demi_gpu.h): the 12-actor shuffle job's first interleavings - its exploration
    # exhausts after 33 529, 1 836 of them violating - and the 11-node raft cluster
    "shuffle12_first_150": lambda: _big("shuffle12", 150),
    "raft11_first_120": lambda: _big("raft11", 120),
    "writers4": lambda: (writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), 0, 0, 2500),
    "raft3": lambda: (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)]), 30, 0, 300),
    "raft3_two_periods": lambda: (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(0, M.M_BOOTSTRAP),
                                  wait_quiescence(), send(1, M.M_BOOTSTRAP), send(2, M.M_BOOTSTRAP)]), 24, 0, 300),
    # DEMI_MODEL_ARRAY: the replicated log (rows LDX / STX, an invariant program that reads the log)
    "replog3_arrays": lambda: (M.replog_model(3, 4, True, False), events_to_array([start(a) for a in range(3)] + [send(0, M.RL_PUT, 9, 0), send(1, M.RL_PUT, 8, 0), send(0, M.RL_PUT, 7, 0)]), 40, 0, 400),
    # BASELINE config 3, its first interleavings (the whole exploration - 60 332, most of an hour in this Python - is what
    # tools/check_golden_dpor_transliteration.py runs; its record: tests/golden/dpor_config3_transliteration.json)
    "raft5_config3_first_250": lambda: _config3(250),
    # config 5's three-job shuffle pipeline (8 actors, 3 classes, depth 40), its first interleavings (the 6 000 that the GPU's
    # reference-order test compares: tools/check_golden_dpor_transliteration.py --config5 6000)
    "shuffle8_pipeline_first_120": lambda: _config5(120),
    "raft3_late_start_and_cap": lambda: (M.raft_model(3), events_to_array([start(0), start(1), send(0, M.M_BOOTSTRAP), send(1, M.M_BOOTSTRAP),
                                         wait_quiescence(), start(2), send(2, M.M_BOOTSTRAP)]), 20, 40, 300),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_whole_exploration_equals_the_scala_transliteration(oracle, case):
    model, ev, depth, maxm, cap, *pp = CASES[case]()
    p_max = pp[1] if len(pp) > 1 else 0          # (the restatement's pending capacity; the reference has none)
    pp = bool(pp and pp[0])
    sc = ScalaDPORwHeuristics(oracle, model, ev, depth_bound=depth, max_messages=maxm, prioritizePendingUponDivergence=pp)
    exhausted = sc.run(cap)
    par = PAR(depth=depth, maxm=maxm)
    par.prioritize_pending = 1 if pp else 0
    if p_max:
        par.p_max = p_max
    nat = native_explore(model, ev, par, 1, cap)
    assert len(nat[0]) == len(sc.verdicts) and bool(nat[4].exhausted) == exhausted
    want = np.array(sc.verdicts, dtype=[("flags", "<u4"), ("fingerprint", "<u4"), ("hash", "<u8")])
    for f in ("flags", "fingerprint", "hash"):
        assert (nat[0][f] == want[f]).all(), f
    assert [int(x) for x in nat[1]] == sc.next_trace_lens
    if case == "writers4":
        assert exhausted and 0 < int((want["flags"] & T.V_VIOLATION != 0).sum()) < len(want)
    if case == "raft3_late_start_and_cap":
        assert int((want["flags"] & T.V_MAXMSG != 0).sum()) > 0
    if case in ("raft3_two_campaigners_prioritize", "shuffle8_dpor_config5_first_250"):
        assert 0 < int((want["flags"] & T.V_VIOLATION != 0).sum()) < len(want)
    if case == "raft3_two_campaigners_prioritize":
        assert exhausted


@pytest.mark.parametrize("case", ["raft3_two_campaigners_prioritize", "raft3", "raft5_dpor_config3_first_250", "raft3_two_periods"])
def test_lean_queue_is_the_literal_queue(oracle, case):
    """ScalaDPORwHeuristics(lean=True) - one queue entry per flipped pair, what the hours-long run of the tool uses - explores the
    interleavings of the literal queue, in its order."""
    model, ev, depth, maxm, cap, *pp = CASES[case]()
    pp = bool(pp and pp[0])
    a = ScalaDPORwHeuristics(oracle, model, ev, depth_bound=depth, max_messages=maxm, prioritizePendingUponDivergence=pp)
    b = ScalaDPORwHeuristics(oracle, model, ev, depth_bound=depth, max_messages=maxm, prioritizePendingUponDivergence=pp, lean=True)
    assert a.run(cap) == b.run(cap)
    assert a.verdicts == b.verdicts and a.next_trace_lens == b.next_trace_lens and len(a.verdicts) > 100


def test_config3_golden_record_is_the_transliterations_too():
    """tests/golden/dpor_config3_reference_order.json - the record the GPU's REFERENCE order is held against, made by the C
    oracle under the product's one-at-a-time loop - was reproduced by ScalaDPORwHeuristics above, which shares neither: all
    60 332 interleavings, 73 minutes on one core (tools/check_golden_dpor_transliteration.py wrote its record beside it).
    Here: the two committed records say the same."""
    import json
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(g, "dpor_config3_reference_order.json")) as f:
        gold = json.load(f)
    with open(os.path.join(g, "dpor_config3_transliteration.json")) as f:
        tr = json.load(f)
    assert "ScalaDPORwHeuristics" in tr["generator"] and tr["equals_dpor_config3_reference_order_json"] is True
    for k in ("interleavings", "exhausted", "sha256_verdicts", "sha256_prefix_lens", "violations", "distinct_schedules"):
        assert tr[k] == gold[k], k
    assert gold["interleavings"] == 60332 and gold["exhausted"]


def test_config5_pipeline_record_of_the_transliteration_is_the_oracles(oracle):
    """tests/golden/dpor_config5_transliteration.json: the first 6 000 interleavings of config 5's three-job pipeline in the
    reference's order as ScalaDPORwHeuristics explored them (tools/check_golden_dpor_transliteration.py --config5 6000, nine
    minutes).  The C oracle under the product's one-at-a-time loop gives the same bytes - and the GPU suite holds the device's
    REFERENCE order against both (test_config5_pipeline_in_reference_order_is_the_one_at_a_time_sequence)."""
    import hashlib
    import json
    import os
    from demi_amd.apps import shuffle8_config5_large
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dpor_config5_transliteration.json")) as f:
        rec = json.load(f)
    assert "ScalaDPORwHeuristics" in rec["generator"] and rec["equals_the_oracles_one_at_a_time_exploration"] is True
    model, ev, depth, _budget = shuffle8_config5_large()
    n = rec["interleavings"]
    one = oracle.dpor_explore(model, ev, PAR(depth=depth), T.DporSearch(1, n, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    assert len(one[0]) == n == 6000
    assert hashlib.sha256(np.ascontiguousarray(one[0], dtype=T.VERDICT_DTYPE).tobytes()).hexdigest() == rec["sha256_verdicts"]
    assert hashlib.sha256(np.ascontiguousarray(one[1], dtype=np.uint32).tobytes()).hexdigest() == rec["sha256_prefix_lens"]


def test_round6_records_of_the_workloads_that_find_the_bugs(oracle):
    """The records the GPU's REFERENCE order is held against on the workloads bench.py times from round 6 on
    (tests/test_dpor_bug_workloads_gpu.py): config 3's golden record (the C oracle, one backtrack point at a time) and the
    transliteration's record of the SAME 258 025 interleavings say the same, and the oracle reproduces the golden record's first
    2^14 verdicts here; config 5's first 6 000 (1 118 violating) by the transliteration are the oracle's bytes."""
    import hashlib
    import json
    import os
    from demi_amd.apps import raft5_dpor_config3, shuffle8_dpor_config5
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    with open(os.path.join(g, "dpor_config3_bug_reference_order.json")) as f:
        gold = json.load(f)
    with open(os.path.join(g, "dpor_config3_bug_transliteration.json")) as f:
        tr = json.load(f)
    assert "ScalaDPORwHeuristics" in tr["generator"] and tr["equals_dpor_config3_bug_reference_order_json"] is True
    for k in ("interleavings", "exhausted", "sha256_verdicts", "sha256_prefix_lens", "violations", "distinct_schedules"):
        assert tr[k] == gold[k], k
    assert gold["interleavings"] == 258025 and gold["exhausted"] and gold["violations"] == 4028 and gold["first_violation"] == 47012
    model, ev, par = raft5_dpor_config3()
    one = oracle.dpor_explore(model, ev, par, T.DporSearch(1, 1 << 14, 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    assert hashlib.sha256(np.ascontiguousarray(one[0], dtype=T.VERDICT_DTYPE).tobytes()).hexdigest() == gold["sha256_first_16384_verdicts"]
    with open(os.path.join(g, "dpor_config5_bug_transliteration.json")) as f:
        rec = json.load(f)
    assert "ScalaDPORwHeuristics" in rec["generator"] and rec["equals_the_oracles_one_at_a_time_exploration"] is True and rec["violations"] == 1118
    model, ev, par, _budget = shuffle8_dpor_config5()
    one = oracle.dpor_explore(model, ev, par, T.DporSearch(1, rec["interleavings"], 0, 1, T.DPOR_ORDER_ROUNDS), n_threads=1)
    assert len(one[0]) == rec["interleavings"] == 6000
    assert hashlib.sha256(np.ascontiguousarray(one[0], dtype=T.VERDICT_DTYPE).tobytes()).hexdigest() == rec["sha256_verdicts"]
    assert hashlib.sha256(np.ascontiguousarray(one[1], dtype=np.uint32).tobytes()).hexdigest() == rec["sha256_prefix_lens"]
