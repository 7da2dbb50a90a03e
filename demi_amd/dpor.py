"""Host-side control loop of DPORwHeuristics over the K3 kernel.

Mirrors schedulers/DPORwHeuristics.scala: the backtrack priority queue with
DefaultBacktrackOrdering (deepest branch first, BacktrackOrdering.scala:58-69), the ExploredTacker
(AuxilaryTypes.scala:209-246), dpor()'s bookkeeping around the pair loop (setExplored :1068-1070,
enqueue :1134) and getNext() (:1142-1162, next trace :1180).  What runs on the GPU is everything
inside one interleaving and the racing-pair analysis of its trace (demi_dpor_batch).

The reference explores one interleaving at a time.  Here a *round* pops up to `batch` backtrack
points (skipping explored pairs exactly as getNext does), runs them as one launch (one lane each,
dealt round-robin to the ranks of a process group when there is one), and then absorbs the results
in pop order.  batch=1 is the reference's order with its unspecified PriorityQueue tie order pinned
to creation order.  Because a backtrack point may be absorbed after newer interleavings, each point
stores its full next trace (`trace.take(branch+1) ++ needToReplay` of the interleaving that found
it) instead of re-deriving the prefix from "the current trace" (:1180), which is only correct in
strict depth-first order (see the TODO at :1173-1176).
"""
import heapq
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Set, Tuple

import numpy as np

from . import types as T
from .schedulers import MinimizationStats, SchedulerConfig, ViolationFingerprint


# verdict flags of an interleaving whose racing pairs are missing or truncated
K3_INCOMPLETE = T.V_PENDING_OVF | T.V_QUEUE_OVF | T.V_TRACE_OVF | T.V_SELFMSG | T.V_PAIRS_OVF


@dataclass
class Interleaving:
    verdict: np.void            # VERDICT_DTYPE row
    trace: np.ndarray           # DPOR_TRACE_DTYPE
    prefix_len: int


@dataclass
class Exploration:
    interleavings: List[Interleaving] = field(default_factory=list)
    violations: List[int] = field(default_factory=list)      # indices into interleavings
    rounds: List[int] = field(default_factory=list)          # batch size of every launch
    exhausted: bool = False                                  # the backtrack queue ran empty and no interleaving was cut short
    aborted: int = 0                                         # interleavings without a valid result (a capacity of the engine:
                                                             # pending set, trace length, racing-pair list; SELFMSG): their
                                                             # backtrack points are missing, the exploration is incomplete

    def schedule_hashes(self) -> Set[int]:
        return {int(i.verdict["hash"]) for i in self.interleavings}


class ExploredTacker:
    """AuxilaryTypes.scala:209-246 (isExplored looks at every index's set)."""

    def __init__(self):
        self.exploredStack: Dict[int, Set[Tuple[int, int]]] = {}
        self._any: Set[Tuple[int, int]] = set()

    def setExplored(self, index: int, pair: Tuple[int, int]):
        self.exploredStack.setdefault(index, set()).add(pair)
        self._any.add(pair)

    def isExplored(self, pair: Tuple[int, int]) -> bool:
        return pair in self._any


class DefaultBacktrackOrdering:
    """BacktrackOrdering.scala:58-69: deeper branch first; no notion of distance."""

    def init(self, sched, originalTrace):
        pass

    def priority(self, trace: np.ndarray, branch: int, later: int, earlier: int) -> Tuple[int, ...]:
        return (branch,)                       # compared as a tuple, larger = dequeued first

    def getDistance(self, priority: Tuple[int, ...]) -> int:
        return 0


class StopImmediatelyOrdering(DefaultBacktrackOrdering):
    """BacktrackOrdering.scala:71-81: combine with setMaxDistance(0)."""

    def getDistance(self, priority):
        return 0x7FFFFFFF


class ArvindDistanceOrdering:
    """BacktrackOrdering.scala:99-173.  Distance of a backtrack point from the original execution: events of its
    path that the original did not contain, plus misordered pairs among those it did.  The path is the one the
    reference builds (:117-123): the causal chain root..later, the events to replay, then (later, earlier).
    Higher distance compares greater, i.e. is dequeued FIRST, ties by depth (:155-165) — and getNext() stops as soon
    as the head's distance reaches the cap (DPORwHeuristics.scala:1145-1146)."""

    def __init__(self):
        self.originalIndices: Dict[int, int] = {}

    def init(self, sched, originalTrace: np.ndarray):
        self.originalTrace = np.ascontiguousarray(originalTrace, dtype=T.DPOR_TRACE_DTYPE)        # (explore_native hands it to the library)
        self.originalIndices = {int(k): i for i, k in enumerate(np.asarray(originalTrace)["key"].tolist())}

    def arvindDistance(self, trace: np.ndarray, branch: int, later: int, earlier: int) -> int:
        keys = trace["key"].tolist()
        parent = trace["parent"].tolist()
        chain = [later]                                     # getCommonPrefix(later, later): root .. later
        while chain[-1] != 0:
            chain.append(parent[chain[-1]])
        path = [keys[i] for i in reversed(chain)]
        path += [keys[i] for i in range(branch + 1, later + 1) if i != earlier]      # needToReplay (:1054-1057)
        path += [keys[later], keys[earlier]]
        idx = np.array([self.originalIndices.get(k, -1) for k in path], dtype=np.int64)
        present = idx >= 0
        distance = int((~present).sum())
        pi = idx[present]
        if len(pi) > 1:
            distance += int(np.triu(pi[:, None] > pi[None, :], 1).sum())    # pred before e with a larger original index
        return distance

    def priority(self, trace, branch, later, earlier):
        return (self.arvindDistance(trace, branch, later, earlier), branch)

    def getDistance(self, priority):
        return priority[0]


class DPORwHeuristics:
    """DPORwHeuristics(schedulerConfig, prioritizePendingUponDivergence=..., backtrackHeuristic=..., depth_bound=...,
    stopIfViolationFound=..., startFromBackTrackPoints=..., trackHistory=...) (DPORwHeuristics.scala:63-90)."""

    def __init__(self, schedulerConfig: SchedulerConfig, depth_bound: Optional[int] = None,
                 stopIfViolationFound: bool = True, trackHistory: bool = True, batch: int = 256,
                 max_pairs: int = 4096, p_max: int = 64, device: int = 0, backend: Optional[Callable] = None,
                 specialize: bool = False, prioritizePendingUponDivergence: bool = False, backtrackHeuristic=None,
                 startFromBackTrackPoints: bool = True, native: bool = False):
        """native: test() runs the whole exploration inside the library (demi_dpor_explore: queue, explored pairs, getNext and
        - for ArvindDistanceOrdering / a distance cap / an initial trace - the resumable state of this instance) instead of this
        Python loop; same explorations."""
        if schedulerConfig.model is None or schedulerConfig.model.inv_kind == T.INV_NONE:
            raise ValueError("Must invoke setInvariant before test()")
        self.schedulerConfig = schedulerConfig
        self.depth_bound = depth_bound
        self.stopIfViolationFound = stopIfViolationFound
        self.trackHistory = trackHistory
        self.batch = batch
        self.max_pairs = max_pairs
        self.p_max = p_max
        self.max_messages = 0
        self.native = native and backend is None
        self._backend = backend          # tests inject the CPU oracle here
        self.specialize = specialize     # compile the model's table to native code first (pays off on long explorations)
        self._device = device
        self._ctx = None
        self.prioritizePendingUponDivergence = prioritizePendingUponDivergence
        self.backtrackHeuristic = backtrackHeuristic or DefaultBacktrackOrdering()
        self.startFromBackTrackPoints = startFromBackTrackPoints
        self.should_cap_distance = False
        self.stop_at_distance = 0
        self.native_budget = 1 << 16     # interleavings per native test() call (the capacity of its output arrays)
        self._initialTrace: Optional[np.ndarray] = None
        self._started = False            # test()/explore() has run before on this instance (ResumableDPOR)
        # dropping a point whose flipped pair is already explored when it is CREATED is only the same exploration
        # if nothing looks at the queue's head before popping: not with a distance cap or a custom ordering
        self._early_drop = backtrackHeuristic is None
        self.backTrack: list = []        # heap of (negated priority..., seq, (later key, earlier key), trace, later, earlier)
        self._seq = 0
        self.exploredTracker = ExploredTacker()
        self.interleavingCounter = 0
        self._next_shared = 0
        self.shortestTraceSoFar: Optional[np.ndarray] = None

    def getName(self) -> str:
        return "DPORwHeuristics"

    def setMaxMessagesToSchedule(self, _max_messages: int):
        self.max_messages = _max_messages

    def setDepthBound(self, d: int):
        self.depth_bound = d

    def setMaxDistance(self, _stop_at_distance: int):
        """:131-134: getNext() gives up once the head of the queue is at least this far from the original."""
        self.should_cap_distance = True
        self.stop_at_distance = _stop_at_distance
        self._early_drop = False

    def setInitialTrace(self, t: np.ndarray):
        """:211-213: the first interleaving replays this trace (DPOR_TRACE_DTYPE: key, word, kind)."""
        self._initialTrace = np.ascontiguousarray(t, dtype=T.DPOR_TRACE_DTYPE)

    # -- one launch
    def _params(self, lookingFor: Optional[ViolationFingerprint]) -> T.DporParams:
        return T.DporParams(self.depth_bound or 0, self.max_messages, 1 if lookingFor is not None else 0,
                            lookingFor.code if lookingFor is not None else 0, self.p_max, self.max_pairs,
                            1 if self.prioritizePendingUponDivergence else 0)

    def _run(self, externals, prefixes, params, shared=None):
        from .distributed import sharded_batch
        if shared is None:
            shared = [0] * len(prefixes)
        items = list(zip(prefixes, shared))
        if self._backend is not None:
            fn = lambda part: self._backend(self.schedulerConfig.model, externals, [p for p, _ in part], params,
                                            [s for _, s in part])
        else:
            if self._ctx is None:
                from . import _native
                self._ctx = _native.Context(self._device)
                self._ctx.model_load(self.schedulerConfig.model.to_struct())
                if self.specialize or getattr(self.schedulerConfig.model, "compiled_only", False):     # (a wide table has no interpreter)
                    self._ctx.model_specialize()
                self._ctx.dpor_load(externals)
            fn = lambda part: self._ctx.dpor_batch([p for p, _ in part], params, [s for _, s in part])
        return sharded_batch(items, fn)

    # -- getNext (:1142-1162): pop the highest-priority unexplored backtrack point
    def _get_next(self):
        while self.backTrack:
            head = self.backTrack[0]
            if self.should_cap_distance and self.backtrackHeuristic.getDistance(tuple(-x for x in head[0])) >= self.stop_at_distance:
                return None                                  # "Tutto finito!" (:1144-1150); the queue is kept
            neg_prio, _, pair, trace, later, earlier = heapq.heappop(self.backTrack)
            if self.trackHistory and pair in self.exploredTracker._any:
                continue
            branch = -neg_prio[-1]
            if self.trackHistory:
                self.exploredTracker.setExplored(branch, pair)
            # next trace = trace.take(branch + 1) ++ needToReplay (:1054-1057, 1180), built on demand.  The racing pairs
            # inside the take() part were absorbed when `trace` ran: with the default ordering their duplicates can never
            # be dequeued live, so the kernel need not report them again (include/demi_gpu.h, demi_dpor_batch)
            idx = [i for i in range(branch + 1, later + 1) if i != earlier]
            self._next_shared = branch + 1 if (self.trackHistory and self._early_drop) else 0
            return np.concatenate([trace[:branch + 1], trace[idx]])
        return None

    # -- dpor()'s bookkeeping for one finished interleaving (:1122-1139)
    def _absorb(self, trace: np.ndarray, pairs: np.ndarray):
        self.interleavingCounter += 1
        if len(pairs) == 0:
            return
        keys = trace["key"].tolist()
        explored = self.exploredTracker._any
        stack = self.exploredTracker.exploredStack
        push = heapq.heappush
        for branch, later, earlier in zip(pairs["branch"].tolist(), pairs["later"].tolist(), pairs["earlier"].tolist()):
            ke, kl = keys[earlier], keys[later]
            if self.trackHistory:
                stack.setdefault(branch, set()).add((ke, kl))        # setExplored(branchI, (earlier, later)) (:1068-1070)
                explored.add((ke, kl))
                if self._early_drop and (kl, ke) in explored:
                    # getNext would skip this point when it is popped (:1153-1157): isExplored only ever grows,
                    # so dropping it now is the same exploration with a shorter queue
                    continue
            prio = self.backtrackHeuristic.priority(trace, branch, later, earlier)
            push(self.backTrack, (tuple(-x for x in prio), self._seq, (kl, ke), trace, later, earlier))
            self._seq += 1

    def explore(self, externals, lookingFor: Optional[ViolationFingerprint] = None,
                max_interleavings: Optional[int] = None, stats: Optional[MinimizationStats] = None) -> Exploration:
        externals = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        params = self._params(lookingFor)
        res = Exploration()
        shared = [0]
        if self._started and self.startFromBackTrackPoints and self.backTrack:
            # test() again on the same instance (ResumableDPOR): continue from the backtrack queue (:1219-1220)
            nxt = self._get_next()
            frontier = [nxt] if nxt is not None else []
            shared = [self._next_shared]
        elif self._initialTrace is not None:
            frontier = [self._initialTrace]                           # setInitialTrace (:1220-1221)
        else:
            frontier = [np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)]        # first run: nextTrace is empty
        self._started = True
        while frontier:
            verdicts, traces, pairs = self._run(externals, frontier, params, shared)
            res.rounds.append(len(frontier))
            if stats is not None:
                stats.increment_replays(len(frontier))
            for k in range(len(frontier)):
                il = Interleaving(verdicts[k], traces[k], len(frontier[k]))
                res.interleavings.append(il)
                if int(verdicts[k]["flags"]) & K3_INCOMPLETE:
                    res.aborted += 1
                if int(verdicts[k]["flags"]) & T.V_VIOLATION:
                    res.violations.append(len(res.interleavings) - 1)
                    if self.shortestTraceSoFar is None or len(traces[k]) < len(self.shortestTraceSoFar):
                        self.shortestTraceSoFar = traces[k]               # checkInvariant (:405-409)
                self._absorb(traces[k], pairs[k])
            if self.stopIfViolationFound and self.shortestTraceSoFar is not None:
                break
            if max_interleavings is not None and len(res.interleavings) >= max_interleavings:
                break
            frontier, shared = [], []
            while len(frontier) < self.batch:
                if max_interleavings is not None and len(res.interleavings) + len(frontier) >= max_interleavings:
                    break
                nxt = self._get_next()
                if nxt is None:
                    break
                frontier.append(nxt)
                shared.append(self._next_shared)
        res.exhausted = not self.backTrack and not frontier and res.aborted == 0
        return res

    def explore_native(self, externals, lookingFor: Optional[ViolationFingerprint] = None, max_interleavings: int = 100000,
                       reference_order: bool = False):
        """The same exploration with the queue / explored-set bookkeeping run natively inside
        libdemi_gpu.so (demi_dpor_explore): identical rounds, verdicts and prefix lengths, two orders of
        magnitude less host time per interleaving than this Python loop.  Single rank only.
        reference_order: commit the interleavings in DPORwHeuristics' own one-at-a-time order (the sequence batch = 1
        gives) while the device speculates `batch` wide (DEMI_DPOR_ORDER_REFERENCE)."""
        from . import _native
        arvind = isinstance(self.backtrackHeuristic, ArvindDistanceOrdering)
        if not arvind and type(self.backtrackHeuristic) is not DefaultBacktrackOrdering:
            raise NotImplementedError("demi_dpor_explore knows DefaultBacktrackOrdering and ArvindDistanceOrdering; use explore()")
        if self.backTrack:
            raise NotImplementedError("this instance has explored through the Python loop: its queue is not the library's")
        ordered = arvind or self.should_cap_distance or self._initialTrace is not None
        if ordered and reference_order:
            raise NotImplementedError("the reference order runs with DefaultBacktrackOrdering, no cap and no initial trace")
        externals = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
            self._ctx.model_load(self.schedulerConfig.model.to_struct())
            if self.specialize or getattr(self.schedulerConfig.model, "compiled_only", False):
                self._ctx.model_specialize()
            self._ctx.dpor_load(externals)
        search = T.DporSearch(self.batch, max_interleavings, 1 if self.stopIfViolationFound else 0,
                              1 if self.trackHistory else 0, T.DPOR_ORDER_REFERENCE if reference_order else T.DPOR_ORDER_ROUNDS, 0,
                              T.DPOR_ORDERING_ARVIND if arvind else T.DPOR_ORDERING_DEFAULT,
                              self.stop_at_distance if self.should_cap_distance else None,
                              # a later test() of this instance continues from the queue the library kept (:1219-1220)
                              resume=1 if (ordered and self._started and self.startFromBackTrackPoints) else 0)
        self._started = True
        if ordered:
            self._ctx.dpor_set_traces(getattr(self.backtrackHeuristic, "originalTrace", None) if arvind else None, self._initialTrace)
        verdicts, plen, rounds, vtrace, stats = self._ctx.dpor_explore(self._params(lookingFor), search)
        res = Exploration()
        res.rounds = [int(r) for r in rounds]
        res.aborted = int(np.count_nonzero(verdicts["flags"] & K3_INCOMPLETE))
        res.exhausted = bool(stats.exhausted) and res.aborted == 0
        empty = np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)
        for k in range(len(verdicts)):
            res.interleavings.append(Interleaving(verdicts[k], empty, int(plen[k])))
            if int(verdicts[k]["flags"]) & T.V_VIOLATION:
                res.violations.append(k)
        if len(vtrace):
            res.interleavings[int(stats.first_violation)].trace = vtrace
            self.shortestTraceSoFar = vtrace
        self.interleavingCounter += len(verdicts)
        self.last_native_stats = stats
        return res

    def test(self, events, violation_fingerprint: ViolationFingerprint, _stats: Optional[MinimizationStats] = None):
        """TestOracle.test (:1193-1242): Some(trace of a matching violation) or None."""
        if self.stopIfViolationFound and self.shortestTraceSoFar is not None:
            return self.shortestTraceSoFar
        if self.native:
            # The reference's test() explores until the queue is empty, its head reaches the distance cap, or a violation is
            # found - never "until a budget".  One native call explores at most native_budget interleavings; a call that stopped
            # BECAUSE of the budget (it used all of it, found nothing, points are still queued) is therefore continued from the
            # queue the library kept (resume, the ordered search) - or refused, where the library keeps no queue between calls:
            # "budget ran out" must never read as "the subsequence does not reproduce the violation".
            res = self.explore_native(events, violation_fingerprint, max_interleavings=self.native_budget)
            total = len(res.interleavings)
            while not res.violations and total and len(res.interleavings) >= self.native_budget and \
                    int(self.last_native_stats.queue_len) > 0 and not bool(self.last_native_stats.exhausted):
                ordered = isinstance(self.backtrackHeuristic, ArvindDistanceOrdering) or self.should_cap_distance or self._initialTrace is not None
                if not (ordered and self.startFromBackTrackPoints):
                    raise RuntimeError("DPORwHeuristics.test(native=True): %d interleavings explored, no violation, %d backtrack points "
                                       "still queued - raise native_budget (the search is not resumable without an ordering, a "
                                       "distance cap or an initial trace)" % (total, int(self.last_native_stats.queue_len)))
                res = self.explore_native(events, violation_fingerprint, max_interleavings=self.native_budget)
                total += len(res.interleavings)
            if _stats is not None:
                _stats.increment_replays(total)
        else:
            res = self.explore(events, violation_fingerprint, stats=_stats)
        return res.interleavings[res.violations[0]].trace if res.violations else None

    def shutdown(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
