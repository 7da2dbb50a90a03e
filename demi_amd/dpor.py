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
    exhausted: bool = False                                  # the backtrack queue ran empty

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


class DPORwHeuristics:
    """DPORwHeuristics(schedulerConfig, depth_bound=..., stopIfViolationFound=..., trackHistory=...)."""

    def __init__(self, schedulerConfig: SchedulerConfig, depth_bound: Optional[int] = None,
                 stopIfViolationFound: bool = True, trackHistory: bool = True, batch: int = 256,
                 max_pairs: int = 4096, p_max: int = 64, device: int = 0, backend: Optional[Callable] = None,
                 specialize: bool = False):
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
        self._backend = backend          # tests inject the CPU oracle here
        self.specialize = specialize     # compile the model's table to native code first (pays off on long explorations)
        self._device = device
        self._ctx = None
        self.backTrack: list = []        # heap of (-branch, seq, (later key, earlier key), trace, later, earlier)
        self._seq = 0
        self.exploredTracker = ExploredTacker()
        self.interleavingCounter = 0
        self.shortestTraceSoFar: Optional[np.ndarray] = None

    def getName(self) -> str:
        return "DPORwHeuristics"

    def setMaxMessagesToSchedule(self, _max_messages: int):
        self.max_messages = _max_messages

    def setDepthBound(self, d: int):
        self.depth_bound = d

    # -- one launch
    def _params(self, lookingFor: Optional[ViolationFingerprint]) -> T.DporParams:
        return T.DporParams(self.depth_bound or 0, self.max_messages, 1 if lookingFor is not None else 0,
                            lookingFor.code if lookingFor is not None else 0, self.p_max, self.max_pairs)

    def _run(self, externals, prefixes, params):
        from .distributed import sharded_batch
        if self._backend is not None:
            fn = lambda part: self._backend(self.schedulerConfig.model, externals, part, params)
        else:
            if self._ctx is None:
                from . import _native
                self._ctx = _native.Context(self._device)
                self._ctx.model_load(self.schedulerConfig.model.to_struct())
                if self.specialize:
                    self._ctx.model_specialize()
                self._ctx.dpor_load(externals)
            fn = lambda part: self._ctx.dpor_batch(part, params)
        return sharded_batch(prefixes, fn)

    # -- getNext (:1142-1162): pop the deepest unexplored backtrack point
    def _get_next(self):
        while self.backTrack:
            neg_branch, _, pair, trace, later, earlier = heapq.heappop(self.backTrack)
            if self.trackHistory and pair in self.exploredTracker._any:
                continue
            branch = -neg_branch
            if self.trackHistory:
                self.exploredTracker.setExplored(branch, pair)
            # next trace = trace.take(branch + 1) ++ needToReplay (:1054-1057, 1180), built on demand
            idx = [i for i in range(branch + 1, later + 1) if i != earlier]
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
                if (kl, ke) in explored:
                    # getNext would skip this point when it is popped (:1153-1157): isExplored only ever grows,
                    # so dropping it now is the same exploration with a shorter queue
                    continue
            push(self.backTrack, (-branch, self._seq, (kl, ke), trace, later, earlier))
            self._seq += 1

    def explore(self, externals, lookingFor: Optional[ViolationFingerprint] = None,
                max_interleavings: Optional[int] = None, stats: Optional[MinimizationStats] = None) -> Exploration:
        externals = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        params = self._params(lookingFor)
        res = Exploration()
        frontier = [np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)]            # first run: nextTrace is empty
        while frontier:
            verdicts, traces, pairs = self._run(externals, frontier, params)
            res.rounds.append(len(frontier))
            if stats is not None:
                stats.increment_replays(len(frontier))
            for k in range(len(frontier)):
                il = Interleaving(verdicts[k], traces[k], len(frontier[k]))
                res.interleavings.append(il)
                if int(verdicts[k]["flags"]) & T.V_VIOLATION:
                    res.violations.append(len(res.interleavings) - 1)
                    if self.shortestTraceSoFar is None or len(traces[k]) < len(self.shortestTraceSoFar):
                        self.shortestTraceSoFar = traces[k]               # checkInvariant (:405-409)
                self._absorb(traces[k], pairs[k])
            if self.stopIfViolationFound and self.shortestTraceSoFar is not None:
                break
            if max_interleavings is not None and len(res.interleavings) >= max_interleavings:
                break
            frontier = []
            while len(frontier) < self.batch:
                if max_interleavings is not None and len(res.interleavings) + len(frontier) >= max_interleavings:
                    break
                nxt = self._get_next()
                if nxt is None:
                    break
                frontier.append(nxt)
        res.exhausted = not self.backTrack and not frontier
        return res

    def explore_native(self, externals, lookingFor: Optional[ViolationFingerprint] = None, max_interleavings: int = 100000):
        """The same exploration with the queue / explored-set bookkeeping run natively inside
        libdemi_gpu.so (demi_dpor_explore): identical rounds, verdicts and prefix lengths, two orders of
        magnitude less host time per interleaving than this Python loop.  Single rank only."""
        from . import _native
        externals = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
            self._ctx.model_load(self.schedulerConfig.model.to_struct())
            if self.specialize:
                self._ctx.model_specialize()
            self._ctx.dpor_load(externals)
        search = T.DporSearch(self.batch, max_interleavings, 1 if self.stopIfViolationFound else 0,
                              1 if self.trackHistory else 0)
        verdicts, plen, rounds, vtrace, stats = self._ctx.dpor_explore(self._params(lookingFor), search)
        res = Exploration()
        res.rounds = [int(r) for r in rounds]
        res.exhausted = bool(stats.exhausted)
        empty = np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)
        for k in range(len(verdicts)):
            res.interleavings.append(Interleaving(verdicts[k], empty, int(plen[k])))
            if int(verdicts[k]["flags"]) & T.V_VIOLATION:
                res.violations.append(k)
        if len(vtrace):
            res.interleavings[int(stats.first_violation)].trace = vtrace
            self.shortestTraceSoFar = vtrace
        self.interleavingCounter += len(verdicts)
        return res

    def test(self, events, violation_fingerprint: ViolationFingerprint, _stats: Optional[MinimizationStats] = None):
        """TestOracle.test (:1193-1242): Some(trace of a matching violation) or None."""
        if self.stopIfViolationFound and self.shortestTraceSoFar is not None:
            return self.shortestTraceSoFar
        res = self.explore(events, violation_fingerprint, stats=_stats)
        return res.interleavings[res.violations[0]].trace if res.violations else None

    def shutdown(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None
