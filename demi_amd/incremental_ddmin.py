"""DDMin over DPOR with a growing edit-distance bound: the reference's `editDistanceDporDDMin`.

Host-side mirror of minification/IncrementalDeltaDebugging.scala (IncrementalDDMin :20-92, ResumableDPOR
:94-122), schedulers/BacktrackOrdering.scala (ArvindDistanceOrdering, in demi_amd/dpor.py) and
RunnerUtils.editDistanceDporDDMin (RunnerUtils.scala:810-879).  Every oracle consultation is one bounded
DPORwHeuristics exploration, i.e. rounds of K3 launches; nothing here executes a schedule on the CPU.

How the pieces map:
  * the original (fuzz) execution's deliveries become DPOR's initial trace: node identity is the hash chain of
    the causal path on both sides (include/demi_gpu.h), so the recorded EventTrace of K1 is converted by
    following its MsgSend -> MsgEvent ids (`dpor_initial_trace`); the reference needs the serialized depGraph of
    the original run for the same purpose (DepTracker.scala:27-175, noopWaitQuiescence: externals hang off the root);
  * `convertToDPORTrace` (DPORwHeuristics.scala:1270-1303): DPOR only understands Start / Send (and optionally
    WaitQuiescence); Kill / Partition / UnPartition are dropped from the externals, exactly as the reference does;
  * ResumableDPOR keeps one DPORwHeuristics per external-event subsequence, so that raising the distance bound
    continues each exploration from its backtrack queue instead of starting over.
"""
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import types as T
from .dpor import ArvindDistanceOrdering, DPORwHeuristics
from .minification import DDMin, EventDagView, UnmodifiedEventDag
from .schedulers import EventTrace, MinimizationStats, SchedulerConfig, ViolationFingerprint


def msg_word(msg_type: int, src: int, dst: int, p0: int, p1: int) -> int:
    return msg_type | (dst << 5) | (src << 8) | (p0 << 16) | (p1 << 24)


def rec_msg_word(e, wide: bool = False, big: bool = False) -> int:
    """The message word of a recorded MsgSend / MsgEvent: 32 bits, or the 64 bits of a wide table (header | payload area << 16,
    the area being the record's p0 | p1 << 16 | p_hi << 32: include/demi_gpu.h)."""
    if not wide:
        return msg_word(int(e["msg_type"]), int(e["snd"]), int(e["rcv"]), int(e["p0"]), int(e["p1"]))
    # (big: the layout of a table with more than 8 actors - the sender field sits behind a 4-bit receiver)
    return int(e["msg_type"]) | (int(e["rcv"]) << 5) | (int(e["snd"]) << (9 if big else 8)) | (T.rec_area(e) << 16)


def dpor_initial_trace(trace: EventTrace, model=None) -> np.ndarray:
    """DepTracker.getInitialTrace (DepTracker.scala:130-133, 173) of a recorded execution: the root followed by
    every delivery, each identified by the hash chain of its causal path (parent = the delivery during which it
    was sent; the root for external messages).  model: needed for a wide table (its node keys hash the 64-bit message word;
    the trace entry reports the word's low half)."""
    wide = bool(model is not None and getattr(model, "wide", False))
    big = bool(model is not None and model.n_actors > T.MAX_ACTORS)
    ev = trace.events
    key_of_id: Dict[int, Tuple[int, int]] = {}          # Uniq id -> (node key, trace index of its producer)
    out = [(T.DPOR_ROOT_KEY, 0, 0, 0, 0, 0)]
    cur_key, cur_idx = T.DPOR_ROOT_KEY, 0
    depth_of = [0]
    for e in ev:
        kind = int(e["kind"])
        if kind == T.REC_MSG_SEND:
            w = rec_msg_word(e, wide, big)
            ext = bool(int(e["flags"]) & 1)
            pk, pi = (T.DPOR_ROOT_KEY, 0) if ext else (cur_key, cur_idx)
            key_of_id[int(e["id"])] = (((pk ^ w) * T.DPOR_PRIME) & 0xFFFFFFFFFFFFFFFF, pi)
        elif kind == T.REC_MSG_EVENT:
            w = rec_msg_word(e, wide, big)
            k, pi = key_of_id[int(e["id"])]
            depth_of.append(depth_of[pi] + 1)
            out.append((k, w & 0xFFFFFFFF, pi & 0xFF, 0, depth_of[-1] & 0xFF, 1))
            cur_key, cur_idx = k, len(out) - 1
    return np.array(out, dtype=T.DPOR_TRACE_DTYPE)


def convertToDPORTrace(externals: np.ndarray, ignoreQuiescence: bool = True) -> np.ndarray:
    """DPORwHeuristicsUtil.convertToDPORTrace (DPORwHeuristics.scala:1279-1303)."""
    keep = [i for i, e in enumerate(externals)
            if int(e["kind"]) in (T.EV_START, T.EV_SEND) or (int(e["kind"]) == T.EV_WAIT_QUIESCENCE and not ignoreQuiescence)]
    return np.ascontiguousarray(externals[keep], dtype=T.EXT_EVENT_DTYPE)


class ResumableDPOR:
    """IncrementalDeltaDebugging.scala:94-122: a TestOracle that keeps one DPOR instance per external-event
    subsequence.  `events` are indices into `externals` (what the EventDag hands to the oracle)."""

    def __init__(self, ctor: Callable[[], DPORwHeuristics], externals: np.ndarray, ignoreQuiescence: bool = True):
        self.ctor = ctor
        self.externals = externals
        self.ignoreQuiescence = ignoreQuiescence
        self.subseqToDPOR: Dict[Tuple[int, ...], DPORwHeuristics] = {}
        self.currentMaxDistance = 0

    def getName(self) -> str:
        return "DPOR"

    def setMaxDistance(self, dist: int):
        self.currentMaxDistance = dist

    def test(self, events: Sequence[int], violation_fingerprint: ViolationFingerprint,
             stats: Optional[MinimizationStats] = None):
        key = tuple(events)
        if key not in self.subseqToDPOR:
            self.subseqToDPOR[key] = self.ctor()
        dpor = self.subseqToDPOR[key]
        dpor.setMaxDistance(self.currentMaxDistance)
        ext = convertToDPORTrace(self.externals[list(key)], self.ignoreQuiescence)
        return dpor.test(ext, violation_fingerprint, stats)

    def shutdown(self):
        for d in self.subseqToDPOR.values():
            d.shutdown()


class IncrementalDDMin:
    """IncrementalDeltaDebugging.scala:20-92: DDMin with maxDistance 0, then 2, 4, ... < maxMaxDistance, each pass
    starting from the previous pass's MCS; stops early once the MCS has at most stopAtSize events."""

    def __init__(self, oracle: ResumableDPOR, maxMaxDistance: int = 256, stopAtSize: int = 1,
                 checkUnmodifed: bool = False, stats: Optional[MinimizationStats] = None):
        self.oracle = oracle
        self.maxMaxDistance = maxMaxDistance
        self.stopAtSize = stopAtSize
        self.checkUnmodifed = checkUnmodifed
        self._stats = stats or MinimizationStats()
        self.ddmin: Optional[DDMin] = None
        self.distances: List[Tuple[int, int]] = []          # (distance, MCS size after the pass)
        self.consulted_all: List[Tuple[Tuple[int, ...], bool, int]] = []      # every pass's consultations: (events, passes, cap)

    def minimize(self, dag, violation_fingerprint: ViolationFingerprint):
        currentDistance = 0
        self.oracle.setMaxDistance(currentDistance)
        if self.checkUnmodifed:
            if self.oracle.test(dag.get_all_events(), violation_fingerprint, self._stats) is None:
                raise ValueError("Unmodified trace does not trigger violation")
        self._stats.total_replays = 0                       # _stats.reset()
        currentMCS = dag
        while currentDistance < self.maxMaxDistance and currentMCS.length > self.stopAtSize:
            self.ddmin = DDMin(self.oracle, checkUnmodifed=False)
            currentMCS = self.ddmin.minimize(currentMCS, violation_fingerprint)
            self._stats.total_replays += self.ddmin._stats.total_replays       # mergeStats (:33-41)
            self.consulted_all += [(c, p, currentDistance) for c, p in self.ddmin.consulted]
            self.distances.append((currentDistance, currentMCS.length))
            currentDistance = 2 if currentDistance == 0 else currentDistance << 1
            self.oracle.setMaxDistance(currentDistance)
        return currentMCS

    def verify_mcs(self, mcs, _violation_fingerprint: ViolationFingerprint):
        return self.oracle.test(mcs.get_all_events(), _violation_fingerprint, MinimizationStats())


class NativeIncrementalDDMin:
    """What editDistanceDporDDMin(native_loop=True) returns in IncrementalDDMin's place: the same read-only facts."""

    def __init__(self, consulted, passes, st):
        self.consulted_all = [(tuple(c), p, d) for c, p, d in consulted]
        self.distances = list(passes)
        self._stats = MinimizationStats()
        self._stats.total_replays = int(st.replays)
        self.native_stats = st


def editDistanceDporDDMin(schedulerConfig: SchedulerConfig, trace: EventTrace, violation: ViolationFingerprint,
                          ignoreQuiescence: bool = True, stats: Optional[MinimizationStats] = None,
                          stopAtSize: int = 6, maxMaxDistance: int = 8, batch: int = 256, backend=None, device: int = 0,
                          native: bool = False, native_loop: bool = False, specialize: bool = False):
    """RunnerUtils.editDistanceDporDDMin (RunnerUtils.scala:810-879).  `trace` is the violating execution found by
    the fuzzer (its recorded events + the externals that drove it).  Returns (mcs indices into
    trace.original_externals, stats, the DPOR trace that reproduces the violation on the MCS or None, violation).
    native: every DPOR consultation runs inside the library (demi_dpor_explore with ArvindDistanceOrdering, the distance cap,
    the initial trace and - for a subsequence consulted again at a larger distance - its resumable state)."""
    initialTrace = dpor_initial_trace(trace, schedulerConfig.model)
    if native_loop:
        # the whole of this function inside the library (demi_edit_distance_dpor_ddmin, csrc/incddmin_host.hpp): one call
        from . import _native
        ctx = _native.Context(device)
        try:
            ctx.model_load(schedulerConfig.model.to_struct())
            if specialize or getattr(schedulerConfig.model, "compiled_only", False):
                ctx.model_specialize()
            par = T.DporParams(0, len(initialTrace), 1, violation.code, 64, 4096, 1)
            ip = T.IncDdminParams(max_max_distance=maxMaxDistance, stop_at_size=stopAtSize, check_unmodified=0,
                                  ignore_quiescence=1 if ignoreQuiescence else 0, verify_mcs=1, batch=batch)
            mcs, consulted, passes, vtrace, st = ctx.edit_distance_dpor_ddmin(trace.original_externals, initialTrace, par, ip)
        finally:
            ctx.close()
        res = NativeIncrementalDDMin(consulted, passes, st)
        if stats is not None:
            stats.total_replays = res._stats.total_replays
        return mcs, res, vtrace, violation

    def dporConstructor() -> DPORwHeuristics:
        heuristic = ArvindDistanceOrdering()
        dpor = DPORwHeuristics(schedulerConfig, prioritizePendingUponDivergence=True, backtrackHeuristic=heuristic,
                               batch=batch, backend=backend, device=device, native=native)
        dpor.setMaxMessagesToSchedule(len(initialTrace))
        dpor.setInitialTrace(initialTrace)
        heuristic.init(dpor, initialTrace)
        return dpor

    externals = trace.original_externals
    dag = UnmodifiedEventDag(externals)
    # convertToDPORTrace: only Start / Send (and WaitQuiescence unless ignored) take part in the minimization
    keep = tuple(i for i in dag.events if int(externals[i]["kind"]) in (T.EV_START, T.EV_SEND) or
                 (int(externals[i]["kind"]) == T.EV_WAIT_QUIESCENCE and not ignoreQuiescence))
    view = EventDagView(dag, keep)
    resumableDPOR = ResumableDPOR(dporConstructor, externals, ignoreQuiescence)
    ddmin = IncrementalDDMin(resumableDPOR, stopAtSize=stopAtSize, maxMaxDistance=maxMaxDistance, stats=stats)
    try:
        mcs = ddmin.minimize(view, violation)
        verified = ddmin.verify_mcs(mcs, violation) if mcs.length < view.length else None
    finally:
        resumableDPOR.shutdown()
    return mcs.get_all_events(), ddmin, verified, violation
