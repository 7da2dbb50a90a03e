"""Internal-event minimization: which deliveries of the MCS execution can be dropped as well.

Host-side mirror of minification/internal_minimization/{RemovalStrategy, OneAtATimeRemoval,
ScheduleCheckers}.scala and RunnerUtils.minimizeInternals (RunnerUtils.scala:980-1003).  The
replays are K2 launches (demi_replay_removal_batch): a removal strategy proposes its candidates
one after another, each assuming the previous one failed, so the whole remaining sequence
"lastFailingTrace minus delivery i" is known up front and is evaluated in ONE launch; the first
candidate (in proposal order) that still triggers the violation is adopted, exactly as the
one-replay-at-a-time loop of STSSchedMinimizer.minimize (ScheduleCheckers.scala:35-107) would.

A delivery is identified the way the reference does, by (snd, rcv, MessageFingerprint); on the
table-encoded model the fingerprint is (msg_type, p0, p1).  BeginUnignorableEvents blocks
(OneAtATimeRemoval.scala:28-46) do not exist in the recorded format; external deliveries are
unignorable as in the reference.
"""
from collections import Counter
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _native
from . import types as T
from .model import Model
from .schedulers import EventTrace, MinimizationStats, SchedulerConfig, ViolationFingerprint

Key = Tuple[int, int, Tuple[int, int, int]]       # (snd, rcv, fingerprint)
NO_SKIP = 0xFFFFFFFF


# ------------------------------------------------------------------ trace helpers
def deliveries(trace: EventTrace) -> List[Tuple[int, Key, int]]:
    """[(index in trace.events, (snd, rcv, fingerprint), flags)] of every MsgEvent / TimerDelivery,
    in trace order (RunnerUtils.getFingerprintedDeliveries, RunnerUtils.scala:1288-1313)."""
    cached = getattr(trace, "_deliveries", None)
    if cached is not None:
        return cached
    ev = trace.events
    idx = np.nonzero(ev["kind"] == T.REC_MSG_EVENT)[0]
    # (the fingerprint of a message: its type and every payload field - p_hi holds the fields past the second one of a
    # DEMI_MODEL_PAYLOADS table and is 0 otherwise)
    out = [(int(i), (int(ev["snd"][i]), int(ev["rcv"][i]), (int(ev["msg_type"][i]), int(ev["p0"][i]), int(ev["p1"][i])) +
                     ((int(ev["p_hi"][i]),) if int(ev["p_hi"][i]) else ())),
            int(ev["flags"][i])) for i in idx]
    trace._deliveries = out
    return out


def countMsgEvents(trace: EventTrace) -> int:
    """RunnerUtils.countMsgEvents (RunnerUtils.scala:1315-1323)."""
    return int(np.count_nonzero(trace.events["kind"] == T.REC_MSG_EVENT))


def getFingerprintedDeliveries(trace: EventTrace) -> List[Key]:
    return [k for _, k, _ in deliveries(trace)]


def executed_trace(trace: EventTrace, kept: np.ndarray, externals: Optional[np.ndarray] = None,
                   subseq: Optional[Sequence[int]] = None) -> EventTrace:
    """The EventTrace STSScheduler.test returns on success (STSScheduler.scala:286-292) from the
    `kept` marks of demi_replay_get_kept: the recorded events that took effect, in order.  With
    `subseq` (indices into trace.original_externals) the result is re-based on that subsequence:
    setOriginalExternalEvents(mcs) (RunnerUtils.scala:698) and ext_idx renumbered to it."""
    ev = trace.events[np.asarray(kept, dtype=bool)].copy()
    ext = trace.original_externals if externals is None else externals
    if subseq is not None:
        subseq = sorted(int(i) for i in subseq)
        remap = np.full(256, 255, dtype=np.uint8)
        for new, old in enumerate(subseq):
            remap[old] = new
        ev["ext_idx"] = remap[ev["ext_idx"]]
        ext = trace.original_externals[subseq].copy()
    return EventTrace(ev, ext)


# ------------------------------------------------------------------ removal strategies
class RemovalStrategy:
    """RemovalStrategy.scala:4-24."""

    @property
    def unignorable(self) -> int:
        raise NotImplementedError

    def next_index(self, lastFailingTrace: EventTrace, alreadyRemoved: Counter, violationTriggered: bool) -> Optional[int]:
        """Index (in lastFailingTrace.events) of the delivery the next schedule drops, None when done."""
        raise NotImplementedError

    def getNextTrace(self, lastFailingTrace: EventTrace, alreadyRemoved: Counter,
                     violationTriggered: bool) -> Optional[EventTrace]:
        i = self.next_index(lastFailingTrace, alreadyRemoved, violationTriggered)
        if i is None:
            return None
        return EventTrace(np.delete(lastFailingTrace.events, i), lastFailingTrace.original_externals)

    def clone(self) -> "RemovalStrategy":
        raise NotImplementedError


class OneAtATimeStrategy(RemovalStrategy):
    """OneAtATimeRemoval.scala:17-131: only ever removes one delivery from the last failing trace."""

    def __init__(self, verified_mcs: EventTrace, model: Model):
        """`model` stands where the reference passes the messageFingerprinter: it carries the fingerprint
        (msg_type, p0, p1) and the application's external-message filter (msg_class == EXTERNAL,
        EventTypes.setExternalMessageFilter, ExternalEvents.scala:157-166)."""
        self.verified_mcs = verified_mcs
        self.deadLetters = T.deadletters_of(model.n_actors)       # (31 for a table of more than 8 actors: include/demi_gpu.h)
        # deliveries we have tried ignoring so far; external messages are never ignored (:32-35)
        self.triedIgnoring: Counter = Counter()
        for _, key, _ in deliveries(verified_mcs):
            if model.msg_class[key[2][0]] == T.MSG_EXTERNAL:
                self.triedIgnoring[key] += 1
        self._unignorable = sum(self.triedIgnoring.values())

    @property
    def unignorable(self) -> int:
        return self._unignorable

    def next_index(self, trace, alreadyRemoved, violationTriggered):
        # :57-124.  keysThisIteration counts the occurrences seen so far, plus everything pruned earlier
        keysThisIteration = Counter(alreadyRemoved)
        for idx, key, _ in deliveries(trace):
            keysThisIteration[key] += 1
            if keysThisIteration[key] > self.triedIgnoring[key] and self.choiceFilter(*key):
                self.triedIgnoring[key] += 1
                return idx
        return None

    def choiceFilter(self, snd: int, rcv: int, fingerprint) -> bool:
        raise NotImplementedError

    def _copy_base(self, other):
        other.verified_mcs = self.verified_mcs
        other.deadLetters = self.deadLetters
        other.triedIgnoring = Counter(self.triedIgnoring)
        other._unignorable = self._unignorable


class LeftToRightOneAtATime(OneAtATimeStrategy):
    """OneAtATimeRemoval.scala:134-139."""

    def choiceFilter(self, snd, rcv, fingerprint):
        return True

    def clone(self):
        c = LeftToRightOneAtATime.__new__(LeftToRightOneAtATime)
        self._copy_base(c)
        return c


class SrcDstFIFORemoval(OneAtATimeStrategy):
    """OneAtATimeRemoval.scala:141-251: per (src, dst) pair only the last message of the FIFO is tried;
    a pair is abandoned as soon as one of its removals fails.  Timers are tried in trace order."""

    def __init__(self, verified_mcs: EventTrace, model: Model):
        super().__init__(verified_mcs, model)
        self.srcDstToMessages = {}
        for _, (snd, rcv, fp), _ in deliveries(verified_mcs):
            if snd == self.deadLetters:
                continue
            self.srcDstToMessages.setdefault((snd, rcv), []).append(fp)
        self.previouslyChosenSrcDst: Optional[Tuple[int, int]] = None
        self.srcDstToCurrentIdx = {}
        self._reset_idx()

    def _reset_idx(self):
        for k in self.srcDstToMessages:
            self.srcDstToCurrentIdx[k] = -1

    def choiceFilter(self, snd, rcv, fingerprint):
        k = (snd, rcv)
        if k in self.srcDstToMessages:
            self.srcDstToCurrentIdx[k] += 1
            lst = self.srcDstToMessages[k]
            if self.srcDstToCurrentIdx[k] == len(lst) - 1:
                self.srcDstToMessages[k] = lst[:-1]
                if not self.srcDstToMessages[k]:
                    del self.srcDstToMessages[k]
                self.previouslyChosenSrcDst = k
                return True
        self.previouslyChosenSrcDst = None
        return snd == self.deadLetters

    def next_index(self, trace, alreadyRemoved, violationTriggeredLastRun):
        if not violationTriggeredLastRun and self.previouslyChosenSrcDst is not None:
            self.srcDstToMessages.pop(self.previouslyChosenSrcDst, None)     # ignoring didn't work: pair is done
        if violationTriggeredLastRun:
            # some FIFO entries may have been pruned as absent "freebies": recompute, in reverse (:222-243)
            self.srcDstToMessages.clear()
            removed = Counter(alreadyRemoved)
            for _, key, _ in reversed(deliveries(self.verified_mcs)):
                snd, rcv, fp = key
                if snd == self.deadLetters:
                    continue
                if removed[key] > 0:
                    removed[key] -= 1
                else:
                    self.srcDstToMessages.setdefault((snd, rcv), []).insert(0, fp)
        self._reset_idx()
        return super().next_index(trace, alreadyRemoved, violationTriggeredLastRun)

    def clone(self):
        c = SrcDstFIFORemoval.__new__(SrcDstFIFORemoval)
        self._copy_base(c)
        c.srcDstToMessages = {k: list(v) for k, v in self.srcDstToMessages.items()}
        c.previouslyChosenSrcDst = self.previouslyChosenSrcDst
        c.srcDstToCurrentIdx = dict(self.srcDstToCurrentIdx)
        return c


# ------------------------------------------------------------------ the replay oracle on the GPU
class StsRemovalOracle:
    """RunnerUtils.testWithStsSched (RunnerUtils.scala:913-943) for the traces a removal strategy
    proposes: `new STSScheduler(config, trace, false).test(mcs, violation)`; K2 with a removed delivery."""

    def __init__(self, schedulerConfig: SchedulerConfig, device: int = 0, p_max: int = 64):
        if schedulerConfig.model is None or schedulerConfig.model.inv_kind == T.INV_NONE:
            raise ValueError("Must invoke setInvariant before test()")
        self.schedulerConfig = schedulerConfig
        self.p_max = p_max
        self._ctx = _native.Context(device)
        self._ctx.model_load(schedulerConfig.model.to_struct())
        if getattr(schedulerConfig.model, "compiled_only", False):
            self._ctx.model_specialize()         # a wide table (DEMI_MODEL_WIDE) runs only as compiled code
        self._loaded = None

    def _limits(self, fp: ViolationFingerprint) -> T.Limits:
        return T.Limits(0, 0, self.p_max, 1, fp.code, 1 if self.schedulerConfig.populate_all_actors else 0, 0,
                        int(self.schedulerConfig.filterKnownAbsents))

    def _load(self, trace: EventTrace):
        if self._loaded is not trace:
            self._ctx.replay_load(trace.original_externals, trace.events)
            self._loaded = trace

    def test_removals(self, trace: EventTrace, skips: Sequence[int], violation: ViolationFingerprint) -> List[bool]:
        """element i: does `trace` minus the delivery at skips[i] still trigger the violation?"""
        from .distributed import sharded_map
        self._load(trace)
        lim = self._limits(violation)

        def run(part):
            # a replay aborted on a capacity is no answer: repeat it with the largest pending set, else give up loudly
            from .schedulers import CapacityExceeded, OVF_FLAGS
            v = self._ctx.replay_removal_batch(part, lim)
            bad = np.nonzero(v["flags"] & OVF_FLAGS)[0]
            if len(bad):
                big = self._limits(violation)
                big.p_max = T.MAX_PENDING
                v[bad] = self._ctx.replay_removal_batch([part[i] for i in bad], big)
                if (v["flags"] & OVF_FLAGS).any():
                    raise CapacityExceeded("a removal candidate's replay exceeds the engine's capacities")
            return [bool(f & T.V_VIOLATION) for f in v["flags"]]
        return sharded_map(list(skips), run)

    def executed(self, trace: EventTrace, skip: int, violation: ViolationFingerprint) -> Optional[EventTrace]:
        """test() of one candidate: Some(executed trace) iff it triggers the violation."""
        from .schedulers import CapacityExceeded, OVF_FLAGS
        self._load(trace)
        lim = self._limits(violation)
        v, kept = self._ctx.replay_get_kept(len(trace.events), skip, lim)
        if (int(v.flags) & OVF_FLAGS) and lim.p_max < T.MAX_PENDING:
            lim.p_max = T.MAX_PENDING
            v, kept = self._ctx.replay_get_kept(len(trace.events), skip, lim)
        if int(v.flags) & OVF_FLAGS:
            raise CapacityExceeded("the replay exceeds the engine's capacities")
        if not (int(v.flags) & T.V_VIOLATION):
            return None
        return executed_trace(trace, kept)

    def shutdown(self):
        self._ctx.close()


# ------------------------------------------------------------------ the minimizer
class STSSchedMinimizer:
    """ScheduleCheckers.scala:19-108.  One-time use.  `max_batch` bounds how many of the strategy's upcoming
    candidates are evaluated per launch (1 = the reference's sequential loop)."""

    def __init__(self, mcs: np.ndarray, verified_mcs: EventTrace, violation: ViolationFingerprint,
                 removalStrategy: RemovalStrategy, oracle, stats: Optional[MinimizationStats] = None,
                 max_batch: int = 1 << 14):
        self.mcs = mcs
        self.verified_mcs = verified_mcs
        self.violation = violation
        self.removalStrategy = removalStrategy
        self.oracle = oracle
        self._stats = stats or MinimizationStats()
        self.max_batch = max(1, int(max_batch))
        self.speculative_replays = 0
        self.batches: List[int] = []
        self.internal_sizes: List[int] = []       # record_internal_size after every (sequential) replay

    def minimize(self) -> Tuple[MinimizationStats, EventTrace]:
        lastFailingTrace = EventTrace(self.verified_mcs.events, self.mcs)
        lastFailingSize = countMsgEvents(lastFailingTrace)
        prunedOverall: Counter = Counter()
        violationTriggered = False
        while True:
            # the strategy's upcoming proposals, each assuming the one before it failed
            spec = self.removalStrategy.clone()
            cands: List[int] = []
            vt = violationTriggered
            while len(cands) < self.max_batch:
                i = spec.next_index(lastFailingTrace, prunedOverall, vt)
                if i is None:
                    break
                cands.append(i)
                vt = False
            if not cands:
                break
            results = self.oracle.test_removals(lastFailingTrace, cands, self.violation)
            self.speculative_replays += len(cands)
            self.batches.append(len(cands))
            j = next((k for k, r in enumerate(results) if r), None)
            consumed = len(cands) if j is None else j + 1
            # bring the real strategy to where the sequential loop would be
            vt = violationTriggered
            for k in range(consumed):
                i = self.removalStrategy.next_index(lastFailingTrace, prunedOverall, vt)
                assert i == cands[k]
                vt = False
            self._stats.increment_replays(consumed)
            self.internal_sizes.extend([lastFailingSize] * (consumed if j is None else consumed - 1))
            if j is None:
                violationTriggered = False
                continue        # batch was cut by max_batch, or the next proposal is None (-> loop ends above)
            trace = self.oracle.executed(lastFailingTrace, cands[j], self.violation)
            assert trace is not None, "batched and single replay of the same candidate disagree"
            # other deliveries may have been pruned by virtue of being absent (:58-92)
            prunedThisRun = Counter(getFingerprintedDeliveries(lastFailingTrace)) - \
                Counter(getFingerprintedDeliveries(trace))
            prunedOverall += prunedThisRun
            lastFailingTrace = EventTrace(trace.events, self.mcs)
            lastFailingSize = countMsgEvents(lastFailingTrace)
            self.internal_sizes.append(lastFailingSize)
            violationTriggered = True
        return self._stats, lastFailingTrace


def minimizeInternals(schedulerConfig: SchedulerConfig, mcs: np.ndarray, verified_mcs: EventTrace,
                      violation: ViolationFingerprint, removalStrategyCtor=None, oracle=None,
                      stats: Optional[MinimizationStats] = None, device: int = 0, p_max: int = 64,
                      max_batch: int = 1 << 14) -> Tuple[MinimizationStats, EventTrace]:
    """RunnerUtils.minimizeInternals (RunnerUtils.scala:980-1003).  pre: replaying verified_mcs reproduces the
    violation.  removalStrategyCtor == None uses LeftToRightOneAtATime."""
    strategy = LeftToRightOneAtATime(verified_mcs, schedulerConfig.model) if removalStrategyCtor is None \
        else removalStrategyCtor()
    own = oracle is None
    if own:
        oracle = StsRemovalOracle(schedulerConfig, device=device, p_max=p_max)
    try:
        return STSSchedMinimizer(mcs, verified_mcs, violation, strategy, oracle, stats=stats,
                                 max_batch=max_batch).minimize()
    finally:
        if own:
            oracle.shutdown()
