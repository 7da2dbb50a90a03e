"""Host-side mirror of the reference's scheduler plugin surface for the GPU path.

Mirrors (same names, argument meaning and error behaviour):
  SchedulerConfig      src/main/scala/verification/SchedulerConfig.scala:9-37
  ViolationFingerprint src/main/scala/verification/minification/TestOracle.scala:9-22
  TestOracle.test      src/main/scala/verification/minification/TestOracle.scala:30-55
  RandomScheduler      src/main/scala/verification/schedulers/RandomScheduler.scala:41-613
  FullyRandom          src/main/scala/verification/schedulers/RandomScheduler.scala:635-697
  MinimizationStats    src/main/scala/verification/minification/Minimizer.scala:30-217 (replay counter only)

In production the host is the Scala adapter of INTEGRATION.md calling the same C ABI through JNI;
this Python mirror exists so tests and bench.py drive the kernels the way RunnerUtils would.
Every compute call goes to libdemi_gpu.so; nothing here simulates an execution on the CPU.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _native
from . import types as T
from .model import Model


@dataclass
class SchedulerConfig:
    """SchedulerConfig.scala:9-37.  On the GPU path the message fingerprinter, the actors and the
    invariant are all carried by the lowered `model`; failure detector and checkpointing are off
    (their defaults, :11-12)."""
    model: Optional[Model] = None
    enableFailureDetector: bool = False
    enableCheckpointing: bool = False
    shouldShutdownActorSystem: bool = True
    filterKnownAbsents: int = False      # SchedulerConfig.scala:14; True / 1 = as the reference computes it (types.FILTER_ABSENTS_LITERAL),
                                         # 2 = FILTER_ABSENTS_CORRECTED (see include/demi_gpu.h demi_filter_absents)
    ignoreTimers: bool = False
    abortUponDivergence: bool = False
    populate_all_actors: bool = False    # setActorNamePropPairs


@dataclass(frozen=True)
class ViolationFingerprint:
    """TestOracle.scala:9-22.  `code` is the invariant descriptor's fingerprint word."""
    code: int
    match_mask: int = 0xFFFFFFFF

    def matches(self, other: "ViolationFingerprint") -> bool:
        return ((self.code ^ other.code) & self.match_mask) == 0

    def affectedNodes(self) -> List[int]:
        return T.fingerprint_actors(self.code)


@dataclass
class EventTrace:
    """EventTrace.scala:20: the recorded events of one execution + the externals that drove it."""
    events: np.ndarray                 # REC_EVENT_DTYPE
    original_externals: np.ndarray     # EXT_EVENT_DTYPE


class MinimizationStats:
    """Minimizer.scala:30-217, replay counter only (`increment_replays` is the reference's own
    'schedules evaluated' counter, RandomScheduler.scala:251-253)."""

    def __init__(self):
        self.total_replays = 0

    def increment_replays(self, n=1):
        self.total_replays += n


@dataclass
class FullyRandom:
    """RandomizationStrategy with a seed (RandomScheduler.scala:635-637); the user-defined filter
    is not supported on the GPU path."""
    seed: int = 0


@dataclass
class SrcDstFIFO:
    """RandomizationStrategy SrcDstFIFO (RandomScheduler.scala:702-909): per (src, dst) FIFO delivery, timers and
    externals at random.  Both of its generators are seeded with the execution's seed (the reference uses the wall
    clock for both)."""
    seed: int = 0


class RandomScheduler:
    """RandomScheduler(schedulerConfig, max_executions, invariant_check_interval, strategy).

    Execution i of explore() is one full run with a fresh `FullyRandom(seed = seed_base + i)`:
    the per-execution-seed shape of RunnerUtils.fuzz (RunnerUtils.scala:75-90), which is the
    parallelisable contract (the carried-RNG mode of one scheduler instance is sequential)."""

    def __init__(self, schedulerConfig: SchedulerConfig, max_executions: int = 1,
                 invariant_check_interval: int = 0, randomizationStrategy: Optional[FullyRandom] = None,
                 seed_base: Optional[int] = None, device: int = 0, p_max: int = 64, specialize: Optional[bool] = None,
                 carried_generator: bool = False):
        """specialize: compile the model's transition table to native code before exploring (demi_model_specialize,
        about a second); None = only when max_executions is large enough to amortise it.
        carried_generator: exactly one reference instance `new RandomScheduler(config, max_executions)` with
        `new FullyRandom(seed_base)`: the generator is NOT reseeded between the executions (reset_all_state only clears the
        pending set, RandomScheduler.scala:575-595, 649-651), lookingFor applies to the first execution only (:586) and the
        executions behind the first violating one are not run (explore() returns there, :257-261).  A sequential chain: for
        comparisons with a JVM run, not for throughput (demi_limits.executions_per_instance)."""
        self.carried_generator = carried_generator
        self.schedulerConfig = schedulerConfig
        self.specialize = (max_executions >= (1 << 18)) if specialize is None else specialize
        self.max_executions = max_executions
        self.invariant_check_interval = invariant_check_interval
        self.seed_base = seed_base if seed_base is not None else (randomizationStrategy or FullyRandom()).seed
        self.strategy = T.STRATEGY_SRC_DST_FIFO if isinstance(randomizationStrategy, SrcDstFIFO) else T.STRATEGY_FULLY_RANDOM
        self.maxMessages = 0x7FFFFFFF            # Int.MaxValue (:54)
        self.p_max = p_max
        self.chunk = 1 << 20                     # executions per device call of explore() (two calls submitted ahead of the one waited for)
        self.stats: Optional[MinimizationStats] = None
        self._model: Optional[Model] = schedulerConfig.model
        self._ctx = _native.Context(device)
        self._loaded_model = False
        self._loaded_trace = None

    def getName(self) -> str:
        return "RandomScheduler"

    def setMaxMessages(self, _maxMessages: int):
        self.maxMessages = _maxMessages

    def setInvariant(self, model: Model):
        """setInvariant (:521-523): on the GPU path the invariant descriptor travels with the model."""
        self._model = model
        self._loaded_model = False

    # -- internals
    def _limits(self, lookingFor: Optional[ViolationFingerprint]) -> T.Limits:
        mm = 0 if self.maxMessages >= 0x7FFFFFFF else self.maxMessages
        return T.Limits(mm, max(0, self.invariant_check_interval), self.p_max,
                        1 if lookingFor is not None else 0, lookingFor.code if lookingFor is not None else 0,
                        1 if self.schedulerConfig.populate_all_actors else 0, self.strategy, 0,
                        self.max_executions if self.carried_generator and self.max_executions > 1 else 0)

    def _prepare(self, trace):
        if self._model is None or self._model.inv_kind == T.INV_NONE:
            # IllegalArgumentException("Must invoke setInvariant before test()") (:244-246)
            raise ValueError("Must invoke setInvariant before test()")
        if not self._loaded_model:
            self._ctx.model_load(self._model.to_struct())
            self._loaded_model = True
            self._loaded_trace = None
            if getattr(self._model, "compiled_only", False):
                self._ctx.model_specialize()     # a wide table (DEMI_MODEL_WIDE) runs only as compiled code: a failure is an error
            elif self.specialize:
                try:
                    self._ctx.model_specialize()
                except _native.DemiError:
                    pass                     # no run-time compiler here: the table interpreter is used
        ev = np.ascontiguousarray(trace, dtype=T.EXT_EVENT_DTYPE)
        key = ev.tobytes()
        if self._loaded_trace != key:
            self._ctx.trace_load(ev)
            self._loaded_trace = key
        return ev

    # -- GPU extension: the full verdict set of max_executions schedules
    def explore_all(self, _trace, _lookingFor: Optional[ViolationFingerprint] = None) -> np.ndarray:
        self._prepare(_trace)
        if self.stats is not None:
            self.stats.increment_replays(self.max_executions)
        v = self._ctx.random_explore(self.max_executions, self._limits(_lookingFor), seed_base=self.seed_base)
        self.last_aborted = int(((v["flags"] & OVF_FLAGS) != 0).sum())      # verdicts without a valid answer (capacity)
        return v

    def explore(self, _trace, _lookingFor: Optional[ViolationFingerprint] = None
                ) -> Optional[Tuple[EventTrace, ViolationFingerprint]]:
        """explore (:234-272): the first violating execution, or None.  The GPU evaluates all
        max_executions schedules; the reference-compatible answer is the lowest index."""
        ev = self._prepare(_trace)
        if self.stats is not None:
            self.stats.increment_replays(self.max_executions)
        if self.carried_generator and self.max_executions > 1:
            # one instance: the chain of executions runs on one lane; the first violating execution ends it
            lim = self._limits(_lookingFor)
            v = self._ctx.random_explore(self.max_executions, lim, seed_base=self.seed_base)
            if (v["flags"] & OVF_FLAGS).any():
                raise CapacityExceeded("an execution of the instance exceeds the engine's capacities")
            hit = np.nonzero(v["flags"] & T.V_VIOLATION)[0]
            if not len(hit):
                return None
            i = int(hit[0])
            v1, rec, ran = self._ctx.random_get_trace_carried(self.seed_base, i, lim)
            assert ran == i and v1.flags & T.V_VIOLATION
            used = ev[:T.verdict_trace_idx(v1.flags)]
            mask = self._model.fp_match_mask if self._model else 0xFFFFFFFF
            return EventTrace(rec, used), ViolationFingerprint(int(v1.fingerprint), mask)
        # only the violating and the aborted executions cross PCIe (16 B each instead of 16 B per schedule).  An execution
        # aborted on a capacity has no valid verdict: it is re-run alone with the largest pending set before any
        # higher index is believed (the reference has no capacities; its answer is the lowest violating index)
        # The executions go to the device in calls of `chunk` (BASELINE config 2's step); while the answer of call k is waited
        # for, calls k + 1 and k + 2 are already submitted in the one context (demi_random_explore_submit / _wait: the tail of
        # every launch is filled by the next one - GpuRandomScheduler.explore in scala/ is this loop), and nothing beyond the
        # calls in flight behind the first violating one is ever submitted.
        lim = self._limits(_lookingFor)
        start, i = 0, None
        self.last_aborted_reruns = 0
        self.last_calls = 0
        ahead = []                                # the calls already submitted, in order: (first execution, executions, ticket)

        def top_up(frm):
            """keep the call at `frm` and the two behind it submitted"""
            nxt = ahead[-1][0] + ahead[-1][1] if ahead else frm
            while len(ahead) < 3 and nxt < self.max_executions:
                n = min(self.chunk, self.max_executions - nxt)
                self.last_calls += 1
                ahead.append((nxt, n, self._ctx.random_explore_submit(n, self._limits(_lookingFor), seed_base=self.seed_base + nxt,
                                                                      flag_mask=T.V_VIOLATION | OVF_FLAGS)))
                nxt += n

        def drain():
            while ahead:
                self._ctx.random_explore_wait(ahead.pop(0)[2], cap=1)
        try:
            while start < self.max_executions and i is None:
                if ahead and ahead[0][0] != start:
                    drain()                       # (a truncated list sent the search back into the middle of a call)
                top_up(start)
                cur = ahead.pop(0)
                span = cur[1]
                hits, n_hits, first = self._ctx.random_explore_wait(cur[2])
                # a truncated list is an arbitrary subset: only the lowest index (computed on the device) is certain then
                cand = [(int(h["index"]), int(h["flags"])) for h in hits] if n_hits <= len(hits) else [(first, None)]
                nxt = start + span
                for idx, flags in cand:
                    if flags is not None and not (flags & OVF_FLAGS):
                        i = start + idx
                        break
                    # aborted (or unknown): decide this execution alone
                    self.last_aborted_reruns += 1
                    big = self._limits(_lookingFor)
                    big.p_max = T.MAX_PENDING
                    v1, _ = self._ctx.random_get_trace(self.seed_base + start + idx, big)
                    if v1.flags & OVF_FLAGS:
                        raise CapacityExceeded("schedule %d exceeds the engine's capacities (flags 0x%x)" % (start + idx, v1.flags & 0xFF))
                    if v1.flags & T.V_VIOLATION:
                        i = start + idx
                        lim = big
                        break
                    if flags is None:
                        nxt = start + idx + 1
                start = nxt
        finally:
            drain()
        if i is None:
            return None
        v, rec = self._ctx.random_get_trace(self.seed_base + i, lim)
        assert v.flags & T.V_VIOLATION
        # checkIfBugFound prunes the externals that were never injected (:160-163)
        used = ev[:T.verdict_trace_idx(v.flags)]
        mask = self._model.fp_match_mask if self._model else 0xFFFFFFFF
        return EventTrace(rec, used), ViolationFingerprint(int(v.fingerprint), mask)

    def _explore_one_call(self, ev, _lookingFor):
        """explore() as rounds 1-5 ran it: ONE device call for all the executions (kept for the comparison in the suite)."""
        lim = self._limits(_lookingFor)
        start, i = 0, None
        self.last_aborted_reruns = 0
        while start < self.max_executions and i is None:
            hits, n_hits, first = self._ctx.random_explore_flagged(self.max_executions - start, lim, T.V_VIOLATION | OVF_FLAGS,
                                                                   seed_base=self.seed_base + start)
            if n_hits == 0:
                return None
            # a truncated list is an arbitrary subset: only the lowest index (computed on the device) is certain then
            cand = [(int(h["index"]), int(h["flags"])) for h in hits] if n_hits <= len(hits) else [(first, None)]
            nxt = self.max_executions
            for idx, flags in cand:
                if flags is not None and not (flags & OVF_FLAGS):
                    i = start + idx
                    break
                # aborted (or unknown): decide this execution alone
                self.last_aborted_reruns += 1
                big = self._limits(_lookingFor)
                big.p_max = T.MAX_PENDING
                v1, _ = self._ctx.random_get_trace(self.seed_base + start + idx, big)
                if v1.flags & OVF_FLAGS:
                    raise CapacityExceeded("schedule %d exceeds the engine's capacities (flags 0x%x)" % (start + idx, v1.flags & 0xFF))
                if v1.flags & T.V_VIOLATION:
                    i = start + idx
                    lim = big
                    break
                if flags is None:
                    nxt = start + idx + 1
            else:
                if n_hits <= len(hits):
                    return None           # every flagged execution was an abort that turned out clean
            start = nxt
        if i is None:
            return None
        v, rec = self._ctx.random_get_trace(self.seed_base + i, lim)
        assert v.flags & T.V_VIOLATION
        # checkIfBugFound prunes the externals that were never injected (:160-163)
        used = ev[:T.verdict_trace_idx(v.flags)]
        mask = self._model.fp_match_mask if self._model else 0xFFFFFFFF
        return EventTrace(rec, used), ViolationFingerprint(int(v.fingerprint), mask)

    def test(self, events, violation_fingerprint: ViolationFingerprint, _stats: Optional[MinimizationStats] = None
             ) -> Optional[EventTrace]:
        """TestOracle.test (:597-612): Some(trace) iff the violation is reproduced within
        max_executions random interleavings of `events`."""
        self.stats = _stats
        r = self.explore(events, violation_fingerprint)
        return r[0] if r is not None else None

    def shutdown(self):
        self._ctx.close()


class STSScheduler:
    """STSScheduler(schedulerConfig, original_trace, allowPeek=false) as DDMin's TestOracle
    (schedulers/STSScheduler.scala:83-310): replays the original execution restricted to a
    subsequence of its external events.  A subsequence is a sequence of indices into
    original_trace.original_externals."""

    def __init__(self, schedulerConfig: SchedulerConfig, original_trace: EventTrace, allowPeek: bool = False,
                 device: int = 0, p_max: int = 64, specialize: bool = False):
        """specialize: compile the model's table to native code first (demi_model_specialize): a minimization consults the
        oracle in many small launches whose time is the serial chain of one replay, which the compiled handlers shorten."""
        if allowPeek:
            raise NotImplementedError("IntervalPeek is not on the GPU path (allowPeek=false, RunnerUtils.scala:332)")
        if schedulerConfig.model is None or schedulerConfig.model.inv_kind == T.INV_NONE:
            raise ValueError("Must invoke setInvariant before test()")
        assert len(original_trace.events) > 0, "assume(!original_trace.isEmpty)"
        self.schedulerConfig = schedulerConfig
        self.original_trace = original_trace
        self.p_max = p_max
        self._ctx = _native.Context(device)
        self._ctx.model_load(schedulerConfig.model.to_struct())
        if getattr(schedulerConfig.model, "compiled_only", False):
            self._ctx.model_specialize()         # a wide table (DEMI_MODEL_WIDE) runs only as compiled code
        elif specialize:
            try:
                self._ctx.model_specialize()
            except _native.DemiError:
                pass                             # no run-time compiler here: the table interpreter is used
        self._ctx.replay_load(original_trace.original_externals, original_trace.events)

    def getName(self) -> str:
        return "STSSchedNoPeek"

    def _limits(self, fp: ViolationFingerprint) -> T.Limits:
        return T.Limits(0, 0, self.p_max, 1, fp.code, 1 if self.schedulerConfig.populate_all_actors else 0, 0,
                        int(self.schedulerConfig.filterKnownAbsents))

    def _masks(self, subseqs) -> np.ndarray:
        from .minification import events_to_masks
        return events_to_masks(subseqs)

    def verdicts(self, subseqs, violationFingerprint: ViolationFingerprint) -> np.ndarray:
        """One verdict per candidate.  A replay aborted on a capacity is no verdict: it is repeated with the largest
        pending set, and if it still does not fit the candidate cannot be decided here (CapacityExceeded) - never
        reported as "does not reproduce"."""
        masks = self._masks(subseqs)
        lim = self._limits(violationFingerprint)
        v = self._ctx.replay_batch(masks, lim)
        bad = np.nonzero(v["flags"] & OVF_FLAGS)[0]
        if len(bad):
            if lim.p_max < T.MAX_PENDING:
                lim.p_max = T.MAX_PENDING
                v[bad] = self._ctx.replay_batch(masks[bad], lim)
            if (v["flags"] & OVF_FLAGS).any():
                raise CapacityExceeded("%d candidate replay(s) exceed the engine's capacities" % int(((v["flags"] & OVF_FLAGS) != 0).sum()))
        return v

    def test_batch(self, subseqs, violationFingerprint: ViolationFingerprint,
                   stats: Optional[MinimizationStats] = None) -> List[bool]:
        """One K2 launch for a whole frontier; element i is True iff subseqs[i] reproduces the
        violation (test() would return Some(trace)).  With a process group the candidates are dealt
        round-robin to the ranks and the verdict bits all-gathered."""
        from .distributed import sharded_map
        if stats is not None:
            stats.increment_replays(len(subseqs))
        return sharded_map(list(subseqs), lambda part: [bool(f & T.V_VIOLATION) for f in
                                                        self.verdicts(part, violationFingerprint)["flags"]])

    def test(self, subseq, violationFingerprint: ViolationFingerprint, stats: Optional[MinimizationStats] = None):
        """TestOracle.test: Some(verdict) iff the subsequence reproduces the violation."""
        assert len(subseq) > 0, "assume(!subseq.isEmpty)"
        if stats is not None:
            stats.increment_replays()
        v = self.verdicts([tuple(subseq)], violationFingerprint)[0]
        return v if (int(v["flags"]) & T.V_VIOLATION) else None

    def executed_trace(self, subseq, violationFingerprint: ViolationFingerprint) -> Optional[EventTrace]:
        """test() with the EventTrace it returns on success (:286-292), re-based on `subseq`
        (trace.setOriginalExternalEvents(mcs), RunnerUtils.scala:698): DDMin.verify_mcs's verified_mcs,
        the input of internal-event minimization."""
        from .internal_minimization import executed_trace
        subseq = tuple(subseq)
        lim = self._limits(violationFingerprint)
        v, kept = self._ctx.replay_get_kept(len(self.original_trace.events), 0xFFFFFFFF, lim, mask=self._masks([subseq])[0])
        if (int(v.flags) & OVF_FLAGS) and lim.p_max < T.MAX_PENDING:
            lim.p_max = T.MAX_PENDING
            v, kept = self._ctx.replay_get_kept(len(self.original_trace.events), 0xFFFFFFFF, lim, mask=self._masks([subseq])[0])
        if int(v.flags) & OVF_FLAGS:
            raise CapacityExceeded("the replay exceeds the engine's capacities")
        if not (int(v.flags) & T.V_VIOLATION):
            return None
        return executed_trace(self.original_trace, kept, subseq=subseq)

    def shutdown(self):
        self._ctx.close()


class CapacityExceeded(RuntimeError):
    """An execution needed more than the engine's largest capacities (pending set of DEMI_MAX_PENDING messages, the
    timer queues, DEMI_FX_CAP effects per delivery): its verdict is invalid (DEMI_V_PENDING_OVF / DEMI_V_QUEUE_OVF) and the
    reference, which has no such capacities, must decide it (JVM fallback)."""


OVF_FLAGS = T.V_PENDING_OVF | T.V_QUEUE_OVF


class ReplayException(Exception):
    """ReplayScheduler.scala:24-25: the recorded execution could not be followed exactly."""


class ReplayScheduler:
    """Strict replay of a full EventTrace (schedulers/ReplayScheduler.scala:71-140, 256-342), used by
    RunnerUtils.fuzz to validate that a found violation is deterministic before it is kept
    (RunnerUtils.scala:101-128).  On the GPU path a strict replay is K2 with the full mask: every
    expected delivery must be present (no DEMI_V_DIVERGED), otherwise ReplayException."""

    def __init__(self, schedulerConfig: SchedulerConfig, device: int = 0, p_max: int = 128):
        if schedulerConfig.model is None or schedulerConfig.model.inv_kind == T.INV_NONE:
            raise ValueError("Must invoke setInvariant before test()")
        self.schedulerConfig = schedulerConfig
        self.p_max = p_max
        self._ctx = _native.Context(device)
        self._ctx.model_load(schedulerConfig.model.to_struct())
        if getattr(schedulerConfig.model, "compiled_only", False):
            self._ctx.model_specialize()         # a wide table (DEMI_MODEL_WIDE) runs only as compiled code

    def replay(self, trace: EventTrace, expected: Optional[ViolationFingerprint] = None):
        """Returns the replay's verdict row; raises ReplayException on divergence."""
        n = len(trace.original_externals)
        self._ctx.replay_load(trace.original_externals, trace.events)
        mask = np.zeros((1, 4), dtype=np.uint64)
        for e in range(n):
            mask[0, e >> 6] |= np.uint64(1) << np.uint64(e & 63)
        lim = T.Limits(0, 0, self.p_max, 1, expected.code if expected is not None else 0,
                       1 if self.schedulerConfig.populate_all_actors else 0, 0, int(self.schedulerConfig.filterKnownAbsents))
        v = self._ctx.replay_batch(mask, lim)[0]
        flags = int(v["flags"])
        if flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF):
            raise ReplayException("capacity exceeded during replay")
        if flags & T.V_DIVERGED:
            raise ReplayException("expected message was not pending: the recorded execution is not reproducible")
        return v

    def shutdown(self):
        self._ctx.close()
