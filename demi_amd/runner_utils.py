"""The drivers either side of the hot path: RunnerUtils.fuzz (RunnerUtils.scala:62-147) and the minimization pipeline
of RunnerUtils.runTheGamut (:165-380, the stages that exist on the GPU path).  Host orchestration only: every execution,
replay and interleaving is a kernel launch through the schedulers of this package.
"""
from typing import Callable, Optional, Sequence, Tuple

import numpy as np

from . import types as T
from .incremental_ddmin import dpor_initial_trace
from .provenance import pruneConcurrentEvents
from .schedulers import (EventTrace, FullyRandom, MinimizationStats, RandomScheduler, ReplayException, ReplayScheduler,
                         SchedulerConfig, ViolationFingerprint)


def fuzz(generateFuzzTest: Callable[[int], np.ndarray], schedulerConfig: SchedulerConfig,
         validate_replay: Optional[Callable[[], ReplayScheduler]] = None, invariant_check_interval: int = 30,
         maxMessages: Optional[int] = None, randomizationStrategyCtor: Callable[[], object] = FullyRandom,
         computeProvenance: bool = True, violationWereLookingFor: Callable[[ViolationFingerprint], bool] = lambda f: True,
         executions_per_test: int = 4096, max_tests: int = 64, scheduler_ctor=RandomScheduler,
         provenance_device: Optional[int] = -1
         ) -> Optional[Tuple[EventTrace, ViolationFingerprint, np.ndarray, np.ndarray]]:
    """RunnerUtils.fuzz: generate a fuzz test, explore it, keep the first violation that (optionally) replays
    deterministically, then prune the deliveries outside the violation's provenance.

    The reference runs ONE random execution per generated test (`new RandomScheduler(config, 1, interval, strategy)`);
    a launch evaluates `executions_per_test` seeded interleavings of the same test and explore() reports the first
    violating one.  generateFuzzTest(i) is Fuzzer.generateFuzzTest for the i-th attempt (fuzzer.generate_fuzz_test /
    raft_trace with a seed derived from i).  Returns (trace, violation, initialTrace, filtered) — the depGraph of the
    reference is implicit in the causal-path keys of initialTrace — or None after max_tests tests without a violation
    (the reference loops forever).  provenance_device: the GPU ProvenanceTracker runs on (demi_provenance_prune); -1 = device
    0 when the executions ran on the GPU scheduler (the default scheduler_ctor), the host class (demi_amd/provenance.py)
    for a caller-supplied scheduler; None = always the host class."""
    if provenance_device == -1:
        provenance_device = 0 if scheduler_ctor is RandomScheduler else None
    for attempt in range(max_tests):
        fuzzTest = np.ascontiguousarray(generateFuzzTest(attempt), dtype=T.EXT_EVENT_DTYPE)
        sched = scheduler_ctor(schedulerConfig, executions_per_test, invariant_check_interval,
                               randomizationStrategy=randomizationStrategyCtor())
        if maxMessages is not None:
            sched.setMaxMessages(maxMessages)
        try:
            found = sched.explore(fuzzTest)
        finally:
            sched.shutdown()
        if found is None:
            continue
        trace, violation = found
        if not violationWereLookingFor(violation):
            continue
        if validate_replay is not None:
            replayer = validate_replay()
            deterministic = True
            try:
                v = replayer.replay(trace, violation)
                if not (int(v["flags"]) & T.V_VIOLATION):          # replayer.violationAtEnd.isEmpty
                    deterministic = False
            except ReplayException:
                deterministic = False
            finally:
                replayer.shutdown()
            if not deterministic:
                continue
        initialTrace = dpor_initial_trace(trace, schedulerConfig.model)
        if not computeProvenance:
            filtered = initialTrace[:0]
        elif provenance_device is not None:
            from . import _native
            pctx = _native.Context(provenance_device)
            try:
                filtered = pruneConcurrentEvents(initialTrace, violation.affectedNodes(), ctx=pctx)
            finally:
                pctx.close()
        else:
            filtered = pruneConcurrentEvents(initialTrace, violation.affectedNodes())
        return trace, violation, initialTrace, filtered
    return None


def run_the_gamut(schedulerConfig: SchedulerConfig, trace: EventTrace, violation: ViolationFingerprint,
                  stages: Sequence[str] = ("DDMin", "IntMin"), device: int = 0, p_max: int = 64):
    """The stages of RunnerUtils.runTheGamut (:165-380) that run on the GPU path, in the reference's order:
    stsSchedDDMin (external events), then minimizeInternals with LeftToRightOneAtATime.  Returns a dict with the MCS
    (indices into trace.original_externals), the verified MCS execution, the internally minimized execution and the
    replay counts of each stage."""
    from .internal_minimization import countMsgEvents, minimizeInternals
    from .minification import stsSchedDDMin
    from .schedulers import STSScheduler
    out = {"original_externals": len(trace.original_externals), "original_deliveries": countMsgEvents(trace)}
    cur_trace, mcs = trace, tuple(range(len(trace.original_externals)))
    if "DDMin" in stages:
        sts = STSScheduler(schedulerConfig, trace, device=device, p_max=p_max)
        try:
            stats = MinimizationStats()
            mcs, ddmin, _ = stsSchedDDMin(sts, trace.original_externals, violation, stats=stats)
            verified = sts.executed_trace(mcs, violation)
        finally:
            sts.shutdown()
        out.update(mcs=mcs, ddmin_replays=stats.total_replays, verified_mcs=verified)
        if verified is not None:
            cur_trace = verified
    if "IntMin" in stages and cur_trace is not None:
        stats = MinimizationStats()
        _, minimized = minimizeInternals(schedulerConfig, cur_trace.original_externals, cur_trace, violation, stats=stats,
                                         device=device, p_max=p_max)
        out.update(intmin_replays=stats.total_replays, minimized=minimized, minimized_deliveries=countMsgEvents(minimized))
    return out
