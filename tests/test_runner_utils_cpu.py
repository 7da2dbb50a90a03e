"""CPU suite, part 11: the fuzz driver's control flow (RunnerUtils.fuzz) with stand-in schedulers."""
import numpy as np

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.runner_utils import fuzz
from demi_amd.schedulers import EventTrace, ReplayException, SchedulerConfig, ViolationFingerprint


def test_fuzz_driver_control_flow(oracle):
    model, events, lim = raft5_config2()
    v = oracle.random_explore(model, events, 512, seed_base=SEED_BASE, limits=lim, n_threads=4)
    hit = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    log = []

    class StandInScheduler:                      # RandomScheduler's surface, executions by the oracle (tests only)
        def __init__(self, cfg, max_executions, interval, randomizationStrategy=None):
            self.n, self.interval, self.maxm = max_executions, interval, 0

        def setMaxMessages(self, m):
            self.maxm = m

        def explore(self, trace):
            log.append(("explore", len(trace)))
            if len(trace) < len(events):             # the first generated tests are too short to fail
                return None
            vv, rec, _ = oracle.random_execute(model, trace, SEED_BASE + hit, T.Limits(self.maxm, self.interval, 64, 0, 0, 0))
            return EventTrace(rec, trace[:T.verdict_trace_idx(vv.flags)]), ViolationFingerprint(vv.fingerprint)

        def shutdown(self):
            log.append("shutdown")

    class FlakyReplayer:
        calls = 0

        def replay(self, trace, expected):
            FlakyReplayer.calls += 1
            if FlakyReplayer.calls == 1:
                raise ReplayException("not deterministic")
            return {"flags": T.V_VIOLATION}

        def shutdown(self):
            log.append("replayer shutdown")

    tests = [events[:10], events[:20], events, events]
    res = fuzz(lambda i: tests[i], SchedulerConfig(model=model), validate_replay=FlakyReplayer, maxMessages=200,
               executions_per_test=1, max_tests=4, scheduler_ctor=StandInScheduler)
    assert res is not None
    trace, violation, initial, filtered = res
    assert [e for e in log if e != "shutdown" and e != "replayer shutdown"] == [("explore", 10), ("explore", 20), ("explore", 50), ("explore", 50)]
    assert FlakyReplayer.calls == 2 and log.count("shutdown") == 4 and log.count("replayer shutdown") == 2
    assert len(initial) == 1 + int((trace.events["kind"] == T.REC_MSG_EVENT).sum())
    assert 0 < len(filtered) <= len(initial) and set(filtered["key"]) <= set(initial["key"])
    # every kept event is in the causal past of a last receive at an affected node, so the root is kept
    assert int(filtered[0]["key"]) == T.DPOR_ROOT_KEY
    # a filter that rejects every violation: the driver gives up after max_tests
    assert fuzz(lambda i: events, SchedulerConfig(model=model), violationWereLookingFor=lambda f: False,
                executions_per_test=1, max_tests=2, scheduler_ctor=StandInScheduler) is None
