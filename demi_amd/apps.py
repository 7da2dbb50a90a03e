"""The synthetic applications and frozen inputs restating BASELINE.json's configs.

The reference's applications (akka-raft, Spark: NetSys/demi-applications) are not part of
/root/reference; these table-encoded stand-ins are defined here and frozen under tests/golden/.
"""
from . import types as T
from .fuzzer import events_to_array, raft_trace
from .model import raft_model

SEED_BASE = 0x5EED0000     # schedule i uses java.util.Random(SEED_BASE + i)
TRACE_SEED = 0xDE31        # the frozen external traces are generated from this seed


def raft5_config2():
    """BASELINE config 2: akka-raft-like, 5 actors, 50-event trace, maxMessages 200, invariant
    every 30 deliveries (RunnerUtils.scala:65), pending capacity 64."""
    model = raft_model(5)
    events = events_to_array(raft_trace(5, 50, TRACE_SEED))
    limits = T.Limits(200, 30, 64, 0, 0, 0)
    return model, events, limits


def raft3_config1():
    """BASELINE config 1 restated: 3 actors, 20 events, 100 schedules (the reference's own
    CPU-runnable case; runs on the oracle, never reported as "DEMi JVM")."""
    model = raft_model(3)
    events = events_to_array(raft_trace(3, 20, TRACE_SEED))
    limits = T.Limits(200, 30, 64, 0, 0, 0)
    return model, events, limits


def raft5_config4(n_events=200):
    """BASELINE config 4: DDMin of a failing Raft-5 execution with 200 external events.  The failing
    execution is the first violating schedule of the frozen 200-event trace (found by K1, recorded
    on the GPU); callers minimise its externals with DDMin over the STSSched (no-peek) oracle.
    The invariant is only checked at the end (interval 0) so the failing execution consumes the
    whole trace."""
    model = raft_model(5)
    events = events_to_array(raft_trace(5, n_events, TRACE_SEED + 4))
    limits = T.Limits(4000, 0, 128, 0, 0, 0)
    return model, events, limits


def raft5_config3(n_sends=5):
    """BASELINE config 3: DPORwHeuristics, depth_bound 30, Start x 5 + Send x k (DPOR supports only
    Start / Send / WaitQuiescence), stopIfViolationFound=false, trackHistory=true."""
    from .fuzzer import send, start
    from .model import M_BOOTSTRAP
    model = raft_model(5)
    events = events_to_array([start(a) for a in range(5)] + [send(a % 5, M_BOOTSTRAP) for a in range(n_sends)])
    return model, events, 30


def shuffle8_config5():
    """BASELINE config 5: the Spark-like 2-stage shuffle job (8 actors, 3 actor classes) for bounded
    DPOR exploration (Start / Send only) and for fuzzing.  Spark's actors are not in the reference:
    synthetic only, parity vs real Spark unpinned."""
    from .fuzzer import send, start, wait_quiescence
    from .model import SH_SPECULATE, SH_SUBMIT, shuffle_model
    model = shuffle_model()
    dpor_events = events_to_array([start(a) for a in range(8)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1)])
    fuzz_events = events_to_array([start(a) for a in range(8)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1),
                                                                  send(0, SH_SPECULATE, 3), wait_quiescence()])
    return model, dpor_events, fuzz_events, T.Limits(400, 0, 64, 0, 0, 0)


def shuffle8_config5_large(jobs=3):
    """BASELINE config 5 at a size worth sharding over 8 GPUs ("bounded exhaustive search across 8 GPUs"): the same 8-actor
    shuffle application as a PIPELINE of `jobs` shuffle jobs (model.shuffle_model(jobs): the reduce stage reports back and the
    driver launches the next job itself), explored by DPORwHeuristics with depth_bound 40 until a budget of interleavings is
    spent.  More Submit / Speculate externals do not enlarge the one-job exploration (1 685 -> 1 787 interleavings with five
    Speculates: every racing pair is flipped once, ExploredTacker, and the job's causal structure fixes the pairs); chaining
    jobs does, because the next job's messages descend from whichever delivery completed the previous one and are new DPOR nodes
    for each way it can end: 2 jobs exhaust after 722 376 interleavings, 3 jobs are not exhausted at 2^20 (the bench line's
    budget; the backtrack queue still holds millions of points).  Returns (model, externals, depth_bound, budget)."""
    from .fuzzer import send, start
    from .model import SH_SPECULATE, SH_SUBMIT, shuffle_model
    model = shuffle_model(jobs=jobs)
    dpor_events = events_to_array([start(a) for a in range(8)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1)])
    return model, dpor_events, 40, 1 << 20
