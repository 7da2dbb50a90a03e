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


def raft5_dpor_config3():
    """BASELINE config 3 as bench.py times it from round 6 on: DPORwHeuristics over raft5-synth - the SAME rows as config 2's
    table, seeded bug included - depth_bound 30, Start x 5 + Send(Bootstrap) x 5, trackHistory, stopIfViolationFound = false,
    explored until the backtrack queue is empty.  Two things differ from raft5_config3() below, both so that the exploration
    FINDS the seeded bug (rounds 1-5 explored 60 332 interleavings and the violating set was empty):
      * prioritizePendingUponDivergence = true (DPORwHeuristics.scala:65-68, 537-550; what RunnerUtils.editDistanceDporDDMin
        constructs its DPOR with, RunnerUtils.scala:824-827).  Without it a flipped pair is undone on the spot: the first
        expected event that is not pending - a reply to the message the flip postponed - makes getMatchingMessage fall back to
        getPendingEvent (:594-625), whose first queue holds exactly the postponed message.
      * nodes 3 and 4 never campaign (election budgets 1, 1, 1, 0, 0: initial field values, no row changes).  The invariant is
        checked on the FINAL state (notify_quiescence, :877-902); with five campaigning nodes every execution ends after the
        last node's election in a fresh term, which hides an earlier term's two leaders.
    297 396 interleavings in ROUNDS order, 7 237 of them violating (oracle; the counts of the reference's order are in
    tests/golden/dpor_config3_bug_reference_order.json).  Returns (model, externals, DporParams)."""
    from .fuzzer import send, start
    from .model import M_BOOTSTRAP
    model = raft_model(5, election_budget=(1, 1, 1, 0, 0))
    events = events_to_array([start(a) for a in range(5)] + [send(a, M_BOOTSTRAP) for a in range(5)])
    return model, events, T.DporParams(30, 0, 0, 0, 64, 4096, 1)


def raft5_config3(n_sends=5):
    """BASELINE config 3 as rounds 1-5 timed it (kept: a parity workload of the suites, 60 332 interleavings in the reference's
    order, none violating - see raft5_dpor_config3): DPORwHeuristics, depth_bound 30, Start x 5 + Send x k (DPOR supports only
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


def shuffle8_dpor_config5(jobs=3):
    """BASELINE config 5 as bench.py times it from round 6 on: shuffle8_config5_large's pipeline with the pipeline's second
    seeded bug (model.shuffle_model(early_cleanup=True): reducer 7 frees its map output as soon as it has finished) and
    prioritizePendingUponDivergence = true (see raft5_dpor_config3).  The duplicate-MapDone bug alone needs the FIRST job's
    straggler detector - the shallow end of a deepest-first search: with two jobs the first violating interleaving is number
    1 652 292, with three it lies beyond any budget run here - so the 2^20-interleaving record of rounds 4-5 had an empty
    violating set.  With the reduce-phase race the search reports its first violation within a few hundred interleavings.
    Returns (model, externals, DporParams, budget)."""
    from .fuzzer import send, start
    from .model import SH_SPECULATE, SH_SUBMIT, shuffle_model
    model = shuffle_model(jobs=jobs, early_cleanup=True)
    dpor_events = events_to_array([start(a) for a in range(8)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1)])
    return model, dpor_events, T.DporParams(40, 0, 0, 0, 64, 4096, 1), 1 << 20


# ---- more than 8 actors: the BIG layout of include/demi_gpu.h (4-bit receiver / 5-bit sender fields, deadLetters = 31; a wide,
# compiled table).  The same applications with more nodes - the reference puts no bound on actor names (ExternalEvents.scala:62-91).
def raft11_config2(n_events=50):
    """Config 2's shape with 11 raft nodes (majority 6; four of them ever campaign - the seeded bug needs an odd cluster: two
    leaders in a term take 2 x majority votes, one more than there are nodes): 50-event fuzz trace, maxMessages 1000, invariant
    every 30 deliveries, pending capacity 128.  About 1 % of the schedules violate (oracle)."""
    model = raft_model(11, election_budget=[1] * 4 + [0] * 7)
    events = events_to_array(raft_trace(11, n_events, TRACE_SEED + 11))
    return model, events, T.Limits(1000, 30, 128, 0, 0, 0)


def raft11_dpor(campaigners=2):
    """DPOR over 11 raft nodes of which `campaigners` ever campaign (see raft5_dpor_config3): Start x 11 + Bootstrap x 11."""
    from .fuzzer import send, start
    from .model import M_BOOTSTRAP
    model = raft_model(11, election_budget=[1] * campaigners + [0] * (11 - campaigners))
    events = events_to_array([start(a) for a in range(11)] + [send(a, M_BOOTSTRAP) for a in range(11)])
    return model, events, T.DporParams(30, 0, 0, 0, 128, 4096, 1)


def shuffle12_config5(jobs=1, early_cleanup=False):
    """Config 5's application with nine workers (12 actors, three classes): (model, DPOR externals, fuzz externals, Limits,
    DporParams)."""
    from .fuzzer import send, start, wait_quiescence
    from .model import SH_SPECULATE, SH_SUBMIT, shuffle_model
    model = shuffle_model(jobs=jobs, early_cleanup=early_cleanup, n_workers=9)
    dpor_events = events_to_array([start(a) for a in range(12)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1)])
    fuzz_events = events_to_array([start(a) for a in range(12)] + [send(0, SH_SUBMIT), send(0, SH_SPECULATE, 1),
                                                                   send(0, SH_SPECULATE, 7), wait_quiescence()])
    return model, dpor_events, fuzz_events, T.Limits(600, 0, 128, 0, 0, 0), T.DporParams(40, 0, 0, 0, 128, 4096, 1)
