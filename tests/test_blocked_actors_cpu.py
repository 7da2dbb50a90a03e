"""CPU suite: crashed actors and Util.find_non_blocked_message.

DEMI_OP_CRASH models a receive that throws (Instrumenter.actorCrashed, Instrumenter.scala:184-199): the actor joins
blockedActors, and RandomScheduler.schedule_new_message then draws through Util.find_non_blocked_message
(Util.scala:470-489) - a drawn message whose receiver is blocked is set aside, the draw repeated, and the rejected ones
re-appended in draw order, which permutes RandomizedHashSet.arr.  The oracle's restatement is checked against a literal
transliteration of those two Scala pieces driven by the JDK generator."""
import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.fuzzer import JavaRandom, events_to_array, send, start, wait_quiescence
from demi_amd.model import Asm, build_model

MSGS = [("Boom", T.MSG_EXTERNAL), ("Note", T.MSG_EXTERNAL)]


def sink_model():
    """Note: remember the payload.  Boom: the receive throws."""
    h = {(0, "Boom"): Asm().add(M.F[1], M.F[1], 1).crash().add(M.F[1], M.F[1], 100),   # the row after the crash never runs
         (0, "Note"): Asm().mov(M.F[0], M.P0).add(M.F[2], M.F[2], 1)}
    return build_model("sink", 3, MSGS, h, [[0] * 8] * 3, (T.INV_NEVER, 0, 200, 0))


class RandomizedHashSet:
    """schedulers/Util.scala:110-185, literally: arr + swap-with-last removal + java.util.Random."""

    def __init__(self, seed):
        self.arr, self.rand = [], JavaRandom(seed)

    def insert(self, e):
        self.arr.append(e)

    def isEmpty(self):
        return not self.arr

    def removeRandomElement(self):
        idx = self.rand.next_int(len(self.arr))
        v = self.arr[idx]
        self.arr[idx] = self.arr[-1]
        self.arr.pop()
        return v


def find_non_blocked_message(blockedActors, collection, getActor):
    """schedulers/Util.scala:470-489, literally."""
    if collection.isEmpty():
        return None
    blocked = []
    e = collection.removeRandomElement()
    while getActor(e) in blockedActors:
        blocked.append(e)
        if collection.isEmpty():
            for b in blocked:
                collection.insert(b)
            return None
        e = collection.removeRandomElement()
    for b in blocked:
        collection.insert(b)
    return e


def literal_execution(sends, seed, restart_after=None):
    """One RandomScheduler execution of the sink model with the Scala pieces above: `sends` = (rcv, type, p0) in Send order,
    all flushed at the first scheduling step.  Returns the delivery sequence."""
    pending = RandomizedHashSet(seed)
    for s in sends:
        pending.insert(s)
    blocked, out = set(), []
    while True:
        e = find_non_blocked_message(blocked, pending, lambda x: x[0])
        if e is None:
            return out, pending.arr
        out.append(e)
        if e[1] == 0:                     # Boom: actorCrashed
            blocked.add(e[0])


@pytest.mark.parametrize("seed", [1, 2, 3, 7, 0x5EED0000, 12345])
def test_find_non_blocked_message_redraws_and_reappends_like_the_scala(oracle, seed):
    model = sink_model()
    sends = [(1, 0, 0)] + [(0, 1, 10 + i) for i in range(6)] + [(1, 1, 20 + i) for i in range(4)] + [(2, 1, 30 + i) for i in range(3)]
    ev = events_to_array([start(0), start(1), start(2)] + [send(r, ty, p0) for r, ty, p0 in sends])
    lim = T.Limits(0, 0, 64, 0, 0, 0)
    v, rec, states = oracle.random_execute(model, ev, seed, lim)
    got = [(int(e["rcv"]), int(e["msg_type"]), int(e["p0"])) for e in rec if e["kind"] == T.REC_MSG_EVENT]
    want, left = literal_execution(sends, seed)
    assert got == want
    # everything addressed to the crashed actor after its Boom stays pending; the row after the crash did not run
    boom = want.index((1, 0, 0))
    assert all(r != 1 for r, _, _ in want[boom + 1:]) and len(left) == sum(1 for s in sends if s[0] == 1) - 1 - sum(1 for r, _, _ in want[:boom] if r == 1)
    assert (int(states[1]) >> 8) & 0xFF == 1


def test_start_unblocks_a_crashed_actor(oracle):
    """trigger_start: "If actor was previously killed, allow scheduler to send messages to it again" (EventOrchestrator.scala:224-227)."""
    model = sink_model()
    ev = events_to_array([start(0), start(1), send(1, 0, 0), send(1, 1, 5), send(1, 1, 6), send(0, 1, 7), wait_quiescence(),
                          start(1), wait_quiescence()])
    for seed in range(20):
        v, rec, states = oracle.random_execute(model, ev, seed, T.Limits(0, 0, 64, 0, 0, 0))
        kinds = rec["kind"].tolist()
        q = kinds.index(T.REC_QUIESCENCE)
        first = [(int(e["rcv"]), int(e["p0"])) for e in rec[:q] if e["kind"] == T.REC_MSG_EVENT]
        second = [(int(e["rcv"]), int(e["p0"])) for e in rec[q:] if e["kind"] == T.REC_MSG_EVENT]
        boom_at = first.index((1, 0))
        held = 2 - sum(1 for r, p in first[:boom_at] if r == 1)
        assert sum(1 for r, p in first[boom_at + 1:] if r == 1) == 0 and len(second) == held      # delivered after the re-Start
        assert T.verdict_deliveries(int(v.flags)) == 4


def armed_model():
    """Arm / Disarm set and clear a flag; a Note received while armed makes the receive throw."""
    msgs = [("Arm", T.MSG_EXTERNAL), ("Disarm", T.MSG_EXTERNAL), ("Note", T.MSG_EXTERNAL)]
    h = {(0, "Arm"): Asm().mov(M.F[3], 1), (0, "Disarm"): Asm().mov(M.F[3], 0),
         (0, "Note"): Asm().add(M.F[2], M.F[2], 1).skipz(M.F[3], "ok").crash().label("ok").mov(M.F[0], M.P0)}
    return build_model("armed", 2, msgs, h, [[0] * 8] * 2, (T.INV_NEVER, 2, 2, 0))      # "violation": two Notes were received


def test_crash_in_replay_and_dpor(oracle):
    """STSSched: an expected delivery to a blocked actor is not pending (STSScheduler.scala:392-402) and is ignored;
    DPOR: blocked receivers are skipped by getPendingEvent / getMatchingMessage (DPORwHeuristics.scala:455, 478, 518)."""
    model = armed_model()
    ev = events_to_array([start(0), start(1), send(1, 0), send(1, 1), send(1, 2, 5), send(1, 2, 6)])
    lim = T.Limits(0, 0, 64, 0, 0, 0)
    for k in range(400):               # an original execution Arm, Disarm, Note, Note: both Notes are received
        seed = (k * 0x9E3779B97F4A7C15 + 12345) & 0xFFFFFFFFFFFF    # (consecutive small seeds all start with the same draw)
        v, rec, _ = oracle.random_execute(model, ev, seed, lim)
        order = [int(e["msg_type"]) for e in rec if e["kind"] == T.REC_MSG_EVENT]
        if order == [0, 1, 2, 2]:
            break
    else:
        pytest.skip("no such interleaving among the seeds")
    assert int(v.flags) & T.V_VIOLATION
    target = T.Limits(0, 0, 64, 1, v.fingerprint, 0)
    keep_all, no_disarm = 0b111111, 0b110111
    r = oracle.sts_replay_batch(model, ev, rec, np.array([[keep_all, 0, 0, 0], [no_disarm, 0, 0, 0]], dtype=np.uint64), target)
    assert int(r[0]["flags"]) & T.V_VIOLATION and not (int(r[0]["flags"]) & T.V_DIVERGED) and T.verdict_deliveries(int(r[0]["flags"])) == 4
    # without the Disarm the first Note crashes actor 1: the second expected Note is pending but its receiver is blocked
    assert not (int(r[1]["flags"]) & T.V_VIOLATION) and (int(r[1]["flags"]) & T.V_DIVERGED) and T.verdict_deliveries(int(r[1]["flags"])) == 2
    # DPOR, pinned divergent order (externals to actor 1 in FIFO order): Arm, Disarm, Note, Note - nothing blocked;
    # without the Disarm: Arm, Note (crash), and the last Note is never delivered
    for events, n_deliveries in (([start(0), start(1), send(1, 0), send(1, 1), send(1, 2, 5), send(1, 2, 6)], 4),
                                 ([start(0), start(1), send(1, 0), send(1, 2, 5), send(1, 2, 6)], 2)):
        dv, traces, pairs = oracle.dpor_batch(model, events_to_array(events), [np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)],
                                              T.DporParams(0, 0, 0, 0, 64, 256))
        assert len(traces[0]) == 1 + n_deliveries


# ------------------------------------------------------------------ application randomness (DEMI_OP_RND)
def test_rnd_draws_from_a_second_generator_that_restarts_at_seed_zero(oracle):
    """Instrumenter().seededRandom = scala.util.Random(0), recreated with every ActorSystem (Instrumenter.scala:212, 226-229,
    570): the values an actor draws are java.util.Random(0).nextInt(bound) in delivery order, the same in every execution
    whatever the scheduler's own seed, and independent of the scheduler's generator."""
    msgs = [("Draw", T.MSG_EXTERNAL)]
    h = {(0, "Draw"): Asm().rnd(M.T0, M.P0).shl(M.F[1], M.F[1], 0).add(M.F[2], M.F[2], 1).mov(M.F[0], M.T0).rnd(M.F[3], 0)}
    model = build_model("dice", 1, msgs, h, [[0] * 8], (T.INV_NEVER, 4, 9, 0))
    bounds = [10, 5, 7, 255, 1, 128, 3]
    ev = events_to_array([start(0)] + [send(0, 0, b) for b in bounds])
    for seed in (1, 99, 0x5EED0000):
        v, rec, states = oracle.random_execute(model, ev, seed, T.Limits(0, 0, 64, 0, 0, 0))
        order = [int(e["p0"]) for e in rec if e["kind"] == T.REC_MSG_EVENT]
        assert sorted(order) == sorted(bounds)
        jr = JavaRandom(0)
        drawn = [jr.next_int(b) for b in order]           # one draw per delivery, in delivery order (bound 0 draws nothing)
        assert int(states[0]) & 0xFF == drawn[-1] and (int(states[0]) >> 24) & 0xFF == 0
    # the JDK known answers: new Random(0).nextInt(5) x 10 = 0,3,4,2,0,3,1,1,4,4
    ev5 = events_to_array([start(0), send(0, 0, 5)])
    got = []
    model_acc = build_model("dice5", 1, msgs, {(0, "Draw"): Asm().rnd(M.F[0], 5).rnd(M.F[1], 5).rnd(M.F[2], 5).rnd(M.F[3], 5).rnd(M.F[4], 5)
                                               .rnd(M.F[5], 5).rnd(M.F[6], 5).rnd(M.F[7], 5)}, [[0] * 8], (T.INV_NEVER, 0, 9, 0))
    v, rec, states = oracle.random_execute(model_acc, ev5, 7, T.Limits(0, 0, 64, 0, 0, 0))
    assert [(int(states[0]) >> (8 * i)) & 0xFF for i in range(8)] == [0, 3, 4, 2, 0, 3, 1, 1]
