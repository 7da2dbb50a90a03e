"""CPU suite, part 3: delta debugging host logic (EventDag atoms, split_list, ddmin2, speculation)
and the STS replay restatement in the oracle."""
import itertools
import os
import subprocess
import sys

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import (events_to_array, kill, partition, send, start, unpartition, wait_quiescence)
from demi_amd.minification import (DDMin, EventDagView, SpeculativeDDMin, UnmodifiedEventDag, events_to_mask, split_list,
                                   stsSchedDDMin)
from demi_amd.schedulers import MinimizationStats, ViolationFingerprint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_list_matches_reference_chunking():
    assert split_list(list(range(5)), 2) == [[0, 1, 2], [3, 4]]
    assert split_list(list(range(4)), 2) == [[0, 1], [2, 3]]
    assert split_list([7], 2) == [[7], []]
    assert split_list(list(range(7)), 3) == [[0, 1, 2], [3, 4], [5, 6]]
    with pytest.raises(ValueError):
        split_list([1], 0)


def test_atoms_pair_kill_with_start_and_unpartition_with_partition():
    ev = events_to_array([start(0), start(1), send(0, M.M_BOOTSTRAP), partition(0, 1), kill(1), send(0, M.M_CLIENT, 1),
                          unpartition(0, 1), start(1), partition(1, 0), wait_quiescence()])
    dag = UnmodifiedEventDag(ev)
    assert dag.get_atomic_events() == [(0,), (1, 4), (2,), (3, 6), (5,), (7,), (8,), (9,)]
    view = dag.remove_events([(1, 4), (5,)])
    assert view.get_all_events() == (0, 2, 3, 6, 7, 8, 9)
    assert view.get_atomic_events() == [(0,), (2,), (3, 6), (7,), (8,), (9,)]
    u = view.remove_events([(3, 6)]).union(EventDagView(dag, (3, 6)))
    assert u.get_all_events() == view.get_all_events()      # union re-sorts by original index
    with pytest.raises(RuntimeError):
        UnmodifiedEventDag(events_to_array([kill(0)])).get_atomic_events()
    with pytest.raises(RuntimeError):
        UnmodifiedEventDag(events_to_array([unpartition(0, 1)])).get_atomic_events()
    with pytest.raises(AssertionError):      # two Starts of one actor without a Kill: the reference's assume() fails
        UnmodifiedEventDag(events_to_array([start(0), start(0)])).get_atomic_events()
    explicit = UnmodifiedEventDag(events_to_array([start(0), send(0, M.M_BOOTSTRAP), send(0, M.M_CLIENT)]))
    explicit.conjoinAtoms(1, 2)
    assert explicit.get_atomic_events() == [(0,), (1, 2)]


class SetOracle:
    """Fails (reproduces) iff the candidate contains every index of one of the `cores`."""

    def __init__(self, cores):
        self.cores = [set(c) for c in cores]
        self.calls = 0

    def _rep(self, sub):
        return any(c <= set(sub) for c in self.cores)

    def test(self, sub, fp, stats):
        self.calls += 1
        if stats is not None:
            stats.increment_replays()
        return True if self._rep(sub) else None

    def test_batch(self, subs, fp, stats):
        return [self._rep(s) for s in subs]

    def getName(self):
        return "SetOracle"


@pytest.mark.parametrize("cores", [[{3}], [{0, 9}], [{2, 5, 11}], [{1, 2}, {7}], [{0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}], [{4, 6}, {4, 7}]])
@pytest.mark.parametrize("depth", [1, 2, 4])
def test_speculative_ddmin_equals_sequential(cores, depth):
    ev = events_to_array([send(0, M.M_CLIENT, i) for i in range(12)])
    fp = ViolationFingerprint(1)
    d1 = DDMin(SetOracle(cores))
    m1 = d1.minimize(UnmodifiedEventDag(ev), fp)
    d2 = SpeculativeDDMin(SetOracle(cores), depth=depth)
    m2 = d2.minimize(UnmodifiedEventDag(ev), fp)
    assert m1.get_all_events() == m2.get_all_events()
    assert d1.consulted == d2.consulted                    # same verdict consulted at every step
    assert d1._stats.total_replays == d2._stats.total_replays == len(d1.consulted)
    assert d2.speculative_replays >= len({c for c, _ in d2.consulted})
    assert len(d2.batches) <= len(d1.consulted)
    # ddmin's result is 1-minimal: removing any single atom no longer reproduces
    mcs = set(m1.get_all_events())
    assert SetOracle(cores)._rep(mcs)
    for e in mcs:
        assert not SetOracle(cores)._rep(mcs - {e})
    # the reference's accounting assert
    assert d1.original_num_events - d1.total_inputs_pruned == m1.length


def _violating_execution(oracle, model, events, lim, skip=0):
    v = oracle.random_explore(model, events, 4000, seed_base=SEED_BASE, limits=lim, n_threads=4)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][skip])
    vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + i, lim)
    return vv, rec, events[:T.verdict_trace_idx(vv.flags)]


class OracleSTS:
    def __init__(self, oracle, model, used, rec, fp):
        self.o, self.model, self.used, self.rec, self.fp = oracle, model, used, rec, fp

    def _v(self, subs):
        masks = np.array([events_to_mask(s) for s in subs], dtype=np.uint64).reshape(-1, 4)
        return self.o.sts_replay_batch(self.model, self.used, self.rec, masks, T.Limits(0, 0, 64, 1, self.fp, 0))

    def test(self, sub, fp, stats):
        if stats is not None:
            stats.increment_replays()
        r = self._v([sub])[0]
        return r if r["flags"] & T.V_VIOLATION else None

    def test_batch(self, subs, fp, stats):
        return [bool(f & T.V_VIOLATION) for f in self._v(subs)["flags"]]


def test_sts_replay_of_the_unmodified_trace_reproduces_the_execution(oracle):
    """Replaying ALL externals follows the recorded schedule exactly: nothing is ignored, the same
    messages are delivered in the same order (equal delivery hash), the violation reappears."""
    model, events, lim = raft5_config2()
    for skip in range(5):
        vv, rec, used = _violating_execution(oracle, model, events, lim, skip)
        full = np.array([events_to_mask(range(len(used)))], dtype=np.uint64)
        r = oracle.sts_replay_batch(model, used, rec, full, T.Limits(0, 0, 64, 1, vv.fingerprint, 0))[0]
        assert r["flags"] & T.V_VIOLATION and not (r["flags"] & T.V_DIVERGED)
        assert T.verdict_deliveries(int(r["flags"])) == T.verdict_deliveries(vv.flags)
        assert int(r["hash"]) == vv.hash and int(r["fingerprint"]) == vv.fingerprint
        # the empty subsequence delivers nothing; a wrong target never matches
        none = oracle.sts_replay_batch(model, used, rec, np.zeros((1, 4), dtype=np.uint64), T.Limits(0, 0, 64, 1, vv.fingerprint, 0))[0]
        assert T.verdict_deliveries(int(none["flags"])) == 0 and not (none["flags"] & T.V_VIOLATION)
        wrong = oracle.sts_replay_batch(model, used, rec, full, T.Limits(0, 0, 64, 1, vv.fingerprint ^ 0x100, 0))[0]
        assert not (wrong["flags"] & T.V_VIOLATION)


def test_sts_projection_rules(oracle):
    """Pruned Sends drop their MsgSend and MsgEvent; pruned Starts leave the actor isolated; absent
    deliveries are ignored; name-based matching of Spawn/Kill against the cursor head."""
    MSGS = [("Kick", T.MSG_EXTERNAL), ("Ping", T.MSG_INTERNAL)]
    h = {(0, "Kick"): M.Asm().add(M.F[0], M.F[0], 1).mov(M.T0, 1).if_eq(M.ME, 0, "x").send(1, M.T0, M.T1, 0).label("x"),
         (0, "Ping"): M.Asm().add(M.F[1], M.F[1], 1)}
    model = M.build_model("p", 2, MSGS, h, [[0] * 8] * 2, (T.INV_NEVER, 1, 3, 0))    # violation: actor 1 saw 3 Pings
    ev = events_to_array([start(0), start(1), send(0, 0, 1), send(0, 0, 2), wait_quiescence(), kill(1), start(1), send(0, 0, 3)])
    lim = T.Limits(0, 0, 64, 0, 0, 0)
    vv, rec, st = oracle.random_execute(model, ev, 5, lim)
    assert vv.flags & T.V_VIOLATION
    target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)

    def run(keep):
        return oracle.sts_replay_batch(model, ev, rec, np.array([events_to_mask(keep)], dtype=np.uint64), target)[0]

    full = run(range(8))
    assert full["flags"] & T.V_VIOLATION and not full["flags"] & T.V_DIVERGED and T.verdict_deliveries(int(full["flags"])) == 6
    # drop one Kick: its delivery and its Ping disappear (the Ping's expected delivery is ignored)
    r = run([0, 1, 2, 4, 5, 6, 7])
    assert T.verdict_deliveries(int(r["flags"])) == 4 and r["flags"] & T.V_DIVERGED
    # drop the first Start(1) and, as its atom, Kill(1): the later Start(1) is matched BY NAME against the
    # first recorded SpawnEvent of actor 1 (EventTrace.scala:345-354), so actor 1 is up from the beginning,
    # every Ping arrives and the violation is reproduced
    r = run([0, 2, 3, 4, 6, 7])
    assert T.verdict_deliveries(int(r["flags"])) == 6 and r["flags"] & T.V_VIOLATION
    # without any Start(1) the actor stays isolated: every Ping is dropped at send time
    r = run([0, 2, 3, 4, 7])
    assert T.verdict_deliveries(int(r["flags"])) == 3 and not r["flags"] & T.V_VIOLATION and r["flags"] & T.V_DIVERGED


def test_ddmin_over_the_sts_oracle_raft(oracle):
    model, events, lim = raft5_config2()
    vv, rec, used = _violating_execution(oracle, model, events, lim)
    fp = ViolationFingerprint(vv.fingerprint)
    sts = OracleSTS(oracle, model, used, rec, vv.fingerprint)
    mcs1, d1, ver1 = stsSchedDDMin(sts, used, fp, speculative_depth=0)
    mcs2, d2, ver2 = stsSchedDDMin(sts, used, fp, speculative_depth=3)
    assert mcs1 == mcs2 and d1.consulted == d2.consulted and ver1 is not None and ver2 is not None
    assert len(mcs1) < len(used)
    assert all(int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE for i in mcs1)


_GLOO_WORKER = r'''
import os, sys
import torch.distributed as dist
sys.path.insert(0, %(root)r)
from demi_amd.distributed import sharded_map
rank = int(os.environ["RANK"])
dist.init_process_group("gloo")
calls = []
def fn(part):
    calls.append(list(part))
    return [x %% 3 == 0 for x in part]
for n in (0, 1, 2, 7, 64):
    items = list(range(n))
    out = sharded_map(items, fn)
    assert out == [x %% 3 == 0 for x in items], (n, out)
assert all(all(x %% 2 == rank for x in c) for c in calls)      # each rank only evaluated its own share
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_sharded_frontier_evaluation_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o


def _scala_subsequence_intersection(rec, externals, subseq, model):
    """EventTrace.subsequenceIntersection + filterSends (EventTrace.scala:290-452, filterKnownAbsents = false)
    transliterated over the recorded-event array; returns the indices of the events that survive."""
    EXT_OF_REC = {T.REC_SPAWN: T.EV_START, T.REC_KILL: T.EV_KILL, T.REC_PARTITION: T.EV_PARTITION,
                  T.REC_UNPARTITION: T.EV_UNPARTITION}
    remaining = [i for i in subseq if int(externals[i]["kind"]) != T.EV_SEND]
    result = []
    for idx, e in enumerate(rec):
        kind = int(e["kind"])
        is_msg = kind in (T.REC_MSG_SEND, T.REC_MSG_EVENT)
        is_external_kind = kind in EXT_OF_REC
        if not remaining:
            if is_msg or not is_external_kind:
                result.append(idx)
            continue
        if is_external_kind:
            x = externals[remaining[0]]
            if kind in (T.REC_SPAWN, T.REC_KILL):
                same = int(x["kind"]) == EXT_OF_REC[kind] and int(x["a"]) == int(e["rcv"])
            else:
                same = int(x["kind"]) == EXT_OF_REC[kind] and int(x["a"]) == int(e["snd"]) and int(x["b"]) == int(e["rcv"])
            if same:
                result.append(idx)
                remaining = remaining[1:]
        else:
            result.append(idx)
    # filterSends: the k-th external MsgSend belongs to the k-th Send of the original externals
    original_sends = [i for i in range(len(externals)) if int(externals[i]["kind"]) == T.EV_SEND]
    subseq_sends = {i for i in subseq if int(externals[i]["kind"]) == T.EV_SEND}
    missing = {k for k, i in enumerate(original_sends) if i not in subseq_sends}
    msg_send_idx, pruned_ids, out = -1, set(), []
    for idx in result:
        e = rec[idx]
        kind = int(e["kind"])
        if kind == T.REC_MSG_SEND:
            if model.msg_class[int(e["msg_type"])] == T.MSG_EXTERNAL:
                msg_send_idx += 1
                if msg_send_idx in missing:
                    pruned_ids.add(int(e["id"]))
                    continue
            out.append(idx)
        elif kind == T.REC_MSG_EVENT:
            if int(e["id"]) not in pruned_ids:
                out.append(idx)
        else:
            out.append(idx)
    return out


def test_projection_equals_the_scala_transliteration(oracle):
    """For random subsequences (atoms respected or not): what the oracle's replay keeps is exactly what
    subsequenceIntersection + filterSends let through — external events and external MsgSends one for one, and every
    delivered MsgEvent is a projected one (projected but absent ones are the ignored deliveries)."""
    model = M.raft_model(5, election_budget=2)
    from demi_amd.fuzzer import FuzzerWeights, raft_trace
    w = FuzzerWeights(kill=0.12, send=0.4, wait_quiescence=0.13, partition=0.2, unpartition=0.15)
    rng = np.random.default_rng(9)
    checked = 0
    for seed in (1, 2, 3):
        events = events_to_array(raft_trace(5, 70, seed, w, exact=False))
        lim = T.Limits(300, 10, 128, 0, 0, 0)
        vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + seed, lim)
        used = events[:T.verdict_trace_idx(vv.flags)]
        target = T.Limits(0, 0, 128, 1, vv.fingerprint if vv.fingerprint else 0x1000103, 0)
        noq = [i for i in range(len(used)) if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE]
        for _ in range(40):
            subseq = [i for i in noq if rng.random() < rng.choice([0.4, 0.7, 0.95])]
            mask = np.array(events_to_mask(subseq), dtype=np.uint64)
            v, kept = oracle.sts_removal_kept(model, used, rec, 0xFFFFFFFF, target, mask=mask)
            if v.flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF):
                continue
            proj = set(_scala_subsequence_intersection(rec, used, subseq, model))
            kinds = rec["kind"]
            ext_events = set(np.nonzero(kinds <= T.REC_UNPARTITION)[0].tolist())
            ext_sends = set(np.nonzero((kinds == T.REC_MSG_SEND) & ((rec["flags"] & 1) == 1))[0].tolist())
            msg_events = set(np.nonzero(kinds == T.REC_MSG_EVENT)[0].tolist())
            kept_set = set(np.nonzero(kept)[0].tolist())
            assert kept_set & ext_events == proj & ext_events
            assert kept_set & ext_sends == proj & ext_sends
            assert kept_set & msg_events <= proj & msg_events
            assert bool(v.flags & T.V_DIVERGED) == (len(proj & msg_events) > len(kept_set & msg_events))
            checked += 1
    assert checked > 80


def _scala_filter_known_absent_internals(rec, indices, corrected=False):
    """EventTrace.filterKnownAbsentInternals (EventTrace.scala:458-534) transliterated over the recorded-event array:
    `indices` is what subsequenceIntersection + filterSends let through, the result what survives the filter.  As in the
    Scala, a PartitionEvent stores false and an UnPartitionEvent true (:523-528); corrected=True swaps the two and
    looks the pair up in either direction (DEMI_FILTER_ABSENTS_CORRECTED)."""
    alive = {"deadLetters": True, "Timer": True}                 # default false
    partitioned = {}                                               # default false
    pruned_sends = set()

    def name(x):
        return "deadLetters" if int(x) == T.DEADLETTERS else int(x)

    def is_partitioned(snd, rcv):
        if corrected:
            return partitioned.get((snd, rcv), False) or partitioned.get((rcv, snd), False)
        return partitioned.get((snd, rcv), False)

    def sendable(snd, rcv):
        if not alive.get(snd, False):
            return False
        return not is_partitioned(snd, rcv)

    def deliverable(snd, rcv, mid):
        if not alive.get(rcv, False):
            return False
        return not is_partitioned(snd, rcv) and mid not in pruned_sends

    result = []
    for idx in indices:
        e = rec[idx]
        kind = int(e["kind"])
        if kind == T.REC_MSG_SEND:
            if sendable(name(e["snd"]), name(e["rcv"])):
                result.append(idx)
            else:
                pruned_sends.add(int(e["id"]))
        elif kind == T.REC_MSG_EVENT:
            if deliverable(name(e["snd"]), name(e["rcv"]), int(e["id"])):
                result.append(idx)
        elif kind == T.REC_SPAWN:
            alive[name(e["rcv"])] = True
            result.append(idx)
        elif kind == T.REC_KILL:
            alive[name(e["rcv"])] = False
            result.append(idx)
        elif kind == T.REC_PARTITION:
            partitioned[(name(e["snd"]), name(e["rcv"]))] = corrected
            result.append(idx)
        elif kind == T.REC_UNPARTITION:
            partitioned[(name(e["snd"]), name(e["rcv"]))] = not corrected
            result.append(idx)
        else:
            result.append(idx)
    return result


@pytest.mark.parametrize("mode", [T.FILTER_ABSENTS_LITERAL, T.FILTER_ABSENTS_CORRECTED])
def test_filter_known_absent_internals_equals_the_scala_transliteration(oracle, mode):
    """SchedulerConfig.filterKnownAbsents: the replay walks exactly the trace that subsequenceIntersection + filterSends
    + filterKnownAbsentInternals produce - the reference's inverted Partition / UnPartition bookkeeping included (LITERAL)."""
    model = M.raft_model(5, election_budget=2)
    from demi_amd.fuzzer import FuzzerWeights, raft_trace
    w = FuzzerWeights(kill=0.12, send=0.4, wait_quiescence=0.13, partition=0.2, unpartition=0.15)
    rng = np.random.default_rng(17)
    checked = filtered = changed = 0
    for seed in (1, 2, 3, 4):
        events = events_to_array(raft_trace(5, 70, seed, w, exact=False))
        lim = T.Limits(300, 10, 128, 0, 0, 0)
        vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + seed, lim)
        used = events[:T.verdict_trace_idx(vv.flags)]
        fpc = vv.fingerprint if vv.fingerprint else 0x1000103
        target = T.Limits(0, 0, 128, 1, fpc, 0, 0, mode)
        plain = T.Limits(0, 0, 128, 1, fpc, 0, 0, 0)
        noq = [i for i in range(len(used)) if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE]
        kinds = rec["kind"]
        ext_events = set(np.nonzero(kinds <= T.REC_UNPARTITION)[0].tolist())
        ext_sends = set(np.nonzero((kinds == T.REC_MSG_SEND) & ((rec["flags"] & 1) == 1))[0].tolist())
        msg_events = set(np.nonzero(kinds == T.REC_MSG_EVENT)[0].tolist())
        for _ in range(40):
            subseq = [i for i in noq if rng.random() < rng.choice([0.4, 0.7, 0.95, 1.0])]
            mask = np.array(events_to_mask(subseq), dtype=np.uint64)
            v, kept = oracle.sts_removal_kept(model, used, rec, 0xFFFFFFFF, target, mask=mask)
            if v.flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF):
                continue
            proj = _scala_subsequence_intersection(rec, used, subseq, model)
            proj_f = set(_scala_filter_known_absent_internals(rec, proj, corrected=(mode == T.FILTER_ABSENTS_CORRECTED)))
            kept_set = set(np.nonzero(kept)[0].tolist())
            assert kept_set & ext_events == proj_f & ext_events
            assert kept_set & ext_sends == proj_f & ext_sends
            assert kept_set & msg_events <= proj_f & msg_events
            assert bool(v.flags & T.V_DIVERGED) == (len(proj_f & msg_events) > len(kept_set & msg_events))
            filtered += len(set(proj) & msg_events) - len(proj_f & msg_events)
            v0 = oracle.sts_replay_batch(model, used, rec, mask[None, :], plain)[0]
            changed += int(v0["hash"] != v.hash or (int(v0["flags"]) ^ int(v.flags)) & T.V_DIVERGED != 0)
            checked += 1
    assert checked > 100 and filtered > 100 and changed > 0


def test_native_ddmin_loop_equals_the_python_mirror(oracle):
    """demi_ddmin's host loop (demi_amd/csrc/ddmin_host.hpp: atoms, split_list, ddmin2, the speculative frontier) over the CPU
    oracle's replays against SpeculativeDDMin / DDMin of the Python mirror: the same MCS, the same consultations in the same
    order with the same outcomes, the same candidates per launch at a fixed depth - on raft5 executions and on fault-heavy
    traces with Kill / Partition atoms - and with a launch budget instead of a depth only the launches differ."""
    from demi_amd.fuzzer import FuzzerWeights, raft_trace
    from oracle import oracle_py
    cases = []
    model, events, lim = raft5_config2()
    for skip in (0, 2):
        cases.append((model,) + _violating_execution(oracle, model, events, lim, skip))
    fm = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.12, send=0.35, wait_quiescence=0.13, partition=0.25, unpartition=0.15)
    for seed in (2, 7):
        ev = events_to_array(raft_trace(5, 90, seed, w, exact=False))
        v = oracle.random_explore(fm, ev, 2000, seed_base=SEED_BASE, limits=T.Limits(400, 10, 128, 0, 0, 0), n_threads=4)
        hits = np.nonzero(v["flags"] & T.V_VIOLATION)[0]
        if not len(hits):
            continue
        vv, rec, _ = oracle.random_execute(fm, ev, SEED_BASE + int(hits[0]), T.Limits(400, 10, 128, 0, 0, 0))
        cases.append((fm, vv, rec, ev[:T.verdict_trace_idx(vv.flags)]))
    assert len(cases) == 4
    for model, vv, rec, used in cases:
        fp = ViolationFingerprint(vv.fingerprint)
        target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
        sts = OracleSTS(oracle, model, used, rec, vv.fingerprint)
        sts._v = lambda subs, o=oracle, m=model, u=used, r=rec: o.sts_replay_batch(
            m, u, r, np.array([events_to_mask(s) for s in subs], dtype=np.uint64).reshape(-1, 4), target)
        for depth in (1, 3):
            mcs_p, dd_p, _ = stsSchedDDMin(sts, used, fp, speculative_depth=depth)
            mcs_n, cons_n, batches_n, st = oracle_py.ddmin(model, used, rec, target, T.DdminParams(depth, 0, 1, 1))
            assert tuple(mcs_n) == tuple(mcs_p) and st.mcs_len == len(mcs_p) and st.verified == 1
            assert cons_n == [(tuple(c), p) for c, p in dd_p.consulted] and st.consultations == len(dd_p.consulted)
            assert batches_n[1:-1] == dd_p.batches and batches_n[0] == batches_n[-1] == 1        # (the check and the verification)
            assert st.launches == len(dd_p.batches) + 2 and st.replays == dd_p.speculative_replays + 2
        mcs_s, dd_s, _ = stsSchedDDMin(sts, used, fp, speculative_depth=0)
        for budget in (16, 4096):
            mcs_n, cons_n, batches_n, st = oracle_py.ddmin(model, used, rec, target, T.DdminParams(0, budget, 1, 1))
            assert tuple(mcs_n) == tuple(mcs_s) and cons_n == [(tuple(c), p) for c, p in dd_s.consulted]
            assert all(b <= max(budget, 4) * 4 for b in batches_n)
        assert st.launches <= 6                                                                  # a few wide launches instead of one per consultation
    # explicitly conjoined atoms (UnmodifiedEventDag.conjoinAtoms): two pairs of Sends that may only be removed together
    model, vv, rec, used = cases[0]
    fp = ViolationFingerprint(vv.fingerprint)
    target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
    sends = [i for i in range(len(used)) if int(used[i]["kind"]) == T.EV_SEND]
    pairs = [(sends[1], sends[6]), (sends[3], sends[4])]
    dag = UnmodifiedEventDag(used)
    conj = np.full(len(used), 255, dtype=np.uint8)
    for a, b in pairs:
        dag.conjoinAtoms(a, b)
        conj[a], conj[b] = b, a
    sts = OracleSTS(oracle, model, used, rec, vv.fingerprint)
    sts._v = lambda subs: oracle.sts_replay_batch(model, used, rec, np.array([events_to_mask(x) for x in subs], dtype=np.uint64).reshape(-1, 4), target)
    keep = tuple(i for i in dag.events if int(used[i]["kind"]) != T.EV_WAIT_QUIESCENCE)
    dd = DDMin(sts, checkUnmodifed=True)
    mcs_p = dd.minimize(EventDagView(dag, keep), fp).get_all_events()
    mcs_n, cons_n, _, st = oracle_py.ddmin(model, used, rec, target, T.DdminParams(0, 64, 1, 1), conjoined=conj)
    assert tuple(mcs_n) == tuple(mcs_p) and cons_n == [(tuple(c), p) for c, p in dd.consulted]
    for a, b in pairs:
        assert (a in mcs_n) == (b in mcs_n)
    # an unmodified trace that does not reproduce, and a Kill whose Start is not among the events
    model, vv, rec, used = cases[0]
    with pytest.raises(RuntimeError, match="-1"):
        oracle_py.ddmin(model, used, rec, T.Limits(0, 0, 128, 1, 0x7777, 0), T.DdminParams(2, 0, 1, 1))


def test_native_ddmin_loop_on_arbitrary_oracles(oracle):
    """The DDMin host loop (ddmin_host.hpp) around an ARBITRARY oracle - a Python callback - against the mirror's DDMin on random
    external-event lists with Kill / Partition / UnPartition atoms and conjoined pairs: monotone predicates ("contains this
    set"), non-monotone ones (a hash of the candidate), with a fixed depth and with launch budgets.  Checks in particular that
    computing the atoms once (a view ddmin2 can reach is a union of whole atoms) gives what the mirror gets by recomputing them
    for every view, as the reference does."""
    import ctypes as C
    from oracle import oracle_py
    oracle_py.build()
    H = C.CDLL(os.path.join(os.path.dirname(oracle_py.__file__), "_build", "dpor_host_harness.so"))
    CB = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_uint64), C.c_uint32, C.POINTER(C.c_uint8))
    H.harness_ddmin_callback.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(T.DdminParams), C.c_void_p, CB, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(T.DdminStats)]
    rng = np.random.default_rng(77)

    def random_externals(n):
        ev, alive, parts = [], set(), set()
        for a in range(4):
            ev.append(start(a)); alive.add(a)
        while len(ev) < n:
            k = int(rng.integers(0, 10))
            a, b = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            if k < 5:
                ev.append(send(a, 0, int(rng.integers(0, 9))))
            elif k == 5 and a in alive and len(alive) > 1:
                ev.append(kill(a)); alive.discard(a)
            elif k == 6 and a not in alive:
                ev.append(start(a)); alive.add(a)
            elif k == 7 and a != b and (a, b) not in parts:
                ev.append(partition(a, b)); parts.add((a, b))
            elif k == 8 and parts:
                p = sorted(parts)[int(rng.integers(0, len(parts)))]
                ev.append(unpartition(*p)); parts.discard(p)
            elif k == 9:
                ev.append(wait_quiescence())
        return events_to_array(ev)

    class PyOracle:
        def __init__(self, pred):
            self.pred = pred

        def test(self, sub, fp, stats):
            return True if self.pred(tuple(sub)) else None

        def test_batch(self, subs, fp, stats):
            return [bool(self.pred(tuple(s))) for s in subs]

    checked = 0
    for trial in range(60):
        ext = random_externals(int(rng.integers(8, 60)))
        n = len(ext)
        dag = UnmodifiedEventDag(ext)
        keep = tuple(i for i in dag.events if int(ext[i]["kind"]) != T.EV_WAIT_QUIESCENCE)
        conj = np.full(n, 255, dtype=np.uint8)
        sends = [i for i in keep if int(ext[i]["kind"]) == T.EV_SEND]
        if trial % 3 == 0 and len(sends) >= 4:
            a, b = sorted(rng.choice(sends, 2, replace=False).tolist())
            dag.conjoinAtoms(int(a), int(b)); conj[a], conj[b] = b, a
        if trial % 2:
            need = set(rng.choice(keep, int(rng.integers(1, 4)), replace=False).tolist())
            pred = lambda s, need=need: need.issubset(s)                      # monotone: the violation needs these events
        else:
            salt = int(rng.integers(1, 1 << 30))
            full = tuple(keep)
            pred = lambda s, salt=salt, full=full: s == full or (hash((s, salt)) % 3 == 0)      # arbitrary (the whole trace fails)
        try:
            view = EventDagView(dag, keep)
            dd = DDMin(PyOracle(pred), checkUnmodifed=True)
            mcs_p = dd.minimize(view, ViolationFingerprint(1)).get_all_events()
        except (RuntimeError, AssertionError, ValueError):
            continue                                                          # (a list whose atoms the reference rejects)
        for par in (T.DdminParams(2, 0, 1, 0), T.DdminParams(0, 8, 1, 0), T.DdminParams(0, 4096, 1, 0)):
            def cb(masks, cnt, reproduced, pred=pred):
                for i in range(cnt):
                    m = [masks[4 * i + k] for k in range(4)]
                    sub = tuple(e for e in range(n) if (m[e >> 6] >> (e & 63)) & 1)
                    reproduced[i] = 1 if pred(sub) else 0
                return 0
            mcs = np.zeros(4, dtype=np.uint64); consulted = np.zeros((4096, 4), dtype=np.uint64); passed = np.zeros(4096, dtype=np.uint8)
            st = T.DdminStats()
            rc = H.harness_ddmin_callback(ext.ctypes.data, n, C.byref(par), conj.ctypes.data, CB(cb), mcs.ctypes.data, consulted.ctypes.data,
                                          passed.ctypes.data, 4096, None, 0, C.byref(st))
            assert rc == 0, (trial, rc)
            assert T.mask_to_events(mcs) == tuple(mcs_p), (trial, par.depth, par.max_candidates)
            got = [(T.mask_to_events(consulted[i]), bool(passed[i])) for i in range(st.consultations)]
            assert got == [(tuple(c), p) for c, p in dd.consulted], (trial, par.depth, par.max_candidates)
            checked += 1
    assert checked > 100
