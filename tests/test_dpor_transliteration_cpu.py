"""CPU suite: DPORwHeuristics.dpor() / getNext() / ExploredTacker transliterated from the Scala
(schedulers/DPORwHeuristics.scala:994-1185, schedulers/AuxilaryTypes.scala:209-246, BacktrackOrdering.scala:58-62) around single
interleavings of the oracle, against (a) the racing pairs the oracle reports for every interleaving and (b) the whole
exploration of the product's bookkeeping (demi_amd/csrc/dpor_host.hpp) in the reference's order.

The transliteration keeps the reference's own data structures: the dependency graph as "edge unique -> parentEvent"
(`getCommonPrefix` = intersect the two paths to the root), `exploredStack: HashMap[Int, HashSet[(Unique, Unique)]]` with
ORDERED pairs, a priority queue ordered by the branch index only, and `trace.take(maxIndex + 1) ++ replayThis` taken from
the trace that has just run (:1180) - not from the trace that produced the backtrack point, which is what the product
stores; the two coincide in the reference's (depth-first) order, and this test is what shows it.
Pinned, as in the product: PriorityQueue ties pop in creation order; a Unique is identified by its causal-path key."""
import heapq

import numpy as np
import pytest

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.fuzzer import events_to_array, send, start, wait_quiescence

from .test_dpor_cpu import PAR, native_explore, writers_model


class _ScalaDpor:
    def __init__(self, oracle, model, ext, params, trackHistory=True, stopIfViolationFound=False):
        self.oracle, self.model, self.ext, self.params = oracle, model, ext, params
        self.trackHistory, self.stopIfViolationFound = trackHistory, stopIfViolationFound
        self.parentOf = {}            # depGraph: addEdge(unique -> parentEvent)
        self.exploredStack = {}       # ExploredTacker.exploredStack
        self.backTrack = []           # PriorityQueue[(Int, (Unique, Unique), List[Unique])] by _1; creation order on ties
        self.seq = 0
        self.interleavingCounter = 0
        self.shortestTraceSoFar = None
        self.pairs_checked = 0

    # ---- ExploredTacker
    def setExplored(self, index, pair):
        self.exploredStack.setdefault(index, set()).add(pair)

    def isExplored(self, pair):
        for _index, s in self.exploredStack.items():
            if pair in s:
                return True
        return False

    # ---- depGraph
    def pathToRoot(self, u):
        path = [u]
        while u in self.parentOf:
            u = self.parentOf[u]
            path.append(u)
        return path                    # u ... root

    def getCommonPrefix(self, earlier, later):
        laterPath = list(reversed(self.pathToRoot(later)))
        earlierPath = list(reversed(self.pathToRoot(earlier)))
        rest = list(earlierPath)       # Seq.intersect: multiset intersection in the order of the left operand
        out = []
        for x in laterPath:
            if x in rest:
                rest.remove(x)
                out.append(x)
        return out

    def run(self, max_interleavings):
        verdicts, next_traces = [], []
        nextTrace = []
        while len(verdicts) < max_interleavings:
            pf = np.zeros(len(nextTrace), dtype=T.DPOR_TRACE_DTYPE)
            pf["key"] = np.array(nextTrace, dtype=np.uint64)
            v, traces, pairs = self.oracle.dpor_batch(self.model, self.ext, [pf], self.params)
            verdicts.append(v[0].copy())
            next_traces.append(list(nextTrace))
            tr = traces[0]
            if (int(v[0]["flags"]) & T.V_VIOLATION) and self.shortestTraceSoFar is None:
                self.shortestTraceSoFar = tr
            nxt = self.dpor(tr, pairs[0])
            if nxt is None:
                return verdicts, next_traces, True
            nextTrace = nxt
        return verdicts, next_traces, False

    def dpor(self, tr, oracle_pairs):
        self.interleavingCounter += 1
        trace = [int(k) for k in tr["key"]]                    # currentTrace: Queue[Unique]
        kind = [int(k) for k in tr["kind"]]
        rcv = [(int(w) >> 5) & 7 for w in tr["word"]]
        qp = [int(q) for q in tr["qperiod"]]
        for i in range(1, len(trace)):                          # (the scheduling half added these edges while it ran)
            self.parentOf[trace[i]] = trace[int(tr["parent"][i])]

        def isCoEnabeled(e, l):
            if kind[e] == 2 or kind[l] == 2:                    # "Quiescence is never co-enabled"
                return False
            if kind[e] == 0 or kind[l] == 0:                    # the root MsgEvent("null", "null", null): nobody shares its receiver
                return False
            if rcv[e] != rcv[l]:
                return False
            if qp[e] != qp[l]:
                return False
            return trace[e] not in self.pathToRoot(trace[l])    # laterN.pathTo(earlierN) match { case None => true }

        def analyze_dep(earlierI, laterI):
            earlier, later = trace[earlierI], trace[laterI]
            commonPrefix = self.getCommonPrefix(earlier, later)
            lastElement = commonPrefix[-1]
            branchI = trace.index(lastElement)
            needToReplay = [x for x in trace[branchI + 1:][:laterI - branchI] if x != earlier]   # drop(branchI+1).dropRight(size-laterI-1).filter
            assert branchI < laterI
            if self.trackHistory:
                self.setExplored(branchI, (earlier, later))
            return branchI, needToReplay

        mine = []
        for laterI in range(len(trace)):
            for earlierI in range(laterI):
                if isCoEnabeled(earlierI, laterI):
                    branchI, needToReplay = analyze_dep(earlierI, laterI)
                    heapq.heappush(self.backTrack, (-branchI, self.seq, (trace[laterI], trace[earlierI]), needToReplay))
                    self.seq += 1
                    mine.append((branchI, laterI, earlierI))
        # (a) the oracle's racing pairs of this interleaving: the same list in the same order
        got = [(int(p["branch"]), int(p["later"]), int(p["earlier"])) for p in oracle_pairs]
        assert got == mine
        self.pairs_checked += len(mine)

        # getNext
        while True:
            if not self.backTrack or (self.stopIfViolationFound and self.shortestTraceSoFar is not None):
                return None
            negI, _s, (e1, e2), replayThis = heapq.heappop(self.backTrack)
            if self.trackHistory and self.isExplored((e1, e2)):
                continue
            maxIndex = -negI
            if self.trackHistory:
                self.setExplored(maxIndex, (e1, e2))
            return trace[:maxIndex + 1] + replayThis           # trace.take(maxIndex + 1) ++ replayThis


CASES = {
    "writers4": lambda: (writers_model(4), events_to_array([start(a) for a in range(5)] + [send(a, 0) for a in range(1, 5)]), PAR()),
    "raft3": lambda: (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)]),
                      PAR(depth=30)),
    "raft3_two_periods": lambda: (M.raft_model(3), events_to_array([start(a) for a in range(3)] + [send(0, M.M_BOOTSTRAP),
                                  wait_quiescence(), send(1, M.M_BOOTSTRAP), send(2, M.M_BOOTSTRAP)]), PAR(depth=24)),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_reference_order_exploration_equals_the_scala_transliteration(oracle, case):
    model, ev, par = CASES[case]()
    cap = 2500 if case == "writers4" else 500       # (writers4 exhausts after a few dozen; the raft3 cases are cut by the budget)
    sc = _ScalaDpor(oracle, model, ev, par)
    verdicts, next_traces, exhausted = sc.run(cap)
    assert sc.pairs_checked > (20 if case == "writers4" else 2000)
    # (b) the product's bookkeeping, one backtrack point at a time and in the speculating REFERENCE order
    for batch, ref in ((1, False), (64, True)):
        nat = native_explore(model, ev, par, batch, cap, reference_order=ref)
        assert len(nat[0]) == len(verdicts) and bool(nat[4].exhausted) == exhausted
        assert (nat[0] == np.array(verdicts, dtype=T.VERDICT_DTYPE)).all()          # flags, fingerprint, delivery hash, in order
        assert [int(x) for x in nat[1]] == [len(t) for t in next_traces]             # the next trace each one replayed
    viol = [i for i, v in enumerate(verdicts) if int(v["flags"]) & T.V_VIOLATION]
    if case == "writers4":
        assert 0 < len(viol) < len(verdicts) and exhausted
        # stopIfViolationFound: the same sequence up to and including the first violating interleaving
        s = _ScalaDpor(oracle, model, ev, par, stopIfViolationFound=True)
        sv, _, _ = s.run(cap)
        assert len(sv) == viol[0] + 1
        nat = native_explore(model, ev, par, 1, cap, stop=True)
        assert len(nat[0]) == len(sv) and (nat[0] == np.array(sv, dtype=T.VERDICT_DTYPE)).all()


def test_without_history_every_backtrack_point_is_explored(oracle):
    """trackHistory = false: no ExploredTacker at all (:1066, 1153, 1166): the queue only drains by the interleaving budget."""
    model, ev, par = CASES["writers4"]()
    sc = _ScalaDpor(oracle, model, ev, par, trackHistory=False)
    verdicts, next_traces, exhausted = sc.run(300)
    nat = native_explore(model, ev, par, 1, 300, track=False)
    assert len(nat[0]) == len(verdicts) == 300 and not exhausted
    assert (nat[0] == np.array(verdicts, dtype=T.VERDICT_DTYPE)).all()
