"""CPU suite, part 8: ProvenanceTracker.pruneConcurrentEvents against a literal set-of-pairs transliteration."""
import itertools

import numpy as np

from demi_amd import types as T
from demi_amd.provenance import ProvenanceTracker, pruneConcurrentEvents


def _trace(rows):
    """rows: (receiver, parent index); index 0 is the root."""
    a = np.zeros(len(rows) + 1, dtype=T.DPOR_TRACE_DTYPE)
    a["key"][0] = T.DPOR_ROOT_KEY
    for i, (rcv, parent) in enumerate(rows, start=1):
        a["key"][i] = 1000 + i
        a["word"][i] = (rcv << 5) | 1
        a["parent"][i] = parent
        a["kind"][i] = 1
    return a


def _literal(trace, affected):
    """schedulers/Util.scala:267-376 with sets of pairs and a naive closure."""
    n = len(trace)
    rcv = [-1] + [(int(w) >> 5) & 7 for w in trace["word"][1:]]
    hb = set()
    prior = {}
    for u in range(n):
        prior.setdefault(rcv[u], []).append(u)
        for p in prior[rcv[u]]:
            hb.add((p, u))
        for s in range(1, n):
            if int(trace["parent"][s]) == u and s != u:
                hb.add((u, s))
    changed = True
    while changed:
        changed = False
        for (a, b), (c, d) in itertools.product(list(hb), repeat=2):
            if b == c and (a, d) not in hb:
                hb.add((a, d)); changed = True
    last = []
    for node in affected:
        idx = [i for i in range(n) if rcv[i] == node]
        if idx:
            last.append(idx[-1])
    keep = []
    for u in range(n):
        conc_or_after_all = all((not ((o, u) in hb or (u, o) in hb)) or ((o, u) in hb) for o in last)
        if not conc_or_after_all:
            keep.append(u)
    return keep, hb


def test_hand_example():
    # root -> a(rcv 0) -> b(rcv 1) -> c(rcv 0);  root -> d(rcv 2) (independent);  root -> e(rcv 1) after b
    tr = _trace([(0, 0), (1, 1), (0, 2), (2, 0), (1, 0)])
    pt = ProvenanceTracker(tr)
    assert pt.happensBefore[1, 3] and pt.happensBefore[2, 5] and not pt.happensBefore[4, 3]
    assert pt.concurrent(4, 3) and not pt.concurrent(1, 3)
    # violation at node 0: its last receive is c (index 3); kept = strict causal past of c
    assert list(pt.pruneConcurrentEvents([0])) == [0, 1, 2]
    # nodes 0 and 1: last receives c and e; c itself precedes nothing, e is after b
    assert list(pt.pruneConcurrentEvents([0, 1])) == [0, 1, 2]
    assert len(pruneConcurrentEvents(tr, [5])) == 0          # no such node: everything is pruned


def test_random_traces_match_the_literal_algorithm():
    rng = np.random.default_rng(3)
    for _ in range(40):
        n = int(rng.integers(2, 14))
        rows = [(int(rng.integers(0, 4)), int(rng.integers(0, i + 1))) for i in range(n)]
        tr = _trace(rows)
        pt = ProvenanceTracker(tr)
        affected = [int(x) for x in rng.choice(4, size=int(rng.integers(1, 4)), replace=False)]
        keep, hb = _literal(tr, affected)
        got_hb = {(int(a), int(b)) for a, b in zip(*np.nonzero(pt.happensBefore))}
        assert got_hb == hb
        assert list(pt.pruneConcurrentEvents(affected)) == keep
