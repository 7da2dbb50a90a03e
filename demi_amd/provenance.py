"""ProvenanceTracker: prune the deliveries that are not in the causal past of the violation.

Mirror of schedulers/Util.scala:267-376 (ProvenanceTracker) and RunnerUtils.pruneConcurrentEvents
(RunnerUtils.scala:149-163), over the delivery trace of one execution as `dpor_initial_trace` produces it (root
followed by the deliveries, each with the trace index of the delivery that sent it).  Two forms: the class below (host,
numpy bit matrix - the shape of the reference's own class, one trace at a time) and `pruneConcurrentEventsBatch`, which
hands any number of traces to the library's bitset kernel (demi_provenance_prune, demi_amd/csrc/k_provenance.hpp: one
wavefront per trace) - what RunnerUtils.fuzz uses when it runs on a device, and the way to prune every violating
execution of a fuzz run at once.

happens-before, first order (:283-299): every earlier receive on the same machine precedes a receive (the pair
(u, u) included, as in the reference), and a receive precedes the messages sent while it was handled (its
children in the dep-graph).  Then the transitive closure (:316-349).  `pruneConcurrentEvents` (:355-375) keeps an
event iff it strictly precedes the last receive of at least one node named by the violation fingerprint.
"""
from typing import List, Sequence

import numpy as np

from . import types as T


class ProvenanceTracker:
    def __init__(self, trace: np.ndarray, big: bool = False):
        """trace: DPOR_TRACE_DTYPE (key, word, parent, kind), index 0 = root.  big: the trace of a table with more than 8 actors
        (the BIG layout of include/demi_gpu.h: a 4-bit receiver field in the entries' words)."""
        self.trace = np.ascontiguousarray(trace, dtype=T.DPOR_TRACE_DTYPE)
        n = len(self.trace)
        word = self.trace["word"].astype(np.int64)
        rcv = (word >> 5) & (15 if big else 7)
        is_msg = (self.trace["kind"] == 1) | (np.arange(n) == 0)         # the root is a MsgEvent("null", "null", null)
        rcv = np.where(np.arange(n) == 0, -1, rcv)
        hb = np.zeros((n, n), dtype=bool)
        parent = self.trace["parent"].astype(np.int64)
        for u in range(n):
            if not is_msg[u]:
                continue
            prior = is_msg[:u + 1] & (rcv[:u + 1] == rcv[u])              # priorReceives, u itself included
            hb[:u + 1, u] |= prior
        for s in range(1, n):
            if is_msg[s] and is_msg[parent[s]]:
                hb[parent[s], s] = True                                   # sends that result from the receive
        # transitive closure, last event first: every edge goes forward in trace order
        reach = hb.copy()
        for u in range(n - 1, -1, -1):
            succ = np.nonzero(hb[u])[0]
            succ = succ[succ != u]
            if len(succ):
                reach[u] |= reach[succ].any(axis=0)
        self.happensBefore = reach
        self.is_msg = is_msg
        self.rcv = rcv

    def concurrent(self, a: int, b: int) -> bool:
        return not (self.happensBefore[a, b] or self.happensBefore[b, a])

    def pruneConcurrentEvents(self, affectedNodes: Sequence[int]) -> np.ndarray:
        """Indices (into the trace) of the events that are kept."""
        n = len(self.trace)
        last: List[int] = []
        for node in affectedNodes:
            idx = np.nonzero(self.is_msg & (self.rcv == node))[0]
            if len(idx):
                last.append(int(idx[-1]))
        if not last:
            return np.zeros(0, dtype=np.int64)         # forall over an empty set: everything is pruned (:366-368)
        hb = self.happensBefore
        # removed iff for every last event o: concurrent(o, u) or o happens-before u
        o = np.array(last)
        removed = np.ones(n, dtype=bool)
        for oi in o:
            conc = ~hb[oi, :] & ~hb[:, oi]
            removed &= conc | hb[oi, :]
        return np.nonzero(~removed)[0]


def pruneConcurrentEvents(initialTrace: np.ndarray, affectedNodes: Sequence[int], ctx=None, big: bool = False) -> np.ndarray:
    """RunnerUtils.pruneConcurrentEvents: the initial trace restricted to the provenance of the violation.
    ctx: a demi_amd._native.Context - the closure and the pruning then run on its device (in the layout of the table it holds)."""
    if ctx is not None:
        return pruneConcurrentEventsBatch(ctx, [initialTrace], [affectedNodes])[0]
    keep = ProvenanceTracker(initialTrace, big).pruneConcurrentEvents(affectedNodes)
    return np.ascontiguousarray(initialTrace)[keep]


def pruneConcurrentEventsBatch(ctx, initialTraces: Sequence[np.ndarray], affectedNodes: Sequence[Sequence[int]]) -> List[np.ndarray]:
    """pruneConcurrentEvents for many executions in one launch (one wavefront per trace)."""
    traces = [np.ascontiguousarray(t, dtype=T.DPOR_TRACE_DTYPE) for t in initialTraces]
    masks = [sum(1 << int(a) for a in set(nodes) if 0 <= int(a) < T.MAX_ACTORS_BIG) for nodes in affectedNodes]
    kept = ctx.provenance_prune(traces, masks)
    return [t[k] for t, k in zip(traces, kept)]
