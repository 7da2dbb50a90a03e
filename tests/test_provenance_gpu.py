"""GPU suite: the provenance kernel (demi_provenance_prune) against the literal set-of-pairs transliteration of
schedulers/Util.scala:267-376 (small traces) and the host class (recorded executions, random forests up to 256 events)."""
import numpy as np
import pytest

from demi_amd import types as T
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.incremental_ddmin import dpor_initial_trace
from demi_amd.provenance import ProvenanceTracker, pruneConcurrentEvents, pruneConcurrentEventsBatch
from demi_amd.schedulers import EventTrace

from .test_provenance_cpu import _literal, _trace

pytestmark = pytest.mark.gpu


def test_small_random_traces_equal_the_literal_algorithm(gpu_ctx):
    rng = np.random.default_rng(3)
    traces, affected, want = [], [], []
    for _ in range(300):
        n = int(rng.integers(1, 14))
        rows = [(int(rng.integers(0, 4)), int(rng.integers(0, i + 1))) for i in range(n)]
        tr = _trace(rows)
        if rng.integers(0, 3) == 0 and n > 3:                 # a WaitQuiescence marker in the middle (neither before nor after anything)
            k = int(rng.integers(1, n))
            tr["kind"][k] = 2
            tr["word"][k] = 0
        aff = [int(x) for x in rng.choice(5, size=int(rng.integers(1, 4)), replace=False)]
        traces.append(tr); affected.append(aff)
        if (tr["kind"][1:] == 1).all():
            want.append(_literal(tr, aff)[0])
        else:
            want.append(list(ProvenanceTracker(tr).pruneConcurrentEvents(aff)))
    got = pruneConcurrentEventsBatch(gpu_ctx, traces, affected)
    for tr, g, w in zip(traces, got, want):
        assert [int(k) for k in g["key"]] == [int(tr["key"][i]) for i in w]
    assert len(pruneConcurrentEventsBatch(gpu_ctx, [traces[0]], [[7]])[0]) == 0      # no such node: everything is pruned
    assert pruneConcurrentEventsBatch(gpu_ctx, [], []) == []


def test_large_forests_and_recorded_executions_equal_the_host_class(gpu_ctx):
    rng = np.random.default_rng(8)
    traces, affected = [], []
    for _ in range(200):
        n = int(rng.choice([40, 130, 255]))
        # parents biased to recent events, as in a real execution; 5 machines
        rows = [(int(rng.integers(0, 5)), int(max(0, i - rng.integers(0, 12)))) for i in range(n)]
        traces.append(_trace(rows))
        affected.append([int(x) for x in rng.choice(5, size=int(rng.integers(1, 4)), replace=False)])
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    v = gpu_ctx.random_explore(3000, lim, seed_base=SEED_BASE)
    for i in np.nonzero(v["flags"] & T.V_VIOLATION)[0][:40]:
        vv, rec = gpu_ctx.random_get_trace(SEED_BASE + int(i), lim)
        it = dpor_initial_trace(EventTrace(rec, events[:T.verdict_trace_idx(vv.flags)]))
        if len(it) <= T.DPOR_MAX_TRACE:
            traces.append(it)
            affected.append([a for a in range(T.MAX_ACTORS) if (vv.fingerprint >> a) & 1] or [0])
    assert len(traces) > 220
    got = pruneConcurrentEventsBatch(gpu_ctx, traces, affected)
    kept_some = 0
    for tr, aff, g in zip(traces, affected, got):
        w = pruneConcurrentEvents(tr, aff)
        assert len(g) == len(w) and (g == w).all()
        kept_some += int(0 < len(g) < len(tr))
    assert kept_some > 100
    # the single-trace form and the API errors
    assert (pruneConcurrentEvents(traces[0], affected[0], ctx=gpu_ctx) == got[0]).all()
    from demi_amd._native import DemiError
    with pytest.raises(DemiError):
        gpu_ctx.provenance_prune([np.zeros(T.DPOR_MAX_TRACE + 1, dtype=T.DPOR_TRACE_DTYPE)], [1])
