"""GPU suite: RunnerUtils.randomDDMin natively (demi_random_ddmin) and the kernel variant under it (K1 over a frontier of
candidate subsequences, a workgroup per candidate: k1_random_explore<.., MULTI>, demi_random_explore_candidates).

Held against (a) the CPU oracle one candidate at a time - orc_random_explore on the candidate's own external events, which is
what `trace_load(candidate)` + demi_random_explore computes - and (b) the reference's loop written out in Python
(minification.DDMin: sequential, one consultation at a time) around that oracle: same MCS, same consultation sequence."""
import os

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2, raft5_config4
from demi_amd.minification import DDMin, UnmodifiedEventDag, randomDDMin
from demi_amd.schedulers import EventTrace, SchedulerConfig, ViolationFingerprint

pytestmark = pytest.mark.gpu
EMU = os.environ.get("DEMI_EMU") == "1"


def _masks(cands, n):
    m = np.zeros((len(cands), 4), dtype=np.uint64)
    for k, c in enumerate(cands):
        for i in c:
            m[k, i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    return m


def _atom_candidates(events, rng, n_cand):
    """subsequences that are unions of whole atoms (a Kill goes with its Start, an UnPartition with its Partition): what DDMin asks"""
    dag = UnmodifiedEventDag(events)
    atoms = dag.get_atomic_events()
    out = [tuple(range(len(events))), tuple(sorted(i for a in atoms[:1] for i in a))]
    while len(out) < n_cand:
        keep = rng.random(len(atoms)) < rng.choice([0.3, 0.6, 0.9])
        out.append(tuple(sorted(i for a, k in zip(atoms, keep) if k for i in a)))
    return out


@pytest.mark.parametrize("specialize", [False, True])
@pytest.mark.parametrize("strategy", [T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO])
def test_candidate_batch_equals_the_oracle_per_candidate(oracle, specialize, strategy):
    model, events, lim = raft5_config2()
    lim.strategy = strategy
    R = 70 if not EMU else 40                 # (not a multiple of 64: the last wave of a candidate's workgroup is partly idle)
    rng = np.random.default_rng(11)
    cands = _atom_candidates(events, rng, 24 if not EMU else 6) + [()]
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    if specialize:
        ctx.model_specialize()
    ctx.trace_load(events)
    for looking_for in (None, 0):
        l2 = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 0, 0, 0, strategy)
        v, f = ctx.random_explore_candidates(_masks(cands, len(events)), R, l2, seed_base=SEED_BASE)
        assert v.shape == (len(cands), R)
        for k, c in enumerate(cands):
            want = oracle.random_explore(model, events[list(c)], R, seed_base=SEED_BASE, limits=l2)
            assert (v[k] == want).all(), (k, c)
            assert bool(f[k] & 1) == bool((want["flags"] & T.V_VIOLATION).any())
            assert bool(f[k] & 2) == bool((want["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)).any())
        break
    # more executions than one workgroup holds: ceil(R / 256) workgroups per candidate
    if not EMU:
        v, f = ctx.random_explore_candidates(_masks(cands[:3], len(events)), 600, lim, seed_base=7)
        for k in range(3):
            assert (v[k] == oracle.random_explore(model, events[list(cands[k])], 600, seed_base=7, limits=lim)).all()
    ctx.close()


def _failing_execution(oracle, n_events):
    model, events, lim = raft5_config4(n_events)
    v = oracle.random_explore(model, events, 4000, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count() or 1)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    return model, EventTrace(rec, used), ViolationFingerprint(int(vv.fingerprint), model.fp_match_mask), lim


class _OracleRandomScheduler:
    """RandomScheduler.test (:597-612) over the CPU oracle: Some(...) iff one of R interleavings of the subsequence reproduces."""

    def __init__(self, oracle, model, externals, lim, R, seed_base):
        self.o, self.model, self.ext, self.lim, self.R, self.seed_base = oracle, model, externals, lim, R, seed_base

    def getName(self):
        return "RandomScheduler"

    def test(self, events, fp, stats=None):
        v = self.o.random_explore(self.model, self.ext[list(events)], self.R, seed_base=self.seed_base, limits=self.lim)
        assert not (v["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)).any()
        return True if (v["flags"] & T.V_VIOLATION).any() else None


@pytest.mark.parametrize("specialize", [True, False])
def test_native_random_ddmin_equals_the_reference_loop_over_the_oracle(oracle, specialize):
    """demi_random_ddmin against DDMin.minimize written out (minification.DDMin) around the oracle's RandomScheduler: the MCS and
    every consultation in order; the speculative frontier must not change either, and `sequential` must consult launch by launch."""
    R = 100 if not EMU else 60
    model, trace, fp, lim0 = _failing_execution(oracle, 200 if not EMU else 90)
    ext = trace.original_externals
    lim = T.Limits(len(trace.events), 0, 128, 1, fp.code, 0)          # sched.setMaxMessages(trace.size), lookingFor = the violation
    ref = DDMin(_OracleRandomScheduler(oracle, model, ext, lim, R, SEED_BASE), checkUnmodifed=False)
    want = ref.minimize(UnmodifiedEventDag(ext), fp).get_all_events()
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    if specialize:
        ctx.model_specialize()
    ctx.trace_load(ext)
    for par in (T.RandomDdminParams(R, 0, 256), T.RandomDdminParams(R, 0, 16), T.RandomDdminParams(R, 2, 0), T.RandomDdminParams(R, sequential=1)):
        mcs, cons, batches, st = ctx.random_ddmin(lim, par, seed_base=SEED_BASE)
        assert tuple(mcs) == tuple(want)
        assert [(tuple(c), p) for c, p in cons] == [(tuple(c), p) for c, p in ref.consulted]
        assert st.consultations == len(ref.consulted) and st.mcs_len == len(want)
        assert 0 < len(want) < len(ext) and st.verified == 1 and len(ref.consulted) > 8
        assert st.replays == R * sum(batches) and len(batches) == st.launches      # every launched candidate ran R executions (verify_mcs last)
        if par.sequential:
            assert batches == [1] * (len(ref.consulted) + 1)
        else:
            assert st.launches < len(ref.consulted)
    ctx.close()


def test_python_entry_point_native_and_mirror_agree():
    """minification.randomDDMin(native=True) - what a Python host calls - against its own loop (native=False) on the device."""
    from oracle import oracle_py as O
    R = 50 if not EMU else 16
    model, trace, fp, _ = _failing_execution(O, 120 if not EMU else 40)
    a = randomDDMin(SchedulerConfig(model=model), trace, fp, max_executions=R, seed_base=SEED_BASE, p_max=128, native=True)
    b = randomDDMin(SchedulerConfig(model=model), trace, fp, max_executions=R, seed_base=SEED_BASE, p_max=128, native=False)
    assert tuple(a[0]) == tuple(b[0]) and a[1].consulted == [(tuple(c), p) for c, p in b[1].consulted]
    assert (a[2] is None) == (b[2] is None)
