"""GPU parity suite for K1 (RandomScheduler executions).  Every call goes through the C ABI of
libdemi_gpu.so; the CPU oracle is only the checker.  Bit-exact bar: the 16-byte verdict (flags,
delivery count, traceIdx, fingerprint, FNV hash of every delivered message and every final actor
state) must be identical for every schedule."""
import json
import os

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft3_config1, raft5_config2
from demi_amd.fuzzer import (FuzzerWeights, events_to_array, kill, partition, raft_trace, send, start, unpartition,
                             wait_quiescence)
from demi_amd.model import Asm, build_model, load_model

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def both(gpu_ctx, oracle, model, events, n, lim, seed_base=SEED_BASE, seeds=None, jit=False):
    """GPU verdicts and oracle verdicts of the same schedules.  jit=True also runs the kernel compiled for the
    model's table (demi_model_specialize) and requires it to agree with the table interpreter bit for bit."""
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    g = gpu_ctx.random_explore(n, lim, seed_base=seed_base, seeds=seeds)
    if jit:
        assert not gpu_ctx.is_specialized()
        gpu_ctx.model_specialize()
        assert gpu_ctx.is_specialized()
        gj = gpu_ctx.random_explore(n, lim, seed_base=seed_base, seeds=seeds)
        assert_same(gj, g)
        gpu_ctx.model_specialize(False)
        assert not gpu_ctx.is_specialized()
    c = oracle.random_explore(model, events, n, seed_base=seed_base, seeds=seeds, limits=lim, n_threads=os.cpu_count())
    return g, c


def assert_same(g, c):
    if not (g == c).all():
        bad = np.nonzero(g != c)[0]
        raise AssertionError("%d of %d verdicts differ; first at %d: gpu=%s cpu=%s" % (len(bad), len(g), bad[0], g[bad[0]], c[bad[0]]))


def test_native_library_is_the_one_running(gpu_ctx):
    import ctypes
    from demi_amd import _native
    assert os.path.samefile(_native.LIB_PATH, os.path.join(os.path.dirname(_native.__file__), "libdemi_gpu.so"))
    maps = open("/proc/self/maps").read()
    assert "libdemi_gpu.so" in maps


@pytest.mark.parametrize("name", ["raft5_config2", "raft3_config1"])
def test_golden_fixtures_on_gpu(gpu_ctx, name):
    model = load_model(os.path.join(G, name + "_model.json"))
    meta = json.load(open(os.path.join(G, name + "_trace.json")))
    events = events_to_array([tuple(e) for e in meta["events"]])
    want = np.load(os.path.join(G, name + "_verdicts.npy"))
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    got = gpu_ctx.random_explore(len(want), T.Limits(*meta["limits"]), seed_base=meta["seed_base"])
    assert_same(got, want)


@pytest.mark.parametrize("p_max", [32, 64, 128])
def test_raft5_parity_all_capacities(gpu_ctx, oracle, p_max):
    model, events, lim = raft5_config2()
    lim.p_max = p_max
    g, c = both(gpu_ctx, oracle, model, events, 50000, lim, jit=True)
    assert_same(g, c)
    assert (g["flags"] & T.V_VIOLATION).sum() > 100
    if p_max == 32:
        assert (g["flags"] & T.V_PENDING_OVF).sum() > 0      # overflow is a verdict, identical on both sides


@pytest.mark.parametrize("lanes", [None, 1, 3, 8])
def test_small_launches_on_few_lanes_of_many_waves(gpu_ctx, oracle, monkeypatch, lanes):
    """The SPREAD variant of K1 (k1_random_explore.hpp): a launch far smaller than the chip runs on the first
    lanes of as many wavefronts as it fills.  lanes = None: as the host picks them for each size (on the MI355X 1 lane up to a few
    thousand schedules, more beyond, the plain launch from ~32 768 on); otherwise forced.  Interpreter and compiled table, against
    the plain launch (DEMI_K1_NO_SPREAD) and the oracle: the same verdicts whatever a launch's shape."""
    model, events, lim = raft5_config2()
    sizes = (1, 100, 1000, 5000, 20000) if lanes is None else (100, 3000)
    if os.environ.get("DEMI_EMU") == "1":
        sizes = sizes[:3]
    if lanes == 3:                      # (one of the forced cases under SrcDstFIFO: the variant has its own compiled kernel)
        lim = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 0, 0, lim.populate_all, T.STRATEGY_SRC_DST_FIFO)
    for n in sizes:
        monkeypatch.delenv("DEMI_K1_NO_SPREAD", raising=False)
        if lanes is None:
            monkeypatch.delenv("DEMI_K1_LANES_PER_WAVE", raising=False)
        else:
            monkeypatch.setenv("DEMI_K1_LANES_PER_WAVE", str(lanes))
        g, c = both(gpu_ctx, oracle, model, events, n, lim, jit=True)
        assert_same(g, c)
        monkeypatch.setenv("DEMI_K1_NO_SPREAD", "1")
        gpu_ctx.model_specialize()
        plain = gpu_ctx.random_explore(n, lim, seed_base=SEED_BASE)
        gpu_ctx.model_specialize(False)
        assert_same(plain, c)


def test_raft3_config1_and_fixed_model(gpu_ctx, oracle):
    model, events, lim = raft3_config1()
    g, c = both(gpu_ctx, oracle, model, events, 100, lim)
    assert_same(g, c)
    model5, events5, lim5 = raft5_config2()
    g, c = both(gpu_ctx, oracle, M.raft_model(5, buggy=False), events5, 20000, lim5)
    assert_same(g, c)
    assert not (g["flags"] & T.V_VIOLATION).any()


@pytest.mark.parametrize("limits", [(0, 0, 64), (0, 7, 64), (50, 1, 64), (1000, 30, 128), (200, 30, 64)])
def test_limits_matrix(gpu_ctx, oracle, limits):
    model, events, _ = raft5_config2()
    lim = T.Limits(limits[0], limits[1], limits[2], 0, 0, 0)
    g, c = both(gpu_ctx, oracle, M.raft_model(5, election_budget=2), events, 8000, lim, jit=(limits[1] in (0, 30)))
    assert_same(g, c)


def test_looking_for_and_populate_all(gpu_ctx, oracle):
    model, events, lim = raft5_config2()
    g, c = both(gpu_ctx, oracle, model, events, 20000, lim)
    fps, counts = np.unique(g["fingerprint"][g["fingerprint"] != 0], return_counts=True)
    target = int(fps[np.argmax(counts)])
    lim2 = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 1, target, 0)
    g2, c2 = both(gpu_ctx, oracle, model, events, 20000, lim2)
    assert_same(g2, c2)
    hit = (g2["flags"] & T.V_VIOLATION) != 0
    assert hit.sum() > 0 and (g2["fingerprint"][hit] == target).all()
    # a schedule that violated with another fingerprint keeps running under lookingFor
    assert hit.sum() < ((g["flags"] & T.V_VIOLATION) != 0).sum()
    # populate_all: a trace that never Starts actor 4 still creates it (setActorNamePropPairs)
    ev = events_to_array([start(a) for a in range(4)] + [send(a, M.M_BOOTSTRAP) for a in range(5)] + [wait_quiescence()])
    for pa in (0, 1):
        lim3 = T.Limits(200, 30, 64, 0, 0, pa)
        g3, c3 = both(gpu_ctx, oracle, model, ev, 4000, lim3)
        assert_same(g3, c3)


def test_explicit_seeds_ragged_sizes_and_partitioned_batches(gpu_ctx, oracle):
    model, events, lim = raft5_config2()
    rng = np.random.default_rng(5)
    seeds = rng.integers(0, 2 ** 63, size=3001, dtype=np.uint64)
    g, c = both(gpu_ctx, oracle, model, events, len(seeds), lim, seeds=seeds)
    assert_same(g, c)
    for n in (0, 1, 63, 64, 65, 255, 257, 1025):
        g, c = both(gpu_ctx, oracle, model, events, n, lim)
        assert len(g) == n
        assert_same(g, c)
    # verdict i depends only on (seed_base + i): any split of the index range gives the same array
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    whole = gpu_ctx.random_explore(10000, lim, seed_base=SEED_BASE)
    parts = [gpu_ctx.random_explore(hi - lo, lim, seed_base=SEED_BASE + lo) for lo, hi in ((0, 1), (1, 4097), (4097, 10000))]
    assert_same(np.concatenate(parts), whole)


def test_fault_heavy_and_edge_traces(gpu_ctx, oracle):
    model = M.raft_model(5, election_budget=2)
    lim = T.Limits(300, 10, 128, 0, 0, 0)
    w = FuzzerWeights(kill=0.15, send=0.3, wait_quiescence=0.15, partition=0.25, unpartition=0.15)
    for seed in (1, 2, 3):
        events = events_to_array(raft_trace(5, 80, seed, w, exact=False))
        g, c = both(gpu_ctx, oracle, model, events, 6000, lim, jit=True)
        assert_same(g, c)
    edge = [[], [wait_quiescence()], [start(0)], [send(0, M.M_BOOTSTRAP)],
            [start(0), kill(0), start(0), send(0, M.M_BOOTSTRAP), wait_quiescence(), kill(0), wait_quiescence()],
            [start(a) for a in range(5)] + [partition(0, 1), partition(1, 0), unpartition(0, 1)] +
            [send(a, M.M_BOOTSTRAP) for a in range(5)],
            [start(a) for a in range(5)] + [send(a, M.M_BOOTSTRAP) for a in range(5)] * 6]
    for ev in edge:
        g, c = both(gpu_ctx, oracle, model, events_to_array(ev), 500, lim)
        assert_same(g, c)
    # the longest trace the boundary accepts
    long_ev = [start(a) for a in range(5)] + [send(a, M.M_BOOTSTRAP) for a in range(5)]
    k = 0
    while len(long_ev) < T.MAX_EXT_EVENTS:
        long_ev.append(send(k % 5, M.M_CLIENT, k & 255) if k % 7 else wait_quiescence())
        k += 1
    g, c = both(gpu_ctx, oracle, model, events_to_array(long_ev), 3000, T.Limits(1000, 30, 128, 0, 0, 0), jit=True)
    assert_same(g, c)


def test_timer_semantics_models_on_gpu(gpu_ctx, oracle):
    """Repeating timers, cancels, queue capacities: the tiny models of the CPU suite, in bulk."""
    MSGS = [("Kick", T.MSG_EXTERNAL), ("Ping", T.MSG_INTERNAL), ("Tick", T.MSG_TIMER), ("RTick", T.MSG_TIMER)]
    K, PI, TI, RT = range(4)
    CNT = M.F[0]
    kick = Asm().eq(M.T0, M.P0, 0).skipz(M.T0, "a").trep(RT).tset(TI).halt().label("a")
    kick.eq(M.T0, M.P0, 1).skipz(M.T0, "b").tcancel(RT).tset(TI).tset(TI).halt().label("b")
    kick.eq(M.T0, M.P0, 2).skipz(M.T0, "c").tcancel(TI).bcast(PI, M.P0, 1).halt().label("c")
    kick.trep(RT).trep(RT).tcancel(TI).tset(TI)
    h = {(0, "Kick"): kick,
         (0, "Ping"): Asm().add(CNT, CNT, 1).lt(M.T0, CNT, 6).skipz(M.T0, "x").send(PI, M.SRC, CNT, 0).tset(TI).label("x"),
         (0, "Tick"): Asm().add(M.F[1], M.F[1], 1).and_(M.T0, M.F[1], 3).skipnz(M.T0, "y").tcancel(RT).label("y"),
         (0, "RTick"): Asm().add(M.F[2], M.F[2], 1).ge(M.T0, M.F[2], 14).skipz(M.T0, "z").mov(M.F[3], 1).label("z")}
    model = build_model("timers", 4, MSGS, h, [[0] * 8 for _ in range(4)], (T.INV_NEVER, 3, 1, 0))
    rng = np.random.default_rng(11)
    ev = [start(a) for a in range(4)]
    for i in range(120):
        r = rng.integers(0, 10)
        ev.append(wait_quiescence() if (r < 3 and ev[-1][0] != T.EV_WAIT_QUIESCENCE) else
                  send(int(rng.integers(0, 4)), K, int(rng.integers(0, 4))))
    for lim in (T.Limits(400, 5, 64, 0, 0, 0), T.Limits(250, 0, 32, 0, 0, 0), T.Limits(0, 0, 64, 0, 0, 0)):
        g, c = both(gpu_ctx, oracle, model, events_to_array(ev), 8000, lim, jit=True)
        assert_same(g, c)
        assert len(np.unique(g["hash"])) > 7000
        if lim.max_messages != 250:
            assert (g["flags"] & T.V_VIOLATION).any()
    # a model that overflows the timer queue: 9 distinct sets in one handler is impossible with 4 timer
    # types, so overflow comes from duplicates of one-shot timers
    many = Asm()
    for _ in range(9):
        many.tset(TI)
    model2 = build_model("tq", 2, MSGS, {(0, "Kick"): many}, [[0] * 8] * 2, (T.INV_NEVER, 3, 1, 0))
    g, c = both(gpu_ctx, oracle, model2, events_to_array([start(0), send(0, K)]), 64, T.Limits(0, 0, 64, 0, 0, 0), jit=True)
    assert_same(g, c)
    assert (g["flags"] == T.V_QUEUE_OVF).all()


def test_recorded_event_traces_are_identical(gpu_ctx, oracle):
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    v = gpu_ctx.random_explore(2000, lim, seed_base=SEED_BASE)
    hits = np.nonzero(v["flags"] & T.V_VIOLATION)[0][:6]
    for i in list(hits) + [0, 1, 2, 1999]:
        gv, grec = gpu_ctx.random_get_trace(SEED_BASE + int(i), lim)
        cv, crec, _ = oracle.random_execute(model, events, SEED_BASE + int(i), lim)
        assert (gv.flags, gv.fingerprint, gv.hash) == (cv.flags, cv.fingerprint, cv.hash)
        assert (gv.flags, gv.hash) == (int(v["flags"][i]), int(v["hash"][i]))
        assert len(grec) == len(crec) and (grec == crec).all()
        # every delivery pairs with an earlier send of the same id; dropped sends are never delivered
        sends = {int(e["id"]): e for e in grec if e["kind"] == T.REC_MSG_SEND}
        for e in grec:
            if e["kind"] == T.REC_MSG_EVENT:
                s = sends[int(e["id"])]
                assert not (s["flags"] & 4) and (s["rcv"], s["msg_type"], s["p0"], s["p1"]) == (e["rcv"], e["msg_type"], e["p0"], e["p1"])


def test_scheduler_mirror_explore_and_test(oracle):
    from demi_amd.schedulers import MinimizationStats, RandomScheduler, SchedulerConfig, ViolationFingerprint
    model, events, lim = raft5_config2()
    with pytest.raises(ValueError):
        s0 = RandomScheduler(SchedulerConfig(model=None), max_executions=4)
        try:
            s0.explore(events)
        finally:
            s0.shutdown()
    sched = RandomScheduler(SchedulerConfig(model=model), max_executions=2000, invariant_check_interval=30, seed_base=SEED_BASE)
    sched.setMaxMessages(200)
    found = sched.explore(events)
    assert found is not None
    trace, fp = found
    want = oracle.random_explore(model, events, 2000, seed_base=SEED_BASE, limits=lim)
    first = int(np.nonzero(want["flags"] & T.V_VIOLATION)[0][0])
    assert fp.code == int(want["fingerprint"][first])
    assert len(trace.original_externals) == T.verdict_trace_idx(int(want["flags"][first]))
    # TestOracle.test: reproduces the same violation, counts replays like increment_replays()
    stats = MinimizationStats()
    assert sched.test(events, fp, stats) is not None and stats.total_replays == 2000
    other = ViolationFingerprint(fp.code ^ 0x0100)       # same nodes, another term: never matches here
    got = sched.test(events, other, stats)
    want2 = oracle.random_explore(model, events, 2000, seed_base=SEED_BASE,
                                  limits=T.Limits(200, 30, 64, 1, other.code, 0))
    assert (got is not None) == bool((want2["flags"] & T.V_VIOLATION).any())
    sched.shutdown()


def test_full_size_properties_1m(gpu_ctx, oracle):
    """BASELINE config 2 at full size (2^20 schedules): properties that do not need the oracle for
    every index, plus an oracle spot check on random windows."""
    import ctypes as C
    import torch
    model, events, lim = raft5_config2()
    n = 1 << 20
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    dev = torch.device("cuda", 0)
    out = torch.empty((n, 2), dtype=torch.int64, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gpu_ctx.random_explore_dev(n, lim, out.data_ptr(), seed_base=SEED_BASE, stream=sp)
    torch.cuda.synchronize()
    a = out.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1).copy()
    out.zero_()
    gpu_ctx.random_explore_dev(n, lim, out.data_ptr(), seed_base=SEED_BASE, stream=sp)   # idempotent / deterministic
    torch.cuda.synchronize()
    b = out.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1).copy()
    assert_same(a, b)
    out.zero_()
    gpu_ctx.model_specialize()           # the same 2^20 schedules through the kernel compiled for this table
    gpu_ctx.random_explore_dev(n, lim, out.data_ptr(), seed_base=SEED_BASE, stream=sp)
    torch.cuda.synchronize()
    b = out.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)
    assert_same(a, b)
    gpu_ctx.model_specialize(False)
    # all 2^20 of them are the verdicts of the Scala RandomScheduler as transliterated (tools/check_fuzz_transliteration.py ran it
    # over the whole step; the CPU suite holds the C oracle against the same record)
    import hashlib
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_config2_transliteration.json")) as f:
        translit = json.load(f)
    assert translit["seed_base"] == SEED_BASE and translit["equals_the_oracle"] is True
    assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == translit["sha256_verdicts_of_the_first"][str(n)]
    flags = a["flags"]
    assert not (flags & (T.V_PENDING_OVF | T.V_QUEUE_OVF)).any()
    deliveries = (flags >> 16) & 0xFFFF
    assert deliveries.max() == 201 and ((deliveries == 201) == ((flags & T.V_MAXMSG) != 0)).all()
    viol = (flags & T.V_VIOLATION) != 0
    assert ((a["fingerprint"] != 0) == viol).all() and not (viol & ((flags & T.V_MAXMSG) != 0)).any()
    assert len(np.unique(a["hash"])) > 0.999 * n            # every schedule is a distinct interleaving
    # the device compaction kernel returns exactly the violating set
    cap = 1 << 16
    lst = torch.zeros((cap + 1, 2), dtype=torch.int64, device=dev)
    gpu_ctx.collect_violations_dev(out.data_ptr(), n, 7_000_000, lst[1:].data_ptr(), cap, lst[0:1].data_ptr(), stream=sp)
    torch.cuda.synchronize()
    from demi_amd.distributed import merge_violation_sets
    got = merge_violation_sets([lst.cpu().numpy()], cap)
    idx = np.nonzero(viol)[0]
    assert len(got) == len(idx) and (got["index"] == idx + 7_000_000).all() and (got["fingerprint"] == a["fingerprint"][idx]).all()
    # oracle spot check on 8 random windows of 4096
    rng = np.random.default_rng(0)
    for lo in rng.integers(0, n - 4096, size=8):
        c = oracle.random_explore(model, events, 4096, seed_base=SEED_BASE + int(lo), limits=lim, n_threads=os.cpu_count())
        assert_same(a[lo:lo + 4096], c)


def test_shuffle8_three_actor_classes(gpu_ctx, oracle):
    """BASELINE config 5's application (8 actors, 3 classes): class-indexed handler lookup."""
    from demi_amd.apps import shuffle8_config5
    model, _, events, lim = shuffle8_config5()
    g, c = both(gpu_ctx, oracle, model, events, 30000, lim, jit=True)
    assert_same(g, c)
    assert (g["flags"] & T.V_VIOLATION).sum() > 500
    g, c = both(gpu_ctx, oracle, M.shuffle_model(buggy=False), events, 30000, lim, jit=True)
    assert_same(g, c)
    assert not (g["flags"] & T.V_VIOLATION).any()


def test_violation_set_entry_point(gpu_ctx, oracle):
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    want = oracle.random_explore(model, events, 50000, seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
    idx = np.nonzero(want["flags"] & T.V_VIOLATION)[0]
    got, n = gpu_ctx.random_explore_violations(50000, lim, seed_base=SEED_BASE)
    assert n == len(idx) == len(got) and (got["index"] == idx).all()
    assert (got["fingerprint"] == want["fingerprint"][idx]).all() and (got["flags"] == want["flags"][idx]).all()
    few, n2 = gpu_ctx.random_explore_violations(50000, lim, seed_base=SEED_BASE, cap=8)
    assert n2 == len(idx) and len(few) == 8 and set(few["index"]) <= set(idx)
    none, n3 = gpu_ctx.random_explore_violations(0, lim, seed_base=SEED_BASE)
    assert n3 == 0 and len(none) == 0


# ----------------------------------------------------------------------------------------------
# SrcDstFIFO randomization strategy (RandomScheduler.scala:702-909)
def _fifo(lim):
    return T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, lim.looking_for_valid, lim.looking_for,
                    lim.populate_all, T.STRATEGY_SRC_DST_FIFO)


@pytest.mark.parametrize("p_max", [24, 64, 128])
def test_srcdst_fifo_parity_raft5(gpu_ctx, oracle, p_max):
    model, events, lim = raft5_config2()
    lim.p_max = p_max
    g, c = both(gpu_ctx, oracle, model, events, 40000, _fifo(lim), jit=True)
    assert_same(g, c)
    plain = gpu_ctx.random_explore(40000, lim, seed_base=SEED_BASE)
    assert (g["hash"] != plain["hash"]).mean() > 0.9              # a different delivery discipline
    assert (g["flags"] & T.V_VIOLATION).sum() > 10
    if p_max == 24:
        assert (g["flags"] & T.V_PENDING_OVF).sum() > 0


def test_srcdst_fifo_parity_other_models_and_traces(gpu_ctx, oracle):
    from .test_srcdst_fifo_cpu import gossip_model
    rng = np.random.default_rng(1)
    ev = [start(a) for a in range(4)]
    for i in range(60):
        ev.append(wait_quiescence() if rng.integers(0, 6) == 0 and ev[-1][0] != T.EV_WAIT_QUIESCENCE
                  else send(int(rng.integers(0, 4)), 0, int(rng.integers(0, 8))))
    for lim in (T.Limits(300, 0, 128, 0, 0, 0), T.Limits(120, 7, 40, 0, 0, 0)):
        g, c = both(gpu_ctx, oracle, gossip_model(), events_to_array(ev), 12000, _fifo(lim), jit=True)
        assert_same(g, c)
    # kills / partitions, the 8-actor 3-class application, explicit seeds
    model = M.raft_model(5, election_budget=2)
    w = FuzzerWeights(kill=0.15, send=0.3, wait_quiescence=0.15, partition=0.25, unpartition=0.15)
    for seed in (1, 2):
        events = events_to_array(raft_trace(5, 80, seed, w, exact=False))
        g, c = both(gpu_ctx, oracle, model, events, 6000, _fifo(T.Limits(300, 10, 128, 0, 0, 0)))
        assert_same(g, c)
    from demi_amd.apps import shuffle8_config5
    model, _, events, lim = shuffle8_config5()
    g, c = both(gpu_ctx, oracle, model, events, 20000, _fifo(lim), jit=True)
    assert_same(g, c)
    seeds = rng.integers(0, 1 << 62, size=3001, dtype=np.uint64)
    g, c = both(gpu_ctx, oracle, model, events, 3001, _fifo(lim), seeds=seeds)
    assert_same(g, c)


def test_srcdst_fifo_recorded_traces_and_scheduler_mirror(gpu_ctx, oracle):
    """The recording variant under SrcDstFIFO: identical EventTrace, deliveries in per-pair send order; the
    RandomScheduler mirror takes the strategy object."""
    from demi_amd.schedulers import RandomScheduler, SchedulerConfig, SrcDstFIFO
    model, events, lim = raft5_config2()
    fl = _fifo(lim)
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    v = gpu_ctx.random_explore(3000, fl, seed_base=SEED_BASE)
    idx = list(np.nonzero(v["flags"] & T.V_VIOLATION)[0][:3]) + [0, 1, 2999]
    for i in idx:
        gv, grec = gpu_ctx.random_get_trace(SEED_BASE + int(i), fl)
        cv, crec, _ = oracle.random_execute(model, events, SEED_BASE + int(i), fl)
        assert gv.flags == cv.flags and gv.hash == cv.hash == int(v[i]["hash"])
        assert len(grec) == len(crec) and (grec == crec).all()
    sched = RandomScheduler(SchedulerConfig(model=model), max_executions=3000, invariant_check_interval=30,
                            randomizationStrategy=SrcDstFIFO(SEED_BASE))
    sched.setMaxMessages(200)
    assert_same(sched.explore_all(events), v)
    sched.shutdown()


def test_fifo_golden_fixture_on_gpu(gpu_ctx):
    model, events, lim = raft5_config2()
    want = np.load(os.path.join(G, "raft5_config2_fifo_verdicts.npy"))
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    assert_same(gpu_ctx.random_explore(len(want), _fifo(lim), seed_base=SEED_BASE), want)


@pytest.mark.parametrize("seed", [1, 11, 13])
def test_random_programs_interpreter_specialised_and_oracle_agree(gpu_ctx, oracle, seed):
    """Random transition tables (every op, random forward control flow, two actor classes, timers): the table
    interpreter, the kernel compiled from the table and the oracle give the same verdict for every schedule."""
    from .test_jit_cpu import _random_handler
    rng = np.random.default_rng(seed)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for cls in range(2):
        for name, _ in MSGS:
            if rng.integers(5):
                h[(cls, name)] = _random_handler(rng, int(rng.integers(3, 30)), len(MSGS), few_effects=True)
    model = M.build_model("rand%d" % seed, 5, MSGS, h, [[0] * 8] * 5, (T.INV_NEVER, 0, 200, 0),
                          actor_class=[0, 1, 0, 1, 1], n_classes=2)
    ev = [start(a) for a in range(5)]
    for i in range(40):
        ev.append(wait_quiescence() if rng.integers(0, 7) == 0 and ev[-1][0] != T.EV_WAIT_QUIESCENCE
                  else send(int(rng.integers(0, 5)), 0, int(rng.integers(0, 256))))
    g, c = both(gpu_ctx, oracle, model, events_to_array(ev), 4000, T.Limits(150, 9, 64, 0, 0, 0), jit=True)
    assert_same(g, c)
    assert len(np.unique(g["hash"])) > 900
    gf, cf = both(gpu_ctx, oracle, model, events_to_array(ev), 2000, _fifo(T.Limits(60, 0, 20, 0, 0, 0)), jit=True)
    assert_same(gf, cf)


def test_launches_of_one_ctx_on_two_streams_are_ordered(oracle):
    """The *_dev entry points do not synchronise and a ctx's launches share its work counter and scratch: launches that
    alternate between two streams must still each evaluate exactly their own schedules (include/demi_gpu.h: a launch on
    another stream waits for the ctx's previous one), and a trace (re)load waits for the launch in flight."""
    import ctypes as C
    import torch
    from demi_amd import _native
    model, events, lim = raft5_config2()
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        n, k = 1 << 17, 6
        want = [ctx.random_explore(n, lim, seed_base=SEED_BASE + j * n) for j in range(k)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.zeros((n, 2), dtype=torch.int64, device="cuda") for _ in range(k)]
        for j in range(k):
            ctx.random_explore_dev(n, lim, outs[j].data_ptr(), seed_base=SEED_BASE + j * n, stream=C.c_void_p(streams[j % 2].cuda_stream))
        ctx.trace_load(events[:10])                 # must not disturb the launches in flight
        torch.cuda.synchronize()
        for j in range(k):
            assert_same(outs[j].cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1), want[j])
    finally:
        ctx.close()


def test_code_id_names_the_compiled_kernel(oracle):
    """demi_model_code_id: 0 while the table is interpreted, the same value for the same table in two contexts, another one
    for another table (bench.py quotes hardware counters only for the build they were taken on)."""
    from demi_amd import _native
    model, events, lim = raft5_config2()
    ids = []
    for m in (model, model, M.raft_model(3)):
        ctx = _native.Context(0)
        try:
            ctx.model_load(m.to_struct())
            assert ctx.code_id() == 0
            ctx.model_specialize()
            ids.append(ctx.code_id())
            ctx.model_specialize(False)
            assert ctx.code_id() == 0
        finally:
            ctx.close()
    assert ids[0] != 0 and ids[0] == ids[1] and ids[2] not in (0, ids[0])


@pytest.mark.parametrize("strategy", [T.STRATEGY_FULLY_RANDOM, T.STRATEGY_SRC_DST_FIFO])
def test_carried_generator_instances(gpu_ctx, oracle, strategy):
    """demi_limits.executions_per_instance = k: one lane = one `new RandomScheduler(config, max_executions = k)` whose
    generator(s) run on through its executions (RandomScheduler.scala:248-269, 575-595, 649-651), lookingFor on the first
    execution only (:586), nothing run behind the first violating execution.  Kernel = oracle on every verdict (the oracle's
    chain is pinned to the Scala explore loop in tests/test_random_scheduler_transliteration_cpu.py), with a verdict count
    that is not a multiple of k, explicit per-instance seeds, a lookingFor, and the recorded trace of a chained execution."""
    model, events, lim0 = raft5_config2()
    k, n = 7, 20003
    for looking in (0, 1):
        fp = 0
        if looking:                                   # a fingerprint that occurs: from the independent executions
            ind = oracle.random_explore(model, events, 400, seed_base=SEED_BASE, limits=lim0, n_threads=os.cpu_count())
            fp = int(ind["fingerprint"][np.nonzero(ind["flags"] & T.V_VIOLATION)[0][0]])
        lim = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, looking, fp, 0, strategy, 0, k)
        g, c = both(gpu_ctx, oracle, model, events, n, lim)
        assert_same(g, c)
        viol = np.nonzero(g["flags"] & T.V_VIOLATION)[0]
        assert len(viol) > 20
        not_run = (g["flags"] == 0) & (g["hash"] == 0)
        assert not_run.sum() > 0                      # executions behind a violating one
        for i in viol[:50]:                           # ... and exactly those
            end = min((i // k + 1) * k, n)
            assert not_run[i + 1:end].all() and not not_run[(i // k) * k:i + 1].any()
    # explicit seeds are per instance
    seeds = np.arange(1000, 1000 + (n + k - 1) // k, dtype=np.uint64) * np.uint64(7919)
    lim = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, strategy, 0, k)
    g, c = both(gpu_ctx, oracle, model, events, n, lim, seeds=seeds)
    assert_same(g, c)
    # a chained execution differs from the independent execution of the same index
    lim1 = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, strategy)
    ind = gpu_ctx.random_explore(2 * k, lim1, seed_base=SEED_BASE)
    lim = T.Limits(lim0.max_messages, lim0.invariant_check_interval, 64, 0, 0, 0, strategy, 0, k)
    chained = gpu_ctx.random_explore(2 * k, lim, seed_base=SEED_BASE)
    assert chained[0] == ind[0] and chained[k] == ind[1] and chained[1] != ind[1]
    # the recorded trace of execution e of an instance
    for e in (0, 3):
        v, rec, ran = gpu_ctx.random_get_trace_carried(SEED_BASE + 2, e, lim)
        vo, reco, rano = oracle.random_execute_carried(model, events, SEED_BASE + 2, e, lim)
        assert ran == rano and (int(v.flags), int(v.fingerprint), int(v.hash)) == (int(vo.flags), int(vo.fingerprint), int(vo.hash))
        assert len(rec) == len(reco) and (rec == reco).all()


@pytest.mark.gpu
def test_first_schedules_against_the_random_scheduler_transliterations_record(gpu_ctx):
    """The bench's fixed-seed step again, through the host-buffer entry point and without torch (so that the wave64 emulator runs it
    too): the first 2^17 schedules (2^14 on the emulator), interpreter and compiled table, are byte for byte the verdicts the
    transliteration of the Scala RandomScheduler produced (tests/golden/fuzz_config2_transliteration.json,
    tools/check_fuzz_transliteration.py).  (Checked once by hand on the emulator for all 2^20: both kernels, the same SHA-256.)"""
    import hashlib
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_config2_transliteration.json")) as f:
        rec = json.load(f)
    n = 1 << 14 if os.environ.get("DEMI_EMU") == "1" else 1 << 17
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    for specialised in (False, True):
        if specialised:
            gpu_ctx.model_specialize()
        v = gpu_ctx.random_explore(n, lim, seed_base=SEED_BASE)
        assert hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() == rec["sha256_verdicts_of_the_first"][str(n)], specialised
    gpu_ctx.model_specialize(False)


def test_two_calls_in_flight_in_one_context(oracle):
    """demi_random_explore_submit / _wait: explore() in pieces, up to three calls outstanding in ONE context, two of their kernels
    running at a time (a second set of K1's per-launch scratch, two streams of the context's own).  Every call returns what
    demi_random_explore_flagged returns for its seeds and, on request, all its verdicts = the oracle's; a fourth submit is
    refused until a ticket is waited for; tickets are single-use; a (re)load of the trace waits for the calls in flight."""
    from demi_amd import _native
    model, events, lim = raft5_config2()
    ctx = _native.Context(0)
    try:
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        n, k = (1 << 12, 5) if os.environ.get("DEMI_EMU") == "1" else (1 << 18, 7)
        mask = T.V_VIOLATION | T.V_PENDING_OVF | T.V_QUEUE_OVF
        want_flagged = [ctx.random_explore_flagged(n, lim, mask, seed_base=SEED_BASE + j * n) for j in range(k)]
        # (every verdict asked for at the submit for call 0 only: the last call's are fetched by its wait)
        tickets = [ctx.random_explore_submit(n, lim, seed_base=SEED_BASE + j * n, flag_mask=mask, want_verdicts=(j == 0)) for j in range(3)]
        assert len(set(tickets)) == 3 and all(t > 0 for t in tickets)
        with pytest.raises(_native.DemiError):
            ctx.random_explore_submit(n, lim, seed_base=SEED_BASE, flag_mask=mask)       # three are outstanding
        for j in range(k):
            out = np.zeros(n, dtype=T.VERDICT_DTYPE) if j in (0, k - 1) else None
            hits, cnt, first = ctx.random_explore_wait(tickets[j], out=out)
            with pytest.raises(_native.DemiError):
                ctx.random_explore_wait(tickets[j])                  # a ticket is waited for once
            if j + 3 < k:
                tickets.append(ctx.random_explore_submit(n, lim, seed_base=SEED_BASE + (j + 3) * n, flag_mask=mask))
            wh, wc, wf = want_flagged[j]
            assert cnt == wc and first == wf and len(hits) == len(wh) and (hits == wh).all(), j
            if out is not None:
                assert_same(out, oracle.random_explore(model, events, n, seed_base=SEED_BASE + j * n, limits=lim, n_threads=os.cpu_count()))
        # a call in flight when the trace is replaced: the load waits, the call's answer is the old trace's
        t = ctx.random_explore_submit(n, lim, seed_base=SEED_BASE, flag_mask=mask)
        ctx.trace_load(events[:10])
        hits, cnt, first = ctx.random_explore_wait(t)
        assert cnt == want_flagged[0][1] and (hits == want_flagged[0][0]).all()
    finally:
        ctx.close()


def test_explore_in_calls_is_explore_in_one_call(oracle):
    """RandomScheduler.explore (the loop GpuRandomScheduler.explore in scala/ runs): the executions in calls of `chunk`, two
    submitted ahead of the one waited for, nothing submitted beyond those behind the first violating call - the same first violating execution, trace and fingerprint as
    ONE call over all executions (rounds 1-5), on a trace whose first violation lies several calls in, and None alike when
    there is none."""
    from demi_amd.schedulers import RandomScheduler, SchedulerConfig
    model, events, lim = raft5_config2()
    n = 40000
    want = oracle.random_explore(model, events, n, seed_base=SEED_BASE + 7000, limits=lim, n_threads=os.cpu_count())
    viol = np.nonzero(want["flags"] & T.V_VIOLATION)[0]
    assert len(viol) and viol[0] >= 48
    chunk = int(viol[0]) // 3                       # the first violating execution is in the fourth call
    sched = RandomScheduler(SchedulerConfig(model=model), max_executions=n, invariant_check_interval=30, seed_base=SEED_BASE + 7000)
    sched.setMaxMessages(200)
    sched.chunk = chunk
    a = sched.explore(events)
    assert sched.last_calls in (4, 5, 6)             # (the fourth call, and the two already submitted behind it)
    ev = sched._prepare(events)
    b = sched._explore_one_call(ev, None)
    assert a is not None and b is not None
    assert (a[0].events == b[0].events).all() and a[1].code == b[1].code == int(want["fingerprint"][viol[0]])
    sched.shutdown()
    fixed = M.raft_model(5, buggy=False)
    s2 = RandomScheduler(SchedulerConfig(model=fixed), max_executions=5000, invariant_check_interval=30, seed_base=SEED_BASE)
    s2.setMaxMessages(200)
    s2.chunk = 1024
    assert s2.explore(events) is None and s2.last_calls == 5
    s2.shutdown()
