"""GPU suite: the multi-GPU entry points of the C ABI (demi_comm_*, demi_random_explore_sharded, demi_replay_batch_sharded).
The driver's 8-GPU run is the only place with more than one GPU, so here (a) two processes share the one GPU and exchange
their blocks through a host-supplied all-gather (gloo) - the sharding logic and the C entry points, everything but the
RCCL call itself - and (b) the RCCL communicator is created and used with a world of one."""
import os
import subprocess
import sys

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd.apps import SEED_BASE, raft5_config2

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
if os.environ.get("DEMI_EMU") == "1":      # the CPU suite's run of this test (tests/test_emu_suite_cpu.py): same worker, emulated device
    import tests.conftest
import numpy as np
import torch
import torch.distributed as dist
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
model, events, lim = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events)
def allgather(block: bytes) -> bytes:
    mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return b"".join(bytes(p.numpy().tobytes()) for p in parts)
ctx.comm_create_host(rank, world, allgather)
assert ctx.comm_rank() == (rank, world)
N = 30000
got, n = ctx.random_explore_sharded(N, lim, seed_base=SEED_BASE)
ctx.comm_destroy()
want, n1 = ctx.random_explore_violations(N, lim, seed_base=SEED_BASE)
assert n == n1 == len(want) and (got == want).all(), (n, n1)
# K2: a frontier of candidate masks in blocks over the ranks
v = ctx.random_explore(2000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.replay_load(used, rec)
rng = np.random.default_rng(7)
masks = rng.integers(0, 1 << 63, size=(301, 4), dtype=np.uint64)
target = T.Limits(0, 0, 64, 1, vv.fingerprint, 0)
single = ctx.replay_batch(masks, target)
ctx.comm_create_host(rank, world, allgather)
sharded = ctx.replay_batch_sharded(masks, target)
assert (sharded == single).all()
# K3: the whole bounded exploration, rounds split over the ranks, explored-pair table owner-sharded
from demi_amd import model as M
from demi_amd.fuzzer import events_to_array, send, start
m3 = M.raft_model(3)
ev3 = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
ctx.comm_destroy()
ctx.model_load(m3.to_struct()); ctx.dpor_load(ev3)
par = T.DporParams(30, 0, 0, 0, 64, 4096)
# (rounds of 5: ragged blocks and padding ids at every world size, a budget of a few hundred interleavings - each round is
# three collectives over gloo; rounds of 256: the whole exploration, until the queue is empty)
for batch, budget in ((5, 400), (256, 5000)):
    srch = T.DporSearch(batch, budget, 0, 1)
    ctx.comm_destroy()
    one = ctx.dpor_explore(par, srch)
    ctx.comm_create_host(rank, world, allgather)
    two = ctx.dpor_explore(par, srch)
    assert len(one[0]) == len(two[0]) and (one[0] == two[0]).all() and (one[1] == two[1]).all() and (one[2] == two[2]).all(), batch
    assert len(one[0]) >= 400 and (batch == 5 or (one[4].exhausted and two[4].exhausted))
ctx.comm_destroy()
dist.barrier(); dist.destroy_process_group()
ctx.close()
print("rank", rank, "ok", n)
'''


def _run_ranks(tmp_path, text, world, port, timeout=900, per_rank_env=None, expect="ok"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / ("w%d.py" % world)
    script.write_text(text % {"root": root})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    if os.environ.get("DEMI_EMU") == "1":
        env["W64_THREADS"] = "1"     # (tests/emu/README: the explored-pair table's bounded wait assumes waves that keep running)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r), **((per_rank_env or {}).get(r, {}))),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout)[0].decode())
    finally:
        for p in procs:          # (a hung rank must not outlive the test)
            if p.poll() is None:
                p.kill()
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and expect in o, o
    return outs


def test_sharded_entry_points_two_ranks_on_one_gpu(tmp_path):
    _run_ranks(tmp_path, _WORKER, 2, 29541)


@pytest.mark.parametrize("world", [3, 8])
def test_sharded_entry_points_odd_and_full_world_sizes(tmp_path, world):
    """The same worker with 3 and with 8 ranks (the driver's 8-GPU shape, here on one device): an odd world exercises the
    W * ceil(n / W) padding of every block exchange - empty and ragged last blocks, arena ids that are padding - and the
    status agreement before each collective with more than two parties."""
    _run_ranks(tmp_path, _WORKER, world, 29550 + world)


def test_rccl_communicator_world_of_one(gpu_ctx):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather through the library's dlopen'ed RCCL (what the 8-GPU run uses),
    with one rank: the gathered block is the rank's own, and the sharded fuzz equals the plain one."""
    import ctypes as C
    import torch
    model, events, lim = raft5_config2()
    gpu_ctx.model_load(model.to_struct())
    gpu_ctx.trace_load(events)
    uid = gpu_ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    gpu_ctx.comm_create(uid, 0, 1)
    assert gpu_ctx.comm_rank() == (0, 1)
    dev = torch.device("cuda", 0)
    a = torch.arange(4096, dtype=torch.int64, device=dev)
    b = torch.zeros_like(a)
    gpu_ctx.comm_allgather_dev(a.data_ptr(), b.data_ptr(), a.numel() * 8, stream=C.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert bool((a == b).all())
    got, n = gpu_ctx.random_explore_sharded(20000, lim, seed_base=SEED_BASE)
    gpu_ctx.comm_destroy()
    want, n1 = gpu_ctx.random_explore_violations(20000, lim, seed_base=SEED_BASE)
    assert n == n1 and (got == want).all()


def test_bench_py_two_ranks_on_one_gpu(tmp_path):
    """The command the driver runs for its multi-GPU scaling bench - torch.distributed.run ... bench.py --gpus N - with N = 2
    on this box's one GPU: gloo for torch.distributed, both ranks on cuda:0, the library's communicator over the host
    all-gather callback (RCCL refuses two ranks on one device).  Everything but the RCCL transport itself: rank / world
    plumbing, index-range sharding, the all-gather of the violation sets through demi_comm_allgather_dev, the max over ranks,
    the one JSON line from rank 0."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEMI_BENCH_BACKEND="gloo", DEMI_BENCH_ONE_GPU="1", DEMI_BENCH_COMM="host")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--schedules", "65536", "--no-prewarm", "--config5-budget", "40000"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert "demi_comm_allgather_dev" in d["config"]["collective"]
    # every timed step evaluates fresh seeds: 3 steps x 2 ranks x 65536 schedules, their violating executions counted once each
    tr = d["timed_region"]
    assert tr["schedules"] == 3 * 2 * 65536 and tr["violating_executions"] >= tr["distinct_violating_delivery_hashes"] > 0
    assert tr["distinct_violating_delivery_hashes"] > 2 * d["violations_last_step"] and d["bugs_per_hr"] > 0
    assert d["config"]["jit_compile_s"] > 0
    # BASELINE configs 4 and 5 as N-rank records of the same line
    dd, c5 = d["secondary"]["ddmin"], d["secondary"]["config5"]
    assert "error" not in dd and "error" not in c5, (dd, c5)
    e2e = dd["ddmin_end_to_end"]
    assert dd["n_gpus"] == 2 and e2e["same_mcs_as_single_rank"] and e2e["same_consultation_sequence_as_single_rank"] and e2e["mcs_len"] > 0
    assert dd["every_ranks_block_arrived_everywhere"] and dd["value"] > 0
    rd = dd["random_ddmin_R100"]         # randomDDMin with every frontier split over the two ranks
    assert "error" not in rd, rd
    assert rd["n_gpus"] == 2 and rd["same_mcs_and_consultations_on_every_rank"] and rd["same_mcs_as_single_rank_sequential"] and rd["mcs_len"] > 0
    assert c5["n_gpus"] == 2 and c5["interleavings"] == 40000 and not c5["exhausted"] and c5["same_verdict_sequence_on_every_rank"]
    # both ranks' violations are in the merged set: rank 1 evaluates the indices [65536, 131072)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--schedules", "65536",
                          "--no-prewarm", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600)
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d["violations_last_step"] > d1["violations_last_step"] > 0


def test_bench_py_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO external launcher (the shape of the driver's 1-GPU command, with N = 2): bench.py
    re-runs itself under torch.distributed.run on a free port and rank 0's one JSON line arrives on the caller's stdout.
    Same one-GPU plumbing environment as the test above."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEMI_BENCH_BACKEND="gloo", DEMI_BENCH_ONE_GPU="1", DEMI_BENCH_COMM="host")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--schedules", "65536",
           "--no-prewarm", "--no-cpu-baseline", "--no-secondary"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["timed_region"]["schedules"] == 2 * 2 * 65536
    assert "demi_comm_allgather_dev" in d["config"]["collective"]


# BASELINE configs 4 and 5 over the ranks, exactly as bench.py --gpus N runs them: demi_ddmin with the communicator (every
# speculative frontier split over the ranks inside the library) and the bounded DPOR exploration of the shuffle pipeline
# (apps.shuffle8_config5_large; a smaller budget here), each against the single-rank call on the same inputs.
_WORKER_CFG45 = r'''
import os, sys
sys.path.insert(0, %(root)r)
if os.environ.get("DEMI_EMU") == "1":
    import tests.conftest
import numpy as np
import torch
import torch.distributed as dist
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config4, shuffle8_config5_large
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
def allgather(block: bytes) -> bytes:
    mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return b"".join(bytes(p.numpy().tobytes()) for p in parts)
emu = os.environ.get("DEMI_EMU") == "1"
# ---- config 4: DDMin of the 200-event failing execution
n_ev = 60 if emu else 200
model, events, lim = raft5_config4(n_ev)
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events)
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.replay_load(used, rec)
if not emu:
    ctx.model_specialize()
target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
mcs1, cons1, batches1, st1 = ctx.ddmin(target, T.DdminParams(0, 256, 1, 1))
ctx.comm_create_host(rank, world, allgather)
for budget in (256, 256 * world):
    mcsw, consw, batchesw, stw = ctx.ddmin(target, T.DdminParams(0, budget, 1, 1))
    assert tuple(mcsw) == tuple(mcs1) and stw.verified == st1.verified and stw.consultations == st1.consultations
    assert [(tuple(c), p) for c, p in consw] == [(tuple(c), p) for c, p in cons1]
    if budget == 256:
        assert batchesw == batches1          # same frontiers, each split over the ranks
    else:
        assert len(batchesw) <= len(batches1)
ctx.comm_destroy()
# ---- config 4 with the RandomScheduler as DDMin's oracle (randomDDMin, R executions per candidate): every frontier's candidates
# split over the ranks, their verdict bits all-gathered - same MCS, consultations and frontiers as the single-rank call
R = 24 if emu else 100
ctx.trace_load(used)
rl = T.Limits(len(rec), 0, 128, 1, vv.fingerprint, 0)
r1 = ctx.random_ddmin(rl, T.RandomDdminParams(R, 0, 64), seed_base=SEED_BASE)
ctx.comm_create_host(rank, world, allgather)
for budget in (64, 64 * world):
    rw = ctx.random_ddmin(rl, T.RandomDdminParams(R, 0, budget), seed_base=SEED_BASE)
    assert tuple(rw[0]) == tuple(r1[0]) and rw[3].verified == r1[3].verified and rw[3].consultations == r1[3].consultations
    assert [(tuple(c), p) for c, p in rw[1]] == [(tuple(c), p) for c, p in r1[1]]
    if budget == 64:
        assert rw[2] == r1[2]
        # this rank ran its blocks only: fewer executions than the single-rank call launched (odd frontiers: the last rank's block is short)
        assert rw[3].replays < r1[3].replays or max(r1[2]) == 1
ctx.comm_destroy()
# ---- config 5: the shuffle pipeline, bounded DPOR exploration
model5, ev5, depth, _budget = shuffle8_config5_large()
budget = int(os.environ.get("CFG5_BUDGET", "3000" if emu else "40000"))
ctx.model_load(model5.to_struct())
if not emu:
    ctx.model_specialize()
ctx.dpor_load(ev5)
par = T.DporParams(depth, 0, 0, 0, 64, 4096)
srch = T.DporSearch(1024 if emu else 4096, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
one = ctx.dpor_explore(par, srch)
ctx.comm_create_host(rank, world, allgather)
two = ctx.dpor_explore(par, srch)
assert len(one[0]) == len(two[0]) == budget and (one[0] == two[0]).all() and (one[1] == two[1]).all() and (one[2] == two[2]).all()
assert not one[4].exhausted and not two[4].exhausted and one[4].queue_len == two[4].queue_len
ctx.comm_destroy()
dist.barrier(); dist.destroy_process_group()
ctx.close()
print("rank", rank, "ok", len(mcs1), len(one[0]))
'''


@pytest.mark.parametrize("world", [2, 3])
def test_config4_ddmin_and_config5_dpor_over_the_ranks(tmp_path, world):
    _run_ranks(tmp_path, _WORKER_CFG45, world, 29560 + world, timeout=1500)


# One rank whose share of the explored-pair table is full (a 64-entry table on rank 1 only): every rank must return the
# capacity error from the same call - nobody is left waiting in the round's next all-gather (comm_agree).
_WORKER_FAIL = r'''
import os, sys
sys.path.insert(0, %(root)r)
if os.environ.get("DEMI_EMU") == "1":
    import tests.conftest
import numpy as np
import torch
import torch.distributed as dist
from demi_amd import _native, types as T, model as M
from demi_amd.fuzzer import events_to_array, send, start
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
def allgather(block: bytes) -> bytes:
    mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return b"".join(bytes(p.numpy().tobytes()) for p in parts)
m3 = M.raft_model(3)
ev3 = events_to_array([start(a) for a in range(3)] + [send(a, M.M_BOOTSTRAP) for a in range(3)])
ctx = _native.Context(0)
ctx.model_load(m3.to_struct()); ctx.dpor_load(ev3)
ctx.comm_create_host(rank, world, allgather)
par = T.DporParams(30, 0, 0, 0, 64, 4096)
try:
    ctx.dpor_explore(par, T.DporSearch(64, 5000, 0, 1))
    print("rank", rank, "no error")
except _native.DemiError as e:
    assert e.code == -7, e           # DEMI_ERR_CAPACITY on every rank
    print("rank", rank, "stopped", "own table" if "table full" in str(e) else "with the failing rank")
ctx.comm_destroy()
dist.barrier(); dist.destroy_process_group()
ctx.close()
'''


def test_one_ranks_full_table_stops_every_rank(tmp_path):
    outs = _run_ranks(tmp_path, _WORKER_FAIL, 3, 29571, timeout=600, expect="stopped",
                      per_rank_env={1: {"DEMI_EXPERIMENT": "1", "DEMI_K3_TABLE_ENTRIES": "64"}})
    assert "own table" in outs[1] and all("with the failing rank" in outs[r] for r in (0, 2)), outs
