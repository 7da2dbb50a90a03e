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
for batch in (5, 256):
    srch = T.DporSearch(batch, 5000, 0, 1)
    ctx.comm_destroy()
    one = ctx.dpor_explore(par, srch)
    ctx.comm_create_host(rank, world, allgather)
    two = ctx.dpor_explore(par, srch)
    assert len(one[0]) == len(two[0]) and (one[0] == two[0]).all() and (one[1] == two[1]).all() and (one[2] == two[2]).all(), batch
    assert one[4].exhausted and two[4].exhausted
ctx.comm_destroy()
dist.barrier(); dist.destroy_process_group()
ctx.close()
print("rank", rank, "ok", n)
'''


def test_sharded_entry_points_two_ranks_on_one_gpu(tmp_path):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": root})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=500)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "ok" in o, o


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
           "--schedules", "65536", "--no-prewarm"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert "demi_comm_allgather_dev" in d["config"]["collective"]
    # both ranks' violations are in the merged set: rank 1 evaluates the indices [65536, 131072)
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "1", "--warmup", "0", "--schedules", "65536",
                          "--no-prewarm", "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600)
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d["violations_last_step"] > d1["violations_last_step"] > 0
