"""Round 4: the convergence CEILING of K1.  How fast would the kernel be if the 64 lanes of a wave always did the same thing -
the best any re-binning of simulators over lanes / waves could achieve?  Launch 2^20 schedules whose seeds repeat in groups of
g consecutive indices (g = 1: the headline workload; g = 64: every wave runs 64 copies of ONE execution, i.e. perfectly
convergent control flow with unchanged memory traffic per lane) and time the launches."""
import ctypes as C, json, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
model, events, limits = raft5_config2()
n = 1 << 20
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
dev = torch.device("cuda", 0)
out = torch.empty((n, 2), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(); sp = C.c_void_p(stream.cuda_stream)
t0 = time.time()
while time.time() - t0 < 1.5:
    ctx.random_explore_dev(n, limits, out.data_ptr(), seed_base=SEED_BASE, stream=sp); torch.cuda.synchronize()
res = {}
for g in (1, 2, 4, 8, 16, 32, 64):
    seeds = (SEED_BASE + (np.arange(n, dtype=np.uint64) // np.uint64(g)) * np.uint64(g)).astype(np.uint64)
    d_seeds = torch.from_numpy(seeds.view(np.int64)).to(dev)
    for _ in range(3):
        ctx.random_explore_dev(n, limits, out.data_ptr(), d_seeds_ptr=C.c_void_p(d_seeds.data_ptr()), stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        ctx.random_explore_dev(n, limits, out.data_ptr(), d_seeds_ptr=C.c_void_p(d_seeds.data_ptr()), stream=sp)
    e1.record(stream); torch.cuda.synchronize()
    res[g] = e0.elapsed_time(e1) / 10
    print("lanes per distinct execution %2d : %.3f ms per 2^20 schedules" % (g, res[g]), flush=True)
print(json.dumps({"k1_convergence_ceiling_ms": res}))
