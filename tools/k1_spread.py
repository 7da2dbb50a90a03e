"""Run-to-run spread of the specialised K1 inside ONE process: a fresh demi_ctx (fresh scratch allocation) per trial,
with allocations of other sizes in between.  Tells whether the two timing modes seen across processes (4.37 / 4.58 ms)
follow the placement of the scratch or the process."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2

model, events, limits = raft5_config2()
limits.p_max = 64
n = 1 << 20
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
sp = C.c_void_p(stream.cuda_stream)
keep = []
rng = np.random.default_rng(int(os.environ.get("SPREAD_SEED", "0")))
for trial in range(int(os.environ.get("SPREAD_TRIALS", "6"))):
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    ctx.model_specialize()
    verdicts = torch.empty((n, 2), dtype=torch.int64, device=dev)
    for _ in range(150 if trial == 0 else 20):
        ctx.random_explore_dev(n, limits, verdicts.data_ptr(), seed_base=SEED_BASE, stream=sp)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(30):
        ctx.random_explore_dev(n, limits, verdicts.data_ptr(), seed_base=SEED_BASE, stream=sp)
    e1.record(stream)
    torch.cuda.synchronize()
    print("trial", trial, "kernel_ms %.3f" % (e0.elapsed_time(e1) / 30), "out=%x" % verdicts.data_ptr(), flush=True)
    junk = torch.empty(int(rng.integers(1, 200)) << 20, dtype=torch.uint8, device=dev)     # shifts what the next ctx gets
    if trial % 2 == 0:
        keep.append(junk)
    ctx.close()
