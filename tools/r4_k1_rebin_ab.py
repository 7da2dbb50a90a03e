"""Round 4: the re-binned K1 (k1_random_explore<.., REBIN>) against the plain specialised kernel, in ONE process: ms per 2^20
schedules of the headline workload (HIP events around 10 launches each, after a warm-up), and that the two write the same 2^20
verdicts.  Variants of the class function through DEMI_JIT_DEFINES (experiment knobs)."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DEMI_EXPERIMENT"] = "1"
import numpy as np, torch
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2
model, events, limits = raft5_config2()
n = 1 << 20
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(); sp = C.c_void_p(stream.cuda_stream)

def run(tag, rebin, defines=None, extra_env=None):
    os.environ["DEMI_K1_REBIN"] = "1" if rebin else "0"
    if defines: os.environ["DEMI_JIT_DEFINES"] = defines
    else: os.environ.pop("DEMI_JIT_DEFINES", None)
    for k, v in (extra_env or {}).items(): os.environ[k] = v
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
    out = torch.empty((n, 2), dtype=torch.int64, device=dev)
    t0 = time.time()
    ctx.random_explore_dev(n, limits, out.data_ptr(), seed_base=SEED_BASE, stream=sp); torch.cuda.synchronize()
    first = time.time() - t0
    while time.time() - t0 < 1.5:
        ctx.random_explore_dev(n, limits, out.data_ptr(), seed_base=SEED_BASE, stream=sp); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10):
        ctx.random_explore_dev(n, limits, out.data_ptr(), seed_base=SEED_BASE, stream=sp)
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    for k in (extra_env or {}): os.environ.pop(k, None)
    print("%-46s %.3f ms per 2^20 (first launch incl. compile %.2f s)" % (tag, ms, first), flush=True)
    return ms, out

res = {}
res["plain"], ref = run("plain specialised kernel", False)
variants = [("rebin: class = message type", None, None),
            ("rebin: one class (re-dealt, unsorted)", "DEMI_K1_RB_CLASS(W,S)=0", None),
            ("rebin: class = type*2 + (role==0)", "DEMI_K1_RB_CLASS(W,S)=((w_type(W)<<1)|(((S)&255u)==0u))", None),
            ("rebin: class = type*2 + (role==2)", "DEMI_K1_RB_CLASS(W,S)=((w_type(W)<<1)|(((S)&255u)==2u))", None),
            ("rebin: class = type, 4 WG/CU cap", None, {"DEMI_K1_MAX_WG_PER_CU": "4"}),
            ]
for tag, d, env in variants:
    try:
        ms, out = run(tag, True, d, env)
        same = bool(torch.equal(out, ref))
        res[tag] = {"ms": ms, "same_verdicts_as_plain": same}
        if not same: print("   !!! verdicts differ from the plain kernel", flush=True)
    except Exception as e:
        print("%-46s failed: %s" % (tag, e), flush=True)
res["plain_again"], _ = run("plain specialised kernel (again)", False)
print(json.dumps({"k1_rebin_ab": res}))
