"""K1 launches of 64 .. 2^18 schedules (config 2's raft5 table and trace, through demi_random_explore): the plain launch against the
SPREAD variant with the lanes per wave the host picks, and with forced ones.  ms per call, best of 5."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from demi_amd import _native
from demi_amd.apps import SEED_BASE, raft5_config2

model, events, limits = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
os.environ["DEMI_EXPERIMENT"] = "1"
ref = {}
for n in (64, 256, 1024, 4096, 16384, 65536, 262144):
    row = []
    for tag, env in (("plain", {"DEMI_K1_NO_SPREAD": "1"}), ("auto", {}), ("1", {"DEMI_K1_LANES_PER_WAVE": "1"}), ("4", {"DEMI_K1_LANES_PER_WAVE": "4"}),
                     ("16", {"DEMI_K1_LANES_PER_WAVE": "16"}), ("32", {"DEMI_K1_LANES_PER_WAVE": "32"})):
        for k in ("DEMI_K1_NO_SPREAD", "DEMI_K1_LANES_PER_WAVE"):
            os.environ.pop(k, None)
        os.environ.update(env)
        v = ctx.random_explore(n, limits, seed_base=SEED_BASE)
        if tag == "plain":
            ref[n] = v
        assert (v == ref[n]).all(), (n, tag)
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            ctx.random_explore(n, limits, seed_base=SEED_BASE)
            best = min(best, time.perf_counter() - t)
        row.append("%s %.3f" % (tag, best * 1e3))
    print("n=%7d  ms per call: %s" % (n, "  ".join(row)), flush=True)
