import numpy as np, time, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from demi_amd import types as T, _native
from demi_amd.model import raft_model
from demi_amd.fuzzer import raft_trace, events_to_array
from oracle import oracle_py as O
m = raft_model(5)
ev = events_to_array(raft_trace(5,50,0xDE31))
ctx = _native.Context(0)
ctx.model_load(m.to_struct()); ctx.trace_load(ev)
for pmax in (64, 32, 128):
    lim = T.Limits(200, 30, pmax, 0, 0, 0)
    n = 20000
    g = ctx.random_explore(n, lim, seed_base=0x5EED0000)
    c = O.random_explore(m, ev, n, seed_base=0x5EED0000, limits=lim, n_threads=8)
    same = (g == c)
    print("pmax", pmax, "match", same.all(), "mismatches", (~same).sum(), "viol", (g['flags']&1).sum(), (c['flags']&1).sum())
    if not same.all():
        bad = np.nonzero(~same)[0][:5]
        for b in bad: print(b, g[b], c[b])
lim = T.Limits(200, 30, 64, 0, 0, 0)
for n in (1<<16, 1<<20):
    t=time.time(); g = ctx.random_explore(n, lim, seed_base=0x5EED0000); dt=time.time()-t
    print("n", n, "host-inclusive %.3fs -> %.3g schedules/s"%(dt, n/dt))
v, rec = ctx.random_get_trace(0x5EED0000+3, lim)
vc, recc, _ = O.random_execute(m, ev, 0x5EED0000+3, lim)
print("rec", len(rec), len(recc), (rec==recc).all() if len(rec)==len(recc) else None, v.flags, vc.flags, hex(v.hash), hex(vc.hash))
