"""Round 6: config 5 (and config 3) in ROUNDS order with rounds of 16 384 .. 262 144 backtrack points.  k3_dpor's launch is as long as
one interleaving's execution whatever its width (DESIGN_HISTORY 0.2: 0.52-0.61 ms for ONE interleaving, 0.81 ms for 16 384), so wider
rounds are fewer rounds of nearly the same length.  Prints seconds / kernels / launches / violations per width."""
import json, sys, time
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from demi_amd import _native, types as T
from demi_amd.apps import shuffle8_dpor_config5, raft5_dpor_config3


def run(name, model, events, par, budget, widths):
    out = {}
    for batch in widths:
        ctx = _native.Context(0)
        ctx.model_load(model.to_struct()); ctx.model_specialize(); ctx.dpor_load(events)
        s = T.DporSearch(batch, budget, 0, 1, T.DPOR_ORDER_ROUNDS)
        try:
            bufs = _native.Context.dpor_buffers(budget)
            ctx.dpor_explore(par, s, buffers=bufs)
            best = None
            for _ in range(3):
                t = time.perf_counter()
                v, plen, rounds, vt, st = ctx.dpor_explore(par, s, buffers=bufs)
                dt = time.perf_counter() - t
                if best is None or dt < best[0]:
                    best = (dt, float(st.kernel_ms), int(st.launches))
            viol = int(np.count_nonzero(v["flags"] & T.V_VIOLATION))
            vh = len(np.unique(v["hash"][(v["flags"] & T.V_VIOLATION) != 0]))
            out[batch] = {"seconds": best[0], "kernel_ms": best[1], "launches": best[2], "interleavings": len(v), "violations": viol,
                          "distinct_violating": vh, "rate": len(v) / best[0], "exhausted": bool(st.exhausted)}
        except Exception as e:
            out[batch] = {"error": str(e)}
        print(name, batch, json.dumps(out[batch]), flush=True)
        ctx.close()
    return out


if __name__ == "__main__":
    res = {}
    m, ev, par, budget = shuffle8_dpor_config5()
    if len(sys.argv) > 1:           # one width of config 5 only (under the profiler)
        run("config5", m, ev, par, budget, [int(a) for a in sys.argv[1:]])
        sys.exit(0)
    res["config5"] = run("config5", m, ev, par, budget, [16384, 32768, 65536, 131072, 262144])
    r = raft5_dpor_config3()
    m3, ev3, par3 = r[0], r[1], r[2]
    res["config3"] = run("config3", m3, ev3, par3, 1 << 20, [16384, 32768, 65536, 131072])
    print(json.dumps({"r6_batch_sweep": res}))
