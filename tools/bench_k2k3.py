#!/usr/bin/env python
"""Throughput of K2 (STSSched replays/s) and K3 (DPOR interleavings/s, kernel + host loop) on one
MI355X, for DESIGN.md §5.  Not the headline bench (that is bench.py / K1)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import _native, types as T  # noqa: E402
from demi_amd.apps import SEED_BASE, raft5_config3, raft5_config4  # noqa: E402
from demi_amd.dpor import DPORwHeuristics  # noqa: E402
from demi_amd.minification import events_to_mask, stsSchedDDMin  # noqa: E402
from demi_amd.schedulers import EventTrace, STSScheduler, SchedulerConfig, ViolationFingerprint  # noqa: E402

out = {}
ctx = _native.Context(0)
model, events, lim = raft5_config4()
ctx.model_load(model.to_struct()); ctx.trace_load(events)
v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
used = events[:T.verdict_trace_idx(vv.flags)]
ctx.replay_load(used, rec)
rng = np.random.default_rng(0)
for n in (1024, 65536, 1 << 20):
    masks = np.zeros((n, 4), dtype=np.uint64)
    keep = rng.random((n, len(used))) < 0.7
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
    ctx.replay_batch(masks[:64], target)
    t = time.perf_counter(); r = ctx.replay_batch(masks, target); dt = time.perf_counter() - t
    out["k2_replays_per_s_n%d" % n] = n / dt
ctx.model_specialize()
ctx.replay_batch(masks[:64], target)          # compiles K2 for this table
t = time.perf_counter(); rj = ctx.replay_batch(masks, target); dt = time.perf_counter() - t
out["k2_replays_per_s_n%d_specialised" % n] = n / dt
out["k2_specialised_bit_identical"] = bool((rj == r).all())
ctx.model_specialize(False)
# CPU baseline beside it: the oracle's STSSched restatement on the host cores, same candidates (bounded sample)
from oracle import oracle_py as O  # noqa: E402
cores = os.cpu_count() or 1
sample = masks[:262144]
t = time.perf_counter(); c = O.sts_replay_batch(model, used, rec, sample, target, n_threads=cores); dt = time.perf_counter() - t
out["k2_cpu_baseline"] = {"value": len(sample) / dt, "unit": "replays/s", "cores": cores, "kind": "port",
                          "sample": "first %d of the same candidate masks" % len(sample),
                          "bit_identical_to_gpu": bool((c == r[:len(sample)]).all())}
# algorithmic bytes per replay: 32 B mask in + 16 B verdict out + the expected events (8 B each) read once per lane
out["k2_algorithmic_bytes_per_replay"] = 48 + 8 * int(sum(1 for e in rec if e["kind"] in (0, 1, 2, 3, 7) or (e["kind"] == 6 and e["flags"] & 1)))
out["k2_original_trace"] = {"externals": int(len(used)), "recorded_events": int(len(rec)), "deliveries": T.verdict_deliveries(vv.flags)}
# DDMin end to end (config 4)
sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used), p_max=128)
t = time.perf_counter()
mcs, d, ver = stsSchedDDMin(sts, used, ViolationFingerprint(vv.fingerprint), speculative_depth=4)
out["ddmin_config4"] = {"seconds": time.perf_counter() - t, "mcs_len": len(mcs), "oracle_consultations": len(d.consulted),
                        "launches": len(d.batches), "replays_launched": d.speculative_replays}
verified = sts.executed_trace(mcs, ViolationFingerprint(vv.fingerprint))
sts.shutdown()
# internal-event minimization of the verified MCS execution (config 4), both removal strategies
from demi_amd import internal_minimization as IM  # noqa: E402
for name in ("LeftToRightOneAtATime", "SrcDstFIFORemoval"):
    orc = IM.StsRemovalOracle(SchedulerConfig(model=model), p_max=128)
    mz = IM.STSSchedMinimizer(verified.original_externals, verified, ViolationFingerprint(vv.fingerprint),
                              getattr(IM, name)(verified, model), orc)
    t = time.perf_counter(); st, tr = mz.minimize(); dt = time.perf_counter() - t
    out["intmin_config4_" + name] = {"seconds": dt, "deliveries_before": IM.countMsgEvents(verified),
                                     "deliveries_after": IM.countMsgEvents(tr), "sequential_replays": st.total_replays,
                                     "replays_launched": mz.speculative_replays, "launches": len(mz.batches)}
    orc.shutdown()
# K3: kernel-only rate on a fixed batch of prefixes, then the whole loop
model3, ev3, depth = raft5_config3()
d = DPORwHeuristics(SchedulerConfig(model=model3), depth_bound=depth, stopIfViolationFound=False, batch=2048)
t = time.perf_counter(); res = d.explore(ev3, max_interleavings=8192); dt = time.perf_counter() - t
out["k3_python_loop_interleavings_per_s"] = len(res.interleavings) / dt
for nb in (2048, 16384):
    dn = DPORwHeuristics(SchedulerConfig(model=model3), depth_bound=depth, stopIfViolationFound=False, batch=nb)
    dn.explore_native(ev3, max_interleavings=64)         # context + first launch out of the timing
    dn.shutdown()
    dn = DPORwHeuristics(SchedulerConfig(model=model3), depth_bound=depth, stopIfViolationFound=False, batch=nb)
    t = time.perf_counter(); rn = dn.explore_native(ev3, max_interleavings=1 << 17); dt = time.perf_counter() - t
    out["k3_native_loop_batch%d" % nb] = {"interleavings_per_s": len(rn.interleavings) / dt, "interleavings": len(rn.interleavings),
                                          "launches": len(rn.rounds), "seconds": dt, "exhausted": rn.exhausted,
                                          "distinct_schedules": len(rn.schedule_hashes())}
    dn.shutdown()
out["k3_native_loop_interleavings_per_s"] = out["k3_native_loop_batch2048"]["interleavings_per_s"]
# the same exploration with the kernel compiled for the model's table (compilation outside the timing)
dn = DPORwHeuristics(SchedulerConfig(model=model3), depth_bound=depth, stopIfViolationFound=False, batch=16384, specialize=True)
dn.explore_native(ev3, max_interleavings=64)
t = time.perf_counter(); rn = dn.explore_native(ev3, max_interleavings=1 << 17); dt = time.perf_counter() - t
out["k3_native_loop_batch16384_specialised"] = {"interleavings_per_s": len(rn.interleavings) / dt, "interleavings": len(rn.interleavings),
                                                "seconds": dt}
dn.shutdown()
pref = [il.trace[:max(1, il.prefix_len)] for il in res.interleavings[:8192]]
par = T.DporParams(depth, 0, 0, 0, 64, 4096)
d._ctx.dpor_batch(pref[:64], par)
t = time.perf_counter(); d._ctx.dpor_batch(pref, par); dt = time.perf_counter() - t
out["k3_batch_interleavings_per_s_incl_copies"] = len(pref) / dt
t = time.perf_counter(); cb = O.dpor_batch(model3, ev3, pref[:512], par); dt = time.perf_counter() - t
out["k3_cpu_baseline"] = {"value": 512 / dt, "unit": "interleavings/s", "cores": 1, "kind": "port",
                          "sample": "first 512 of the same prefixes, oracle via ctypes (one call per interleaving)"}
out["k3_mean_trace_len"] = float(np.mean([len(il.trace) for il in res.interleavings]))
d.shutdown()
print(json.dumps(out, indent=1))
