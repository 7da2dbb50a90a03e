#!/usr/bin/env python
"""Dump (model, external trace, seeds, verdicts, delivery lists) of a RandomScheduler fuzz run in the flat experiment
format of demi_amd/serialization.py, so that a box with a JVM can diff the engine against DEMi's own RandomScheduler
mechanically (SURVEY 8c ii): for every seed s in seeds.bin run

    new RandomScheduler(config, 1, interval, new FullyRandom(seed = s))        (RunnerUtils.fuzz's shape, RunnerUtils.scala:75-90)

on an Akka implementation of model.json's transition table with externals.bin as the trace, and compare
  * violated / not, and the ViolationFingerprint code                         -> verdicts.bin   (demi_verdict[], 16 B each)
  * the delivery sequence (snd, rcv, fingerprint) of the execution             -> deliveries_<k>.bin (demi_rec_event[] of seed k:
                                                                                  MsgSend / MsgEvent / Spawn / ... records, 12 B each)
The verdict's 64-bit hash is FNV-1a over every delivered message word then every actor's final state (include/demi_gpu.h),
so equal hashes mean equal interleavings and handler results.

Source of the numbers: the GPU engine when an MI355X is visible (--source gpu, default when available), else the CPU
oracle (--source oracle: the restatement, useful for producing the file set on a box without a GPU).

  python tools/dump_verdicts.py out_dir [--config raft5_config2] [--n 1024] [--traces 16]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import apps, types as T  # noqa: E402
from demi_amd.model import save_model  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("out")
ap.add_argument("--config", default="raft5_config2", choices=["raft5_config2", "raft3_config1", "raft5_config4"])
ap.add_argument("--n", type=int, default=1024, help="executions (seeds seed_base .. seed_base + n - 1)")
ap.add_argument("--traces", type=int, default=16, help="recorded executions to dump in full (violating ones first)")
ap.add_argument("--source", choices=["gpu", "oracle", "auto"], default="auto")
args = ap.parse_args()

model, events, limits = getattr(apps, args.config)()
seed_base = apps.SEED_BASE
source = args.source
if source == "auto":
    import torch
    source = "gpu" if torch.cuda.is_available() else "oracle"
if source == "gpu":
    from demi_amd import _native
    ctx = _native.Context(0)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    verdicts = ctx.random_explore(args.n, limits, seed_base=seed_base)
    record = lambda s: ctx.random_get_trace(s, limits)[1]
else:
    from oracle import oracle_py as O
    verdicts = O.random_explore(model, events, args.n, seed_base=seed_base, limits=limits, n_threads=os.cpu_count())
    record = lambda s: O.random_execute(model, events, s, limits)[1]

os.makedirs(args.out, exist_ok=True)
save_model(model, os.path.join(args.out, "model.json"))
np.ascontiguousarray(events, dtype=T.EXT_EVENT_DTYPE).tofile(os.path.join(args.out, "externals.bin"))
(seed_base + np.arange(args.n, dtype=np.uint64)).tofile(os.path.join(args.out, "seeds.bin"))
np.ascontiguousarray(verdicts, dtype=T.VERDICT_DTYPE).tofile(os.path.join(args.out, "verdicts.bin"))
viol = np.nonzero(verdicts["flags"] & T.V_VIOLATION)[0]
rest = np.setdiff1d(np.arange(args.n), viol)
picked = [int(k) for k in np.concatenate([viol, rest])[:args.traces]]
for k in picked:
    T.rec_events(record(seed_base + k)).tofile(os.path.join(args.out, "deliveries_%d.bin" % k))
meta = {"format": 1, "config": args.config, "source": source, "seed_base": seed_base, "n": args.n,
        "limits": {"max_messages": int(limits.max_messages), "invariant_check_interval": int(limits.invariant_check_interval),
                   "p_max": int(limits.p_max), "strategy": int(limits.strategy)},
        "violations": int(len(viol)), "recorded": picked,
        "layouts": {"externals.bin": "demi_ext_event[] (8 B)", "seeds.bin": "uint64[]", "verdicts.bin": "demi_verdict[] (16 B: flags u32, fingerprint u32, hash u64)",
                    "deliveries_<k>.bin": "demi_rec_event[] (16 B) of execution k"},
        "jvm_side": "RunnerUtils.fuzz shape: one RandomScheduler + FullyRandom(seed) per execution; with max_executions > 1 on ONE "
                    "instance the reference carries the generator over (RandomScheduler.scala:584, 649-651) and is not comparable"}
with open(os.path.join(args.out, "meta.json"), "w") as f:
    json.dump(meta, f, indent=1)
print("wrote %s: %d executions (%s), %d violating, %d recorded" % (args.out, args.n, source, len(viol), len(picked)))
