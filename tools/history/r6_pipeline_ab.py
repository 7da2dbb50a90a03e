#!/usr/bin/env python
"""Round 6: where the two-calls-in-flight pipeline of ONE demi_ctx loses or keeps the overlap (ms per 2^20 schedules, 40 steps each):
  dev_two_streams        demi_random_explore_dev alternating over two torch streams, no host synchronisation inside (bench.py's loop)
  dev_two_streams_sync   the same with the host waiting for launch k right after enqueuing launch k + 1 (what a submit / wait loop does)
  dev_one_stream         one stream
  submit_wait_flagged    demi_random_explore_submit / _wait, flagged executions only: submit(k + 2); wait(k)  (_1_ahead: submit(k + 1); wait(k))
  submit_wait_verdicts   ... every verdict into a pageable host buffer
  host_sync_call         demi_random_explore (one synchronous call per step, every verdict to the host)"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from demi_amd import _native, types as T
from demi_amd.apps import SEED_BASE, raft5_config2

model, events, limits = raft5_config2()
ctx = _native.Context(0)
ctx.model_load(model.to_struct()); ctx.trace_load(events); ctx.model_specialize()
n, K = 1 << 20, 40
dev = torch.device("cuda", 0)
outs = [torch.empty((n, 2), dtype=torch.int64, device=dev) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
sps = [C.c_void_p(s.cuda_stream) for s in streams]
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:          # clock ramp
    ctx.random_explore_dev(n, limits, outs[0].data_ptr(), seed_base=SEED_BASE, stream=sps[0]); torch.cuda.synchronize()
res = {}

def timed(name, fn):
    fn(4); torch.cuda.synchronize()
    t = time.perf_counter(); fn(K); torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t) / K * 1e3

def dev_two(k):
    for i in range(k):
        ctx.random_explore_dev(n, limits, outs[i & 1].data_ptr(), seed_base=SEED_BASE + (i + 1) * n, stream=sps[i & 1])
def dev_two_sync(k):
    evs = [torch.cuda.Event() for _ in range(k)]
    for i in range(k):
        ctx.random_explore_dev(n, limits, outs[i & 1].data_ptr(), seed_base=SEED_BASE + (i + 1) * n, stream=sps[i & 1])
        evs[i].record(streams[i & 1])
        if i:
            evs[i - 1].synchronize()
def dev_one(k):
    for i in range(k):
        ctx.random_explore_dev(n, limits, outs[0].data_ptr(), seed_base=SEED_BASE + (i + 1) * n, stream=sps[0])
hv = [np.ones(n, dtype=T.VERDICT_DTYPE) for _ in range(2)]
def piped(with_verdicts, ahead=2):
    def f(k):
        tk = [ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + 1) * n, want_verdicts=with_verdicts) for j in range(min(ahead, k))]
        for j in range(k):
            if j + ahead < k:
                tk.append(ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (j + ahead + 1) * n, want_verdicts=with_verdicts))
            ctx.random_explore_wait(tk[j], out=hv[j & 1] if with_verdicts else None)
    return f
def host_sync(k):
    for i in range(k):
        rc = _native.lib().demi_random_explore(ctx._h, C.c_uint64(SEED_BASE + (i + 1) * n), None, n, C.byref(limits), hv[0].ctypes.data)
        assert rc == 0
viol = [torch.zeros((65537, 2), dtype=torch.int64, device=dev) for _ in range(3)]
pin = [torch.zeros((65537, 2), dtype=torch.int64).pin_memory() for _ in range(3)]
def mimic(collect, copy, ahead=2):
    """what submit / wait enqueues, on two torch streams: K1 [+ the compaction kernel] [+ the copy of its list to pinned memory], an
    event; the host waits for call k after call k + `ahead` is enqueued"""
    def f(k):
        evs = [torch.cuda.Event() for _ in range(k)]
        def enq(i):
            st = i & 1
            o = outs[st]
            ctx.random_explore_dev(n, limits, o.data_ptr(), seed_base=SEED_BASE + (i + 1) * n, stream=sps[st])
            if collect:
                ctx.collect_violations_dev(o.data_ptr(), n, 0, viol[i % 3][1:].data_ptr(), 65536, viol[i % 3][0:1].data_ptr(), stream=sps[st])
            if copy:
                with torch.cuda.stream(streams[st]):
                    pin[i % 3].copy_(viol[i % 3], non_blocking=True)
            evs[i].record(streams[st])
        for i in range(min(ahead, k)):
            enq(i)
        for i in range(k):
            if i + ahead < k:
                enq(i + ahead)
            evs[i].synchronize()
    return f
def piped_env(val, with_verdicts=False):
    inner = piped(with_verdicts)
    def f(k):
        os.environ["DEMI_EXPERIMENT"] = "1"; os.environ["DEMI_TICKET_BISECT"] = val
        try:
            inner(k)
        finally:
            del os.environ["DEMI_TICKET_BISECT"]
    return f
for name, fn in (("mimic_k1_only_2_ahead", mimic(False, False)), ("mimic_k1_collect_2_ahead", mimic(True, False)),
                 ("mimic_k1_collect_copy_2_ahead", mimic(True, True)), ("mimic_k1_collect_copy_1_ahead", mimic(True, True, 1)),
                 ("submit_wait_no_collect_no_copy", piped_env("3")), ("submit_wait_no_copy", piped_env("2")), ("submit_wait_no_collect", piped_env("1")),
                 ("dev_two_streams", dev_two), ("dev_two_streams_sync", dev_two_sync), ("dev_one_stream", dev_one),
                 ("submit_wait_flagged_1_ahead", piped(False, 1)), ("submit_wait_flagged", piped(False)), ("submit_wait_verdicts_1_ahead", piped(True, 1)),
                 ("submit_wait_verdicts", piped(True)), ("host_sync_call", host_sync),
                 ("dev_two_streams_again", dev_two)):
    timed(name, fn)
print(json.dumps({"ms_per_2^20_schedules": res, "schedules_per_s": {k: n / (v * 1e-3) for k, v in res.items()}}, indent=1))
