#!/usr/bin/env python
"""bench.py — candidate schedules evaluated per second on the Raft-5 / 50-event fuzz workload.

One "step" = one pass of the hot path over one batch: every rank runs N_PER_GPU RandomScheduler
executions (BASELINE config 2: akka-raft-like 5 actors, 50-event external trace, maxMessages 200,
invariant every 30 deliveries) on its own slice of the schedule-index space, compacts its
found-violation set on the device and (N > 1) all-gathers the per-rank sets over RCCL.
Inputs (transition table, trace) and outputs (verdicts) are resident in HBM during the timed region.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...          (no launcher: bench.py starts the N ranks itself, the same way, on a free port)

Every timed step evaluates FRESH seeds: step i of rank r runs the schedule indices [(i * W + r) * n, ... + n), so that
`bugs_per_hr` counts the distinct violating executions (by the 64-bit hash over every delivered message and every final
state) found in the timed region per wall-clock hour; one more, untimed, step on the fixed indices [r * n, ...) follows for the
bit-for-bit comparison with the CPU oracle.

The one JSON line also carries, under "secondary" (after the timed region), the other loops of the hot path on their BASELINE
configurations, each with its own roofline and CPU baseline:
  config1  raft3, 100 schedules per call through host buffers (N = 1 only);
  dpor     config 3: DPORwHeuristics, raft5, depth 30 — the whole bounded exploration (interleavings/s), in ROUNDS order and
           in the reference's own order (N = 1 only: the reference order's commit is one sequential loop);
  ddmin    config 4: DDMin of a 200-event failing execution over the STSSched replay oracle.  N = 1: replays/s of 2^20 resident
           candidates + demi_ddmin end to end.  N > 1: demi_ddmin with every speculative frontier split over the ranks
           (demi_replay_batch_sharded inside the library: all-gather of the verdicts), and the aggregate replay rate;
  config5  the 8-actor shuffle pipeline, bounded DPOR exploration with a budget of 2^20 interleavings (apps.shuffle8_dpor_config5),
           in ROUNDS order and (N = 1) in the reference's own order.
           N > 1: demi_dpor_explore with the communicator (rounds dealt over the ranks, explored-pair table owner-sharded).
`--workload dpor|ddmin|config1|config5` prints that record alone as the line (same contract fields; ddmin and config5 also
under torch.distributed.run with N ranks).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The HIP runtime multiplexes a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues; this process owns torch's
# streams AND the library's own two (demi_random_explore_submit / _wait, the host-buffer record), which must not end up on one
# queue (include/demi_gpu.h; profiles/r06_pipeline_ab.txt).  Read by the runtime when it initialises: set before torch is imported.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

N_PER_GPU = 1 << 20           # "1M random interleavings on 1 MI355X"
VIOL_CAP = 1 << 16            # found-violation list capacity per rank and step
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak
PREWARM_S = 1.5               # untimed launches before the warmup steps: the shader clock ramps over ~1 s (DVFS)
K1_COUNTERS = os.path.join(ROOT, "profiles", "k1_counters.json")   # rocprofv3 --pmc passes of this workload (tools/profile_r5.sh)


def roofline(bound_bytes, kernel_ms, traffic, kernel, note, extra=None):
    achieved = bound_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
    r = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
         "traffic": traffic, "kernel": kernel, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": bound_bytes, "note": note}
    if extra:
        r.update(extra)
    return r


def _seq_digest(verdicts):
    """Order-sensitive 64-bit digest of a verdict array (flags, fingerprint, hash of every entry in sequence)."""
    import numpy as np
    if not len(verdicts):
        return 0
    idx = np.arange(1, len(verdicts) + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        x = (verdicts["hash"] ^ (verdicts["flags"].astype(np.uint64) << np.uint64(32)) ^ verdicts["fingerprint"].astype(np.uint64)) * \
            (idx * np.uint64(0x9E3779B97F4A7C15) | np.uint64(1))
    return int(np.bitwise_xor.reduce(x))


def bench_dpor(ctx_device, cpu_baseline=True, batch=16384, orders=("rounds", "reference_order"), rounds_batch=32768):
    """BASELINE config 3: the whole bounded DPOR exploration of raft5 (depth 30, Start x 5 + Send x 5) - from round 6 on the
    workload that FINDS the seeded bug (apps.raft5_dpor_config3 says what changed and why).  `value` is the rate in the
    REFERENCE order - the order whose explored set and found-violation set are DPORwHeuristics' own - with the ROUNDS order
    beside it; the record states how the two orders' violating sets relate (`violating_sets`)."""
    import hashlib
    import numpy as np
    from demi_amd import types as T
    from demi_amd.apps import raft5_dpor_config3
    model, ev, par = raft5_dpor_config3()
    depth = int(par.depth_bound)
    out = {"metric": "interleavings explored/sec, DPORwHeuristics bounded exploration (raft5, depth 30)", "unit": "interleavings/s",
           "config": {"workload": "raft5-synth (config 2's rows; election budgets 1,1,1,0,0), Start x 5 + Send(Bootstrap) x 5, depth_bound 30, "
                                  "prioritizePendingUponDivergence, trackHistory, stopIfViolationFound = false, explored until the "
                                  "backtrack queue is empty", "batch": batch, "rounds_batch": rounds_batch,
                       "note": "ROUNDS of rounds_batch backtrack points (the width is the caller's: a k3_dpor launch is as long as one "
                               "interleaving whatever its width, profiles/r06_config5_round_width_sweep.txt); the reference's order "
                               "speculates `batch` wide"}}
    runs = {}
    viol_sets = {}
    ref_verdicts = None
    from demi_amd import _native
    for name, ref in (("rounds", False), ("reference_order", True)):
        if name not in orders:
            continue
        # timed: demi_dpor_explore itself, the C entry point a JVM host binds (the Python mirror DPORwHeuristics.explore_native
        # wraps the same call and then builds one object per interleaving, which is not the library's time)
        ctx = _native.Context(ctx_device)
        ctx.model_load(model.to_struct())
        ctx.model_specialize()
        ctx.dpor_load(ev)
        srch = T.DporSearch(batch if ref else rounds_batch, 1 << 20, 0, 1, T.DPOR_ORDER_REFERENCE if ref else T.DPOR_ORDER_ROUNDS)
        # a first whole exploration outside the timing: compilation for this table, the device arenas (a long-lived demi_ctx
        # keeps them); every call is a fresh exploration
        # (the output arrays are the caller's, as a JVM host's are: allocated and touched before the timed call)
        bufs = _native.Context.dpor_buffers(srch.max_interleavings)
        ctx.dpor_explore(par, srch, buffers=bufs)
        t = time.perf_counter()
        verdicts, plen, rounds, vtrace, st = ctx.dpor_explore(par, srch, buffers=bufs)
        dt = time.perf_counter() - t
        verdicts, plen = verdicts.copy(), plen.copy()
        vh = np.unique(verdicts["hash"][(verdicts["flags"] & T.V_VIOLATION) != 0])
        viol_sets[name] = set(vh.tolist())
        if ref:
            ref_verdicts = verdicts
        runs[name] = {"value": len(verdicts) / dt, "seconds": dt, "interleavings": len(verdicts),
                      "mean_prefix_len": float(np.mean(plen)) if len(plen) else 0.0, "backtrack_points": int(st.backtrack_points),
                      "sequence_digest": "%016x" % _seq_digest(verdicts),
                      "sha256_verdicts": hashlib.sha256(np.ascontiguousarray(verdicts).tobytes()).hexdigest(),
                      "executed_on_device": int(st.executed), "launches": int(st.launches), "exhausted": bool(st.exhausted),
                      "violations": int(np.count_nonzero(verdicts["flags"] & T.V_VIOLATION)),
                      "distinct_violating_schedules": int(len(vh)), "bugs_per_hr": len(vh) / dt * 3600.0,
                      "sha256_sorted_violating_hashes": hashlib.sha256(vh.tobytes()).hexdigest(),
                      "first_violation": int(np.nonzero(verdicts["flags"] & T.V_VIOLATION)[0][0]) if len(vh) else None,
                      "distinct_schedules": int(len(np.unique(verdicts["hash"]))),
                      "kernel_ms_total": float(st.kernel_ms), "h2d_bytes": int(st.h2d_bytes), "d2h_bytes": int(st.d2h_bytes)}
        if ref:
            runs[name]["launches_for_results_the_speculation_lacked"] = int(st.cache_misses)
            runs[name]["record_fetches"] = int(st.fetches)
        ctx.close()
    # the headline of this record is the reference's own order (its explored set and its found-violation set are
    # DPORwHeuristics'); the ROUNDS order explores another - larger - set, whose violating set is compared below
    head = "reference_order" if "reference_order" in runs else next(iter(runs))
    first = "rounds" if "rounds" in runs else next(iter(runs))
    out["value"] = runs[head]["value"]
    out["value_order"] = head
    out["violations"] = runs[head]["violations"]
    out["bugs_per_hr"] = runs[head]["bugs_per_hr"]
    out["orders"] = runs
    if len(viol_sets) == 2:
        a, b = viol_sets["rounds"], viol_sets["reference_order"]
        out["violating_sets"] = {"common": len(a & b), "rounds_only": len(a - b), "reference_order_only": len(b - a),
                                 "note": "distinct violating executions by the 64-bit hash over every delivery and final state; the "
                                         "two orders flip each racing pair once but in different contexts (ExploredTacker is global), "
                                         "so neither explored set contains the other"}
    try:
        with open(os.path.join(ROOT, "tests", "golden", "dpor_config3_bug_reference_order.json")) as f:
            gold = json.load(f)
        if "reference_order" in runs:
            out["reference_order_equals_golden_record"] = all(runs["reference_order"][k] == gold[k] for k in
                                                              ("interleavings", "violations", "sha256_verdicts", "sha256_sorted_violating_hashes"))
    except (OSError, ValueError, KeyError):
        pass
    r = runs[first]
    # algorithmic bytes per interleaving = 4 x depth (its prefix in) + 8 (verdict out) + 12 x r (its r new
    # backtrack points out), with the MEASURED mean prefix length and r = backtrack points enqueued / interleavings.  What the
    # kernels additionally move inside HBM (finished traces into the arena and back into the pair kernels, racing pairs, one
    # 64-byte explored-pair entry per pair) is working-set traffic, reported beside it as a model and, when a counters profile of
    # this workload exists (tools/profile_r5_k2k3.sh), as measured `traffic` - never as algorithmic bytes.
    n_il = r["interleavings"]
    per_il = 4.0 * r["mean_prefix_len"] + 8.0 + 12.0 * (r["backtrack_points"] / max(1, n_il))
    alg = n_il * per_il
    traffic = (_counters_profile("dpor_counters.json") or {}).get("fabric_bytes_per_exploration")
    out["roofline"] = roofline(alg, r["kernel_ms_total"], traffic,
                               "k3_dpor + k3_pairs_mark/insert/decide (specialised, hiprtc), %d rounds" % r["launches"],
                               "ROUNDS order. Algorithmic bytes per SURVEY 8(d): 4 x mean prefix length (%.1f events) + 8 + 12 x r "
                               "(r = %.2f backtrack points per interleaving) = %.0f B per interleaving. kernel_ms = sum over the "
                               "rounds (HIP events in the library). Latency / issue bound: a round of the backtrack queue is far "
                               "smaller than the chip" % (r["mean_prefix_len"], r["backtrack_points"] / max(1, n_il), per_il),
                               {"working_set_bytes_model": n_il * (2 * 16 * 190 + 600 * (3 * 4 + 2 * 64) + 8 + 16),
                                "working_set_note": "model of what the round's kernels move inside HBM per interleaving: the finished trace "
                                                    "(16 B x ~190 events) written to the arena and read by the pair kernels, ~600 racing pairs "
                                                    "written and read twice (4 B), one 64 B explored-pair entry per pair in insert and in decide",
                                "pcie_bytes": {"h2d": r["h2d_bytes"], "d2h": r["d2h_bytes"]},
                                "reference_order": None if "reference_order" not in runs else
                                {"algorithmic_bytes": runs["reference_order"]["interleavings"] *
                                 (4.0 * runs["reference_order"]["mean_prefix_len"] + 8.0 + 12.0 * runs["reference_order"]["backtrack_points"] /
                                  max(1, runs["reference_order"]["interleavings"])),
                                 "kernel_ms_total": runs["reference_order"]["kernel_ms_total"],
                                 "pcie_bytes": {"h2d": runs["reference_order"]["h2d_bytes"], "d2h": runs["reference_order"]["d2h_bytes"]}}})
    # continuity: the workload rounds 1-5 timed (apps.raft5_config3: 60 332 interleavings in the reference's order, none
    # violating - flips that getMatchingMessage undoes), both orders, so that the rates of the two workloads can be read side by side
    try:
        from demi_amd.apps import raft5_config3
        m5, e5, d5 = raft5_config3()
        p5 = T.DporParams(d5, 0, 0, 0, 64, 4096)
        old = {}
        for name, ref in (("rounds", False), ("reference_order", True)):
            if name not in orders:
                continue
            c5 = _native.Context(ctx_device)
            c5.model_load(m5.to_struct()); c5.model_specialize(); c5.dpor_load(e5)
            s5 = T.DporSearch(batch, 1 << 17, 0, 1, T.DPOR_ORDER_REFERENCE if ref else T.DPOR_ORDER_ROUNDS)
            c5.dpor_explore(p5, s5)
            t = time.perf_counter()
            v5, _pl5, _r5, _t5, st5 = c5.dpor_explore(p5, s5)
            d = time.perf_counter() - t
            old[name] = {"value": len(v5) / d, "seconds": d, "interleavings": len(v5), "violations": int(np.count_nonzero(v5["flags"] & T.V_VIOLATION)),
                         "launches": int(st5.launches), "sequence_digest": "%016x" % _seq_digest(v5), "kernel_ms_total": float(st5.kernel_ms)}
            c5.close()
        out["round5_workload"] = {"workload": "apps.raft5_config3 (prioritizePendingUponDivergence = false, five campaigning nodes)", "orders": old}
    except Exception as e:
        out["round5_workload"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if "rounds" in runs:
        pctx = _native.Context(ctx_device)
        out["roofline"]["issue_model"] = issue_model(pctx, "dpor", runs["rounds"]["kernel_ms_total"],
                                                     lambda tr: tr.get("sequence_digest") == runs["rounds"]["sequence_digest"])
        pctx.close()
    if cpu_baseline:
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        base = {}
        for name, ref in (("rounds", False), ("reference_order", True)):
            if name not in orders:
                continue
            # (the oracle's reference order runs one interleaving at a time: a speculation `batch` wide costs a host thousands of
            # discarded executions per committed one)
            # discarded executions per committed one) - and a bounded sample of it: the first 2^14 interleavings
            srch = T.DporSearch(1 if ref else rounds_batch, 1 << 14 if ref else 1 << 20, 0, 1, T.DPOR_ORDER_ROUNDS)
            t = time.perf_counter()
            v, plen, rounds, vt, st, secs = O.dpor_explore(model, ev, par, srch, n_threads=1 if ref else cores)
            dt = time.perf_counter() - t
            base[name] = {"value": len(v) / dt, "seconds": dt, "interleavings": len(v), "sequence_digest": "%016x" % _seq_digest(v),
                          "sha256_verdicts": hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest()}
            if ref:
                base[name]["same_as_the_gpus_first_interleavings"] = bool(
                    hashlib.sha256(np.ascontiguousarray(ref_verdicts[:len(v)]).tobytes()).hexdigest() == base[name]["sha256_verdicts"])
        out["cpu_baseline"] = {"value": base[head]["value"], "unit": "interleavings/s", "cores": 1 if head == "reference_order" else cores, "kind": "port",
                               "sample": "the same whole exploration: oracle/demi_oracle.c interleavings under the same host bookkeeping "
                                         "(demi_amd/csrc/dpor_host.hpp) - the reference's order one interleaving at a time on one thread "
                                         "(as DPORwHeuristics runs), the ROUNDS order on %d host threads (one next trace per thread)" % cores,
                               "orders": base,
                               # every verdict (flags, fingerprint, delivery hash) in exploration order, GPU = oracle (the reference
                               # order: over the sample's interleavings; the whole sequence is held against the golden record above)
                               "same_verdict_sequence_as_gpu": {k: (base[k]["same_as_the_gpus_first_interleavings"] if k == "reference_order"
                                                                    else base[k]["sha256_verdicts"] == runs[k]["sha256_verdicts"]) for k in base}}
    return out


def bench_config1(ctx_device, cpu_baseline=True):
    """BASELINE config 1 restated: raft3-synth, 20-event trace, RandomScheduler, 100 schedules - the reference's own
    CPU-runnable case.  The reference's akka-raft / JVM cannot run here: the oracle on ONE thread stands in for it and is
    labelled as such; the same 100 schedules through the GPU path are checked bit for bit and timed beside it (a launch of
    100 schedules is launch-latency, not throughput)."""
    import numpy as np
    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE, raft3_config1
    model, events, limits = raft3_config1()
    n = 100
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    ctx.model_specialize()
    got = ctx.random_explore(n, limits, seed_base=SEED_BASE)
    reps = 20
    t = time.perf_counter()
    for _ in range(reps):
        got = ctx.random_explore(n, limits, seed_base=SEED_BASE)
    dt = (time.perf_counter() - t) / reps
    ctx.close()
    out = {"metric": "candidate schedules evaluated/sec, config 1 (raft3-synth, RandomScheduler, 100 schedules)", "unit": "schedules/s",
           "value": n / dt, "seconds_per_100_schedules": dt,
           "config": {"workload": "raft3-synth, 3 actors, frozen 20-event trace, 100 schedules per call through demi_random_explore "
                                  "(host buffers: what a JVM caller gets)", "max_messages": int(limits.max_messages)},
           "violations": int(np.count_nonzero(got["flags"] & T.V_VIOLATION))}
    if cpu_baseline:
        from oracle import oracle_py as O
        O.random_explore(model, events, n, seed_base=SEED_BASE, limits=limits, n_threads=1)
        t = time.perf_counter()
        for _ in range(reps):
            cpu = O.random_explore(model, events, n, seed_base=SEED_BASE, limits=limits, n_threads=1)
        dc = (time.perf_counter() - t) / reps
        out["cpu_baseline"] = {"value": n / dc, "unit": "schedules/s", "cores": 1, "kind": "port",
                               "sample": "the same 100 schedules, oracle/demi_oracle.c on one thread (restated CPU baseline, never "
                                         "'DEMi JVM': no JVM / akka-raft in this image)", "seconds": dc,
                               "bit_identical_to_gpu": bool((cpu == got).all())}
    return out


def bench_big_tables(ctx_device, cpu_baseline=True, n=1 << 20):
    """Tables of MORE THAN 8 ACTORS (the BIG layout of include/demi_gpu.h; round 6): config 2's step - 2^20 RandomScheduler executions,
    verdicts resident in HBM, one launch at a time on the null stream - on the 11-node raft cluster (apps.raft11_config2) and the
    12-actor shuffle job (apps.shuffle12_config5), and the 12-actor job's DPOR exploration to exhaustion in ROUNDS order.  Each is
    checked against the oracle (a 2^14 prefix of the step; the whole exploration) and timed beside it on every host thread."""
    import hashlib
    import numpy as np
    import torch
    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE, raft11_config2, shuffle12_config5
    out = {"metric": "candidate schedules evaluated/sec, tables of more than 8 actors (BIG layout)", "unit": "schedules/s", "workloads": {}}
    m2, dev, fev, lim2, par = shuffle12_config5()
    threads = os.cpu_count() or 1
    for name, (model, events, lim) in (("raft11", raft11_config2()), ("shuffle12", (m2, fev, lim2))):
        ctx = _native.Context(ctx_device)
        ctx.model_load(model.to_struct())
        ctx.trace_load(events)
        ctx.model_specialize()
        vbuf = torch.empty(n * 16, dtype=torch.uint8, device="cuda:%d" % ctx_device)
        ctx.random_explore_dev(n, lim, vbuf.data_ptr(), seed_base=SEED_BASE)
        torch.cuda.synchronize()
        steps = 5
        t = time.perf_counter()
        for i in range(steps):
            ctx.random_explore_dev(n, lim, vbuf.data_ptr(), seed_base=SEED_BASE + (i + 1) * n)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        ctx.random_explore_dev(n, lim, vbuf.data_ptr(), seed_base=SEED_BASE)
        torch.cuda.synchronize()
        got = vbuf.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)
        ctx.close()
        rec = {"value": n / dt, "unit": "schedules/s", "ms_per_step": dt * 1e3, "schedules_per_step": n, "n_actors": model.n_actors,
               "model": model.name, "violations": int(np.count_nonzero(got["flags"] & T.V_VIOLATION)),
               "capacity_aborts": int(np.count_nonzero(got["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF))),
               "mean_deliveries": float((got["flags"] >> 16).mean()),
               "roofline": roofline(16.0 * n, dt * 1e3, None, "k1_random_explore<false,false> compiled for the table (BIG layout, wide)",
                                    "16 B verdict per schedule; the launch's wall time (one launch at a time)")}
        if cpu_baseline:
            from oracle import oracle_py as O
            k = 1 << 14
            t = time.perf_counter()
            cpu = O.random_explore(model, events, k, seed_base=SEED_BASE, limits=lim, n_threads=threads)
            dc = time.perf_counter() - t
            rec["cpu_baseline"] = {"value": k / dc, "unit": "schedules/s", "cores": threads, "kind": "port",
                                   "sample": "the first 2^14 schedules of the step, oracle/demi_oracle.c on every host thread",
                                   "bit_identical_to_gpu": bool((cpu == got[:k]).all())}
        out["workloads"][name] = rec
    # the 12-actor job's DPOR exploration, exhausted (33 529 interleavings, 1 836 violating: tests/golden/big_tables.json)
    ctx = _native.Context(ctx_device)
    ctx.model_load(m2.to_struct())
    ctx.model_specialize()
    ctx.dpor_load(dev)
    srch = T.DporSearch(4096, 1 << 17, 0, 1, T.DPOR_ORDER_ROUNDS)
    ctx.dpor_explore(par, srch)
    t = time.perf_counter()
    g = ctx.dpor_explore(par, srch)
    dd = time.perf_counter() - t
    ctx.close()
    drec = {"value": len(g[0]) / dd, "unit": "interleavings/s", "seconds": dd, "interleavings": int(len(g[0])), "violations": int(g[4].violations),
            "exhausted": bool(g[4].exhausted), "order": "rounds (batch 4096)", "sha256_verdicts": hashlib.sha256(np.ascontiguousarray(g[0]).tobytes()).hexdigest(),
            "kernel_ms_total": float(g[4].kernel_ms)}
    if cpu_baseline:
        from oracle import oracle_py as O
        t = time.perf_counter()
        c = O.dpor_explore(m2, dev, par, srch, threads)
        dc = time.perf_counter() - t
        drec["cpu_baseline"] = {"value": len(c[0]) / dc, "unit": "interleavings/s", "cores": threads, "kind": "port",
                                "sample": "the whole exploration, the product's host bookkeeping around the oracle's interleavings",
                                "bit_identical_to_gpu": bool(len(c[0]) == len(g[0]) and (c[0] == g[0]).all())}
    out["workloads"]["shuffle12_dpor"] = drec
    out["value"] = out["workloads"]["raft11"]["value"]
    out["config"] = {"workload": "raft11-synth (11 actors, 50-event fuzz trace), shuffle12-synth (12 actors), 2^20 schedules per step; "
                                 "shuffle12 DPOR exhausted"}
    return out


def issue_model(ctx, name, kernel_ms, same_run):
    """The integer-issue model of a secondary record, as the fuzz line has it for K1: the committed instruction counters of this
    workload (tools/profile_r6_k2k3.sh -> profiles/<tag>_<name>_insts.json: wave-instructions per launch / per exploration) priced
    with the SIMD cycles per instruction and the shader clock measured in THIS run (demi_device_probe).  Quoted only when the
    counters belong to this run's work: same_run(traced_run) must hold (same exploration digest / same candidates)."""
    prof = _counters_profile("%s_insts.json" % name)
    if not prof:
        return None
    tr = prof.get("traced_run", {})
    try:
        if not same_run(tr):
            return {"stale": "%s describes another run (%s)" % (prof["profile_file"], {k: tr.get(k) for k in ("sequence_digest", "interleavings", "still_violating")})}
        # (the counters are COUNTS of the same work - the digest says so - and do not depend on the clock; the traced run's kernel
        # time, taken with the profiler serialising every dispatch, is quoted for information only)
        pms = float(tr.get("kernel_ms") or 0.0)
        p6 = ctx.device_probe(6, 6000)
        m6, d6 = ctx.device_probe(6, 6000, 2), ctx.device_probe(6, 6000, 3)
        s6 = ctx.device_probe(6, 6000, 1)
        import torch
        cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
        c = prof["counters"]
        valu, salu = c.get("SQ_INSTS_VALU", 0.0), c.get("SQ_INSTS_SALU", 0.0)
        other = c.get("SQ_INSTS_LDS", 0.0) + c.get("SQ_INSTS_VMEM_RD", 0.0) + c.get("SQ_INSTS_VMEM_WR", 0.0)
        clk = p6.shader_clock_ghz * 1e9
        cyc = kernel_ms * 1e-3 * clk
        simds = cus * 4
        return {"valu_insts": valu, "salu_insts": salu, "lds_vmem_insts": other, "unit": prof.get("unit"),
                "active_lanes_per_valu_inst": (c.get("SQ_THREAD_CYCLES_VALU", 0.0) / valu) if valu else None,
                "waves": c.get("SQ_WAVES"),
                "wait_share_of_wave_cycles": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                "valu_alone_frac": valu * p6.cycles_per_valu / simds / cyc, "salu_alone_frac": salu * s6.cycles_per_valu / simds / cyc,
                "issue_frac_straight_line": (valu + salu + other) * m6.cycles_per_valu / simds / cyc,
                "issue_frac_branchy": (valu + salu + other) * d6.cycles_per_valu / simds / cyc,
                "clock_hz": clk, "kernel_ms": kernel_ms, "source": "%s (rocprofv3 --pmc; kernel time %.3f ms in the traced run); clock and SIMD cycles "
                "per instruction measured in this run. The share of the kernels' duration that issuing their instructions would take on "
                "ALL SIMDs lies between the straight-line and the branchy figure: far below 1 = the launches do not fill the chip / "
                "wait on memory, not issue-bound" % (prof["profile_file"], pms)}
    except Exception as e:          # a model must never cost the record
        return {"error": "%s: %s" % (type(e).__name__, e)}


class Ranks:
    """What the records below need from the process group: one instance per process.  world == 1: everything is local."""

    def __init__(self, rank=0, world=1, attach=None, dist=None, cdev=None):
        self.rank, self.world, self.attach, self.dist, self.cdev = rank, world, attach, dist, cdev

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def max(self, x):
        if self.world == 1:
            return float(x)
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, obj):
        """every rank's (small, picklable) object, in rank order, on every rank"""
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out


def _counters_profile(name):
    """fabric-side bytes of a secondary record from its FETCH_SIZE / WRITE_SIZE passes (tools/profile_r5_k2k3.sh), if committed:
    `name` is the file's name without its round tag; the newest round's file wins"""
    for tag in ("r06", "r05", "r04", "r03"):
        try:
            with open(os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))) as f:
                d = json.load(f)
                d["profile_file"] = "profiles/%s_%s" % (tag, name)
                return d
        except (OSError, ValueError):
            continue
    return None


def bench_config5(ctx_device, cpu_baseline=True, max_interleavings=None, batch=65536, ranks=None, small=True, reference_order=True,
                  reference_batch=16384):
    """BASELINE config 5: shuffle8-synth as a pipeline of three jobs (8 actors, 3 classes), bounded DPOR exploration with a budget
    of 2^20 interleavings (apps.shuffle8_dpor_config5; shuffle8_config5_large says why more externals do not enlarge the one-job exploration and
    chained jobs do).  With N ranks: demi_dpor_explore with the communicator - a round's backtrack points dealt over the ranks
    in contiguous blocks, the explored-pair table sharded by owner = hash(unordered pair) mod N, three all-gathers per round
    (DESIGN section 6); every rank ends with the same verdict sequence (checked)."""
    import hashlib
    import numpy as np
    from demi_amd import _native, types as T
    from demi_amd.apps import shuffle8_config5, shuffle8_dpor_config5
    ranks = ranks or Ranks()
    if ranks.world > 1:
        batch = min(batch, 16384)     # (the multi-rank rounds stage at most 2^21 backtrack points per round - DESIGN section 6)
    model, dpor_events, par, budget = shuffle8_dpor_config5()
    depth = int(par.depth_bound)
    if max_interleavings is None:
        max_interleavings = budget
    srch = T.DporSearch(batch, max_interleavings, 0, 1, T.DPOR_ORDER_ROUNDS)
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    t = time.perf_counter()
    ctx.model_specialize()
    ctx.dpor_load(dpor_events)
    collective = ranks.attach(ctx) if ranks.world > 1 else "none (1 rank)"
    # compilation for this table and the device arenas (sized by the budget: 4 KB of trace arena per interleaving and the
    # explored-pair table are allocated when a call first needs them - the same budget, so that the timed call allocates nothing;
    # every call is a fresh exploration)
    warm = T.DporSearch(batch, max_interleavings, 0, 1, T.DPOR_ORDER_ROUNDS)
    bufs = _native.Context.dpor_buffers(max_interleavings)        # (the caller's output arrays, as a JVM host's are: touched before the timed call)
    ctx.dpor_explore(par, warm, buffers=bufs)
    setup_s = time.perf_counter() - t
    ranks.barrier()
    t = time.perf_counter()
    verdicts, plen, rounds, vtrace, st = ctx.dpor_explore(par, srch, buffers=bufs)
    dt = ranks.max(time.perf_counter() - t)
    verdicts, plen, rounds = verdicts.copy(), plen.copy(), rounds.copy()
    n_il = len(verdicts)
    digest = "%016x" % _seq_digest(verdicts)
    per_il = 4.0 * float(np.mean(plen)) + 8.0 + 12.0 * (int(st.backtrack_points) / max(1, n_il))
    flags, counts = np.unique(verdicts["flags"] & 0xFF, return_counts=True)
    vh = np.unique(verdicts["hash"][(verdicts["flags"] & T.V_VIOLATION) != 0])
    prof = _counters_profile("config5_counters.json")
    out = {"metric": "interleavings explored/sec, bounded DPOR (shuffle8-synth pipeline of 3 jobs, depth 40, budget %d)" % max_interleavings,
           "unit": "interleavings/s", "value": n_il / dt, "seconds": dt, "interleavings": n_il, "exhausted": bool(st.exhausted),
           "budget": int(max_interleavings), "backtrack_points_still_queued": int(st.queue_len), "launches": int(st.launches),
           "n_gpus": ranks.world, "collective": collective, "setup_s_untimed": setup_s,
           "violations": int(np.count_nonzero(verdicts["flags"] & T.V_VIOLATION)), "sequence_digest": digest,
           "distinct_violating_schedules": int(len(vh)), "bugs_per_hr": len(vh) / dt * 3600.0,
           "first_violation": int(np.nonzero(verdicts["flags"] & T.V_VIOLATION)[0][0]) if len(vh) else None,
           "sha256_verdicts": hashlib.sha256(np.ascontiguousarray(verdicts).tobytes()).hexdigest(),
           "value_order": "rounds",
           "distinct_schedules": int(len(np.unique(verdicts["hash"]))),
           "verdict_flag_histogram": {"0x%02x" % int(f): int(c) for f, c in zip(flags, counts)},
           "mean_prefix_len": float(np.mean(plen)), "kernel_ms_total": float(st.kernel_ms),
           "pcie_bytes": {"h2d": int(st.h2d_bytes), "d2h": int(st.d2h_bytes)},
           "config": {"workload": "shuffle8-synth pipeline (2-stage shuffle stand-in, 8 actors, 3 actor classes, 3 jobs back to back, both "
                                  "seeded bugs: apps.shuffle8_dpor_config5), Start x 8 + Submit + Speculate, depth_bound 40, "
                                  "prioritizePendingUponDivergence, ROUNDS of %d, budget %d interleavings (not exhausted: "
                                  "the budget bounds the search, RunnerUtils.boundedDPOR's shape)" % (batch, max_interleavings)},
           "roofline": roofline(n_il * per_il, float(st.kernel_ms), (prof or {}).get("fabric_bytes_per_exploration"),
                                "k3_dpor + k3_pairs_* (specialised)",
                                "SURVEY 8(d): 4 x mean prefix (%.1f) + 8 + 12 x r (%.2f) B per interleaving; kernel_ms = this rank's launches" %
                                (float(np.mean(plen)), int(st.backtrack_points) / max(1, n_il)))}
    if ranks.world > 1:
        per_rank = ranks.gather({"digest": digest, "interleavings": n_il, "kernel_ms": float(st.kernel_ms)})
        out["per_rank"] = per_rank
        out["same_verdict_sequence_on_every_rank"] = all(r["digest"] == digest and r["interleavings"] == n_il for r in per_rank)
    if ranks.world == 1:
        out["roofline"]["issue_model"] = issue_model(ctx, "config5", float(st.kernel_ms), lambda tr: tr.get("sequence_digest") == digest)
    if ranks.world == 1 and reference_order:
        # the same budget in the REFERENCE order: the order whose 2^20 interleavings are the ones DPORwHeuristics itself would
        # explore under this budget (ROUNDS takes the 2^20 from another frontier), single-rank by construction
        rs = T.DporSearch(reference_batch, max_interleavings, 0, 1, T.DPOR_ORDER_REFERENCE)
        try:
            # (untimed: this order's own staging and kernels with a small budget - the trace arena and the explored-pair table
            # of the full budget exist since the ROUNDS run above; the whole exploration takes seconds in this order)
            ctx.dpor_explore(par, T.DporSearch(reference_batch, min(max_interleavings, 1 << 15), 0, 1, T.DPOR_ORDER_REFERENCE), buffers=bufs)
            t = time.perf_counter()
            rv, rplen, _rr, _rt, rst = ctx.dpor_explore(par, rs, buffers=bufs)
            rdt = time.perf_counter() - t
            rvh = np.unique(rv["hash"][(rv["flags"] & T.V_VIOLATION) != 0])
            out["reference_order"] = {"value": len(rv) / rdt, "seconds": rdt, "interleavings": len(rv), "exhausted": bool(rst.exhausted),
                                      "violations": int(np.count_nonzero(rv["flags"] & T.V_VIOLATION)),
                                      "distinct_violating_schedules": int(len(rvh)), "bugs_per_hr": len(rvh) / rdt * 3600.0,
                                      "first_violation": int(np.nonzero(rv["flags"] & T.V_VIOLATION)[0][0]) if len(rvh) else None,
                                      "sequence_digest": "%016x" % _seq_digest(rv),
                                      "sha256_verdicts": hashlib.sha256(np.ascontiguousarray(rv).tobytes()).hexdigest(),
                                      "sha256_first_6000_verdicts": hashlib.sha256(np.ascontiguousarray(rv[:6000]).tobytes()).hexdigest(),
                                      "executed_on_device": int(rst.executed), "launches": int(rst.launches), "record_fetches": int(rst.fetches),
                                      "kernel_ms_total": float(rst.kernel_ms),
                                      "pcie_bytes": {"h2d": int(rst.h2d_bytes), "d2h": int(rst.d2h_bytes)}}
            a, b = set(vh.tolist()), set(rvh.tolist())
            out["violating_sets"] = {"common": len(a & b), "rounds_only": len(a - b), "reference_order_only": len(b - a),
                                     "note": "under a budget the two orders explore different interleavings (the search is not exhausted)"}
            try:
                with open(os.path.join(ROOT, "tests", "golden", "dpor_config5_bug_transliteration.json")) as f:
                    g5 = json.load(f)
                if g5["interleavings"] == 6000 and len(rv) >= 6000:
                    out["reference_order"]["first_6000_equal_the_transliterations_record"] = bool(
                        g5["sha256_verdicts"] == out["reference_order"]["sha256_first_6000_verdicts"])
            except (OSError, ValueError, KeyError):
                pass
        except Exception as e:        # reported, never hidden: the ROUNDS record above stands on its own
            out["reference_order"] = {"error": str(e)}
    ctx.close()
    if small and ranks.world == 1:
        # the one-job table of rounds 1-3 (1 653 interleavings, exhausted in three launches: launch latency, kept for continuity)
        m1, ev1, _f, _l = shuffle8_config5()
        c1 = _native.Context(ctx_device)
        c1.model_load(m1.to_struct()); c1.model_specialize(); c1.dpor_load(ev1)
        s1 = T.DporSearch(batch, 1 << 17, 0, 1, T.DPOR_ORDER_ROUNDS)
        c1.dpor_explore(par, s1)
        t = time.perf_counter()
        v1, _p1, _r1, _t1, st1 = c1.dpor_explore(par, s1)
        d1 = time.perf_counter() - t
        c1.close()
        out["one_job"] = {"interleavings": len(v1), "seconds": d1, "value": len(v1) / d1, "exhausted": bool(st1.exhausted),
                          "violations": int(np.count_nonzero(v1["flags"] & T.V_VIOLATION)),
                          "launches": int(st1.launches), "sequence_digest": "%016x" % _seq_digest(v1)}
    if cpu_baseline and ranks.rank == 0:
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        m = min(n_il, 1 << 16)          # a bounded sample: the same exploration cut at 2^16 interleavings (same order, same prefix)
        cs = T.DporSearch(batch, m, 0, 1, T.DPOR_ORDER_ROUNDS)
        t = time.perf_counter()
        v, _pl, _r, _vt, _st, _s = O.dpor_explore(model, dpor_events, par, cs, n_threads=cores)
        dc = time.perf_counter() - t
        out["cpu_baseline"] = {"value": len(v) / dc, "unit": "interleavings/s", "cores": cores, "kind": "port", "seconds": dc,
                               "sample": "the first %d interleavings of the same exploration (budget %d), oracle interleavings on %d host "
                                         "threads under the same host bookkeeping" % (len(v), m, cores),
                               "same_verdict_sequence_as_gpu_prefix": "%016x" % _seq_digest(v) == "%016x" % _seq_digest(verdicts[:len(v)])}
    return out


def bench_ddmin_ranks(ctx_device, ranks, n_per_rank=1 << 18):
    """BASELINE config 4 with N ranks ("DDMin ... subsequence frontier sharded across 8 GPUs"): demi_ddmin with the communicator -
    every speculative frontier of the decision tree is split over the ranks in contiguous blocks inside the library
    (demi_replay_batch_sharded: one all-gather of the verdicts per launch), all ranks walk the same tree; the launch budget
    grows with N (a frontier N times as wide in the time of one launch).  Beside it the aggregate replay rate: every rank
    replays its own n_per_rank resident candidates and the verdicts are all-gathered."""
    import numpy as np
    import torch
    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE, raft5_config4
    model, events, lim = raft5_config4()
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    ctx.replay_load(used, rec)
    ctx.model_specialize()
    target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
    W = ranks.world
    # the single-rank call first (no communicator yet): every rank computes the same MCS - the reference for the sharded one
    par1 = T.DdminParams(0, 1024, 1, 1)
    ctx.ddmin(target, par1)
    t = time.perf_counter()
    mcs1, cons1, batches1, st1 = ctx.ddmin(target, par1)
    single_s = time.perf_counter() - t
    collective = ranks.attach(ctx) if W > 1 else "none (1 rank)"
    best = None
    for budget in (1024 * W, 4096 * W):
        parw = T.DdminParams(0, budget, 1, 1)
        ctx.ddmin(target, parw)
        for _ in range(3):
            ranks.barrier()
            t = time.perf_counter()
            mcsw, consw, batchesw, stw = ctx.ddmin(target, parw)
            dt = ranks.max(time.perf_counter() - t)
            if best is None or dt < best["seconds"]:
                best = {"seconds": dt, "max_candidates_per_launch_all_ranks": budget, "launches": int(stw.launches),
                        "replays_launched": int(stw.replays), "candidates_per_launch": batchesw, "mcs_len": len(mcsw),
                        "oracle_consultations": int(stw.consultations),
                        "same_mcs_as_single_rank": tuple(mcsw) == tuple(mcs1),
                        "same_consultation_sequence_as_single_rank": [(tuple(c), p) for c, p in consw] == [(tuple(c), p) for c, p in cons1]}
    # aggregate replay rate: n_per_rank random candidates per rank, resident in HBM, verdicts all-gathered through the library
    dev = torch.device("cuda", ctx_device)
    rng = np.random.default_rng(1000 + ranks.rank)
    keep = rng.random((n_per_rank, len(used))) < 0.7
    masks = np.zeros((n_per_rank, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    d_masks = torch.from_numpy(masks.view(np.int64)).to(dev)
    d_out = torch.empty((n_per_rank, 2), dtype=torch.int64, device=dev)
    d_all = torch.empty((W, n_per_rank, 2), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    def one():
        ctx.replay_batch_dev(d_masks.data_ptr(), n_per_rank, target, d_out.data_ptr(), stream=sp)
        ctx.comm_allgather_dev(d_out.data_ptr(), d_all.data_ptr(), n_per_rank * 16, stream=sp)
    one()
    torch.cuda.synchronize(dev)
    ranks.barrier()
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        one()
    torch.cuda.synchronize(dev)
    ranks.barrier()
    dt = ranks.max(time.perf_counter() - t)
    got = d_all.cpu().numpy().view(T.VERDICT_DTYPE).reshape(W, n_per_rank)
    digests = ranks.gather("%016x" % _seq_digest(got[ranks.rank]))
    out = {"metric": "candidate subsequences replayed/sec (STSScheduler.test without peek), %d ranks" % W, "unit": "replays/s",
           "value": W * n_per_rank * reps / dt, "n_gpus": W, "collective": collective,
           "config": {"workload": "raft5-synth, 200-event failing execution (%d externals), %d random candidates per rank per pass, "
                                  "resident in HBM, verdicts all-gathered" % (len(used), n_per_rank)},
           "ms_per_pass": dt / reps * 1e3,
           "every_ranks_block_arrived_everywhere": all(d == "%016x" % _seq_digest(got[k]) for k, d in enumerate(digests)),
           "ddmin_end_to_end": dict(best, externals=int(len(used))),
           "ddmin_end_to_end_single_rank": {"seconds": single_s, "launches": int(st1.launches), "replays_launched": int(st1.replays),
                                            "mcs_len": len(mcs1), "max_candidates": 1024}}
    ctx.close()
    # randomDDMin with the frontier split over the ranks (demi_random_ddmin with the communicator): the DDMin shape whose launches
    # have real width - (candidates x 100 executions)
    try:
        from demi_amd.schedulers import ViolationFingerprint
        out["random_ddmin_R100"] = _bench_random_ddmin(ctx_device, model, used, rec, ViolationFingerprint(vv.fingerprint), False, ranks=ranks)
    except Exception as e:
        out["random_ddmin_R100"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return out


def _bench_random_ddmin(ctx_device, model, used, rec, fp, cpu_baseline, R=100, ranks=None):
    import numpy as np
    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE
    from demi_amd.minification import randomDDMin
    from demi_amd.schedulers import EventTrace, SchedulerConfig
    ranks = ranks or Ranks()
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.trace_load(used)
    collective = ranks.attach(ctx) if ranks.world > 1 else "none (1 rank)"
    lim = T.Limits(len(rec), 0, 64, 1, fp.code, 0)          # sched.setMaxMessages(trace.size), lookingFor = the violation
    best = None
    for budget in (64 * ranks.world, 256 * ranks.world, 1024 * ranks.world):
        par = T.RandomDdminParams(R, 0, budget)
        ctx.random_ddmin(lim, par, seed_base=SEED_BASE)
        for _ in range(3):
            ranks.barrier()
            t = time.perf_counter()
            mcs, cons, batches, st = ctx.random_ddmin(lim, par, seed_base=SEED_BASE)
            dt = ranks.max(time.perf_counter() - t)
            if best is None or dt < best["seconds"]:
                best = {"seconds": dt, "mcs_len": len(mcs), "oracle_consultations": int(st.consultations), "launches": int(st.launches),
                        "executions": int(st.consultations) * R, "executions_per_s": int(st.consultations) * R / dt,
                        "executions_launched": int(st.replays), "executions_launched_per_s": int(st.replays) / dt,
                        "candidates_per_launch": batches, "max_candidates_all_ranks": budget, "verified": bool(st.verified),
                        "mcs": [int(i) for i in mcs],
                        "consulted_digest": __import__("hashlib").sha256(repr([(tuple(int(i) for i in c), bool(p)) for c, p in cons]).encode()).hexdigest()[:16]}
    # the sequential algorithm on the device: one launch per consultation (what the reference's DDMin does with its oracle)
    par = T.RandomDdminParams(R, sequential=1)
    ctx.random_ddmin(lim, par, seed_base=SEED_BASE)
    t = time.perf_counter()
    mcs_s, cons_s, _b, st_s = ctx.random_ddmin(lim, par, seed_base=SEED_BASE)
    dts = ranks.max(time.perf_counter() - t)
    best["sequential_native"] = {"seconds": dts, "launches": int(st_s.launches), "executions_per_s": int(st_s.consultations) * R / dts,
                                 "same_mcs": tuple(mcs_s) == tuple(best["mcs"])}
    best["n_gpus"] = ranks.world
    best["collective"] = collective
    if ranks.world > 1:
        every = ranks.gather({"mcs": best["mcs"], "digest": best["consulted_digest"]})
        best["same_mcs_and_consultations_on_every_rank"] = all(r["mcs"] == best["mcs"] and r["digest"] == best["consulted_digest"] for r in every)
        best["same_mcs_as_single_rank_sequential"] = best["sequential_native"]["same_mcs"]
    ctx.close()
    if ranks.world == 1:
        # round 4's path: the Python loop, sequential DDMin, the table interpreted
        randomDDMin(SchedulerConfig(model=model), EventTrace(rec, used), fp, max_executions=R, seed_base=SEED_BASE)
        t = time.perf_counter()
        mcs_r, dd_r, _v = randomDDMin(SchedulerConfig(model=model), EventTrace(rec, used), fp, max_executions=R, seed_base=SEED_BASE)
        dtr = time.perf_counter() - t
        best["python_loop_interpreted"] = {"seconds": dtr, "oracle_consultations": len(dd_r.consulted), "executions_per_s": len(dd_r.consulted) * R / dtr,
                                           "same_mcs": tuple(mcs_r) == tuple(best["mcs"]),
                                           "same_consultation_sequence": [(tuple(c), p) for c, p in dd_r.consulted] == [(tuple(c), p) for c, p in cons_s]}
        if cpu_baseline:
            # the reference's loop around the CPU oracle's RandomScheduler, one host thread per execution batch
            from oracle import oracle_py as O
            from demi_amd.minification import DDMin, UnmodifiedEventDag
            cores = os.cpu_count() or 1

            class _Sched:
                def getName(self):
                    return "RandomScheduler"

                def test(self, events, fp_, stats=None):
                    v = O.random_explore(model, used[list(events)], R, seed_base=SEED_BASE, limits=lim, n_threads=min(cores, R))
                    return True if (v["flags"] & T.V_VIOLATION).any() else None
            dd = DDMin(_Sched(), checkUnmodifed=False)
            t = time.perf_counter()
            want = dd.minimize(UnmodifiedEventDag(used), fp).get_all_events()
            dc = time.perf_counter() - t
            best["cpu_baseline"] = {"value": len(dd.consulted) * R / dc, "unit": "executions/s", "cores": min(cores, R), "kind": "port", "seconds": dc,
                                    "sample": "the same minimization: the reference's DDMin loop (Python) around oracle/demi_oracle.c, "
                                              "%d executions per consultation on %d host threads" % (R, min(cores, R)),
                                    "same_mcs_as_gpu": tuple(want) == tuple(best["mcs"]),
                                    "same_consultation_sequence_as_gpu": [(tuple(c), p) for c, p in dd.consulted] == [(tuple(c), p) for c, p in cons_s]}
    return best


def bench_ddmin(ctx_device, cpu_baseline=True, n=1 << 20):
    """BASELINE config 4: STSSched (no peek) replays of candidate subsequences of a 200-event failing Raft-5 execution."""
    import numpy as np
    import torch
    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE, raft5_config4
    model, events, lim = raft5_config4()
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    v = ctx.random_explore(4000, lim, seed_base=SEED_BASE)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec = ctx.random_get_trace(SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    ctx.replay_load(used, rec)
    ctx.model_specialize()
    rng = np.random.default_rng(0)
    masks = np.zeros((n, 4), dtype=np.uint64)
    keep = rng.random((n, len(used))) < 0.7
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    target = T.Limits(0, 0, 128, 1, vv.fingerprint, 0)
    dev = torch.device("cuda", ctx_device)
    d_masks = torch.from_numpy(masks.view(np.int64)).to(dev)
    d_out = torch.empty((n, 2), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    ctx.replay_batch_dev(d_masks.data_ptr(), 256, target, d_out.data_ptr(), stream=sp)      # compiles K2 for this table
    torch.cuda.synchronize(dev)
    res = {}
    for m, reps in ((256, 50), (4096, 20), (65536, 10), (n, 5)):
        if m > n:
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.replay_batch_dev(d_masks.data_ptr(), m, target, d_out.data_ptr(), stream=sp)
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        e0.record(stream)
        for _ in range(reps):
            ctx.replay_batch_dev(d_masks.data_ptr(), m, target, d_out.data_ptr(), stream=sp)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        res[m] = {"kernel_ms": e0.elapsed_time(e1) / reps, "wall_ms": (time.perf_counter() - t) * 1e3 / reps}
    got = d_out.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)
    # two launches in flight (round 6: a K2 launch has K1's shape - a resident grid that drains a work counter - and a context holds
    # a second set of K2's per-launch scratch): the 2^20-candidate launches dealt over two streams, each with its own verdict array
    two = None
    try:
        st2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        sp2 = [C.c_void_p(x.cuda_stream) for x in st2]
        d_out2 = [d_out, torch.empty_like(d_out)]
        torch.cuda.synchronize(dev)
        for i in range(4):
            ctx.replay_batch_dev(d_masks.data_ptr(), n, target, d_out2[i & 1].data_ptr(), stream=sp2[i & 1])
        torch.cuda.synchronize(dev)
        reps2 = 10
        t = time.perf_counter()
        for i in range(reps2):
            ctx.replay_batch_dev(d_masks.data_ptr(), n, target, d_out2[i & 1].data_ptr(), stream=sp2[i & 1])
        torch.cuda.synchronize(dev)
        two = {"wall_ms": (time.perf_counter() - t) * 1e3 / reps2, "launches": reps2,
               "same_verdicts": bool(torch.equal(d_out2[0], d_out2[1]))}
        two["replays_per_s"] = n / (two["wall_ms"] * 1e-3)
    except Exception as e:
        two = {"error": "%s: %s" % (type(e).__name__, e)}
    n_exp = int(sum(1 for e in rec if e["kind"] in (0, 1, 2, 3, 7) or (e["kind"] == 6 and e["flags"] & 1)))
    kms = res[n]["kernel_ms"]
    out = {"metric": "candidate subsequences replayed/sec (STSScheduler.test without peek, DDMin's oracle)", "unit": "replays/s",
           "value": n / ((two["wall_ms"] if two and "wall_ms" in two else res[n]["wall_ms"]) * 1e-3),
           "launches_in_flight": 2 if two and "wall_ms" in two else 1, "two_launches_in_flight": two,
           "one_launch_at_a_time": {"replays_per_s": n / (res[n]["wall_ms"] * 1e-3), "wall_ms": res[n]["wall_ms"], "kernel_ms": res[n]["kernel_ms"]},
           "config": {"workload": "raft5-synth, 200-event failing execution (%d externals used, %d recorded events, %d deliveries), "
                                  "%d random candidate subsequences per launch, masks and verdicts resident in HBM" %
                                  (len(used), len(rec), T.verdict_deliveries(vv.flags), n), "candidates_per_launch": n},
           "launch_floor": {"candidates": 256, "kernel_us": res[256]["kernel_ms"] * 1e3, "wall_us": res[256]["wall_ms"] * 1e3},
           "frontiers": {str(m): {"kernel_us": r["kernel_ms"] * 1e3, "wall_us": r["wall_ms"] * 1e3, "replays_per_s": m / (r["wall_ms"] * 1e-3)}
                         for m, r in res.items()},
           "still_violating": int((got["flags"] & T.V_VIOLATION).sum())}
    # algorithmic HBM bytes per candidate: 32 B mask in + 16 B verdict out; the lowered original trace (8 B x expected events)
    # is shared by every lane (read once per workgroup into LDS)
    alg = 48 * n + 8 * n_exp * ((n + 255) // 256)
    # (measured fabric-side bytes of one 2^20-candidate launch, when the counters of this round are committed: tools/profile_r4_k2k3.sh)
    k2_traffic = (_counters_profile("ddmin_counters.json") or {}).get("fabric_bytes_per_launch") if n == (1 << 20) else None
    out["roofline"] = roofline(alg, kms, k2_traffic, "k2_replay (specialised, hiprtc)",
                               "32 B mask + 16 B verdict per candidate, the lowered original trace (%d x 8 B) once per workgroup; "
                               "the replay itself is integer / LDS work" % n_exp)
    if n == (1 << 20):
        nviol = out["still_violating"]
        out["roofline"]["issue_model"] = issue_model(ctx, "ddmin", kms, lambda tr: tr.get("still_violating") == nviol)
    ctx.close()
    # what the replays are for: RunnerUtils.stsSchedDDMin on this execution, end to end (DDMin's decision tree on the host,
    # every frontier one launch), through the Python mirror of the reference's classes
    from demi_amd.minification import stsSchedDDMin
    from demi_amd.schedulers import EventTrace, STSScheduler, SchedulerConfig, ViolationFingerprint
    fp = ViolationFingerprint(vv.fingerprint)
    sts = STSScheduler(SchedulerConfig(model=model), EventTrace(rec, used), p_max=128, device=ctx_device)
    stsSchedDDMin(sts, used, fp, speculative_depth=4)            # (compilation, buffers)
    t = time.perf_counter()
    mcs, dd, _ver = stsSchedDDMin(sts, used, fp, speculative_depth=4)
    out["ddmin_end_to_end_python_mirror"] = {"seconds": time.perf_counter() - t, "externals": int(len(used)), "mcs_len": len(mcs),
                                             "oracle_consultations": len(dd.consulted), "launches": len(dd.batches),
                                             "replays_launched": int(dd.speculative_replays),
                                             "note": "the same search with DDMin's decision tree enumerated by demi_amd/minification.py"}
    sts.shutdown()
    # the same minimization in ONE call of the library (demi_ddmin: atoms, ddmin2, the speculative frontier and the K2 launches
    # natively - what a JVM host binds); the best of a few launch budgets, each timed as the best of 5
    ctx = _native.Context(ctx_device)
    ctx.model_load(model.to_struct())
    ctx.model_specialize()
    ctx.replay_load(used, rec)
    best = None
    for budget in (256, 1024, 4096):
        par = T.DdminParams(0, budget, 1, 1)
        ctx.ddmin(target, par)
        for _ in range(5):
            t = time.perf_counter()
            mcs_n, cons_n, batches_n, st = ctx.ddmin(target, par)
            dt = time.perf_counter() - t
            if best is None or dt < best["seconds"]:
                best = {"seconds": dt, "externals": int(len(used)), "mcs_len": len(mcs_n), "oracle_consultations": int(st.consultations),
                        "launches": int(st.launches), "replays_launched": int(st.replays), "candidates_per_launch": batches_n,
                        "max_candidates": budget, "same_mcs_as_the_python_mirror": tuple(mcs_n) == tuple(mcs)}
    out["ddmin_end_to_end"] = best
    ctx.close()
    # RunnerUtils.randomDDMin (RunnerUtils.scala:601-623): DDMin whose oracle is the RandomScheduler itself, R = 100 random
    # interleavings per candidate.  demi_random_ddmin: decision tree + speculative frontier in the library, one
    # launch = (frontier candidates x R) executions of the compiled K1, a workgroup per candidate; beside it round 4's shape - the
    # Python loop, one interpreted 100-lane launch per consultation.  executions_per_s counts what the SEQUENTIAL algorithm asks for
    # (consultations x R): speculation that is launched and not consulted is not throughput.
    try:
        out["random_ddmin_R100"] = _bench_random_ddmin(ctx_device, model, used, rec, fp, cpu_baseline)
    except Exception as e:           # never let a side measurement cost the record
        out["random_ddmin_R100"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if cpu_baseline:
        from oracle import oracle_py as O
        cores = os.cpu_count() or 1
        sample = masks[:min(n, 1 << 20)]
        c = O.sts_replay_batch(model, used, rec, sample[:4096], target, n_threads=cores)      # (threads and pages warm)
        dt, reps = 0.0, 0
        while dt < 3.0 and reps < 64:                   # a launch of the sample is short on many cores: repeat for a stable rate
            t = time.perf_counter()
            c = O.sts_replay_batch(model, used, rec, sample, target, n_threads=cores)
            dt += time.perf_counter() - t
            reps += 1
        fr = {}
        for m in (256, 4096, 65536):
            if m > len(sample):
                continue
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                O.sts_replay_batch(model, used, rec, sample[:m], target, n_threads=min(cores, m))
                ts.append(time.perf_counter() - t)
            fr[str(m)] = {"wall_us": sorted(ts)[2] * 1e6}
        # the same minimization with the oracle as DDMin's TestOracle: one replay at a time on one core (what the reference's
        # loop does), and the same speculative frontiers on all host threads
        class _OracleSTS:
            def __init__(self, threads):
                self.threads = threads

            def getName(self):
                return "OracleSTS"

            def _v(self, subs):
                from demi_amd.minification import events_to_mask
                mk = np.array([events_to_mask(x) for x in subs], dtype=np.uint64).reshape(-1, 4)
                return O.sts_replay_batch(model, used, rec, mk, target, n_threads=min(self.threads, max(1, len(subs))))

            def test(self, sub, fpx, stats):
                r = self._v([sub])[0]
                return r if r["flags"] & T.V_VIOLATION else None

            def test_batch(self, subs, fpx, stats):
                return [bool(f & T.V_VIOLATION) for f in self._v(subs)["flags"]]

        e2e = {}
        for name, depth_, threads in (("sequential_one_core", 0, 1), ("speculative_all_threads", 4, cores)):
            t = time.perf_counter()
            m2, d2, _ = stsSchedDDMin(_OracleSTS(threads), used, fp, speculative_depth=depth_)
            e2e[name] = {"seconds": time.perf_counter() - t, "mcs_len": len(m2), "same_mcs_as_gpu": list(m2) == list(mcs)}
        # and the library's own DDMin loop (demi_amd/csrc/ddmin_host.hpp) around the oracle's replays, one at a time on one core
        try:
            ts = []
            for _ in range(5):
                t = time.perf_counter()
                m3, _c3, _b3, st3 = O.ddmin(model, used, rec, target, T.DdminParams(1, 0, 1, 1), n_threads=1)
                ts.append(time.perf_counter() - t)
            e2e["native_loop_one_core"] = {"seconds": min(ts), "mcs_len": len(m3), "same_mcs_as_gpu": list(m3) == list(mcs),
                                           "replays": int(st3.replays)}
        except Exception as e:
            e2e["native_loop_one_core"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["cpu_baseline"] = {"value": len(sample) * reps / dt, "unit": "replays/s", "cores": cores, "kind": "port",
                               "ddmin_end_to_end": e2e,
                               "sample": "first %d of the same candidate masks x %d passes, oracle/demi_oracle.c on %d pthreads" % (len(sample), reps, cores),
                               "seconds": dt, "frontiers": fr, "bit_identical_to_gpu": bool((c == got[:len(sample)]).all())}
    return out


def _self_launch(n_gpus):
    """Re-run this command as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <same arguments>`; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["fuzz", "dpor", "ddmin", "config1", "config5", "big"], default="fuzz")
    ap.add_argument("--dpor-order", choices=["both", "rounds", "reference_order"], default="both",
                    help="--workload dpor / config5: which exploration order(s) to run (profiling passes use one; config5 always runs ROUNDS)")
    ap.add_argument("--schedules", type=int, default=N_PER_GPU, help="schedules per GPU per step")
    ap.add_argument("--config5-budget", type=int, default=None,
                    help="interleavings the config 5 record may explore (default: apps.shuffle8_dpor_config5's 2^20)")
    ap.add_argument("--p-max", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="fuzz line only (no dpor / ddmin records)")
    ap.add_argument("--no-specialize", action="store_true", help="interpret the transition table instead of compiling it")
    ap.add_argument("--no-prewarm", action="store_true")
    ap.add_argument("--launches-in-flight", type=int, choices=[1, 2], default=2,
                    help="fuzz: 2 = the steps dealt over two streams of the one context (the tail of a launch overlaps "
                         "the start of the next); 1 = one context, one stream")
    ap.add_argument("--cpu-sample", type=int, default=1 << 20)
    ap.add_argument("--wide-term0", type=int, default=0,
                    help="experiment, not the headline: the same raft lowered as DEMI_MODEL_WIDE with terms starting here (> 255)")
    ap.add_argument("--log-cap", type=int, default=0,
                    help="experiment, not the headline: the raft with a REAL log of this many entries in the nodes' arrays "
                         "(DEMI_MODEL_ARRAY, raft_model(log_cap)); parity green on the GPU, not yet timed (DESIGN section 8, item 0)")
    ap.add_argument("--real-fields", action="store_true",
                    help="with --log-cap: the messages carry akka-raft's own field sets (DEMI_MODEL_PAYLOADS(5): AppendEntries(term, "
                         "prevLogIndex, prevLogTerm, entry, leaderCommit), RequestVote(term, candidateId, lastLogTerm, lastLogIndex))")
    ap.add_argument("--strategy", choices=["random", "fifo"], default="random",
                    help="RandomizationStrategy: FullyRandom (the headline workload) or SrcDstFIFO")
    args = ap.parse_args()
    if args.real_fields and not args.log_cap:
        ap.error("--real-fields needs --log-cap N (the field sets include the log's prevLogIndex / prevLogTerm / entry)")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become the launcher.  One rank per GPU under torch.distributed.run
        # (the command the contract names), rendezvous on 127.0.0.1 at a port the kernel hands out; rank 0's one JSON line goes
        # to this process's stdout unchanged.  Under torchrun WORLD_SIZE is set and this branch is never taken.
        raise SystemExit(_self_launch(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist

    from demi_amd import _native, types as T
    from demi_amd.apps import SEED_BASE, raft5_config2

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Plumbing test of the N > 1 path on a box with ONE GPU (tests/test_comm_gpu.py): DEMI_BENCH_BACKEND=gloo,
    # DEMI_BENCH_ONE_GPU=1 (every rank on cuda:0) and DEMI_BENCH_COMM=host (the library's communicator over a host all-gather
    # callback instead of RCCL, which refuses two ranks on one device).  The driver's 8-GPU run uses none of them.
    backend = os.environ.get("DEMI_BENCH_BACKEND", "nccl")
    one_gpu = os.environ.get("DEMI_BENCH_ONE_GPU") == "1"
    host_comm = os.environ.get("DEMI_BENCH_COMM") == "host"
    if one_gpu:
        local_rank = 0
    if args.gpus > 1 or world > 1:
        assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if backend == "nccl" else torch.device("cpu")       # where torch.distributed's own small tensors live

    def attach(ctx):
        """The library's own communicator for this demi_ctx on every rank (demi_comm_*: ncclAllGather over xGMI behind the C ABI,
        what a JVM host would call); its unique id travels over torch.distributed.  Returns what carries the collectives; raises
        on every rank alike when it cannot be created (the sharded entry points need it)."""
        if host_comm:
            def _host_allgather(block: bytes) -> bytes:
                mine = torch.frombuffer(bytearray(block), dtype=torch.uint8)
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine)
                return b"".join(bytes(q.numpy().tobytes()) for q in parts)
            ctx.comm_create_host(rank, world, _host_allgather)
            return "demi_comm (host all-gather callback over torch.distributed %s: plumbing test)" % backend
        # (every rank takes part in the broadcast and the agreement whatever happens locally: nobody is left waiting)
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)
        err = None
        if rank == 0:
            try:
                uid[:128] = torch.tensor(list(_native.Context.comm_unique_id()), dtype=torch.uint8, device=dev)
                uid[128] = 1
            except Exception as e:
                err = "no RCCL unique id: %s" % e
        dist.broadcast(uid, 0)
        ok = int(uid[128].item()) == 1
        if ok:
            try:
                ctx.comm_create(bytes(uid[:128].cpu().tolist()), rank, world)
            except Exception as e:
                ok, err = False, "ncclCommInitRank: %s" % e
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if ok:
                ctx.comm_destroy()
            raise RuntimeError("library communicator unavailable (%s)" % (err or "another rank failed"))
        return "demi_comm (RCCL ncclAllGather inside libdemi_gpu.so)"

    ranks = Ranks(rank, world, attach if world > 1 else None, dist, cdev)

    if args.workload != "fuzz":
        assert world == 1 or args.workload in ("ddmin", "config5"), "the dpor / config1 records are single-GPU"
        t = time.perf_counter()
        if args.workload == "dpor":
            rec = bench_dpor(local_rank, cpu_baseline=not args.no_cpu_baseline,
                             orders=("rounds", "reference_order") if args.dpor_order == "both" else (args.dpor_order,))
        elif args.workload == "config5":
            rec = bench_config5(local_rank, cpu_baseline=not args.no_cpu_baseline, ranks=ranks, max_interleavings=args.config5_budget,
                                reference_order=args.dpor_order in ("both", "reference_order"))
        elif args.workload == "ddmin" and world > 1:
            rec = bench_ddmin_ranks(local_rank, ranks)
        else:
            rec = {"ddmin": bench_ddmin, "config1": bench_config1, "big": bench_big_tables}[args.workload](local_rank, cpu_baseline=not args.no_cpu_baseline)
        rec.update({"n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": (time.perf_counter() - t) * 1e3, "higher_is_better": True,
                    "scaling": "strong" if args.workload == "config5" else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic"})
        if rank == 0:
            print(json.dumps(rec))
        if world > 1:
            dist.destroy_process_group()
        return

    model, events, limits = raft5_config2()
    if args.wide_term0:
        from demi_amd.model import raft_model
        model = raft_model(5, term0=args.wide_term0, loglen0=300)
    if args.log_cap:
        from demi_amd.model import raft_model
        model = raft_model(5, log_cap=args.log_cap, real_fields=args.real_fields)
    limits.p_max = args.p_max
    limits.strategy = T.STRATEGY_SRC_DST_FIFO if args.strategy == "fifo" else T.STRATEGY_FULLY_RANDOM
    n = args.schedules
    ctx = _native.Context(local_rank)
    ctx.model_load(model.to_struct())
    ctx.trace_load(events)
    specialized = False
    jit_compile_s = None
    if not args.no_specialize:
        # compile the loaded table to native code once (hiprtc, outside the timed region: reported as config.jit_compile_s); a
        # failure leaves the table interpreter in place and is reported in the JSON line
        try:
            tj = time.perf_counter()
            ctx.model_specialize()
            jit_compile_s = time.perf_counter() - tj
            specialized = ctx.is_specialized()
        except _native.DemiError as e:
            print("bench: specialisation unavailable, interpreting the table: %s" % e, file=sys.stderr)

    # one verdict array per timed step (16 MB each at 2^20 schedules; HBM has 288 GB): the violating executions of the whole
    # timed region are counted after it, on the device, without a synchronisation inside it
    keep_steps = min(args.steps, 256)
    vbuf = torch.empty((max(1, keep_steps), n, 2), dtype=torch.int64, device=dev)     # demi_verdict[n] per step
    verdicts = torch.empty((n, 2), dtype=torch.int64, device=dev)                     # warm-up / fixed-seed step
    # A K1 launch is a resident grid that drains a work counter: its last schedules run on a thinning device.  With a second
    # launch queued on another stream the workgroups that retire are replaced by the next launch's.  ONE context does that since
    # round 6 (it holds two sets of K1's per-launch scratch and alternates between them: include/demi_gpu.h): the timed steps
    # are dealt over two streams of the same context (--launches-in-flight 1: one stream).
    n_lanes = args.launches_in_flight
    ctxs = [ctx]
    # the found-violation sets are all-gathered by the library's own communicator on a stream of its own: the exchanges of all
    # steps follow each other in step order on every rank, whatever the lanes do; if RCCL cannot be initialised
    # there, torch.distributed's all_gather does the exchange (and the line says so)
    collective = "none (1 rank)"
    if world > 1:
        try:
            collective = attach(ctx) + ": demi_comm_allgather_dev"
        except RuntimeError as e:
            collective = "torch.distributed.all_gather (%s)" % e
    use_lib_comm = collective.startswith("demi_comm")

    class Lane:
        def __init__(self, c, st):
            self.ctx, self.stream, self.sp = c, st, C.c_void_p(st.cuda_stream)
            self.viol = torch.zeros((VIOL_CAP + 1, 2), dtype=torch.int64, device=dev)  # row 0 = count, then demi_violation[]
            self.gathered = torch.empty((world, VIOL_CAP + 1, 2), dtype=torch.int64, device=dev) if world > 1 else None
            self.extracted, self.exchanged = torch.cuda.Event(), torch.cuda.Event()
    lanes = [Lane(ctx, torch.cuda.current_stream() if n_lanes == 1 else torch.cuda.Stream(device=dev)) for _ in range(n_lanes)]
    comm_stream = lanes[0].stream if n_lanes == 1 or world == 1 else torch.cuda.Stream(device=dev)
    csp = C.c_void_p(comm_stream.cuda_stream)
    torch.cuda.synchronize()                 # (the buffers above exist before another stream touches them)
    stream, sp = lanes[0].stream, lanes[0].sp
    viol, gathered = lanes[0].viol, lanes[0].gathered

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    def step(i=None, out=None, index_base=None):
        """one pass of the hot path: K1 over n schedules, the found-violation set compacted, all-gathered with N > 1.
        Timed step i: fresh indices [(i * W + r) * n, ...), on lane i mod (launches in flight); otherwise the fixed indices
        [r * n, ...) on lane 0."""
        ln = lanes[i % n_lanes] if i is not None else lanes[0]
        if index_base is None:
            index_base = ((i * world + rank) if i is not None else rank) * n
        if out is None:
            out = vbuf[i % keep_steps] if i is not None else verdicts
        if i is not None:
            ev0[i].record(ln.stream)
        ln.ctx.random_explore_dev(n, limits, out.data_ptr(), seed_base=SEED_BASE + index_base, stream=ln.sp)
        if i is not None:
            ev1[i].record(ln.stream)
        ln.ctx.collect_violations_dev(out.data_ptr(), n, index_base, ln.viol[1:].data_ptr(), VIOL_CAP,
                                      ln.viol[0:1].data_ptr(), stream=ln.sp)
        if world > 1:
            # the exchange waits for this lane's extraction, and the lane's next extraction (which overwrites viol) for the exchange
            ln.extracted.record(ln.stream)
            comm_stream.wait_event(ln.extracted)
            if use_lib_comm:
                ctx.comm_allgather_dev(ln.viol.data_ptr(), ln.gathered.data_ptr(), ln.viol.numel() * 8, stream=csp)
            else:
                with torch.cuda.stream(comm_stream):
                    dist.all_gather_into_tensor(ln.gathered.view(-1, 2), ln.viol)
            ln.exchanged.record(comm_stream)
            ln.stream.wait_event(ln.exchanged)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # a freshly leased GPU idles at a low shader clock and ramps over about a second of load: run the same launches
    # untimed first so that the W warmup steps and the K timed steps see the clock a long-running job sees
    prewarm_s = 0.0
    k1_launches_before = 0            # K1 launches this process issues before the timed region (tools/summarize_prof.py finds the timed ones by it)
    if not args.no_prewarm:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < PREWARM_S:
            ctx.random_explore_dev(n, limits, verdicts.data_ptr(), seed_base=SEED_BASE + rank * n, stream=sp)
            torch.cuda.synchronize()
            k1_launches_before += 1
        prewarm_s = time.perf_counter() - t0
    for _ in range(args.warmup):
        step()
        k1_launches_before += 1
    # with two launches in flight a launch's duration includes its wait for the other's workgroups to retire; the same
    # launches one at a time (one context, one stream), untimed and BEFORE the timed region: what a launch takes with the
    # device to itself - the duration the committed counters and the issue model below belong to
    alone = None
    if n_lanes == 2:
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(2 * args.steps)]
        l0 = lanes[0]
        ta = time.perf_counter()
        for i in range(args.steps):
            base_a = (((args.steps + 1 + i) * world) + rank) * n
            ea[2 * i].record(l0.stream)
            l0.ctx.random_explore_dev(n, limits, verdicts.data_ptr(), seed_base=SEED_BASE + base_a, stream=l0.sp)
            ea[2 * i + 1].record(l0.stream)
            l0.ctx.collect_violations_dev(verdicts.data_ptr(), n, base_a, l0.viol[1:].data_ptr(), VIOL_CAP, l0.viol[0:1].data_ptr(), stream=l0.sp)
        torch.cuda.synchronize()
        ta = time.perf_counter() - ta
        k1_launches_before += args.steps
        alone = {"streams": 1, "steps": args.steps, "ms_per_step": ta / args.steps * 1e3, "value": n * args.steps / ta,
                 "unit": "schedules/s (this rank)", "kernel_ms": float(np.mean([ea[2 * i].elapsed_time(ea[2 * i + 1]) for i in range(args.steps)])),
                 "note": "untimed side run on lane 0's stream with an event pair per launch; its kernel_ms is what a launch takes with the "
                         "device to itself.  The one-stream bench line itself (--launches-in-flight 1) runs 4.06-4.08 ms per step on an MI355X"}
    if n_lanes == 2:                          # (the second scratch set's first launches: its allocation)
        for _ in range(max(2, args.warmup)):
            lanes[1].ctx.random_explore_dev(n, limits, verdicts.data_ptr(), seed_base=SEED_BASE + rank * n, stream=lanes[1].sp)
            k1_launches_before += 1
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    kernel_ms_alone = alone["kernel_ms"] if alone else kernel_ms

    # ---- what the timed region found: the violating executions of its steps, by delivery-sequence hash and by fingerprint
    # (bugs/hr = distinct violating schedules per wall-clock hour); counted here, after the timed region
    kept = vbuf[:keep_steps].reshape(-1, 2)
    vmask = (kept[:, 0] & T.V_VIOLATION) != 0
    my_hashes = torch.unique(kept[vmask, 1]).cpu().numpy()
    my_fps = torch.unique((kept[vmask, 0] >> 32) & 0xFFFFFFFF).cpu().numpy()
    my_viol = int(vmask.sum().item())
    parts = ranks.gather((my_viol, my_hashes, my_fps))
    timed_viol = sum(p[0] for p in parts)
    timed_hashes = int(len(np.unique(np.concatenate([p[1] for p in parts])))) if parts else 0
    timed_fps = int(len(np.unique(np.concatenate([p[2] for p in parts])))) if parts else 0
    counted_frac = keep_steps / float(args.steps)

    # ---- one more, untimed, step on the fixed indices [r * n, ...): the found-violation set every rank ends up with, and the
    # verdicts the CPU oracle is compared with bit for bit
    from demi_amd.distributed import merge_violation_sets
    step()
    sync()
    vparts = [gathered[r] for r in range(world)] if world > 1 else [viol]
    vset = merge_violation_sets([p.cpu().numpy() for p in vparts], VIOL_CAP)
    index_base = rank * n

    out = None
    if rank == 0:
        total = world * n * args.steps
        value = total / dt
        # algorithmic bytes of one K1 launch: 16 B verdict per schedule out, plus the
        # trace and the transition table streamed once per resident workgroup
        blocks = min((n + 255) // 256, torch.cuda.get_device_properties(dev).multi_processor_count *
                     ((3 if specialized else 2) if args.strategy == "fifo" else 4 if getattr(model, "wide", False) else (6 if specialized else 3)))
        shared = 8 * len(events) + 4 * len(model.code) + 4 * len(model.handler_start) + 8 * 8 + 32 * 4 + 132 * 4 + 64 * 4
        alg_bytes = 16 * n + blocks * shared
        # measured on this box, right after the timed region: the shader clock under load and the SIMD cycles per wave64
        # integer VALU instruction (alone on the SIMD / with 6 waves competing as in K1)
        probe = None
        try:
            p1, p6 = ctx.device_probe(1, 20000), ctx.device_probe(6, 6000)
            s6, m6, d6 = ctx.device_probe(6, 6000, 1), ctx.device_probe(6, 6000, 2), ctx.device_probe(6, 6000, 3)
            probe = {"shader_clock_ghz": p6.shader_clock_ghz, "simd_cycles_per_int_valu_1_wave": p1.cycles_per_valu,
                     "simd_cycles_per_int_valu_6_waves": p6.cycles_per_valu,
                     "simd_cycles_per_salu_6_waves": s6.cycles_per_valu,
                     "simd_cycles_per_inst_valu_salu_alternating_6_waves": m6.cycles_per_valu,
                     "simd_cycles_per_inst_divergent_if_6_waves": d6.cycles_per_valu,
                     "how": "demi_device_probe_mix: s_memtime cycles per 100 MHz wall_clock64 tick; unrolled passes of "
                            "independent integer VALU instructions / SALU instructions / the two alternating / the code of a "
                            "divergent two-instruction if, with 6 waves per SIMD as in K1"}
        except Exception as e:
            print("bench: device probe failed: %s" % e, file=sys.stderr)
        # rocprofv3 counters of THIS kernel build on this workload (tools/profile_r4.sh writes profiles/k1_counters.json with
        # the duration it saw): used only for the SAME kernel build - the profile records the code object's identity
        # (demi_model_code_id); a profile without one is accepted when its duration is within 10 % of this run's.
        traffic, issue, stale = None, None, None
        code_id = "%016x" % ctx.code_id() if specialized else None
        if os.path.exists(K1_COUNTERS) and specialized and args.strategy == "random" and n == N_PER_GPU:
            with open(K1_COUNTERS) as f:
                ctr = json.load(f)
            pms = float(ctr.get("kernel_ms", 0.0))
            same = (ctr["code_id"] == code_id) if ctr.get("code_id") else (pms > 0 and abs(pms - kernel_ms_alone) / pms <= 0.10)
            if same:
                traffic = ctr.get("fabric_bytes_per_launch")
                if probe and "SQ_INSTS_VALU" in ctr:
                    props = torch.cuda.get_device_properties(dev)
                    cus = props.multi_processor_count
                    clk = probe["shader_clock_ghz"] * 1e9
                    valu, salu = ctr["SQ_INSTS_VALU"], ctr["SQ_INSTS_SALU"]
                    other = ctr.get("SQ_INSTS_LDS", 0) + ctr.get("SQ_INSTS_VMEM_RD", 0) + ctr.get("SQ_INSTS_VMEM_WR", 0)
                    cyc = kernel_ms_alone * 1e-3 * clk          # (a launch with the device to itself: what the counters were taken on)
                    issue = {"valu_insts_per_launch": valu, "salu_insts_per_launch": salu,
                             "active_lanes_per_valu_inst": ctr.get("SQ_THREAD_CYCLES_VALU", 0) / valu if valu else None,
                             "valu_alone_frac": valu * probe["simd_cycles_per_int_valu_6_waves"] / (cus * 4) / cyc,
                             "salu_alone_frac": salu * probe["simd_cycles_per_salu_6_waves"] / (cus * 4) / cyc,
                             "issue_frac_straight_line": (valu + salu + other) * probe["simd_cycles_per_inst_valu_salu_alternating_6_waves"] / (cus * 4) / cyc,
                             "issue_frac_branchy": (valu + salu + other) * probe["simd_cycles_per_inst_divergent_if_6_waves"] / (cus * 4) / cyc,
                             "clock_hz": clk, "source": "profiles/k1_counters.json (rocprofv3 --pmc, kernel_ms %.3f there); the clock and "
                             "the SIMD cycles per instruction of each kind are measured in this run (roofline.probe): the share "
                             "of the kernel's duration that issuing its instructions takes lies between the straight-line and "
                             "the branchy figure" % pms}
            else:
                stale = "profiles/k1_counters.json describes another kernel build (code id %s, %.3f ms there; this run: %s, %.3f ms): counters not quoted" % (ctr.get("code_id"), pms, code_id, kernel_ms)
        out = {
            "metric": "candidate schedules evaluated/sec on Raft-5 fuzz (RandomScheduler executions)",
            "value": value, "unit": "schedules/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "raft5-synth (table-encoded akka-raft stand-in), 5 actors, frozen 50-event "
                                   "Fuzzer-distribution trace, %d random interleavings per GPU per step" % n,
                       "schedules_per_gpu_per_step": n, "max_messages": int(limits.max_messages),
                       "invariant_check_interval": int(limits.invariant_check_interval), "p_max": int(limits.p_max),
                       "randomization_strategy": "SrcDstFIFO" if args.strategy == "fifo" else "FullyRandom",
                       "table_compiled_to_native_code": specialized, "jit_compile_s": jit_compile_s,
                       "wide_register_window": bool(getattr(model, "wide", False)), "array_elements_per_actor": int(getattr(model, "array_len", 0)),
                       "payload_fields_per_message": int(getattr(model, "payloads", 2)), "seed_base": SEED_BASE,
                       "seeds": "timed step i of rank r: schedule indices [(i * %d + r) * n, ... + n) - every timed step evaluates fresh "
                                "seeds; the untimed last step: [r * n, ... + n)" % world,
                       "parallelism": "schedule-index range sharded, %d rank(s)" % world, "collective": collective,
                       "launches_in_flight": "%d (ONE demi_ctx; the timed steps dealt over %d stream(s))" % (n_lanes, n_lanes),
                       "untimed_prewarm_s": prewarm_s, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
            # distinct bugs found in the TIMED region per wall-clock hour of it: by delivery-sequence hash - two
            # executions count once only if every delivered message and every final state agree - and by fingerprint
            "bugs_per_hr": timed_hashes / counted_frac / dt * 3600.0,
            "bugs_per_hr_by_fingerprint": timed_fps / dt * 3600.0,
            "timed_region": {"schedules": total, "seconds": dt, "violating_executions": timed_viol,
                             "k1_launches_before": k1_launches_before, "k1_launches": args.steps,
                             "distinct_violating_delivery_hashes": timed_hashes, "distinct_fingerprints": timed_fps,
                             "steps_counted": keep_steps},
            "violations_last_step": int(len(vset)),
            "distinct_fingerprints_last_step": int(len(np.unique(vset["fingerprint"]))) if len(vset) else 0,
            "roofline": roofline(alg_bytes, kernel_ms, traffic,
                                 ("k1_random_explore<false, true>" if args.strategy == "fifo" else "k1_random_explore<false, false>") +
                                 (" (specialised, hiprtc)" if specialized else ""),
                                 "integer / latency-bound simulation: algorithmic HBM traffic is 16 B per schedule, so the HBM "
                                 "fraction is tiny by construction (SURVEY 8d). `traffic` = fabric-side bytes (L2 <-> Infinity Cache / "
                                 "HBM; FETCH_SIZE x 2 + WRITE_SIZE as calibrated in profiles/), dominated by the pending sets the "
                                 "specialised build keeps in a [slot][lane] scratch instead of LDS (24 waves per CU): a measured trade, "
                                 "DESIGN.md section 4 K1; that working set (~47 MB) fits the 256 MiB Infinity Cache",
                                 {"issue_model": issue, "probe": probe, "counters_stale": stale, "kernel_code_id": code_id,
                                  "kernel_ms_alone": kernel_ms_alone,
                                  "kernel_ms_note": "kernel_ms = a launch's duration in the timed region, by events on its stream; with two "
                                                    "launches in flight that includes its wait for the other launch's workgroups to retire (the "
                                                    "device's time per launch is ms_per_step); kernel_ms_alone = the same launch with the "
                                                    "device to itself (one_launch_at_a_time), what the counters and the issue model belong to"}),
        }
        if alone:
            out["one_launch_at_a_time"] = alone
        if world == 1:
            # what a JVM host sees through demi_random_explore (caller's pageable host buffer for the verdicts): the same
            # launch plus the copy over PCIe.  Never `value` (inputs / outputs resident in HBM); reported beside it.
            try:
                hv = np.zeros(n, dtype=T.VERDICT_DTYPE)
                hv["hash"] = 1                        # (pages touched: the caller's buffer exists before the call)

                def host_call():
                    rc = _native.lib().demi_random_explore(ctx._h, C.c_uint64(SEED_BASE + index_base), None, n, C.byref(limits), hv.ctypes.data)
                    assert rc == 0
                host_call()
                th = time.perf_counter()
                for _ in range(3):
                    host_call()
                th = (time.perf_counter() - th) / 3
                out["pcie_inclusive"] = {"entry_point": "demi_random_explore (16 B verdict per schedule copied into a pageable host buffer)",
                                         "ms_per_step": th * 1e3, "value": n / th, "unit": "schedules/s",
                                         "same_verdicts_as_the_resident_path": bool((hv == verdicts.cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)).all())}
                # the same through demi_random_explore_submit / _wait - what GpuRandomScheduler.explore calls: three calls
                # outstanding in the one context, (a) every verdict copied into the caller's host buffer, (b) only the flagged
                # executions (what explore() needs); K steps of fresh seeds each, host wall clock around the whole loop
                hv2 = [np.zeros(n, dtype=T.VERDICT_DTYPE) for _ in range(2)]
                for h_ in hv2:
                    h_["hash"] = 1
                def piped(with_verdicts, steps):
                    # submit(k + 2); wait(k): two calls submitted ahead of the one waited for (include/demi_gpu.h says why)
                    tk = [ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + index_base, want_verdicts=with_verdicts)]
                    if steps > 1:
                        tk.append(ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (steps + 2) * n, want_verdicts=with_verdicts))
                    found = 0
                    for j in range(steps):
                        if j + 2 < steps:
                            tk.append(ctx.random_explore_submit(n, limits, seed_base=SEED_BASE + (steps + 3 + j) * n, want_verdicts=with_verdicts))
                        _h, cnt, _f = ctx.random_explore_wait(tk[j], out=hv2[j & 1] if with_verdicts else None)
                        found += cnt
                    return found
                piped(True, 2)
                same_p = bool((hv2[0] == hv).all())            # (the first call of the loop runs the fixed seeds)
                ksteps = max(4, args.steps)
                tp = time.perf_counter(); piped(True, ksteps); tp = (time.perf_counter() - tp) / ksteps
                tf = time.perf_counter(); nf = piped(False, ksteps); tf = (time.perf_counter() - tf) / ksteps
                out["pcie_inclusive"]["pipelined"] = {
                    "entry_point": "demi_random_explore_submit / demi_random_explore_wait (submit(k + 2), wait(k): three calls outstanding in one demi_ctx, two streams of its own)",
                    "every_verdict_to_host": {"ms_per_step": tp * 1e3, "value": n / tp, "unit": "schedules/s", "steps": ksteps,
                                              "same_verdicts_as_the_resident_path": same_p},
                    "flagged_executions_to_host": {"ms_per_step": tf * 1e3, "value": n / tf, "unit": "schedules/s", "steps": ksteps,
                                                   "violating_executions_returned": int(nf)}}
            except Exception as e:           # never let a side measurement break the bench line
                print("bench: host-buffer measurement unavailable: %s" % e, file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle_py as O
            cores = os.cpu_count() or 1
            m = min(args.cpu_sample, n)
            tc = time.perf_counter()
            cpu = O.random_explore(model, events, m, seed_base=SEED_BASE, limits=limits, n_threads=cores)
            tcpu = time.perf_counter() - tc
            same = bool((cpu == verdicts[:m].cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)).all())
            # the oracle on ONE thread as well (what a single JVM scheduler corresponds to: the reference cannot
            # use more than one core for executions, Instrumenter.scala:1289-1296)
            m1 = min(m, 1 << 15)
            t1 = time.perf_counter()
            cpu1 = O.random_explore(model, events, m1, seed_base=SEED_BASE, limits=limits, n_threads=1)
            t1 = time.perf_counter() - t1
            out["cpu_baseline"] = {"value": m / tcpu, "unit": "schedules/s", "cores": cores, "kind": "port",
                                   "sample": "first %d schedules of the untimed fixed-seed step, oracle/demi_oracle.c with %d "
                                             "pthreads (restated CPU oracle, not the DEMi JVM)" % (m, cores),
                                   "seconds": tcpu, "bit_identical_to_gpu": same,
                                   "single_thread": {"value": m1 / t1, "unit": "schedules/s", "cores": 1, "seconds": t1,
                                                     "sample": "first %d schedules, one thread" % m1,
                                                     "bit_identical_to_gpu": bool((cpu1 == verdicts[:m1].cpu().numpy().view(T.VERDICT_DTYPE).reshape(-1)).all())}}
    for c in ctxs:
        c.close()
    del vbuf
    torch.cuda.empty_cache()
    if not args.no_secondary:
        sec = {}
        if world == 1:
            for name, fn in (("config1", bench_config1), ("dpor", bench_dpor), ("ddmin", bench_ddmin), ("more_than_8_actors", bench_big_tables),
                             ("config5", lambda d, cpu_baseline: bench_config5(d, cpu_baseline=cpu_baseline, max_interleavings=args.config5_budget))):
                try:
                    sec[name] = fn(local_rank, cpu_baseline=not args.no_cpu_baseline)
                except Exception as e:          # a secondary record must never cost the headline line
                    sec[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        else:
            # BASELINE configs 4 and 5 over the N ranks.  Every rank runs the record (the sharded entry points are collective);
            # an exception on one rank is agreed on before anybody enters the record's collectives a second time.
            for name, fn in (("ddmin", lambda: bench_ddmin_ranks(local_rank, ranks)),
                             ("config5", lambda: bench_config5(local_rank, cpu_baseline=False, ranks=ranks, small=False,
                                                               max_interleavings=args.config5_budget))):
                try:
                    r = fn()
                    err = None
                except Exception as e:
                    r, err = None, "%s: %s" % (type(e).__name__, e)
                errs = [x for x in ranks.gather(err) if x]
                sec[name] = r if not errs else {"error": errs[0], "failed_ranks": len(errs)}
                if errs:
                    break                       # (a rank that failed inside a collective may have left the others' communicators unusable)
        if out is not None:
            out["secondary"] = sec
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
