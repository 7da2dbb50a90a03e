"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never from the
product package (demi_amd/)."""
import ctypes as C
import os
import subprocess

import numpy as np

from demi_amd import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """make decides what is stale (the oracle itself and the DPOR bookkeeping harness)."""
    so = os.path.join(_HERE, "_build", "liboracle.so")
    subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return so


def dpor_explore_reference_resident(model, externals, params, search, n_threads=None):
    """REFERENCE order with the results resident on the (restated) device: explore_reference_resident of dpor_host.hpp over the
    CPU stand-in of ResidentDev::round_ref.  Returns (verdicts, prefix_len, rounds, first violating trace, stats, pair counts
    [reported, after the parent filter, after the snapshot filter = held by the device, after the fetch's second filter =
    what crosses PCIe, record fetches, interleavings fetched])."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    H.harness_dpor_explore_reference_resident.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.POINTER(T.DporParams),
                                                          C.POINTER(T.DporSearch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                                          C.POINTER(C.c_uint32), C.POINTER(T.DporStats), C.c_void_p, C.c_void_p]
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    cap = search.max_interleavings
    verdicts = np.zeros(cap, dtype=T.VERDICT_DTYPE)
    plen = np.zeros(cap, dtype=np.uint32)
    rounds = np.zeros(cap, dtype=np.uint32)
    vtrace = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
    vlen = C.c_uint32(0)
    stats = T.DporStats()
    counts = np.zeros(6, dtype=np.uint64)
    secs = np.zeros(8, dtype=np.float64)
    rc = H.harness_dpor_explore_reference_resident(C.byref(ms), ev.ctypes.data, len(ev), C.byref(params), C.byref(search),
                                                   n_threads or (os.cpu_count() or 1), verdicts.ctypes.data, plen.ctypes.data,
                                                   rounds.ctypes.data, vtrace.ctypes.data, C.byref(vlen), C.byref(stats), secs.ctypes.data, counts.ctypes.data)
    assert rc == 0, rc
    n = int(stats.interleavings)
    return verdicts[:n].copy(), plen[:n].copy(), rounds[:int(stats.launches)].copy(), vtrace[:vlen.value].copy(), stats, counts


def dpor_explore(model, externals, params, search, n_threads=None, resident=False):
    """The whole DPOR exploration on the CPU: the product's host bookkeeping (demi_amd/csrc/dpor_host.hpp) around this
    oracle's interleavings, `n_threads` of them at a time.  Returns (verdicts, prefix_len, rounds, first violating trace,
    stats, seconds[run, fetch + absorb, get_next]).
    resident: the device-resident bookkeeping (explored-pair table, enqueue decision, trace arena: k3_pairs.hpp) restated
    sequentially, under the host loop that drives the kernels (explore_rounds_resident)."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    H.harness_dpor_explore.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.POINTER(T.DporParams),
                                       C.POINTER(T.DporSearch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_uint32), C.POINTER(T.DporStats), C.c_void_p]
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    cap = search.max_interleavings
    verdicts = np.zeros(cap, dtype=T.VERDICT_DTYPE)
    plen = np.zeros(cap, dtype=np.uint32)
    rounds = np.zeros(cap, dtype=np.uint32)
    vt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
    vl = C.c_uint32(0)
    stats = T.DporStats()
    secs = np.zeros(8, dtype=np.float64)
    args = [C.byref(ms), ev.ctypes.data, len(ev), C.byref(params), C.byref(search),
            n_threads or (os.cpu_count() or 1), verdicts.ctypes.data, plen.ctypes.data, rounds.ctypes.data,
            vt.ctypes.data, C.byref(vl), C.byref(stats), secs.ctypes.data]
    if resident:
        H.harness_dpor_explore_resident.argtypes = H.harness_dpor_explore.argtypes + [C.POINTER(C.c_uint64)]
        entries = C.c_uint64(0)
        rc = H.harness_dpor_explore_resident(*args, C.byref(entries))
        stats.table_entries = int(entries.value)
    else:
        rc = H.harness_dpor_explore(*args)
    assert rc == 0
    n = int(stats.interleavings)
    return verdicts[:n], plen[:n], rounds[:int(stats.launches)], vt[:vl.value], stats, secs


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_jrandom_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_jrandom_next_int.argtypes = [C.c_void_p]
        L.orc_jrandom_next_int.restype = C.c_int32
        L.orc_jrandom_next_int_bound.argtypes = [C.c_void_p, C.c_int32]
        L.orc_jrandom_next_int_bound.restype = C.c_int32
        L.orc_jrandom_next_double.argtypes = [C.c_void_p]
        L.orc_jrandom_next_double.restype = C.c_double
        L.orc_model_validate.argtypes = [C.POINTER(T.ModelStruct), C.c_char_p, C.c_size_t]
        L.orc_trace_validate.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_char_p, C.c_size_t]
        L.orc_invariant.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32]
        L.orc_invariant.restype = C.c_uint32
        L.orc_random_execute.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_uint64,
                                         C.POINTER(T.Limits), C.POINTER(T.Verdict), C.c_void_p, C.c_uint32,
                                         C.POINTER(C.c_uint32), C.c_void_p]
        L.orc_random_explore.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_uint64,
                                         C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p, C.c_int]
        L.orc_vm_run.argtypes = [C.POINTER(T.ModelStruct), C.c_uint32, C.c_void_p, C.c_uint8, C.c_uint8,
                                 C.c_uint16, C.c_uint16, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_vm_run_area.argtypes = [C.POINTER(T.ModelStruct), C.c_uint32, C.c_void_p, C.c_uint8, C.c_uint8,
                                      C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_pay_area.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p]
        L.orc_pay_area.restype = C.c_uint64
        L.orc_sts_replay_batch.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p, C.c_int]
        L.orc_sts_removal_batch.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                            C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p, C.c_int]
        L.orc_sts_removal.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                      C.c_void_p, C.c_uint32, C.POINTER(T.Limits), C.POINTER(T.Verdict), C.c_void_p]
        L.orc_dpor_execute.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                       C.POINTER(T.DporParams), C.POINTER(T.Verdict), C.c_void_p, C.POINTER(C.c_uint32),
                                       C.c_void_p, C.POINTER(C.c_uint32)]
        _LIB = L
    return _LIB


class Effect(C.Structure):
    """orc_effect (demi_oracle.h): kind 0 send, 1 tset, 2 trep, 3 tcancel, 4 crash."""
    _fields_ = [("kind", C.c_uint8), ("target", C.c_uint8), ("msg_type", C.c_uint8), ("p0", C.c_uint16), ("p1", C.c_uint16),
                ("area", C.c_uint64)]


class JRandom:
    def __init__(self, seed):
        self._s = C.c_uint64(0)
        lib().orc_jrandom_seed(C.byref(self._s), C.c_uint64(seed & 0xFFFFFFFFFFFFFFFF))

    def next_int(self, bound=None):
        if bound is None:
            return lib().orc_jrandom_next_int(C.byref(self._s))
        return lib().orc_jrandom_next_int_bound(C.byref(self._s), bound)

    def next_double(self):
        return lib().orc_jrandom_next_double(C.byref(self._s))


_EXT_AREAS = None       # (kept alive while the oracle holds the pointer)


def set_ext_areas(areas):
    """orc_set_ext_areas: the payload areas of the external Sends of the trace the NEXT calls are given (a DEMI_MODEL_PAYLOADS
    table); None clears."""
    global _EXT_AREAS
    L = lib()
    L.orc_set_ext_areas.argtypes = [C.c_void_p, C.c_uint32]
    L.orc_set_ext_areas.restype = None
    if areas is None:
        _EXT_AREAS = None
        L.orc_set_ext_areas(None, 0)
        return
    _EXT_AREAS = np.ascontiguousarray(areas, dtype=np.uint64)
    L.orc_set_ext_areas(_EXT_AREAS.ctypes.data, len(_EXT_AREAS))


def model_validate(model):
    err = C.create_string_buffer(256)
    ms = model.to_struct()
    rc = lib().orc_model_validate(C.byref(ms), err, 256)
    return rc, err.value.decode()


def trace_validate(model, events):
    err = C.create_string_buffer(256)
    ms = model.to_struct()
    ev = np.ascontiguousarray(events)
    rc = lib().orc_trace_validate(C.byref(ms), ev.ctypes.data, len(ev), err, 256)
    return rc, err.value.decode()


def random_explore(model, events, n, seed_base=0, seeds=None, limits=None, n_threads=1):
    """n RandomScheduler executions on the CPU; returns a VERDICT_DTYPE array."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(events)
    out = np.zeros(n, dtype=T.VERDICT_DTYPE)
    sp = None
    if seeds is not None:
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        sp = seeds.ctypes.data
    rc = lib().orc_random_explore(C.byref(ms), ev.ctypes.data, len(ev), C.c_uint64(seed_base), sp, n,
                                  C.byref(limits), out.ctypes.data, n_threads)
    assert rc == 0
    return out


def random_execute(model, events, seed, limits, record=True):
    """One execution; returns (verdict, recorded events array, final states)."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(events)
    v = T.Verdict()
    rec = np.zeros(T.MAX_REC_EVENTS, dtype=T.REC_EVENT_DTYPE)
    n_rec = C.c_uint32(0)
    states = np.zeros(model.n_actors * model.state_words, dtype=np.uint64)
    rc = lib().orc_random_execute(C.byref(ms), ev.ctypes.data, len(ev), C.c_uint64(seed), C.byref(limits),
                                  C.byref(v), rec.ctypes.data if record else None, len(rec), C.byref(n_rec),
                                  states.ctypes.data)
    assert rc == 0
    return v, rec[:min(n_rec.value, len(rec))].copy(), states


def random_execute_carried(model, events, seed, exec_index, limits):
    """Execution number exec_index of the RandomScheduler instance seeded `seed` (limits.executions_per_instance mode: one
    generator through the instance's executions); returns (verdict, recorded events, index of the execution the instance
    stopped at - exec_index, or an earlier violating one)."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(events)
    v = T.Verdict()
    rec = np.zeros(T.MAX_REC_EVENTS, dtype=T.REC_EVENT_DTYPE)
    n_rec, ran = C.c_uint32(0), C.c_uint32(0)
    L = lib()
    L.orc_random_execute_carried.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    rc = L.orc_random_execute_carried(C.byref(ms), ev.ctypes.data, len(ev), C.c_uint64(seed), exec_index, C.byref(limits),
                                      C.byref(v), rec.ctypes.data, len(rec), C.byref(n_rec), C.byref(ran))
    assert rc == 0
    return v, rec[:min(n_rec.value, len(rec))].copy(), ran.value


def sts_replay_batch(model, original_externals, original_trace, masks, limits, n_threads=1):
    """STSScheduler.test (no peek) for every candidate mask (uint64[n, 4]); VERDICT_DTYPE array."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(original_externals, dtype=T.EXT_EVENT_DTYPE)
    rec = T.rec_events(original_trace)
    masks = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(len(masks), dtype=T.VERDICT_DTYPE)
    rc = lib().orc_sts_replay_batch(C.byref(ms), ev.ctypes.data, len(ev), rec.ctypes.data, len(rec),
                                    masks.ctypes.data, len(masks), C.byref(limits), out.ctypes.data, n_threads)
    assert rc == 0
    return out


def sts_removal_batch(model, original_externals, original_trace, skips, limits, masks=None, n_threads=1):
    """STSScheduler.test of the trace minus the MsgEvent at index skips[i] (0xFFFFFFFF = none), all externals
    kept unless masks is given; VERDICT_DTYPE array."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(original_externals, dtype=T.EXT_EVENT_DTYPE)
    rec = T.rec_events(original_trace)
    skips = np.ascontiguousarray(skips, dtype=np.uint32)
    if masks is not None:
        masks = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
        assert len(masks) == len(skips)
    out = np.zeros(len(skips), dtype=T.VERDICT_DTYPE)
    rc = lib().orc_sts_removal_batch(C.byref(ms), ev.ctypes.data, len(ev), rec.ctypes.data, len(rec),
                                     masks.ctypes.data if masks is not None else None, skips.ctypes.data, len(skips),
                                     C.byref(limits), out.ctypes.data, n_threads)
    assert rc == 0
    return out


def sts_removal_kept(model, original_externals, original_trace, skip, limits, mask=None):
    """(verdict, kept uint8[n_rec]) of one removal candidate: kept marks the executed trace."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(original_externals, dtype=T.EXT_EVENT_DTYPE)
    rec = T.rec_events(original_trace)
    if mask is not None:
        mask = np.ascontiguousarray(mask, dtype=np.uint64).reshape(4)
    v = T.Verdict()
    kept = np.zeros(max(len(rec), 1), dtype=np.uint8)
    rc = lib().orc_sts_removal(C.byref(ms), ev.ctypes.data, len(ev), rec.ctypes.data, len(rec),
                               mask.ctypes.data if mask is not None else None, C.c_uint32(int(skip) & 0xFFFFFFFF),
                               C.byref(limits), C.byref(v), kept.ctypes.data)
    assert rc == 0
    return v, kept[:len(rec)]


def dpor_batch(model, externals, prefixes, params, shared=None):
    """One DPORwHeuristics interleaving per prefix on the CPU; same return shape as Context.dpor_batch.
    shared[i]: leading events of prefix i whose racing pairs the caller already has (0 = report all)."""
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    verdicts = np.zeros(len(prefixes), dtype=T.VERDICT_DTYPE)
    traces, pairs = [], []
    for i, p in enumerate(prefixes):
        keys = np.ascontiguousarray(np.asarray(p, dtype=T.DPOR_TRACE_DTYPE)["key"], dtype=np.uint64)
        v = T.Verdict()
        tr = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
        pr = np.zeros(max(1, params.max_pairs), dtype=T.DPOR_PAIR_DTYPE)
        tl, npr = C.c_uint32(0), C.c_uint32(0)
        rc = lib().orc_dpor_execute(C.byref(ms), ev.ctypes.data, len(ev), keys.ctypes.data if len(keys) else None, len(keys),
                                    int(shared[i]) if shared is not None else 0, C.byref(params), C.byref(v), tr.ctypes.data, C.byref(tl), pr.ctypes.data, C.byref(npr))
        assert rc == 0
        verdicts[i] = (v.flags, v.fingerprint, v.hash)
        traces.append(tr[:tl.value].copy())
        pairs.append(pr[:npr.value].copy())
    return verdicts, traces, pairs


def dpor_explore_sharded(model, externals, params, search, world):
    """The multi-GPU rounds of the device-resident exploration restated for `world` host ranks (threads exchanging their
    blocks through an in-process all-gather): returns one (verdicts, prefix_len, rounds, first violating trace, stats)
    per rank - they must all be equal, and equal to the single-rank exploration."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    H.harness_dpor_explore_sharded.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.POINTER(T.DporParams),
                                               C.POINTER(T.DporSearch), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    cap = search.max_interleavings
    verdicts = np.zeros((world, cap), dtype=T.VERDICT_DTYPE)
    plen = np.zeros((world, cap), dtype=np.uint32)
    rounds = np.zeros((world, cap), dtype=np.uint32)
    vt = np.zeros((world, T.DPOR_MAX_TRACE), dtype=T.DPOR_TRACE_DTYPE)
    vl = np.zeros(world, dtype=np.uint32)
    stats = (T.DporStats * world)()
    rc = H.harness_dpor_explore_sharded(C.byref(ms), ev.ctypes.data, len(ev), C.byref(params), C.byref(search), world,
                                        verdicts.ctypes.data, plen.ctypes.data, rounds.ctypes.data, vt.ctypes.data, vl.ctypes.data,
                                        C.cast(stats, C.c_void_p))
    assert rc == 0
    out = []
    for r in range(world):
        n = int(stats[r].interleavings)
        out.append((verdicts[r, :n], plen[r, :n], rounds[r, :int(stats[r].launches)], vt[r, :vl[r]], stats[r]))
    return out


def ddmin(model, original_externals, original_trace, limits, params=None, conjoined=None, n_threads=1, cap=4096):
    """demi_ddmin's host loop (demi_amd/csrc/ddmin_host.hpp) over this oracle's STSScheduler replays.  Returns (mcs indices,
    [(candidate indices, passes)] in consultation order, candidates per launch, stats)."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    ms = model.to_struct()
    ev = np.ascontiguousarray(original_externals, dtype=T.EXT_EVENT_DTYPE)
    rec = T.rec_events(original_trace)
    params = params or T.DdminParams()
    mcs = np.zeros(4, dtype=np.uint64)
    consulted = np.zeros((cap, 4), dtype=np.uint64)
    passed = np.zeros(cap, dtype=np.uint8)
    batches = np.zeros(cap, dtype=np.uint32)
    st = T.DdminStats()
    conj = None
    if conjoined is not None:
        conj = np.ascontiguousarray(conjoined, dtype=np.uint8)
    H.harness_ddmin.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(T.Limits),
                                C.POINTER(T.DdminParams), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                C.c_void_p, C.c_uint32, C.POINTER(T.DdminStats)]
    rc = H.harness_ddmin(C.byref(ms), ev.ctypes.data, len(ev), rec.ctypes.data, len(rec), C.byref(limits), C.byref(params),
                         conj.ctypes.data if conj is not None else None, n_threads, mcs.ctypes.data, consulted.ctypes.data,
                         passed.ctypes.data, cap, batches.ctypes.data, cap, C.byref(st))
    if rc:
        raise RuntimeError("harness_ddmin: %d" % rc)
    return T.mask_to_events(mcs), [(T.mask_to_events(consulted[i]), bool(passed[i])) for i in range(min(cap, st.consultations))], \
        [int(b) for b in batches[:st.launches]], st


def edit_distance_dpor_ddmin(model, externals, initial_trace, params, ip=None, n_threads=2, cap=4096):
    """demi_edit_distance_dpor_ddmin's host loop (demi_amd/csrc/incddmin_host.hpp) with this oracle's interleavings under every
    DPOR consultation.  Returns what demi_amd._native.Context.edit_distance_dpor_ddmin returns."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    init = np.ascontiguousarray(initial_trace, dtype=T.DPOR_TRACE_DTYPE)
    ip = ip or T.IncDdminParams()
    mcs = np.zeros(4, dtype=np.uint64)
    consulted = np.zeros((cap, 4), dtype=np.uint64)
    passed = np.zeros(cap, dtype=np.uint8)
    dist = np.zeros(cap, dtype=np.uint32)
    vt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
    st = T.IncDdminStats()
    H.harness_edit_distance_dpor_ddmin.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                                   C.POINTER(T.DporParams), C.POINTER(T.IncDdminParams), C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(T.IncDdminStats)]
    rc = H.harness_edit_distance_dpor_ddmin(C.byref(ms), ev.ctypes.data if len(ev) else None, len(ev), init.ctypes.data, len(init),
                                            C.byref(params), C.byref(ip), n_threads, mcs.ctypes.data, consulted.ctypes.data,
                                            passed.ctypes.data, dist.ctypes.data, cap, vt.ctypes.data, C.byref(st))
    if rc:
        raise RuntimeError("harness_edit_distance_dpor_ddmin: %d" % rc)
    n = min(cap, st.consultations)
    return T.mask_to_events(mcs), [(T.mask_to_events(consulted[i]), bool(passed[i]), int(dist[i])) for i in range(n)], \
        [(int(st.pass_distance[i]), int(st.pass_mcs_len[i])) for i in range(min(16, st.passes))], \
        (vt[:st.violation_len].copy() if st.violation_len else None), st


class OrderedState:
    """What one DPORwHeuristics instance keeps between its test() calls (queue, explored pairs): hand the same object to
    consecutive dpor_explore_ordered calls with search.resume = 1."""

    def __init__(self):
        build()
        self._H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
        self._H.harness_ordered_state_new.restype = C.c_void_p
        self._H.harness_ordered_state_free.argtypes = [C.c_void_p]
        self.ptr = C.c_void_p(self._H.harness_ordered_state_new())

    def __del__(self):
        if getattr(self, "ptr", None):
            self._H.harness_ordered_state_free(self.ptr)
            self.ptr = None


def dpor_explore_ordered(model, externals, params, search, original_trace=None, initial_trace=None, n_threads=None, state=None):
    """DPORwHeuristics with ArvindDistanceOrdering / setMaxDistance / setInitialTrace natively (dpor_host.hpp explore_rounds_ordered,
    what demi_dpor_explore runs for them) around this oracle's interleavings.  Returns (verdicts, prefix_len, rounds, first
    violating trace, stats)."""
    build()
    H = C.CDLL(os.path.join(_HERE, "_build", "dpor_host_harness.so"))
    H.harness_dpor_explore_ordered.argtypes = [C.POINTER(T.ModelStruct), C.c_void_p, C.c_uint32, C.POINTER(T.DporParams), C.POINTER(T.DporSearch),
                                               C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(T.DporStats), C.c_void_p]
    ms = model.to_struct()
    ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
    keys = np.ascontiguousarray(np.asarray(original_trace)["key"], dtype=np.uint64) if original_trace is not None else np.zeros(0, dtype=np.uint64)
    init = np.ascontiguousarray(initial_trace, dtype=T.DPOR_TRACE_DTYPE) if initial_trace is not None else np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)
    cap = search.max_interleavings
    verdicts = np.zeros(cap, dtype=T.VERDICT_DTYPE)
    plen = np.zeros(cap, dtype=np.uint32)
    rounds = np.zeros(cap, dtype=np.uint32)
    vt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
    vl = C.c_uint32(0)
    stats = T.DporStats()
    rc = H.harness_dpor_explore_ordered(C.byref(ms), ev.ctypes.data, len(ev), C.byref(params), C.byref(search),
                                        keys.ctypes.data if len(keys) else None, len(keys), init.ctypes.data if len(init) else None, len(init),
                                        n_threads or (os.cpu_count() or 1), verdicts.ctypes.data, plen.ctypes.data, rounds.ctypes.data,
                                        vt.ctypes.data, C.byref(vl), C.byref(stats), state.ptr if state is not None else None)
    assert rc == 0, rc
    n = int(stats.interleavings)
    return verdicts[:n].copy(), plen[:n].copy(), rounds[:int(stats.launches)].copy(), vt[:vl.value].copy(), stats
