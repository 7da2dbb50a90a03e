"""ctypes binding of libdemi_gpu.so (the C ABI of include/demi_gpu.h).

There is no fallback: if the shared library is missing or no MI355X is visible, the product path
raises.  Build with `python -c "import __graft_entry__ as g; g.build()"` or `make -C demi_amd/csrc`.
"""
import ctypes as C
import os

from . import types as T

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdemi_gpu.so")

EXPORTS = ["demi_ctx_create", "demi_ctx_destroy", "demi_last_error", "demi_version", "demi_model_load",
           "demi_trace_load", "demi_random_explore", "demi_random_explore_dev", "demi_random_get_trace", "demi_collect_violations_dev",
           "demi_replay_load", "demi_replay_batch", "demi_replay_batch_dev", "demi_dpor_load", "demi_dpor_batch", "demi_dpor_explore", "demi_random_explore_violations", "demi_random_get_trace_carried",
           "demi_replay_removal_batch", "demi_replay_get_kept", "demi_replay_recorded_len", "demi_ddmin", "demi_dpor_set_traces", "demi_model_specialize", "demi_model_is_specialized", "demi_model_code_id",
           "demi_specialize_check", "demi_specialize_source", "demi_specialize_source_k1", "demi_provenance_prune", "demi_device_probe", "demi_device_probe_mix", "demi_calib_rw", "demi_random_explore_flagged", "demi_collect_flagged_dev", "demi_random_explore_submit", "demi_random_explore_wait", "demi_trace_len",
           "demi_comm_unique_id", "demi_comm_create", "demi_comm_create_host", "demi_comm_destroy", "demi_comm_rank",
           "demi_comm_allgather_dev", "demi_random_explore_sharded", "demi_replay_batch_sharded", "demi_abi_version", "demi_replay_externals_len", "demi_edit_distance_dpor_ddmin", "demi_dpor_explored", "demi_random_ddmin", "demi_random_explore_candidates", "demi_ext_payload_areas"]

_lib = None
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)     # demi_allgather_fn


class DemiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("demi_gpu error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libdemi_gpu.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
    if not os.environ.get("DEMI_NO_TORCH"):
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64; loading it first makes
        # libdemi_gpu.so bind to the same runtime, so torch tensors / streams and our kernels share a
        # device context (loading the system runtime first leaves torch with "No HIP GPUs").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(LIB_PATH)
    L.demi_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.demi_ctx_destroy.argtypes = [C.c_void_p]
    L.demi_ctx_destroy.restype = None
    L.demi_last_error.argtypes = [C.c_void_p]
    L.demi_last_error.restype = C.c_char_p
    L.demi_version.restype = C.c_char_p
    L.demi_abi_version.argtypes = []
    L.demi_abi_version.restype = C.c_uint32
    if L.demi_abi_version() != T.ABI_VERSION:       # (struct layouts of another generation of include/demi_gpu.h)
        raise ImportError("libdemi_gpu.so has ABI generation %d, this binding was written for %d: rebuild (__graft_entry__.build())"
                          % (L.demi_abi_version(), T.ABI_VERSION))
    L.demi_replay_externals_len.argtypes = [C.c_void_p]
    L.demi_replay_externals_len.restype = C.c_uint32
    L.demi_trace_len.argtypes = [C.c_void_p]
    L.demi_trace_len.restype = C.c_uint32
    L.demi_model_load.argtypes = [C.c_void_p, C.POINTER(T.ModelStruct)]
    L.demi_model_specialize.argtypes = [C.c_void_p, C.c_int]
    L.demi_model_is_specialized.argtypes = [C.c_void_p]
    L.demi_model_code_id.argtypes = [C.c_void_p]
    L.demi_model_code_id.restype = C.c_uint64
    L.demi_specialize_check.argtypes = [C.POINTER(T.ModelStruct), C.c_char_p, C.c_size_t]
    L.demi_specialize_check.restype = C.c_long
    L.demi_dpor_set_traces.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.demi_dpor_set_traces.restype = C.c_int
    L.demi_ddmin.argtypes = [C.c_void_p, C.POINTER(T.Limits), C.POINTER(T.DdminParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(T.DdminStats)]
    L.demi_ddmin.restype = C.c_int
    L.demi_edit_distance_dpor_ddmin.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(T.DporParams),
                                                C.POINTER(T.IncDdminParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32,
                                                C.c_void_p, C.POINTER(T.IncDdminStats)]
    L.demi_edit_distance_dpor_ddmin.restype = C.c_int
    L.demi_specialize_source.argtypes = [C.POINTER(T.ModelStruct), C.c_char_p, C.c_size_t]
    L.demi_specialize_source.restype = C.c_long
    L.demi_specialize_source_k1.argtypes = [C.POINTER(T.ModelStruct), C.c_char_p, C.c_size_t]
    L.demi_specialize_source_k1.restype = C.c_long
    L.demi_trace_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.demi_ext_payload_areas.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.demi_random_explore.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p]
    L.demi_random_explore_dev.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.POINTER(T.Limits),
                                          C.c_void_p, C.c_void_p]
    L.demi_random_get_trace.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.POINTER(T.Verdict),
                                        C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    L.demi_random_get_trace_carried.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(T.Limits), C.POINTER(T.Verdict),
                                                C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.demi_collect_violations_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p,
                                              C.c_uint32, C.c_void_p, C.c_void_p]
    L.demi_replay_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.demi_replay_recorded_len.argtypes = [C.c_void_p]
    L.demi_replay_recorded_len.restype = C.c_uint32
    L.demi_replay_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p]
    L.demi_replay_batch_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p, C.c_void_p]
    L.demi_replay_removal_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p]
    L.demi_replay_get_kept.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(T.Limits), C.POINTER(T.Verdict),
                                       C.c_void_p]
    L.demi_dpor_load.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    L.demi_dpor_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(T.DporParams),
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.demi_dpor_explore.argtypes = [C.c_void_p, C.POINTER(T.DporParams), C.POINTER(T.DporSearch), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(T.DporStats)]
    L.demi_random_explore_candidates.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(T.Limits), C.c_void_p, C.c_void_p]
    L.demi_random_ddmin.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.POINTER(T.RandomDdminParams), C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(T.DdminStats)]
    L.demi_dpor_explored.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_void_p,
                                     C.POINTER(C.c_uint32)]
    L.demi_random_explore_violations.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(T.Limits), C.c_void_p,
                                                 C.c_uint32, C.POINTER(C.c_uint64)]
    L.demi_random_explore_flagged.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(T.Limits), C.c_uint32, C.c_void_p,
                                              C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.demi_random_explore_submit.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(T.Limits), C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.demi_random_explore_wait.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64),
                                           C.POINTER(C.c_uint64)]
    L.demi_collect_flagged_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_uint32,
                                           C.c_void_p, C.c_void_p]
    L.demi_comm_unique_id.argtypes = [C.c_void_p]
    L.demi_comm_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.demi_comm_create_host.argtypes = [C.c_void_p, C.c_int, C.c_int, ALLGATHER_FN, C.c_void_p]
    L.demi_comm_destroy.argtypes = [C.c_void_p]
    L.demi_comm_rank.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.demi_comm_allgather_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    L.demi_random_explore_sharded.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(T.Limits), C.c_void_p, C.c_uint32,
                                              C.POINTER(C.c_uint64)]
    L.demi_replay_batch_sharded.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(T.Limits), C.c_void_p]
    L.demi_device_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(T.ProbeResult)]
    L.demi_provenance_prune.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
    L.demi_device_probe_mix.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(T.ProbeResult)]
    L.demi_calib_rw.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint32]
    # every export has its argument types declared: an undeclared one would silently truncate pointers to 32 bits
    for name in EXPORTS:
        fn = getattr(L, name)
        assert fn.argtypes is not None or name in ("demi_version",), "no argtypes for %s" % name
    _lib = L
    return L


def specialize_check(model_struct):
    """Generate + compile the specialised K1 kernel for a model without a device: (code size, kernel name)."""
    log = C.create_string_buffer(4096)
    n = lib().demi_specialize_check(C.byref(model_struct), log, len(log))
    if n < 0:
        raise DemiError(n, log.value.decode(errors="replace"))
    return int(n), log.value.decode()


def specialize_source(model_struct, k1=False):
    """The C++ the specialiser generates for the model's handlers (k1: the RandomScheduler kernel's flavour, with the
    table's effect-slot schedule when it has one)."""
    buf = C.create_string_buffer(1 << 20)
    n = (lib().demi_specialize_source_k1 if k1 else lib().demi_specialize_source)(C.byref(model_struct), buf, len(buf))
    if n < 0:
        raise DemiError(n, buf.value.decode(errors="replace"))
    return buf.value.decode()


class Context:
    """demi_ctx: one per host thread, owns the device copies of the model and the trace."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().demi_ctx_create(device, C.byref(self._h))
        if rc != 0:
            raise DemiError(rc, "demi_ctx_create failed (no MI355X visible? the GPU path has no CPU fallback)")
        self.device = device

    def close(self):
        if self._h:
            lib().demi_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise DemiError(rc, lib().demi_last_error(self._h).decode())

    def model_load(self, model_struct):
        self._check(lib().demi_model_load(self._h, C.byref(model_struct)))

    def model_specialize(self, enable=True):
        """Compile the loaded table to native code (hiprtc) for the following random_explore launches."""
        self._check(lib().demi_model_specialize(self._h, 1 if enable else 0))

    def is_specialized(self):
        return bool(lib().demi_model_is_specialized(self._h))

    def code_id(self):
        """64-bit identity of the compiled K1 of the loaded model (0: interpreted)."""
        return int(lib().demi_model_code_id(self._h))

    def ext_payload_areas(self, areas):
        """demi_ext_payload_areas: the 48-bit payload areas (T.pay_area) of the external events of the NEXT trace_load / dpor_load -
        a DEMI_MODEL_PAYLOADS table's external Sends with all their fields.  None forgets a staged array."""
        import numpy as np
        if areas is None:
            self._check(lib().demi_ext_payload_areas(self._h, None, 0))
            return
        a = np.ascontiguousarray(areas, dtype=np.uint64)
        self._check(lib().demi_ext_payload_areas(self._h, a.ctypes.data if len(a) else None, len(a)))

    def trace_load(self, events, areas=None):
        import numpy as np
        ev = np.ascontiguousarray(events, dtype=T.EXT_EVENT_DTYPE)
        if areas is not None:
            self.ext_payload_areas(areas)
        self._check(lib().demi_trace_load(self._h, ev.ctypes.data if len(ev) else None, len(ev)))

    def random_explore(self, n, limits, seed_base=0, seeds=None):
        import numpy as np
        out = np.zeros(n, dtype=T.VERDICT_DTYPE)
        sp = None
        if seeds is not None:
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            k = max(1, int(limits.executions_per_instance))           # carried-generator mode: one seed per instance
            assert len(seeds) == (n + k - 1) // k
            sp = seeds.ctypes.data
        self._check(lib().demi_random_explore(self._h, C.c_uint64(seed_base), sp, n, C.byref(limits),
                                              out.ctypes.data if n else None))
        return out

    def random_explore_violations(self, n, limits, seed_base=0, cap=1 << 16):
        """n schedules; only the violating ones come back (VIOLATION_DTYPE, sorted by index) + their count."""
        import numpy as np
        out = np.zeros(cap, dtype=T.VIOLATION_DTYPE)
        cnt = C.c_uint64(0)
        self._check(lib().demi_random_explore_violations(self._h, C.c_uint64(seed_base), n, C.byref(limits),
                                                         out.ctypes.data, cap, C.byref(cnt)))
        return out[:min(cnt.value, cap)].copy(), int(cnt.value)

    def random_explore_flagged(self, n, limits, flag_mask, seed_base=0, cap=1 << 16):
        """n schedules; the entries whose verdict flags intersect flag_mask (sorted by index), their count and the lowest
        such index (exact even when the list is truncated to `cap`)."""
        import numpy as np
        out = np.zeros(cap, dtype=T.VIOLATION_DTYPE)
        cnt, first = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().demi_random_explore_flagged(self._h, C.c_uint64(seed_base), n, C.byref(limits), flag_mask,
                                                      out.ctypes.data, cap, C.byref(cnt), C.byref(first)))
        return out[:min(cnt.value, cap)].copy(), int(cnt.value), int(first.value)

    def random_explore_submit(self, n, limits, seed_base=0, flag_mask=0, want_verdicts=False):
        """explore() in pieces, up to three outstanding: enqueue n executions on a stream of the context's own; returns the ticket.
        want_verdicts: every verdict travels to the library's pinned memory behind the call (random_explore_wait(out=...))."""
        t = C.c_uint32(0)
        self._check(lib().demi_random_explore_submit(self._h, C.c_uint64(seed_base), n, C.byref(limits), flag_mask,
                                                     1 if want_verdicts else 0, C.byref(t)))
        return int(t.value)

    def random_explore_wait(self, ticket, out=None, cap=1 << 16):
        """-> (flagged entries sorted by index, their exact count, the lowest flagged index or 2^64 - 1); out: a VERDICT_DTYPE
        array of the call's n entries to receive every verdict, or None"""
        import numpy as np
        fl = np.zeros(cap, dtype=T.VIOLATION_DTYPE)
        cnt, first = C.c_uint64(0), C.c_uint64(0)
        self._check(lib().demi_random_explore_wait(self._h, ticket, out.ctypes.data if out is not None else None, fl.ctypes.data, cap,
                                                   C.byref(cnt), C.byref(first)))
        return fl[:min(cnt.value, cap)].copy(), int(cnt.value), int(first.value)

    def random_explore_dev(self, n, limits, d_out_ptr, seed_base=0, d_seeds_ptr=None, stream=None):
        self._check(lib().demi_random_explore_dev(self._h, C.c_uint64(seed_base), d_seeds_ptr, n, C.byref(limits),
                                                  d_out_ptr, stream))

    def collect_violations_dev(self, d_verdicts_ptr, n, index_base, d_out_ptr, cap, d_count_ptr, stream=None):
        self._check(lib().demi_collect_violations_dev(self._h, d_verdicts_ptr, n, C.c_uint64(index_base), d_out_ptr,
                                                      cap, d_count_ptr, stream))

    def replay_load(self, original_externals, original_trace):
        import numpy as np
        ev = np.ascontiguousarray(original_externals, dtype=T.EXT_EVENT_DTYPE)
        rec = T.rec_events(original_trace)
        self._check(lib().demi_replay_load(self._h, ev.ctypes.data if len(ev) else None, len(ev),
                                           rec.ctypes.data if len(rec) else None, len(rec)))

    def replay_batch(self, masks, limits):
        """masks: uint64[n, 4] (bit i of the 256-bit row = external event i kept)."""
        import numpy as np
        masks = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(len(masks), dtype=T.VERDICT_DTYPE)
        self._check(lib().demi_replay_batch(self._h, masks.ctypes.data if len(masks) else None, len(masks),
                                            C.byref(limits), out.ctypes.data if len(masks) else None))
        return out

    def replay_batch_dev(self, d_masks_ptr, n, limits, d_out_ptr, stream=None):
        """Device-resident form: masks [n][4] u64 and verdicts [n] stay in HBM; enqueued on `stream`, not synchronised."""
        self._check(lib().demi_replay_batch_dev(self._h, C.c_void_p(d_masks_ptr), C.c_uint64(n), C.byref(limits),
                                                C.c_void_p(d_out_ptr), stream))

    def replay_removal_batch(self, skips, limits, masks=None):
        """One STSScheduler.test per entry of skips: the loaded trace minus the delivery at that recorded-event
        index (0xFFFFFFFF = nothing removed); masks (uint64[n, 4]) default to every external kept."""
        import numpy as np
        skips = np.ascontiguousarray(skips, dtype=np.uint32)
        out = np.zeros(len(skips), dtype=T.VERDICT_DTYPE)
        if len(skips) == 0:
            return out
        mp = None
        if masks is not None:
            masks = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
            if len(masks) != len(skips):
                raise ValueError("one mask per removal candidate")
            mp = masks.ctypes.data
        self._check(lib().demi_replay_removal_batch(self._h, mp, skips.ctypes.data, len(skips), C.byref(limits),
                                                    out.ctypes.data))
        return out

    def ddmin(self, limits, params=None, conjoined=None, cap=4096):
        """RunnerUtils.stsSchedDDMin on the loaded replay, natively (demi_ddmin): (mcs indices, [(candidate indices, passes)] in
        consultation order, candidates per launch, stats)."""
        import numpy as np
        params = params or T.DdminParams()
        mcs = np.zeros(4, dtype=np.uint64)
        consulted = np.zeros((cap, 4), dtype=np.uint64)
        passed = np.zeros(cap, dtype=np.uint8)
        batches = np.zeros(cap, dtype=np.uint32)
        st = T.DdminStats()
        conj = np.ascontiguousarray(conjoined, dtype=np.uint8) if conjoined is not None else None
        self._check(lib().demi_ddmin(self._h, C.byref(limits), C.byref(params), conj.ctypes.data if conj is not None else None,
                                     mcs.ctypes.data, consulted.ctypes.data, passed.ctypes.data, cap, batches.ctypes.data, cap,
                                     C.byref(st)))
        return T.mask_to_events(mcs), [(T.mask_to_events(consulted[i]), bool(passed[i])) for i in range(min(cap, st.consultations))], \
            [int(b) for b in batches[:st.launches]], st

    def random_explore_candidates(self, masks, executions, limits, seed_base=0):
        """K1 over candidate subsequences of the loaded trace: (verdicts [n_cand, executions], flags [n_cand])."""
        import numpy as np
        m = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
        v = np.zeros((len(m), executions), dtype=T.VERDICT_DTYPE)
        f = np.zeros(len(m), dtype=np.uint32)
        self._check(lib().demi_random_explore_candidates(self._h, C.c_uint64(seed_base), m.ctypes.data, len(m), executions, C.byref(limits),
                                                         v.ctypes.data, f.ctypes.data))
        return v, f

    def random_ddmin(self, limits, params=None, seed_base=0, conjoined=None, cap=4096):
        """RunnerUtils.randomDDMin on the loaded trace, natively (demi_random_ddmin): (mcs indices, [(candidate indices, passes)]
        in consultation order, candidates per launch, stats)."""
        import numpy as np
        params = params or T.RandomDdminParams()
        mcs = np.zeros(4, dtype=np.uint64)
        consulted = np.zeros((cap, 4), dtype=np.uint64)
        passed = np.zeros(cap, dtype=np.uint8)
        batches = np.zeros(cap, dtype=np.uint32)
        st = T.DdminStats()
        conj = np.ascontiguousarray(conjoined, dtype=np.uint8) if conjoined is not None else None
        self._check(lib().demi_random_ddmin(self._h, C.c_uint64(seed_base), C.byref(limits), C.byref(params),
                                            conj.ctypes.data if conj is not None else None, mcs.ctypes.data, consulted.ctypes.data,
                                            passed.ctypes.data, cap, batches.ctypes.data, cap, C.byref(st)))
        return T.mask_to_events(mcs), [(T.mask_to_events(consulted[i]), bool(passed[i])) for i in range(min(cap, st.consultations))], \
            [int(b) for b in batches[:min(cap, st.launches)]], st

    def replay_get_kept(self, n_rec, skip, limits, mask=None):
        """(Verdict, uint8[n_rec]) of one candidate: which recorded events make up its executed trace."""
        import numpy as np
        mp = None
        if mask is not None:
            mask = np.ascontiguousarray(mask, dtype=np.uint64).reshape(4)
            mp = mask.ctypes.data
        v = T.Verdict()
        kept = np.zeros(max(int(n_rec), 1), dtype=np.uint8)
        self._check(lib().demi_replay_get_kept(self._h, mp, C.c_uint32(int(skip) & 0xFFFFFFFF), C.byref(limits),
                                               C.byref(v), kept.ctypes.data))
        return v, kept[:int(n_rec)]

    def dpor_load(self, externals, areas=None):
        import numpy as np
        ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        if areas is not None:
            self.ext_payload_areas(areas)
        self._check(lib().demi_dpor_load(self._h, ev.ctypes.data if len(ev) else None, len(ev)))

    def dpor_batch(self, prefixes, params, shared=None):
        """prefixes: list of DPOR_TRACE_DTYPE arrays (nextTrace of each interleaving); shared[i]: leading events of
        prefix i whose racing pairs the caller already has (None / 0 = report all).  Returns
        (verdicts, [trace arrays], [pair arrays])."""
        import numpy as np
        n = len(prefixes)
        stride = max([len(p) for p in prefixes] + [1])
        pf = np.zeros((n, stride), dtype=T.DPOR_TRACE_DTYPE)
        pl = np.zeros(n, dtype=np.uint32)
        for i, p in enumerate(prefixes):
            pf[i, :len(p)] = p
            pl[i] = len(p)
        verdicts = np.zeros(n, dtype=T.VERDICT_DTYPE)
        traces = np.zeros((n, T.DPOR_MAX_TRACE), dtype=T.DPOR_TRACE_DTYPE)
        tl = np.zeros(n, dtype=np.uint32)
        pairs = np.zeros((n, max(1, params.max_pairs)), dtype=T.DPOR_PAIR_DTYPE)
        npairs = np.zeros(n, dtype=np.uint32)
        sh = np.ascontiguousarray(shared, dtype=np.uint32) if shared is not None else None
        if n:
            self._check(lib().demi_dpor_batch(self._h, pf.ctypes.data, pl.ctypes.data, sh.ctypes.data if sh is not None else None,
                                              stride, n, C.byref(params),
                                              verdicts.ctypes.data, traces.ctypes.data, tl.ctypes.data,
                                              pairs.ctypes.data, npairs.ctypes.data))
        return verdicts, [traces[i, :tl[i]].copy() for i in range(n)], [pairs[i, :npairs[i]].copy() for i in range(n)]

    def dpor_set_traces(self, original_trace=None, initial_trace=None):
        """ArvindDistanceOrdering.init(sched, originalTrace) and DPORwHeuristics.setInitialTrace for the following dpor_explore
        calls (DPOR_TRACE_DTYPE arrays; None clears)."""
        import numpy as np
        keys = np.ascontiguousarray(np.asarray(original_trace)["key"], dtype=np.uint64) if original_trace is not None else np.zeros(0, dtype=np.uint64)
        init = np.ascontiguousarray(initial_trace, dtype=T.DPOR_TRACE_DTYPE) if initial_trace is not None else np.zeros(0, dtype=T.DPOR_TRACE_DTYPE)
        self._check(lib().demi_dpor_set_traces(self._h, keys.ctypes.data if len(keys) else None, len(keys),
                                               init.ctypes.data if len(init) else None, len(init)))

    def edit_distance_dpor_ddmin(self, externals, initial_trace, params, ip=None, cap=4096):
        """RunnerUtils.editDistanceDporDDMin natively (demi_edit_distance_dpor_ddmin: IncrementalDDMin over ResumableDPOR, every
        consultation K3 launches): (mcs indices, [(subsequence indices, passes, distance cap)] in consultation order,
        [(cap, MCS size)] per pass, the reproducing interleaving or None, stats)."""
        import numpy as np
        ev = np.ascontiguousarray(externals, dtype=T.EXT_EVENT_DTYPE)
        init = np.ascontiguousarray(initial_trace, dtype=T.DPOR_TRACE_DTYPE)
        ip = ip or T.IncDdminParams()
        mcs = np.zeros(4, dtype=np.uint64)
        consulted = np.zeros((cap, 4), dtype=np.uint64)
        passed = np.zeros(cap, dtype=np.uint8)
        dist = np.zeros(cap, dtype=np.uint32)
        vt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
        st = T.IncDdminStats()
        self._check(lib().demi_edit_distance_dpor_ddmin(self._h, ev.ctypes.data if len(ev) else None, len(ev), init.ctypes.data, len(init),
                                                        C.byref(params), C.byref(ip), mcs.ctypes.data, consulted.ctypes.data, passed.ctypes.data,
                                                        dist.ctypes.data, cap, vt.ctypes.data, C.byref(st)))
        n = min(cap, st.consultations)
        return T.mask_to_events(mcs), [(T.mask_to_events(consulted[i]), bool(passed[i]), int(dist[i])) for i in range(n)], \
            [(int(st.pass_distance[i]), int(st.pass_mcs_len[i])) for i in range(min(16, st.passes))], \
            (vt[:st.violation_len].copy() if st.violation_len else None), st

    @staticmethod
    def dpor_buffers(max_interleavings):
        """Output arrays for dpor_explore(..., buffers=...) - what a host that calls demi_dpor_explore repeatedly keeps (a JVM's
        arrays): allocated and touched once, so that a call's time is the library's and not the page faults of fresh arrays."""
        import numpy as np
        cap = int(max_interleavings)
        b = (np.zeros(cap, dtype=T.VERDICT_DTYPE), np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32),
             np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE))
        for a in b:
            a.view(np.uint8)[::4096] = 0          # (np.zeros maps its pages lazily)
        return b

    def dpor_explore(self, params, search, buffers=None):
        """The whole exploration natively: returns (verdicts, prefix_len, rounds, first violating trace, stats).  With `buffers`
        (dpor_buffers) the results are VIEWS of those arrays - valid until the next call that uses them."""
        import numpy as np
        cap = search.max_interleavings
        if buffers is not None:
            verdicts, plen, rounds, vt = buffers
            assert len(verdicts) >= cap and len(plen) >= cap and len(rounds) >= cap
        else:
            verdicts = np.zeros(cap, dtype=T.VERDICT_DTYPE)
            plen = np.zeros(cap, dtype=np.uint32)
            rounds = np.zeros(cap, dtype=np.uint32)
            vt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
        vl = C.c_uint32(0)
        stats = T.DporStats()
        self._check(lib().demi_dpor_explore(self._h, C.byref(params), C.byref(search), verdicts.ctypes.data,
                                            plen.ctypes.data, rounds.ctypes.data, vt.ctypes.data, C.byref(vl), C.byref(stats)))
        n = int(stats.interleavings)
        if buffers is not None:
            return verdicts[:n], plen[:n], rounds[:int(stats.launches)], vt[:vl.value], stats
        return verdicts[:n].copy(), plen[:n].copy(), rounds[:int(stats.launches)].copy(), vt[:vl.value].copy(), stats

    def dpor_explored(self, index):
        """Interleaving `index` of the last dpor_explore: (next trace it was started from, shared length, executed trace)."""
        import numpy as np
        nt = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
        tr = np.zeros(T.DPOR_MAX_TRACE, dtype=T.DPOR_TRACE_DTYPE)
        nl, sl, tl = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(lib().demi_dpor_explored(self._h, C.c_uint64(int(index)), nt.ctypes.data, C.byref(nl), C.byref(sl), tr.ctypes.data, C.byref(tl)))
        return nt[:nl.value].copy(), int(sl.value), tr[:tl.value].copy()

    # ---- multi-GPU (one process and one Context per GPU)
    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the RCCL unique id (128 bytes) every rank passes to comm_create."""
        buf = C.create_string_buffer(128)
        rc = lib().demi_comm_unique_id(buf)
        if rc != 0:
            raise DemiError(rc, "RCCL unavailable")
        return buf.raw

    def comm_create(self, unique_id: bytes, rank: int, world: int):
        """RCCL communicator over xGMI (ncclCommInitRank: collective over the ranks)."""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._check(lib().demi_comm_create(self._h, buf, rank, world))

    def comm_create_host(self, rank: int, world: int, allgather):
        """A communicator whose all-gather the host supplies: allgather(send: bytes) -> bytes of every rank's block in rank
        order (e.g. torch.distributed over gloo).  Blocks are staged through host memory."""
        def thunk(_user, send, recv, nbytes):
            try:
                out = allgather(C.string_at(send, nbytes))
                C.memmove(recv, out, len(out))
                return 0
            except Exception:        # reported as DEMI_ERR_DEVICE by the library
                return 1
        self._allgather_cb = ALLGATHER_FN(thunk)          # keep the callback alive as long as the communicator
        self._check(lib().demi_comm_create_host(self._h, rank, world, self._allgather_cb, None))

    def comm_destroy(self):
        self._check(lib().demi_comm_destroy(self._h))

    def comm_rank(self):
        r, w = C.c_int(0), C.c_int(1)
        self._check(lib().demi_comm_rank(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_allgather_dev(self, d_send_ptr, d_recv_ptr, nbytes, stream=None):
        self._check(lib().demi_comm_allgather_dev(self._h, C.c_void_p(d_send_ptr), C.c_void_p(d_recv_ptr), nbytes, stream))

    def random_explore_sharded(self, n_total, limits, seed_base=0, cap=1 << 16):
        """n_total schedules split by index range over the communicator's ranks; the merged violation set on every rank."""
        import numpy as np
        out = np.zeros(cap, dtype=T.VIOLATION_DTYPE)
        cnt = C.c_uint64(0)
        self._check(lib().demi_random_explore_sharded(self._h, C.c_uint64(seed_base), n_total, C.byref(limits), out.ctypes.data,
                                                      cap, C.byref(cnt)))
        return out[:min(cnt.value, cap)].copy(), int(cnt.value)

    def replay_batch_sharded(self, masks, limits):
        import numpy as np
        masks = np.ascontiguousarray(masks, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros(len(masks), dtype=T.VERDICT_DTYPE)
        if len(masks):
            self._check(lib().demi_replay_batch_sharded(self._h, masks.ctypes.data, len(masks), C.byref(limits), out.ctypes.data))
        return out

    def provenance_prune(self, traces, affected_masks):
        """demi_provenance_prune: traces = sequence of DPOR_TRACE_DTYPE arrays, affected_masks = one actor bitmask per trace;
        returns a list of index arrays (the kept events of every trace)."""
        import numpy as np
        n = len(traces)
        if n == 0:
            return []
        stride = max(1, max(len(t) for t in traces))
        buf = np.zeros((n, stride), dtype=T.DPOR_TRACE_DTYPE)
        lens = np.zeros(n, dtype=np.uint32)
        for i, t in enumerate(traces):
            buf[i, :len(t)] = t
            lens[i] = len(t)
        aff = np.ascontiguousarray(affected_masks, dtype=np.uint32)
        assert len(aff) == n
        words = T.DPOR_MAX_TRACE // 64
        keep = np.zeros((n, words), dtype=np.uint64)
        self._check(lib().demi_provenance_prune(self._h, buf.ctypes.data, lens.ctypes.data, aff.ctypes.data, stride, n, keep.ctypes.data))
        out = []
        for i in range(n):
            bits = np.unpackbits(keep[i].view(np.uint8), bitorder="little")[:int(lens[i])]
            out.append(np.nonzero(bits)[0].astype(np.int64))
        return out

    def device_probe(self, waves_per_simd=1, iters=20000, kind=0):
        """Shader clock under load (GHz) and SIMD cycles per wave64 instruction of the probe's kind: 0 integer VALU,
        1 SALU, 2 VALU / SALU alternating, 3 a divergent `if` (demi_device_probe_mix)."""
        r = T.ProbeResult()
        self._check(lib().demi_device_probe_mix(self._h, waves_per_simd, iters, kind, C.byref(r)))
        return r

    def calib_rw(self, mode, nbytes, repeats=1):
        self._check(lib().demi_calib_rw(self._h, mode, nbytes, repeats))

    def random_get_trace_carried(self, seed, exec_index, limits):
        """(verdict, recorded events, executed index) of execution `exec_index` of the carried-generator instance seeded `seed`."""
        import numpy as np
        rec = np.zeros(T.MAX_REC_EVENTS, dtype=T.REC_EVENT_DTYPE)
        v = T.Verdict()
        n, ran = C.c_uint32(0), C.c_uint32(0)
        self._check(lib().demi_random_get_trace_carried(self._h, C.c_uint64(seed), exec_index, C.byref(limits), C.byref(v),
                                                        rec.ctypes.data, len(rec), C.byref(n), C.byref(ran)))
        return v, rec[:n.value].copy(), ran.value

    def random_get_trace(self, seed, limits):
        import numpy as np
        rec = np.zeros(T.MAX_REC_EVENTS, dtype=T.REC_EVENT_DTYPE)
        v = T.Verdict()
        n = C.c_uint32(0)
        self._check(lib().demi_random_get_trace(self._h, C.c_uint64(seed), C.byref(limits), C.byref(v),
                                                rec.ctypes.data, len(rec), C.byref(n)))
        return v, rec[:n.value].copy()
