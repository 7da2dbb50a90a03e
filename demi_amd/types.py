"""ctypes mirrors of include/demi_gpu.h (flat data formats crossing the C ABI)."""
import ctypes as C

MAX_ACTORS = 8
DEADLETTERS = 15
MAX_ACTORS_BIG = 16       # a table with more than MAX_ACTORS actors: the BIG layout (include/demi_gpu.h)
DEADLETTERS_BIG = 31


def deadletters_of(n_actors):
    """The sender id of externals and timers in the layout of a table with n_actors actors."""
    return DEADLETTERS_BIG if n_actors > MAX_ACTORS else DEADLETTERS


def fingerprint_actors(code):
    """The actors a ViolationFingerprint word names: its low 8 bits, or - the BIG layout, kind in bits 30..31 - its low 16."""
    n = MAX_ACTORS_BIG if (int(code) >> 30) else MAX_ACTORS
    return [i for i in range(n) if (int(code) >> i) & 1]
MAX_MSG_TYPES = 32
MAX_CLASSES = 4
MAX_CODE = 1024
MAX_TIMER_TYPES = 4
MAX_EXT_EVENTS = 255
MAX_REC_EVENTS = 16384
ABI_VERSION = 4             # DEMI_ABI_VERSION of include/demi_gpu.h (struct-layout generation)
MAX_PENDING = 128          # DEMI_MAX_PENDING

# demi_status
OK = 0
ERR_INVALID_ARG = -1
ERR_INVALID_MODEL = -2
ERR_INVALID_TRACE = -3
ERR_NO_MODEL = -4
ERR_NO_TRACE = -5
ERR_DEVICE = -6
ERR_CAPACITY = -7

# demi_ext_kind  (ExternalEvents.scala:62-91)
EV_START, EV_KILL, EV_SEND, EV_PARTITION, EV_UNPARTITION, EV_WAIT_QUIESCENCE = range(6)
EV_NAMES = ["Start", "Kill", "Send", "Partition", "UnPartition", "WaitQuiescence"]

# demi_msg_class
MSG_INTERNAL, MSG_EXTERNAL, MSG_TIMER = 0, 1, 2

# demi_strategy
STRATEGY_FULLY_RANDOM, STRATEGY_SRC_DST_FIFO = 0, 1

# demi_inv_kind
INV_NONE, INV_AT_MOST_ONE, INV_NEVER, INV_AGREE = 0, 1, 2, 3
INV_PROGRAM = 0x100          # OR-ed into inv_kind: the per-actor predicate / key is a row program starting at row inv_fa

# verdict flags
V_VIOLATION = 0x1
V_MAXMSG = 0x2
V_PENDING_OVF = 0x4
V_QUEUE_OVF = 0x8
V_DIVERGED = 0x10
V_TRACE_OVF = 0x20
V_PAIRS_OVF = 0x40
V_SELFMSG = 0x80
DPOR_MAX_TRACE = 256
DPOR_ROOT_KEY = 0xCBF29CE484222325
DPOR_PRIME = 0x100000001B3


def dpor_marker_key(ext_idx):
    return DPOR_ROOT_KEY ^ (0x5155494553434500 | ext_idx)

# demi_rec_kind
(REC_SPAWN, REC_KILL, REC_PARTITION, REC_UNPARTITION, REC_BEGIN_WAIT_QUIESCENCE, REC_QUIESCENCE,
 REC_MSG_SEND, REC_MSG_EVENT) = range(8)


class ExtEvent(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("a", C.c_uint8), ("b", C.c_uint8), ("msg_type", C.c_uint8),
                ("p0", C.c_uint8), ("p1", C.c_uint8), ("p0_hi", C.c_uint8), ("p1_hi", C.c_uint8)]


class ModelStruct(C.Structure):
    _fields_ = [("n_actors", C.c_uint32), ("n_msg_types", C.c_uint32), ("n_classes", C.c_uint32),
                ("code_len", C.c_uint32),
                ("msg_class", C.POINTER(C.c_uint8)), ("actor_class", C.POINTER(C.c_uint8)),
                ("handler_start", C.POINTER(C.c_uint16)), ("code", C.POINTER(C.c_uint32)),
                ("init_state", C.POINTER(C.c_uint64)),
                ("inv_kind", C.c_uint32), ("inv_fa", C.c_uint32), ("inv_va", C.c_uint32),
                ("inv_fb", C.c_uint32), ("fp_match_mask", C.c_uint32), ("flags", C.c_uint32)]


MODEL_WIDE = 0x1            # demi_model.flags: 16 x u16 register window (include/demi_gpu.h, DEMI_MODEL_WIDE)
MAX_ARRAY = 64              # DEMI_MAX_ARRAY


def MODEL_ARRAY(n):
    """demi_model.flags: every actor owns an array of n elements (DEMI_MODEL_ARRAY, rows LDX / STX)."""
    return (int(n) & 0xFF) << 8


MAX_PAYLOADS = 6            # DEMI_MAX_PAYLOADS


def MODEL_PAYLOADS(n):
    """demi_model.flags: a message carries n = 3..6 payload fields (DEMI_MODEL_PAYLOADS; wide tables only); 2 = the default."""
    assert n == 2 or 3 <= n <= MAX_PAYLOADS
    return 0 if n == 2 else int(n) << 16


def payload_bits(n):
    """DEMI_PAYLOAD_BITS: the width of one payload field of a wide table whose messages carry n fields."""
    return 16 if n <= 3 else 48 // n


def payload_area(fields, n):
    """The 48-bit payload area (demi_rec_event p0 | p1 << 16 | p_hi << 32) of a wide table's message with these field values."""
    w = payload_bits(n)
    return sum((int(v) & ((1 << w) - 1)) << (k * w) for k, v in enumerate(list(fields)[:n]))


def payload_fields(area, n):
    """The n payload fields of a 48-bit payload area (DEMI_PAYLOAD_OF)."""
    w = payload_bits(n)
    return [(int(area) >> (k * w)) & ((1 << w) - 1) for k in range(n)]


def pay_area(fields, n):
    """The 48-bit payload area of a message with n payload fields (3..6: DEMI_MODEL_PAYLOADS; include/demi_gpu.h), P0 first."""
    w = payload_bits(n)
    return sum((int(v) & ((1 << w) - 1)) << (k * w) for k, v in enumerate(list(fields)[:n]))


def rec_area(e):
    """DEMI_REC_AREA of one record (numpy record or ctypes RecEvent)."""
    return int(e["p0"]) | (int(e["p1"]) << 16) | (int(e["p_hi"]) << 32)



FILTER_ABSENTS_OFF, FILTER_ABSENTS_LITERAL, FILTER_ABSENTS_CORRECTED = 0, 1, 2     # demi_filter_absents


class Limits(C.Structure):
    _fields_ = [("max_messages", C.c_uint32), ("invariant_check_interval", C.c_uint32),
                ("p_max", C.c_uint32), ("looking_for_valid", C.c_uint32), ("looking_for", C.c_uint32),
                ("populate_all", C.c_uint32), ("strategy", C.c_uint32),
                ("filter_known_absents", C.c_uint32), ("executions_per_instance", C.c_uint32)]


class Verdict(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("fingerprint", C.c_uint32), ("hash", C.c_uint64)]


class RecEvent(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("snd", C.c_uint8), ("rcv", C.c_uint8), ("msg_type", C.c_uint8),
                ("p0", C.c_uint16), ("p1", C.c_uint16), ("flags", C.c_uint8), ("ext_idx", C.c_uint8),
                ("p_hi", C.c_uint16), ("id", C.c_uint32)]


class DdminParams(C.Structure):
    """demi_ddmin_params"""
    _fields_ = [("depth", C.c_uint32), ("max_candidates", C.c_uint32), ("check_unmodified", C.c_uint32), ("verify_mcs", C.c_uint32)]

    def __init__(self, depth=0, max_candidates=0, check_unmodified=1, verify_mcs=1):
        super().__init__(depth, max_candidates, check_unmodified, verify_mcs)


class RandomDdminParams(C.Structure):
    """demi_random_ddmin_params"""
    _fields_ = [("executions", C.c_uint32), ("depth", C.c_uint32), ("max_candidates", C.c_uint32), ("check_unmodified", C.c_uint32),
                ("verify_mcs", C.c_uint32), ("sequential", C.c_uint32), ("reserved", C.c_uint32 * 2)]

    def __init__(self, executions=100, depth=0, max_candidates=256, check_unmodified=0, verify_mcs=1, sequential=0):
        super().__init__(executions, depth, max_candidates, check_unmodified, verify_mcs, sequential)


class DdminStats(C.Structure):
    """demi_ddmin_stats"""
    _fields_ = [("consultations", C.c_uint32), ("launches", C.c_uint32), ("mcs_len", C.c_uint32), ("verified", C.c_uint32),
                ("replays", C.c_uint64)]


class IncDdminParams(C.Structure):
    """demi_incddmin_params"""
    _fields_ = [("max_max_distance", C.c_uint32), ("stop_at_size", C.c_uint32), ("check_unmodified", C.c_uint32),
                ("ignore_quiescence", C.c_uint32), ("verify_mcs", C.c_uint32), ("batch", C.c_uint32), ("budget", C.c_uint32),
                ("reserved", C.c_uint32)]

    def __init__(self, max_max_distance=256, stop_at_size=1, check_unmodified=0, ignore_quiescence=1, verify_mcs=1, batch=256, budget=1 << 16):
        super().__init__(max_max_distance, stop_at_size, check_unmodified, ignore_quiescence, verify_mcs, batch, budget, 0)


class IncDdminStats(C.Structure):
    """demi_incddmin_stats"""
    _fields_ = [("replays", C.c_uint64), ("interleavings", C.c_uint64), ("consultations", C.c_uint32), ("instances", C.c_uint32),
                ("passes", C.c_uint32), ("mcs_len", C.c_uint32), ("verified", C.c_int32), ("violation_len", C.c_uint32),
                ("pass_distance", C.c_uint32 * 16), ("pass_mcs_len", C.c_uint32 * 16)]


class DporParams(C.Structure):
    _fields_ = [("depth_bound", C.c_uint32), ("max_messages", C.c_uint32), ("looking_for_valid", C.c_uint32),
                ("looking_for", C.c_uint32), ("p_max", C.c_uint32), ("max_pairs", C.c_uint32),
                ("prioritize_pending", C.c_uint32)]


class DporSearch(C.Structure):
    _fields_ = [("batch", C.c_uint32), ("max_interleavings", C.c_uint32), ("stop_if_violation", C.c_uint32),
                ("track_history", C.c_uint32), ("order", C.c_uint32), ("cache_mb", C.c_uint32),
                ("ordering", C.c_uint32), ("max_distance_plus1", C.c_uint32), ("resume", C.c_uint32)]

    def __init__(self, batch=1, max_interleavings=1, stop_if_violation=0, track_history=1, order=0, cache_mb=0, ordering=0,
                 max_distance=None, resume=0):
        """ordering: DPOR_ORDERING_*; max_distance: None = no cap, k = setMaxDistance(k); resume: continue from the queue the
        previous ordered exploration of the context left"""
        super().__init__(batch, max_interleavings, stop_if_violation, track_history, order, cache_mb, ordering,
                         0 if max_distance is None else int(max_distance) + 1, resume)


class DporStats(C.Structure):
    _fields_ = [("interleavings", C.c_uint64), ("launches", C.c_uint64), ("violations", C.c_uint64),
                ("first_violation", C.c_uint64), ("queue_len", C.c_uint64), ("exhausted", C.c_uint32),
                ("fetches", C.c_uint32), ("executed", C.c_uint64), ("cache_misses", C.c_uint64), ("kernel_ms", C.c_double),
                ("h2d_bytes", C.c_uint64), ("d2h_bytes", C.c_uint64), ("backtrack_points", C.c_uint64)]


class ProbeResult(C.Structure):
    _fields_ = [("shader_clock_ghz", C.c_double), ("cycles_per_valu", C.c_double), ("waves_per_simd", C.c_uint32),
                ("num_cu", C.c_uint32)]


DPOR_ORDERING_DEFAULT, DPOR_ORDERING_ARVIND = 0, 1      # demi_dpor_ordering (BacktrackOrdering.scala)
DPOR_ORDER_ROUNDS = 0       # demi_dpor_order
DPOR_ORDER_REFERENCE = 1


assert C.sizeof(ExtEvent) == 8 and C.sizeof(Verdict) == 16 and C.sizeof(RecEvent) == 16

import numpy as np  # noqa: E402

VERDICT_DTYPE = np.dtype([("flags", "<u4"), ("fingerprint", "<u4"), ("hash", "<u8")])
EXT_EVENT_DTYPE = np.dtype([("kind", "u1"), ("a", "u1"), ("b", "u1"), ("msg_type", "u1"),
                            ("p0", "u1"), ("p1", "u1"), ("p0_hi", "u1"), ("p1_hi", "u1")])
REC_EVENT_DTYPE = np.dtype([("kind", "u1"), ("snd", "u1"), ("rcv", "u1"), ("msg_type", "u1"), ("p0", "<u2"), ("p1", "<u2"),
                            ("flags", "u1"), ("ext_idx", "u1"), ("p_hi", "<u2"), ("id", "<u4")])


def rec_events(a):
    """A recorded trace as a contiguous demi_rec_event[] (16-byte records).  Accepts any structured array with the record's
    field names - e.g. the 12-byte layout (8-bit payloads) that fixtures written before the wide models hold - and copies field
    by field."""
    a = np.asarray(a)
    if a.dtype == REC_EVENT_DTYPE:
        return np.ascontiguousarray(a)
    out = np.zeros(a.shape, dtype=REC_EVENT_DTYPE)
    for name in REC_EVENT_DTYPE.names:
        if a.dtype.names and name in a.dtype.names:
            out[name] = a[name]
    if a.dtype.names and "reserved" in a.dtype.names:      # (records written before the field had a meaning: always 0)
        out["p_hi"] = a["reserved"]
    return out
DPOR_TRACE_DTYPE = np.dtype([("key", "<u8"), ("word", "<u4"), ("parent", "u1"), ("qperiod", "u1"), ("depth", "u1"),
                             ("kind", "u1")])
DPOR_PAIR_DTYPE = np.dtype([("branch", "u1"), ("later", "u1"), ("earlier", "u1"), ("pad", "u1")])
assert DPOR_TRACE_DTYPE.itemsize == 16 and DPOR_PAIR_DTYPE.itemsize == 4
VIOLATION_DTYPE = np.dtype([("index", "<u8"), ("fingerprint", "<u4"), ("flags", "<u4")])
assert VERDICT_DTYPE.itemsize == 16 and VIOLATION_DTYPE.itemsize == 16 and EXT_EVENT_DTYPE.itemsize == 8 and REC_EVENT_DTYPE.itemsize == 16


def verdict_violation(flags):
    return (flags & V_VIOLATION) != 0


def verdict_deliveries(flags):
    return (flags >> 16) & 0xFFFF


def verdict_trace_idx(flags):
    return (flags >> 8) & 0xFF


def mask_to_events(mask) -> tuple:
    """The external-event indices of a 256-bit candidate mask (uint64[4])."""
    bits = np.unpackbits(np.ascontiguousarray(mask, dtype=np.uint64).view(np.uint8), bitorder="little")
    return tuple(int(i) for i in np.flatnonzero(bits))
