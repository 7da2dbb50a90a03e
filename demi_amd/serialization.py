"""Flat experiment directories: the GPU path's replacement for the reference's Java
ObjectOutputStream experiment dirs (Serialization.scala:57-74, 122-155, 176-254: event_trace.bin,
mcs.bin, ...).  Everything is a little-endian array of the structs in include/demi_gpu.h plus one
JSON file, so a JVM (or anything else) can read it without this package:

  model.json                the lowered application (demi_model)
  externals.bin             demi_ext_event[]   (original_externals)
  event_trace.bin           demi_rec_event[]   (16-byte records: the recorded violating execution)
  mcs.bin                   uint32[]           (indices of the minimal causal sequence, optional)
  meta.json                 fingerprint code, limits, seed, format version, rec_event_size

Format 2 (round 3 on): demi_rec_event is 16 bytes (16-bit payload fields, a reserved half-word).  Format 1 directories hold the
12-byte record of rounds 1-2 (8-bit payloads); load_experiment reads them with that layout and converts field by field.  A file
whose size is not a whole number of records of its format is refused, never truncated.
"""
import json
import os
from typing import Optional, Sequence

import numpy as np

from . import types as T
from .model import Model, load_model, save_model
from .schedulers import EventTrace, ViolationFingerprint

FORMAT_VERSION = 2
# format 1: kind, snd, rcv, msg_type, p0, p1 (one byte each), flags, ext_idx, id - 12 bytes, no padding
REC_EVENT_DTYPE_V1 = np.dtype([("kind", "u1"), ("snd", "u1"), ("rcv", "u1"), ("msg_type", "u1"), ("p0", "u1"), ("p1", "u1"),
                               ("flags", "u1"), ("ext_idx", "u1"), ("id", "<u4")])
assert REC_EVENT_DTYPE_V1.itemsize == 12


def save_experiment(path: str, model: Model, trace: EventTrace, fingerprint: ViolationFingerprint,
                    limits: Optional[T.Limits] = None, seed: Optional[int] = None, mcs: Optional[Sequence[int]] = None):
    os.makedirs(path, exist_ok=True)
    save_model(model, os.path.join(path, "model.json"))
    np.ascontiguousarray(trace.original_externals, dtype=T.EXT_EVENT_DTYPE).tofile(os.path.join(path, "externals.bin"))
    T.rec_events(trace.events).tofile(os.path.join(path, "event_trace.bin"))
    if mcs is not None:
        np.asarray(mcs, dtype=np.uint32).tofile(os.path.join(path, "mcs.bin"))
    meta = {"format": FORMAT_VERSION, "rec_event_size": T.REC_EVENT_DTYPE.itemsize, "fingerprint": int(fingerprint.code), "match_mask": int(fingerprint.match_mask),
            "seed": seed,
            "limits": None if limits is None else [limits.max_messages, limits.invariant_check_interval, limits.p_max,
                                                   limits.looking_for_valid, limits.looking_for, limits.populate_all]}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)


def load_experiment(path: str):
    """Returns (model, EventTrace, ViolationFingerprint, meta dict, mcs or None)."""
    model = load_model(os.path.join(path, "model.json"))
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    fmt = meta.get("format")
    if fmt not in (1, FORMAT_VERSION):
        raise ValueError("unknown experiment format %r" % fmt)
    rec_dtype = REC_EVENT_DTYPE_V1 if fmt == 1 else T.REC_EVENT_DTYPE
    if fmt != 1 and meta.get("rec_event_size", rec_dtype.itemsize) != rec_dtype.itemsize:
        raise ValueError("event_trace.bin was written with %r-byte records, this build reads %d" % (meta.get("rec_event_size"), rec_dtype.itemsize))
    for name, dt in (("externals.bin", T.EXT_EVENT_DTYPE), ("event_trace.bin", rec_dtype)):
        size = os.path.getsize(os.path.join(path, name))
        if size % dt.itemsize:
            raise ValueError("%s: %d bytes is not a whole number of %d-byte records (format %r)" % (name, size, dt.itemsize, fmt))
    ext = np.fromfile(os.path.join(path, "externals.bin"), dtype=T.EXT_EVENT_DTYPE)
    rec = T.rec_events(np.fromfile(os.path.join(path, "event_trace.bin"), dtype=rec_dtype))
    mcs_path = os.path.join(path, "mcs.bin")
    mcs = np.fromfile(mcs_path, dtype=np.uint32) if os.path.exists(mcs_path) else None
    return model, EventTrace(rec, ext), ViolationFingerprint(meta["fingerprint"], meta["match_mask"]), meta, mcs
