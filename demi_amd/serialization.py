"""Flat experiment directories: the GPU path's replacement for the reference's Java
ObjectOutputStream experiment dirs (Serialization.scala:57-74, 122-155, 176-254: event_trace.bin,
mcs.bin, ...).  Everything is a little-endian array of the structs in include/demi_gpu.h plus one
JSON file, so a JVM (or anything else) can read it without this package:

  model.json                the lowered application (demi_model)
  externals.bin             demi_ext_event[]   (original_externals)
  event_trace.bin           demi_rec_event[]   (16-byte records: the recorded violating execution)
  mcs.bin                   uint32[]           (indices of the minimal causal sequence, optional)
  meta.json                 fingerprint code, limits, seed, format version, rec_event_size

Format 2: demi_rec_event is 16 bytes (16-bit payload fields, the payload area's high half-word) and meta.json says so
(rec_event_size).  "format": 1 was written by two generations of this package: rounds 1-2 with the 12-byte record (8-bit
payloads), and round 3 - before the version was bumped - already with the 16-byte record.  A format-1 directory is therefore
read by what its bytes are: rec_event_size from meta.json if present, else whichever of the two record sizes divides the file AND
parses as a recorded execution (kinds and actor ids in range, every MsgEvent preceded by the MsgSend of the same id and
message); when both or neither do, the directory is refused - never guessed.  A file whose size is not a whole number of
records of its format is refused, never truncated.
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


def _plausible_recording(rec) -> bool:
    """Is this array a recorded execution?  Kinds and actor ids in range, ids of MsgSend records distinct, every MsgEvent
    delivers a message a MsgSend with the same id, sender, receiver and type produced earlier (RandomScheduler.scala:319-320,
    460-463: the two records of one Uniq)."""
    if len(rec) == 0:
        return True
    if int(rec["kind"].max()) > T.REC_MSG_EVENT:
        return False
    snd = rec["snd"]
    # (either layout: actors 0..7 with deadLetters 15, or - a table of more than 8 actors - 0..15 with deadLetters 31)
    big = int(rec["rcv"].max()) >= T.MAX_ACTORS or bool((snd == T.DEADLETTERS_BIG).any())
    cap, dl = (T.MAX_ACTORS_BIG, T.DEADLETTERS_BIG) if big else (T.MAX_ACTORS, T.DEADLETTERS)
    if int(rec["rcv"].max()) >= cap or bool(((snd >= cap) & (snd != dl)).any()):
        return False
    sent = {}
    for e in rec:
        k = int(e["kind"])
        if k == T.REC_MSG_SEND:
            if int(e["id"]) in sent:
                return False
            sent[int(e["id"])] = (int(e["snd"]), int(e["rcv"]), int(e["msg_type"]), int(e["p0"]), int(e["p1"]))
        elif k == T.REC_MSG_EVENT:
            if sent.get(int(e["id"])) != (int(e["snd"]), int(e["rcv"]), int(e["msg_type"]), int(e["p0"]), int(e["p1"])):
                return False
    return True


def _format1_record(trace_path: str, declared_size):
    """Which record a "format": 1 event_trace.bin holds (see the module docstring)."""
    by_size = {REC_EVENT_DTYPE_V1.itemsize: REC_EVENT_DTYPE_V1, T.REC_EVENT_DTYPE.itemsize: T.REC_EVENT_DTYPE}
    if declared_size is not None:
        if declared_size not in by_size:
            raise ValueError("event_trace.bin was written with %r-byte records, this build reads 12 or 16" % (declared_size,))
        return by_size[declared_size]
    size = os.path.getsize(trace_path)
    fits = [dt for sz, dt in sorted(by_size.items()) if size % sz == 0]
    good = [dt for dt in fits if _plausible_recording(np.fromfile(trace_path, dtype=dt))]
    if len(good) == 1:
        return good[0]
    raise ValueError("event_trace.bin (%d bytes, format 1 without rec_event_size): %s - refusing to guess" %
                     (size, "neither the 12-byte nor the 16-byte record parses as a recorded execution" if not good else
                      "both the 12-byte and the 16-byte record parse as a recorded execution"))


def load_experiment(path: str):
    """Returns (model, EventTrace, ViolationFingerprint, meta dict, mcs or None)."""
    model = load_model(os.path.join(path, "model.json"))
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    fmt = meta.get("format")
    if fmt not in (1, FORMAT_VERSION):
        raise ValueError("unknown experiment format %r" % fmt)
    trace_path = os.path.join(path, "event_trace.bin")
    if fmt == 1:
        rec_dtype = _format1_record(trace_path, meta.get("rec_event_size"))
    else:
        rec_dtype = T.REC_EVENT_DTYPE
        if meta.get("rec_event_size", rec_dtype.itemsize) != rec_dtype.itemsize:
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
