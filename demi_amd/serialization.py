"""Flat experiment directories: the GPU path's replacement for the reference's Java
ObjectOutputStream experiment dirs (Serialization.scala:57-74, 122-155, 176-254: event_trace.bin,
mcs.bin, ...).  Everything is a little-endian array of the structs in include/demi_gpu.h plus one
JSON file, so a JVM (or anything else) can read it without this package:

  model.json                the lowered application (demi_model)
  externals.bin             demi_ext_event[]   (original_externals)
  event_trace.bin           demi_rec_event[]   (16-byte records: the recorded violating execution)
  mcs.bin                   uint32[]           (indices of the minimal causal sequence, optional)
  meta.json                 fingerprint code, limits, seed, format version
"""
import json
import os
from typing import Optional, Sequence

import numpy as np

from . import types as T
from .model import Model, load_model, save_model
from .schedulers import EventTrace, ViolationFingerprint

FORMAT_VERSION = 1


def save_experiment(path: str, model: Model, trace: EventTrace, fingerprint: ViolationFingerprint,
                    limits: Optional[T.Limits] = None, seed: Optional[int] = None, mcs: Optional[Sequence[int]] = None):
    os.makedirs(path, exist_ok=True)
    save_model(model, os.path.join(path, "model.json"))
    np.ascontiguousarray(trace.original_externals, dtype=T.EXT_EVENT_DTYPE).tofile(os.path.join(path, "externals.bin"))
    T.rec_events(trace.events).tofile(os.path.join(path, "event_trace.bin"))
    if mcs is not None:
        np.asarray(mcs, dtype=np.uint32).tofile(os.path.join(path, "mcs.bin"))
    meta = {"format": FORMAT_VERSION, "fingerprint": int(fingerprint.code), "match_mask": int(fingerprint.match_mask),
            "seed": seed,
            "limits": None if limits is None else [limits.max_messages, limits.invariant_check_interval, limits.p_max,
                                                   limits.looking_for_valid, limits.looking_for, limits.populate_all]}
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)


def load_experiment(path: str):
    """Returns (model, EventTrace, ViolationFingerprint, meta dict, mcs or None)."""
    model = load_model(os.path.join(path, "model.json"))
    ext = np.fromfile(os.path.join(path, "externals.bin"), dtype=T.EXT_EVENT_DTYPE)
    rec = np.fromfile(os.path.join(path, "event_trace.bin"), dtype=T.REC_EVENT_DTYPE)
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT_VERSION:
        raise ValueError("unknown experiment format %r" % meta.get("format"))
    mcs_path = os.path.join(path, "mcs.bin")
    mcs = np.fromfile(mcs_path, dtype=np.uint32) if os.path.exists(mcs_path) else None
    return model, EventTrace(rec, ext), ViolationFingerprint(meta["fingerprint"], meta["match_mask"]), meta, mcs
