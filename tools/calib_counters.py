#!/usr/bin/env python
"""Known-byte-count kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (run under rocprofv3 --pmc by
tools/profile_r5.sh): four dispatches of demi::k_calib_rw in this order - write 4 B/lane, read 4 B/lane, write 16 B/lane,
read 16 B/lane - over BYTES each (1 GiB: four times the Infinity Cache, so reads come from HBM)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import _native  # noqa: E402

BYTES = 1 << 30
ctx = _native.Context(0)
for mode in (0, 1, 2, 3):
    ctx.calib_rw(mode, BYTES, 1)
ctx.close()
print("calib bytes", BYTES)
