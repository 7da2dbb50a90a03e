#!/usr/bin/env python
"""Host-time split of the native DPOR loop (run / absorb / get_next) on BASELINE config 3, with the CPU test
harness standing in for the K3 kernel.  Diagnostic for dpor_host.hpp; needs no GPU."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import raft5_config3  # noqa: E402
from tests.test_dpor_cpu import native_explore  # noqa: E402

model, ev, depth = raft5_config3()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
budget = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
par = T.DporParams(depth, 0, 0, 0, 64, 4096)
t = time.perf_counter()
v, plen, rounds, vt, stats, secs = native_explore(model, ev, par, batch, budget, n_threads=os.cpu_count())
dt = time.perf_counter() - t
print("interleavings %d launches %d exhausted %d total %.2fs run %.2fs absorb %.2fs get_next %.2fs" %
      (stats.interleavings, stats.launches, stats.exhausted, dt, secs[0], secs[1], secs[2]))
