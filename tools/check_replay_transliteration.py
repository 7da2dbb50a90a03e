#!/usr/bin/env python
"""The bench's replay workload (BASELINE config 4: the 200-event failing raft5 execution, candidate subsequences drawn with
numpy's default_rng(0) at 0.7 per event - bench.py bench_ddmin) replayed by the literal Python transliteration of the Scala
STSScheduler (tests/test_sts_scheduler_transliteration_cpu.py ScalaSTSScheduler: its own pendingEvents, expected-event walk and
injector state; only the actors' row interpreter is shared) for the first N candidates: verdict for verdict against the C oracle,
and the SHA-256 of the N verdicts into tests/golden/replay_config4_transliteration.json - which the CPU suite holds the oracle
against and the GPU suite the device's verdicts of the same candidates.  Under a millisecond per candidate and core on average (most candidates diverge early); N = 2^20 - the bench's whole
workload - by default; the record also holds the SHA-256 of the first 2^16, which is what the CPU suite re-computes.
Usage: python tools/check_replay_transliteration.py [N [processes]]"""
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from demi_amd import types as T  # noqa: E402
from demi_amd.apps import SEED_BASE, raft5_config4  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tests.test_sts_scheduler_transliteration_cpu import ScalaSTSScheduler  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
procs = int(sys.argv[2]) if len(sys.argv) > 2 else (os.cpu_count() or 1)


def workload(n):
    """bench.py bench_ddmin's inputs, from the oracle: the first violating execution among 4000, recorded; the candidate masks"""
    model, events, lim = raft5_config4()
    v = O.random_explore(model, events, 4000, seed_base=SEED_BASE, limits=lim)
    i = int(np.nonzero(v["flags"] & T.V_VIOLATION)[0][0])
    vv, rec, _ = O.random_execute(model, events, SEED_BASE + i, lim)
    used = events[:T.verdict_trace_idx(vv.flags)]
    keep = np.random.default_rng(0).random((n, len(used))) < 0.7          # (row-major: the first n rows of bench.py's 2^20 x 200 draw)
    masks = np.zeros((n, 4), dtype=np.uint64)
    for w in range(4):
        bits = keep[:, 64 * w:64 * (w + 1)]
        masks[:, w] = (bits.astype(np.uint64) << np.arange(bits.shape[1], dtype=np.uint64)).sum(axis=1)
    return model, used, rec, vv, keep, masks


model, used, rec, vv, keep, masks = workload(n)


def chunk(lo_hi):
    lo, hi = lo_hi
    out = np.zeros(hi - lo, dtype=T.VERDICT_DTYPE)
    for i in range(lo, hi):
        subseq = [int(j) for j in np.nonzero(keep[i])[0] if int(used[j]["kind"]) != T.EV_WAIT_QUIESCENCE]
        out[i - lo] = ScalaSTSScheduler(O, model, used, rec, subseq, 0).test(int(vv.fingerprint), model.fp_match_mask)
    return lo, out


if __name__ == "__main__":
    import multiprocessing as mp
    want = O.sts_replay_batch(model, used, rec, masks, T.Limits(0, 0, 128, 1, vv.fingerprint, 0), n_threads=os.cpu_count() or 1)
    # (a capacity flag of the restatement - more pending messages than its largest set - is not a behaviour of the reference: such
    # a candidate keeps the oracle's verdict and is counted)
    capped = (want["flags"] & (T.V_PENDING_OVF | T.V_QUEUE_OVF)) != 0
    got = np.zeros(n, dtype=T.VERDICT_DTYPE)
    t0 = time.perf_counter()
    with mp.Pool(procs) as pool:
        for lo, out in pool.imap_unordered(chunk, [(lo, min(n, lo + 512)) for lo in range(0, n, 512)]):
            got[lo:lo + len(out)] = out
    seconds = time.perf_counter() - t0
    got[capped] = want[capped]
    same = bool((got == want).all())
    rec_out = {"generator": "tools/check_replay_transliteration.py %d (ScalaSTSScheduler, %d processes, %.0f s)" % (n, procs, seconds),
               "candidates": n, "skipped_for_a_capacity_flag": int(capped.sum()), "externals": int(len(used)), "recorded_events": int(len(rec)),
               "still_violating": int(((got["flags"] & T.V_VIOLATION) != 0).sum()), "diverged": int(((got["flags"] & T.V_DIVERGED) != 0).sum()),
               "sha256_verdicts": hashlib.sha256(got[:1 << 16].tobytes()).hexdigest(), "sha256_masks": hashlib.sha256(masks[:1 << 16].tobytes()).hexdigest(),
               "first": min(n, 1 << 16),
               "still_violating_of_the_first": int(((got[:1 << 16]["flags"] & T.V_VIOLATION) != 0).sum()),
               "sha256_verdicts_of_all": hashlib.sha256(got.tobytes()).hexdigest(), "sha256_masks_of_all": hashlib.sha256(masks.tobytes()).hexdigest(),
               "equals_the_oracle": same}
    print(rec_out)
    if not same:
        bad = np.nonzero(got != want)[0]
        print("first differing candidate:", int(bad[0]), got[bad[0]], want[bad[0]])
        sys.exit(1)
    with open(os.path.join(ROOT, "tests", "golden", "replay_config4_transliteration.json"), "w") as f:
        json.dump(rec_out, f, indent=1)
