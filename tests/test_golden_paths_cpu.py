"""CPU suite, part 10: the committed fixtures of the SrcDstFIFO, replay / removal and DPOR paths are what the oracle
computes today (tools/make_golden.py wrote them; they are regression pins of our restatement, not JVM outputs)."""
import hashlib
import json
import os

import numpy as np

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.fuzzer import events_to_array

G = os.path.join(os.path.dirname(__file__), "golden")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_fifo_fixture(oracle):
    model, events, lim = raft5_config2()
    want = np.load(os.path.join(G, "raft5_config2_fifo_verdicts.npy"))
    fifo = T.Limits(lim.max_messages, lim.invariant_check_interval, lim.p_max, 0, 0, 0, T.STRATEGY_SRC_DST_FIFO)
    assert (oracle.random_explore(model, events, len(want), seed_base=SEED_BASE, limits=fifo) == want).all()


def test_replay_fixture(oracle):
    model, events, lim = raft5_config2()
    z = np.load(os.path.join(G, "raft5_config2_replay.npz"))
    vv, rec, _ = oracle.random_execute(model, events, SEED_BASE + int(z["index"]), lim)
    assert vv.fingerprint == int(z["fingerprint"]) and (rec == T.rec_events(z["rec"])).all()
    target = T.Limits(0, 0, 64, 1, int(z["fingerprint"]), 0)
    assert (oracle.sts_replay_batch(model, z["used"], z["rec"], z["masks"], target) == z["mask_verdicts"]).all()
    assert (oracle.sts_removal_batch(model, z["used"], z["rec"], z["skips"], target) == z["skip_verdicts"]).all()
    for k in range(len(z["kept"])):
        assert (oracle.sts_removal_kept(model, z["used"], z["rec"], int(z["skips"][k]), target)[1] == z["kept"][k]).all()


def test_dpor_fixture(oracle):
    z = np.load(os.path.join(G, "raft3_dpor.npz"))
    prefixes = [z["prefixes"][k, :int(n)] for k, n in enumerate(z["prefix_len"])]
    dv, dt, dp = oracle.dpor_batch(M.raft_model(3), z["externals"], prefixes, T.DporParams(30, 0, 0, 0, 64, 4096, 0))
    assert (dv == z["verdicts"]).all()
    assert [len(t) for t in dt] == list(z["trace_len"]) and [len(p) for p in dp] == list(z["n_pairs"])
    assert [_sha(t) for t in dt] == list(z["trace_sha"]) and [_sha(p) for p in dp] == list(z["pairs_sha"])


def test_array_model_fixtures(oracle):
    """DEMI_MODEL_ARRAY: the committed models (rows with LDX / STX, the array length in `flags`) and the oracle's verdicts on
    them, both strategies - the raft with a real log on the bench trace, the replicated log with its hole."""
    import pytest
    for name in ("raft5_log8", "replog4_6", "raft5_log8_fields"):
        model = M.load_model(os.path.join(G, name + "_model.json"))
        assert model.array_len > 0 and model.payloads == (5 if name.endswith("fields") else 2)
        z = np.load(os.path.join(G, name + "_verdicts.npz"))
        mm, ic, pm = (int(x) for x in z["limits"])
        for sname, strat in (("random", T.STRATEGY_FULLY_RANDOM), ("fifo", T.STRATEGY_SRC_DST_FIFO)):
            lim = T.Limits(mm, ic, pm, 0, 0, 0, strat)
            got = oracle.random_explore(model, z["events"], len(z[sname]), seed_base=SEED_BASE, limits=lim, n_threads=os.cpu_count())
            assert (got == z[sname]).all(), (name, sname)
            assert (got["flags"] & T.V_VIOLATION).sum() > 10
    # the constructors still produce the committed tables
    assert M.raft_model(5, log_cap=8).code == M.load_model(os.path.join(G, "raft5_log8_model.json")).code
    assert M.raft_model(5, log_cap=8, real_fields=True).code == M.load_model(os.path.join(G, "raft5_log8_fields_model.json")).code
    assert M.replog_model(4, 6, True, False).code == M.load_model(os.path.join(G, "replog4_6_model.json")).code
