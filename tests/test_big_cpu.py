"""Tables of more than 8 actors (the BIG layout of include/demi_gpu.h) on the CPU: the oracle against its golden record
(tests/golden/big_tables.json, tools/make_golden_big.py), the layout's visible parts - deadLetters = 31 in recorded traces, 16-bit
actor masks in the fingerprints - and the validation rules.  The oracle's BIG layout is held against the literal transliterations
of the Scala schedulers in tests/test_random_scheduler_transliteration_cpu.py and tests/test_dpor_scheduler_transliteration_cpu.py;
the kernels against the oracle in tests/test_big_gpu.py (part of the emulator selection of the CPU suite)."""
import hashlib
import json
import os

import numpy as np

from demi_amd import model as M
from demi_amd import types as T
from demi_amd.apps import SEED_BASE, raft11_config2, shuffle12_config5

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "big_tables.json")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_oracle_gives_the_golden_record(oracle):
    with open(GOLDEN) as f:
        gold = json.load(f)
    assert gold["seed_base"] == SEED_BASE
    m, ev, lim = raft11_config2()
    m2, dev, fev, lim2, par = shuffle12_config5()
    for name, model, events, limits in (("raft11", m, ev, lim), ("shuffle12", m2, fev, lim2)):
        g = gold[name]
        assert model.name == g["model"] and model.n_actors == g["n_actors"] > T.MAX_ACTORS and model.wide
        v = oracle.random_explore(model, events, g["fuzz_prefix"], seed_base=SEED_BASE, limits=limits, n_threads=os.cpu_count())
        assert _sha(v) == g["sha256_fuzz_verdicts"]
        viol = v[(v["flags"] & T.V_VIOLATION) != 0]
        assert len(viol) == g["violating_executions"] and ((viol["fingerprint"] >> 30) == g["fingerprint_kind"]).all()
        # the fingerprint's low 16 bits are actors of THIS table
        assert ((viol["fingerprint"] & 0xFFFF) < (1 << model.n_actors)).all() and ((viol["fingerprint"] & 0xFFFF) != 0).all()
    d = gold["shuffle12"]["dpor_rounds_batch_4096"]
    r = oracle.dpor_explore(m2, dev, par, T.DporSearch(4096, 1 << 17, 0, 1, T.DPOR_ORDER_ROUNDS), os.cpu_count())
    assert int(r[4].exhausted) == 1 and len(r[0]) == d["interleavings"] and int(r[4].violations) == d["violating"]
    assert _sha(r[0]) == d["sha256_verdicts"] and _sha(r[1]) == d["sha256_prefix_lengths"] and int(r[4].first_violation) == d["first_violation"]


def test_recorded_traces_name_deadletters_31_and_replay(oracle):
    """A recorded execution of a big table: externals and timers come from sender 31; the STSScheduler replay of the whole
    trace reproduces the violation, a candidate without the Bootstraps does not."""
    m, ev, lim = raft11_config2()
    l0 = T.Limits(lim.max_messages, 0, lim.p_max, 0, 0, 0)
    v = oracle.random_explore(m, ev, 3000, seed_base=SEED_BASE, limits=l0, n_threads=os.cpu_count())
    idx = int(np.nonzero((v["flags"] & T.V_VIOLATION) != 0)[0][0])
    vd, rec, _st = oracle.random_execute(m, ev, SEED_BASE + idx, l0)
    ev_rows = rec[rec["kind"] == T.REC_MSG_EVENT]
    assert int(ev_rows["snd"].max()) == T.DEADLETTERS_BIG and ((ev_rows["snd"] < m.n_actors) | (ev_rows["snd"] == T.DEADLETTERS_BIG)).all()
    assert int(ev_rows["rcv"].max()) >= T.MAX_ACTORS
    lr = T.Limits(lim.max_messages, 0, lim.p_max, 1, int(vd.fingerprint), 0)
    masks = np.full((2, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    masks[1, 0] &= ~np.uint64(((1 << 11) - 1) << 11)           # the eleven Bootstrap Sends removed
    r = oracle.sts_replay_batch(m, ev, rec, masks, lr, n_threads=1)
    assert r[0]["flags"] & T.V_VIOLATION and int(r[0]["hash"]) == int(vd.hash) and not (r[1]["flags"] & T.V_VIOLATION)


def test_validation_of_big_tables(oracle):
    m = M.raft_model(9)
    assert m.wide and oracle.model_validate(m)[0] == 0
    m.wide = False
    m.init_state = m.init_state[::2]
    assert "DEMI_MODEL_WIDE" in oracle.model_validate(m)[1]
    m = M.raft_model(16)
    assert oracle.model_validate(m)[0] == 0
    m.n_actors = 17
    assert "n_actors" in oracle.model_validate(m)[1]
    # the 8-actor tables are what they were: narrow, deadLetters 15
    m8 = M.shuffle_model()
    assert m8.n_actors == 8 and not m8.wide and m8.name == "shuffle8-synth"


def test_oracle_gives_the_transliterations_record(oracle):
    """tests/golden/big_tables_transliteration.json (tools/check_big_transliteration.py): ScalaRandomScheduler over the first 4 096
    schedules of both fuzz steps, ScalaDPORwHeuristics over the 12-actor job in the reference's own order until its queue was
    empty (28 767 interleavings, 16 minutes of Python).  The C oracle - its batch, and the product's one-at-a-time loop around its
    interleavings - gives those bytes."""
    with open(os.path.join(os.path.dirname(GOLDEN), "big_tables_transliteration.json")) as f:
        tl = json.load(f)
    m, ev, lim = raft11_config2()
    m2, dev, fev, lim2, par = shuffle12_config5()
    for name, model, events, limits in (("raft11", m, ev, lim), ("shuffle12", m2, fev, lim2)):
        v = oracle.random_explore(model, events, tl[name]["schedules"], seed_base=SEED_BASE, limits=limits, n_threads=os.cpu_count())
        assert _sha(v) == tl[name]["sha256_verdicts"] and tl[name]["equals_the_oracle"] is True
    r = tl["shuffle12_dpor_reference_order"]
    one = oracle.dpor_explore(m2, dev, par, T.DporSearch(1, r["interleavings"] + 64, 0, 1, T.DPOR_ORDER_ROUNDS), 1)
    assert int(one[4].exhausted) == 1 and len(one[0]) == r["interleavings"] == 28767 and int(one[4].violations) == r["violations"]
    assert _sha(one[0]) == r["sha256_verdicts"] and _sha(np.ascontiguousarray(one[1], dtype=np.uint32)) == r["sha256_prefix_lens"]
    assert int(one[4].first_violation) == r["first_violation"]
