"""CPU suite, part 2: host logic — the C-ABI library loads and exports every symbol declared in
include/demi_gpu.h (no compute calls without a GPU), the scheduler mirror's error behaviour, the
model assembler, the Fuzzer-distribution trace generator, and the N>1 sharding / violation-set
all-gather over gloo (world_size 2)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from demi_amd import types as T
from demi_amd import model as M
from demi_amd.distributed import merge_violation_sets, pack_violation_list, shard_range
from demi_amd.fuzzer import FuzzerWeights, events_to_array, raft_trace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from demi_amd import _native
    L = _native.lib()
    header = open(os.path.join(ROOT, "include", "demi_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(demi_[a-z0-9_]+)\s*\(", header)) - {"demi_status"})
    assert len(declared) >= 10
    for name in declared:
        assert hasattr(L, name), "libdemi_gpu.so does not export %s" % name
    assert sorted(_native.EXPORTS) == declared
    assert b"gfx950" in L.demi_version()
    # the shared object carries gfx950 code objects, and no CPU execution path
    blob = open(_native.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"k1_random_explore" in blob


def test_struct_layouts_match_the_header():
    assert C.sizeof(T.ExtEvent) == 8 and C.sizeof(T.Verdict) == 16 and C.sizeof(T.RecEvent) == 16 and T.REC_EVENT_DTYPE.itemsize == 16
    assert C.sizeof(T.Limits) == 36 and T.Limits.strategy.offset == 24 and T.Limits.filter_known_absents.offset == 28 and T.Limits.executions_per_instance.offset == 32
    assert C.sizeof(T.DporParams) == 28 and T.DporParams.prioritize_pending.offset == 24
    assert T.ModelStruct.msg_class.offset == 16 and T.ModelStruct.inv_kind.offset == 56 and C.sizeof(T.ModelStruct) == 80


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="only meaningful without a GPU")
def test_product_path_fails_loudly_without_a_gpu():
    from demi_amd import _native
    from demi_amd.schedulers import RandomScheduler, SchedulerConfig
    with pytest.raises(_native.DemiError):
        _native.Context(0)
    with pytest.raises(_native.DemiError):
        RandomScheduler(SchedulerConfig(model=M.raft_model(3)))


def test_assembler_labels_and_rows():
    a = M.Asm().eq(M.T0, M.F[0], 3).skipz(M.T0, "end").add(M.F[1], M.F[1], 1).label("end")
    rows = a.finish()
    assert len(rows) == 4 and rows[-1] & 0xFF == 0
    assert rows[1] >> 24 == 1 and rows[1] & 0xFF == M.OPS["SKIPZ"]
    assert rows[0] == M.row(M.OPS["EQ"], 8, 0, 1, 0, 3)
    with pytest.raises(AssertionError):
        M.Asm().skipz(M.T0, "back").label("x").finish() if False else M.Asm().label("b").skipz(M.T0, "b").finish()
    m = M.raft_model(5)
    assert m.n_msg_types == 8 and len(m.handler_start) == 8 and all(h != 0xFFFF for h in m.handler_start)
    assert M.Model.from_json(m.to_json()).to_json() == m.to_json()


def test_fuzzer_distribution_and_shape():
    tr = raft_trace(5, 50, 0xDE31)
    assert len(tr) == 50 and tr[-1][0] == T.EV_WAIT_QUIESCENCE
    assert [e[0] for e in tr[:5]] == [T.EV_START] * 5
    # never two WaitQuiescence in a row (Fuzzer.scala:126-146)
    assert all(not (a[0] == b[0] == T.EV_WAIT_QUIESCENCE) for a, b in zip(tr, tr[1:]))
    # an UnPartition always undoes an earlier Partition of the same pair
    open_pairs = set()
    for e in tr:
        if e[0] == T.EV_PARTITION:
            assert (e[1], e[2]) not in open_pairs
            open_pairs.add((e[1], e[2]))
        if e[0] == T.EV_UNPARTITION:
            open_pairs.remove((e[1], e[2]))
    # deterministic in the seed, different across seeds, weights respected in the large
    assert raft_trace(5, 50, 0xDE31) == tr and raft_trace(5, 50, 0xDE32) != tr
    big = raft_trace(5, 250, 5, FuzzerWeights(kill=0.0, send=0.5, wait_quiescence=0.5, partition=0.0, unpartition=0.0))
    kinds = [e[0] for e in big[10:]]
    assert set(kinds) <= {T.EV_SEND, T.EV_WAIT_QUIESCENCE}


def test_shard_range_and_merge():
    for n in (0, 1, 7, 1000, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))
    a = np.zeros(3, dtype=T.VIOLATION_DTYPE)
    a["index"] = [5, 1, 9]
    a["fingerprint"] = [7, 8, 9]
    b = np.zeros(1, dtype=T.VIOLATION_DTYPE)
    b["index"] = [3]
    merged = merge_violation_sets([pack_violation_list(a, 8), pack_violation_list(b, 8), pack_violation_list(b[:0], 8)], 8)
    assert list(merged["index"]) == [1, 3, 5, 9] and list(merged["fingerprint"]) == [8, 0, 7, 9]
    # truncated list: count exceeds cap
    assert len(merge_violation_sets([pack_violation_list(a, 2)], 2)) == 2


_GLOO_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from demi_amd import types as T
from demi_amd.apps import SEED_BASE, raft5_config2
from demi_amd.distributed import allgather_violation_sets, pack_violation_list, shard_range
from oracle import oracle_py as O          # tests may use the oracle as the per-rank "device"
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
model, events, lim = raft5_config2()
N, CAP = 6000, 512
lo, hi = shard_range(N, rank, world)
v = O.random_explore(model, events, hi - lo, seed_base=SEED_BASE + lo, limits=lim)
hit = np.nonzero(v["flags"] & T.V_VIOLATION)[0]
local = np.zeros(len(hit), dtype=T.VIOLATION_DTYPE)
local["index"] = hit + lo; local["fingerprint"] = v["fingerprint"][hit]; local["flags"] = v["flags"][hit]
merged = allgather_violation_sets(torch.from_numpy(pack_violation_list(local, CAP)), CAP)
full = O.random_explore(model, events, N, seed_base=SEED_BASE, limits=lim)
fh = np.nonzero(full["flags"] & T.V_VIOLATION)[0]
assert len(merged) == len(fh) > 0 and (merged["index"] == fh).all() and (merged["fingerprint"] == full["fingerprint"][fh]).all()
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", len(merged))
'''


def test_sharded_violation_set_allgather_gloo_world2(tmp_path, oracle):
    """N>1 path: index range sharded across ranks, one fixed-size all-gather of the per-rank
    violation lists; the merged set equals the single-process set."""
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
        assert "ok" in o


def test_experiment_dir_roundtrip(tmp_path, oracle):
    from demi_amd.apps import SEED_BASE, raft3_config1
    from demi_amd.schedulers import EventTrace, ViolationFingerprint
    from demi_amd.serialization import load_experiment, save_experiment
    model, events, lim = raft3_config1()
    v, rec, _ = oracle.random_execute(model, events, SEED_BASE + 1, lim)
    tr = EventTrace(rec, events[:T.verdict_trace_idx(v.flags)])
    save_experiment(str(tmp_path / "e"), model, tr, ViolationFingerprint(0x1000103), limits=lim, seed=SEED_BASE + 1, mcs=[0, 2, 5])
    m2, t2, fp2, meta, mcs = load_experiment(str(tmp_path / "e"))
    assert m2.to_json() == model.to_json() and (t2.events == rec).all() and (t2.original_externals == tr.original_externals).all()
    assert fp2.code == 0x1000103 and meta["seed"] == SEED_BASE + 1 and list(mcs) == [0, 2, 5]
    assert os.path.getsize(str(tmp_path / "e" / "event_trace.bin")) == 16 * len(rec)


def test_format1_directories_of_both_generations_load(tmp_path, oracle):
    """"format": 1 was written with 12-byte records (rounds 1-2) and, before the version was bumped, with 16-byte records
    (round 3): load_experiment tells them apart by content - also when the byte count is a multiple of both sizes - and refuses
    what it cannot tell apart."""
    import json
    from demi_amd.apps import SEED_BASE, raft3_config1
    from demi_amd.schedulers import EventTrace, ViolationFingerprint
    from demi_amd.serialization import REC_EVENT_DTYPE_V1, load_experiment, save_experiment
    model, events, lim = raft3_config1()
    v, rec, _ = oracle.random_execute(model, events, SEED_BASE + 1, lim)
    rec = rec[:len(rec) - len(rec) % 3]                       # 16 n divisible by 12 and 12 n by 16 (n % 12 == 0 below): the ambiguous sizes
    rec = rec[:len(rec) - len(rec) % 12]
    assert len(rec) >= 12
    tr = EventTrace(rec, events[:T.verdict_trace_idx(v.flags)])

    def write(name, dtype, declare):
        d = str(tmp_path / name)
        save_experiment(d, model, tr, ViolationFingerprint(7), limits=lim)
        out = np.zeros(len(rec), dtype=dtype)
        for f in dtype.names:
            out[f] = rec[f]
        out.tofile(os.path.join(d, "event_trace.bin"))
        meta = json.load(open(os.path.join(d, "meta.json")))
        meta["format"] = 1
        if declare:
            meta["rec_event_size"] = dtype.itemsize
        else:
            meta.pop("rec_event_size")
        json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
        return d

    for dtype in (REC_EVENT_DTYPE_V1, T.REC_EVENT_DTYPE):        # the round-1/2 writer and the round-3 writer, undeclared and declared
        for declare in (False, True):
            _m, t2, _fp, meta, _mcs = load_experiment(write("e%d_%d" % (dtype.itemsize, declare), dtype, declare))
            assert meta["format"] == 1 and (t2.events == rec).all()
    # bytes that parse under neither record: refused, not guessed
    d = write("garbage", T.REC_EVENT_DTYPE, False)
    raw = np.fromfile(os.path.join(d, "event_trace.bin"), dtype=np.uint8)
    raw[::16] = 0xEE
    raw.tofile(os.path.join(d, "event_trace.bin"))
    with pytest.raises(ValueError, match="refusing to guess"):
        load_experiment(d)
    # an unknown declared size is refused too
    d = write("odd", T.REC_EVENT_DTYPE, True)
    meta = json.load(open(os.path.join(d, "meta.json")))
    meta["rec_event_size"] = 20
    json.dump(meta, open(os.path.join(d, "meta.json"), "w"))
    with pytest.raises(ValueError):
        load_experiment(d)


def test_jni_shim_compiles_against_the_stub_header_and_covers_the_adapter():
    """jni/demi_jni.c must stay in step with include/demi_gpu.h (no JDK here: a compile-only <jni.h> stands in), and every
    @native method the Scala adapter declares has its Java_..._DemiGpu_<name> function."""
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-s", "-C", os.path.join(root, "jni"), "check"])
    shim = open(os.path.join(root, "jni", "demi_jni.c")).read()
    scala = open(os.path.join(root, "scala", "akka", "dispatch", "verification", "gpu", "DemiGpu.scala")).read()
    natives = set(re.findall(r"@native def (\w+)\(", scala))
    shims = set(re.findall(r"FN\((\w+)\)\(JNIEnv", shim))
    assert natives and natives == shims, (natives - shims, shims - natives)


def test_verdict_dump_for_the_deferred_jvm_check(tmp_path):
    """tools/dump_verdicts.py writes the file set a JVM box diffs against RandomScheduler + FullyRandom(seed) (SURVEY 8c ii)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "dump"
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "dump_verdicts.py"), str(out), "--n", "128", "--traces", "3",
                           "--source", "oracle"])
    meta = json.load(open(out / "meta.json"))
    v = np.fromfile(out / "verdicts.bin", dtype=T.VERDICT_DTYPE)
    seeds = np.fromfile(out / "seeds.bin", dtype=np.uint64)
    assert len(v) == len(seeds) == 128 and meta["violations"] == int((v["flags"] & T.V_VIOLATION != 0).sum())
    for k in meta["recorded"]:
        rec = np.fromfile(out / ("deliveries_%d.bin" % k), dtype=T.REC_EVENT_DTYPE)
        assert int((rec["kind"] == T.REC_MSG_EVENT).sum()) == T.verdict_deliveries(int(v["flags"][k]))


def test_experiment_knobs_are_ignored_without_the_switch():
    """csrc/knobs.hpp: the library's experiment variables select an engine only together with DEMI_EXPERIMENT=1.  Checked on a
    knob whose effect is visible without a GPU: DEMI_JIT_IFCONVERT changes the generated handler source."""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from demi_amd import _native, model as M\n"
            "print(hash(_native.specialize_source(M.raft_model(5).to_struct())))\n") % ROOT
    def run(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("DEMI_EXPERIMENT", "DEMI_JIT_IFCONVERT")}
        e.update(env, PYTHONHASHSEED="0", DEMI_NO_TORCH="1")
        return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, check=True).stdout.strip()
    plain = run()
    assert run(DEMI_JIT_IFCONVERT="3") == plain                                # a stray knob: ignored
    assert run(DEMI_JIT_IFCONVERT="3", DEMI_EXPERIMENT="1") != plain           # with the switch: it acts
    assert run(DEMI_EXPERIMENT="1") == plain


def test_abi_generation_is_checked():
    from demi_amd import _native
    L = _native.lib()
    assert L.demi_abi_version() == T.ABI_VERSION
    hdr = open(os.path.join(ROOT, "include", "demi_gpu.h")).read()
    assert "#define DEMI_ABI_VERSION %du" % T.ABI_VERSION in hdr
