"""CPU suite, part 6: the table -> native code specialiser (demi_model_specialize).  Without a GPU we can still
(a) run the code generator and the hiprtc compilation for gfx950 and (b) compile the generated handler code
for the host and compare it, delivery by delivery, with the oracle's row interpreter."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from demi_amd import _native, types as T
from demi_amd import model as M
from oracle.oracle_py import Effect

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FX_CAP = 8            # DEMI_FX_CAP
MODELS = [("raft5", lambda: M.raft_model(5)), ("raft3_fixed", lambda: M.raft_model(3, election_budget=2, buggy=False)),
          ("shuffle8", lambda: M.shuffle_model(True))]


@pytest.mark.parametrize("name,mk", MODELS)
def test_specialised_kernel_compiles_for_gfx950(name, mk):
    try:
        size, kernel = _native.specialize_check(mk().to_struct())
    except _native.DemiError as e:
        if "hiprtc not found" in str(e):
            pytest.skip("no hiprtc in this environment")
        raise
    assert size > 10000 and "k1_random_explore" in kernel


def test_generated_source_shape():
    m = M.raft_model(5)
    src = _native.specialize_source(m.to_struct())
    assert src.count("\n  L") == len(m.code)                 # one labelled statement per row
    assert "vm_run_jit" in src and "goto done;" in src
    for st in {s for s in m.handler_start if s != 0xFFFF}:
        assert "case %du: goto L%d;" % (st, st) in src


def _fx_schedule(src):
    """The effect-slot schedule of a K1-flavoured source: [(kind, op, type)] per slot, None when the table has none."""
    import re
    line = [l for l in src.splitlines() if l.startswith("#define DEMI_JIT_FX_APPLY")]
    if not line:
        return None
    return [(int(k), int(op), int(t), int(q)) for _, k, op, t, _, q in re.findall(r"DEMI_FX_SLOT\((\d+), (\d+)u, (\d+)u, (\d+)u, (\d+)u, (\d+)u\)", line[0])]


def _host_vm(model, tmp_path, k1=False):
    src = _native.specialize_source(model.to_struct(), k1=k1)
    wide = getattr(model, "wide", False)
    cpp = tmp_path / (("vm_host_wide" if wide else "vm_host") + ("_k1.cpp" if k1 else ".cpp"))
    cpp.write_text('#include "%s"\n%s\nstatic uint64_t g_app = 0x5DEECE66DULL;   // Instrumenter().seededRandom: seed 0\n'
                   'extern "C" void app_reset() { g_app = 0x5DEECE66DULL; }\n'
                   'extern "C" uint32_t run(const uint32_t* hs, uint32_t ac, uint32_t nt, uint64_t* st, '
                   'demi::word_t* fxq, demi::word_t w, uint32_t* flags) {\n  demi::Tables t{hs, ac, nt, nullptr}; demi::LaneMem m{st, fxq};\n'
                   '  uint32_t f = *flags; uint32_t n = demi::vm_run_jit(t, m, w, f, g_app); *flags = f; return n; }\n'
                   % (os.path.join(ROOT, "tests", "jit_host_shim.hpp"), src))
    so = tmp_path / (("vm_host_wide" if wide else "vm_host") + ("_k1.so" if k1 else ".so"))
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-label"] + (["-DDEMI_WIDE"] if wide else []) +
                          ["-DDEMI_JIT_ARR_LEN=%d" % getattr(model, "array_len", 0), "-DDEMI_JIT_NPAY=%d" % getattr(model, "payloads", 2),
                           "-o", str(so), str(cpp)])
    L = C.CDLL(str(so))
    L.run.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64 if wide else C.c_uint32, C.POINTER(C.c_uint32)]
    L.run.restype = C.c_uint32
    L.app_rng = C.c_uint64(0x5DEECE66D)          # the oracle's copy of the application generator (seed 0), advanced in step
    L.fx_schedule = _fx_schedule(src) if k1 else None
    return L


@pytest.mark.parametrize("k1", [False, True], ids=["queue", "k1-schedule"])
@pytest.mark.parametrize("name,mk", MODELS)
def test_generated_handlers_equal_the_row_interpreter(oracle, tmp_path, name, mk, k1):
    """Random states x random messages: same new state, same effect rows (after expanding SEND / BCAST the way the
    apply phase does), same FX_CAP overflow.  k1: the RandomScheduler kernel's flavour, where an effect row fills a fixed
    slot of the table's effect schedule and the apply phase walks the schedule - the filled slots in schedule order must be
    the rows in program order."""
    model = mk()
    L = _host_vm(model, tmp_path, k1)
    sched = L.fx_schedule
    if k1:
        assert sched is not None and len(sched) <= FX_CAP, "these tables have an effect schedule"
    ms = model.to_struct()
    A, NT = model.n_actors, len(model.msg_names)
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    ac = sum((c & 15) << (4 * i) for i, c in enumerate(model.actor_class))
    exists = (1 << A) - 1
    rng = np.random.default_rng(7)
    st = np.zeros(8 * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint32)
    fx = (Effect * 64)()
    n_fx_seen = n_ovf = 0
    for it in range(30000):
        me = int(rng.integers(A))
        typ = int(rng.integers(NT))
        src = int(rng.choice([int(rng.integers(A)), T.DEADLETTERS]))
        p0, p1 = int(rng.integers(256)), int(rng.integers(256))
        # states near the reachable region (small field values) and fully random ones
        state = int.from_bytes(bytes(int(x) for x in (rng.integers(0, 6, 8) if it % 3 else rng.integers(0, 256, 8))), "little")
        w = typ | (me << 5) | (src << 8) | (p0 << 16) | (p1 << 24)
        st[me * 64] = state
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, ac, NT, st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        want_state = C.c_uint64(state)
        wn = oracle.lib().orc_vm_run(C.byref(ms), me, C.byref(want_state), typ, src, p0, p1, exists, fx, 64, C.byref(L.app_rng))
        if wn < 0:
            assert flags.value & T.V_QUEUE_OVF
            n_ovf += 1
            continue
        assert not flags.value
        assert int(st[me * 64]) == want_state.value, (it, me, typ, hex(state))
        got = []
        for k in (range(n) if sched is None else [j for j in range(len(sched)) if (n >> j) & 1]):
            f = int(fxq[(k if sched is None else sched[k][3]) * 64])     # (a send slot's entry of the effect queue)
            op, t_, target, q0, q1 = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & 255, (f >> 22) & 255
            if sched is not None and sched[k][0] != 0:       # a timer slot carries no data: the slot is the row
                op, t_ = sched[k][1], sched[k][2]
            elif sched is not None:
                assert op in (M.OPS["SEND"], M.OPS["BCAST"])
            if op == M.OPS["SEND"]:
                if target < A:
                    got.append((0, target, t_, q0, q1))
            elif op == M.OPS["BCAST"]:
                got += [(0, r, t_, q0, q1) for r in range(A) if r != me]
            else:
                got.append((1 + op - M.OPS["TSET"], me, t_, 0, 0))
        want = [(e.kind, e.target, e.msg_type, e.p0, e.p1) for e in fx[:wn]]
        assert got == want, (it, me, typ, hex(state), got, want)
        n_fx_seen += len(got)
    assert n_fx_seen > 1000


def test_missing_runtime_compiler_is_reported_not_fatal(tmp_path):
    """Without hiprtc the specialiser says so (the launch paths then keep the table interpreter); checked in a fresh
    process because the hiprtc handle is opened once per process."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from demi_amd import _native, model as M\n"
            "try:\n    _native.specialize_check(M.raft_model(3).to_struct())\n"
            "except _native.DemiError as e:\n    print('ERR', e)\n" % ROOT)
    env = dict(os.environ, DEMI_HIPRTC_LIB=str(tmp_path / "no_such_libhiprtc.so"))
    out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "ERR" in out.stdout and "hiprtc not found" in out.stdout, out.stdout + out.stderr


def _random_handler(rng, n_rows, n_types, few_effects=False):
    """A random valid handler: every op of the table, forward skips / guards of random length, registers and
    immediates mixed.  few_effects: effect rows are rare (whole executions then stay within the capacities)."""
    a = M.Asm()
    regs = [M.Reg(i) for i in range(16)]
    alu = ["add", "sub", "and_", "or_", "xor", "shl", "shr", "bitset", "eq", "ne", "lt", "ge", "le", "gt", "min", "max"]
    pending = []                      # labels that must still be placed
    for i in range(n_rows):
        for lab in [l for l in pending if l[1] == i]:
            a.label(lab[0]); pending.remove(lab)
        k = int(rng.integers(0, 100))
        if few_effects and k >= 78 and rng.integers(0, 4):
            k = int(rng.integers(0, 78))
        breg = lambda: regs[int(rng.integers(16))] if rng.integers(2) else int(rng.integers(256))
        if k < 45:
            getattr(a, alu[int(rng.integers(len(alu)))])(regs[int(rng.integers(12))], regs[int(rng.integers(16))], breg())
        elif k < 52:
            a.mov(regs[int(rng.integers(12))], breg())
        elif k < 56:
            a.popc(regs[int(rng.integers(12))], breg())
        elif k < 78:
            name = "L%d" % i
            tgt = i + 1 + int(rng.integers(0, min(6, n_rows - i)))
            pending.append((name, tgt))
            c = int(rng.integers(0, 9))
            if c < 6:
                getattr(a, ["if_eq", "if_ne", "if_lt", "if_ge", "if_le", "if_gt"][c])(regs[int(rng.integers(16))], breg(), name)
            elif c == 6:
                a.skipz(regs[int(rng.integers(16))], name)
            elif c == 7:
                a.skipnz(regs[int(rng.integers(16))], name)
            else:
                a.skip(name)
        elif k < 88:
            a.send(int(rng.integers(1, 3)), regs[int(rng.integers(16))], regs[int(rng.integers(16))], breg())   # internal types
        elif k < 92:
            a.bcast(int(rng.integers(1, 3)), regs[int(rng.integers(16))], breg())
        elif k < 93:
            a.rnd(regs[int(rng.integers(12))], breg() if rng.integers(2) else int(rng.integers(0, 256)))
        elif k < 95:
            a.tset(n_types - 1)
        elif k < 97:
            a.trep(n_types - 1)
        elif k < 99:
            a.tcancel(n_types - 1)
        else:
            a.halt()
    for lab in pending:
        if lab[0] not in a._labels:
            a.label(lab[0])
    return a


@pytest.mark.parametrize("seed,ifconvert,k1", [(s, c, False) for s in (1, 2, 3, 4) for c in (0, 4)] + [(s, 0, True) for s in (1, 2, 3, 11, 12, 13, 14)])
def test_random_programs_through_the_code_generator(oracle, tmp_path, seed, ifconvert, k1, monkeypatch):
    """Every op, random control flow, two actor classes: generated C++ == the oracle's row interpreter, delivery by
    delivery (state, effect rows, FX_CAP overflow).  k1: the RandomScheduler kernel's flavour of the generated code; seeds
    above 10 build tables with few effect rows, which have an effect-slot schedule (jit.hpp fx_schedule) - the filled slots
    in schedule order must then be the effect rows in program order, on every path the random control flow takes."""
    # ifconvert: the experimental select-based emission of short guarded ALU runs (DEMI_JIT_IFCONVERT), same semantics
    monkeypatch.setenv("DEMI_JIT_IFCONVERT", str(ifconvert))
    rng = np.random.default_rng(seed)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {}
    for cls in range(2):
        for name, _ in MSGS:
            if rng.integers(5):
                h[(cls, name)] = _random_handler(rng, int(rng.integers(3, 40)), len(MSGS), few_effects=seed > 10)
    A = 5
    model = M.build_model("rand%d" % seed, A, MSGS, h, [[0] * 8] * A, (T.INV_NEVER, 0, 200, 0),
                          actor_class=[0, 1, 0, 1, 1], n_classes=2)
    L = _host_vm(model, tmp_path, k1)
    sched = L.fx_schedule
    if seed > 10:
        assert sched is not None and 2 <= len(sched) <= FX_CAP, "a table with few effect rows has a schedule"
    ms = model.to_struct()
    NT = len(MSGS)
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    ac = sum((c & 15) << (4 * i) for i, c in enumerate(model.actor_class))
    st = np.zeros(8 * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint32)
    fx = (Effect * 64)()
    seen_ovf = seen_fx = 0
    for it in range(6000):
        me, typ = int(rng.integers(A)), int(rng.integers(NT))
        src = int(rng.choice([int(rng.integers(A)), T.DEADLETTERS]))
        p0, p1 = int(rng.integers(256)), int(rng.integers(256))
        state = int.from_bytes(bytes(int(x) for x in rng.integers(0, 256, 8)), "little")
        w = typ | (me << 5) | (src << 8) | (p0 << 16) | (p1 << 24)
        st[me * 64] = state
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, ac, NT, st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        want_state = C.c_uint64(state)
        wn = oracle.lib().orc_vm_run(C.byref(ms), me, C.byref(want_state), typ, src, p0, p1, (1 << A) - 1, fx, 64, C.byref(L.app_rng))
        if wn < 0:
            assert flags.value & T.V_QUEUE_OVF
            seen_ovf += 1
            continue
        assert not flags.value and int(st[me * 64]) == want_state.value, (it, me, typ, hex(state))
        got = []
        for k in (range(n) if sched is None else [j for j in range(len(sched)) if (n >> j) & 1]):
            f = int(fxq[(k if sched is None else sched[k][3]) * 64])     # (a send slot's entry of the effect queue)
            op, t_, target, q0, q1 = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & 255, (f >> 22) & 255
            if sched is not None and sched[k][0] != 0:       # a timer slot carries no data: the slot is the row
                op, t_ = sched[k][1], sched[k][2]
            elif sched is not None:
                assert op in (M.OPS["SEND"], M.OPS["BCAST"])
            if op == M.OPS["SEND"]:
                if target < A:
                    got.append((0, target, t_, q0, q1))
            elif op == M.OPS["BCAST"]:
                got += [(0, r, t_, q0, q1) for r in range(A) if r != me]
            else:
                got.append((1 + op - M.OPS["TSET"], me, t_, 0, 0))
        assert got == [(e.kind, e.target, e.msg_type, e.p0, e.p1) for e in fx[:wn]], (it, me, typ)
        seen_fx += len(got)
    assert seen_fx > 200


def test_if_conversion_knob_converts_short_guarded_alu_runs(oracle, tmp_path, monkeypatch):
    """DEMI_JIT_IFCONVERT: `if (cond) { two ALU rows }` becomes two selects (no branch); a guard over an effect row, or
    one whose body is a jump target, stays a branch; results are the interpreter's either way."""
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL)]
    a = M.Asm()
    a.if_eq(M.F[0], 1, "x").add(M.F[1], M.F[1], 1).mov(M.F[2], 7).label("x")          # convertible
    a.if_gt(M.P0, 9, "y").mov(M.T0, 1).send(1, M.T0, M.F[1], 0).label("y")             # effect row inside: stays a branch
    a.skipz(M.P1, "in").if_ne(M.F[3], 0, "z").add(M.F[4], M.F[4], 2).label("in").add(M.F[5], M.F[5], 1).label("z")   # target inside
    model = M.build_model("ifc", 2, MSGS, {(0, "E"): a}, [[0] * 8] * 2, (T.INV_NEVER, 0, 200, 0))
    monkeypatch.setenv("DEMI_JIT_IFCONVERT", "3")
    src = _native.specialize_source(model.to_struct())
    assert src.count(" ? (") == 2 and "bool c0 = false;" in src and src.count("bool c") == 1
    L = _host_vm(model, tmp_path)
    ms = model.to_struct()
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    st = np.zeros(8 * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint32)
    fx = (Effect * 64)()
    rng = np.random.default_rng(0)
    for _ in range(3000):
        state = int.from_bytes(bytes(int(x) for x in rng.integers(0, 3, 8)), "little")
        p0, p1 = int(rng.integers(0, 20)), int(rng.integers(0, 2))
        w = 0 | (0 << 5) | (T.DEADLETTERS << 8) | (p0 << 16) | (p1 << 24)
        st[0] = state
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, 0, len(MSGS), st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        want = C.c_uint64(state)
        wn = oracle.lib().orc_vm_run(C.byref(ms), 0, C.byref(want), 0, T.DEADLETTERS, p0, p1, 3, fx, 64, C.byref(L.app_rng))
        assert int(st[0]) == want.value and n == wn


def _random_handler_wide(rng, n_rows, n_types):
    """_random_handler plus 16-bit constants (LDI16 / MOVHI): a wide table's rows."""
    a = _random_handler(rng, n_rows, n_types, few_effects=True)
    b = M.Asm()
    regs = [M.Reg(i) for i in range(12)]
    for _ in range(int(rng.integers(1, 4))):
        b.ldi16(regs[int(rng.integers(12))], int(rng.integers(0, 65536)))
    b.movhi(regs[int(rng.integers(12))], M.Reg(int(rng.integers(16))), int(rng.integers(256)))
    b.rows += a.rows
    b._fix = [(i + len(b.rows) - len(a.rows), lab) for i, lab in a._fix]
    b._labels = {k: v + len(b.rows) - len(a.rows) for k, v in a._labels.items()}
    return b


def _random_handler_array(rng, n_rows, n_types, wide):
    """A random handler whose rows also load from and store to the actor's array (LDX / STX): indices from registers and
    constants, in and out of range."""
    base = _random_handler_wide(rng, n_rows, n_types) if wide else _random_handler(rng, n_rows, n_types, few_effects=True)
    b = M.Asm()
    regs = [M.Reg(i) for i in range(16)]
    pre = int(rng.integers(2, 7))
    for _ in range(pre):
        idx = regs[int(rng.integers(16))] if rng.integers(3) else int(rng.integers(0, 80))
        if rng.integers(2):
            b.ldx(regs[int(rng.integers(12))], idx)
        else:
            b.stx(idx, regs[int(rng.integers(16))])
    b.rows += base.rows
    b._fix = [(i + pre, lab) for i, lab in base._fix]
    b._labels = {k: v + pre for k, v in base._labels.items()}
    # ... and a few more after the random rows, where the registers hold computed values (reached when no row halts before)
    for _ in range(int(rng.integers(1, 4))):
        if rng.integers(2):
            b.stx(regs[int(rng.integers(16))], regs[int(rng.integers(16))])
        else:
            b.ldx(regs[int(rng.integers(8))], regs[int(rng.integers(16))])
    return b


@pytest.mark.parametrize("seed,wide,alen", [(1, False, 20), (2, False, 64), (3, True, 10), (4, True, 64), (5, False, 3)])
def test_array_tables_through_the_code_generator(oracle, tmp_path, seed, wide, alen):
    """DEMI_MODEL_ARRAY: the generated LDX / STX statements (a narrow LDS access to the element inside the actor's array
    words, which lie behind its field words) equal the oracle's row interpreter on random rows x random (fields, array,
    message), both window widths, lengths that do and do not fill their last word; the kernels compile for gfx950."""
    rng = np.random.default_rng(100 + seed)
    MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
    h = {(0, name): _random_handler_array(rng, int(rng.integers(4, 30)), len(MSGS), wide) for name, _ in MSGS}
    A, NT = 4, len(MSGS)
    model = M.build_model("rand_arr%d" % seed, A, MSGS, h, [[0] * 8] * A, (T.INV_NEVER, 0, 200, 0), wide=wide, array_len=alen)
    assert oracle.model_validate(model)[0] == 0
    stw = model.state_words
    assert stw == (2 if wide else 1) + (alen + (3 if wide else 7)) // (4 if wide else 8)
    if seed in (1, 3):
        try:
            size, kernel = _native.specialize_check(model.to_struct())
            assert size > 10000 and kernel.count("k1_random_explore") == 13 and "k2_replay" in kernel and "k3_dpor" in kernel
        except _native.DemiError as e:
            if "hiprtc not found" not in str(e):
                raise
    L = _host_vm(model, tmp_path)
    ms = model.to_struct()
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    st = np.zeros(8 * stw * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint64 if wide else np.uint32)
    fx = (Effect * 64)()
    want_state = (C.c_uint64 * stw)()
    hi = 65536 if wide else 256
    changed = loaded = 0
    for it in range(5000):
        me, typ = int(rng.integers(A)), int(rng.integers(NT))
        src = int(rng.choice([int(rng.integers(A)), T.DEADLETTERS]))
        p0, p1 = int(rng.integers(hi)), int(rng.integers(hi))
        fields = [int(x) for x in rng.integers(0, min(hi, 90 if it % 3 else hi), 8)]       # (small values index the array)
        words = (M.pack_state_wide(fields) if wide else [M.pack_state(fields)]) + [int(x) for x in rng.integers(0, 1 << 63, stw - (2 if wide else 1), dtype=np.uint64)]
        # the unused tail of the last array word stays zero, as in an execution (no STX reaches it)
        per, bits = (4, 16) if wide else (8, 8)
        if alen % per:
            words[-1] &= (1 << (bits * (alen % per))) - 1
        for k, wv in enumerate(words):
            st[(stw * me + k) * 64] = wv
            want_state[k] = wv
        w = typ | (me << 5) | (src << 8) | (p0 << 16) | (p1 << (32 if wide else 24))
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, 0, NT, st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        wn = oracle.lib().orc_vm_run(C.byref(ms), me, want_state, typ, src, p0, p1, (1 << A) - 1, fx, 64, C.byref(L.app_rng))
        if wn < 0:
            assert flags.value & T.V_QUEUE_OVF
            continue
        assert not flags.value
        got_state = [int(st[(stw * me + k) * 64]) for k in range(stw)]
        assert got_state == [int(x) for x in want_state], (it, me, typ, fields)
        changed += got_state[(2 if wide else 1):] != words[(2 if wide else 1):]
        loaded += got_state[:(2 if wide else 1)] != words[:(2 if wide else 1)]
        got = []
        for k in range(n):
            f = int(fxq[k * 64])
            if wide:
                op, t_, target, q0, q1 = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & 0xFFFF, (f >> 30) & 0xFFFF
            else:
                op, t_, target, q0, q1 = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & 0xFF, (f >> 22) & 0xFF
            if op == M.OPS["SEND"]:
                if target < A:
                    got.append((0, target, t_, q0, q1))
            elif op == M.OPS["BCAST"]:
                got += [(0, r, t_, q0, q1) for r in range(A) if r != me]
            else:
                got.append((1 + op - M.OPS["TSET"], me, t_, 0, 0))
        assert got == [(e.kind, e.target, e.msg_type, e.p0, e.p1) for e in fx[:wn]], (it, me, typ)
    assert changed > 300 and loaded > 300


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_wide_tables_through_the_code_generator(oracle, tmp_path, seed):
    """DEMI_MODEL_WIDE: the generated handlers over the 16 x u16 window (masks, shifts by b & 15, 16-bit POPC, MOVHI, state in
    two words, 64-bit message / effect words) equal the oracle's wide row interpreter on random rows and on the raft lowered
    with terms above 255; the kernel source compiles for gfx950."""
    rng = np.random.default_rng(seed)
    if seed == 1:
        model = M.raft_model(5, term0=1000, loglen0=300)
        A, NT = 5, model.n_msg_types
    else:
        MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
        h = {(0, name): _random_handler_wide(rng, int(rng.integers(4, 40)), len(MSGS)) for name, _ in MSGS}
        A, NT = 4, len(MSGS)
        model = M.build_model("rand_wide%d" % seed, A, MSGS, h, [[0] * 8] * A, (T.INV_NEVER, 0, 60000, 0), wide=True)
    assert oracle.model_validate(model)[0] == 0
    try:
        size, kernel = _native.specialize_check(model.to_struct())
        assert size > 10000 and "k1_random_explore" in kernel
    except _native.DemiError as e:
        if "hiprtc not found" not in str(e):
            raise
    L = _host_vm(model, tmp_path)
    ms = model.to_struct()
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    st = np.zeros(16 * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint64)
    fx = (Effect * 64)()
    want_state = (C.c_uint64 * 2)()
    seen_fx = seen_big = 0
    for it in range(6000):
        me, typ = int(rng.integers(A)), int(rng.integers(NT))
        src = int(rng.choice([int(rng.integers(A)), T.DEADLETTERS]))
        hi = 65536 if it % 2 else 1200
        p0, p1 = int(rng.integers(hi)), int(rng.integers(hi))
        fields = [int(x) for x in rng.integers(0, hi, 8)]
        if seed == 1:
            fields[0], fields[2] = int(rng.integers(3)), int(rng.choice([M.NOBODY, int(rng.integers(A))]))
        w0, w1 = M.pack_state_wide(fields)
        st[(2 * me) * 64], st[(2 * me + 1) * 64] = w0, w1
        w = typ | (me << 5) | (src << 8) | (p0 << 16) | (p1 << 32)
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, 0, NT, st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        want_state[0], want_state[1] = w0, w1
        wn = oracle.lib().orc_vm_run(C.byref(ms), me, want_state, typ, src, p0, p1, (1 << A) - 1, fx, 64, C.byref(L.app_rng))
        if wn < 0:
            assert flags.value & T.V_QUEUE_OVF
            continue
        assert not flags.value
        assert (int(st[(2 * me) * 64]), int(st[(2 * me + 1) * 64])) == (want_state[0], want_state[1]), (it, me, typ, fields)
        got = []
        for k in range(n):
            f = int(fxq[k * 64])
            op, t_, target, q0, q1 = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & 0xFFFF, (f >> 30) & 0xFFFF
            if op == M.OPS["SEND"]:
                if target < A:
                    got.append((0, target, t_, q0, q1))
            elif op == M.OPS["BCAST"]:
                got += [(0, r, t_, q0, q1) for r in range(A) if r != me]
            else:
                got.append((1 + op - M.OPS["TSET"], me, t_, 0, 0))
        assert got == [(e.kind, e.target, e.msg_type, e.p0, e.p1) for e in fx[:wn]], (it, me, typ)
        seen_fx += len(got)
        seen_big += any(v > 255 for v in (int(st[(2 * me) * 64]) >> 16 & 0xFFFF, *[g[3] for g in got]))
    assert seen_fx > 200 and seen_big > 100


def _random_handler_payloads(rng, n_rows, n_types, npay):
    """A random wide handler whose rows also read the message's payload fields (LDP, in and out of range) and stage the
    further fields of what it sends (PSET, registers and constants)."""
    base = _random_handler_wide(rng, n_rows, n_types)
    b = M.Asm()
    regs = [M.Reg(i) for i in range(16)]
    pre = int(rng.integers(2, 7))
    for _ in range(pre):
        if rng.integers(2):
            b.ldp(regs[int(rng.integers(12))], int(rng.integers(0, 6)))
        else:
            b.pset(int(rng.integers(2, 6)), regs[int(rng.integers(16))] if rng.integers(3) else int(rng.integers(0, 256)))
    b.rows += base.rows
    b._fix = [(i + pre, lab) for i, lab in base._fix]
    b._labels = {k: v + pre for k, v in base._labels.items()}
    return b


@pytest.mark.parametrize("npay", [3, 4, 5, 6])
def test_payload_tables_through_the_code_generator(oracle, tmp_path, npay):
    """DEMI_MODEL_PAYLOADS(n): the generated handlers - payload fields read with LDP (and P0 / P1), staged with PSET, packed into
    the 48-bit area of the effect word at the width of the model - equal the oracle's row interpreter on random rows, and on the
    raft with akka-raft's field sets (n = 5); the kernel source compiles for gfx950."""
    rng = np.random.default_rng(70 + npay)
    if npay == 5:
        model = M.raft_model(5, log_cap=8, real_fields=True)
        A, NT = 5, model.n_msg_types
    else:
        MSGS = [("E", T.MSG_EXTERNAL), ("A", T.MSG_INTERNAL), ("B", T.MSG_INTERNAL), ("Tm", T.MSG_TIMER)]
        h = {(0, name): _random_handler_payloads(rng, int(rng.integers(4, 40)), len(MSGS), npay) for name, _ in MSGS}
        A, NT = 4, len(MSGS)
        model = M.build_model("rand_pay%d" % npay, A, MSGS, h, [[0] * 8] * A, (T.INV_NEVER, 0, 60000, 0), wide=True, payloads=npay)
    assert model.payloads == npay and oracle.model_validate(model)[0] == 0
    try:
        size, kernel = _native.specialize_check(model.to_struct())
        assert size > 10000 and "k1_random_explore" in kernel and "k2_replay" in kernel and "k3_dpor" in kernel
    except _native.DemiError as e:
        if "hiprtc not found" not in str(e):
            raise
    L = _host_vm(model, tmp_path)
    ms = model.to_struct()
    hs = np.full(T.MAX_CLASSES * T.MAX_MSG_TYPES, 0xFFFF, dtype=np.uint32)
    hs[:len(model.handler_start)] = model.handler_start
    sw = model.state_words
    st = np.zeros(8 * sw * 64, dtype=np.uint64)
    fxq = np.zeros(FX_CAP * 64, dtype=np.uint64)
    fx = (Effect * 64)()
    want_state = (C.c_uint64 * sw)()
    bits = T.payload_bits(npay)
    seen_fx = seen_far = 0
    for it in range(5000):
        me, typ = int(rng.integers(A)), int(rng.integers(NT))
        src = int(rng.choice([int(rng.integers(A)), T.DEADLETTERS]))
        hi = 65536 if it % 2 else 9
        pay = [int(x) for x in rng.integers(0, hi, 6)]
        fields = [int(x) for x in rng.integers(0, hi, 8)]
        if npay == 5:
            fields[0], fields[2], fields[5] = int(rng.integers(3)), int(rng.choice([M.NOBODY, int(rng.integers(A))])), int(rng.integers(9))
        words = M.pack_state_wide(fields) + [int(x) for x in rng.integers(0, 1 << 63, sw - 2)]
        for k in range(sw):
            st[(sw * me + k) * 64] = words[k]
            want_state[k] = words[k]
        area = T.payload_area(pay, npay)
        w = typ | (me << 5) | (src << 8) | (area << 16)
        flags = C.c_uint32(0)
        n = L.run(hs.ctypes.data, 0, NT, st.ctypes.data, fxq.ctypes.data, w, C.byref(flags))
        wn = oracle.lib().orc_vm_run_area(C.byref(ms), me, want_state, typ, src, area, (1 << A) - 1, fx, 64, C.byref(L.app_rng))
        if wn < 0:
            assert flags.value & T.V_QUEUE_OVF
            continue
        assert not flags.value
        assert [int(st[(sw * me + k) * 64]) for k in range(sw)] == list(want_state), (it, me, typ, fields)
        got = []
        for k in range(n):
            f = int(fxq[k * 64])
            op, t_, target, ar = f & 31, (f >> 5) & 31, (f >> 10) & 15, (f >> 14) & ((1 << 48) - 1)
            if op == M.OPS["SEND"]:
                if target < A:
                    got.append((0, target, t_, ar))
            elif op == M.OPS["BCAST"]:
                got += [(0, r, t_, ar) for r in range(A) if r != me]
            else:
                got.append((1 + op - M.OPS["TSET"], me, t_, 0))
        assert got == [(e.kind, e.target, e.msg_type, e.area if e.kind == 0 else 0) for e in fx[:wn]], (it, me, typ)
        assert all(e.kind != 0 or e.area < (1 << (npay * bits)) for e in fx[:wn])
        seen_fx += len(got)
        seen_far += any(g[0] == 0 and (g[3] >> (2 * bits)) for g in got)
    assert seen_fx > 200 and seen_far > 50, (seen_fx, seen_far)


def _vgpr_count(image):
    """.vgpr_count of a code object (msgpack metadata note): fixint / uint8 / uint16 after the key."""
    best, key = -1, b".vgpr_count"
    i = image.find(key)
    while i >= 0:
        t = image[i + len(key)]
        v = t if t <= 0x7F else image[i + len(key) + 1] if t == 0xCC else (image[i + len(key) + 1] << 8 | image[i + len(key) + 2]) if t == 0xCD else -1
        best = max(best, v)
        i = image.find(key, i + 1)
    return best


@pytest.mark.parametrize("system_comgr", [False, True])
def test_specialised_k1_keeps_six_waves_per_simd_with_either_compiler(tmp_path, system_comgr):
    """The specialised K1 needs <= 80 VGPRs for 6 waves per SIMD (with 5 it is 11 % slower).  Round 2's kernel sat at 79
    with PyTorch's bundled compiler (ROCm 7.0.2) and at 83 with /opt/rocm's (7.2), and a second compilation with the occupancy
    stated pulled the latter back; round 3's kernel (effect-slot schedule, timer directory instead of a 64-bit timer mask)
    needs 72 / 74, so the rule is gone and this test pins the margin: <= 80 and no spills with either compiler."""
    comgr = "/opt/rocm/lib/libamd_comgr.so.3"
    if system_comgr and not os.path.exists(comgr):
        pytest.skip("no system comgr next to the bundled one")
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from demi_amd import _native, model as M\n"
            "try:\n"
            "    print('SIZE', _native.specialize_check(M.raft_model(5).to_struct())[0])\n"
            "except _native.DemiError as e:\n"
            "    print('ERR', e)\n" % ROOT)
    env = dict(os.environ, DEMI_JIT_DUMP=str(tmp_path / "img"), DEMI_SPECIALIZE_CHECK_K1_ONLY="1")
    if system_comgr:
        env["LD_PRELOAD"] = comgr
    out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    if "hiprtc not found" in out.stdout:
        pytest.skip("no hiprtc in this environment")
    assert "SIZE" in out.stdout, out.stdout + out.stderr
    image = open(str(tmp_path / "img") + ".0", "rb").read()          # kernel 0 = K1 (FullyRandom)
    assert 0 < _vgpr_count(image) <= 80
    assert b".vgpr_spill_count" in image and image[image.find(b".vgpr_spill_count") + len(b".vgpr_spill_count")] == 0


def _meta_values(image, key):
    """every value of msgpack key `key` in a code object's metadata note (fixint / uint8 / uint16 / uint32)"""
    vals, k = [], key.encode()
    i = image.find(k)
    while i >= 0:
        p = i + len(k)
        t = image[p]
        vals.append(t if t <= 0x7F else image[p + 1] if t == 0xCC else int.from_bytes(image[p + 1:p + 3], "big") if t == 0xCD
                    else int.from_bytes(image[p + 1:p + 5], "big") if t == 0xCE else -1)
        i = image.find(k, i + 1)
    return vals


@pytest.mark.parametrize("wide", [False, True, "array"])
def test_no_specialised_kernel_touches_scratch_memory(tmp_path, wide):
    """None of the compiled kernels may keep anything in the stack frame (scratch = HBM-backed private memory): the replay and
    DPOR kernels are latency-bound chains, and a variable that lives there costs a memory round trip per use.  Round 3 found
    K2 with 240 bytes of it - the candidate mask, the cursor and the kernel arguments, pinned by ONE expression that selected
    among four variables and was compiled into a table of their addresses."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from demi_amd import _native, model as M\n"
            "w = %r\n"
            "m = M.replog_model(5, 16, True, False) if w == 'array' else M.raft_model(5, term0=1000, loglen0=300) if w else M.raft_model(5)\n"
            "try:\n"
            "    print('SIZE', _native.specialize_check(m.to_struct())[0])\n"
            "except _native.DemiError as e:\n"
            "    print('ERR', e)\n" % (ROOT, wide))
    env = dict(os.environ, DEMI_JIT_DUMP=str(tmp_path / "img"))
    out = subprocess.run([os.sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
    if "hiprtc not found" in out.stdout:
        pytest.skip("no hiprtc in this environment")
    assert "SIZE" in out.stdout, out.stdout + out.stderr
    seen = 0
    for k in range(13):            # (demi_gpu.hip JK_COUNT: a wide table gets every variant of K1 compiled)
        path = str(tmp_path / "img") + ".%d" % k
        if not os.path.exists(path):
            continue
        image = open(path, "rb").read()
        sizes = _meta_values(image, ".private_segment_fixed_size")
        assert sizes and all(v == 0 for v in sizes), (k, sizes)
        assert all(v == 0 for v in _meta_values(image, ".vgpr_spill_count")), k
        seen += 1
    assert seen >= (10 if wide else 4)


@pytest.mark.parametrize("wide", [False, True])
def test_generated_invariant_program_equals_the_oracle(oracle, tmp_path, wide):
    """DEMI_INV_PROGRAM: the per-actor program as generated code (jit.hpp inv_prog_jit), compiled for the host, against the
    oracle's row interpreter - hit and key of every actor on random states, random programs; the later trials with DEMI_OP_PEER
    rows (the fields of the other actors, some of them not created)."""
    from tests.test_oracle_cpu import _random_pure_program, _py_rows
    rng = np.random.default_rng(21 + wide)
    msgs = [("E", T.MSG_EXTERNAL)]
    h = {(0, "E"): M.Asm().add(M.F[3], M.F[3], 1)}
    hi = 65536 if wide else 256
    for trial in range(9):
        model = M.build_model("ip%d" % trial, 5, msgs, h, [[0] * 8] * 5,
                              (T.INV_AT_MOST_ONE, _random_pure_program(rng, int(rng.integers(4, 30)), peers=trial >= 6)), wide=wide)
        src = _native.specialize_source(model.to_struct())
        assert "inv_prog_jit" in src
        cpp = tmp_path / ("inv%d_%d.cpp" % (trial, wide))
        cpp.write_text('#include "%s"\n%s\nextern "C" uint32_t inv(const uint64_t* st, uint32_t actor, uint32_t* key, uint32_t exists) {\n'
                       '  uint32_t k = 0; const uint32_t hit = demi::inv_prog_jit(st, actor, k, exists, 5); *key = k; return hit; }\n'
                       % (os.path.join(ROOT, "tests", "jit_host_shim.hpp"), src))
        so = tmp_path / ("inv%d_%d.so" % (trial, wide))
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-label", "-Wno-unused-variable"] +
                              (["-DDEMI_WIDE"] if wide else []) + ["-o", str(so), str(cpp)])
        L = C.CDLL(str(so))
        L.inv.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.c_uint32]
        L.inv.restype = C.c_uint32
        st = np.zeros(16 * 64, dtype=np.uint64)
        hits = 0
        for it in range(400):
            actor = int(rng.integers(5))
            allf = [[int(x) for x in rng.integers(0, hi if it % 2 else 5, 8)] for _ in range(5)]
            exists = int(rng.integers(0, 32)) | (1 << actor)
            fields = allf[actor]
            for a_ in range(5):
                words = M.pack_state_wide(allf[a_]) if wide else [M.pack_state(allf[a_])]
                for k, wv in enumerate(words):
                    st[(len(words) * a_ + k) * 64] = wv
            key = C.c_uint32(0)
            hit = L.inv(st.ctypes.data, actor, C.byref(key), exists)
            r = _py_rows(model.code, model.inv_fa, fields + [0] * 7 + [actor], hi - 1, 15 if wide else 7, peers=(allf, exists))
            assert (hit != 0) == (r[8] != 0) and (hit == 0 or key.value == r[9]), (trial, it, fields, hit, key.value, r[8], r[9])
            hits += hit != 0
        assert hits > 20 or trial > 0


def test_committed_k1_counters_describe_the_kernel_this_tree_compiles(tmp_path):
    """profiles/k1_counters.json (the rocprofv3 --pmc passes bench.py quotes as roofline.traffic / issue_model) names the code
    object it was taken on: `code_id` = FNV-1a over .text of the K1 specialised for the default raft5 table, what
    demi_model_code_id returns on the GPU box.  The same compiler runs here, so the id of what THIS tree compiles is known
    without a GPU - and a change to the kernel, the code generator or the table that is committed without re-taking the
    profile (tools/profile_r5.sh) fails here instead of silently nulling the driver line's counters."""
    import json
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from jit_stats import text_hash
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from demi_amd import _native\n"
            "from demi_amd.apps import raft5_config2\n"
            "try:\n"
            "    print('SIZE', _native.specialize_check(raft5_config2()[0].to_struct())[0])\n"
            "except _native.DemiError as e:\n"
            "    print('ERR', e)\n" % ROOT)
    env = dict(os.environ, DEMI_EXPERIMENT="1", DEMI_JIT_DUMP=str(tmp_path / "img"), DEMI_SPECIALIZE_CHECK_K1_ONLY="1")
    for k in ("DEMI_JIT_FLAGS", "DEMI_JIT_DEFINES", "DEMI_JIT_K1_HOT", "LD_PRELOAD"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    if "hiprtc not found" in out.stdout:
        pytest.skip("no hiprtc in this environment")
    assert "SIZE" in out.stdout, out.stdout + out.stderr
    here = text_hash(open(str(tmp_path / "img") + ".0", "rb").read())
    with open(os.path.join(ROOT, "profiles", "k1_counters.json")) as f:
        prof = json.load(f)
    assert prof.get("code_id") == here, ("profiles/k1_counters.json was taken on code id %s, this tree compiles %s: "
                                         "re-run tools/profile_r5.sh on the GPU box and commit its k1_counters.json" % (prof.get("code_id"), here))
