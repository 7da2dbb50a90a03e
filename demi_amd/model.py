"""Host-side lowering of an application's actors to the flat transition table of
include/demi_gpu.h, plus the synthetic applications the benchmarks run.

The reference schedules real Akka actors (NetSys/demi-applications: akka-raft, Spark) whose
`receive` functions are arbitrary JVM code outside /root/reference.  On the GPU path the Scala
adapter lowers each (actor class, message type) handler to a micro-program of guarded rows;
`Asm` below is that lowering's assembler.
"""
import ctypes as C
import json
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

from . import types as T


class Reg(int):
    """Register operand r0..r15 of a handler's 16 x u8 window."""
    def __repr__(self):
        return "r%d" % int(self)


F = [Reg(i) for i in range(8)]          # receiving actor's persisted state fields
T0, T1, T2, T3 = Reg(8), Reg(9), Reg(10), Reg(11)   # temporaries, zero at handler entry
P0, P1 = Reg(12), Reg(13)               # message payload
SRC, ME = Reg(14), Reg(15)              # sender id (15 = deadLetters), own id

OPS = dict(HALT=0, MOV=1, ADD=2, SUB=3, AND=4, OR=5, XOR=6, SHL=7, SHR=8, BITSET=9, POPC=10,
           EQ=11, NE=12, LT=13, GE=14, LE=15, GT=16, MIN=17, MAX=18, MOVHI=19, SKIPZ=20, SKIPNZ=21, SKIP=22,
           SEND=24, BCAST=25, TSET=26, TREP=27, TCANCEL=28, CRASH=29, RND=30, IFEQ=32, IFNE=33, IFLT=34, IFGE=35, IFLE=36, IFGT=37,
           LDX=38, STX=39, PEER=40, LDP=41, PSET=42)


PEER_CREATED = 8     # Asm.peer(dst, actor, PEER_CREATED): is that actor created


def row(op, dst=0, a=0, bimm=0, aux=0, b=0):
    assert 0 <= dst < 16 and 0 <= a < 16 and 0 <= aux < 128 and 0 <= b < 256
    return op | (dst << 8) | (a << 12) | (bimm << 16) | (aux << 17) | (b << 24)


class Asm:
    """Assembler for one handler (forward-only control flow)."""

    def __init__(self):
        self.rows: List[int] = []
        self._fix = []       # (row index, label)
        self._labels: Dict[str, int] = {}

    def _b(self, b):
        if isinstance(b, Reg):
            return 0, int(b)
        assert 0 <= int(b) < 256
        return 1, int(b)

    def alu(self, name, dst, a, b):
        assert isinstance(dst, Reg) and isinstance(a, Reg)
        bimm, bv = self._b(b)
        self.rows.append(row(OPS[name], int(dst), int(a), bimm, 0, bv))
        return self

    def mov(self, dst, b):
        bimm, bv = self._b(b)
        self.rows.append(row(OPS["MOV"], int(dst), 0, bimm, 0, bv))
        return self

    def movhi(self, dst, a, hi):
        """DEMI_MODEL_WIDE only: dst = (a & 0xFF) | (hi << 8)."""
        assert isinstance(dst, Reg) and isinstance(a, Reg) and 0 <= int(hi) < 256
        self.rows.append(row(OPS["MOVHI"], int(dst), int(a), 1, 0, int(hi)))
        return self

    def ldi16(self, dst, value):
        """dst = a 16-bit constant (wide models): MOV of the lower half, MOVHI of the upper one when it is not zero."""
        assert 0 <= int(value) < 65536
        self.mov(dst, int(value) & 0xFF)
        if int(value) >> 8:
            self.movhi(dst, dst, int(value) >> 8)
        return self

    def ldx(self, dst, index):
        """dst = ARRAY[index] (DEMI_MODEL_ARRAY: the actor's array beside its eight fields; index a register or a constant;
        past the end: 0)."""
        bimm, bv = self._b(index)
        self.rows.append(row(OPS["LDX"], int(dst), 0, bimm, 0, bv))
        return self

    def stx(self, index, value):
        """ARRAY[index] = value (a register; past the end: nothing)."""
        assert isinstance(value, Reg)
        bimm, bv = self._b(index)
        self.rows.append(row(OPS["STX"], 0, int(value), bimm, 0, bv))
        return self

    def peer(self, dst, actor, field):
        """DEMI_INV_PROGRAM rows only: dst = field `field` (0..7, or a field register F[k]) of the actor whose id is in register
        `actor`; field 8 (PEER_CREATED) = 1 if that actor is created, else 0.  An id that is not a created actor reads 0."""
        assert isinstance(dst, Reg) and isinstance(actor, Reg)
        f = int(field)
        assert 0 <= f <= 8
        self.rows.append(row(OPS["PEER"], int(dst), int(actor), 0, f, 0))
        return self

    def rnd(self, dst, bound):
        """dst = Instrumenter().seededRandom.nextInt(bound): the application's own generator (restarts at seed 0 with every
        execution); bound 1..255, an immediate or a register."""
        bimm, bv = self._b(bound)
        self.rows.append(row(OPS["RND"], int(dst), 0, bimm, 0, bv))
        return self

    def popc(self, dst, b):
        bimm, bv = self._b(b)
        self.rows.append(row(OPS["POPC"], int(dst), 0, bimm, 0, bv))
        return self

    def _skip(self, name, a, label):
        self._fix.append((len(self.rows), label))
        self.rows.append(row(OPS[name], 0, int(a), 1, 0, 0))
        return self

    def _if(self, name, a, b, label):
        """Fused guard: rows up to `label` execute only if (a OP b)."""
        assert isinstance(a, Reg)
        bimm, bv = self._b(b)
        self._fix.append((len(self.rows), label))
        self.rows.append(row(OPS[name], 0, int(a), bimm, 0, bv))
        return self

    def skipz(self, a, label):
        return self._skip("SKIPZ", a, label)

    def skipnz(self, a, label):
        return self._skip("SKIPNZ", a, label)

    def skip(self, label):
        return self._skip("SKIP", Reg(0), label)

    def label(self, name):
        assert name not in self._labels
        self._labels[name] = len(self.rows)
        return self

    def ldp(self, dst, k):
        """dst = payload field k (0..5) of the message being handled (DEMI_MODEL_PAYLOADS; P0 / P1 are also registers)."""
        assert isinstance(dst, Reg) and 0 <= int(k) < T.MAX_PAYLOADS
        self.rows.append(row(OPS["LDP"], int(dst), 0, 1, 0, int(k)))
        return self

    def pset(self, k, value):
        """Payload field k (2..5) of the messages sent next = value (register or immediate); kept until overwritten."""
        assert 2 <= int(k) < T.MAX_PAYLOADS
        bimm, bv = self._b(value)
        self.rows.append(row(OPS["PSET"], 0, 0, bimm, int(k), bv))
        return self

    def _more(self, more):
        for k, v in enumerate(more):
            self.pset(2 + k, v)

    def send(self, msg_type, target, p0=0, p1=0, *more):
        """`target ! Msg(p0, p1, ...)`; p0 must be a register (use a temp for constants).  Further fields (a table with
        DEMI_MODEL_PAYLOADS) are staged with PSET rows in front of the SEND."""
        assert isinstance(target, Reg) and isinstance(p0, Reg)
        self._more(more)
        bimm, bv = self._b(p1)
        self.rows.append(row(OPS["SEND"], int(p0), int(target), bimm, msg_type, bv))
        return self

    def bcast(self, msg_type, p0, p1=0, *more):
        assert isinstance(p0, Reg)
        self._more(more)
        bimm, bv = self._b(p1)
        self.rows.append(row(OPS["BCAST"], int(p0), 0, bimm, msg_type, bv))
        return self

    def tset(self, msg_type):
        self.rows.append(row(OPS["TSET"], 0, 0, 1, msg_type, 0))
        return self

    def trep(self, msg_type):
        self.rows.append(row(OPS["TREP"], 0, 0, 1, msg_type, 0))
        return self

    def tcancel(self, msg_type):
        self.rows.append(row(OPS["TCANCEL"], 0, 0, 1, msg_type, 0))
        return self

    def crash(self):
        """The receive throws here (Instrumenter.actorCrashed): the actor is blocked until a Start() of its name."""
        self.rows.append(row(OPS["CRASH"], 0, 0, 1, 0, 0))
        return self

    def halt(self):
        self.rows.append(row(OPS["HALT"]))
        return self

    def finish(self) -> List[int]:
        if not self.rows or (self.rows[-1] & 0xFF) != OPS["HALT"]:
            self.halt()
        for idx, label in self._fix:
            dist = self._labels[label] - (idx + 1)
            if (self.rows[idx] & 0xFF) >= OPS["IFEQ"]:
                assert 0 <= dist < 128, "guarded block must be forward, < 128 rows"
                self.rows[idx] |= dist << 17
            else:
                assert 0 <= dist < 256, "skip must be forward, < 256 rows"
                self.rows[idx] |= dist << 24
        return self.rows


for _n in ("ADD", "SUB", "AND", "OR", "XOR", "SHL", "SHR", "BITSET", "EQ", "NE", "LT", "GE", "LE", "GT",
           "MIN", "MAX"):
    def _mk(n):
        def f(self, dst, a, b):
            return self.alu(n, dst, a, b)
        return f
    setattr(Asm, _n.lower() if _n not in ("AND", "OR") else _n.lower() + "_", _mk(_n))


for _n in ("IFEQ", "IFNE", "IFLT", "IFGE", "IFLE", "IFGT"):
    def _mk2(n):
        def f(self, a, b, label):
            return self._if(n, a, b, label)
        return f
    setattr(Asm, "if_" + _n[2:].lower(), _mk2(_n))


@dataclass
class Model:
    """The application lowered to a table (demi_model)."""
    name: str
    n_actors: int
    msg_names: List[str]
    msg_class: List[int]
    actor_class: List[int]
    n_classes: int
    handler_start: List[int]          # [n_classes * n_msg_types]
    code: List[int]
    init_state: List[int]             # u64 per actor (wide: two per actor, F0..F3 then F4..F7, 16 bits each)
    inv_kind: int = T.INV_NONE
    inv_fa: int = 0
    inv_va: int = 0
    inv_fb: int = 0
    fp_match_mask: int = 0xFFFFFFFF
    wide: bool = False                # DEMI_MODEL_WIDE: 16 x u16 register window
    array_len: int = 0                # DEMI_MODEL_ARRAY: elements of every actor's array (LDX / STX), 0 = none
    payloads: int = 2                 # DEMI_MODEL_PAYLOADS: payload fields per message (3..6 need a wide table)
    _keep: list = field(default_factory=list, repr=False, compare=False)

    @property
    def n_msg_types(self):
        return len(self.msg_names)

    @property
    def compiled_only(self):
        """A wide table, and one with arrays, has no interpreter on the device: the schedulers compile it right after loading it."""
        return bool(self.wide or self.array_len)

    @property
    def state_words(self):
        """64-bit words of one actor's state: its field word(s), then its array (8 elements to a word, 4 when wide)."""
        per = 4 if self.wide else 8
        return (2 if self.wide else 1) + (self.array_len + per - 1) // per

    def to_struct(self) -> T.ModelStruct:
        mc = (C.c_uint8 * len(self.msg_class))(*self.msg_class)
        ac = (C.c_uint8 * len(self.actor_class))(*self.actor_class)
        hs = (C.c_uint16 * len(self.handler_start))(*self.handler_start)
        code = (C.c_uint32 * len(self.code))(*self.code)
        init = (C.c_uint64 * len(self.init_state))(*self.init_state)
        s = T.ModelStruct(self.n_actors, self.n_msg_types, self.n_classes, len(self.code),
                          C.cast(mc, C.POINTER(C.c_uint8)), C.cast(ac, C.POINTER(C.c_uint8)),
                          C.cast(hs, C.POINTER(C.c_uint16)), C.cast(code, C.POINTER(C.c_uint32)),
                          C.cast(init, C.POINTER(C.c_uint64)),
                          self.inv_kind, self.inv_fa, self.inv_va, self.inv_fb, self.fp_match_mask,
                          (T.MODEL_WIDE if self.wide else 0) | T.MODEL_ARRAY(self.array_len) | T.MODEL_PAYLOADS(self.payloads))
        self._keep = [mc, ac, hs, code, init]   # keep the buffers alive as long as the Model
        return s

    def to_json(self) -> dict:
        d = {k: getattr(self, k) for k in ("name", "n_actors", "msg_names", "msg_class", "actor_class",
                                            "n_classes", "handler_start", "code", "init_state", "inv_kind",
                                            "inv_fa", "inv_va", "inv_fb", "fp_match_mask")}
        if self.wide:
            d["wide"] = True
        if self.array_len:
            d["array_len"] = self.array_len
        if self.payloads != 2:
            d["payloads"] = self.payloads
        return d

    @staticmethod
    def from_json(d: dict) -> "Model":
        return Model(**d)


def pack_state(fields: List[int]) -> int:
    assert len(fields) <= 8
    s = 0
    for i, v in enumerate(fields):
        assert 0 <= v < 256
        s |= v << (8 * i)
    return s


def pack_state_wide(fields: List[int]) -> List[int]:
    """The two state words of an actor of a wide model: F0..F3 and F4..F7, 16 bits each."""
    f = list(fields) + [0] * (8 - len(fields))
    assert len(f) == 8 and all(0 <= v < 65536 for v in f)
    return [sum(v << (16 * i) for i, v in enumerate(f[:4])), sum(v << (16 * i) for i, v in enumerate(f[4:]))]


def build_model(name, n_actors, msgs, handlers, init_fields, invariant, actor_class=None, n_classes=1,
                fp_match_mask=0xFFFFFFFF, wide=False, array_len=0, payloads=2) -> Model:
    """msgs: list of (name, class); handlers: {(actor_class, msg name): Asm}."""
    assert 0 <= array_len <= T.MAX_ARRAY and (payloads == 2 or (wide and 3 <= payloads <= T.MAX_PAYLOADS))
    names = [m[0] for m in msgs]
    code: List[int] = []
    hs = [0xFFFF] * (n_classes * len(msgs))
    for (cls, mname), asm in handlers.items():
        hs[cls * len(msgs) + names.index(mname)] = len(code)
        code.extend(asm.finish())
    if not code:
        code = [row(OPS["HALT"])]
    if len(invariant) == 2:
        # (combining kind, Asm): DEMI_INV_PROGRAM - the per-actor predicate / key as rows after the handlers (T0 = counts,
        # T1 = key; include/demi_gpu.h)
        kind, prog = invariant
        inv_kind, fa, va, fb = kind | T.INV_PROGRAM, len(code), 0, 0
        code.extend(prog.finish())
    else:
        inv_kind, fa, va, fb = invariant
    return Model(name=name, n_actors=n_actors, msg_names=names, msg_class=[m[1] for m in msgs],
                 actor_class=list(actor_class or [0] * n_actors), n_classes=n_classes, handler_start=hs,
                 code=code, init_state=([w for f in init_fields for w in pack_state_wide(f)] if wide else
                                        [pack_state(f) for f in init_fields]),
                 inv_kind=inv_kind, inv_fa=fa, inv_va=va, inv_fb=fb, fp_match_mask=fp_match_mask, wide=wide,
                 array_len=array_len, payloads=payloads)


# --------------------------------------------------------------------------- raft-synth
# Raft-like leader election + log replication with one seeded protocol bug, standing in for
# akka-raft (NetSys/demi-applications, branches raft-45 ... raft-66), which is not in the reference.
ROLE, TERM, VOTED, VOTES, BUDGET, LOGLEN, COMMIT, BOOTED = F
FOLLOWER, CANDIDATE, LEADER = 0, 1, 2
NOBODY = 0xFF

RAFT_MSGS = [("Bootstrap", T.MSG_EXTERNAL), ("ClientCommand", T.MSG_EXTERNAL),
             ("ElectionTimeout", T.MSG_TIMER), ("RequestVote", T.MSG_INTERNAL),
             ("VoteReply", T.MSG_INTERNAL), ("AppendEntries", T.MSG_INTERNAL),
             ("AppendReply", T.MSG_INTERNAL), ("Heartbeat", T.MSG_TIMER)]
(M_BOOTSTRAP, M_CLIENT, M_ELECTION_TIMEOUT, M_REQUEST_VOTE, M_VOTE_REPLY, M_APPEND_ENTRIES,
 M_APPEND_REPLY, M_HEARTBEAT) = range(8)


RAFT_LOG_MAX = 15      # log_cap of raft_model: AppendEntries packs index (4 bits), prevLogTerm and the entry's term (6 bits each)


def raft_entry_word(idx, prev_term, term):
    """p1 of an AppendEntries that carries log entry `idx` (1-based; 0 = none) of raft_model(log_cap > 0)."""
    return idx | (prev_term << 4) | (term << 10)


def raft_model(n_actors=5, election_budget=1, buggy=True, term0=0, loglen0=0, invariant=None, log_cap=0, real_fields=False) -> Model:
    """term0 / loglen0: the term and the log length every node starts with.  Values above 255 (a cluster that has been
    running for a while) need 16-bit fields and payloads: the model is then lowered as DEMI_MODEL_WIDE, same handlers.
    invariant: None = "at most one leader per term" as a descriptor; or what build_model takes, e.g. (kind, Asm) for a
    DEMI_INV_PROGRAM invariant over the fields ROLE, TERM, ... of this module.
    log_cap > 0: the nodes keep a REAL log (DEMI_MODEL_ARRAY(log_cap): element i = the term of entry i + 1, 0 = none) instead
    of its length only - akka-raft's `replicatedLog` - and AppendEntries carries one entry with the consistency check of the
    protocol: p1 = index | prevLogTerm << 4 | entry term << 10 (raft_entry_word; a wide table).  A follower appends the entry
    (or overwrites a conflicting suffix) when the entry before it matches, else answers with a hint (0x8000 | where to retry)
    and the leader backs up, reading its own log at the computed index; an answer that shows a follower behind is followed by
    the next entry.  Elections (and the seeded bug) are as without a log.
    real_fields (with log_cap > 0): the messages carry akka-raft's own field sets instead of hand-packed values - a table with
    DEMI_MODEL_PAYLOADS(5), 9-bit fields: AppendEntries(term, prevLogIndex, prevLogTerm, entry term (0 = no entry), leaderCommit),
    RequestVote(term, candidateId, lastLogTerm, lastLogIndex) with the up-to-date rule of the protocol (a voter refuses a
    candidate whose log is behind its own), AppendReply(term, lastIndex / hint, success); a follower that appended advances
    its commit index to min(leaderCommit, the entry's index)."""
    assert 0 <= log_cap <= RAFT_LOG_MAX and not (log_cap and (term0 > 50 or loglen0)) and not (real_fields and not log_cap)
    wide = max(term0, loglen0) > 200 or log_cap > 0 or n_actors > T.MAX_ACTORS      # (more than 8 actors: the BIG layout, a wide table's)
    majority = n_actors // 2 + 1
    h = {}

    def entry_word(a, k, uniq):
        """rows: T3 = raft_entry_word(k, log[k - 2] (0 when k < 2), log[k - 1]) for the 1-based index in register k (T0..T2
        are free to use except k itself); k == 0 gives 0."""
        a.mov(T3, 0).if_ne(k, 0, "ew%s" % uniq)
        a.sub(T3, k, 1).ldx(T3, T3).shl(T3, T3, 10).or_(T3, T3, k)              # entry term << 10 | index
        a.if_ge(k, 2, "ew%s" % uniq).sub(T2, k, 2).ldx(T2, T2).shl(T2, T2, 4).or_(T3, T3, T2)
        a.label("ew%s" % uniq)

    def entry_fields(a, k, uniq):
        """rows (real_fields): T3 = prevLogIndex = k - 1, T1 = prevLogTerm = log[k - 2] (0 when k < 2), T2 = the term of
        entry k = log[k - 1], for the 1-based index in register k (not T1..T3); k == 0 (an empty log): all 0 = no entry."""
        a.mov(T1, 0).mov(T2, 0).mov(T3, 0).if_ne(k, 0, "ef%s" % uniq)
        a.sub(T3, k, 1).ldx(T2, T3)
        a.if_ge(k, 2, "ef%s" % uniq).sub(T1, k, 2).ldx(T1, T1)
        a.label("ef%s" % uniq)

    def append_entries(a, k, uniq, target=None):
        """rows: send (target register) or broadcast the AppendEntries that carries entry k"""
        if real_fields:
            entry_fields(a, k, uniq)
            if target is None:
                a.bcast(M_APPEND_ENTRIES, TERM, T3, T1, T2, COMMIT)
            else:
                a.send(M_APPEND_ENTRIES, target, TERM, T3, T1, T2, COMMIT)
        else:
            entry_word(a, k, uniq)
            if target is None:
                a.bcast(M_APPEND_ENTRIES, TERM, T3)
            else:
                a.send(M_APPEND_ENTRIES, target, TERM, T3)

    # Bootstrap (the ChangeConfiguration the DEMi raft runner Sends after Start): begin as follower.
    a = Asm()
    a.if_eq(BOOTED, 0, "done").mov(BOOTED, 1).tset(M_ELECTION_TIMEOUT).label("done")
    h[(0, "Bootstrap")] = a

    # ClientCommand: a leader appends and replicates; everyone else ignores it.
    a = Asm()
    a.if_eq(ROLE, LEADER, "done")
    if log_cap:
        a.if_lt(LOGLEN, log_cap, "done").stx(LOGLEN, TERM).add(LOGLEN, LOGLEN, 1)
        append_entries(a, LOGLEN, "c")
        a.label("done")
    else:
        a.add(LOGLEN, LOGLEN, 1).bcast(M_APPEND_ENTRIES, TERM, LOGLEN).label("done")
    h[(0, "ClientCommand")] = a

    # ElectionTimeout: start an election (bounded number per node so that executions quiesce).
    a = Asm()
    a.if_ne(ROLE, LEADER, "done")
    a.if_ne(BUDGET, 0, "done")
    a.sub(BUDGET, BUDGET, 1).mov(ROLE, CANDIDATE).add(TERM, TERM, 1).mov(VOTED, ME)
    a.bitset(VOTES, T0, ME)                                  # votes = {self}  (T0 is zero at entry)
    if real_fields:                                          # RequestVote(term, candidateId, lastLogTerm, lastLogIndex)
        a.mov(T1, 0).if_ne(LOGLEN, 0, "lt").sub(T1, LOGLEN, 1).ldx(T1, T1).label("lt")
        a.bcast(M_REQUEST_VOTE, TERM, ME, T1, LOGLEN).tset(M_ELECTION_TIMEOUT).label("done")
    else:
        a.bcast(M_REQUEST_VOTE, TERM, 0).tset(M_ELECTION_TIMEOUT).label("done")
    h[(0, "ElectionTimeout")] = a

    # RequestVote(term) from SRC, lowered as a decision tree (guards skip forward):
    #   newer term -> step down;  grant iff same term and (not voted | voted for SRC | <seeded bug>)
    a = Asm()
    a.if_gt(P0, TERM, "same")
    a.if_eq(ROLE, LEADER, "nl").tcancel(M_HEARTBEAT).label("nl")
    a.mov(TERM, P0).mov(ROLE, FOLLOWER).mov(VOTED, NOBODY)
    a.label("same")
    a.if_eq(P0, TERM, "deny")
    if real_fields:
        # the candidate's log must be at least as up to date as the voter's: a later last term, or the same and no shorter
        a.mov(T0, 0).if_ne(LOGLEN, 0, "lt").sub(T0, LOGLEN, 1).ldx(T0, T0).label("lt")
        a.ldp(T1, 2).ldp(T2, 3)
        a.if_ge(T1, T0, "deny").if_eq(T1, T0, "utd").if_ge(T2, LOGLEN, "deny").label("utd")
    a.if_ne(VOTED, NOBODY, "grant")                          # already voted for somebody ...
    a.if_ne(VOTED, SRC, "grant")                             # ... else than SRC
    if buggy:
        # seeded bug: a candidate forgets its own vote when its right-hand neighbour (id + 1)
        # asks for a vote in the same term -> two leaders can be elected in one term
        a.if_eq(ROLE, CANDIDATE, "deny").sub(T3, SRC, 1).if_eq(T3, ME, "deny")
    else:
        a.skip("deny")
    a.label("grant")
    a.mov(VOTED, SRC).tcancel(M_ELECTION_TIMEOUT).tset(M_ELECTION_TIMEOUT)
    a.send(M_VOTE_REPLY, SRC, TERM, 1).halt()
    a.label("deny").send(M_VOTE_REPLY, SRC, TERM, 0)
    h[(0, "RequestVote")] = a

    # VoteReply(term, granted) from SRC.
    a = Asm()
    a.if_gt(P0, TERM, "cur")
    a.if_eq(ROLE, LEADER, "nl").tcancel(M_HEARTBEAT).label("nl")
    a.mov(TERM, P0).mov(ROLE, FOLLOWER).mov(VOTED, NOBODY).halt()
    a.label("cur")
    a.if_eq(ROLE, CANDIDATE, "done").if_eq(P0, TERM, "done").and_(T0, P1, 1).if_ne(T0, 0, "done")
    a.bitset(VOTES, VOTES, SRC).popc(T1, VOTES).if_ge(T1, majority, "done")
    a.mov(ROLE, LEADER).tcancel(M_ELECTION_TIMEOUT).trep(M_HEARTBEAT)
    if log_cap:
        append_entries(a, LOGLEN, "v")                       # its last entry (nothing when the log is empty)
        a.label("done")
    else:
        a.bcast(M_APPEND_ENTRIES, TERM, LOGLEN).label("done")
    h[(0, "VoteReply")] = a

    # AppendEntries(term, loglen) from SRC.
    a = Asm()
    a.if_lt(P0, TERM, "ok").send(M_APPEND_REPLY, SRC, TERM, 0).halt()          # (real_fields: success = 0, P2 staged as 0)
    a.label("ok")
    a.if_gt(P0, TERM, "sameterm")                            # newer term: forget the vote, step down
    a.mov(VOTED, NOBODY)
    a.if_eq(ROLE, LEADER, "nl").tcancel(M_HEARTBEAT).label("nl")
    a.mov(ROLE, FOLLOWER).skip("keep")
    a.label("sameterm")
    a.if_ne(ROLE, LEADER, "keep").mov(ROLE, FOLLOWER)       # same term: a candidate steps down, a leader stays
    a.label("keep")
    if log_cap:
        a.mov(TERM, P0).tcancel(M_ELECTION_TIMEOUT).tset(M_ELECTION_TIMEOUT)
        if real_fields:          # T0 = the entry's index = prevLogIndex + 1 (0: no entry), T1 = prevLogTerm, T2 = entry term
            a.ldp(T2, 3).if_ne(T2, 0, "noent").add(T0, P1, 1).label("noent").ldp(T1, 2)
        else:
            a.and_(T0, P1, 15).shr(T1, P1, 4).and_(T1, T1, 63).shr(T2, P1, 10)     # T0 = index, T1 = prevLogTerm, T2 = entry term
        a.if_ne(T0, 0, "none")
        a.if_ge(T0, 2, "prevok")                                               # the entry before it: there, with that term?
        a.sub(T3, T0, 1).if_le(T3, LOGLEN, "nack").sub(T3, T0, 2).ldx(T3, T3).if_eq(T3, T1, "nack")
        a.label("prevok")
        a.sub(T3, T0, 1).mov(T1, 0).if_le(T0, LOGLEN, "cmp").ldx(T1, T3)       # T1 = what it holds at that index (0 past the end)
        a.label("cmp").if_ne(T1, T2, "have").stx(T3, T2).mov(LOGLEN, T0)       # append, or overwrite and cut a conflicting suffix
        if real_fields:
            a.label("have").ldp(T3, 4).min(T3, T3, T0).max(COMMIT, COMMIT, T3)     # commitIndex = min(leaderCommit, this entry)
            a.send(M_APPEND_REPLY, SRC, TERM, T0, 1).halt()
            a.label("nack").sub(T3, T0, 2).min(T3, T3, LOGLEN).send(M_APPEND_REPLY, SRC, TERM, T3, 0).halt()
            a.label("none").mov(T3, 0).send(M_APPEND_REPLY, SRC, TERM, T3, 1)
        else:
            a.label("have").send(M_APPEND_REPLY, SRC, TERM, T0).halt()
            # (the hint: one back, or the end of a shorter log)
            a.label("nack").sub(T3, T0, 2).min(T3, T3, LOGLEN).movhi(T3, T3, 0x80).send(M_APPEND_REPLY, SRC, TERM, T3).halt()
            a.label("none").mov(T3, 0).send(M_APPEND_REPLY, SRC, TERM, T3)
    else:
        a.mov(TERM, P0).max(LOGLEN, LOGLEN, P1)
        a.tcancel(M_ELECTION_TIMEOUT).tset(M_ELECTION_TIMEOUT)
        a.send(M_APPEND_REPLY, SRC, TERM, P1)
    h[(0, "AppendEntries")] = a

    # AppendReply(term, acked) from SRC.
    a = Asm()
    a.if_gt(P0, TERM, "cur")
    a.if_eq(ROLE, LEADER, "nl").tcancel(M_HEARTBEAT).label("nl")
    a.mov(TERM, P0).mov(ROLE, FOLLOWER).mov(VOTED, NOBODY).halt()
    a.label("cur")
    a.if_eq(ROLE, LEADER, "done").if_eq(P0, TERM, "done")
    if log_cap:
        if real_fields:
            a.mov(T0, P1).ldp(T1, 2).if_ne(T1, 0, "hint")                         # AppendReply(term, index, success): matched up to T0
        else:
            a.and_(T0, P1, 255).shr(T1, P1, 15).if_eq(T1, 0, "hint")              # matched up to T0
        a.if_le(T0, LOGLEN, "done").max(COMMIT, COMMIT, T0).skip("next")
        a.label("hint")                                                        # refused: retry right after the hinted index
        a.label("next").if_lt(T0, LOGLEN, "done").add(T0, T0, 1)
        append_entries(a, T0, "r", target=SRC)
        a.label("done")
    else:
        a.max(COMMIT, COMMIT, P1).label("done")
    h[(0, "AppendReply")] = a

    # Heartbeat (repeating timer): a leader re-sends uncommitted entries.
    a = Asm()
    a.if_ne(ROLE, LEADER, "lead").tcancel(M_HEARTBEAT).halt()
    a.label("lead").if_gt(LOGLEN, COMMIT, "done")
    if log_cap:
        append_entries(a, LOGLEN, "h")
        a.label("done")
    else:
        a.bcast(M_APPEND_ENTRIES, TERM, LOGLEN).label("done")
    h[(0, "Heartbeat")] = a

    # election_budget: one number, or one per node (the rows do not change, only the nodes' initial BUDGET field: a node with
    # budget 0 votes and replicates but never campaigns)
    budgets = list(election_budget) if isinstance(election_budget, (list, tuple)) else [election_budget] * n_actors
    assert len(budgets) == n_actors
    init = [[FOLLOWER, term0, NOBODY, 0, b, loglen0, loglen0, 0] for b in budgets]
    return build_model("raft%d-synth%s%s%s%s%s" % (n_actors, "" if buggy else "-fixed", "-wide" if wide else "", "-log%d" % log_cap if log_cap else "",
                                                   "-fields" if real_fields else "",
                                                   "-eb" + "".join(str(b) for b in budgets) if len(set(budgets)) > 1 else ""),
                       n_actors, RAFT_MSGS, h, init, invariant=invariant or (T.INV_AT_MOST_ONE, int(ROLE), LEADER, int(TERM)), wide=wide,
                       array_len=log_cap, payloads=5 if real_fields else 2)


def save_model(model: Model, path: str):
    with open(path, "w") as f:
        json.dump(model.to_json(), f)


def load_model(path: str) -> Model:
    with open(path) as f:
        return Model.from_json(json.load(f))


# --------------------------------------------------------------------------- shuffle-synth
# A 2-stage shuffle job standing in for the Spark application of NetSys/demi-applications
# (branches spark-2294/3150/9256), which is not in the reference: one driver, two stage
# coordinators, five workers, three actor classes.  Seeded bug: the driver starts stage 2 as soon
# as it has counted `n_workers` MapDone reports, and counts a duplicate report from a re-launched
# task twice, so stage 2 can start while a map output is still missing.
SH_MSGS = [("Submit", T.MSG_EXTERNAL), ("Speculate", T.MSG_EXTERNAL), ("LaunchStage", T.MSG_INTERNAL),
           ("RunTask", T.MSG_INTERNAL), ("MapDone", T.MSG_INTERNAL), ("StageDone", T.MSG_INTERNAL),
           ("Fetch", T.MSG_INTERNAL), ("FetchReply", T.MSG_INTERNAL), ("TaskTimeout", T.MSG_TIMER)]
(SH_SUBMIT, SH_SPECULATE, SH_LAUNCH, SH_RUN, SH_MAPDONE, SH_STAGEDONE, SH_FETCH, SH_FETCHREPLY, SH_TIMEOUT) = range(9)
SH_REDUCEDONE = 9      # the pipeline variant only (shuffle_model(jobs > 1))
CLS_DRIVER, CLS_COORD, CLS_WORKER = 0, 1, 2


def shuffle_model(buggy=True, jobs=1, early_cleanup=False, n_workers=5) -> Model:
    """Actors: 0 driver, 1 map-stage coordinator, 2 reduce-stage coordinator, 3..7 workers (n_workers = 5: the table of every
    fixture; up to 13 workers = 16 actors - more than 8 actors is the BIG layout of include/demi_gpu.h, a wide table).

    jobs > 1 is the PIPELINE variant (BASELINE config 5 at a size worth sharding, apps.shuffle8_config5_large): the reduce stage
    reports back - a worker that has all its FetchReplies sends ReduceDone, the reduce coordinator counts them and reports
    StageDone(2) - and the driver then launches the next job itself, `jobs` of them back to back.  The next job's messages
    descend from the delivery that completed the previous one, so its whole subtree of DPOR nodes is new for every way the
    previous job can end: the space of racing pairs multiplies per job instead of adding up.  jobs = 1 is the table every
    fixture of rounds 1-3 was taken on, row for row.
    early_cleanup (pipeline only): the pipeline's second seeded bug - reducer 7, once it has all its FetchReplies, frees its own
    map output at once although other reducers may not have fetched it yet (shuffle files removed too early); their Fetch then
    finds nothing and the invariant's sticky flag F2 is raised.  Unlike the duplicate MapDone (which needs the straggler
    detector of the FIRST job and so sits at the shallow end of a depth-first exploration) this race lives in every job's
    reduce phase: a bounded DPOR search meets it within its first few hundred interleavings (round 6)."""
    n_actors = 3 + n_workers
    pipeline = jobs > 1
    early_cleanup = bool(early_cleanup and buggy and pipeline)
    assert 1 <= jobs <= 255 and 1 <= n_workers <= T.MAX_ACTORS_BIG - 3
    h = {}
    # ---- driver: F0 phase (0 idle, 1 map, 2 reduce, 3 done), F1 reports counted, F2 bitmask of workers reported, F6 job number
    a = Asm()
    a.if_eq(F[0], 0, "x").mov(F[0], 1).mov(T0, 1).send(SH_LAUNCH, T0, T1, 0).label("x")
    h[(CLS_DRIVER, "Submit")] = a
    a = Asm()          # Speculate(w): ask the map coordinator to re-launch worker w's task
    a.if_eq(F[0], 1, "x").mov(T0, 1).send(SH_LAUNCH, T0, P0, 1).label("x")
    h[(CLS_DRIVER, "Speculate")] = a
    a = Asm()          # StageDone(stage) from a coordinator
    a.if_eq(P0, 1, "s2")
    a.if_eq(F[0], 1, "x").mov(F[0], 2).mov(T0, 2).send(SH_LAUNCH, T0, T1, 0).halt()
    a.label("s2").if_eq(F[0], 2, "x").mov(F[0], 3)
    if pipeline:       # the next job: LaunchStage(job number) to the map coordinator
        a.if_lt(F[6], jobs - 1, "x").add(F[6], F[6], 1).mov(F[0], 1).mov(T0, 1).send(SH_LAUNCH, T0, F[6], 0)
    a.label("x")
    h[(CLS_DRIVER, "StageDone")] = a
    # ---- stage coordinators (actors 1 and 2 share a class): F0 started, F1 done count, F2 done mask, F3 reported, F4 job number
    a = Asm()          # LaunchStage(w, relaunch)
    a.if_eq(ME, 2, "map")            # on the reduce coordinator: RunTask(reduce) to every worker
    if pipeline:
        a.mov(F[1], 0).mov(F[3], 0)
    if n_workers == 5:
        for w in range(n_workers):
            a.mov(T0, 3 + w).mov(T1, 2).send(SH_RUN, T0, T1, 0)
    else:              # (one effect row instead of n_workers: a delivery has DEMI_FX_CAP of them; driver and coordinators ignore RunTask)
        a.mov(T1, 2).bcast(SH_RUN, T1, 0)
    a.halt()
    a.label("map")                   # on the map coordinator: relaunch=0 -> all workers; relaunch=1 -> worker 3+w only
    a.if_eq(P1, 0, "re")
    if pipeline:                     # (a new job number: the coordinator starts over)
        a.if_ne(P0, F[4], "same").mov(F[4], P0).mov(F[0], 0).mov(F[1], 0).mov(F[2], 0).mov(F[3], 0).label("same")
    a.if_eq(F[0], 0, "x").mov(F[0], 1)
    if n_workers == 5:
        for w in range(n_workers):
            a.mov(T0, 3 + w).mov(T1, 1).send(SH_RUN, T0, T1, 0)
    else:
        a.mov(T1, 1).bcast(SH_RUN, T1, 0)
    a.halt()
    a.label("re").lt(T2, P0, n_workers).if_ne(T2, 0, "x").add(T0, P0, 3).mov(T1, 1).send(SH_RUN, T0, T1, 0).label("x")
    h[(CLS_COORD, "LaunchStage")] = a
    a = Asm()          # MapDone from worker SRC
    a.sub(T0, SRC, 3).mov(T1, 1).shl(T1, T1, T0)            # bit of the reporting worker
    if buggy:
        a.add(F[1], F[1], 1).or_(F[2], F[2], T1)            # bug: duplicates are counted
    else:
        a.and_(T2, F[2], T1).if_eq(T2, 0, "x").add(F[1], F[1], 1).or_(F[2], F[2], T1)
    a.if_eq(F[1], n_workers, "x").if_eq(F[3], 0, "x").mov(F[3], 1).mov(T0, 0).mov(T1, 1).send(SH_STAGEDONE, T0, T1, 0).label("x")
    h[(CLS_COORD, "MapDone")] = a
    if pipeline:
        a = Asm()      # ReduceDone from a worker (at the reduce coordinator): the fifth one ends the stage
        a.add(F[1], F[1], 1).if_eq(F[1], n_workers, "x").if_eq(F[3], 0, "x").mov(F[3], 1).mov(T0, 0).mov(T1, 2).send(SH_STAGEDONE, T0, T1, 0).label("x")
        h[(CLS_COORD, "ReduceDone")] = a
    # ---- workers: F0 has map output, F1 fetched count, F2 fetch-missing flag (violation), F3 ran reduce
    a = Asm()          # RunTask(kind): 1 = map (arm a task timeout, produce output, report); 2 = reduce (fetch from all)
    a.if_eq(P0, 1, "red").mov(F[0], 1)
    if pipeline:
        a.mov(F[1], 0).mov(F[3], 0)
    a.tcancel(SH_TIMEOUT).tset(SH_TIMEOUT).mov(T0, 1).send(SH_MAPDONE, T0, T1, 0).halt()
    a.label("red").if_eq(F[3], 0, "x").mov(F[3], 1).bcast(SH_FETCH, T1, 0).label("x")
    h[(CLS_WORKER, "RunTask")] = a
    a = Asm()          # Fetch from a reducer: reply with whether the map output exists
    a.send(SH_FETCHREPLY, SRC, F[0], 0)
    h[(CLS_WORKER, "Fetch")] = a
    a = Asm()          # FetchReply(has_output)
    a.add(F[1], F[1], 1).if_eq(P0, 0, "y").mov(F[2], 1).label("y")
    if pipeline:       # the last of the n_workers - 1 replies: this reducer is done
        a.if_eq(F[1], n_workers - 1, "x")
        if early_cleanup:
            a.if_eq(ME, n_actors - 1, "keep").mov(F[0], 0).label("keep")
        a.mov(T0, 2).send(SH_REDUCEDONE, T0, T1, 0).label("x")
    h[(CLS_WORKER, "FetchReply")] = a
    a = Asm()          # TaskTimeout: a straggler detector re-reports (duplicate MapDone)
    a.if_eq(F[0], 1, "x").if_lt(F[4], 1, "x").add(F[4], F[4], 1).mov(T0, 1).send(SH_MAPDONE, T0, T1, 0).label("x")
    h[(CLS_WORKER, "TaskTimeout")] = a
    # coordinators ignore worker-only messages and vice versa (handler_start 0xFFFF)
    init = [[0] * 8 for _ in range(n_actors)]
    return build_model("shuffle%d-synth%s%s%s" % (n_actors, "" if buggy else "-fixed", "-x%d" % jobs if pipeline else "", "-c" if early_cleanup else ""), n_actors,
                       SH_MSGS + ([("ReduceDone", T.MSG_INTERNAL)] if pipeline else []), h, init,
                       invariant=(T.INV_NEVER, 2, 1, 0), actor_class=[CLS_DRIVER, CLS_COORD, CLS_COORD] + [CLS_WORKER] * n_workers,
                       n_classes=3, wide=n_actors > T.MAX_ACTORS)


# --------------------------------------------------------------------------- replicated log (DEMI_MODEL_ARRAY)
# Primary-backup log replication with the LOG ITSELF in the actors' state - what raft-synth above abstracts to a length
# (akka-raft keeps `replicatedLog: Vector[Entry]`; eight fields cannot).  The array of every actor is its log.  Whoever
# receives a client's Put appends the value and broadcasts Append(index, value); a backup appends the entry that comes next,
# ignores one it already has, and answers with its length either way; the primary answers an Ack that shows a backup behind
# with the entry that backup is missing (read from its own log at a computed index: LDX).  Messages overtake each other under
# the random schedulers, so gaps are the normal case.  The seeded bug: a backup that receives an entry beyond its length stores
# it where it belongs anyway and advances its length past the gap - a log with a hole.  Invariant (a DEMI_INV_PROGRAM that
# reads the array): no actor has an empty slot (value 0; clients never Put 0) below its length.
REPLOG_MSGS = [("Put", T.MSG_EXTERNAL), ("Append", T.MSG_INTERNAL), ("Ack", T.MSG_INTERNAL)]
RL_PUT, RL_APPEND, RL_ACK = range(3)
RL_LEN = F[0]


def replog_model(n_actors=3, log_len=6, buggy=True, wide=False) -> Model:
    assert 1 <= log_len <= T.MAX_ARRAY
    put = Asm().if_lt(RL_LEN, log_len, "x").stx(RL_LEN, P0).mov(T0, RL_LEN).add(RL_LEN, RL_LEN, 1).bcast(RL_APPEND, T0, P0).label("x")
    ap = Asm().if_eq(P0, RL_LEN, "other").stx(P0, P1).add(RL_LEN, RL_LEN, 1).send(RL_ACK, SRC, RL_LEN, 0).halt()
    ap.label("other").if_gt(P0, RL_LEN, "x")                     # beyond the end: a gap
    if buggy:
        ap.stx(P0, P1).add(RL_LEN, P0, 1)
    ap.send(RL_ACK, SRC, RL_LEN, 0).label("x")
    ack = Asm().if_lt(P0, RL_LEN, "x").ldx(T0, P0).send(RL_APPEND, SRC, P0, T0).label("x")
    inv = Asm()
    for i in range(log_len):
        inv.if_gt(RL_LEN, i, "n%d" % i).ldx(T1, i).if_eq(T1, 0, "n%d" % i).mov(T0, 1).label("n%d" % i)
    h = {(0, "Put"): put, (0, "Append"): ap, (0, "Ack"): ack}
    return build_model("replog%d-%d%s%s" % (n_actors, log_len, "" if buggy else "-fixed", "-wide" if wide else ""), n_actors,
                       REPLOG_MSGS, h, [[0] * 8] * n_actors, invariant=(T.INV_NEVER, inv), wide=wide, array_len=log_len)
