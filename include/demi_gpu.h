/*
 * demi_gpu.h — C ABI of libdemi_gpu.so, the MI355X schedule-exploration engine for DEMi.
 *
 * The reference (NetSys/demi) has no FFI: its plugin boundary is the Scala traits
 *   trait Scheduler   (src/main/scala/verification/schedulers/Scheduler.scala:13-104)
 *   trait TestOracle  (src/main/scala/verification/minification/TestOracle.scala:30-55)
 * driven by RunnerUtils (src/main/scala/verification/RunnerUtils.scala:62-147, 601-707, 881-911).
 * A Scala adapter class `GpuRandomScheduler extends Scheduler with TestOracle`
 * lowers the closures of that boundary (Props, message constructors, Invariant) to the flat
 * tables below and calls these entry points through a 1:1 JNI shim.  Every entry point names the
 * reference method(s) it replaces.
 *
 * Conventions: plain C, caller-allocated buffers, `int` return (0 = ok, <0 = demi_status),
 * no callbacks into the host from device code (the transition function and the invariant are
 * data, not closures), one ctx per host thread, the ctx owns its device memory.
 * The *_dev entry points enqueue on the caller's stream and return without synchronising; a ctx runs ONE launch at a
 * time (its work counter and scratch are shared): a launch on a different stream than the ctx's previous one waits for
 * that one on the device, and demi_model_load / demi_trace_load / demi_replay_load wait for the last launch before they
 * replace what the kernels read.  The exception is the RandomScheduler kernel (demi_random_explore_dev and everything
 * built on it): a ctx holds TWO sets of its per-launch scratch and alternates between them, so two of its launches on
 * two streams overlap - the tail of one with the start of the next (demi_random_explore_submit / _wait do that with
 * two streams of the ctx's own, for hosts that have no HIP streams).  Other concurrent launches need one ctx each.
 * Pointers named d_* are DEVICE pointers; all others are host pointers.
 *
 * Environment.  The library reads exactly two environment variables on its own account:
 *   DEMI_HIPRTC_LIB   the hiprtc library demi_model_specialize dlopens (default: the one next to the process's HIP runtime);
 *   DEMI_RCCL_LIB     the RCCL library demi_comm_create dlopens (default: librccl.so next to the HIP runtime, then the loader path).
 * Every other DEMI_* variable of the source tree (kernel-variant, launch-shape and bookkeeping selectors such as DEMI_K2_MODE,
 * DEMI_JIT_K1_HOT, DEMI_DPOR_HOST_BOOKKEEPING; diagnostics such as DEMI_K1_VERBOSE, DEMI_DPOR_TIMING, DEMI_JIT_DUMP) is an
 * experiment knob of the test suites and the profiling scripts and is IGNORED unless DEMI_EXPERIMENT=1 is set as well
 * (demi_amd/csrc/knobs.hpp): a host process that inherits a stray one runs the default engine.
 */
#ifndef DEMI_GPU_H
#define DEMI_GPU_H

#ifndef __HIPCC_RTC__   /* the run-time compiler (demi_model_specialize) supplies the fixed-width types */
#include <stdint.h>
#include <stddef.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ limits */
#define DEMI_MAX_ACTORS      8      /* actor ids 0..7: the layout every table of up to 8 actors uses */
#define DEMI_DEADLETTERS     15     /* sender id of externals and timers ("deadLetters") */
#define DEMI_MAX_ACTORS_BIG  16     /* actor ids 0..15: the BIG layout of a table with more than 8 actors (below) */
#define DEMI_DEADLETTERS_BIG 31     /* its deadLetters id (r14 of a handler run for an external message or a timer) */
#define DEMI_MAX_MSG_TYPES   32
#define DEMI_MAX_CLASSES     4
#define DEMI_MAX_CODE        1024   /* delta-table rows */
#define DEMI_MAX_TIMER_TYPES 4      /* timer-class message types per model */
#define DEMI_MAX_EXT_EVENTS  255    /* external events per trace */
#define DEMI_TQ_CAP          8      /* messagesToSend timers between two scheduling steps */
#define DEMI_RESEND_CAP      8      /* timersToResend */
#define DEMI_FX_CAP          8      /* effect rows (SEND/BCAST/TSET/TREP/TCANCEL) executed per delivery */
#define DEMI_MAX_REC_EVENTS  16384  /* recorded events of one execution */
#define DEMI_MAX_PENDING     128    /* largest p_max */

/* ------------------------------------------------------------------ status */
typedef enum {
  DEMI_OK = 0,
  DEMI_ERR_INVALID_ARG = -1,
  DEMI_ERR_INVALID_MODEL = -2,   /* model rejected by validation (see demi_last_error) */
  DEMI_ERR_INVALID_TRACE = -3,   /* e.g. WaitCondition / CodeBlock / HardKill: JVM fallback */
  DEMI_ERR_NO_MODEL = -4,        /* IllegalArgumentException("Must invoke setInvariant...") analogue */
  DEMI_ERR_NO_TRACE = -5,
  DEMI_ERR_DEVICE = -6,          /* HIP runtime error */
  DEMI_ERR_CAPACITY = -7         /* caller buffer too small */
} demi_status;

/* ------------------------------------------------------ external events
 * ExternalEvent ADT, src/main/scala/verification/ExternalEvents.scala:62-91.          */
typedef enum {
  DEMI_EV_START = 0,            /* Start(propCtor, name)       a = actor            */
  DEMI_EV_KILL = 1,             /* Kill(name)                  a = actor            */
  DEMI_EV_SEND = 2,             /* Send(name, messageCtor)     a = receiver, msg    */
  DEMI_EV_PARTITION = 3,        /* Partition(a, b)                                  */
  DEMI_EV_UNPARTITION = 4,      /* UnPartition(a, b)                                */
  DEMI_EV_WAIT_QUIESCENCE = 5   /* WaitQuiescence()                                 */
  /* WaitCondition, CodeBlock, HardKill are closures / real actor stops: rejected.   */
} demi_ext_kind;

typedef struct {
  uint8_t kind;       /* demi_ext_kind */
  uint8_t a, b;       /* actors */
  uint8_t msg_type;   /* SEND: message type (class EXTERNAL) */
  uint8_t p0, p1;     /* SEND: payload (low bytes) */
  uint8_t p0_hi, p1_hi; /* SEND, DEMI_MODEL_WIDE only: bits 8..15 of the payload fields; must be 0 for other models */
} demi_ext_event;     /* 8 bytes */

/* ------------------------------------------------------ transition table
 * The application's actors (NOT in the reference: NetSys/demi-applications) lowered to
 * a table: one micro-program ("handler") per (actor class, message type), each row one
 * guarded micro-op over a 16 x u8 register window:
 *   r0..r7  = the receiving actor's state fields F0..F7 (persisted across deliveries)
 *   r8..r11 = temporaries T0..T3 (zero at handler entry)
 *   r12,r13 = message payload P0,P1     r14 = sender id (15 = deadLetters; 31 in the BIG layout)   r15 = own id
 * Row word: op[7:0] | dst[11:8] | a[15:12] | bimm[16] | aux[23:17] | b[31:24]
 *   (bimm=1: b is an 8-bit immediate, else b[3:0] is a register).  All arithmetic mod 256,
 *   comparisons unsigned.  Control flow is forward-only (SKIP*), so a handler terminates.   */
typedef enum {
  DEMI_OP_HALT = 0,
  DEMI_OP_MOV = 1,  DEMI_OP_ADD = 2,  DEMI_OP_SUB = 3,  DEMI_OP_AND = 4,  DEMI_OP_OR = 5,
  DEMI_OP_XOR = 6,  DEMI_OP_SHL = 7,  DEMI_OP_SHR = 8,  DEMI_OP_BITSET = 9, /* dst = a | 1<<(b&7) */
  DEMI_OP_POPC = 10, /* dst = popcount(b) */
  DEMI_OP_EQ = 11,  DEMI_OP_NE = 12,  DEMI_OP_LT = 13,  DEMI_OP_GE = 14,  DEMI_OP_LE = 15,
  DEMI_OP_GT = 16,  DEMI_OP_MIN = 17, DEMI_OP_MAX = 18,
  DEMI_OP_MOVHI = 19,   /* DEMI_MODEL_WIDE only: dst = (a & 0xFF) | (b << 8), b an 8-bit immediate: the upper half of a
                           16-bit constant (MOV loads the lower half)                                               */
  DEMI_OP_SKIPZ = 20,   /* if reg a == 0 skip the next b rows  (b immediate) */
  DEMI_OP_SKIPNZ = 21,  /* if reg a != 0 skip the next b rows */
  DEMI_OP_SKIP = 22,    /* skip the next b rows */
  /* fused guard: if !(reg a OP b) skip the next `aux` rows */
  DEMI_OP_IFEQ = 32, DEMI_OP_IFNE = 33, DEMI_OP_IFLT = 34, DEMI_OP_IFGE = 35, DEMI_OP_IFLE = 36, DEMI_OP_IFGT = 37,
  DEMI_OP_SEND = 24,    /* `target ! msg`: type = aux, target = reg a, p0 = reg dst, p1 = b    */
  DEMI_OP_BCAST = 25,   /* SEND to every other created actor, ascending id: p0 = reg dst, p1 = b */
  DEMI_OP_TSET = 26,    /* scheduler.scheduleOnce(self, msg type aux)                          */
  DEMI_OP_TREP = 27,    /* scheduler.schedule(self, msg type aux)  (repeating)                 */
  DEMI_OP_TCANCEL = 28, /* cancellable.cancel() of timer (self, type aux)                      */
  DEMI_OP_CRASH = 29,   /* the receive throws: Instrumenter.actorCrashed (Instrumenter.scala:184-199).  The handler
                           stops here (state changes and effects so far stand) and the actor joins blockedActors: the
                           schedulers no longer deliver to it (Util.find_non_blocked_message, Util.scala:470-489: a
                           drawn message for a blocked receiver is set aside, the draw repeated, and the rejected
                           ones re-appended in draw order) until a Start() of the same name (EventOrchestrator
                           trigger_start :219-231).  Counts as an effect row.                                      */
  DEMI_OP_RND = 30      /* dst = Instrumenter().seededRandom.nextInt(b), b = bound 1..255 (0: dst = 0, nothing drawn).
                           Applications draw their randomness (akka-raft: election timeouts) from this generator, a
                           scala.util.Random(0) - the same JDK LCG - recreated with every ActorSystem, i.e. a second
                           deterministic stream that restarts at seed 0 in every execution (Instrumenter.scala:212,
                           226-229, 570).                                                                          */,
  /* Indexed state (DEMI_MODEL_ARRAY below): the actor's array - a log, a vote table - beside its eight fields. */
  DEMI_OP_LDX = 38,     /* dst = ARRAY[b]  (b register or immediate; an index >= the array's length reads 0)         */
  DEMI_OP_STX = 39,     /* ARRAY[b] = a    (an index >= the array's length stores nothing)                             */
  /* DEMI_INV_PROGRAM rows only: ANOTHER actor's state, so that an invariant can relate two actors by more than equal keys. */
  DEMI_OP_PEER = 40,    /* dst = field F[aux] (aux 0..7) of the actor whose id is in reg a; aux = 8: dst = 1 if that actor is
                           created, else 0.  An id that is not a created actor reads 0.                                 */
  /* Messages with more than two payload fields (DEMI_MODEL_PAYLOADS below). */
  DEMI_OP_LDP = 41,     /* dst = payload field b of the message being handled (b an immediate 0..5; a field the model's
                           messages do not have reads 0).  P0 / P1 are also r12 / r13 as always.  A pure row.           */
  DEMI_OP_PSET = 42     /* outgoing payload field aux (2..5) = b (register or immediate): SEND / BCAST rows take P0 / P1 from
                           their operands and P2..P5 from these staging values - 0 at handler entry, kept until
                           overwritten (several SENDs may share them).  Not an effect row.                             */
} demi_op;

#define DEMI_ROW(op, dst, a, bimm, aux, b) \
  ((uint32_t)(op) | ((uint32_t)(dst) << 8) | ((uint32_t)(a) << 12) | ((uint32_t)(bimm) << 16) | \
   ((uint32_t)(aux) << 17) | ((uint32_t)(b) << 24))

typedef enum { DEMI_MSG_INTERNAL = 0, DEMI_MSG_EXTERNAL = 1, DEMI_MSG_TIMER = 2 } demi_msg_class;

/* Invariant (TestOracle.scala:9-28 `Invariant`) as a descriptor evaluated on simulated state. */
typedef enum {
  DEMI_INV_NONE = 0,
  DEMI_INV_AT_MOST_ONE = 1, /* no two created actors with F[fa]==va and equal F[fb]               */
  DEMI_INV_NEVER = 2,       /* no created actor with F[fa]==va                                     */
  DEMI_INV_AGREE = 3        /* created actors with F[fa]!=0 agree on F[fb]                         */
} demi_inv_kind;
/* DEMI_INV_PROGRAM, OR-ed into inv_kind: the per-actor part of the invariant - "does this actor count" and "under which key" -
 * is a ROW PROGRAM instead of the two field tests.  inv_fa = its first row in `code` (inv_va, inv_fb: 0).  It is run on every
 * created actor with the register window r0..r7 = that actor's F0..F7, r15 = its id, everything else 0, and ends (HALT, or the
 * end of the table) with T0 (r8) != 0 = the actor counts ("hit"), T1 (r9) = its key; what it writes to r0..r7 is discarded.
 * Only ALU, SKIP* and IF* rows (no effects, no RND).  inv_kind & 0xFF combines the actors as before: NEVER = some actor
 * counts; AT_MOST_ONE = two counting actors have equal keys; AGREE = the counting actors' keys differ - with the same
 * fingerprint layouts.  So an arbitrary predicate over one actor's state (ranges, bit tests, several fields, its id) and a
 * computed key replace `F[fa] == va` / `F[fb]`.  A program may also read the fields of the OTHER actors (DEMI_OP_PEER, actor id
 * from a register or unrolled over the ids): "this actor is a leader and some created actor has a larger term", "my commit index
 * exceeds what a majority has logged" - a relation between two (or all) actors, evaluated from each actor's point of view and
 * combined as before.  (The hit of one actor then depends on the others' states: the kernels re-evaluate every actor at a
 * check instead of only the receiver of the last delivery.)  What still needs the JVM: invariants over state the table does not
 * model. */
#define DEMI_INV_PROGRAM 0x100u

typedef struct {
  uint32_t n_actors;        /* 1..DEMI_MAX_ACTORS; 9..DEMI_MAX_ACTORS_BIG with DEMI_MODEL_WIDE: the BIG layout (below) */
  uint32_t n_msg_types;     /* 1..DEMI_MAX_MSG_TYPES */
  uint32_t n_classes;       /* 1..DEMI_MAX_CLASSES */
  uint32_t code_len;        /* rows in `code` */
  const uint8_t*  msg_class;     /* [n_msg_types] demi_msg_class */
  const uint8_t*  actor_class;   /* [n_actors] */
  const uint16_t* handler_start; /* [n_classes * n_msg_types] row index, 0xFFFF = message ignored */
  const uint32_t* code;          /* [code_len] */
  const uint64_t* init_state;    /* [n_actors] F0 in bits 0..7 ... F7 in bits 56..63 */
  uint32_t inv_kind, inv_fa, inv_va, inv_fb;
  uint32_t fp_match_mask;   /* ViolationFingerprint.matches: ((x ^ y) & mask) == 0 */
  uint32_t flags;           /* DEMI_MODEL_*; occupies what used to be the struct's tail padding (sizeof is unchanged) */
} demi_model;

/* DEMI_MODEL_WIDE: the register window is 16 x u16 instead of 16 x u8 - state fields F0..F7, temporaries and the two
 * payload fields are 16 bits wide, all arithmetic is mod 65536, shifts / BITSET use b & 15, POPC counts 16 bits, sender
 * and own id are unchanged - so that a protocol whose terms and log indices exceed 255 lowers without multi-byte
 * arithmetic.  What changes at the boundary:
 *   init_state       [2 * n_actors]: word 2i holds F0..F3 of actor i (16 bits each, F0 lowest), word 2i + 1 holds F4..F7;
 *   row immediates   stay 8 bits; DEMI_OP_MOVHI supplies the upper half of a constant;
 *   DEMI_OP_RND      the bound is b & 0xFF;
 *   external Sends   carry 16-bit payloads (demi_ext_event.p0_hi / p1_hi);
 *   inv_va           up to 65535; the fingerprint of DEMI_INV_AT_MOST_ONE keeps the layout 1 << 24 | key << 8 | mask with a
 *                    16-bit key;
 *   the message word hashed into demi_verdict.hash is 64 bits: type[4:0] | dst[7:5] | src[11:8] | p0[31:16] | p1[47:32], and
 *                    an actor's final state enters the hash as its two words.
 *   demi_rec_event   carries the 16-bit payloads (p0, p1 are uint16_t for every model);
 *   demi_dpor_trace_entry.word  is the LOW half of the 64-bit message word (type, dst, src, p0): enough to build next traces
 *                    from (only key, word and kind of a prefix entry are read, and the node key hashes the whole word);
 *                    p1 is not reported there.
 * A wide model runs only as a compiled table (demi_model_specialize must succeed: there is no interpreter for it; the kernels
 * are compiled for it at their first launch and a failure is an error, never a fallback): the RandomScheduler entry points
 * (both strategies, independent or carried generators, incl. demi_random_get_trace: every variant of the kernel is compiled
 * for the table), the replay entry points (the pending-set scan variant of the kernel) and the DPOR entry points.  The 8-bit
 * layout is untouched by the option (same code, same verdict hashes as before). */
#define DEMI_MODEL_WIDE 0x1u

/* MORE THAN 8 ACTORS - the BIG layout (round 6).  The reference puts no bound on actor names (ExternalEvents.scala:62-91,
 * EventOrchestrator.scala:203-217, 345-351); a table with n_actors = 9 .. DEMI_MAX_ACTORS_BIG (16) is accepted when it is also
 * DEMI_MODEL_WIDE (it runs only as compiled code, like every wide table), and everything that names an actor is laid out for 16:
 *   message word      type[4:0] | dst[8:5] | src[13:9] | payload area[63:16]  (4-bit receiver, 5-bit sender; tables of up to
 *                     8 actors keep dst[7:5] | src[11:8]).  demi_verdict.hash hashes these words, demi_dpor_trace_entry.word
 *                     is their low half;
 *   deadLetters       DEMI_DEADLETTERS_BIG (31): r14 of a handler run for an external message or a timer, demi_rec_event.snd of
 *                     such a message (15 stays what it is for tables of up to 8 actors - there it could be an actor's id here);
 *   SEND rows         a target register above 15 addresses nobody, as a target above 7 does in the small layout;
 *   fingerprints      kind[31:30] | key[29:16] | actors[15:0]: DEMI_INV_AT_MOST_ONE = 1 << 30 | (key & 0x3FFF) << 16 | mask of the
 *                     actors that share the key, DEMI_INV_NEVER = 2 << 30 | hit actors, DEMI_INV_AGREE = 3 << 30 | hit actors (the
 *                     small layout: kind << 24 | key << 8 | 8-bit mask).  A key wider than 14 bits is truncated IN THE
 *                     FINGERPRINT only (the comparison of keys is exact); fp_match_mask applies to this word;
 *   inside the engine 16 x 16 partition / reach matrices, 64 timer bits (actor x DEMI_MAX_TIMER_TYPES), 256 (src, dst) pairs of
 *                     SrcDstFIFO - none of it visible at the boundary.
 * Every entry point takes such a table - the RandomScheduler kernel in all its variants (recording, SrcDstFIFO, carried generators,
 * candidate frontiers: demi_random_ddmin), the replay kernel with demi_ddmin and the internal-event minimization, DPORwHeuristics
 * in both orders with the device-resident queue and the checkpoints, demi_provenance_prune (which reads its traces in the layout
 * of the table the context holds).  Tables of up to 8 actors are untouched: same layout, same
 * verdict hashes, the same instructions in their kernels (the message word of the small layout has no room for a second
 * layout's fields, so the big one is a layout of its own rather than a widening of the old).  tests/test_big_gpu.py. */

/* DEMI_MODEL_ARRAY(n): every actor owns, beside its eight state fields, an ARRAY of n elements (1..DEMI_MAX_ARRAY) of the
 * register window's width (u8, or u16 with DEMI_MODEL_WIDE) - the part of an actor's state that eight fields cannot hold: a
 * replicated log (akka-raft's `replicatedLog`), a table of votes, a mailbox of deferred requests.  Rows reach it through
 * DEMI_OP_LDX / DEMI_OP_STX with a computed index; the arrays start all-zero (a log starts empty; `init_state` keeps its
 * layout and holds the fields only); they are part of the actor's state in every respect: persisted across deliveries,
 * hashed into demi_verdict.hash after the actor's field word(s) - element 0 in the lowest bits of the first array word, 8 (wide:
 * 4) elements per 64-bit word - and visible to a DEMI_INV_PROGRAM (LDX is a pure row; STX is not allowed there).  The length
 * lives in bits 8..15 of `flags`.  Like a wide table, a table with an array runs only as compiled code
 * (demi_model_specialize).  Budget: a schedule's actor states live in LDS, n_actors x (1 (wide: 2) + ceil(n / 8 (wide: 4)))
 * words of 8 bytes per simulated schedule, 256 schedules to a workgroup of the RandomScheduler kernel - 2 KB per state word and
 * actor next to the tables, within the 160 KB of a CU (5 actors with 64 narrow elements: 90 KB; with 64 wide ones the launch is
 * refused: DEMI_ERR_INVALID_ARG, "LDS budget exceeded"). */
#define DEMI_MAX_ARRAY 64
#define DEMI_MODEL_ARRAY(n) ((uint32_t)(n) << 8)
#define DEMI_MODEL_ARRAY_LEN(flags) (((flags) >> 8) & 0xFFu)

/* DEMI_MODEL_PAYLOADS(n), n = 3..6 (with DEMI_MODEL_WIDE only: the message word is then 64 bits): a message carries n payload
 * fields instead of two - akka-raft's AppendEntries(term, prevLogIndex, prevLogTerm, entries, leaderCommit) or
 * RequestVote(term, candidateId, lastLogTerm, lastLogIndex) lower field by field instead of being packed into two values by
 * hand.  The reference identifies a message by what its MessageFingerprinter makes of the whole object
 * (V/MessageFingerprints.scala:83-101); here the identity of a message IS its word, so every field is part of it: of the
 * pending set's equality (FullyRandom.remove, the replay's messagePending), of the DPOR node key and of demi_verdict.hash.
 * Layout: the 48 bits above the 16-bit header are the PAYLOAD AREA; field k occupies bits [k * w, (k + 1) * w) of it with
 * w = DEMI_PAYLOAD_BITS(n) = 16, 12, 9, 8 for n = 3, 4, 5, 6 (and 16 for the two fields of a plain wide table - the same
 * rule, so nothing changes for those):
 *     word = type[4:0] | dst[7:5] | src[11:8] | area[63:16],    area = sum of (P_k mod 2^w) << (k * w)
 * A SEND truncates each field to w bits (the register window stays 16 bits wide; a protocol whose terms outgrow w bits needs a
 * smaller n).  Rows: P0 / P1 are r12 / r13 and the SEND / BCAST operands as before; DEMI_OP_LDP reads any field, DEMI_OP_PSET
 * stages P2..P5 of the messages sent next.  At the boundary: external Sends carry P0 and P1 only (demi_ext_event; the other
 * fields of an external message are 0); demi_rec_event.p0 / p1 / p_hi are bits 0..15 / 16..31 / 32..47 of the area (for n = 3
 * exactly P0, P1, P2; DEMI_REC_PAYLOAD extracts field k for any n); demi_dpor_trace_entry.word stays the low half of the
 * word.  Like every wide table it runs only as compiled code.  The count lives in bits 16..18 of `flags` (0 = two fields). */
#define DEMI_MODEL_PAYLOADS(n) ((uint32_t)(n) << 16)
#define DEMI_MODEL_PAYLOADS_N(flags) ((((flags) >> 16) & 7u) ? (((flags) >> 16) & 7u) : 2u)
#define DEMI_MAX_PAYLOADS 6
#define DEMI_PAYLOAD_BITS(n) ((n) <= 3 ? 16u : 48u / (uint32_t)(n))
/* field k of a 48-bit payload area for a model with n payload fields */
#define DEMI_PAYLOAD_OF(area, n, k) \
  ((k) < (n) ? (uint32_t)(((uint64_t)(area) >> ((k) * DEMI_PAYLOAD_BITS(n))) & ((1u << DEMI_PAYLOAD_BITS(n)) - 1u)) : 0u)

typedef struct {
  uint32_t max_messages;              /* RandomScheduler.setMaxMessages (RandomScheduler.scala:54-57); 0 = unbounded */
  uint32_t invariant_check_interval;  /* RandomScheduler ctor arg (RandomScheduler.scala:43); 0 = only at the end */
  uint32_t p_max;                     /* capacity of the pending set per schedule, 1..128 (0 = 64) */
  uint32_t looking_for_valid;         /* explore(_trace, _lookingFor) (RandomScheduler.scala:234-237) */
  uint32_t looking_for;               /* target fingerprint code */
  uint32_t populate_all;              /* setActorNamePropPairs: create all actors, not only Start()ed ones */
  uint32_t strategy;                  /* RandomizationStrategy (RandomScheduler.scala:614-633): demi_strategy */
  uint32_t filter_known_absents;      /* replays only: SchedulerConfig.filterKnownAbsents (SchedulerConfig.scala:14) ->
                                         EventTrace.filterKnownAbsentInternals (EventTrace.scala:458-534): demi_filter_absents */
  uint32_t executions_per_instance;   /* RandomScheduler entry points: 0 / 1 = every execution is a scheduler of its own, seeded
                                         seed_base + i (RunnerUtils.fuzz's shape).  k > 1 = the carried-generator mode of ONE
                                         `new RandomScheduler(config, max_executions = k)`: verdict index i belongs to instance
                                         i / k, seeded seed_base + i / k (seeds[i / k]), and is its execution number i % k; the
                                         instance's generator(s) are NOT reseeded between its executions (reset_all_state only
                                         clears the pending set, RandomScheduler.scala:575-595, 649-651), the application's
                                         seededRandom restarts at 0 with every execution as always, lookingFor only applies to
                                         execution 0 (reset_all_state sets it to None, :586), and explore() returns at the first
                                         violating execution (:257-261): the instance's later verdicts are all-zero ("not run").
                                         An instance is a sequential chain; instances run in parallel. */
} demi_limits;

/* EventTrace.filterKnownAbsentInternals drops from the projected trace every internal MsgSend whose sender is not alive or
 * is "partitioned" from the receiver, the MsgEvents of those sends, and every MsgEvent whose receiver is not alive or is
 * "partitioned" from the sender.  The reference's partition bookkeeping is inverted (EventTrace.scala:523-528: a
 * PartitionEvent stores false, an UnPartitionEvent true, under the ordered pair of the event): LITERAL reproduces exactly
 * that, CORRECTED treats a pair as cut off between Partition and UnPartition, in either direction. */
typedef enum { DEMI_FILTER_ABSENTS_OFF = 0, DEMI_FILTER_ABSENTS_LITERAL = 1, DEMI_FILTER_ABSENTS_CORRECTED = 2 } demi_filter_absents;

/* The pending-message container of RandomScheduler.
 * FULLY_RANDOM: FullyRandom (RandomScheduler.scala:635-697), one RandomizedHashSet seeded with the execution's seed.
 * SRC_DST_FIFO: SrcDstFIFO (:702-909): TCP-like delivery, one FIFO per (src, dst) pair of actors, an ordered list of
 *   the pairs that currently have a queue, timers and external messages (sender deadLetters) in a FullyRandom of
 *   their own; RandomScheduler asks it through getNonBlockedMessage (:443-449, 716-760).  The reference seeds both of
 *   its generators with System.currentTimeMillis(); here both are `new Random(seed)` of the execution's seed.  K1 only. */
typedef enum { DEMI_STRATEGY_FULLY_RANDOM = 0, DEMI_STRATEGY_SRC_DST_FIFO = 1 } demi_strategy;

/* One verdict per candidate schedule. */
#define DEMI_V_VIOLATION     0x1u  /* invariant violated (and matching looking_for when set)         */
#define DEMI_V_MAXMSG        0x2u  /* messagesScheduledSoFar > maxMessages: not bug-checked (:256)   */
#define DEMI_V_PENDING_OVF   0x4u  /* pending set exceeded p_max: schedule aborted, verdict invalid   */
#define DEMI_V_QUEUE_OVF     0x8u  /* timer queues / DEMI_FX_CAP exceeded: aborted, verdict invalid       */
#define DEMI_V_DIVERGED      0x10u /* replay kernels: an expected delivery was absent (ignored)       */
typedef struct {
  uint32_t flags;        /* bits 0..7 DEMI_V_*; bits 8..15 traceIdx at the end; bits 16..31 deliveries, saturating at 65535
                            (max_messages = 0 is unbounded: a longer execution reports 65535, never a wrapped count) */
  uint32_t fingerprint;  /* ViolationFingerprint code, 0 if none */
  uint64_t hash;         /* FNV-1a over every delivered message word, then every actor's final state */
} demi_verdict;          /* 16 bytes */

/* Recorded EventTrace of one execution (EventTrace.scala:20, AuxilaryTypes.scala:34-69). */
typedef enum {
  DEMI_REC_SPAWN = 0, DEMI_REC_KILL = 1, DEMI_REC_PARTITION = 2, DEMI_REC_UNPARTITION = 3,
  DEMI_REC_BEGIN_WAIT_QUIESCENCE = 4, DEMI_REC_QUIESCENCE = 5,
  DEMI_REC_MSG_SEND = 6,   /* UniqueMsgSend(MsgSend(snd, rcv, msg), id) */
  DEMI_REC_MSG_EVENT = 7   /* UniqueMsgEvent(MsgEvent(snd, rcv, msg), id) */
} demi_rec_kind;

typedef struct {
  uint8_t kind;            /* demi_rec_kind */
  uint8_t snd, rcv;        /* MSG_*: sender (15 = deadLetters, 31 for a table of more than 8 actors; timers are recorded as "Timer"), receiver;
                              SPAWN/KILL: rcv = actor; (UN)PARTITION: snd = a, rcv = b */
  uint8_t msg_type;
  uint16_t p0, p1;         /* the message's payload fields: below 256 unless the model is DEMI_MODEL_WIDE (with
                              DEMI_MODEL_PAYLOADS: bits 0..15 / 16..31 of the payload area) */
  uint8_t flags;           /* bit0: external message, bit1: timer, bit2: dropped at send (crosses_partition) */
  uint8_t ext_idx;         /* index of the ExternalEvent that caused this record, 255 = none */
  uint16_t p_hi;           /* DEMI_MODEL_PAYLOADS(n >= 3): bits 32..47 of the payload area; otherwise 0 */
  uint32_t id;             /* Uniq id pairing a MSG_SEND with its MSG_EVENT */
} demi_rec_event;          /* 16 bytes */
/* payload field k of a recorded message of a model with n payload fields (n = 2: p0 / p1 themselves) */
#define DEMI_REC_AREA(e) ((uint64_t)(e).p0 | ((uint64_t)(e).p1 << 16) | ((uint64_t)(e).p_hi << 32))
#define DEMI_REC_PAYLOAD(e, n, k) DEMI_PAYLOAD_OF(DEMI_REC_AREA(e), n, k)

typedef struct demi_ctx demi_ctx;

/* ---------------------------------------------------------- lifecycle */
int demi_ctx_create(int device_ordinal, demi_ctx** out);
void demi_ctx_destroy(demi_ctx* ctx);
const char* demi_last_error(const demi_ctx* ctx);
const char* demi_version(void);
/* The layout generation of this header's structs.  It changes whenever a struct a caller fills or reads changes size or
 * meaning (4: demi_rec_event is 16 bytes with 16-bit payloads; demi_limits ends with executions_per_instance; demi_dpor_search
 * ends with ordering / max_distance_plus1 / resume; demi_dpor_stats ends with backtrack_points).  A binding built against
 * another generation must refuse to run: compare demi_abi_version() with the DEMI_ABI_VERSION it was compiled with right
 * after loading the library (jni/demi_jni.c does so in JNI_OnLoad, demi_amd/_native.py in lib()). */
#define DEMI_ABI_VERSION 4u
uint32_t demi_abi_version(void);

/* SchedulerConfig (SchedulerConfig.scala:9-37) + the application actors lowered to a table. */
int demi_model_load(demi_ctx* ctx, const demi_model* model);
/* Compile the loaded transition table to native gfx950 code (hiprtc, ~1 s) and use that kernel for the
 * following demi_random_explore* launches (and, compiled at their first launch, demi_replay_* / demi_dpor_*) instead
 * of the table interpreter: the reference hands every delivery to the application's own JVM-compiled `receive`
 * (Instrumenter.dispatch_new_message, Instrumenter.scala:913-1017; WeaveActor.aj around-advice on receive), this is
 * the same step for the lowered table.  Verdicts are
 * bit-identical with and without it.  enable = 0 returns to the interpreter.  A failure (no hiprtc, compile
 * error) leaves the interpreter in place and returns the reason through demi_last_error.  demi_model_load
 * drops the specialisation of the previous model.                                                         */
int demi_model_specialize(demi_ctx* ctx, int enable);
int demi_model_is_specialized(const demi_ctx* ctx);
/* Identity of the compiled RandomScheduler kernel of the loaded model: a 64-bit hash of its code object, 0 when the table is
 * interpreted.  Measurement aid: bench.py quotes hardware counters from profiles/ only for the build they were taken on. */
uint64_t demi_model_code_id(const demi_ctx* ctx);
/* Device-free check of the same code generation + compilation (build / CI): code-object size in bytes, or the
 * (negative) demi_status with the reason in `log`.                                                         */
long demi_specialize_check(const demi_model* model, char* log, size_t log_cap);
/* The generated handler source (C++), NUL-terminated, truncated to cap; returns its full length. */
long demi_specialize_source(const demi_model* model, char* out, size_t cap);
/* The same for the RandomScheduler kernel's flavour of it: when the table admits one, every effect row is given a fixed slot
 * of a per-table schedule of effect classes (program order kept along every path), so that the kernel applies slot j of all
 * its lanes as one class; the handler then returns the mask of filled slots instead of a count. */
long demi_specialize_source_k1(const demi_model* model, char* out, size_t cap);
/* The external-event trace handed to explore()/test() (RandomScheduler.scala:226-237). */
int demi_trace_load(demi_ctx* ctx, const demi_ext_event* events, uint32_t n_events);
/* External Sends of a DEMI_MODEL_PAYLOADS table with ALL their fields (Send(name, messageCtor), ExternalEvents.scala:62-91: the
 * constructor's message is whatever the application sends; demi_ext_event has room for P0 and P1).  areas[i] = the 48-bit payload
 * area (DEMI_MODEL_PAYLOADS above: field k in bits [k * w, (k + 1) * w)) of external event i of the trace that the NEXT
 * demi_trace_load / demi_dpor_load of this context loads - read for its Send events only, n must equal that load's count, the
 * load consumes it (a later load without a new call takes the fields from the events again: P0, P1, the others 0).  A table with
 * two payload fields ignores it.  demi_replay_load needs none: the recorded MsgSend of an external message carries its area
 * (demi_rec_event p0 / p1 / p_hi), and that is what a replay enqueues.  NULL, 0: forget a staged array. */
int demi_ext_payload_areas(demi_ctx* ctx, const uint64_t* areas, uint32_t n);

/* ---------------------------------------------------------- K1: RandomScheduler
 * Replaces the loop of RandomScheduler.explore (RandomScheduler.scala:234-272) in the
 * fresh-scheduler-per-execution shape of RunnerUtils.fuzz (RunnerUtils.scala:75-90): schedule i
 * is one full execution with `new FullyRandom(seed = seeds[i])` (RandomScheduler.scala:635-697).
 * `seeds == NULL` means seeds[i] = seed_base + i.  Host-buffer form (what JNI binds).
 * Any n is a sensible call: a launch far smaller than the chip runs on few lanes of many wavefronts (the SPREAD variant of the
 * kernel - a wavefront's step costs the union of its lanes' paths): 100 schedules of config 1 take 0.17 ms, 1 024 of config 2
 * 0.40 ms, 16 384 0.69 ms, 2^20 4.0 ms.  Same verdicts whatever the launch's shape.            */
int demi_random_explore(demi_ctx* ctx, uint64_t seed_base, const uint64_t* seeds, uint64_t n,
                        const demi_limits* limits, demi_verdict* out);
/* Device-resident form: d_seeds (or NULL) and d_out are device pointers, the launch is enqueued
 * on `hip_stream` (a hipStream_t, NULL = default stream) and not synchronised.                 */
int demi_random_explore_dev(demi_ctx* ctx, uint64_t seed_base, const uint64_t* d_seeds, uint64_t n,
                            const demi_limits* limits, demi_verdict* d_out, void* hip_stream);
/* The EventTrace of one execution (what explore() returns for a violating schedule,
 * RandomScheduler.scala:156-180), recorded on the GPU by re-running that seed.                 */
int demi_random_get_trace(demi_ctx* ctx, uint64_t seed, const demi_limits* limits,
                          demi_verdict* verdict, demi_rec_event* out, uint32_t cap, uint32_t* n_out);
/* The same for execution number exec_index of the instance seeded `seed` in the carried-generator mode
 * (demi_limits.executions_per_instance): the chain is re-run from its first execution; *executed_index (may be NULL) is the
 * execution the instance stopped at - exec_index, or an earlier one that violated, whose verdict and trace are then returned. */
int demi_random_get_trace_carried(demi_ctx* ctx, uint64_t seed, uint32_t exec_index, const demi_limits* limits,
                                  demi_verdict* verdict, demi_rec_event* out, uint32_t cap, uint32_t* n_out,
                                  uint32_t* executed_index);

/* ---------------------------------------------------------- K2: DDMin's replay oracle
 * Replaces STSScheduler.test without peek (STSScheduler.scala:199-310) as called by DDMin.ddmin2
 * (minification/DeltaDebugging.scala:73-109) once per candidate: the original failing execution
 * (its external events + its recorded EventTrace, e.g. from demi_random_get_trace) is loaded once;
 * every candidate subsequence is a 256-bit mask over the external events (bit i = event i kept;
 * WaitQuiescence bits are ignored, RunnerUtils.scala:680-684).  limits->looking_for is the target
 * ViolationFingerprint (limits->looking_for_valid must be 1); p_max and populate_all as in K1.
 * Verdict: DEMI_V_VIOLATION = "the subsequence still triggers the violation" (test() returned Some),
 * DEMI_V_DIVERGED = at least one expected delivery was absent and ignored.                      */
int demi_replay_load(demi_ctx* ctx, const demi_ext_event* original_externals, uint32_t n_ext,
                     const demi_rec_event* original_trace, uint32_t n_rec);
int demi_replay_batch(demi_ctx* ctx, const uint64_t* masks /* [n][4] */, uint64_t n, const demi_limits* limits,
                      demi_verdict* out);
int demi_replay_batch_dev(demi_ctx* ctx, const uint64_t* d_masks, uint64_t n, const demi_limits* limits,
                          demi_verdict* d_out, void* hip_stream);

/* ---------------------------------------------------------- K2 for internal-event minimization
 * Replaces RunnerUtils.testWithStsSched (RunnerUtils.scala:913-943) as called by
 * STSSchedMinimizer.minimize (minification/internal_minimization/ScheduleCheckers.scala:50-57) on the
 * traces a OneAtATimeStrategy proposes (OneAtATimeRemoval.scala:57-124): candidate i is the loaded trace
 * (demi_replay_load, normally the verified MCS execution and its externals) minus the one MsgEvent /
 * TimerDelivery at index skip[i] of original_trace (0xFFFFFFFF removes nothing).  A removed delivery is
 * not "ignored": it does not set DEMI_V_DIVERGED.  masks == NULL keeps every external (test(mcs)).
 * A removal strategy proposes its candidates one after another assuming each fails; the mirror
 * (demi_amd/internal_minimization.py) enumerates that whole sequence and evaluates it in one launch.  */
int demi_replay_removal_batch(demi_ctx* ctx, const uint64_t* masks /* [n][4] or NULL */, const uint32_t* skip /* [n] */,
                              uint64_t n, const demi_limits* limits, demi_verdict* out);
/* The executed trace of one candidate, i.e. what test() returns on success (STSScheduler.scala:286-292):
 * kept[i] = 1 iff original_trace[i] took effect in the replay (external event applied, external MsgSend
 * enqueued, MsgEvent delivered, internal / timer MsgSend whose delivery took effect); absent deliveries, pruned
 * events, sends of messages that were never delivered and the quiescence markers are 0.  kept: [n_rec].  */
int demi_replay_get_kept(demi_ctx* ctx, const uint64_t* mask /* [4] or NULL */, uint32_t skip,
                         const demi_limits* limits, demi_verdict* verdict, uint8_t* kept);
/* Number of recorded events of the execution loaded by demi_replay_load (0 without one): the size demi_replay_get_kept's
 * out_kept needs.  Lets a binding check its buffer before the call. */
uint32_t demi_replay_recorded_len(const demi_ctx* ctx);
/* Number of external events of that execution (0 without one): the length demi_ddmin's `conjoined` must have. */
uint32_t demi_replay_externals_len(const demi_ctx* ctx);
/* external events of the trace demi_trace_load holds (0: none) - what demi_random_ddmin's `conjoined` array must cover */
uint32_t demi_trace_len(const demi_ctx* ctx);

/* ---------------------------------------------------------- DDMin over the replay oracle, in one call
 * Replaces RunnerUtils.stsSchedDDMin (RunnerUtils.scala:642-707): DDMin.minimize / ddmin2 (minification/DeltaDebugging.scala:27-109)
 * over the EventDag of the loaded execution's external events (demi_replay_load), WaitQuiescence events stripped (:680-684), atomic
 * events as EventDag.get_atomic_events builds them (minification/Util.scala:197-265: a Kill with the Start of its actor, an
 * UnPartition with its Partition, pairs given in `conjoined`, singletons), MinificationUtil.split_list (:9-37), with
 * STSScheduler.test as the oracle (demi_replay_batch).  ddmin2 consults its oracle once per node of a sequential decision
 * tree; here the candidates the next levels could ask for, whatever the outcomes, are replayed together in one launch, and
 * the real path walks through the verdicts - the MCS and the sequence of consultations are those of the sequential algorithm.
 * A candidate whose replay exceeds limits->p_max is replayed with the largest pending set; DEMI_ERR_CAPACITY if it still does
 * not fit (never "does not reproduce").  DEMI_ERR_INVALID_ARG: the unmodified trace does not trigger the violation
 * (check_unmodified).  DEMI_ERR_INVALID_TRACE: the externals' atoms do not partition them (e.g. a Kill without its Start).
 * With a communicator (demi_comm_create*) every rank calls demi_ddmin with the same arguments: the candidates of each launch are
 * split over the ranks in contiguous blocks (demi_replay_batch_sharded) and one all-gather of the verdicts lets every rank walk
 * the same decision tree - same MCS, consultations and launch sizes as the single-rank call on every rank; max_candidates is the
 * width of a launch over ALL ranks, so W ranks test a frontier W times as wide in the time of one. */
typedef struct {
  uint32_t depth;            /* levels of the decision tree tested ahead per launch; 0 = as many as fit max_candidates */
  uint32_t max_candidates;   /* per launch when depth = 0 (0 = 4096) */
  uint32_t check_unmodified; /* DDMin(checkUnmodifed): test the whole trace first */
  uint32_t verify_mcs;       /* DDMin.verify_mcs: replay the MCS once more */
} demi_ddmin_params;
typedef struct {
  uint32_t consultations;    /* oracle.test calls of the sequential algorithm (MinimizationStats.total_replays) */
  uint32_t launches;         /* K2 launches, the two optional single replays included */
  uint32_t mcs_len;
  uint32_t verified;         /* verify_mcs: the MCS reproduces the violation */
  uint64_t replays;          /* candidates replayed, speculation included */
} demi_ddmin_stats;
/* conjoined (may be NULL): [n_ext] index of the event this one is conjoined with (UnmodifiedEventDag.conjoinAtoms), 255 = none.
 * out_mcs: the minimal causal sequence as a mask over the external events.  out_consulted [cap][4] / out_passed [cap] (may be
 * NULL): the candidates in consultation order and whether each "passed" (did not reproduce).  out_batches [batches_cap] (may be
 * NULL): candidates per launch. */
int demi_ddmin(demi_ctx* ctx, const demi_limits* limits, const demi_ddmin_params* params, const uint8_t* conjoined,
               uint64_t out_mcs[4], uint64_t* out_consulted, uint8_t* out_passed, uint32_t cap, uint32_t* out_batches,
               uint32_t batches_cap, demi_ddmin_stats* stats);

/* ---------------------------------------------------------- randomDDMin: DDMin over the RandomScheduler itself
 * RunnerUtils.randomDDMin (RunnerUtils.scala:601-623) in one call: DDMin (minification/DeltaDebugging.scala:27-109) whose
 * TestOracle is RandomScheduler.test (RandomScheduler.scala:597-612) - a candidate subsequence of the external events "fails"
 * (still triggers the violation) iff one of `executions` random interleavings of it ends in a violation that matches
 * limits.looking_for.  The externals are those of demi_trace_load (trace.original_externals, WaitQuiescence events included:
 * :609-610 minimises them as they are); limits as for demi_random_explore (RunnerUtils sets max_messages = the recorded
 * trace's length); execution k of every candidate runs under java.util.Random(seed_base + k).  The decision tree is evaluated
 * speculatively as in demi_ddmin: ONE launch runs (frontier candidates x executions) - a workgroup per candidate, whose
 * projected external trace is workgroup-shared - and the tree then walks through the verdicts; the MCS and the sequence of
 * consultations are those of the sequential algorithm.  With a communicator (demi_comm_*) every frontier is split over the ranks
 * in contiguous blocks and the candidates' verdict bits are all-gathered.
 * stats: consultations, launches, replays (executions run on the device, speculation included), mcs_len, verified (verify_mcs
 * when the MCS is smaller than the externals, :611-618: 1 = it reproduces). */
/* RandomScheduler.test for a batch of SUBSEQUENCES of the loaded external trace (what demi_random_ddmin launches per frontier):
 * masks [n_cand][4], bit i = external event i takes part.  out_verdicts [n_cand * executions] (may be NULL): entry c * executions + k
 * is demi_random_explore's verdict for demi_trace_load(the events of masks[c]) under seed_base + k.  out_flags [n_cand] (may be
 * NULL): bit 0 = some execution of the candidate violates (limits.looking_for applies), bit 1 = some execution was aborted on a
 * capacity (DEMI_V_PENDING_OVF / DEMI_V_QUEUE_OVF). */
int demi_random_explore_candidates(demi_ctx* ctx, uint64_t seed_base, const uint64_t* masks, uint32_t n_cand, uint32_t executions,
                                   const demi_limits* limits, demi_verdict* out_verdicts, uint32_t* out_flags);
typedef struct demi_random_ddmin_params {
  uint32_t executions;        /* R: RandomScheduler(config, max_executions = R) per consultation (0 = 1, the reference's own value) */
  uint32_t depth;             /* 0: as many unknown levels of the decision tree per launch as fit max_candidates; k: k levels below the node */
  uint32_t max_candidates;    /* candidates per launch (0 = 256) */
  uint32_t check_unmodified;  /* DDMin's checkUnmodifed (RunnerUtils passes false) */
  uint32_t verify_mcs;
  uint32_t sequential;        /* 1: no speculation - one launch per consultation, as the reference consults its oracle */
  uint32_t reserved[2];
} demi_random_ddmin_params;
int demi_random_ddmin(demi_ctx* ctx, uint64_t seed_base, const demi_limits* limits, const demi_random_ddmin_params* params,
                      const uint8_t* conjoined /* [n_ext] or NULL: partner index of explicitly conjoined events, 255 = none */,
                      uint64_t out_mcs[4], uint64_t* out_consulted /* [cap][4] or NULL */, uint8_t* out_passed /* [cap] or NULL */,
                      uint32_t cap, uint32_t* out_batches /* candidates per launch, or NULL */, uint32_t batches_cap,
                      demi_ddmin_stats* stats);

/* ---------------------------------------------------------- K3: DPORwHeuristics interleavings
 * Replaces, per interleaving, DPORwHeuristics.schedule_new_message / event_produced / getMessage /
 * runExternal / notify_quiescence (DPORwHeuristics.scala:421-648, 803-847, 773-801, 684-721,
 * 855-942) and the racing-pair analysis of dpor() (:1020-1139: isCoEnabeled, analyze_dep,
 * getCommonPrefix).  The backtrack priority queue, the ExploredTacker and getNext() (:1142-1185)
 * stay on the host (demi_amd/dpor.py here, the Scala DPORwHeuristics in production); a round of the
 * queue is one batch: lane = one interleaving = one `nextTrace` prefix.
 *
 * Message identity across interleavings: the dep-graph is a tree (every node has one out-edge,
 * :844-845) and getMessage() collapses equal (snd, rcv, fingerprint) children of one parent
 * (:773-801), so a node is identified by the 64-bit hash chain of its path,
 *   key(child) = (key(parent) ^ word) * FNV_PRIME,   key(root) = FNV_OFFSET,
 *   key(WaitQuiescence marker of external event i) = FNV_OFFSET ^ (0x5155494553434500 | i)
 * which needs no global id counter and is identical on every GPU.
 * External events: Start, Send, WaitQuiescence only (:692-710); all actors exist and start isolated
 * (setActorNameProps, :666-679).                                                                   */
#define DEMI_DPOR_MAX_TRACE 256   /* root + deliveries + quiescence markers of one interleaving */
typedef struct {
  uint32_t depth_bound;        /* depth_bound ctor arg / setDepthBound (:81, 108-115); 0 = unbounded */
  uint32_t max_messages;       /* setMaxMessagesToSchedule (:118-122); 0 = unbounded; counts scheduler calls */
  uint32_t looking_for_valid;  /* 0: any violation counts */
  uint32_t looking_for;
  uint32_t p_max;              /* pending capacity, 1..128 (0 = 64) */
  uint32_t max_pairs;          /* capacity of the racing-pair list per interleaving */
  uint32_t prioritize_pending; /* prioritizePendingUponDivergence (:65-68, 537-550, 594-597): when the expected head of
                                  nextTrace is not pending, keep popping heads until one is, before diverging */
} demi_dpor_params;

typedef struct {
  uint64_t key;      /* node identity (see above) */
  uint32_t word;     /* message word (0 for root and markers) */
  uint8_t parent;    /* trace index of the delivery (or root / marker) that produced it */
  uint8_t qperiod;   /* quiescentPeriod(node): 0, or 1 + index of the WaitQuiescence that opened it */
  uint8_t depth;     /* getPathLength(node) */
  uint8_t kind;      /* 0 root, 1 message delivery, 2 WaitQuiescence marker */
} demi_dpor_trace_entry; /* 16 bytes */

typedef struct { uint8_t branch, later, earlier, pad; } demi_dpor_pair;  /* trace indices (:1043-1077) */

#define DEMI_V_TRACE_OVF 0x20u  /* interleaving longer than DEMI_DPOR_MAX_TRACE: aborted           */
#define DEMI_V_PAIRS_OVF 0x40u  /* more racing pairs than max_pairs: list truncated                 */
#define DEMI_V_SELFMSG   0x80u  /* "self message without prior messages!" (:631-633): aborted       */

int demi_dpor_load(demi_ctx* ctx, const demi_ext_event* externals, uint32_t n_ext);
/* prefixes: [n][stride] entries of nextTrace (root and markers included, as getNext builds them,
 * :1180; only key, word and kind are read); outputs are [n], [n][DEMI_DPOR_MAX_TRACE], [n],
 * [n][max_pairs], [n].  All host pointers.
 * shared_len (may be NULL = all 0): shared_len[i] leading events of prefix i are the `trace.take(branch + 1)` part of the
 * next trace (:1054-1057, 1180), i.e. events of the interleaving that produced this backtrack point.  The replay
 * reproduces them at the same trace indices, so every racing pair (later, earlier) with later < shared_len[i] was
 * already reported - same node keys, same branch - when that interleaving ran, and dpor()'s bookkeeping for it is a
 * no-op now: (earlier, later) is in the ExploredTacker (:1068-1070), and the backtrack point it would enqueue (:1134) was
 * either enqueued then with the same branch and an older creation time, so it is dequeued first, or its flipped pair
 * was explored already; getNext() drops the duplicate either way (:1153-1157).  Those pairs are not reported.  Only
 * valid with trackHistory and DefaultBacktrackOrdering (the duplicate's priority equals the original's); pass 0
 * otherwise.                                                                                         */
int demi_dpor_batch(demi_ctx* ctx, const demi_dpor_trace_entry* prefixes, const uint32_t* prefix_len,
                    const uint32_t* shared_len, uint32_t stride,
                    uint64_t n, const demi_dpor_params* params, demi_verdict* out_verdicts,
                    demi_dpor_trace_entry* out_traces, uint32_t* out_trace_len, demi_dpor_pair* out_pairs,
                    uint32_t* out_n_pairs);

/* The whole bounded exploration in one call: the backtrack priority queue with
 * DefaultBacktrackOrdering (BacktrackOrdering.scala:58-69), the ExploredTacker
 * (AuxilaryTypes.scala:209-246), dpor()'s bookkeeping (:1068-1070, 1134) and getNext() (:1142-1185)
 * inside the library around demi_dpor_batch-sized launches.  A round pops up to `batch`
 * unexplored backtrack points (batch = 1 is the reference's one-at-a-time order; PriorityQueue ties
 * pop in creation order) and each point carries its own next trace.
 * With trackHistory and the default ordering all of it is DEVICE-RESIDENT on one rank (round 5): the traces stay in an arena
 * (a backtrack point is 8 bytes: the interleaving that found it + three trace indices), the ExploredTacker is a hash table over
 * pairs of node keys, a racing pair the interleaving's parent provably applied never reaches it, the queue is a pool of points
 * sorted per round with getNext() as one probe + take over the candidates in dequeue order, and an interleaving starts from a
 * record of its parent's state below the branch point instead of re-executing the shared prefix.  None of it changes what is
 * explored: same rounds, verdicts and queue as the bookkeeping on host threads.          */
/* In which order the backtrack points are explored.
 * ROUNDS: a round pops up to `batch` unexplored points, runs them as one launch and absorbs their racing pairs in pop
 *   order.  batch = 1 is the reference's order; the explored-pair heuristic (ExploredTacker, AuxilaryTypes.scala:209-246)
 *   makes the SET of explored interleavings depend on the order, so a wider round explores a slightly different set
 *   (DESIGN.md section 4 K3 has the measured difference).  How wide: a launch takes about as long as ONE interleaving whatever
 *   its width, so rounds should be as wide as the queue fills them - 32 768 .. 65 536 on an MI355X (config 5's 2^20 interleavings:
 *   0.103 / 0.074 / 0.065 / 0.079 s in rounds of 16 384 / 32 768 / 65 536 / 131 072); beyond that the
 *   deepest-first order spends the width on shallow points.
 * REFERENCE: exactly the sequence of interleavings of batch = 1 - DPORwHeuristics' own depth-first order
 *   (:1142-1185) with PriorityQueue ties in creation order - whatever `batch` is.  The device runs ahead speculatively
 *   (a ROUNDS exploration of width `batch` whose results are cached by next-trace identity) and the host commits the
 *   cached results strictly one at a time, launching what the speculation missed.  An interleaving is a pure function of
 *   its next trace, so the committed sequence, verdicts and violating set do not depend on what was speculated.        */
typedef enum { DEMI_DPOR_ORDER_ROUNDS = 0, DEMI_DPOR_ORDER_REFERENCE = 1 } demi_dpor_order;

typedef struct {
  uint32_t batch;               /* backtrack points per launch (>= 1); REFERENCE order: width of the speculation */
  uint32_t max_interleavings;   /* budget; also the capacity of the output arrays */
  uint32_t stop_if_violation;   /* stopIfViolationFound */
  uint32_t track_history;       /* trackHistory */
  uint32_t order;               /* demi_dpor_order */
  uint32_t cache_mb;            /* REFERENCE order: host memory for speculated results not yet committed (0 = 1024) */
  uint32_t ordering;            /* demi_dpor_ordering: the backtrackHeuristic (DPORwHeuristics.scala:69) */
  uint32_t max_distance_plus1;  /* 0 = no cap; k + 1 = setMaxDistance(k) (:131-134): getNext() gives up - the queue is kept - once its
                                   head is at least k away from the original execution */
  uint32_t resume;              /* with an ordering / cap / initial trace: 1 = continue from the backtrack queue the previous such
                                   exploration of this demi_ctx left (a later test() of the same DPORwHeuristics, :1219-1220:
                                   ResumableDPOR); 0, or no queue left = start from the initial trace */
} demi_dpor_search;

/* BacktrackOrdering.scala: DEFAULT = DefaultBacktrackOrdering (:58-69, deepest branch first, distance 0);
 * ARVIND = ArvindDistanceOrdering (:99-173): a backtrack point's priority is (distance from the original execution, branch) -
 * events of its path the original trace lacks + misordered pairs among those it has; needs demi_dpor_set_traces.
 * With ARVIND, a distance cap or an initial trace, demi_dpor_explore runs the plain loop of the reference (ROUNDS order,
 * host bookkeeping, every racing pair enqueued and explored flips skipped at pop time) - the shortcuts of the default path
 * assume the default priority.  Single rank; DEMI_DPOR_ORDER_REFERENCE is refused with them.  The queue and the explored pairs
 * of such an exploration stay with the demi_ctx until the next demi_dpor_load: demi_dpor_search.resume continues from them. */
typedef enum { DEMI_DPOR_ORDERING_DEFAULT = 0, DEMI_DPOR_ORDERING_ARVIND = 1 } demi_dpor_ordering;
/* ArvindDistanceOrdering.init(sched, originalTrace) (:115-123) and DPORwHeuristics.setInitialTrace (:211-213): the node keys of
 * the original execution, and the next trace of the first interleaving (key, word, kind are read).  (NULL, 0) clears either.
 * Kept until the next demi_dpor_load. */
int demi_dpor_set_traces(demi_ctx* ctx, const uint64_t* original_keys, uint32_t n_original,
                         const demi_dpor_trace_entry* initial_trace, uint32_t n_initial);

typedef struct {
  uint64_t interleavings;       /* executed */
  uint64_t launches;
  uint64_t violations;
  uint64_t first_violation;     /* index into out_verdicts, ~0 if none */
  uint64_t queue_len;           /* backtrack points still queued at return */
  uint32_t exhausted;           /* the queue ran empty */
  uint32_t fetches;             /* REFERENCE order: how many times the commit asked the device for racing-pair records;
                                   ROUNDS order on one GPU: how many times the device-resident queue's pool was compacted */
  uint64_t executed;            /* interleavings run on the device (REFERENCE order: committed + speculated in vain) */
  uint64_t cache_misses;        /* REFERENCE order: committed interleavings the speculation had not run */
  double kernel_ms;             /* sum of the K3 launches' durations (HIP events on the launch stream) */
  uint64_t h2d_bytes;           /* next traces uploaded */
  uint64_t d2h_bytes;           /* verdicts, traces and racing pairs fetched */
  uint64_t backtrack_points;    /* backtrack points the bookkeeping enqueued (dpor() :1134 after the drops that getNext() would make
                                   anyway): the "r new backtrack points out" of an interleaving, summed */
} demi_dpor_stats;

/* out_verdicts / out_prefix_len: [max_interleavings], in execution order.  first_violation_trace:
 * [DEMI_DPOR_MAX_TRACE] (may be NULL).  out_rounds: launch sizes, [max_interleavings] (may be NULL).
 * Device memory follows the budget: 4 KB of trace arena per interleaving run, the explored-pair table (128 entries of 64 B per
 * interleaving of the budget: at most 4 GB up to 2^21 interleavings, 16 GB beyond), checkpoint records for as many interleavings
 * as a quarter of the free memory holds.  DEMI_ERR_CAPACITY: the explored-pair table is full (ROUNDS order; in the REFERENCE order
 * the device's table only steers the speculation, which then stops), or - several ranks - a round left more than 2^21 backtrack
 * points (on one rank the area they are staged in grows).  Measured on one MI355X: 2^24 interleavings of config 5 in 2.9 s. */
int demi_dpor_explore(demi_ctx* ctx, const demi_dpor_params* params, const demi_dpor_search* search,
                      demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                      demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                      demi_dpor_stats* stats);

/* What interleaving `index` of the last demi_dpor_explore of this context WAS (for auditing an exploration against another
 * implementation of DPORwHeuristics - an exploration hands back verdicts only): the next trace it was started from
 * (out_next_trace, *out_next_len entries; `trace.take(maxIndex + 1) ++ replayThis` of the backtrack point it was dequeued as,
 * DPORwHeuristics.scala:1054-1057, 1180; empty for the first interleaving), how many leading entries of it are the take() part
 * (*out_shared_len, may be NULL: demi_dpor_batch's shared_len), and the trace it executed (out_trace, *out_trace_len).  Both
 * buffers [DEMI_DPOR_MAX_TRACE].  Available after the explorations whose traces stay in the device's arena - trackHistory with
 * DefaultBacktrackOrdering, no distance cap, no initial trace, in either order - until the next demi_dpor_explore / demi_dpor_load
 * of the context; DEMI_ERR_INVALID_ARG otherwise.  With several ranks every rank holds every trace. */
int demi_dpor_explored(demi_ctx* ctx, uint64_t index, demi_dpor_trace_entry* out_next_trace, uint32_t* out_next_len,
                       uint32_t* out_shared_len, demi_dpor_trace_entry* out_trace, uint32_t* out_trace_len);

/* ---------------------------------------------------------- DDMin over DPOR with a growing edit-distance bound
 * RunnerUtils.editDistanceDporDDMin (RunnerUtils.scala:810-879) in one call: IncrementalDDMin (minification/
 * IncrementalDeltaDebugging.scala:20-92) - DDMin with the distance cap 0, then 2, 4, ... < max_max_distance, each pass starting
 * from the previous pass's MCS, until the MCS has at most stop_at_size events - over ResumableDPOR (:94-122): every
 * consultation is a DPORwHeuristics exploration (K3 launches) of one subsequence of the external events with
 * ArvindDistanceOrdering, `initial_trace` (DepTracker.getInitialTrace of the original execution: demi_dpor_trace_entry as K3 /
 * dpor_initial_trace produce them; also ArvindDistanceOrdering's original trace) as the first interleaving and
 * setMaxDistance(cap); the backtrack queue and explored pairs of a subsequence are kept, a later consultation of the same
 * subsequence continues from them.  `externals`: the original execution's; only Start / Send (and WaitQuiescence with
 * ignore_quiescence = 0) take part (convertToDPORTrace, DPORwHeuristics.scala:1279-1303).  `params` as for demi_dpor_explore
 * (looking_for = the violation; max_messages = n_initial is what RunnerUtils sets).  Replaces what demi_dpor_load /
 * demi_dpor_set_traces loaded.  Single rank.  Same MCS, consultations, caps and replay count as the reference's loop
 * (tests: against the Python mirror demi_amd/incremental_ddmin.py around the CPU oracle and on the GPU).                   */
typedef struct demi_incddmin_params {
  uint32_t max_max_distance;    /* IncrementalDDMin.maxMaxDistance (0 = 256, the class's default; RunnerUtils passes its own) */
  uint32_t stop_at_size;        /* stopAtSize */
  uint32_t check_unmodified;    /* checkUnmodifed: consult the whole view first (DEMI_ERR_INVALID_ARG if it does not reproduce) */
  uint32_t ignore_quiescence;   /* RunnerUtils' ignoreQuiescence (1 = WaitQuiescence events are not part of the minimization) */
  uint32_t verify_mcs;          /* verify_mcs when the MCS is smaller than the view (:868-873) */
  uint32_t batch;               /* interleavings per K3 launch (0 = 256) */
  uint32_t budget;              /* interleavings per internal exploration call (0 = 65536); a consultation continues until the
                                   queue is empty, its head reaches the cap, or a violation is found - never "until the budget" */
  uint32_t reserved;
} demi_incddmin_params;
typedef struct demi_incddmin_stats {
  uint64_t replays;             /* MinimizationStats.total_replays as IncrementalDDMin merges it (interleavings of the passes) */
  uint64_t interleavings;       /* every interleaving explored, checks and verification included */
  uint32_t consultations;       /* DDMin consultations over all passes */
  uint32_t instances;           /* DPORwHeuristics instances (distinct subsequences consulted) */
  uint32_t passes;              /* DDMin passes run (distance caps tried) */
  uint32_t mcs_len;
  int32_t verified;             /* -1: not verified (the MCS removed nothing, or verify_mcs = 0); 0 / 1: verify_mcs' answer */
  uint32_t violation_len;       /* entries written to out_violation_trace */
  uint32_t pass_distance[16];   /* per pass (the first 16): its cap, and the MCS size after it */
  uint32_t pass_mcs_len[16];
} demi_incddmin_stats;
/* out_mcs: bit i = external event i is in the MCS.  out_consulted [cap][4] / out_passed [cap] / out_distance [cap] (may be
 * NULL): every DDMin consultation in order - the subsequence, whether it "passes" (no violation), the cap it ran under.
 * out_violation_trace [DEMI_DPOR_MAX_TRACE] (may be NULL): the interleaving that reproduces the violation on the MCS. */
int demi_edit_distance_dpor_ddmin(demi_ctx* ctx, const demi_ext_event* externals, uint32_t n_ext,
                                  const demi_dpor_trace_entry* initial_trace, uint32_t n_initial, const demi_dpor_params* params,
                                  const demi_incddmin_params* ip, uint64_t out_mcs[4], uint64_t* out_consulted, uint8_t* out_passed,
                                  uint32_t* out_distance, uint32_t cap, demi_dpor_trace_entry* out_violation_trace,
                                  demi_incddmin_stats* stats);

/* ---------------------------------------------------------- provenance of a violation
 * ProvenanceTracker.pruneConcurrentEvents (schedulers/Util.scala:267-376; RunnerUtils.pruneConcurrentEvents,
 * RunnerUtils.scala:149-163) for n delivery traces at once: happens-before (same-machine receive order + "sent while
 * handling") closed transitively, then an event is kept iff it strictly precedes the last receive of at least one actor
 * of affected[i] (bit a = actor a, ViolationFingerprint.affectedNodes).  traces: [n][stride] entries as K3 /
 * dpor_initial_trace produce them (word, parent and kind are read; trace_len[i] <= DEMI_DPOR_MAX_TRACE);
 * out_keep: [n][DEMI_DPOR_MAX_TRACE / 64] words, bit u = event u of trace i is kept.  All host pointers.  The receiver of an
 * entry is read in the layout of the table the context holds (4 bits and up to 16 affected actors for a table of more than 8).  */
int demi_provenance_prune(demi_ctx* ctx, const demi_dpor_trace_entry* traces, const uint32_t* trace_len,
                          const uint32_t* affected, uint32_t stride, uint64_t n, uint64_t* out_keep);

/* ---------------------------------------------------------- found-violation set
 * One entry per violating schedule (what RunnerUtils.fuzz keeps: the violating execution's
 * index + fingerprint, RunnerUtils.scala:91-128).  Compacts a device verdict array into a device
 * list; *d_count receives the number of violations (may exceed cap: the list is then truncated).
 * Entry order is unspecified: it is a set (sort by index on the host if an order is needed).     */
typedef struct {
  uint64_t index;        /* index_base + position in the verdict array */
  uint32_t fingerprint;
  uint32_t flags;
} demi_violation;        /* 16 bytes */

/* explore() only ever needs the violating executions (RandomScheduler.scala:257-261 returns the first one): run n
 * schedules, keep the verdicts on the device, and copy back just the compacted violation set.  The entries returned
 * are sorted by index; when *n_violations exceeds cap the list is truncated to an arbitrary subset of `cap` entries.
 * 16 bytes per violation cross PCIe instead of 16 bytes per schedule.                                              */
int demi_random_explore_violations(demi_ctx* ctx, uint64_t seed_base, uint64_t n, const demi_limits* limits,
                                   demi_violation* out, uint32_t cap, uint64_t* n_violations);
/* The same with a caller-chosen selection: entries whose verdict flags intersect flag_mask.  With
 * DEMI_V_VIOLATION | DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF the list also names the executions that were ABORTED on a
 * capacity (their verdicts are invalid): the reference has no such capacities, so a driver re-runs those seeds with a
 * larger p_max (demi_random_get_trace) before it trusts "no violation at a lower index".  *first_index (may be NULL) is
 * the lowest selected index, computed on the device: exact even when the list is truncated.                        */
int demi_random_explore_flagged(demi_ctx* ctx, uint64_t seed_base, uint64_t n, const demi_limits* limits, uint32_t flag_mask,
                                demi_violation* out, uint32_t cap, uint64_t* n_flagged, uint64_t* first_index);

/* explore() in pieces, two of them in flight (RandomScheduler.scala:234-272: the loop over executions, here cut into calls of n
 * executions each).  demi_random_explore_submit enqueues the n executions seeded seed_base .. seed_base + n - 1 and the
 * compaction of the verdicts whose flags intersect flag_mask (0 = DEMI_V_VIOLATION) on a stream of the ctx's own and returns
 * at once with a ticket; at most THREE tickets are outstanding (call k runs on stream k mod 2).  demi_random_explore_wait
 * blocks until that call has finished and returns what demi_random_explore_flagged returns (list sorted by index, truncated to
 * cap <= 65536; *n_flagged and *first_index exact) and, if `out` is not NULL, all n verdicts (want_verdicts != 0 at the submit
 * sends them to pinned memory of the library behind the call, so that the wait only copies them on with the CPU; otherwise the
 * wait fetches them).  The loop to write is
 * `submit(k + 2); wait(k)`: while a host waits for ticket k the kernel of ticket k + 1 runs and that of k + 2 is queued behind
 * k's compaction, so the thinning tail of every launch is filled by the next launch's workgroups and the answer of k crosses
 * PCIe under them - a JVM gets the rate bench.py reports without owning a HIP stream.  (`submit(k + 1); wait(k)` is correct
 * but no faster than one call at a time: the compaction of k only runs in the tail of k + 1, so k + 2 would come too late.)
 * Hardware queues: the HIP runtime multiplexes a process's streams over GPU_MAX_HW_QUEUES (default 4) hardware queues, and two
 * streams on one queue do not overlap.  In a process that owns no other streams (a JVM) the library's two streams get queues of
 * their own; a process that has created three or more streams before its first submit (a PyTorch process) should start with
 * GPU_MAX_HW_QUEUES=8 in its environment, or it measures one call at a time (profiles/r06_pipeline_ab.txt: 4.18 against 3.23 ms
 * per 2^20 schedules).
 * A driver that wants the reference's answer (the lowest violating index) stops submitting at the first call that reports one.
 * Not for the carried-generator mode (one chain of executions: nothing to overlap). */
int demi_random_explore_submit(demi_ctx* ctx, uint64_t seed_base, uint64_t n, const demi_limits* limits, uint32_t flag_mask,
                               uint32_t want_verdicts, uint32_t* ticket);
int demi_random_explore_wait(demi_ctx* ctx, uint32_t ticket, demi_verdict* out, demi_violation* flagged, uint32_t cap,
                             uint64_t* n_flagged, uint64_t* first_index);

int demi_collect_violations_dev(demi_ctx* ctx, const demi_verdict* d_verdicts, uint64_t n, uint64_t index_base,
                                demi_violation* d_out, uint32_t cap, unsigned long long* d_count,
                                void* hip_stream);
int demi_collect_flagged_dev(demi_ctx* ctx, const demi_verdict* d_verdicts, uint64_t n, uint64_t index_base,
                             uint32_t flag_mask, demi_violation* d_out, uint32_t cap, unsigned long long* d_count,
                             void* hip_stream);

/* ---------------------------------------------------------- multi-GPU (no reference counterpart)
 * The reference evaluates one execution at a time in one JVM (Instrumenter.scala:1289-1296).  Here the candidates of
 * each loop are split over the GPUs of a node, one process (and one demi_ctx) per GPU; what the ranks exchange is always
 * an all-gather of fixed-size blocks: RCCL over xGMI when the communicator is created from an RCCL unique id (rank 0
 * calls demi_comm_unique_id and hands the 128 bytes to the other ranks by whatever means the host has - a JVM would use
 * its own control channel), or a host-supplied all-gather (demi_comm_create_host: any transport; blocks are staged
 * through host memory).  Without a communicator every sharded entry point degenerates to its single-GPU form.       */
typedef struct { char bytes[128]; } demi_comm_id;                 /* = ncclUniqueId */
typedef int (*demi_allgather_fn)(void* user, const void* send, void* recv /* world * bytes */, size_t bytes);
int demi_comm_unique_id(demi_comm_id* out);
int demi_comm_create(demi_ctx* ctx, const demi_comm_id* id, int rank, int world);
int demi_comm_create_host(demi_ctx* ctx, int rank, int world, demi_allgather_fn fn, void* user);
int demi_comm_destroy(demi_ctx* ctx);
int demi_comm_rank(const demi_ctx* ctx, int* rank, int* world);
/* every rank's `bytes` at d_send -> world * bytes at d_recv (rank r's block at r * bytes), device pointers, on hip_stream */
int demi_comm_allgather_dev(demi_ctx* ctx, const void* d_send, void* d_recv, size_t bytes, void* hip_stream);
/* RandomScheduler executions [0, n_total) split by index range over the ranks (RunnerUtils.fuzz's executions are
 * independent given per-execution seeds); every rank returns the merged found-violation set, sorted by schedule index
 * (*n_violations counts all of them; at most `cap` per rank are listed).                                           */
int demi_random_explore_sharded(demi_ctx* ctx, uint64_t seed_base, uint64_t n_total, const demi_limits* limits,
                                demi_violation* out, uint32_t cap, uint64_t* n_violations);
/* demi_replay_batch with the candidates in contiguous blocks over the ranks; every rank receives all n verdicts. */
int demi_replay_batch_sharded(demi_ctx* ctx, const uint64_t* masks /* [n][4] */, uint64_t n, const demi_limits* limits,
                              demi_verdict* out);

/* ---------------------------------------------------------- measurement helpers (no reference counterpart)
 * Used by bench.py and the profiling scripts so that the figures beside the throughput are measured on the box that
 * prints them: the shader clock under load (s_memtime cycles per 100 MHz wall_clock64 tick) and the cycles one SIMD spends
 * per wave64 integer VALU instruction with `waves_per_simd` waves issuing (the unit of K1's issue-rate model).          */
typedef struct {
  double shader_clock_ghz;
  double cycles_per_valu;
  uint32_t waves_per_simd;
  uint32_t num_cu;
} demi_probe_result;
int demi_device_probe(demi_ctx* ctx, uint32_t waves_per_simd, uint32_t iters, demi_probe_result* out);
/* The same for other instruction kinds (cycles_per_valu is then SIMD cycles per instruction of the probe's mix): kind 0 =
 * integer VALU (demi_device_probe), 1 = SALU only, 2 = VALU and SALU alternating, 3 = the code of a divergent two-instruction
 * `if` (saveexec / branch / body / restore).                                                                             */
int demi_device_probe_mix(demi_ctx* ctx, uint32_t waves_per_simd, uint32_t iters, uint32_t kind, demi_probe_result* out);
/* A kernel with a known byte count for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE: mode 0 writes / 1 reads `bytes`
 * with 4 B per lane (K1's scratch rows), 2 writes / 3 reads with 16 B per lane (the verdict array's pattern).          */
int demi_calib_rw(demi_ctx* ctx, uint32_t mode, uint64_t bytes, uint32_t repeats);

#ifdef __cplusplus
}
#endif
#endif /* DEMI_GPU_H */
