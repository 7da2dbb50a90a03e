/*
 * demi_oracle.c — CPU restatement of DEMi's RandomScheduler hot path.  TEST INFRASTRUCTURE,
 * see demi_oracle.h (PARITY UNPINNED vs the JVM reference; java.util.Random is pinned by KATs).
 *
 * Paths cited as V/... are /root/reference/src/main/scala/verification/...
 */
#include "demi_oracle.h"

#include <pthread.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ===================================================================== java.util.Random
 * JDK javadoc: seed scrambling, 48-bit LCG, next(bits), nextInt(bound) with the power-of-two
 * fast path and the rejection loop.  Call sites: V/schedulers/Util.scala:115,172.             */
#define JR_MULT 0x5DEECE66DULL
#define JR_MASK ((1ULL << 48) - 1)

void orc_jrandom_seed(orc_jrandom* r, uint64_t seed) { r->s = (seed ^ JR_MULT) & JR_MASK; }

int32_t orc_jrandom_next(orc_jrandom* r, int bits) {
  r->s = (r->s * JR_MULT + 0xBULL) & JR_MASK;
  return (int32_t)(uint32_t)(r->s >> (48 - bits));
}

int32_t orc_jrandom_next_int(orc_jrandom* r) { return orc_jrandom_next(r, 32); }

int32_t orc_jrandom_next_int_bound(orc_jrandom* r, int32_t bound) {
  int32_t v = orc_jrandom_next(r, 31);
  int32_t m = bound - 1;
  if ((bound & m) == 0) return (int32_t)(((int64_t)bound * (int64_t)v) >> 31);
  /* for (u = v; u - (v = u % bound) + m < 0; u = next(31)) with int32 wrap-around */
  int32_t u = v;
  for (;;) {
    v = u % bound;
    int32_t t = (int32_t)((uint32_t)u - (uint32_t)v + (uint32_t)m);
    if (t >= 0) break;
    u = orc_jrandom_next(r, 31);
  }
  return v;
}

double orc_jrandom_next_double(orc_jrandom* r) {
  int64_t hi = (int64_t)orc_jrandom_next(r, 26);
  int64_t lo = (int64_t)orc_jrandom_next(r, 27);
  return (double)((hi << 27) + lo) * (1.0 / (double)(1LL << 53));
}

/* ===================================================================== the two message-word layouts (include/demi_gpu.h)
 * A table with more than DEMI_MAX_ACTORS actors (DEMI_MODEL "big": up to DEMI_MAX_ACTORS_BIG, a wide table) has a 4-bit
 * receiver and a 5-bit sender field in its message word, deadLetters = DEMI_DEADLETTERS_BIG and 16-bit actor masks in its
 * fingerprints; every entry point sets the layout of the model it was given (per thread) before it touches a word. */
#define ORC_MAX_ACTORS DEMI_MAX_ACTORS_BIG
static _Thread_local int g_big;
static inline int model_big(const demi_model* m) { return m->n_actors > DEMI_MAX_ACTORS; }
#define ORC_DL (g_big ? (uint32_t)DEMI_DEADLETTERS_BIG : (uint32_t)(DEMI_DEADLETTERS))      /* sender id of externals and timers */
#define ORC_IS_ACTOR(a) ((a) < (g_big ? (uint32_t)DEMI_MAX_ACTORS_BIG : (uint32_t)DEMI_MAX_ACTORS))
/* ordered-pair matrices (partitions): bit a * 16 + b of 256 */
typedef struct { uint64_t w[4]; } orc_pairset;
static inline void ps_set(orc_pairset* p, uint32_t a, uint32_t b) { const uint32_t i = a * 16 + b; p->w[i >> 6] |= 1ULL << (i & 63); }
static inline void ps_clear(orc_pairset* p, uint32_t a, uint32_t b) { const uint32_t i = a * 16 + b; p->w[i >> 6] &= ~(1ULL << (i & 63)); }
static inline int ps_get(const orc_pairset* p, uint32_t a, uint32_t b) { const uint32_t i = a * 16 + b; return (int)((p->w[i >> 6] >> (i & 63)) & 1); }

/* The payload areas of external Sends (include/demi_gpu.h demi_ext_payload_areas): a table with DEMI_MODEL_PAYLOADS has messages
 * of up to six fields, demi_ext_event room for two.  Set before an entry point is called with the trace the areas belong to
 * (areas[i] = the area of external event i when it is a Send), cleared with NULL; a process-wide pointer (the worker threads of
 * the batch entry points read it), test infrastructure like everything here. */
static const uint64_t* g_ext_areas;
static uint32_t g_n_ext_areas;
void orc_set_ext_areas(const uint64_t* areas, uint32_t n) { g_ext_areas = areas; g_n_ext_areas = areas ? n : 0; }
#define ORC_EXT_AREA(m, e, idx) ((g_ext_areas && (idx) < g_n_ext_areas && DEMI_MODEL_PAYLOADS_N((m)->flags) > 2) ? (g_ext_areas[(idx)] & 0xFFFFFFFFFFFFull) \
                                 : area_of2((m), (e)->p0 | ((uint32_t)(e)->p0_hi << 8), (e)->p1 | ((uint32_t)(e)->p1_hi << 8)))

/* ===================================================================== model helpers */
static int timer_index(const demi_model* m, uint32_t type) {
  /* dense index of a TIMER-class type among the model's timer types, ascending */
  int k = 0;
  for (uint32_t t = 0; t < type; t++)
    if (m->msg_class[t] == DEMI_MSG_TIMER) k++;
  return k;
}

static uint64_t timer_bit(const demi_model* m, uint32_t rcv, uint32_t type) {
  return 1ULL << (rcv * DEMI_MAX_TIMER_TYPES + (uint32_t)timer_index(m, type));
}

/* state words of one actor: its field word(s) - one, two for DEMI_MODEL_WIDE - then its array (DEMI_MODEL_ARRAY: 8 elements
 * per word, 4 for a wide model), include/demi_gpu.h */
#define ORC_MAX_STW (2 + DEMI_MAX_ARRAY / 4)
static inline uint32_t model_fw(const demi_model* m) { return (m->flags & DEMI_MODEL_WIDE) ? 2u : 1u; }
static inline uint32_t model_arr_words(const demi_model* m) {
  const uint32_t per = (m->flags & DEMI_MODEL_WIDE) ? 4u : 8u;
  return (DEMI_MODEL_ARRAY_LEN(m->flags) + per - 1) / per;
}
static inline uint32_t model_stw(const demi_model* m) { return model_fw(m) + model_arr_words(m); }
uint32_t orc_model_state_words(const demi_model* m) { return model_stw(m); }
/* the initial state: the fields from init_state, the arrays empty */
static void state_init(const demi_model* m, uint64_t* st) {
  const uint32_t fw = model_fw(m), stw = model_stw(m);
  for (uint32_t a = 0; a < m->n_actors; a++)
    for (uint32_t k = 0; k < stw; k++) st[a * stw + k] = k < fw ? m->init_state[a * fw + k] : 0;
}

#define FAIL(...)                                  \
  do {                                             \
    if (err) snprintf(err, err_cap, __VA_ARGS__);  \
    return DEMI_ERR_INVALID_MODEL;                 \
  } while (0)

int orc_model_validate(const demi_model* m, char* err, size_t err_cap) {
  if (!m) FAIL("null model");
  if (m->n_actors < 1 || m->n_actors > DEMI_MAX_ACTORS_BIG) FAIL("n_actors out of range");
  if (m->n_actors > DEMI_MAX_ACTORS && !(m->flags & DEMI_MODEL_WIDE))
    FAIL("more than %d actors need DEMI_MODEL_WIDE (the big layout is a compiled table's)", DEMI_MAX_ACTORS);
  if (m->n_msg_types < 1 || m->n_msg_types > DEMI_MAX_MSG_TYPES) FAIL("n_msg_types out of range");
  if (m->n_classes < 1 || m->n_classes > DEMI_MAX_CLASSES) FAIL("n_classes out of range");
  if (m->code_len < 1 || m->code_len > DEMI_MAX_CODE) FAIL("code_len out of range");
  if (!m->msg_class || !m->actor_class || !m->handler_start || !m->code || !m->init_state)
    FAIL("null table pointer");
  uint32_t n_timers = 0;
  for (uint32_t t = 0; t < m->n_msg_types; t++) {
    if (m->msg_class[t] > DEMI_MSG_TIMER) FAIL("msg_class[%u] invalid", t);
    if (m->msg_class[t] == DEMI_MSG_TIMER) n_timers++;
  }
  if (n_timers > DEMI_MAX_TIMER_TYPES) FAIL("more than %d timer types", DEMI_MAX_TIMER_TYPES);
  for (uint32_t a = 0; a < m->n_actors; a++)
    if (m->actor_class[a] >= m->n_classes) FAIL("actor_class[%u] out of range", a);
  for (uint32_t i = 0; i < m->n_classes * m->n_msg_types; i++)
    if (m->handler_start[i] != 0xFFFF && m->handler_start[i] >= m->code_len)
      FAIL("handler_start[%u] out of range", i);
  for (uint32_t pc = 0; pc < m->code_len; pc++) {
    uint32_t w = m->code[pc];
    uint32_t op = w & 0xFF, bimm = (w >> 16) & 1, aux = (w >> 17) & 0x7F, b = w >> 24;
    if (!bimm && b > 15) FAIL("row %u: register operand b out of range", pc);
    switch (op) {
      case DEMI_OP_HALT: case DEMI_OP_MOV: case DEMI_OP_ADD: case DEMI_OP_SUB: case DEMI_OP_AND:
      case DEMI_OP_OR: case DEMI_OP_XOR: case DEMI_OP_SHL: case DEMI_OP_SHR: case DEMI_OP_BITSET:
      case DEMI_OP_POPC: case DEMI_OP_EQ: case DEMI_OP_NE: case DEMI_OP_LT: case DEMI_OP_GE:
      case DEMI_OP_LE: case DEMI_OP_GT: case DEMI_OP_MIN: case DEMI_OP_MAX: case DEMI_OP_RND:
        break;
      case DEMI_OP_MOVHI:
        if (!(m->flags & DEMI_MODEL_WIDE)) FAIL("row %u: MOVHI in a model without DEMI_MODEL_WIDE", pc);
        if (!bimm) FAIL("row %u: MOVHI takes an immediate", pc);
        break;
      case DEMI_OP_SKIPZ: case DEMI_OP_SKIPNZ: case DEMI_OP_SKIP:
        if (!bimm) FAIL("row %u: skip distance must be an immediate", pc);
        if (pc + 1 + b > m->code_len) FAIL("row %u: skip past the end of the table", pc);
        break;
      case DEMI_OP_IFEQ: case DEMI_OP_IFNE: case DEMI_OP_IFLT: case DEMI_OP_IFGE: case DEMI_OP_IFLE:
      case DEMI_OP_IFGT:
        if (pc + 1 + aux > m->code_len) FAIL("row %u: skip past the end of the table", pc);
        break;
      case DEMI_OP_SEND: case DEMI_OP_BCAST:
        /* the reference assumes "external message objects never == internal message objects"
           (V/schedulers/ExternalEventInjector.scala:101-104); timers are only ever produced by
           the akka scheduler.  Actors may therefore only `!` INTERNAL-class types. */
        if (aux >= m->n_msg_types || m->msg_class[aux] != DEMI_MSG_INTERNAL)
          FAIL("row %u: SEND/BCAST of a non-internal message type %u", pc, aux);
        break;
      case DEMI_OP_TSET: case DEMI_OP_TREP: case DEMI_OP_TCANCEL:
        if (aux >= m->n_msg_types || m->msg_class[aux] != DEMI_MSG_TIMER)
          FAIL("row %u: timer op on a non-timer message type %u", pc, aux);
        break;
      case DEMI_OP_CRASH:
        break;
      case DEMI_OP_PEER:
        if (!(m->inv_kind & DEMI_INV_PROGRAM) || pc < m->inv_fa) FAIL("PEER outside an invariant program");
        if (aux > 8) FAIL("PEER reads field 0..7 or 8 (created)");
        break;
      case DEMI_OP_LDX: case DEMI_OP_STX:
        if (!DEMI_MODEL_ARRAY_LEN(m->flags)) FAIL("row %u: LDX / STX in a model without DEMI_MODEL_ARRAY", pc);
        break;
      case DEMI_OP_LDP: case DEMI_OP_PSET:
        if (DEMI_MODEL_PAYLOADS_N(m->flags) < 3) FAIL("row %u: LDP / PSET in a model without DEMI_MODEL_PAYLOADS", pc);
        if (op == DEMI_OP_LDP && (!bimm || b >= DEMI_MAX_PAYLOADS)) FAIL("row %u: LDP takes an immediate field index 0..5", pc);
        if (op == DEMI_OP_PSET && (aux < 2 || aux >= DEMI_MAX_PAYLOADS)) FAIL("row %u: PSET stages payload field 2..5", pc);
        if ((m->inv_kind & DEMI_INV_PROGRAM) && pc >= m->inv_fa) FAIL("row %u: LDP / PSET in an invariant program", pc);
        break;
      default:
        FAIL("row %u: unknown op %u", pc, op);
    }
  }
  if ((m->inv_kind & ~DEMI_INV_PROGRAM) > DEMI_INV_AGREE) FAIL("inv_kind invalid");
  if (m->flags & ~(DEMI_MODEL_WIDE | 0xFF00u | 0x70000u)) FAIL("unknown model flags 0x%x", m->flags);
  if ((m->flags >> 16) & 7u) {
    const uint32_t np = (m->flags >> 16) & 7u;
    if (np < 3 || np > DEMI_MAX_PAYLOADS) FAIL("DEMI_MODEL_PAYLOADS: 3..%d fields", DEMI_MAX_PAYLOADS);
    if (!(m->flags & DEMI_MODEL_WIDE)) FAIL("DEMI_MODEL_PAYLOADS needs DEMI_MODEL_WIDE (a 64-bit message word)");
  }
  if (DEMI_MODEL_ARRAY_LEN(m->flags) > DEMI_MAX_ARRAY) FAIL("DEMI_MODEL_ARRAY: at most %d elements", DEMI_MAX_ARRAY);
  if (m->inv_kind & DEMI_INV_PROGRAM) {
    /* the per-actor predicate / key as rows (include/demi_gpu.h): pure rows only, from inv_fa to the end of the table */
    if ((m->inv_kind & 0xFFu) == DEMI_INV_NONE) FAIL("DEMI_INV_PROGRAM needs a combining kind");
    if (m->inv_fa >= m->code_len) FAIL("invariant program starts past the end of the table");
    for (uint32_t pc = m->inv_fa; pc < m->code_len; pc++) {
      const uint32_t op = m->code[pc] & 0xFF;
      if ((op >= DEMI_OP_SEND && op <= DEMI_OP_RND) || op == DEMI_OP_STX)
        FAIL("row %u: an invariant program has no effects and draws no random numbers", pc);
    }
  } else if (m->inv_fa > 7 || m->inv_fb > 7 || m->inv_va > ((m->flags & DEMI_MODEL_WIDE) ? 65535u : 255u)) FAIL("invariant field out of range");
  return DEMI_OK;
}

#undef FAIL
#define FAIL(...)                                  \
  do {                                             \
    if (err) snprintf(err, err_cap, __VA_ARGS__);  \
    return DEMI_ERR_INVALID_TRACE;                 \
  } while (0)

int orc_trace_validate(const demi_model* m, const demi_ext_event* ev, uint32_t n, char* err, size_t err_cap) {
  if (n > DEMI_MAX_EXT_EVENTS) FAIL("more than %d external events", DEMI_MAX_EXT_EVENTS);
  for (uint32_t i = 0; i < n; i++) {
    const demi_ext_event* e = &ev[i];
    switch (e->kind) {
      case DEMI_EV_START: case DEMI_EV_KILL:
        if (e->a >= m->n_actors) FAIL("event %u: actor out of range", i);
        break;
      case DEMI_EV_SEND:
        if (e->a >= m->n_actors) FAIL("event %u: receiver out of range", i);
        /* MessageTypes.sanityCheckTrace analogue (V/ExternalEvents.scala:138-149) */
        if (e->msg_type >= m->n_msg_types || m->msg_class[e->msg_type] != DEMI_MSG_EXTERNAL)
          FAIL("event %u: Send of a non-external message type", i);
        if (!(m->flags & DEMI_MODEL_WIDE) && (e->p0_hi | e->p1_hi))
          FAIL("event %u: 16-bit payload in a model without DEMI_MODEL_WIDE", i);
        break;
      case DEMI_EV_PARTITION: case DEMI_EV_UNPARTITION:
        if (e->a >= m->n_actors || e->b >= m->n_actors) FAIL("event %u: actor out of range", i);
        break;
      case DEMI_EV_WAIT_QUIESCENCE:
        break;
      default:
        FAIL("event %u: unsupported external event kind %u (WaitCondition/CodeBlock/HardKill "
             "need the JVM scheduler)", i, e->kind);
    }
  }
  return DEMI_OK;
}
#undef FAIL

/* ===================================================================== payload area (include/demi_gpu.h DEMI_MODEL_PAYLOADS)
 * A message's payload as ONE value, the way demi_rec_event stores it (p0 | p1 << 16 | p_hi << 32): a narrow model's two bytes
 * sit in bits 0..7 and 16..23; a wide model's area is the 48 bits above the word's header, n fields of DEMI_PAYLOAD_BITS(n). */
static inline uint64_t pay_area(const demi_model* m, const uint16_t* p /* [DEMI_MAX_PAYLOADS] */) {
  if (!(m->flags & DEMI_MODEL_WIDE)) return (uint64_t)(p[0] & 0xFFu) | ((uint64_t)(p[1] & 0xFFu) << 16);
  const uint32_t n = DEMI_MODEL_PAYLOADS_N(m->flags), bits = DEMI_PAYLOAD_BITS(n);
  uint64_t a = 0;
  for (uint32_t k = 0; k < n; k++) a |= (uint64_t)(p[k] & ((1u << bits) - 1u)) << (k * bits);
  return a;
}
static inline uint32_t area_field(const demi_model* m, uint64_t area, uint32_t k) {
  if (!(m->flags & DEMI_MODEL_WIDE)) return k == 0 ? (uint32_t)area & 0xFFu : k == 1 ? (uint32_t)(area >> 16) & 0xFFu : 0u;
  return DEMI_PAYLOAD_OF(area, DEMI_MODEL_PAYLOADS_N(m->flags), k);
}
static inline uint64_t area_of2(const demi_model* m, uint32_t p0, uint32_t p1) {
  const uint16_t p[DEMI_MAX_PAYLOADS] = {(uint16_t)p0, (uint16_t)p1, 0, 0, 0, 0};
  return pay_area(m, p);
}

/* ===================================================================== delta-table VM
 * The application's `receive` (not in the reference; see include/demi_gpu.h for the row
 * format).  Effects are returned in program order; the scheduler applies them in that order,
 * as Akka would call `!` / scheduleOnce / cancel inside receive (WeaveActor.aj:224-279).       */
static int vm_run_at(const demi_model* m, uint32_t start, uint32_t me, uint64_t* state, uint8_t src, uint64_t area,
                     uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app, uint16_t* regs_out, const uint64_t* all_states);
int orc_vm_run_area(const demi_model* m, uint32_t me, uint64_t* state, uint8_t msg_type, uint8_t src,
                    uint64_t area, uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app) {
  uint16_t start = m->handler_start[m->actor_class[me] * m->n_msg_types + msg_type];
  if (start == 0xFFFF) return 0;
  return vm_run_at(m, start, me, state, src, area, exists_mask, fx, fx_cap, app, NULL, NULL);
}
int orc_vm_run(const demi_model* m, uint32_t me, uint64_t* state, uint8_t msg_type, uint8_t src,
               uint16_t p0, uint16_t p1, uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app) {
  return orc_vm_run_area(m, me, state, msg_type, src, area_of2(m, p0, p1), exists_mask, fx, fx_cap, app);
}
uint64_t orc_pay_area(const demi_model* m, const uint16_t* p) { return pay_area(m, p); }
/* the rows from `start` on; regs_out (may be NULL) receives the final register window; all_states (invariant programs only):
 * every actor's state, for DEMI_OP_PEER */
static int vm_run_at(const demi_model* m, uint32_t start, uint32_t me, uint64_t* state, uint8_t src, uint64_t area,
                     uint32_t exists_mask, orc_effect* fx, uint32_t fx_cap, orc_jrandom* app, uint16_t* regs_out, const uint64_t* all_states) {
  uint32_t nfx = 0, n_fx_rows = 0;
  /* the register window: 16 x u8, or 16 x u16 for DEMI_MODEL_WIDE (state = two words, four 16-bit fields each) */
  const int wide = (m->flags & DEMI_MODEL_WIDE) != 0;
  const uint32_t M = wide ? 0xFFFFu : 0xFFu, SH = wide ? 15u : 7u;
  /* the actor's array (DEMI_MODEL_ARRAY): behind its field word(s), `per` elements of `bits` bits to a word */
  const uint32_t arr_len = DEMI_MODEL_ARRAY_LEN(m->flags), per = wide ? 4u : 8u, bits = wide ? 16u : 8u;
  uint64_t* const arr = state + (wide ? 2 : 1);
  uint16_t r[16];
  for (int i = 0; i < 8; i++) r[i] = wide ? (uint16_t)(state[i >> 2] >> (16 * (i & 3))) : (uint16_t)((*state >> (8 * i)) & 0xFF);
  r[8] = r[9] = r[10] = r[11] = 0;
  r[12] = (uint16_t)area_field(m, area, 0); r[13] = (uint16_t)area_field(m, area, 1); r[14] = src; r[15] = (uint16_t)me;
  uint16_t q[DEMI_MAX_PAYLOADS] = {0, 0, 0, 0, 0, 0};   /* DEMI_OP_PSET: the staged payload fields P2..P5 of the messages sent next */
  uint32_t pc = start;
  while (pc < m->code_len) {
    uint32_t w = m->code[pc++];
    uint32_t op = w & 0xFF, dst = (w >> 8) & 15, ai = (w >> 12) & 15, bimm = (w >> 16) & 1;
    uint32_t aux = (w >> 17) & 0x7F, braw = w >> 24;
    uint32_t a = r[ai];
    uint32_t b = bimm ? braw : r[braw & 15];
    if (op == DEMI_OP_HALT) break;
    switch (op) {
      case DEMI_OP_MOV: r[dst] = (uint16_t)b; break;
      case DEMI_OP_MOVHI: r[dst] = (uint16_t)(((a & 0xFFu) | (b << 8)) & M); break;   /* (wide models only: validation) */
      case DEMI_OP_ADD: r[dst] = (uint16_t)((a + b) & M); break;
      case DEMI_OP_SUB: r[dst] = (uint16_t)((a - b) & M); break;
      case DEMI_OP_AND: r[dst] = (uint16_t)(a & b); break;
      case DEMI_OP_OR: r[dst] = (uint16_t)(a | b); break;
      case DEMI_OP_XOR: r[dst] = (uint16_t)(a ^ b); break;
      case DEMI_OP_SHL: r[dst] = (uint16_t)((a << (b & SH)) & M); break;
      case DEMI_OP_SHR: r[dst] = (uint16_t)(a >> (b & SH)); break;
      case DEMI_OP_BITSET: r[dst] = (uint16_t)((a | (1u << (b & SH))) & M); break;
      case DEMI_OP_POPC: r[dst] = (uint16_t)__builtin_popcount(b); break;
      case DEMI_OP_EQ: r[dst] = a == b; break;
      case DEMI_OP_NE: r[dst] = a != b; break;
      case DEMI_OP_LT: r[dst] = a < b; break;
      case DEMI_OP_GE: r[dst] = a >= b; break;
      case DEMI_OP_LE: r[dst] = a <= b; break;
      case DEMI_OP_GT: r[dst] = a > b; break;
      case DEMI_OP_MIN: r[dst] = a < b ? a : b; break;
      case DEMI_OP_MAX: r[dst] = a > b ? a : b; break;
      case DEMI_OP_RND: /* Instrumenter().seededRandom.nextInt(bound) (V/Instrumenter.scala:212, 226-229) */
        r[dst] = (b & 0xFFu) ? (uint16_t)orc_jrandom_next_int_bound(app, (int32_t)(b & 0xFFu)) : 0;   /* bound = b & 0xFF */
        break;
      case DEMI_OP_LDX: r[dst] = b < arr_len ? (uint16_t)((arr[b / per] >> (bits * (b % per))) & M) : 0; break;
      case DEMI_OP_STX:
        if (b < arr_len) arr[b / per] = (arr[b / per] & ~((uint64_t)M << (bits * (b % per)))) | ((uint64_t)a << (bits * (b % per)));
        break;
      case DEMI_OP_PEER: {    /* invariant programs: field aux (8 = "is created") of actor a, 0 when a is not a created actor */
        uint32_t v = 0;
        if (all_states && a < m->n_actors && ((exists_mask >> a) & 1)) {
          const uint32_t stw_ = (wide ? 2u : 1u) + (arr_len + per - 1) / per;
          const uint64_t* ps = all_states + (size_t)stw_ * a;
          v = aux >= 8 ? 1u : wide ? (uint32_t)(ps[aux >> 2] >> (16 * (aux & 3))) & 0xFFFFu : (uint32_t)(ps[0] >> (8 * aux)) & 0xFFu;
        }
        r[dst] = (uint16_t)v;
        break;
      }
      case DEMI_OP_LDP: r[dst] = (uint16_t)area_field(m, area, b); break;
      case DEMI_OP_PSET: if (aux >= 2 && aux < DEMI_MAX_PAYLOADS) q[aux] = (uint16_t)b; break;
      case DEMI_OP_SKIPZ: if (a == 0) pc += braw; break;
      case DEMI_OP_SKIPNZ: if (a != 0) pc += braw; break;
      case DEMI_OP_SKIP: pc += braw; break;
      case DEMI_OP_IFEQ: if (!(a == b)) pc += aux; break;
      case DEMI_OP_IFNE: if (!(a != b)) pc += aux; break;
      case DEMI_OP_IFLT: if (!(a < b)) pc += aux; break;
      case DEMI_OP_IFGE: if (!(a >= b)) pc += aux; break;
      case DEMI_OP_IFLE: if (!(a <= b)) pc += aux; break;
      case DEMI_OP_IFGT: if (!(a > b)) pc += aux; break;
      case DEMI_OP_SEND:
        if (++n_fx_rows > DEMI_FX_CAP) return -1;
        /* a message to a name that was never created reaches no scheduler (deadLetters) */
        if (a < m->n_actors && ((exists_mask >> a) & 1)) {
          if (nfx >= fx_cap) return -1;
          q[0] = r[dst]; q[1] = (uint16_t)b;
          fx[nfx++] = (orc_effect){0, (uint8_t)a, (uint8_t)aux, r[dst], (uint16_t)b, pay_area(m, q)};
        }
        break;
      case DEMI_OP_BCAST:
        if (++n_fx_rows > DEMI_FX_CAP) return -1;
        for (uint32_t j = 0; j < m->n_actors; j++) {
          if (j == me || !((exists_mask >> j) & 1)) continue;
          if (nfx >= fx_cap) return -1;
          q[0] = r[dst]; q[1] = (uint16_t)b;
          fx[nfx++] = (orc_effect){0, (uint8_t)j, (uint8_t)aux, r[dst], (uint16_t)b, pay_area(m, q)};
        }
        break;
      case DEMI_OP_TSET: case DEMI_OP_TREP: case DEMI_OP_TCANCEL:
        if (++n_fx_rows > DEMI_FX_CAP) return -1;
        if (nfx >= fx_cap) return -1;
        fx[nfx++] = (orc_effect){(uint8_t)(1 + (op - DEMI_OP_TSET)), (uint8_t)me, (uint8_t)aux, 0, 0, 0};
        break;
      case DEMI_OP_CRASH: /* the receive throws (V/Instrumenter.scala:184-199): recorded as the last effect, rows stop */
        if (++n_fx_rows > DEMI_FX_CAP) return -1;
        if (nfx >= fx_cap) return -1;
        fx[nfx++] = (orc_effect){4, (uint8_t)me, 0, 0, 0, 0};
        pc = m->code_len;
        break;
      default: break;
    }
  }
  if (wide) {
    state[0] = (uint64_t)r[0] | ((uint64_t)r[1] << 16) | ((uint64_t)r[2] << 32) | ((uint64_t)r[3] << 48);
    state[1] = (uint64_t)r[4] | ((uint64_t)r[5] << 16) | ((uint64_t)r[6] << 32) | ((uint64_t)r[7] << 48);
  } else {
    uint64_t s = 0;
    for (int i = 0; i < 8; i++) s |= (uint64_t)r[i] << (8 * i);
    *state = s;
  }
  if (regs_out) memcpy(regs_out, r, sizeof r);
  return (int)nfx;
}

/* ===================================================================== invariant
 * `Invariant` closure (V/minification/TestOracle.scala:27) as a descriptor; evaluated on the
 * simulated actor state instead of CheckpointReply maps (checkpointing is off by default,
 * V/SchedulerConfig.scala:11-12).  Returns the ViolationFingerprint code.                      */
/* field f of actor i: 8 bits of its one state word, or 16 bits of its two (DEMI_MODEL_WIDE) */
static inline uint32_t fldw(int wide, const uint64_t* st, uint32_t stw, uint32_t i, uint32_t f) {
  return wide ? (uint32_t)(st[stw * i + (f >> 2)] >> (16 * (f & 3))) & 0xFFFFu : (uint32_t)(st[stw * i] >> (8 * f)) & 0xFFu;
}

/* per actor: does it count ("hit") and under which key.  Descriptor: F[fa] == va (AGREE: F[fa] != 0), key F[fb].
 * DEMI_INV_PROGRAM: the rows from inv_fa on, run on a COPY of the actor's state with r15 = its id and everything else 0;
 * hit = T0 != 0, key = T1 (include/demi_gpu.h). */
static void inv_actor(const demi_model* m, const uint64_t* st, uint32_t i, uint32_t* hit, uint32_t* key, uint32_t exists) {
  const int wide = (m->flags & DEMI_MODEL_WIDE) != 0;
  const uint32_t stw = model_stw(m);
  if (m->inv_kind & DEMI_INV_PROGRAM) {
    uint64_t copy[ORC_MAX_STW];
    memcpy(copy, st + (size_t)stw * i, sizeof(uint64_t) * stw);
    uint16_t r[16];
    orc_jrandom none;
    orc_jrandom_seed(&none, 0);
    vm_run_at(m, m->inv_fa, i, copy, 0, 0, exists, NULL, 0, &none, r, st);
    *hit = r[8] != 0;
    *key = r[9];
    return;
  }
  const uint32_t a = fldw(wide, st, stw, i, m->inv_fa);
  *hit = ((m->inv_kind & 0xFFu) == DEMI_INV_AGREE) ? (a != 0) : (a == m->inv_va);
  *key = fldw(wide, st, stw, i, m->inv_fb);
}

uint32_t orc_invariant(const demi_model* m, const uint64_t* st, uint32_t exists) {
  const uint32_t A = m->n_actors;
  uint32_t hit[ORC_MAX_ACTORS], key[ORC_MAX_ACTORS], hits = 0;
  /* fingerprint layouts: kind << 24 | key << 8 | actors (8 bits); big tables: kind << 30 | (key & 0x3FFF) << 16 | actors (16 bits) */
  const int big = model_big(m);
  if ((m->inv_kind & 0xFFu) == DEMI_INV_NONE) return 0;
  for (uint32_t i = 0; i < A; i++) {
    hit[i] = 0; key[i] = 0;
    if ((exists >> i) & 1) inv_actor(m, st, i, &hit[i], &key[i], exists);
    hits |= hit[i] << i;
  }
  switch (m->inv_kind & 0xFFu) {
    case DEMI_INV_AT_MOST_ONE:
      for (uint32_t i = 0; i < A; i++) {
        if (!hit[i]) continue;
        for (uint32_t j = i + 1; j < A; j++) {
          if (!hit[j] || key[i] != key[j]) continue;
          uint32_t mask = 0;
          for (uint32_t k = 0; k < A; k++)
            if (hit[k] && key[k] == key[i]) mask |= 1u << k;
          return big ? (1u << 30) | ((key[i] & 0x3FFFu) << 16) | mask : (1u << 24) | (key[i] << 8) | mask;
        }
      }
      return 0;
    case DEMI_INV_NEVER:
      return hits ? (big ? 2u << 30 : 2u << 24) | hits : 0;
    case DEMI_INV_AGREE: {
      uint32_t first = 0xFFFFFFFFu, bad = 0;
      for (uint32_t k = 0; k < A; k++) {
        if (!hit[k]) continue;
        if (first == 0xFFFFFFFFu) first = key[k];
        else if (key[k] != first) bad = 1;
      }
      return bad ? (big ? 3u << 30 : 3u << 24) | hits : 0;
    }
    default:
      return 0;
  }
}

/* ===================================================================== one execution */
#define PEND_HARD_CAP DEMI_MAX_PENDING
#define MTS_CAP 512

typedef struct { uint64_t word; uint32_t id; } pend_entry;

/* message word: type[4:0] | dst[7:5] | src[11:8] | p0[23:16] | p1[31:24]; DEMI_MODEL_WIDE: p0[31:16] | p1[47:32] */
/* big tables (more than 8 actors): type[4:0] | dst[8:5] | src[13:9] | area[63:16] */
static inline uint32_t msg_word(uint32_t type, uint32_t src, uint32_t dst, uint32_t p0, uint32_t p1) {
  if (g_big) return type | (dst << 5) | (src << 9) | (p0 << 16);      /* (only ever called with p0 = p1 = 0 for a wide table) */
  return type | (dst << 5) | (src << 8) | (p0 << 16) | (p1 << 24);
}
#define W_TYPE(w) ((uint32_t)(w) & 31u)
#define W_DST(w) (g_big ? (((uint32_t)(w) >> 5) & 15u) : (((uint32_t)(w) >> 5) & 7u))
#define W_SRC(w) (g_big ? (((uint32_t)(w) >> 9) & 31u) : (((uint32_t)(w) >> 8) & 15u))
#define W_P0(w) (((w) >> 16) & 255u)
#define W_P1(w) ((w) >> 24)
/* the word of a message whose payload is `area`, and the area of a word (see "payload area" above) */
static inline uint64_t msg_word_a(int wide, uint32_t type, uint32_t src, uint32_t dst, uint64_t area) {
  if (!wide) return msg_word(type, src, dst, (uint32_t)area & 0xFFu, (uint32_t)(area >> 16) & 0xFFu);
  return (uint64_t)(type | (dst << 5) | (src << (g_big ? 9 : 8))) | (area << 16);
}
#define WX_AREA(wide, w) ((wide) ? (uint64_t)(w) >> 16 : (uint64_t)(W_P0(w) | (W_P1(w) << 16)))

typedef struct { uint8_t rcv, type; uint64_t area; uint8_t is_external, ext_idx; } mts_entry; /* messagesToSend */

typedef struct {
  const demi_model* m;
  const demi_ext_event* trace;
  uint32_t n_ev;
  const demi_limits* lim;
  orc_jrandom rng;
  int wide;                             /* DEMI_MODEL_WIDE */
  uint64_t state[ORC_MAX_STW * ORC_MAX_ACTORS];  /* model_stw words per actor: field word(s), then the array */
  uint32_t exists, inaccessible, killed;
  uint32_t blocked;     /* Instrumenter().blockedActors (crashed actors), V/Instrumenter.scala:116, 184-199 */
  orc_pairset partitioned; /* ordered pair (a,b), V/schedulers/EventOrchestrator.scala:51 */
  uint32_t trace_idx;
  pend_entry pend[PEND_HARD_CAP]; /* RandomizedHashSet.arr, V/schedulers/Util.scala:112 (SrcDstFIFO: timersAndExternals) */
  uint32_t n_pend, p_max;
  /* SrcDstFIFO (V/schedulers/RandomScheduler.scala:702-909): the actor-to-actor messages in arrival order (the
   * per-pair queues are the sub-sequences with equal (src, dst)), srcDsts, the second generator */
  int fifo;
  pend_entry norm[PEND_HARD_CAP];
  uint32_t n_norm;
  uint8_t pairs[256];             /* srcDsts: src * 16 + dst, in queue-creation order */
  uint32_t n_pairs;
  orc_jrandom te_rng;             /* timersAndExternals' RandomizedHashSet generator */
  orc_jrandom app_rng;            /* Instrumenter().seededRandom = scala.util.Random(0), new per ActorSystem (V/Instrumenter.scala:226-229) */
  mts_entry mts[MTS_CAP]; /* messagesToSend, V/schedulers/ExternalEventInjector.scala:109 */
  uint32_t n_mts;
  uint32_t n_mts_timers;  /* timers among them; capacity DEMI_TQ_CAP is part of the spec */
  uint64_t just_scheduled; /* justScheduledTimers, V/schedulers/RandomScheduler.scala:109 */
  uint64_t repeating;      /* timerToCancellable of ongoing timers, V/Instrumenter.scala:141 */
  uint8_t resend[DEMI_RESEND_CAP][2]; /* timersToResend (rcv, type), RandomScheduler.scala:113 */
  uint32_t n_resend;
  uint32_t count;     /* messagesScheduledSoFar */
  uint32_t violation; /* violationFound fingerprint code, 0 = None */
  uint32_t flags;
  uint32_t next_id;   /* Uniq ids of produced messages, per execution, from 1 */
  uint64_t hash;
  demi_rec_event* rec;
  uint32_t rec_cap, n_rec;
  orc_effect fx[DEMI_MAX_CODE * ORC_MAX_ACTORS];
} exec_t;

static void rec_push(exec_t* x, uint8_t kind, uint8_t snd, uint8_t rcv, uint8_t type, uint64_t area,
                     uint8_t flags, uint8_t ext_idx, uint32_t id) {
  if (!x->rec) return;
  if (x->n_rec < x->rec_cap) {
    demi_rec_event* e = &x->rec[x->n_rec];
    e->kind = kind; e->snd = snd; e->rcv = rcv; e->msg_type = type; e->p0 = (uint16_t)area; e->p1 = (uint16_t)(area >> 16);
    e->flags = flags; e->ext_idx = ext_idx; e->p_hi = (uint16_t)(area >> 32); e->id = id;
  }
  x->n_rec++;
}

/* V/schedulers/EventOrchestrator.scala:345-351 */
static int crosses_partition(const exec_t* x, uint32_t snd, uint32_t rcv) {
  if (snd == rcv && !((x->killed >> snd) & 1)) return 0;
  int part = 0;
  if (ORC_IS_ACTOR(snd) && ORC_IS_ACTOR(rcv))
    part = ps_get(&x->partitioned, snd, rcv) | ps_get(&x->partitioned, rcv, snd);
  int inacc_r = ORC_IS_ACTOR(rcv) ? (int)((x->inaccessible >> rcv) & 1) : 0;
  int inacc_s = ORC_IS_ACTOR(snd) ? (int)((x->inaccessible >> snd) & 1) : 0;
  return part || inacc_r || inacc_s;
}

/* RandomizedHashSet.insert, V/schedulers/Util.scala:126-136 */
#define OVF_ANY (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)
static void pend_insert(exec_t* x, uint64_t word, uint32_t id) {
  if (x->flags & OVF_ANY) return; /* only the first capacity overflow is reported */
  if (x->n_pend + x->n_norm >= x->p_max) { x->flags |= DEMI_V_PENDING_OVF; return; }
  if (x->fifo && W_SRC(word) != ORC_DL) {
    /* SrcDstFIFO.+= (:791-811): append to the pair's queue, creating it (and its srcDsts entry) if absent */
    uint8_t pair = (uint8_t)(W_SRC(word) * 16 + W_DST(word));
    int have = 0;
    for (uint32_t i = 0; i < x->n_pairs; i++) have |= (x->pairs[i] == pair);
    if (!have) x->pairs[x->n_pairs++] = pair;
    x->norm[x->n_norm].word = word;
    x->norm[x->n_norm].id = id;
    x->n_norm++;
    return;
  }
  x->pend[x->n_pend].word = word;
  x->pend[x->n_pend].id = id;
  x->n_pend++;
}

/* SrcDstFIFO.dequeue (:764-774): the head of the pair's queue; the queue and its srcDsts entry go when it empties */
static pend_entry fifo_dequeue(exec_t* x, uint32_t pi) {
  const uint8_t pair = x->pairs[pi];
  uint32_t k = 0;
  while (W_SRC(x->norm[k].word) * 16 + W_DST(x->norm[k].word) != pair) k++;
  pend_entry v = x->norm[k];
  memmove(&x->norm[k], &x->norm[k + 1], (x->n_norm - k - 1) * sizeof(pend_entry));
  x->n_norm--;
  int more = 0;
  for (uint32_t i = k; i < x->n_norm; i++) more |= (W_SRC(x->norm[i].word) * 16 + W_DST(x->norm[i].word) == pair);
  if (!more) {
    memmove(&x->pairs[pi], &x->pairs[pi + 1], x->n_pairs - pi - 1);   /* ArrayList.remove(idx) */
    x->n_pairs--;
  }
  return v;
}

/* RandomizedHashSet.remove: swap with last, V/schedulers/Util.scala:146-163 */
static pend_entry pend_remove_at(exec_t* x, uint32_t i) {
  pend_entry v = x->pend[i];
  x->pend[i] = x->pend[x->n_pend - 1];
  x->n_pend--;
  return v;
}

/* ExternalEventInjector.handle_timer, V/schedulers/ExternalEventInjector.scala:282-297 */
static void handle_timer(exec_t* x, uint32_t rcv, uint32_t type) {
  if (x->flags & OVF_ANY) return;
  if (x->n_mts >= MTS_CAP || x->n_mts_timers >= DEMI_TQ_CAP) { x->flags |= DEMI_V_QUEUE_OVF; return; }
  x->mts[x->n_mts++] = (mts_entry){(uint8_t)rcv, (uint8_t)type, 0, 0, 255};
  x->n_mts_timers++;
}

/* RandomScheduler.enqueue_timer, V/schedulers/RandomScheduler.scala:549-559 */
static void enqueue_timer(exec_t* x, uint32_t rcv, uint32_t type) {
  if (x->flags & OVF_ANY) return;
  if (x->just_scheduled & timer_bit(x->m, rcv, type)) {
    if (x->n_resend >= DEMI_RESEND_CAP) { x->flags |= DEMI_V_QUEUE_OVF; return; }
    x->resend[x->n_resend][0] = (uint8_t)rcv;
    x->resend[x->n_resend][1] = (uint8_t)type;
    x->n_resend++;
    return;
  }
  handle_timer(x, rcv, type);
}

/* Instrumenter.registerCancellable + handleTick, V/Instrumenter.scala:1145-1200
 * (WeaveActor.aj:234-279: scheduleOnce -> ongoing=false, schedule -> ongoing=true).            */
static void register_cancellable(exec_t* x, int ongoing, uint32_t rcv, uint32_t type) {
  uint64_t bit = timer_bit(x->m, rcv, type);
  if (x->repeating & bit) return; /* "Non-unique timer" (:1154-1157) */
  if (ongoing) x->repeating |= bit;
  enqueue_timer(x, rcv, type);    /* "Schedule it immediately!" */
  /* one-shot: removeCancellable right after the tick (:1194-1196), i.e. never registered */
}

/* Instrumenter.cancelTimer (V/Instrumenter.scala:159-168) -> RandomScheduler.notify_timer_cancel
 * (V/schedulers/RandomScheduler.scala:525-534) -> handle_timer_cancel
 * (V/schedulers/ExternalEventInjector.scala:601-610) / FullyRandom.remove (:653-664).          */
static void cancel_timer(exec_t* x, uint32_t rcv, uint32_t type) {
  x->repeating &= ~timer_bit(x->m, rcv, type);
  for (uint32_t i = 0; i < x->n_mts; i++) {
    if (!x->mts[i].is_external && x->mts[i].rcv == rcv && x->mts[i].type == type) {
      memmove(&x->mts[i], &x->mts[i + 1], (x->n_mts - i - 1) * sizeof(mts_entry));
      x->n_mts--;
      x->n_mts_timers--;
      return;
    }
  }
  for (uint32_t i = 0; i < x->n_pend; i++) {
    uint64_t w = x->pend[i].word;
    if (W_SRC(w) == ORC_DL && W_DST(w) == rcv && W_TYPE(w) == type && (w >> 16) == 0) {
      pend_remove_at(x, i);
      return;
    }
  }
}

/* RandomScheduler.event_produced(cell, envelope), V/schedulers/RandomScheduler.scala:274-321 */
static void event_produced(exec_t* x, uint32_t snd, uint32_t rcv, uint32_t type, uint64_t area,
                           int is_external, uint8_t ext_idx) {
  uint32_t id = x->next_id++;
  int is_timer = 0, dropped = 0;
  if (!is_external) {
    if (snd == ORC_DL) is_timer = 1;
    if (!crosses_partition(x, snd, rcv)) pend_insert(x, msg_word_a(x->wide, type, snd, rcv, area), id);
    else dropped = 1;
  } else {
    pend_insert(x, msg_word_a(x->wide, type, snd, rcv, area), id); /* externals: no partition check (:298-308) */
  }
  rec_push(x, DEMI_REC_MSG_SEND, (uint8_t)snd, (uint8_t)rcv, (uint8_t)type, area,
           (uint8_t)((is_external ? 1 : 0) | (is_timer ? 2 : 0) | (dropped ? 4 : 0)), ext_idx, id);
}

/* ExternalEventInjector.send_external_messages, V/schedulers/ExternalEventInjector.scala:306-365 */
static void send_external_messages(exec_t* x) {
  for (uint32_t i = 0; i < x->n_mts; i++) {
    mts_entry* e = &x->mts[i];
    event_produced(x, ORC_DL, e->rcv, e->type, e->area, e->is_external, e->ext_idx);
  }
  x->n_mts = 0;
  x->n_mts_timers = 0;
}

/* EventOrchestrator.inject_until_quiescence, V/schedulers/EventOrchestrator.scala:132-189 */
static void inject_until_quiescence(exec_t* x) {
  int loop = 1;
  while (loop && x->trace_idx < x->n_ev) {
    const demi_ext_event* e = &x->trace[x->trace_idx];
    uint8_t idx = (uint8_t)x->trace_idx;
    switch (e->kind) {
      case DEMI_EV_START: /* trigger_start :219-231 -> unisolate_node :211-217 */
        rec_push(x, DEMI_REC_SPAWN, 0, e->a, 0, 0, 0, idx, 0);
        x->inaccessible &= ~(1u << e->a);
        x->killed &= ~(1u << e->a);
        x->blocked &= ~(1u << e->a);     /* "allow scheduler to send messages to it again" (:224-227) */
        break;
      case DEMI_EV_KILL: /* trigger_kill :233-241 */
        rec_push(x, DEMI_REC_KILL, 0, e->a, 0, 0, 0, idx, 0);
        x->killed |= 1u << e->a;
        x->inaccessible |= 1u << e->a;
        break;
      case DEMI_EV_SEND: /* enqueue_message, V/schedulers/ExternalEventInjector.scala:250-279 */
        if ((x->exists >> e->a) & 1) {
          if (x->n_mts >= MTS_CAP) { x->flags |= DEMI_V_QUEUE_OVF; break; }
          x->mts[x->n_mts++] = (mts_entry){e->a, e->msg_type, ORC_EXT_AREA(x->m, e, x->trace_idx), 1, idx};
        } /* else: "Unknown message receiver" (:254) */
        break;
      case DEMI_EV_PARTITION: /* trigger_partition :314-322 */
        rec_push(x, DEMI_REC_PARTITION, e->a, e->b, 0, 0, 0, idx, 0);
        ps_set(&x->partitioned, e->a, e->b);
        break;
      case DEMI_EV_UNPARTITION: /* trigger_unpartition :324-332 (ordered pair as given) */
        rec_push(x, DEMI_REC_UNPARTITION, e->a, e->b, 0, 0, 0, idx, 0);
        ps_clear(&x->partitioned, e->a, e->b);
        break;
      case DEMI_EV_WAIT_QUIESCENCE: /* :182-184 */
        rec_push(x, DEMI_REC_BEGIN_WAIT_QUIESCENCE, 0, 0, 0, 0, 0, idx, 0);
        loop = 0;
        break;
      default: break;
    }
    x->trace_idx++; /* trace_advanced :186 */
  }
}

/* test_invariant + violationMatches, V/schedulers/RandomScheduler.scala:138-154 */
static uint32_t check_invariant(exec_t* x) {
  uint32_t fp = orc_invariant(x->m, x->state, x->exists);
  if (!fp) return 0;
  if (x->lim->looking_for_valid) {
    if (((fp ^ x->lim->looking_for) & x->m->fp_match_mask) == 0) return x->lim->looking_for;
    return 0;
  }
  return fp;
}

static inline void hash_step(uint64_t* h, uint64_t v) { *h = (*h ^ v) * 0x100000001B3ULL; }

/* Apply delta to the delivered message and its effects in program order. */
static void deliver(exec_t* x, uint64_t word) {
  uint32_t me = W_DST(word);
  orc_effect* fx = x->fx; /* DEMI_MAX_CODE rows x at most DEMI_MAX_ACTORS effects each: never full */
  int n = orc_vm_run_area(x->m, me, &x->state[model_stw(x->m) * me], (uint8_t)W_TYPE(word), (uint8_t)W_SRC(word),
                          WX_AREA(x->wide, word), x->exists, fx, DEMI_MAX_CODE * ORC_MAX_ACTORS, &x->app_rng);
  if (n < 0) { x->flags |= DEMI_V_QUEUE_OVF; return; }
  for (int i = 0; i < n; i++) {
    switch (fx[i].kind) {
      case 0: event_produced(x, me, fx[i].target, fx[i].msg_type, fx[i].area, 0, 255); break;
      case 1: register_cancellable(x, 0, me, fx[i].msg_type); break;
      case 2: register_cancellable(x, 1, me, fx[i].msg_type); break;
      case 3: cancel_timer(x, me, fx[i].msg_type); break;
      case 4: x->blocked |= 1u << me; break;    /* actorCrashed: blockedActors += name */
    }
  }
}

/* Util.find_non_blocked_message (V/schedulers/Util.scala:470-489) over a RandomizedHashSet: draw until the receiver is
 * not blocked; the rejected elements are re-appended in draw order (which permutes arr).  Returns 0 when every pending
 * message is for a blocked actor (all of them re-appended, `None`). */
static int find_non_blocked(exec_t* x, orc_jrandom* rng, pend_entry* out) {
  pend_entry rejected[PEND_HARD_CAP];
  uint32_t n_rej = 0;
  int found = 0;
  while (x->n_pend > 0) {
    pend_entry e = pend_remove_at(x, (uint32_t)orc_jrandom_next_int_bound(rng, (int32_t)x->n_pend));
    if ((x->blocked >> W_DST(e.word)) & 1) { rejected[n_rej++] = e; continue; }
    *out = e;
    found = 1;
    break;
  }
  for (uint32_t i = 0; i < n_rej; i++) x->pend[x->n_pend++] = rejected[i];      /* collection ++= blocked */
  return found;
}

/* RandomScheduler.schedule_new_message (V/schedulers/RandomScheduler.scala:352-485) fused with the
 * Instrumenter dispatch it feeds (V/Instrumenter.scala:913-1017).  Returns 0 on `None`.        */
static int schedule_new_message(exec_t* x) {
  uint32_t max_messages = x->lim->max_messages ? x->lim->max_messages : 0x7FFFFFFFu;
  uint32_t interval = x->lim->invariant_check_interval;
  if (x->violation) return 0;                                   /* :354-360 */
  if (x->count > max_messages) {                                /* :369-373 */
    x->flags |= DEMI_V_MAXMSG;
    x->trace_idx = x->n_ev;                                     /* finish_early */
    return 0;
  }
  if (interval > 0 && (x->count % interval) == 0 && x->count != 0 /* lastCheckpoint == 0 */) {
    x->violation = check_invariant(x);                          /* :376-394 */
    if (x->violation) return 0;
  }
  send_external_messages(x);                                    /* :424 */
  if (x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)) return 0;
  if (x->n_pend + x->n_norm == 0) return 0;                     /* find_non_blocked_message, Util.scala:474 */
  pend_entry e;
  if (!x->fifo) {
    /* Util.find_non_blocked_message over FullyRandom.removeRandomElement (:443-458, :666-684) ->
       RandomizedHashSet.removeRandomElement (V/schedulers/Util.scala:171-176) */
    if (!find_non_blocked(x, &x->rng, &e)) return 0;
  } else {
    /* SrcDstFIFO.getNonBlockedMessage (:716-760) */
    int open_pair = 0;                      /* a pair queue whose receiver is not blocked */
    for (uint32_t i = 0; i < x->n_pairs; i++) open_pair |= !((x->blocked >> (x->pairs[i] & 15)) & 1);
    if (!open_pair) {
      /* (:717-729) "only timers left" */
      if (!find_non_blocked(x, &x->te_rng, &e)) return 0;
    } else {
      /* (:731-759) first decide whether to deliver a timer / external, in proportion to their share */
      int timer = 0;
      if ((uint32_t)orc_jrandom_next_int_bound(&x->rng, (int32_t)(x->n_pend + x->n_norm)) < x->n_pend)
        timer = find_non_blocked(x, &x->te_rng, &e);
      if (!timer) {
        uint32_t pi = (uint32_t)orc_jrandom_next_int_bound(&x->rng, (int32_t)x->n_pairs);
        while ((x->blocked >> (x->pairs[pi] & 15)) & 1) pi = (uint32_t)orc_jrandom_next_int_bound(&x->rng, (int32_t)x->n_pairs);
        e = fifo_dequeue(x, pi);
      }
    }
  }
  x->count++;                                                   /* :462 */
  uint64_t w = e.word;
  rec_push(x, DEMI_REC_MSG_EVENT, (uint8_t)W_SRC(w), (uint8_t)W_DST(w), (uint8_t)W_TYPE(w), WX_AREA(x->wide, w), 0, 255, e.id);
  hash_step(&x->hash, w);
  /* updateRepeatingTimer :405-421; isTimer = timerToCancellable contains (rcv, msg) */
  int is_rep = x->m->msg_class[W_TYPE(w)] == DEMI_MSG_TIMER &&
               (x->repeating & timer_bit(x->m, W_DST(w), W_TYPE(w))) != 0;
  if (is_rep) {
    x->just_scheduled |= timer_bit(x->m, W_DST(w), W_TYPE(w));
  } else {
    for (uint32_t i = 0; i < x->n_resend; i++) handle_timer(x, x->resend[i][0], x->resend[i][1]);
    x->n_resend = 0;
    x->just_scheduled = 0;
  }
  /* dispatch_new_message: "Check if it was a repeating timer. If so, retrigger it"
     (V/Instrumenter.scala:1008-1016).  The retrigger races with the actor's receive in the
     reference; this restatement pins it BEFORE the receive. */
  if (is_rep) enqueue_timer(x, W_DST(w), W_TYPE(w));
  deliver(x, w);
  return 1;
}

/* carried: NULL, or the generators an earlier execution of the same RandomScheduler instance left behind ([0] FullyRandom's /
   SrcDstFIFO.rand, [1] SrcDstFIFO's timersAndExternals): reset_all_state only clears the pending containers, it does not
   reseed them (V/schedulers/RandomScheduler.scala:575-595, 649-651).  On return they hold this execution's final states. */
static int random_execute_in2(exec_t* x, const demi_model* m, const demi_ext_event* trace, uint32_t n_ev,
                              uint64_t seed, const demi_limits* lim, demi_verdict* out, demi_rec_event* rec,
                              uint32_t rec_cap, uint32_t* n_rec, uint64_t* final_states, orc_jrandom* carried, int carried_valid);
static int random_execute_in(exec_t* x, const demi_model* m, const demi_ext_event* trace, uint32_t n_ev,
                             uint64_t seed, const demi_limits* lim, demi_verdict* out, demi_rec_event* rec,
                             uint32_t rec_cap, uint32_t* n_rec, uint64_t* final_states) {
  return random_execute_in2(x, m, trace, n_ev, seed, lim, out, rec, rec_cap, n_rec, final_states, NULL, 0);
}
static int random_execute_in2(exec_t* x, const demi_model* m, const demi_ext_event* trace, uint32_t n_ev,
                              uint64_t seed, const demi_limits* lim, demi_verdict* out, demi_rec_event* rec,
                              uint32_t rec_cap, uint32_t* n_rec, uint64_t* final_states, orc_jrandom* carried, int carried_valid) {
  memset(x, 0, offsetof(exec_t, fx));
  g_big = model_big(m);
  x->m = m; x->trace = trace; x->n_ev = n_ev; x->lim = lim;
  x->wide = (m->flags & DEMI_MODEL_WIDE) != 0;
  x->rec = rec; x->rec_cap = rec_cap;
  x->p_max = lim->p_max ? lim->p_max : 64;
  if (x->p_max > PEND_HARD_CAP) x->p_max = PEND_HARD_CAP;
  orc_jrandom_seed(&x->rng, seed); /* new FullyRandom(seed = ...) / SrcDstFIFO.rand */
  orc_jrandom_seed(&x->te_rng, seed);
  if (carried && carried_valid) { x->rng = carried[0]; x->te_rng = carried[1]; }   /* a later execution of the same instance */
  orc_jrandom_seed(&x->app_rng, 0);
  x->fifo = lim->strategy == DEMI_STRATEGY_SRC_DST_FIFO;
  /* populateActorSystem (V/schedulers/ExternalEventInjector.scala:371-378, 397-406):
     every actor that is ever Start()ed is created up front and isolated. */
  for (uint32_t i = 0; i < n_ev; i++)
    if (trace[i].kind == DEMI_EV_START) x->exists |= 1u << trace[i].a;
  if (lim->populate_all) x->exists = (1u << m->n_actors) - 1;
  x->inaccessible = x->exists;
  const uint32_t n_state = m->n_actors * model_stw(m);
  state_init(m, x->state);
  x->next_id = 1;
  x->hash = 0xCBF29CE484222325ULL;

  /* execute_trace -> advanceTrace -> start_dispatch ... handle_quiescence
     (V/schedulers/ExternalEventInjector.scala:382-441, 541-580) */
  for (;;) {
    inject_until_quiescence(x);
    while (schedule_new_message(x)) {
      if (x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)) break;
    }
    if (x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)) break;
    if (x->violation) break;                 /* notify_quiescence :487-500 */
    if (x->trace_idx < n_ev) {
      rec_push(x, DEMI_REC_QUIESCENCE, 0, 0, 0, 0, 0, 255, 0);
      continue;
    }
    break;
  }
  /* explore(): `if (messagesScheduledSoFar <= maxMessages) checkIfBugFound` (:256-262, 156-180) */
  if (!(x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF | DEMI_V_MAXMSG)) && !x->violation)
    x->violation = check_invariant(x);
  for (uint32_t a = 0; a < n_state; a++) hash_step(&x->hash, x->state[a]);

  if (x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)) {
    /* capacity abort: only the overflow bits are defined (the schedule must be re-run on the
       JVM scheduler or with a larger p_max) */
    out->flags = x->flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF);
    out->fingerprint = 0;
    out->hash = 0;
  } else {
    out->flags = (x->flags & 0xFF) | (x->violation ? DEMI_V_VIOLATION : 0) | ((x->trace_idx & 0xFF) << 8) |
                 ((x->count < 0xFFFFu ? x->count : 0xFFFFu) << 16);
    out->fingerprint = x->violation;
    out->hash = x->hash;
  }
  if (n_rec) *n_rec = x->n_rec;
  if (final_states) memcpy(final_states, x->state, sizeof(uint64_t) * n_state);
  if (carried) { carried[0] = x->rng; carried[1] = x->te_rng; }
  return DEMI_OK;
}

int orc_random_execute(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed,
                       const demi_limits* lim, demi_verdict* out, demi_rec_event* rec, uint32_t rec_cap,
                       uint32_t* n_rec, uint64_t* final_states) {
  exec_t* x = (exec_t*)malloc(sizeof(exec_t));
  if (!x) return DEMI_ERR_INVALID_ARG;
  int rc = random_execute_in(x, m, trace, n_ev, seed, lim, out, rec, rec_cap, n_rec, final_states);
  free(x);
  return rc;
}

/* Execution number `exec_index` of the instance seeded `seed` (demi_limits.executions_per_instance mode), recorded: the chain
   is run from its first execution; *ran = the execution the instance stopped at (exec_index, or an earlier violating one). */
int orc_random_execute_carried(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed, uint32_t exec_index,
                               const demi_limits* lim, demi_verdict* out, demi_rec_event* rec, uint32_t rec_cap,
                               uint32_t* n_rec, uint32_t* ran);

/* ===================================================================== batch driver */
typedef struct {
  const demi_model* m; const demi_ext_event* trace; uint32_t n_ev; uint64_t seed_base;
  const uint64_t* seeds; uint64_t lo, hi; const demi_limits* lim; demi_verdict* out; uint64_t n;
} job_t;

/* One RandomScheduler instance with max_executions = k (explore(), V/schedulers/RandomScheduler.scala:248-269): executions
   first .. first + k - 1 of the verdict array (clipped to n), the generators carried from one to the next, lookingFor only
   for the first (reset_all_state sets it to None, :586), and the loop returns at the first violating execution: the
   instance's remaining verdicts stay all-zero ("not run").  rec / n_rec (optional): the recorded trace of the LAST execution
   that ran. */
static uint64_t instance_run(exec_t* x, const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed,
                             const demi_limits* lim, uint64_t first, uint64_t k, uint64_t n, demi_verdict* out,
                             demi_rec_event* rec, uint32_t rec_cap, uint32_t* n_rec) {
  orc_jrandom carried[2];
  demi_limits l = *lim;
  uint64_t last = first;
  for (uint64_t e = 0; e < k && first + e < n; e++) {
    if (e == 1) l.looking_for_valid = 0;
    random_execute_in2(x, m, trace, n_ev, seed, &l, &out[first + e], rec, rec_cap, n_rec, NULL, carried, e > 0);
    last = first + e;
    if (out[first + e].flags & DEMI_V_VIOLATION) {
      for (uint64_t r = e + 1; r < k && first + r < n; r++) memset(&out[first + r], 0, sizeof(demi_verdict));
      break;
    }
  }
  return last;
}

static void* job_main(void* p) {
  job_t* j = (job_t*)p;
  exec_t* x = (exec_t*)malloc(sizeof(exec_t)); /* one simulator per thread, reset per execution */
  if (!x) return NULL;
  const uint64_t k = j->lim->executions_per_instance > 1 ? j->lim->executions_per_instance : 1;
  if (k > 1) {
    /* (lo, hi are INSTANCE indices here; instance i owns the verdicts [i * k, (i + 1) * k)) */
    for (uint64_t i = j->lo; i < j->hi; i++) {
      uint64_t seed = j->seeds ? j->seeds[i] : j->seed_base + i;
      instance_run(x, j->m, j->trace, j->n_ev, seed, j->lim, i * k, k, j->n, j->out, NULL, 0, NULL);
    }
  } else {
    for (uint64_t i = j->lo; i < j->hi; i++) {
      uint64_t seed = j->seeds ? j->seeds[i] : j->seed_base + i;
      random_execute_in(x, j->m, j->trace, j->n_ev, seed, j->lim, &j->out[i], NULL, 0, NULL, NULL);
    }
  }
  free(x);
  return NULL;
}

int orc_random_execute_carried(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed, uint32_t exec_index,
                               const demi_limits* lim, demi_verdict* out, demi_rec_event* rec, uint32_t rec_cap,
                               uint32_t* n_rec, uint32_t* ran) {
  exec_t* x = (exec_t*)malloc(sizeof(exec_t));
  demi_verdict* v = (demi_verdict*)calloc((size_t)exec_index + 1, sizeof(demi_verdict));
  if (!x || !v) { free(x); free(v); return DEMI_ERR_INVALID_ARG; }
  const uint64_t last = instance_run(x, m, trace, n_ev, seed, lim, 0, (uint64_t)exec_index + 1, (uint64_t)exec_index + 1, v, rec, rec_cap, n_rec);
  *out = v[last];
  if (ran) *ran = (uint32_t)last;
  free(x); free(v);
  return DEMI_OK;
}

int orc_random_explore(const demi_model* m, const demi_ext_event* trace, uint32_t n_ev, uint64_t seed_base,
                       const uint64_t* seeds, uint64_t n, const demi_limits* lim, demi_verdict* out,
                       int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  job_t jobs[256];
  const uint64_t k = lim->executions_per_instance > 1 ? lim->executions_per_instance : 1;
  const uint64_t units = (n + k - 1) / k;                 /* executions, or instances in the carried-generator mode */
  for (int t = 0; t < n_threads; t++) {
    jobs[t] = (job_t){m, trace, n_ev, seed_base, seeds, units * (uint64_t)t / (uint64_t)n_threads,
                      units * (uint64_t)(t + 1) / (uint64_t)n_threads, lim, out, n};
    if (n_threads == 1) job_main(&jobs[t]);
    else pthread_create(&th[t], NULL, job_main, &jobs[t]);
  }
  if (n_threads > 1)
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  return DEMI_OK;
}

/* ===================================================================== K2: STSScheduler replay
 * One candidate = one STSScheduler.test (V/schedulers/STSScheduler.scala:199-310, no peek,
 * abortUponDivergence off, filterKnownAbsents off): project the original trace onto the
 * subsequence (EventTrace.subsequenceIntersection + filterSends, V/EventTrace.scala:290-452),
 * walk the expected events (advanceReplay :405-559), deliver an expected MsgEvent iff a message
 * with the same (snd, rcv, fingerprint) is pending (messagePending :381-403, oldest first
 * :709-737), else ignore it (:528-529); at the end evaluate the invariant and match the target
 * fingerprint (:278-300).                                                                        */
typedef struct { uint64_t word; uint32_t seq; } sts_pend;   /* word: 64 bits for a wide model */

typedef struct {
  const demi_model* m;
  int wide;           /* DEMI_MODEL_WIDE: 64-bit message words, two state words per actor */
  uint64_t state[ORC_MAX_STW * ORC_MAX_ACTORS];
  uint32_t exists, inaccessible, killed;
  uint32_t blocked;   /* crashed actors (Instrumenter().blockedActors): an expected delivery to one is not "pending" (:392-402) */
  orc_jrandom app_rng; /* Instrumenter().seededRandom, new with every replay */
  orc_pairset partitioned;
  sts_pend pend[PEND_HARD_CAP];   /* pendingEvents: (snd,rcv) -> fingerprint -> FIFO; seq keeps FIFO order */
  uint32_t n_pend, p_max, next_seq;
  uint8_t mts[DEMI_TQ_CAP][2];    /* messagesToSend timers (rcv, type) */
  uint32_t n_mts;
  uint64_t repeating;
  uint32_t flags, count, ignored;
  uint64_t hash;
  orc_effect fx[DEMI_MAX_CODE * ORC_MAX_ACTORS];
} sts_t;

static int sts_crosses(const sts_t* x, uint32_t snd, uint32_t rcv) {
  if (snd == rcv && !((x->killed >> snd) & 1)) return 0;
  int part = 0;
  if (ORC_IS_ACTOR(snd) && ORC_IS_ACTOR(rcv))
    part = ps_get(&x->partitioned, snd, rcv) | ps_get(&x->partitioned, rcv, snd);
  int ir = ORC_IS_ACTOR(rcv) ? (int)((x->inaccessible >> rcv) & 1) : 0;
  int is = ORC_IS_ACTOR(snd) ? (int)((x->inaccessible >> snd) & 1) : 0;
  return part || ir || is;
}

static void sts_pend_add(sts_t* x, uint64_t word) {
  if (x->flags & OVF_ANY) return;
  if (x->n_pend >= x->p_max) { x->flags |= DEMI_V_PENDING_OVF; return; }
  x->pend[x->n_pend].word = word;
  x->pend[x->n_pend].seq = x->next_seq++;
  x->n_pend++;
}

/* oldest pending message with this (snd, rcv, fingerprint); -1 if none */
static int sts_pend_find(const sts_t* x, uint64_t word) {
  int best = -1;
  for (uint32_t i = 0; i < x->n_pend; i++)
    if (x->pend[i].word == word && (best < 0 || x->pend[i].seq < x->pend[best].seq)) best = (int)i;
  return best;
}

static void sts_pend_remove(sts_t* x, int i) {
  x->pend[i] = x->pend[x->n_pend - 1];
  x->n_pend--;
}

/* STSScheduler.enqueue_timer = handle_timer (:857), no justScheduledTimers parking */
static void sts_handle_timer(sts_t* x, uint32_t rcv, uint32_t type) {
  if (x->flags & OVF_ANY) return;
  if (x->n_mts >= DEMI_TQ_CAP) { x->flags |= DEMI_V_QUEUE_OVF; return; }
  x->mts[x->n_mts][0] = (uint8_t)rcv;
  x->mts[x->n_mts][1] = (uint8_t)type;
  x->n_mts++;
}

/* send_external_messages for timers: internal messages from deadLetters (:583-607) */
static void sts_flush(sts_t* x) {
  for (uint32_t i = 0; i < x->n_mts; i++) {
    uint32_t rcv = x->mts[i][0], type = x->mts[i][1];
    if (!((x->inaccessible >> rcv) & 1)) sts_pend_add(x, msg_word(type, ORC_DL, rcv, 0, 0));
  }
  x->n_mts = 0;
}

static void sts_deliver(sts_t* x, uint64_t w) {
  const demi_model* m = x->m;
  uint32_t me = W_DST(w);
  x->count++;
  hash_step(&x->hash, w);
  /* Instrumenter retrigger of repeating timers (V/Instrumenter.scala:1008-1016), pinned before receive */
  if (m->msg_class[W_TYPE(w)] == DEMI_MSG_TIMER && (x->repeating & timer_bit(m, me, W_TYPE(w))))
    sts_handle_timer(x, me, W_TYPE(w));
  int n = orc_vm_run_area(m, me, &x->state[model_stw(m) * me], (uint8_t)W_TYPE(w), (uint8_t)W_SRC(w), WX_AREA(x->wide, w),
                          x->exists, x->fx, DEMI_MAX_CODE * ORC_MAX_ACTORS, &x->app_rng);
  if (n < 0) { x->flags |= DEMI_V_QUEUE_OVF; return; }
  for (int i = 0; i < n && !(x->flags & OVF_ANY); i++) {
    const orc_effect* e = &x->fx[i];
    uint64_t bit = e->kind ? timer_bit(m, me, e->msg_type) : 0;
    switch (e->kind) {
      case 0: /* event_produced, internal (:590-607) */
        if (!sts_crosses(x, me, e->target)) sts_pend_add(x, msg_word_a(x->wide, e->msg_type, me, e->target, e->area));
        break;
      case 1: case 2: /* registerCancellable + handleTick (V/Instrumenter.scala:1145-1200) */
        if (x->repeating & bit) break;
        if (e->kind == 2) x->repeating |= bit;
        sts_handle_timer(x, me, e->msg_type);
        break;
      case 3: { /* notify_timer_cancel (:828-855) */
        x->repeating &= ~bit;
        int found = 0;
        for (uint32_t k = 0; k < x->n_mts; k++) {
          if (x->mts[k][0] == me && x->mts[k][1] == e->msg_type) {
            memmove(&x->mts[k], &x->mts[k + 1], (x->n_mts - k - 1) * 2);
            x->n_mts--; found = 1; break;
          }
        }
        if (!found) {
          int k = sts_pend_find(x, msg_word(e->msg_type, ORC_DL, me, 0, 0));
          if (k >= 0) sts_pend_remove(x, k);
        }
        break;
      }
      case 4: { x->blocked |= 1u << me; /* actorCrashed (V/Instrumenter.scala:184-199) */
      }
    }
  }
  sts_flush(x); /* schedule_new_message begins with send_external_messages (:655) */
}

static int ext_equal(const demi_ext_event* e, uint32_t kind, uint32_t a, uint32_t b) {
  if (kind == DEMI_REC_SPAWN) return e->kind == DEMI_EV_START && e->a == a;
  if (kind == DEMI_REC_KILL) return e->kind == DEMI_EV_KILL && e->a == a;
  if (kind == DEMI_REC_PARTITION) return e->kind == DEMI_EV_PARTITION && e->a == a && e->b == b;
  if (kind == DEMI_REC_UNPARTITION) return e->kind == DEMI_EV_UNPARTITION && e->a == a && e->b == b;
  return 0;
}

static int sts_replay_in(sts_t* x, const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                         const demi_rec_event* rec, uint32_t n_rec, const uint64_t mask[4], const demi_limits* lim,
                         demi_verdict* out, uint32_t* n_ignored, uint32_t skip, uint8_t* kept) {
  /* skip: index of one MsgEvent removed from the trace before the replay (STSSchedMinimizer's candidate,
   * V/minification/internal_minimization/OneAtATimeRemoval.scala:57-124); kept[i] = 1 iff rec[i] took effect
   * in the replay, i.e. is part of the executed trace test() returns (V/schedulers/STSScheduler.scala:286-292) */
  memset(x, 0, offsetof(sts_t, fx));
  g_big = model_big(m);
  orc_jrandom_seed(&x->app_rng, 0);
  if (kept) memset(kept, 0, n_rec);
  x->m = m;
  x->wide = (m->flags & DEMI_MODEL_WIDE) != 0;
  x->p_max = lim->p_max ? lim->p_max : 64;
  if (x->p_max > PEND_HARD_CAP) x->p_max = PEND_HARD_CAP;
  x->hash = 0xCBF29CE484222325ULL;
  /* populateActorSystem from the ORIGINAL trace's SpawnEvents (:231-236) */
  for (uint32_t i = 0; i < n_rec; i++)
    if (rec[i].kind == DEMI_REC_SPAWN) x->exists |= 1u << rec[i].rcv;
  if (lim->populate_all) x->exists = (1u << m->n_actors) - 1;
  x->inaccessible = x->exists;
  state_init(m, x->state);

#define IN_MASK(i) ((mask[(i) >> 6] >> ((i) & 63)) & 1)
  /* id -> index of the Send that enqueued it (external messages only): filterSends (:382-452) */
  static _Thread_local uint8_t send_of_id[DEMI_MAX_REC_EVENTS * 2];
  memset(send_of_id, 255, sizeof send_of_id);
  for (uint32_t i = 0; i < n_rec; i++)
    if (rec[i].kind == DEMI_REC_MSG_SEND && (rec[i].flags & 1) && rec[i].id < sizeof send_of_id)
      send_of_id[rec[i].id] = rec[i].ext_idx;

  /* EventTrace.filterKnownAbsentInternals (EventTrace.scala:458-534), the last stage of the projection when
   * SchedulerConfig.filterKnownAbsents is set.  actorToAlive: default false, deadLetters / Timer true, SpawnEvent true,
   * KillEvent false.  actorsToPartitioned: default false, and as written PartitionEvent((a,b)) stores FALSE and
   * UnPartitionEvent((a,b)) stores TRUE under the ordered key (a,b) (:523-528) - DEMI_FILTER_ABSENTS_LITERAL keeps that;
   * _CORRECTED is the evident intent (cut off between Partition and UnPartition, either direction).  prunedMessageSends:
   * ids of MsgSends that were not sendable; their MsgEvents go too.                                                   */
  const uint32_t fka = lim->filter_known_absents;
  uint32_t fk_alive = 0;
  orc_pairset fk_part = {{0, 0, 0, 0}};
  static _Thread_local uint8_t fk_pruned[DEMI_MAX_REC_EVENTS * 2];
  if (fka) memset(fk_pruned, 0, sizeof fk_pruned);
#define FK_ALIVE(who) (!ORC_IS_ACTOR(who) ? 1u : ((fk_alive >> (who)) & 1u))
#define FK_PART(s, r) ((!ORC_IS_ACTOR(s) || !ORC_IS_ACTOR(r)) ? 0u                                                   \
                       : (fka == DEMI_FILTER_ABSENTS_LITERAL) ? (uint32_t)ps_get(&fk_part, (s), (r))                  \
                       : (uint32_t)(ps_get(&fk_part, (s), (r)) | ps_get(&fk_part, (r), (s))))

  /* cursor over the subsequence's non-Send externals (subsequenceIntersection :299-304) */
  uint32_t cur = 0;
#define CUR_SKIP() while (cur < n_ext && (!IN_MASK(cur) || ext[cur].kind == DEMI_EV_SEND || \
                                          ext[cur].kind == DEMI_EV_WAIT_QUIESCENCE)) cur++
  CUR_SKIP();
  for (uint32_t idx = 0; idx < n_rec && !(x->flags & OVF_ANY); idx++) {
    const demi_rec_event* e = &rec[idx];
    switch (e->kind) {
      case DEMI_REC_SPAWN: case DEMI_REC_KILL: case DEMI_REC_PARTITION: case DEMI_REC_UNPARTITION: {
        /* kept iff it equals the cursor head (by name, not identity); dropped once the cursor is exhausted */
        uint32_t a = (e->kind <= DEMI_REC_KILL) ? e->rcv : e->snd, b = e->rcv;
        if (cur >= n_ext || !ext_equal(&ext[cur], e->kind, a, b)) break;
        cur++;
        CUR_SKIP();
        if (kept) kept[idx] = 1;
        if (e->kind == DEMI_REC_SPAWN) { x->inaccessible &= ~(1u << a); x->killed &= ~(1u << a); x->blocked &= ~(1u << a); }
        else if (e->kind == DEMI_REC_KILL) { x->killed |= 1u << a; x->inaccessible |= 1u << a; }
        else if (e->kind == DEMI_REC_PARTITION) ps_set(&x->partitioned, a, b);
        else ps_clear(&x->partitioned, a, b);
        if (e->kind == DEMI_REC_SPAWN) fk_alive |= 1u << a;
        else if (e->kind == DEMI_REC_KILL) fk_alive &= ~(1u << a);
        else {
          /* literal: Partition -> false, UnPartition -> true (sic); corrected: the other way round */
          const int set = (fka == DEMI_FILTER_ABSENTS_LITERAL) ? (e->kind == DEMI_REC_UNPARTITION) : (e->kind == DEMI_REC_PARTITION);
          if (set) ps_set(&fk_part, a, b); else ps_clear(&fk_part, a, b);
        }
        break;
      }
      case DEMI_REC_MSG_SEND:
        /* filterKnownAbsentInternals: `if (messageSendable(m.sender, m.receiver)) result += event else prunedMessageSends += id`
         * (an external MsgSend is from deadLetters: always sendable) */
        if (fka && !(FK_ALIVE(e->snd) && !FK_PART(e->snd, e->rcv))) {
          if (e->id < sizeof fk_pruned) fk_pruned[e->id] = 1;
          break;
        }
        /* external MsgSend: enqueue_message (:509-511) unless its Send was pruned; internal: nothing */
        if ((e->flags & 1) && IN_MASK(e->ext_idx) && ((x->exists >> e->rcv) & 1)) {
          sts_pend_add(x, msg_word_a(x->wide, e->msg_type, ORC_DL, e->rcv, DEMI_REC_AREA(*e)));
          if (kept && !(x->flags & OVF_ANY)) kept[idx] = 1;
        }
        break;
      case DEMI_REC_MSG_EVENT: {
        uint8_t s = e->id < sizeof send_of_id ? send_of_id[e->id] : 255;
        if (idx == skip) break;             /* the delivery this candidate removes */
        if (s != 255 && !IN_MASK(s)) break; /* pruned together with its Send */
        /* filterKnownAbsentInternals: messageDeliverable(snd, rcv, id) or the event is not part of the projected trace */
        if (fka && !(FK_ALIVE(e->rcv) && !FK_PART(e->snd, e->rcv) && !(e->id < sizeof fk_pruned && fk_pruned[e->id]))) break;
        uint64_t w = msg_word_a(x->wide, e->msg_type, e->snd, e->rcv, DEMI_REC_AREA(*e));
        int k = ((x->blocked >> e->rcv) & 1) ? -1 : sts_pend_find(x, w);   /* messagePending: "double check that the destination isn't currently blocked" (:392-402) */
        if (k < 0) { x->ignored++; break; } /* "Ignoring message" (:528-529) */
        sts_pend_remove(x, k);
        if (kept) kept[idx] = 1;
        sts_deliver(x, w);
        break;
      }
      default: break; /* Quiescence, BeginWaitQuiescence: nop */
    }
  }
#undef CUR_SKIP
#undef IN_MASK
#undef FK_ALIVE
#undef FK_PART
  uint32_t viol = 0;
  if (!(x->flags & OVF_ANY)) {
    uint32_t fp = orc_invariant(m, x->state, x->exists);
    if (fp && ((fp ^ lim->looking_for) & m->fp_match_mask) == 0) viol = lim->looking_for;
  }
  for (uint32_t a = 0; a < m->n_actors * model_stw(m); a++) hash_step(&x->hash, x->state[a]);
  if (x->flags & OVF_ANY) {
    out->flags = x->flags & OVF_ANY; out->fingerprint = 0; out->hash = 0;
  } else {
    out->flags = (viol ? DEMI_V_VIOLATION : 0) | (x->ignored ? DEMI_V_DIVERGED : 0) | ((x->count < 0xFFFFu ? x->count : 0xFFFFu) << 16);
    out->fingerprint = viol;
    out->hash = x->hash;
  }
  if (n_ignored) *n_ignored = x->ignored;
  return DEMI_OK;
}

int orc_sts_replay(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                   uint32_t n_rec, const uint64_t mask[4], const demi_limits* lim, demi_verdict* out,
                   uint32_t* n_ignored) {
  sts_t* x = (sts_t*)malloc(sizeof(sts_t));
  if (!x) return DEMI_ERR_INVALID_ARG;
  int rc = sts_replay_in(x, m, ext, n_ext, rec, n_rec, mask, lim, out, n_ignored, 0xFFFFFFFFu, NULL);
  free(x);
  return rc;
}

static const uint64_t STS_ALL[4] = {~0ULL, ~0ULL, ~0ULL, ~0ULL};

int orc_sts_removal(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                    uint32_t n_rec, const uint64_t* mask, uint32_t skip, const demi_limits* lim, demi_verdict* out,
                    uint8_t* kept) {
  sts_t* x = (sts_t*)malloc(sizeof(sts_t));
  if (!x) return DEMI_ERR_INVALID_ARG;
  int rc = sts_replay_in(x, m, ext, n_ext, rec, n_rec, mask ? mask : STS_ALL, lim, out, NULL, skip, kept);
  free(x);
  if (rc == DEMI_OK && kept) {
    /* the executed trace holds the MsgSend of every message it delivered (event_produced records the send of each
     * produced message, V/schedulers/STSScheduler.scala:561-623; the recorded internal / timer MsgSends themselves are nops of
     * the replay, :530-538): such a send is kept iff the delivery with its id took effect */
    for (uint32_t i = 0; i < n_rec; i++) {
      if (rec[i].kind != DEMI_REC_MSG_SEND || (rec[i].flags & 1)) continue;
      kept[i] = 0;
      for (uint32_t k = i + 1; k < n_rec; k++)
        if (rec[k].kind == DEMI_REC_MSG_EVENT && rec[k].id == rec[i].id) { kept[i] = kept[k]; break; }
    }
  }
  return rc;
}

typedef struct {
  const demi_model* m; const demi_ext_event* ext; uint32_t n_ext; const demi_rec_event* rec; uint32_t n_rec;
  const uint64_t* masks; uint64_t lo, hi; const demi_limits* lim; demi_verdict* out; const uint32_t* skip;
} sts_job_t;

static void* sts_job_main(void* p) {
  sts_job_t* j = (sts_job_t*)p;
  sts_t* x = (sts_t*)malloc(sizeof(sts_t));
  if (!x) return NULL;
  for (uint64_t i = j->lo; i < j->hi; i++)
    sts_replay_in(x, j->m, j->ext, j->n_ext, j->rec, j->n_rec, j->masks ? &j->masks[4 * i] : STS_ALL, j->lim, &j->out[i],
                  NULL, j->skip ? j->skip[i] : 0xFFFFFFFFu, NULL);
  free(x);
  return NULL;
}

static int sts_batch(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                     uint32_t n_rec, const uint64_t* masks, const uint32_t* skip, uint64_t n, const demi_limits* lim,
                     demi_verdict* out, int n_threads) {
  if (n_threads < 1) n_threads = 1;
  if (n_threads > 256) n_threads = 256;
  pthread_t th[256];
  sts_job_t jobs[256];
  for (int t = 0; t < n_threads; t++) {
    jobs[t] = (sts_job_t){m, ext, n_ext, rec, n_rec, masks, n * (uint64_t)t / (uint64_t)n_threads,
                          n * (uint64_t)(t + 1) / (uint64_t)n_threads, lim, out, skip};
    if (n_threads == 1) sts_job_main(&jobs[t]);
    else pthread_create(&th[t], NULL, sts_job_main, &jobs[t]);
  }
  if (n_threads > 1)
    for (int t = 0; t < n_threads; t++) pthread_join(th[t], NULL);
  return DEMI_OK;
}

int orc_sts_replay_batch(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                         uint32_t n_rec, const uint64_t* masks, uint64_t n, const demi_limits* lim, demi_verdict* out,
                         int n_threads) {
  return sts_batch(m, ext, n_ext, rec, n_rec, masks, NULL, n, lim, out, n_threads);
}

int orc_sts_removal_batch(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec,
                          uint32_t n_rec, const uint64_t* masks, const uint32_t* skip, uint64_t n,
                          const demi_limits* lim, demi_verdict* out, int n_threads) {
  return sts_batch(m, ext, n_ext, rec, n_rec, masks, skip, n, lim, out, n_threads);
}

/* ===================================================================== K3: DPORwHeuristics
 * One interleaving of V/schedulers/DPORwHeuristics.scala (checkpointing off, FD off,
 * prioritizePendingUponDivergence=false, invariant at the end of the interleaving :877-902).
 * Pinned where the reference depends on Scala HashMap iteration order (:454-456): the divergent
 * choice iterates the (snd, rcv) queues in ascending (snd, rcv) order with the scheduler's
 * WaitQuiescence queue last.                                                                     */
#define DPOR_ROOT_KEY 0xCBF29CE484222325ULL
#define DPOR_MARKER_KEY(i) (DPOR_ROOT_KEY ^ (0x5155494553434500ULL | (uint64_t)(i)))
#define DPOR_PRIME 0x100000001B3ULL

typedef struct { uint64_t word; uint32_t seq; uint8_t parent; uint8_t qperiod; } dpor_pend;   /* word: 64 bits for a wide model */

typedef struct {
  const demi_model* m;
  const demi_dpor_params* par;
  int wide;                             /* DEMI_MODEL_WIDE: 64-bit message words, two state words per actor */
  uint64_t state[ORC_MAX_STW * ORC_MAX_ACTORS];
  uint32_t isolated;
  uint32_t blocked;    /* crashed actors: skipped by getPendingEvent (:455) and by getMatchingMessage (:478, 518) */
  orc_jrandom app_rng; /* Instrumenter().seededRandom, new with every interleaving */
  dpor_pend pend[PEND_HARD_CAP];
  uint32_t n_pend, p_max, next_seq;
  int marker_pending;        /* the (SCHEDULER, SCHEDULER) queue holds at most one marker */
  uint32_t marker_ext;
  demi_dpor_trace_entry* trace;
  uint32_t n_trace;
  uint32_t parent, cur_root, qperiod, next_qperiod, awaiting, quiescent_marker_ext;
  uint64_t repeating;
  uint32_t flags, count, deliveries;
  uint64_t hash;
  orc_effect fx[DEMI_MAX_CODE * ORC_MAX_ACTORS];
} dpor_t;

int orc_dpor_trace_validate(const demi_model* m, const demi_ext_event* ev, uint32_t n, char* err, size_t err_cap) {
  int rc = orc_trace_validate(m, ev, n, err, err_cap);
  if (rc) return rc;
  for (uint32_t i = 0; i < n; i++)
    if (ev[i].kind != DEMI_EV_START && ev[i].kind != DEMI_EV_SEND && ev[i].kind != DEMI_EV_WAIT_QUIESCENCE) {
      if (err) snprintf(err, err_cap, "event %u: unsuported external event for DPOR", i); /* :709 */
      return DEMI_ERR_INVALID_TRACE;
    }
  return DEMI_OK;
}

/* event_produced + getMessage (:803-847, 773-801): the node always exists in the graph; it is only
 * enqueued when the depth bound allows (:832-838) */
static void dpor_produce(dpor_t* x, uint64_t word) {
  uint32_t cur_depth = (uint32_t)x->trace[x->parent].depth + 1; /* currentDepth = pathLength(parent)+1 */
  if (x->par->depth_bound && cur_depth >= x->par->depth_bound) return;
  if (x->flags & OVF_ANY) return;
  if (x->n_pend >= x->p_max) { x->flags |= DEMI_V_PENDING_OVF; return; }
  x->pend[x->n_pend].word = word;
  x->pend[x->n_pend].seq = x->next_seq++;
  x->pend[x->n_pend].parent = (uint8_t)x->parent;
  x->pend[x->n_pend].qperiod = (uint8_t)x->qperiod; /* quiescentPeriod(node) = period at production (:284-288) */
  x->n_pend++;
}

static uint64_t dpor_key_of(const dpor_t* x, const dpor_pend* p) {
  return (x->trace[p->parent].key ^ p->word) * DPOR_PRIME;
}

/* runExternal (:684-721) */
static uint32_t dpor_run_external(dpor_t* x, const demi_ext_event* ext, uint32_t n_ext, uint32_t idx) {
  int await = 0;
  while (idx < n_ext && !await) {
    const demi_ext_event* e = &ext[idx];
    if (e->kind == DEMI_EV_START) x->isolated &= ~(1u << e->a);
    else if (e->kind == DEMI_EV_SEND)
      dpor_produce(x, msg_word_a(x->wide, e->msg_type, ORC_DL, e->a, ORC_EXT_AREA(x->m, e, idx)));
    else if (e->kind == DEMI_EV_WAIT_QUIESCENCE) { x->marker_pending = 1; x->marker_ext = idx; await = 1; }
    idx++;
  }
  return idx;
}

/* (a wide model's trace entry reports the low half of the message word - type, dst, src, p0 - include/demi_gpu.h) */
static int dpor_trace_push(dpor_t* x, uint64_t key, uint64_t word, uint32_t parent, uint32_t kind) {
  if (x->n_trace >= DEMI_DPOR_MAX_TRACE) { x->flags |= DEMI_V_TRACE_OVF; return -1; }
  demi_dpor_trace_entry* t = &x->trace[x->n_trace];
  t->key = key; t->word = (uint32_t)word; t->parent = (uint8_t)parent; t->qperiod = (uint8_t)x->qperiod;
  t->depth = (uint8_t)(x->n_trace == 0 ? 0 : x->trace[parent].depth + 1);
  t->kind = (uint8_t)kind;
  return (int)x->n_trace++;
}

static int dpor_cmp_queue(const dpor_pend* a, const dpor_pend* b) {
  /* pinned iteration order of pendingEvents: (snd, rcv) ascending, FIFO inside a queue */
  uint32_t ka = (W_SRC(a->word) << 4) | W_DST(a->word), kb = (W_SRC(b->word) << 4) | W_DST(b->word);
  if (ka != kb) return ka < kb ? -1 : 1;
  return a->seq < b->seq ? -1 : (a->seq > b->seq ? 1 : 0);
}

static void dpor_deliver(dpor_t* x, uint64_t w) {
  const demi_model* m = x->m;
  uint32_t me = W_DST(w);
  x->deliveries++;
  hash_step(&x->hash, w);
  if (m->msg_class[W_TYPE(w)] == DEMI_MSG_TIMER && (x->repeating & timer_bit(m, me, W_TYPE(w))))
    dpor_produce(x, msg_word(W_TYPE(w), ORC_DL, me, 0, 0)); /* retrigger -> enqueue_timer = `!` (Scheduler.scala:73) */
  int n = orc_vm_run_area(m, me, &x->state[model_stw(m) * me], (uint8_t)W_TYPE(w), (uint8_t)W_SRC(w), WX_AREA(x->wide, w),
                          (1u << m->n_actors) - 1, x->fx, DEMI_MAX_CODE * ORC_MAX_ACTORS, &x->app_rng);
  if (n < 0) { x->flags |= DEMI_V_QUEUE_OVF; return; }
  for (int i = 0; i < n && !(x->flags & OVF_ANY); i++) {
    const orc_effect* e = &x->fx[i];
    uint64_t bit = e->kind ? timer_bit(m, me, e->msg_type) : 0;
    switch (e->kind) {
      case 0: dpor_produce(x, msg_word_a(x->wide, e->msg_type, me, e->target, e->area)); break;
      case 1: case 2:
        if (x->repeating & bit) break; /* Non-unique timer */
        if (e->kind == 2) x->repeating |= bit;
        dpor_produce(x, msg_word(e->msg_type, ORC_DL, me, 0, 0));
        break;
      case 3: { /* notify_timer_cancel (:961-984): first in the (deadLetters, rcv) queue with this msg */
        x->repeating &= ~bit;
        uint64_t want = msg_word(e->msg_type, ORC_DL, me, 0, 0);
        int best = -1;
        for (uint32_t k = 0; k < x->n_pend; k++)
          if (x->pend[k].word == want && (best < 0 || x->pend[k].seq < x->pend[best].seq)) best = (int)k;
        if (best >= 0) { x->pend[best] = x->pend[x->n_pend - 1]; x->n_pend--; }
        break;
      }
      case 4: x->blocked |= 1u << me; break;   /* actorCrashed (V/Instrumenter.scala:184-199) */
    }
  }
}

int orc_dpor_execute(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const uint64_t* prefix,
                     uint32_t prefix_len, uint32_t shared_len, const demi_dpor_params* par, demi_verdict* out,
                     demi_dpor_trace_entry* trace, uint32_t* trace_len, demi_dpor_pair* pairs, uint32_t* n_pairs) {
  dpor_t* x = (dpor_t*)calloc(1, sizeof(dpor_t));
  if (!x) return DEMI_ERR_INVALID_ARG;
  g_big = model_big(m);
  x->m = m; x->par = par; x->trace = trace;
  x->wide = (m->flags & DEMI_MODEL_WIDE) != 0;
  x->p_max = par->p_max ? par->p_max : 64;
  if (x->p_max > PEND_HARD_CAP) x->p_max = PEND_HARD_CAP;
  x->hash = 0xCBF29CE484222325ULL;
  x->isolated = (1u << m->n_actors) - 1; /* maybeStartActors: isolatedActors ++= actorNames (:666-679) */
  orc_jrandom_seed(&x->app_rng, 0);
  state_init(m, x->state);
  dpor_trace_push(x, DPOR_ROOT_KEY, 0, 0, 0); /* start_trace: currentTrace += getRootEvent (:336-343) */
  x->parent = 0; x->cur_root = 0;
  uint32_t ext_idx = dpor_run_external(x, ext, n_ext, 0);
  uint32_t pfx = 0;
  uint32_t max_messages = par->max_messages ? par->max_messages : 0x7FFFFFFFu;
  uint32_t viol = 0;

  for (;;) {
    if (x->flags & (OVF_ANY | DEMI_V_TRACE_OVF | DEMI_V_SELFMSG)) break;
    /* ---- schedule_new_message (:421-648) */
    int chosen = -1, chose_marker = 0, none = 0;
    x->count++;                                       /* messagesScheduledSoFar += 1 (:583) */
    if (x->count > max_messages) none = 1;            /* (:584-586) */
    if (!none && !x->awaiting) {
      /* getMatchingMessage: pop nextTrace heads that are root / id 0 (:363-372), then look the head up */
      /* prioritizePendingUponDivergence: getNextMatchingMessage keeps popping heads until one is pending
       * (:537-550); otherwise exactly one head is tried (:594-597) */
      do {
        while (pfx < prefix_len && prefix[pfx] == DPOR_ROOT_KEY) pfx++;
        if (pfx >= prefix_len) break;
        uint64_t want = prefix[pfx++];
        if (x->marker_pending && want == DPOR_MARKER_KEY(x->marker_ext)) chose_marker = 1;
        else {
          for (uint32_t k = 0; k < x->n_pend; k++)
            if (dpor_key_of(x, &x->pend[k]) == want && !((x->blocked >> W_DST(x->pend[k].word)) & 1) &&
                (chosen < 0 || x->pend[k].seq < x->pend[chosen].seq)) chosen = (int)k;
        }
      } while (par->prioritize_pending && chosen < 0 && !chose_marker);
    }
    if (!none && chosen < 0 && !chose_marker) {
      /* divergent / first run / awaiting quiescence: getPendingEvent (:452-472), pinned order */
      for (uint32_t k = 0; k < x->n_pend; k++) {
        if ((x->blocked >> W_DST(x->pend[k].word)) & 1) continue;       /* !(blockedActors contains k._2) (:455) */
        if (chosen < 0 || dpor_cmp_queue(&x->pend[k], &x->pend[chosen]) < 0) chosen = (int)k;
      }
      if (chosen < 0 && x->marker_pending) chose_marker = 1;
      if (chosen < 0 && !chose_marker) none = 1;
    }
    if (chose_marker) {                               /* awaitQuiescenceUpdate (:256-266) */
      x->marker_pending = 0;
      x->awaiting = 1; x->next_qperiod = x->marker_ext + 1; x->quiescent_marker_ext = x->marker_ext;
      continue;
    }
    if (!none) {
      dpor_pend p = x->pend[chosen];
      x->pend[chosen] = x->pend[x->n_pend - 1];
      x->n_pend--;
      uint32_t snd = W_SRC(p.word), rcv = W_DST(p.word);
      if ((ORC_IS_ACTOR(snd) && ((x->isolated >> snd) & 1)) || ((x->isolated >> rcv) & 1)) {
        if (snd == rcv) { x->flags |= DEMI_V_SELFMSG; break; }  /* (:631-633) */
        continue;                                     /* discarded, schedule again (:626-635) */
      }
      uint64_t key = dpor_key_of(x, &p);
      int ti = dpor_trace_push(x, key, p.word, p.parent, 1);
      if (ti < 0) break;
      x->trace[ti].qperiod = p.qperiod;
      x->parent = (uint32_t)ti;                       /* setParentEvent (:636-639) */
      dpor_deliver(x, p.word);
      continue;
    }
    /* ---- notify_quiescence (:855-942) */
    if (x->awaiting) {
      x->awaiting = 0;
      x->qperiod = x->next_qperiod; x->next_qperiod = 0;
      int ti = dpor_trace_push(x, DPOR_MARKER_KEY(x->quiescent_marker_ext), 0, x->cur_root, 2);
      if (ti < 0) break;
      x->trace[ti].qperiod = (uint8_t)x->qperiod;
      x->cur_root = (uint32_t)ti; x->parent = (uint32_t)ti;
      ext_idx = dpor_run_external(x, ext, n_ext, ext_idx);
      continue;
    }
    break; /* end of this interleaving */
  }

  int aborted = (x->flags & (OVF_ANY | DEMI_V_TRACE_OVF | DEMI_V_SELFMSG)) != 0;
  if (!aborted) { /* checkInvariant (:394-418) */
    uint32_t fp = orc_invariant(m, x->state, (1u << m->n_actors) - 1);
    if (fp) {
      if (!par->looking_for_valid) viol = fp;
      else if (((fp ^ par->looking_for) & m->fp_match_mask) == 0) viol = par->looking_for;
    }
  }
  for (uint32_t a = 0; a < m->n_actors * model_stw(m); a++) hash_step(&x->hash, x->state[a]);

  /* ---- dpor(): racing pairs (:1122-1139) with isCoEnabeled (:1091-1110) and analyze_dep (:1043-1077) */
  uint32_t np = 0, pairs_ovf = 0;
  if (!aborted) {
    /* shared_len: the first shared_len events of this trace are those of the interleaving whose backtrack point this
     * prefix is (trace.take(branch + 1), :1180).  A pair inside that part was reported, with the same keys and the same
     * branch, when that interleaving ran, so dpor() has nothing new to learn from it (see include/demi_gpu.h). */
    for (uint32_t l = shared_len; l < x->n_trace; l++) {
      if (trace[l].kind != 1) continue;
      for (uint32_t e = 0; e < l; e++) {
        if (trace[e].kind != 1) continue;
        if (W_DST(trace[e].word) != W_DST(trace[l].word)) continue;
        if (trace[e].qperiod != trace[l].qperiod) continue;
        /* later.pathTo(earlier): is `e` an ancestor of `l`?  also collect l's ancestors for the LCA */
        int anc = 0;
        for (uint32_t k = l; k != 0;) { k = trace[k].parent; if (k == e) { anc = 1; break; } }
        if (anc) continue;
        /* getCommonPrefix(earlier, later).last: deepest common ancestor, as a trace index */
        uint32_t a = e, b = l;
        while (a != b) { if (a > b) a = trace[a].parent; else b = trace[b].parent; }
        if (np < par->max_pairs) { pairs[np].branch = (uint8_t)a; pairs[np].later = (uint8_t)l; pairs[np].earlier = (uint8_t)e; pairs[np].pad = 0; np++; }
        else pairs_ovf = 1;
      }
    }
  }

  if (aborted) {
    out->flags = x->flags & (OVF_ANY | DEMI_V_TRACE_OVF | DEMI_V_SELFMSG); out->fingerprint = 0; out->hash = 0;
    *trace_len = 0; *n_pairs = 0;
  } else {
    out->flags = (viol ? DEMI_V_VIOLATION : 0) | (pairs_ovf ? DEMI_V_PAIRS_OVF : 0) |
                 ((x->count > max_messages) ? DEMI_V_MAXMSG : 0) | ((x->deliveries < 0xFFFFu ? x->deliveries : 0xFFFFu) << 16);
    out->fingerprint = viol; out->hash = x->hash;
    *trace_len = x->n_trace; *n_pairs = np;
  }
  free(x);
  return DEMI_OK;
}
