// demi_device.hpp — device-side building blocks shared by the schedule-exploration kernels
// (gfx950 / CDNA4, wave64).  One lane simulates one candidate schedule; every per-lane array that
// needs dynamic indexing lives in LDS in [slot][lane] order, so any per-lane slot index maps lane l
// to bank l mod 32 (b32) / 2l mod 64 (b64): conflict-free by construction.
#pragma once

#ifndef __HIPCC_RTC__   // hiprtc (jit.hpp) has the device runtime built in
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

#include "../../include/demi_gpu.h"

namespace demi {

// Flat model blob uploaded once per demi_model_load (host fills it from demi_model).
struct DevModel {
  uint32_t n_actors, n_msg_types, n_classes, code_len;
  uint32_t inv_kind, inv_fa, inv_va, inv_fb;
  uint32_t fp_match_mask, wide /* DEMI_MODEL_WIDE */;
  uint32_t n_timer_types;   // TIMER-class message types of the model (their timer indices are 0 .. n_timer_types - 1)
  uint32_t timer_types;     // bit t = message type t is TIMER-class
  uint32_t meta[DEMI_MAX_MSG_TYPES];                           // msg_class | timer_idx << 8
  uint32_t handler_start[DEMI_MAX_CLASSES * DEMI_MAX_MSG_TYPES];  // 0xFFFF = ignored
  uint32_t actor_class[DEMI_MAX_ACTORS];
  uint64_t init_state[DEMI_MAX_ACTORS];
  uint32_t divmagic[257];  // ceil(2^(31+L)/d), L = ceil(log2 d): exact floor(r/d) for r < 2^31 (d <= 128: the pending
  uint32_t arr_len;        //  set's nextInt; d <= 255: DEMI_OP_RND bounds)      arr_len: DEMI_MODEL_ARRAY's length, 0 = none
  uint32_t optab[64];      // per-op control words (sim_core.hpp op_control), filled by the host
  uint32_t code[DEMI_MAX_CODE];
  uint64_t init_state_wide[2 * DEMI_MAX_ACTORS];   // DEMI_MODEL_WIDE: two words per actor (init_state is unused then)
  uint64_t tix_packed;      // timer index of message type t in bits 2t, 2t + 1 (what meta[t] >> 8 holds, without the table read)
  uint32_t npay, pad_;      // DEMI_MODEL_PAYLOADS: payload fields per message (2 unless the table says otherwise)
  // a table with more than DEMI_MAX_ACTORS actors (the BIG layout below; appended, so that every offset above stays what it was)
  uint32_t actor_class_big[DEMI_MAX_ACTORS_BIG];
  uint64_t init_state_big[2 * DEMI_MAX_ACTORS_BIG];    // (a big table is a wide table: two words per actor)
};

// A translation unit compiled for a DEMI_MODEL_WIDE table (-DDEMI_WIDE, only ever by demi_model_specialize) sees 64-bit
// message words and two state words per actor; everything else sees the 8-bit layout, unchanged.
#ifdef DEMI_WIDE
typedef uint64_t word_t;
constexpr uint32_t FLD_WORDS = 2;
constexpr bool WIDE_TU = true;
#else
typedef uint32_t word_t;
constexpr uint32_t FLD_WORDS = 1;
constexpr bool WIDE_TU = false;
#endif
// The BIG layout (-DDEMI_BIG, together with DEMI_WIDE, only ever by demi_model_specialize): a table with 9 .. 16 actors.  The
// message word's receiver field is 4 bits, its sender field 5 (deadLetters = 31), actor masks are 16 bits, the ordered-pair
// matrices (partitions, reach) 16 x 16, the timer bits 64.  Everything else compiles to what it compiled to before.
#ifdef DEMI_BIG
#ifndef DEMI_WIDE
#error "DEMI_BIG is a layout of wide tables"
#endif
constexpr bool BIG_TU = true;
constexpr uint32_t MAX_ACT = DEMI_MAX_ACTORS_BIG, DL = DEMI_DEADLETTERS_BIG, W_SRC_SHIFT = 9, W_SRC_MASK = 31;
typedef uint64_t tmask_t;     // timer bits: actor * DEMI_MAX_TIMER_TYPES + timer index
typedef uint64_t acpack_t;    // actor classes, 4 bits per actor
#else
constexpr bool BIG_TU = false;
constexpr uint32_t MAX_ACT = DEMI_MAX_ACTORS, DL = DEMI_DEADLETTERS, W_SRC_SHIFT = 8, W_SRC_MASK = 15;
typedef uint32_t tmask_t;
typedef uint32_t acpack_t;
#endif
constexpr uint32_t ACT_MASK = (1u << MAX_ACT) - 1u;      // every actor
__host__ __device__ constexpr uint32_t max_act_of(bool big) { return big ? DEMI_MAX_ACTORS_BIG : DEMI_MAX_ACTORS; }

// DEMI_MODEL_ARRAY(n): the actors' arrays (rows LDX / STX) are further state words behind the field word(s) - 8 elements to
// a word, 4 in a wide table - and, like the wide window, exist only in a translation unit compiled for the table
// (DEMI_JIT_ARR_LEN, from demi_model_specialize).  ST_WORDS = all the 64-bit words of one actor's state: what is initialised,
// kept per lane in LDS ([word][lane]) and hashed at the end.
#ifdef DEMI_JIT_ARR_LEN
constexpr uint32_t ARR_LEN = DEMI_JIT_ARR_LEN;
#else
constexpr uint32_t ARR_LEN = 0;
#endif
__host__ __device__ constexpr uint32_t arr_words_of(bool wide, uint32_t len) { return (len + (wide ? 4u : 8u) - 1u) / (wide ? 4u : 8u); }
constexpr uint32_t ARR_WORDS = arr_words_of(WIDE_TU, ARR_LEN);
constexpr uint32_t ST_WORDS = FLD_WORDS + ARR_WORDS;

// ------------------------------------------------------------------ message word
// type[4:0] | dst[7:5] | src[11:8] | p0[23:16] | p1[31:24]   (identical to the oracle's)
// wide: type[4:0] | dst[7:5] | src[11:8] | payload area[63:16]; field k of the area = bits [k * PAY_BITS, (k + 1) * PAY_BITS)
// of it, NPAY fields of PAY_BITS = 16, 16, 12, 9, 8 bits for NPAY = 2 ... 6 (DEMI_MODEL_PAYLOADS, include/demi_gpu.h;
// DEMI_JIT_NPAY from demi_model_specialize; the plain wide table is NPAY = 2: p0[31:16] | p1[47:32])
#ifdef DEMI_JIT_NPAY
constexpr uint32_t NPAY = DEMI_JIT_NPAY;
#else
constexpr uint32_t NPAY = 2;
#endif
// the payload areas of the external events (demi_ext_payload_areas): behind the events in the same device array, word
// EXT_AREA_OFFSET + i for event i; read by the kernels compiled for a DEMI_MODEL_PAYLOADS table only (DEMI_JIT_NPAY)
constexpr uint32_t EXT_AREA_OFFSET = DEMI_MAX_EXT_EVENTS + 1;
constexpr uint32_t PAY_BITS = DEMI_PAYLOAD_BITS(NPAY);
constexpr uint32_t PAY_MASK = (1u << PAY_BITS) - 1u;
#ifdef DEMI_WIDE
// the area of a message with the payload fields p0 .. p5 (each truncated to PAY_BITS; fields past NPAY are dropped)
__device__ __forceinline__ uint64_t pay_area(uint32_t p0, uint32_t p1, uint32_t p2 = 0, uint32_t p3 = 0, uint32_t p4 = 0, uint32_t p5 = 0) {
  uint64_t a = (uint64_t)(p0 & PAY_MASK) | ((uint64_t)(p1 & PAY_MASK) << PAY_BITS);
  if (NPAY > 2) a |= (uint64_t)(p2 & PAY_MASK) << (2 * PAY_BITS);
  if (NPAY > 3) a |= (uint64_t)(p3 & PAY_MASK) << (3 * PAY_BITS);
  if (NPAY > 4) a |= (uint64_t)(p4 & PAY_MASK) << (4 * PAY_BITS);
  if (NPAY > 5) a |= (uint64_t)(p5 & PAY_MASK) << (5 * PAY_BITS);
  return a;
}
__device__ __forceinline__ word_t msg_word_area(uint32_t type, uint32_t src, uint32_t dst, uint64_t area) {
  return (word_t)(type | (dst << 5) | (src << W_SRC_SHIFT)) | ((word_t)area << 16);
}
__device__ __forceinline__ word_t msg_word(uint32_t type, uint32_t src, uint32_t dst, uint32_t p0, uint32_t p1) {
  return msg_word_area(type, src, dst, pay_area(p0, p1));
}
__device__ __forceinline__ uint32_t w_type(word_t w) { return (uint32_t)w & 31u; }
__device__ __forceinline__ uint32_t w_dst(word_t w) { return ((uint32_t)w >> 5) & (MAX_ACT - 1u); }
__device__ __forceinline__ uint32_t w_src(word_t w) { return ((uint32_t)w >> W_SRC_SHIFT) & W_SRC_MASK; }
__device__ __forceinline__ uint64_t w_area(word_t w) { return w >> 16; }
__device__ __forceinline__ uint32_t w_pay(word_t w, uint32_t k) { return k < NPAY ? (uint32_t)(w >> (16 + k * PAY_BITS)) & PAY_MASK : 0u; }
__device__ __forceinline__ uint32_t w_p0(word_t w) { return w_pay(w, 0); }
__device__ __forceinline__ uint32_t w_p1(word_t w) { return w_pay(w, 1); }
#else
__device__ __forceinline__ uint32_t msg_word(uint32_t type, uint32_t src, uint32_t dst, uint32_t p0, uint32_t p1) {
  return type | (dst << 5) | (src << 8) | (p0 << 16) | (p1 << 24);
}
__device__ __forceinline__ uint32_t w_type(uint32_t w) { return w & 31u; }
__device__ __forceinline__ uint32_t w_dst(uint32_t w) { return (w >> 5) & 7u; }
__device__ __forceinline__ uint32_t w_src(uint32_t w) { return (w >> 8) & 15u; }
__device__ __forceinline__ uint32_t w_p0(uint32_t w) { return (w >> 16) & 255u; }
__device__ __forceinline__ uint32_t w_p1(uint32_t w) { return w >> 24; }
// (the "area" of a narrow word, as demi_rec_event stores it: p0 in bits 0..15, p1 in bits 16..31)
__device__ __forceinline__ uint64_t w_area(uint32_t w) { return (uint64_t)(w_p0(w) | (w_p1(w) << 16)); }
#endif

// ------------------------------------------------------------------ java.util.Random
// 48-bit LCG of the JDK javadoc; call sites schedulers/Util.scala:115,172.
__device__ __forceinline__ uint64_t jr_seed(uint64_t seed) { return (seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1); }
__device__ __forceinline__ uint32_t jr_next31(uint64_t& s) {
  s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  return (uint32_t)(s >> 17);
}
// nextInt(bound), 1 <= bound <= 256: power-of-two fast path, else modulo with the rejection loop;
// floor(r / bound) by multiply-high with a precomputed magic (no integer divide on the GPU).
__device__ __forceinline__ uint32_t jr_next_int(uint64_t& s, uint32_t bound, const uint32_t* magic) {
  uint32_t r = jr_next31(s);
  if ((bound & (bound - 1)) == 0) return r >> (__builtin_clz(bound));  // (bound * r) >> 31
  const uint32_t m = magic[bound];
  const uint32_t sh = 31 - __builtin_clz(bound - 1);  // L - 1
  for (;;) {
    const uint32_t q = __umulhi(r, m) >> sh;
    const uint32_t qd = q * bound;
    if ((qd + bound - 1) < 0x80000000u) return r - qd;  // u - (u % bound) + (bound-1) >= 0 as int32
    r = jr_next31(s);
  }
}

// ------------------------------------------------------------------ partitions
// EventOrchestrator.crosses_partition (schedulers/EventOrchestrator.scala:345-351)
#ifdef DEMI_BIG
struct PairSet {         // ordered pairs of actors: bit a * 16 + b of 256
  uint64_t w[4];
};
__device__ __forceinline__ void pairs_clear(PairSet& p) { p.w[0] = 0; p.w[1] = 0; p.w[2] = 0; p.w[3] = 0; }
// (selected by VALUE, not by address: an indexed register array would live in scratch memory)
__device__ __forceinline__ uint32_t pairs_get(const PairSet& p, uint32_t a, uint32_t b) {
  const uint32_t q = a >> 2, sh = ((a & 3u) << 4) | b;
  const uint64_t v = q == 0 ? p.w[0] : q == 1 ? p.w[1] : q == 2 ? p.w[2] : p.w[3];
  return (uint32_t)(v >> sh) & 1u;
}
__device__ __forceinline__ void pairs_put(PairSet& p, uint32_t a, uint32_t b, bool on) {
  const uint32_t q = a >> 2;
  const uint64_t bit = 1ull << (((a & 3u) << 4) | b);
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) {
    const uint64_t m = q == k ? bit : 0ull;
    p.w[k] = on ? (p.w[k] | m) : (p.w[k] & ~m);
  }
}
#else
struct PairSet {         // ordered pairs of actors: bit a * 8 + b
  uint64_t w;
};
__device__ __forceinline__ void pairs_clear(PairSet& p) { p.w = 0; }
__device__ __forceinline__ uint32_t pairs_get(const PairSet& p, uint32_t a, uint32_t b) { return (uint32_t)(p.w >> (a * 8 + b)) & 1u; }
__device__ __forceinline__ void pairs_put(PairSet& p, uint32_t a, uint32_t b, bool on) {
  if (on) p.w |= 1ULL << (a * 8 + b); else p.w &= ~(1ULL << (a * 8 + b));
}
#endif
struct Net {
  uint32_t inaccessible, killed;
  PairSet partitioned;   // the ordered pairs Partition()ed
};
__device__ __forceinline__ bool crosses_partition(const Net& n, uint32_t snd, uint32_t rcv) {
  // snd, rcv actors (actor-to-actor)
  if (snd == rcv && !((n.killed >> snd) & 1)) return false;
#ifdef DEMI_BIG
  const uint32_t part = pairs_get(n.partitioned, snd, rcv) | pairs_get(n.partitioned, rcv, snd);
#else
  const uint32_t part = (uint32_t)((n.partitioned.w >> (snd * 8 + rcv)) | (n.partitioned.w >> (rcv * 8 + snd))) & 1u;
#endif
  return (part | (n.inaccessible >> rcv) | (n.inaccessible >> snd)) & 1u;
}

// messagesToSend's timers / timersToResend: one byte per entry.  (rcv << 5 | type); BIG: (rcv << 2 | timer index) - four bits
// of receiver and five of type do not fit - with the type recovered from the table's constants (DEMI_JIT_TYPE_OF_TIX: the
// message type of timer index k in bits 5k .. 5k + 4; a big table always runs as compiled code)
#ifdef DEMI_BIG
#ifndef DEMI_JIT_TYPE_OF_TIX
#define DEMI_JIT_TYPE_OF_TIX 0u
#endif
__device__ __forceinline__ uint32_t tq_pack(uint32_t rcv, uint32_t type, uint32_t tix) { (void)type; return (rcv << 2) | tix; }
__device__ __forceinline__ uint32_t tq_rcv(uint32_t b) { return b >> 2; }
__device__ __forceinline__ uint32_t tq_type(uint32_t b) { return ((uint32_t)(DEMI_JIT_TYPE_OF_TIX) >> (5u * (b & 3u))) & 31u; }
#else
__device__ __forceinline__ uint32_t tq_pack(uint32_t rcv, uint32_t type, uint32_t tix) { (void)tix; return (rcv << 5) | type; }
__device__ __forceinline__ uint32_t tq_rcv(uint32_t b) { return b >> 5; }
__device__ __forceinline__ uint32_t tq_type(uint32_t b) { return b & 31u; }
#endif

__device__ __forceinline__ void hash_step(uint64_t& h, uint64_t v) { h = (h ^ v) * 0x100000001B3ULL; }

// ------------------------------------------------------------------ invariant
// Invariant descriptor (TestOracle.scala:27 `Invariant`) on the simulated state; returns the
// ViolationFingerprint code.  st: this lane's actor states in LDS, stride 64 u64.
__device__ __forceinline__ uint32_t fld(uint64_t s, uint32_t f) { return (uint32_t)(s >> (8 * f)) & 0xFF; }
// field f of actor a in a lane's state array (stride 64 words): 8 bits of its one word, or 16 bits of its two (wide)
__device__ __forceinline__ uint32_t state_field(const uint64_t* st, uint32_t a, uint32_t f) {
#ifdef DEMI_WIDE
  return (uint32_t)(st[(ST_WORDS * a + (f >> 2)) * 64] >> (16 * (f & 3))) & 0xFFFFu;
#else
  return (uint32_t)(st[(ST_WORDS * a) * 64] >> (8 * f)) & 0xFFu;
#endif
}
// element `idx` of actor a's array (DEMI_OP_LDX / DEMI_OP_STX): element k of an array word is its k-th byte (16-bit half in a
// wide table), so a store is one narrow LDS write, not a read-modify-write of the word; past the end: 0 / nothing
__device__ __forceinline__ uint32_t arr_load(const uint64_t* st, uint32_t a, uint32_t idx) {
  if (ARR_LEN == 0 || idx >= ARR_LEN) return 0;
  const unsigned char* w = reinterpret_cast<const unsigned char*>(st + (size_t)(ST_WORDS * a + FLD_WORDS + idx / (WIDE_TU ? 4u : 8u)) * 64);
  if (WIDE_TU) return reinterpret_cast<const uint16_t*>(w)[idx & 3u];
  return w[idx & 7u];
}
__device__ __forceinline__ void arr_store(uint64_t* st, uint32_t a, uint32_t idx, uint32_t v) {
  if (ARR_LEN == 0 || idx >= ARR_LEN) return;
  unsigned char* w = reinterpret_cast<unsigned char*>(st + (size_t)(ST_WORDS * a + FLD_WORDS + idx / (WIDE_TU ? 4u : 8u)) * 64);
  if (WIDE_TU) reinterpret_cast<uint16_t*>(w)[idx & 3u] = (uint16_t)v;
  else w[idx & 7u] = (unsigned char)v;
}

// (the invariant itself - per-actor hit / key, the combining kinds - lives in sim_core.hpp: a DEMI_INV_PROGRAM invariant runs rows)

}  // namespace demi
