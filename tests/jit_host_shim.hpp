// Host stand-ins for the few device definitions the generated handler code (demi_specialize_source) uses,
// so that the CPU suite can compile it with g++ and compare it with the oracle's row interpreter.
#include <cstdint>
#define __device__
#define DEMI_FX_CAP 8
#define DEMI_V_QUEUE_OVF 0x8u
namespace demi {
constexpr uint32_t FX_NOBODY = 15u;     // (sim_core.hpp: a SEND target beyond the field is nobody; the shim runs tables of up to 8 actors)
struct Tables { const uint32_t* hs; uint32_t ac_packed, NT; const uint32_t* magic; const uint32_t* gmagic; };
#ifdef DEMI_WIDE      // a DEMI_MODEL_WIDE table: 64-bit message / effect words, two state words per actor
typedef uint64_t word_t;
struct LaneMem { uint64_t* st; word_t* fxq; };
static inline uint32_t w_type(word_t w) { return (uint32_t)w & 31u; }
static inline uint32_t w_dst(word_t w) { return ((uint32_t)w >> 5) & 7u; }
static inline uint32_t w_src(word_t w) { return ((uint32_t)w >> 8) & 15u; }
// DEMI_MODEL_PAYLOADS (demi_device.hpp): NPAY fields of PAY_BITS bits in the 48-bit payload area above the word's header
#ifndef DEMI_JIT_NPAY
#define DEMI_JIT_NPAY 2
#endif
constexpr uint32_t NPAY = DEMI_JIT_NPAY, PAY_BITS = NPAY <= 3 ? 16u : 48u / NPAY, PAY_MASK = (1u << PAY_BITS) - 1u;
static inline uint64_t pay_area(uint32_t p0, uint32_t p1, uint32_t p2 = 0, uint32_t p3 = 0, uint32_t p4 = 0, uint32_t p5 = 0) {
  const uint32_t p[6] = {p0, p1, p2, p3, p4, p5};
  uint64_t a = 0;
  for (uint32_t k = 0; k < NPAY; k++) a |= (uint64_t)(p[k] & PAY_MASK) << (k * PAY_BITS);
  return a;
}
static inline uint32_t w_pay(word_t w, uint32_t k) { return k < NPAY ? (uint32_t)(w >> (16 + k * PAY_BITS)) & PAY_MASK : 0u; }
static inline uint32_t w_p0(word_t w) { return w_pay(w, 0); }
static inline uint32_t w_p1(word_t w) { return w_pay(w, 1); }
static inline word_t fx_pack_area(uint32_t op, uint32_t type, uint32_t target, uint64_t area) {
  return (word_t)((op & 31u) | (type << 5) | (target << 10)) | ((word_t)area << 14);
}
static inline word_t fx_pack(uint32_t op, uint32_t type, uint32_t target, uint32_t p0, uint32_t p1) {
  return fx_pack_area(op, type, target, pay_area(p0, p1));
}
#else
typedef uint32_t word_t;
struct LaneMem { uint64_t* st; uint32_t* fxq; };
static inline uint32_t w_type(uint32_t w) { return w & 31u; }
static inline uint32_t w_dst(uint32_t w) { return (w >> 5) & 7u; }
static inline uint32_t w_src(uint32_t w) { return (w >> 8) & 15u; }
static inline uint32_t w_p0(uint32_t w) { return (w >> 16) & 255u; }
static inline uint32_t w_p1(uint32_t w) { return w >> 24; }
static inline uint32_t fx_pack(uint32_t op, uint32_t type, uint32_t target, uint32_t p0, uint32_t p1) {
  return (op & 31u) | (type << 5) | (target << 10) | (p0 << 14) | (p1 << 22);
}
#endif
// DEMI_MODEL_ARRAY (demi_device.hpp): the state words of an actor are its field word(s), then its array - here with shifts and
// masks on the 64-bit words (the device addresses the element's byte / half-word directly)
#ifndef DEMI_JIT_ARR_LEN
#define DEMI_JIT_ARR_LEN 0
#endif
#ifdef DEMI_WIDE
constexpr uint32_t FLD_WORDS = 2, ARR_PER = 4, ARR_BITS = 16;
#else
constexpr uint32_t FLD_WORDS = 1, ARR_PER = 8, ARR_BITS = 8;
#endif
constexpr uint32_t ARR_LEN = DEMI_JIT_ARR_LEN, ST_WORDS = FLD_WORDS + (ARR_LEN + ARR_PER - 1) / ARR_PER;
static inline uint32_t arr_load(const uint64_t* st, uint32_t a, uint32_t idx) {
  if (idx >= ARR_LEN) return 0;
  return (uint32_t)(st[(ST_WORDS * a + FLD_WORDS + idx / ARR_PER) * 64] >> (ARR_BITS * (idx % ARR_PER))) & ((1u << ARR_BITS) - 1u);
}
static inline void arr_store(uint64_t* st, uint32_t a, uint32_t idx, uint32_t v) {
  if (idx >= ARR_LEN) return;
  uint64_t& w = st[(ST_WORDS * a + FLD_WORDS + idx / ARR_PER) * 64];
  const uint64_t m = (uint64_t)((1u << ARR_BITS) - 1u) << (ARR_BITS * (idx % ARR_PER));
  w = (w & ~m) | (((uint64_t)v << (ARR_BITS * (idx % ARR_PER))) & m);
}
// DEMI_OP_PEER (sim_core.hpp peer_field): field f (8 = "is created") of actor `who`; not a created actor: 0
static inline uint32_t peer_field(const uint64_t* st, uint32_t who, uint32_t f, uint32_t exists, uint32_t n_actors) {
  if (who >= n_actors || !((exists >> who) & 1u)) return 0u;
  if (f >= 8u) return 1u;
#ifdef DEMI_WIDE
  return (uint32_t)(st[(ST_WORDS * who + (f >> 2)) * 64] >> (16 * (f & 3))) & 0xFFFFu;
#else
  return (uint32_t)(st[(ST_WORDS * who) * 64] >> (8 * f)) & 0xFFu;
#endif
}
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
// DEMI_OP_RND: java.util.Random.nextInt(bound) on the application's generator (the device uses multiply-high magics for the
// modulo; here the plain JDK algorithm - the results must agree)
static inline uint32_t app_next_int(uint64_t& s, uint32_t bound, const uint32_t*) {
  if (bound == 0) return 0;
  auto next31 = [&]() { s = (s * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1); return (uint32_t)(s >> 17); };
  uint32_t r = next31();
  if ((bound & (bound - 1)) == 0) return (uint32_t)(((uint64_t)bound * r) >> 31);
  uint32_t u = r;
  while ((int32_t)(u - (r = u % bound) + (bound - 1)) < 0) u = next31();
  return r;
}
}  // namespace demi
