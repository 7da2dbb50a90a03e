// sim_core.hpp — the per-lane actor-system simulator shared by the exploration kernels.
//
// A wavefront is a pool of 64 independent simulators.  Everything that needs a per-lane dynamic
// index lives in LDS as [slot][lane] (stride 64 elements), so lane l always hits bank l mod 32
// (b32) / banks 2l,2l+1 mod 64 (b64): no conflicts for ANY combination of per-lane slots.
//
// The row interpreter is branch-free: every lane fetches its own row (per-lane pc) and computes
// all result classes with selects, so lanes sitting in different handlers of the transition
// table do not serialise.  Rows with effects (SEND/BCAST/TSET/TREP/TCANCEL) are only recorded
// while the rows run; they are applied afterwards in program order, slot k of every lane at the
// same time (apply phase), so the expensive bodies are paid once per slot and not once per row.
#pragma once

#include "demi_device.hpp"

namespace demi {

#define DEMI_OVF_ANY (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)

// Per-op control word (what the row does), looked up instead of decoded with compare chains.
// The masks make the ALU a sum of mutually exclusive (candidate & mask) terms: no VCC selects.
enum : uint32_t {
  CW_ADDSUB = 1u << 0,   // r = (a & am) + (b ^ nm) + n1   (MOV: am = 0; SUB: nm = ~0, n1 = 1)
  CW_KEEP_A = 1u << 1,   //   am = ~0
  CW_NEG_B = 1u << 2,    //   nm = ~0, n1 = 1
  CW_AND = 1u << 3, CW_OR = 1u << 4, CW_BITB = 1u << 5 /* OR with b := 1 << (b & 7) */, CW_XOR = 1u << 6,
  CW_SHL = 1u << 7, CW_SHR = 1u << 8, CW_POPC = 1u << 9, CW_CMP = 1u << 10,
  CW_MINMAX = 1u << 11, CW_IS_MIN = 1u << 12,
  CW_ALU = 1u << 13,     // writes dst
  CW_IF = 1u << 14,      // fused guard: skip aux rows unless the relation holds
  CW_SKIPZ = 1u << 15, CW_SKIPNZ = 1u << 16, CW_SKIP = 1u << 17,
  CW_FX = 1u << 18, CW_HALT = 1u << 19,
  CW_REL_SHIFT = 20,     // bits 20..22: accepted relations {lt, eq, gt} of compare / guard rows
  CW_RND = 1u << 23,     // dst = seededRandom.nextInt(b)
  CW_LDX = 1u << 24, CW_STX = 1u << 25,    // dst = ARRAY[b] / ARRAY[b] = a (DEMI_MODEL_ARRAY)
  CW_PEER = 1u << 26,                      // dst = a field of another actor (invariant programs only)
  CW_LDP = 1u << 27, CW_PSET = 1u << 28    // dst = payload field b / staged payload field aux = b (DEMI_MODEL_PAYLOADS: compiled only)
};
// DevModel::inv_kind, set by the host when the invariant's program has a DEMI_OP_PEER row: an actor's hit depends on the other
// actors' states, so K1's incrementally maintained hit mask is rebuilt at every check instead
#define DEMI_INV_PEERS 0x200u

inline uint32_t op_control(uint32_t op) {   // host side: fills DevModel::optab
  // accepted-relation masks in the order EQ NE LT GE LE GT
  const uint32_t rels[6] = {2u, 5u, 1u, 6u, 3u, 4u};
  if (op == DEMI_OP_HALT) return CW_HALT;
  if (op == DEMI_OP_MOV) return CW_ALU | CW_ADDSUB;
  if (op == DEMI_OP_ADD) return CW_ALU | CW_ADDSUB | CW_KEEP_A;
  if (op == DEMI_OP_SUB) return CW_ALU | CW_ADDSUB | CW_KEEP_A | CW_NEG_B;
  if (op == DEMI_OP_AND) return CW_ALU | CW_AND;
  if (op == DEMI_OP_OR) return CW_ALU | CW_OR;
  if (op == DEMI_OP_XOR) return CW_ALU | CW_XOR;
  if (op == DEMI_OP_SHL) return CW_ALU | CW_SHL;
  if (op == DEMI_OP_SHR) return CW_ALU | CW_SHR;
  if (op == DEMI_OP_BITSET) return CW_ALU | CW_OR | CW_BITB;
  if (op == DEMI_OP_POPC) return CW_ALU | CW_POPC;
  if (op >= DEMI_OP_EQ && op <= DEMI_OP_GT) return CW_ALU | CW_CMP | (rels[op - DEMI_OP_EQ] << CW_REL_SHIFT);
  if (op == DEMI_OP_MIN) return CW_ALU | CW_MINMAX | CW_IS_MIN;
  if (op == DEMI_OP_MAX) return CW_ALU | CW_MINMAX;
  if (op == DEMI_OP_SKIPZ) return CW_SKIPZ;
  if (op == DEMI_OP_SKIPNZ) return CW_SKIPNZ;
  if (op == DEMI_OP_SKIP) return CW_SKIP;
  if (op >= DEMI_OP_SEND && op <= DEMI_OP_TCANCEL) return CW_FX;
  if (op == DEMI_OP_CRASH) return CW_FX | CW_HALT;     // recorded as the delivery's last effect, then the rows stop
  if (op == DEMI_OP_RND) return CW_ALU | CW_RND;
  if (op == DEMI_OP_MOVHI) return CW_ALU;     // wide tables only, and those are never interpreted (jit.hpp emits it)
  if (op == DEMI_OP_LDX) return CW_ALU | CW_LDX;   // DEMI_MODEL_ARRAY tables only: compiled, never interpreted (jit.hpp)
  if (op == DEMI_OP_STX) return CW_STX;
  if (op == DEMI_OP_PEER) return CW_ALU | CW_PEER;
  if (op == DEMI_OP_LDP) return CW_ALU | CW_LDP;   // (the interpreter never sees one: validation wants DEMI_MODEL_PAYLOADS, a wide table)
  if (op == DEMI_OP_PSET) return CW_PSET;
  if (op >= DEMI_OP_IFEQ && op <= DEMI_OP_IFGT) return CW_IF | (rels[op - DEMI_OP_IFEQ] << CW_REL_SHIFT);
  return CW_HALT;   // unknown ops are rejected by validation
}

// ------------------------------------------------------------------ workgroup-shared tables
struct Tables {
  const uint64_t* trace;  // [E] external events, one 8-byte word each
  const uint64_t* init;   // [8 * ST_WORDS] initial actor states
  const uint32_t* code;   // [code_len] transition-table rows
  const uint32_t* hs;     // [n_classes * NT] handler starts
  const uint32_t* meta;   // [32] msg_class | timer_idx << 8
  const uint32_t* magic;  // [129] nextInt multiply-high magics of the scheduler's bounds (<= p_max <= 128), in LDS
  const uint32_t* gmagic; // [257] the whole table in the model blob (global memory): DEMI_OP_RND's bounds go up to 255 and are rare
  const uint32_t* optab;  // [64] per-op control words (op_control)
  uint32_t A, NT, code_len, E, exists;
  acpack_t ac_packed;                    // actor classes, 4 bits per actor
  uint32_t inv_kind, inv_fa, inv_va, inv_fb, fp_mask;
  uint32_t n_timer_types, timer_types;   // TIMER-class message types: how many, and which (bit t)
  uint64_t tix_packed;                   // their timer indices, two bits per message type
};

__host__ __device__ inline size_t tables_lds_bytes(uint32_t code_len, uint32_t n_ev, uint32_t n_hs, bool wide = WIDE_TU,
                                                   uint32_t arr_words = ARR_WORDS, bool big = BIG_TU) {
  size_t b = (size_t)n_ev * 8 + max_act_of(big) * 8 * ((wide ? 2 : 1) + arr_words) + (size_t)code_len * 4 + (size_t)n_hs * 4 +
             DEMI_MAX_MSG_TYPES * 4 + 132 * 4 + 64 * 4;
  return (b + 15) & ~(size_t)15;
}

// Streams the trace and the tables from HBM into LDS once per workgroup (coalesced); ends with a
// workgroup barrier.  Returns the first byte after the tables.
__device__ inline unsigned char* tables_load(Tables& t, unsigned char* smem, const DevModel* __restrict__ gm,
                                             const uint64_t* __restrict__ g_trace, uint32_t n_ev, uint32_t exists) {
  t.A = gm->n_actors; t.NT = gm->n_msg_types; t.code_len = gm->code_len; t.E = n_ev; t.exists = exists;
  t.inv_kind = gm->inv_kind; t.inv_fa = gm->inv_fa; t.inv_va = gm->inv_va; t.inv_fb = gm->inv_fb;
  t.fp_mask = gm->fp_match_mask;
  t.n_timer_types = gm->n_timer_types; t.timer_types = gm->timer_types; t.tix_packed = gm->tix_packed;
  const uint32_t n_hs = gm->n_classes * t.NT;
  uint64_t* s_trace = reinterpret_cast<uint64_t*>(smem);
  uint64_t* s_init = s_trace + n_ev;
  uint32_t* s_code = reinterpret_cast<uint32_t*>(s_init + MAX_ACT * ST_WORDS);
  uint32_t* s_hs = s_code + t.code_len;
  uint32_t* s_meta = s_hs + n_hs;
  uint32_t* s_magic = s_meta + DEMI_MAX_MSG_TYPES;
  uint32_t* s_optab = s_magic + 132;
  for (uint32_t i = threadIdx.x; i < n_ev; i += blockDim.x) s_trace[i] = g_trace[i];
  // (a big table's fields and actor classes come from the arrays appended to the model blob)
  const uint64_t* const g_init_wide = BIG_TU ? gm->init_state_big : gm->init_state_wide;
  for (uint32_t i = threadIdx.x; i < MAX_ACT * ST_WORDS; i += blockDim.x) {
    if (ARR_WORDS == 0) s_init[i] = WIDE_TU ? g_init_wide[i] : gm->init_state[i];
    else {                                   // (the fields from the model, the arrays empty)
      const uint32_t a = i / ST_WORDS, k = i % ST_WORDS;
      s_init[i] = k >= FLD_WORDS ? 0ull : WIDE_TU ? g_init_wide[a * FLD_WORDS + k] : gm->init_state[a];
    }
  }
  for (uint32_t i = threadIdx.x; i < t.code_len; i += blockDim.x) s_code[i] = gm->code[i];
  for (uint32_t i = threadIdx.x; i < n_hs; i += blockDim.x) s_hs[i] = gm->handler_start[i];
  for (uint32_t i = threadIdx.x; i < DEMI_MAX_MSG_TYPES; i += blockDim.x) s_meta[i] = gm->meta[i];
  for (uint32_t i = threadIdx.x; i < 129; i += blockDim.x) s_magic[i] = gm->divmagic[i];
  for (uint32_t i = threadIdx.x; i < 64; i += blockDim.x) s_optab[i] = gm->optab[i];
  t.ac_packed = 0;
  for (uint32_t a = 0; a < t.A; a++) t.ac_packed |= (acpack_t)(BIG_TU ? gm->actor_class_big[a] : gm->actor_class[a]) << (4 * a);
  t.trace = s_trace; t.init = s_init; t.code = s_code; t.hs = s_hs; t.meta = s_meta; t.magic = s_magic; t.gmagic = gm->divmagic; t.optab = s_optab;
  __syncthreads();
  return smem + tables_lds_bytes(t.code_len, n_ev, n_hs);     // (this translation unit's layout: the defaults)
}

// ------------------------------------------------------------------ per-lane arrays
// All LDS pointers are already offset by the lane id; element k of an array is p[k * 64].
// The pending set keeps its PEND_HOT lowest slots in LDS; deeper slots (rare: <1% of raft
// schedules ever hold more than 32 pending messages) spill to an HBM scratch laid out
// [slot - PEND_HOT][global lane], so a wave's spill accesses to one slot are coalesced.
#ifndef DEMI_PEND_HOT
#define DEMI_PEND_HOT 32
#endif
constexpr uint32_t PEND_HOT = DEMI_PEND_HOT;

struct LaneMem {
  uint64_t* st;        // [A * ST_WORDS] actor states (LDS)
  word_t* pend;        // [PEND_HOT]   pending message words (LDS)
  uint32_t* pend_aux;  // [PEND_HOT]   optional parallel array (ids / sequence numbers), may be null
  word_t* fxq;         // [FX_CAP]     effect rows recorded by the current delivery (LDS)
  // global scratch: slot s >= hot of this lane is spill[(s - hot) * spill_stride + spill_lane].  The base pointers are the
  // same for every lane (scalar registers) and the index is 32 bits (128 slots x < 2^24 lanes), so an access is one
  // multiply-add for the offset instead of a 64-bit multiply and a 64-bit add
  word_t* spill;
  uint32_t* spill_aux; // parallel to spill (may be null)
  uint32_t spill_stride, spill_lane;
  uint32_t hot;        // slots below `hot` live in LDS (PEND_HOT, or less where occupancy is worth more than residency)
};

// fxq_slots: entries of the effect queue (DEMI_FX_CAP; a K1 compiled with an effect-slot schedule only stores its SEND /
// BCAST slots, jit.hpp fx_schedule)
__host__ __device__ inline size_t lane_mem_wave_bytes(uint32_t n_actors, bool aux, uint32_t hot = PEND_HOT, bool wide = WIDE_TU,
                                                      uint32_t fxq_slots = DEMI_FX_CAP, uint32_t arr_words = ARR_WORDS) {
  const size_t wb = wide ? 8 : 4;     // bytes per message / effect word
  return (size_t)n_actors * 64 * 8 * ((wide ? 2 : 1) + arr_words) + (size_t)hot * 64 * (wb + (aux ? 4 : 0)) + (size_t)fxq_slots * 64 * wb;
}
// HBM scratch words for `lanes` simulators (per array)
__host__ __device__ inline size_t spill_words(size_t lanes, uint32_t hot = PEND_HOT) { return lanes * (DEMI_MAX_PENDING - hot); }

__device__ inline LaneMem lane_mem_carve(unsigned char* wave_base, uint32_t n_actors, bool aux, uint32_t lane,
                                         uint32_t* g_spill_words, size_t global_lane, size_t total_lanes,
                                         uint32_t hot = PEND_HOT) {
  word_t* const g_spill = reinterpret_cast<word_t*>(g_spill_words);
  // (the aux arrays behind the message words are 32-bit also when the message words are 64-bit: the host sizes the scratch
  // as words x (1 or 2) + aux x 1 dwords per slot)
  uint32_t* const g_aux = reinterpret_cast<uint32_t*>(g_spill + spill_words(total_lanes, hot));
  LaneMem m;
  m.st = reinterpret_cast<uint64_t*>(wave_base) + lane;
  unsigned char* q = wave_base + (size_t)n_actors * ST_WORDS * 64 * 8;
  m.pend = reinterpret_cast<word_t*>(q) + lane;
  q += (size_t)hot * 64 * sizeof(word_t);
  m.pend_aux = aux ? reinterpret_cast<uint32_t*>(q) + lane : nullptr;
  if (aux) q += (size_t)hot * 64 * 4;
  m.fxq = reinterpret_cast<word_t*>(q) + lane;
#ifndef DEMI_SPILL_WAVE_BLOCKS
  // one [slot][lane] matrix over every lane of the launch (slot stride = all lanes): the live part of the pending sets
  // (the low slots of every wave) is one contiguous region.  Measured over 8 processes each: 4.37-4.39 ms per 2^20 every
  // time, where one [slot][64] block per wave (DEMI_SPILL_WAVE_BLOCKS, kept for A/B runs: 32 KB stride with only the first
  // few KB of each block live) is 4.37 ms in some processes and 4.58 ms in others with the same binary, depending on
  // where the allocation happens to lie physically
  m.spill = g_spill;
  m.spill_aux = aux ? g_aux : nullptr;
  m.spill_stride = (uint32_t)total_lanes;
  m.spill_lane = (uint32_t)global_lane;
#else
  m.spill = g_spill;
  m.spill_aux = aux ? g_aux : nullptr;
  m.spill_stride = 64;
  m.spill_lane = (uint32_t)((global_lane >> 6) * ((size_t)(DEMI_MAX_PENDING - hot) * 64) + lane);
#endif
  m.hot = hot;
  return m;
}

// (slot < 2^7, stride < 2^24: the 24-bit multiply-add is exact and full rate; the BYTE offset stays below 2^32 - 128 slots x
// < 2^22 lanes x 8 bytes - and is formed in 32 bits so that the access is scalar base + 32-bit vector offset)
__device__ __forceinline__ uint32_t spill_index(const LaneMem& m, uint32_t slot) {
  return __umul24(slot - m.hot, m.spill_stride) + m.spill_lane;
}
template <typename T>
__device__ __forceinline__ T* spill_at(T* base, uint32_t index) {
  return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(base) + (uint32_t)(index * (uint32_t)sizeof(T)));
}
__device__ __forceinline__ word_t pend_load(const LaneMem& m, uint32_t slot) {
  if (slot < m.hot) return m.pend[slot * 64];
  return *spill_at(m.spill, spill_index(m, slot));
}
__device__ __forceinline__ void pend_store(const LaneMem& m, uint32_t slot, word_t v) {
  if (slot < m.hot) m.pend[slot * 64] = v;
  else *spill_at(m.spill, spill_index(m, slot)) = v;
}
__device__ __forceinline__ uint32_t aux_load(const LaneMem& m, uint32_t slot) {
  if (slot < m.hot) return m.pend_aux[slot * 64];
  return *spill_at(m.spill_aux, spill_index(m, slot));
}
__device__ __forceinline__ void aux_store(const LaneMem& m, uint32_t slot, uint32_t v) {
  if (slot < m.hot) m.pend_aux[slot * 64] = v;
  else *spill_at(m.spill_aux, spill_index(m, slot)) = v;
}

// ------------------------------------------------------------------ row interpreter
// effect word recorded per effect row: op[4:0] | type[9:5] | target[13:10] | p0[21:14] | p1[29:22]
// wide: op[4:0] | type[9:5] | target[13:10] | payload area[61:14] (demi_device.hpp pay_area: p0[29:14] | p1[45:30] for the two
// fields of a plain wide table); BIG: target[14:10] (FX_NOBODY = 31) | payload area[62:15]
constexpr uint32_t FX_NOBODY = BIG_TU ? 31u : 15u, FX_AREA_SHIFT = BIG_TU ? 15u : 14u;
__device__ __forceinline__ uint32_t fx_target(uint32_t fx_low) { return (fx_low >> 10) & FX_NOBODY; }
#ifdef DEMI_WIDE
__device__ __forceinline__ word_t fx_pack_area(uint32_t op, uint32_t type, uint32_t target, uint64_t area) {
  return (word_t)((op & 31u) | (type << 5) | (target << 10)) | ((word_t)area << FX_AREA_SHIFT);
}
__device__ __forceinline__ word_t fx_pack(uint32_t op, uint32_t type, uint32_t target, uint32_t p0, uint32_t p1) {
  return fx_pack_area(op, type, target, pay_area(p0, p1));
}
__device__ __forceinline__ uint64_t fx_area(word_t fx) { return (fx >> FX_AREA_SHIFT) & 0xFFFFFFFFFFFFull; }
// the message an effect word sends: type / payload from the word, sender and receiver from the caller
__device__ __forceinline__ word_t fx_msg_word(word_t fx, uint32_t type, uint32_t src, uint32_t dst) {
  return msg_word_area(type, src, dst, fx_area(fx));
}
#else
__device__ __forceinline__ uint32_t fx_pack(uint32_t op, uint32_t type, uint32_t target, uint32_t p0, uint32_t p1) {
  return (op & 31u) | (type << 5) | (target << 10) | (p0 << 14) | (p1 << 22);
}
__device__ __forceinline__ uint32_t fx_p0(uint32_t fx) { return (fx >> 14) & 0xFFu; }
__device__ __forceinline__ uint32_t fx_p1(uint32_t fx) { return (fx >> 22) & 0xFFu; }
__device__ __forceinline__ uint64_t fx_area(uint32_t fx) { return (uint64_t)(fx_p0(fx) | (fx_p1(fx) << 16)); }
__device__ __forceinline__ uint32_t fx_msg_word(uint32_t fx, uint32_t type, uint32_t src, uint32_t dst) {
  return msg_word(type, src, dst, fx_p0(fx), fx_p1(fx));
}
#endif

// 16 x u8 register window held in four VGPRs: w0,w1 = r0..r7 (state), w2,w3 = r8..r15 (temps, payload,
// sender, self).  v_perm_b32 extracts / inserts one byte without variable 64-bit shifts.
__device__ __forceinline__ uint32_t reg_get4(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t i) {
  const uint32_t sel = (i & 7u) | 0x0C0C0C00u;                 // byte i&7 of the 64-bit pair, upper bytes zero
  const uint32_t lo = __builtin_amdgcn_perm(w1, w0, sel);
  const uint32_t hi = __builtin_amdgcn_perm(w3, w2, sel);
  return (i & 8u) ? hi : lo;
}

__device__ __forceinline__ uint32_t mask_of(uint32_t cw, uint32_t bit_index) {
  return (uint32_t)__builtin_amdgcn_sbfe(cw, bit_index, 1);   // v_bfe_i32: 0 or ~0
}

// Runs the handler of message word `w` on its receiver.  State is read from / written to
// mem.st; effect rows are recorded into mem.fxq.  Returns the number of recorded effect rows
// (sets DEMI_V_QUEUE_OVF in flags when more than DEMI_FX_CAP would be recorded).
// DEMI_OP_RND: the application's generator (Instrumenter().seededRandom, restarted at seed 0 by every execution)
__device__ __forceinline__ uint32_t app_next_int(uint64_t& app_rng, uint32_t bound, const uint32_t* magic) {
  return bound == 0 ? 0u : jr_next_int(app_rng, bound, magic);
}

#ifndef DEMI_WIDE     // the row interpreter exists for the 8-bit window only: a wide table always runs as generated code
__device__ inline uint32_t vm_run(const Tables& t, const LaneMem& mem, uint32_t w, uint32_t& flags, uint64_t& app_rng) {
  const uint32_t type = w_type(w), me = w_dst(w);
  uint32_t pc = t.hs[((t.ac_packed >> (4 * me)) & 15u) * t.NT + type];
  if (pc == 0xFFFFu) return 0;
  const uint64_t st0 = mem.st[me * 64];
  uint32_t w0 = (uint32_t)st0, w1 = (uint32_t)(st0 >> 32);
  uint32_t w2 = 0;                                                      // T0..T3
  uint32_t w3 = w_p0(w) | (w_p1(w) << 8) | (w_src(w) << 16) | (me << 24);  // P0 P1 SRC ME
  uint32_t nfx = 0;
  const uint32_t code_len = t.code_len;
  bool running = true;
  while (running) {
    const uint32_t row = t.code[pc];
    pc++;
    const uint32_t cw = t.optab[row & 0x3Fu];
    const uint32_t dsti = (row >> 8) & 15u, ai = (row >> 12) & 15u, aux = (row >> 17) & 0x7Fu, braw = row >> 24;
    const uint32_t a = reg_get4(w0, w1, w2, w3, ai);
    const uint32_t breg = reg_get4(w0, w1, w2, w3, braw);
    const uint32_t b = (row & 0x10000u) ? braw : breg;
    // ---- relation of a and b: 0 (a<b), 1 (a==b), 2 (a>b), tested against the row's accepted set
    const int32_t d = (int32_t)a - (int32_t)b;
    const uint32_t rel = (uint32_t)(min(max(d, -1), 1) + 1);                 // v_med3_i32
    const uint32_t cond = (cw >> (CW_REL_SHIFT + rel)) & 1u;
    const uint32_t ltm = (uint32_t)(d >> 31);                                // ~0 iff a < b
    // ---- ALU: mutually exclusive (candidate & mask) terms
    const uint32_t sh = b & 7u;
    const uint32_t nm = mask_of(cw, 2);
    uint32_t r = ((a & mask_of(cw, 1)) + (b ^ nm) + (nm & 1u)) & mask_of(cw, 0);                 // MOV ADD SUB
    r |= (a & b) & mask_of(cw, 3);                                                                // AND
    const uint32_t bm = mask_of(cw, 5);
    r |= (a | ((b & ~bm) | ((1u << sh) & bm))) & mask_of(cw, 4);                                  // OR BITSET
    r |= (a ^ b) & mask_of(cw, 6);                                                                // XOR
    r |= (a << sh) & mask_of(cw, 7);                                                              // SHL
    r |= (a >> sh) & mask_of(cw, 8);                                                              // SHR
    r |= (uint32_t)__popc(b) & mask_of(cw, 9);                                                    // POPC
    r |= cond & mask_of(cw, 10);                                                                  // EQ..GT
    const uint32_t mn = mask_of(cw, 12);                 // MIN: b ^ ((a^b) & lt)   MAX: a ^ ((a^b) & lt)
    r |= (((b & mn) | (a & ~mn)) ^ ((a ^ b) & ltm)) & mask_of(cw, 11);
    if (cw & CW_RND) r = app_next_int(app_rng, b, t.gmagic);                                       // RND (rare: a real branch)
    // ---- write-back, branch-free: insert byte r into word dsti>>2 when the row is an ALU row
    const uint32_t k8 = (dsti & 3u) * 8u;
    const uint32_t ins = 0x03020100u ^ ((((dsti & 3u) ^ 4u)) << k8);         // selector: byte k := S0.byte0
    const uint32_t wsel = (cw & CW_ALU) ? (dsti >> 2) : 4u;
    w0 = __builtin_amdgcn_perm(r, w0, wsel == 0 ? ins : 0x03020100u);
    w1 = __builtin_amdgcn_perm(r, w1, wsel == 1 ? ins : 0x03020100u);
    w2 = __builtin_amdgcn_perm(r, w2, wsel == 2 ? ins : 0x03020100u);
    w3 = __builtin_amdgcn_perm(r, w3, wsel == 3 ? ins : 0x03020100u);
    // ---- forward skips
    const uint32_t zf = (a == 0) ? CW_SKIPZ : CW_SKIPNZ;
    const bool skip_if = (cw & CW_IF) && (cond == 0);
    const bool skip = skip_if | ((cw & (zf | CW_SKIP)) != 0);
    pc += skip ? (skip_if ? aux : braw) : 0u;
    // ---- effect rows: recorded now, applied after the rows have run
    if (cw & CW_FX) {
      if (nfx >= DEMI_FX_CAP) { flags |= DEMI_V_QUEUE_OVF; }
      else {
        const uint32_t p0 = reg_get4(w0, w1, w2, w3, dsti);
        mem.fxq[nfx * 64] = fx_pack(row & 0xFFu, aux, a > 15u ? 15u : a, p0, b);     // target 15 = nobody
        nfx++;
      }
    }
    running = !(cw & CW_HALT) & (pc < code_len) & !(flags & DEMI_V_QUEUE_OVF);
  }
  mem.st[me * 64] = (uint64_t)w0 | ((uint64_t)w1 << 32);
  return nfx;
}
#endif  // !DEMI_WIDE

// ------------------------------------------------------------------ invariant
// Invariant descriptor (TestOracle.scala:27 `Invariant`) on the simulated state; returns the ViolationFingerprint code.
// st: this lane's actor states in LDS, stride 64 u64.  Per actor: does it count ("hit"), and under which key; the kinds
// combine the actors (include/demi_gpu.h).  With DEMI_INV_PROGRAM the per-actor part is a row program (first row inv_fa)
// run on that actor's state: r0..r7 = its fields, r15 = its id, everything else 0; hit = T0 != 0, key = T1.  A compiled table
// brings it as generated code (jit.hpp: inv_prog_jit); the interpreter below runs it for the others.
#if defined(DEMI_JIT_INV_PROG)
__device__ inline uint32_t inv_prog_jit(const uint64_t* st, uint32_t actor, uint32_t& key, uint32_t exists, uint32_t n_actors);     // (generated: jit.hpp)
#endif
// DEMI_OP_PEER: field `f` (0..7; 8 = "is created") of actor `who` in this lane's state array; not a created actor: 0
__device__ __forceinline__ uint32_t peer_field(const uint64_t* st, uint32_t who, uint32_t f, uint32_t exists, uint32_t n_actors) {
  if (who >= n_actors || !((exists >> who) & 1u)) return 0u;
  if (f >= 8u) return 1u;
#ifdef DEMI_WIDE
  return (uint32_t)(st[(ST_WORDS * who + (f >> 2)) * 64] >> (16 * (f & 3))) & 0xFFFFu;
#else
  return (uint32_t)(st[(ST_WORDS * who) * 64] >> (8 * f)) & 0xFFu;
#endif
}
// the invariant's kind: a compile-time constant in a translation unit compiled for one table (the other kinds' code is not
// even generated there), the loaded model's otherwise
#ifdef DEMI_JIT_INV_KIND
#define DEMI_INV_KIND_OF(T) DEMI_JIT_INV_KIND
#else
#define DEMI_INV_KIND_OF(T) ((T).inv_kind)
#endif
#ifndef DEMI_WIDE
// the pure part of vm_run: ALU / SKIP / IF rows from `pc` until HALT or the end of the table
__device__ inline uint32_t inv_prog_interp(const Tables& t, const uint64_t* st, uint32_t actor, uint32_t& key) {
  uint32_t pc = t.inv_fa;
  const uint64_t st0 = st[actor * 64];
  uint32_t w0 = (uint32_t)st0, w1 = (uint32_t)(st0 >> 32), w2 = 0, w3 = actor << 24;
  bool running = pc < t.code_len;
  while (running) {
    const uint32_t row = t.code[pc];
    pc++;
    const uint32_t cw = t.optab[row & 0x3Fu];
    const uint32_t dsti = (row >> 8) & 15u, ai = (row >> 12) & 15u, aux = (row >> 17) & 0x7Fu, braw = row >> 24;
    const uint32_t a = reg_get4(w0, w1, w2, w3, ai);
    const uint32_t breg = reg_get4(w0, w1, w2, w3, braw);
    const uint32_t b = (row & 0x10000u) ? braw : breg;
    const int32_t d = (int32_t)a - (int32_t)b;
    const uint32_t rel = (uint32_t)(min(max(d, -1), 1) + 1);
    const uint32_t cond = (cw >> (CW_REL_SHIFT + rel)) & 1u;
    const uint32_t ltm = (uint32_t)(d >> 31);
    const uint32_t sh = b & 7u;
    const uint32_t nm = mask_of(cw, 2);
    uint32_t r = ((a & mask_of(cw, 1)) + (b ^ nm) + (nm & 1u)) & mask_of(cw, 0);
    r |= (a & b) & mask_of(cw, 3);
    const uint32_t bm = mask_of(cw, 5);
    r |= (a | ((b & ~bm) | ((1u << sh) & bm))) & mask_of(cw, 4);
    r |= (a ^ b) & mask_of(cw, 6);
    r |= (a << sh) & mask_of(cw, 7);
    r |= (a >> sh) & mask_of(cw, 8);
    r |= (uint32_t)__popc(b) & mask_of(cw, 9);
    r |= cond & mask_of(cw, 10);
    const uint32_t mn = mask_of(cw, 12);
    r |= (((b & mn) | (a & ~mn)) ^ ((a ^ b) & ltm)) & mask_of(cw, 11);
    if (cw & CW_PEER) r = peer_field(st, a, aux, t.exists, t.A);                                  // PEER (rare: a real branch)
    const uint32_t k8 = (dsti & 3u) * 8u;
    const uint32_t ins = 0x03020100u ^ ((((dsti & 3u) ^ 4u)) << k8);
    const uint32_t wsel = (cw & CW_ALU) ? (dsti >> 2) : 4u;
    w0 = __builtin_amdgcn_perm(r, w0, wsel == 0 ? ins : 0x03020100u);
    w1 = __builtin_amdgcn_perm(r, w1, wsel == 1 ? ins : 0x03020100u);
    w2 = __builtin_amdgcn_perm(r, w2, wsel == 2 ? ins : 0x03020100u);
    w3 = __builtin_amdgcn_perm(r, w3, wsel == 3 ? ins : 0x03020100u);
    const uint32_t zf = (a == 0) ? CW_SKIPZ : CW_SKIPNZ;
    const bool skip_if = (cw & CW_IF) && (cond == 0);
    const bool skip = skip_if | ((cw & (zf | CW_SKIP)) != 0);
    pc += skip ? (skip_if ? aux : braw) : 0u;
    running = !(cw & CW_HALT) & (pc < t.code_len);
  }
  key = (w2 >> 8) & 0xFFu;      // T1
  return w2 & 0xFFu;            // T0
}
#endif
// hit (non-zero = the actor counts) and key of one actor under a DEMI_INV_PROGRAM invariant
__device__ __forceinline__ uint32_t inv_prog(const Tables& t, const uint64_t* st, uint32_t actor, uint32_t& key) {
#if defined(DEMI_JIT_INV_PROG)
  return inv_prog_jit(st, actor, key, t.exists, t.A);
#elif !defined(DEMI_WIDE)
  return inv_prog_interp(t, st, actor, key);
#else
  (void)t; (void)st; (void)actor; key = 0;
  return 0u;                    // (a wide table always runs as generated code)
#endif
}

// One actor's contribution to the invariant's "hit" mask: F[fa] == va (F[fa] != 0 for AGREE), or its program's T0
__device__ __forceinline__ uint32_t invariant_hit_at(const Tables& t, const uint64_t* st, uint32_t actor, uint32_t kind, uint32_t fa, uint32_t va) {
  if (kind & DEMI_INV_PROGRAM) { uint32_t key; return inv_prog(t, st, actor, key) != 0u; }
  const uint32_t a = state_field(st, actor, fa);
  return ((kind & 0xFFu) == DEMI_INV_AGREE) ? (a != 0) : (a == va);
}

// The verdict from the hit mask (bit i = created actor i hits).  Almost every check ends in the first two lines
// (fewer than two hits); the group keys are only read after that.
__device__ inline uint32_t invariant_from_hits(const Tables& t, const uint64_t* st, uint32_t vmask, uint32_t A, uint32_t kind, uint32_t fb) {
  vmask &= (1u << A) - 1u;   // (a specialised build knows A: the pair logic below then only exists for real actors)
  const uint32_t comb = kind & 0xFFu;
  // fingerprints: kind << 24 | key << 8 | actors; BIG: kind << 30 | (key & 0x3FFF) << 16 | actors (include/demi_gpu.h)
  constexpr uint32_t FPK = BIG_TU ? 30u : 24u;
  if (comb == DEMI_INV_NEVER) return vmask ? ((2u << FPK) | vmask) : 0u;
  if (comb == DEMI_INV_NONE || (vmask & (vmask - 1)) == 0) return 0u;   // needs at least two hits
  // slow path: group keys of the hit actors
  uint32_t key[MAX_ACT];
#pragma unroll
  for (uint32_t i = 0; i < MAX_ACT; i++) {
    key[i] = 0u;
    if (i < A) {
      if (kind & DEMI_INV_PROGRAM) { if ((vmask >> i) & 1u) (void)inv_prog(t, st, i, key[i]); }
      else key[i] = state_field(st, i, fb);
    }
  }
  if (comb == DEMI_INV_AGREE) {
    bool have = false, bad = false;
    uint32_t first = 0;
#pragma unroll
    for (uint32_t i = 0; i < MAX_ACT; i++) {
      if ((vmask >> i) & 1) {
        if (!have) { have = true; first = key[i]; }
        else if (key[i] != first) bad = true;
      }
    }
    return bad ? ((3u << FPK) | vmask) : 0u;
  }
  // AT_MOST_ONE: lowest (i, j) pair of hits with equal keys
  bool found = false;
  uint32_t k = 0;
#pragma unroll
  for (uint32_t i = 0; i < MAX_ACT; i++) {
#pragma unroll
    for (uint32_t j = i + 1; j < MAX_ACT; j++) {
      if (!found && ((vmask >> i) & 1) && ((vmask >> j) & 1) && key[i] == key[j]) { found = true; k = key[i]; }
    }
  }
  if (!found) return 0u;
  uint32_t mask = 0;
#pragma unroll
  for (uint32_t i = 0; i < MAX_ACT; i++)
    if (((vmask >> i) & 1) && key[i] == k) mask |= 1u << i;
  return BIG_TU ? (1u << 30) | ((k & 0x3FFFu) << 16) | mask : (1u << 24) | (k << 8) | mask;
}

__device__ inline uint32_t invariant_code(const Tables& t, const uint64_t* st, uint32_t exists,
                                          uint32_t A, uint32_t kind, uint32_t fa, uint32_t va, uint32_t fb) {
  uint32_t vmask = 0;
  for (uint32_t i = 0; i < A; i++) if ((exists >> i) & 1u) vmask |= invariant_hit_at(t, st, i, kind, fa, va) << i;
  return invariant_from_hits(t, st, vmask & exists, A, kind, fb);
}

}  // namespace demi
