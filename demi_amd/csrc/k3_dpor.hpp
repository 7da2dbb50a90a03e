// k3_dpor.hpp — K3: one DPORwHeuristics interleaving per wavefront lane, plus its racing pairs.
//
// Restates DPORwHeuristics.schedule_new_message / getMatchingMessage / getPendingEvent /
// event_produced / getMessage / runExternal / notify_quiescence / notify_timer_cancel
// (schedulers/DPORwHeuristics.scala:421-648, 803-847, 773-801, 684-721, 855-942, 961-984) and
// the pair loop of dpor() with isCoEnabeled / analyze_dep / getCommonPrefix (:1020-1139, 994-1018).
//
// lane = one interleaving = one `nextTrace` prefix popped from the host's backtrack queue.
// The dep-graph is a tree, so a node is the hash chain of its causal path (include/demi_gpu.h);
// the lane keeps, for every delivered event, the trace index of the delivery that produced it.
// "earlier happens-before later" (later.pathTo(earlier)) is a walk up those parent indices and the
// branch point (getCommonPrefix(...).last) is the lowest common ancestor of the two trace indices.
// The lane writes its trace (16 B per event) straight into the output array; pending messages live in
// the LDS hot slots / HBM spill of sim_core.hpp with a packed side word (producer index, quiescent
// period, FIFO sequence number).
//
// The racing-pair analysis of a finished interleaving is done by the WHOLE wave (k3_racing_pairs): the
// trace's (parent, period, receiver) words and one 256-bit ancestor set per event are staged in LDS
// (9 KB per wave); lane i owns the candidates "later = i, i + 64, ..." and all lanes walk `earlier`
// together, so every LDS read of the inner loop is a broadcast.  happens-before is one bit test, the
// branch point is the highest common bit of two ancestor sets.  Pairs are written in the sequential
// (later, earlier) order via a count pass, a wave prefix sum and a write pass.
#pragma once

#include "sim_core.hpp"
#include "k3_pairs.hpp"

#ifndef DEMI_VM_RUN   // the table interpreter, unless a specialised build supplies the compiled handlers (jit.hpp)
#define DEMI_VM_RUN vm_run
#endif

namespace demi {

constexpr uint64_t DPOR_ROOT_KEY = 0xCBF29CE484222325ULL;
constexpr uint64_t DPOR_PRIME = 0x100000001B3ULL;
__host__ __device__ inline uint64_t dpor_marker_key(uint32_t ext_idx) {
  return DPOR_ROOT_KEY ^ (0x5155494553434500ULL | (uint64_t)ext_idx);
}

struct K3Args {
  const DevModel* model;
  const uint64_t* ext;               // external events (Start / Send / WaitQuiescence)
  uint32_t n_ext;
  const demi_dpor_trace_entry* prefixes;  // [n][stride]
  const uint32_t* prefix_len;        // [n]
  const uint32_t* shared_len;        // [n] or null: leading events whose racing pairs the caller already has (demi_gpu.h)
  uint32_t stride;
  uint64_t n;
  uint32_t depth_bound, max_messages, looking_for_valid, looking_for, p_max, max_pairs, prioritize;
  uint32_t lanes_per_wave;           // simulators a wave runs at a time (8..64): a round of the backtrack queue is far
                                     // smaller than the chip, and a wave analyses its finished traces one after the
                                     // other, so few interleavings per wave on many waves finish sooner
  demi_verdict* out;
  demi_dpor_trace_entry* traces;     // [n][DEMI_DPOR_MAX_TRACE]
  uint32_t* trace_len;               // [n]
  demi_dpor_pair* pairs;             // [n][max_pairs]
  uint32_t* n_pairs;                 // [n]
  unsigned long long* work_counter;
  uint32_t* spill;
  // device-resident exploration (demi_dpor_explore, ROUNDS order): the next trace of interleaving i is not uploaded but
  // read from the trace of the interleaving that found the backtrack point, which stayed in the arena:
  // trace.take(branch + 1) ++ trace(branch + 1 .. later) minus `earlier` (:1054-1057, 1180).  `traces` then points at this
  // round's slots of the same arena.
  const DporItem* items;             // [n] or null (then prefixes / prefix_len / shared_len are used)
  const demi_dpor_trace_entry* arena;
  unsigned long long* phase_out;     // -DDEMI_K3_PHASES builds only (tools/k3_phases.sh): [waves][16] cycle totals per phase
  // checkpointed interleavings (round 5; `items` launches of one rank; snap == nullptr: off).  See K3Snap below.
  unsigned char* snap;               // [snap_ids][K3_SNAP_SLOTS] records of snap_stride bytes
  uint32_t* snap_owner;              // [snap_ids][16]: which interleaving's record holds the state after c = 16 (k + 1) events of this trace
  uint32_t snap_stride, snap_pend_cap, snap_ids;
  uint32_t first_id;                 // arena id of interleaving 0 of this launch (traces = arena + first_id rows)
};

// Checkpointed interleavings.  The next trace of a child C is its parent's trace up to the branch point, then the events to
// replay (DPORwHeuristics.scala:1054-1057, 1180): C used to re-execute the shared part - 83 of 250 deliveries on average for
// config 5, 60 of 190 for config 3 - only to arrive at the state its parent P had been in at that point.  An interleaving now
// leaves a RECORD of its state every 16 trace entries - everything a scheduling step reads: the scalars of the scheduler, the
// actors' states, the pending set with side words and node keys - and a child starts from the deepest record at or below its
// branch point, copying the trace entries in front of it from P's row of the arena.
// What makes that exact: a prefix head that is matched (getMatchingMessage, :474-537) picks the message the original divergent
// step picked (getPendingEvent, :452-472) - same slot, same swap-remove - so C's state after replaying entries [0, c) IS P's
// state when it had pushed c entries, as long as P's steps up to there were all deliveries or quiescence markers.  The one step
// that leaves no trace entry is the discard of a message to or from an isolated actor (:626-635): P removes it from the pending
// set when it is picked, C - matching prefix heads - never picks it.  An interleaving stops leaving records at its first such
// step (`asym`), and no record is taken between choosing a quiescence marker and pushing it.  Records P did not make itself -
// those below the point where P started from ITS parent's record - are found through an owner table (16 arena ids per
// interleaving, inherited at the start).  Same traces, verdicts and racing pairs (tests run both ways: DEMI_K3_NO_CHECKPOINT).
constexpr uint32_t K3_SNAP_SLOTS = 15, K3_SNAP_EVERY = 16, K3_SNAP_NONE = 0xFFFFFFFFu, K3_SNAP_MAGIC = 0x4B335350u;
struct K3SnapHdr {                   // 96 bytes; the actors' states and the pending entries follow
  unsigned long long hash, app_rng, parent_key;
  uint32_t magic, c, n_pend, flags, next_seq, count, deliveries, ext_idx, parent, parent_depth, cur_root, qperiod, next_qperiod,
      marker_ext, qmarker_ext, isolated, rep, blocked;
#ifdef DEMI_BIG
  uint32_t rep_hi, pad_big;          // (64 timer bits; the host sizes the records with k3_snap_stride(..., big))
#endif
};
__host__ __device__ inline uint32_t k3_snap_pend_bytes(bool wide) { return wide ? 24u : 16u; }       // key 8, word 4 | 8, side word 4 (+ 4 pad)
__host__ __device__ inline uint32_t k3_snap_stride(uint32_t n_actors, uint32_t st_words, uint32_t pend_cap, bool wide, bool big = BIG_TU) {
  return (uint32_t)(((big ? 104u : 96u) + 8u * n_actors * st_words + pend_cap * k3_snap_pend_bytes(wide) + 63u) & ~63u);
}

constexpr int K3_WAVES = 4;
constexpr size_t K3_ANALYSIS_BYTES = DEMI_DPOR_MAX_TRACE * 4 + DEMI_DPOR_MAX_TRACE * 32;   // meta words + ancestor sets

// The key plane (round 4): the node key of every LDS-resident pending message - key(child) = (key(producer) ^ word) * prime,
// computed when the message is produced, while the producer's key is in a register - beside its word and side word.  The prefix
// match and the delivery used to fetch the producer's trace entry from HBM for it: a dependent round trip in each of the two
// longest phases of a scheduling step (slots beyond the LDS-resident ones still do).
__host__ __device__ inline size_t k3_key_wave_bytes(uint32_t hot) { return (size_t)hot * 64 * 8; }
// (waves: wavefronts per workgroup of the launch, at most K3_WAVES; the kernel reads it from blockDim)
__host__ __device__ inline size_t k3_lds_bytes(uint32_t code_len, uint32_t n_ext, uint32_t n_hs, uint32_t n_actors, bool wide = WIDE_TU,
                                               uint32_t hot = PEND_HOT, uint32_t waves = 4, uint32_t arr_words = ARR_WORDS, bool big = BIG_TU) {
  return tables_lds_bytes(code_len, n_ext, n_hs, wide, arr_words, big) +
         waves * (lane_mem_wave_bytes(n_actors, true, hot, wide, DEMI_FX_CAP, arr_words) + k3_key_wave_bytes(hot));
}

__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v, uint32_t lane) {
#pragma unroll
  for (uint32_t d = 1; d < 64; d <<= 1) {
    const uint32_t u = __shfl_up(v, d);
    if (lane >= d) v += u;
  }
  return v;
}

// dpor()'s pair loop (:1122-1139) for one finished trace T[0..n), executed by all 64 lanes of the wave.
// Returns the number of racing pairs (wave-uniform) whose later event is at index >= shared; the first max_pairs of them
// are written to `po` in the order of the sequential loop (later ascending, earlier ascending).
// BIG (a template parameter here, not the translation unit's layout: k3_analyze is a generic kernel of the library): the trace
// entries' words carry a 4-bit receiver field
template <bool BIG>
__device__ inline uint32_t k3_racing_pairs(const demi_dpor_trace_entry* __restrict__ T, uint32_t n, uint32_t* s_meta,
                                           uint64_t* s_anc, demi_dpor_pair* __restrict__ po, uint32_t max_pairs,
                                           uint32_t lane, uint32_t shared) {
  constexpr uint32_t MSG = BIG ? 1u << 20 : 1u << 19;
  // meta word: parent | qperiod << 8 | receiver << 16 | (kind == message delivery) << 19 (BIG: << 20)
  for (uint32_t i = lane; i < n; i += 64) {
    const demi_dpor_trace_entry e = T[i];
    s_meta[i] = (uint32_t)e.parent | ((uint32_t)e.qperiod << 8) | (((e.word >> 5) & (BIG ? 15u : 7u)) << 16) | (e.kind == 1 ? MSG : 0u);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // ancestor set of event i: every trace index on the path from its producer up to the root (parents have
  // smaller indices, the root is its own parent)
  for (uint32_t i = lane; i < n; i += 64) {
    uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    uint32_t k = s_meta[i] & 0xFF;
    for (;;) {
      const uint64_t bit = 1ull << (k & 63);
      const uint32_t q = k >> 6;
      a0 |= (q == 0) ? bit : 0ull; a1 |= (q == 1) ? bit : 0ull; a2 |= (q == 2) ? bit : 0ull; a3 |= (q == 3) ? bit : 0ull;
      if (k == 0) break;
      k = s_meta[k] & 0xFF;
    }
    s_anc[i * 4 + 0] = a0; s_anc[i * 4 + 1] = a1; s_anc[i * 4 + 2] = a2; s_anc[i * 4 + 3] = a3;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // Which pairs race is a matter of 256-bit masks, not of a loop over `earlier`: later = l races with every message delivery
  // e < l that has l's receiver and quiescent period (isCoEnabeled, :1091-1110) and is not an ancestor of l, i.e.
  //   race(l) = SAME[class(l)] & ~anc(l) & below(l),     class = (receiver, quiescent period, is-a-delivery) of the meta word.
  // SAME[c] is one ballot per 64 events and distinct class (a handful: actors x quiescent periods), taken once for all the
  // lanes of that class.  The lane then owns its set of racing `earlier`s as bits, counts them with popcounts (the offsets of
  // the sequential order - later ascending, earlier ascending - are a wave prefix sum per group of 64 laters, groups in
  // order) and walks only ITS set bits to compute the branch points (analyze_dep, :1043-1077: the highest common bit of two
  // ancestor sets) and write the pairs.  The loop over every (later group, earlier) with its dependent LDS read per step that
  // this replaces took 60-75 % of k3_dpor (tools/k3_phases.sh).
  constexpr uint32_t CLS = BIG ? 0x1FFF00u : 0xFFF00u;        // quiescent period | receiver | MSG
  uint32_t mg[4];
#pragma unroll
  for (uint32_t k = 0; k < 4; k++) mg[k] = (k * 64 + lane < n) ? s_meta[k * 64 + lane] : 0u;
  uint32_t total = 0;
#pragma unroll
  for (uint32_t g = 0; g < 4; g++) {
    if (g * 64 >= n || g * 64 + 64 <= shared) continue;        // later < shared: reported by the producing interleaving
    const uint32_t l = g * 64 + lane;
    const bool lv = l < n && l >= shared && (mg[g] & MSG) != 0;
    const uint32_t cls = mg[g] & CLS;
    uint64_t same[4] = {0, 0, 0, 0};
    uint64_t todo = __ballot(lv);
    while (todo) {
      const int lead = __builtin_ctzll(todo);
      const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)cls, lead);
      const bool mine = lv && cls == c;
      uint64_t m[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) m[k] = (k <= g) ? __ballot((mg[k] & CLS) == c) : 0ull;     // (c has the MSG bit: deliveries only)
      if (mine) { same[0] = m[0]; same[1] = m[1]; same[2] = m[2]; same[3] = m[3]; }
      todo &= ~__ballot(mine);
    }
    uint64_t la[4] = {0, 0, 0, 0}, r[4] = {0, 0, 0, 0};
    uint32_t cnt = 0;
    if (lv) {
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        la[k] = s_anc[l * 4 + k];
        const uint64_t below = (k < g) ? ~0ull : (k == g) ? ((1ull << lane) - 1ull) : 0ull;      // e < l
        r[k] = same[k] & ~la[k] & below;
        cnt += (uint32_t)__popcll(r[k]);
      }
    }
    const uint32_t incl = wave_inclusive_sum(cnt, lane);
    uint32_t idx = total + incl - cnt;
    total += __shfl(incl, 63);
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      if (k > g) continue;
      // two earliers per step: the ancestor sets of both are requested from LDS before either is used (one wave, nothing
      // else to hide the latency behind)
      uint64_t bits = r[k];
      while (bits) {
        const uint32_t e0 = k * 64 + (uint32_t)__builtin_ctzll(bits);
        bits &= bits - 1;
        const bool two = bits != 0;
        const uint32_t e1 = two ? k * 64 + (uint32_t)__builtin_ctzll(bits) : e0;
        bits &= bits - 1;                       // (0 & anything stays 0)
        uint64_t x0[4], x1[4];
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
          x0[q] = (q <= k) ? (la[q] & s_anc[e0 * 4 + q]) : 0ull;
          x1[q] = (q <= k) ? (la[q] & s_anc[e1 * 4 + q]) : 0ull;
        }
        // analyze_dep (:1043-1077): branch point = deepest common ancestor of the two producers (ancestors of e are below e)
        const uint32_t b0 = x0[3] ? 255u - (uint32_t)__builtin_clzll(x0[3]) : x0[2] ? 191u - (uint32_t)__builtin_clzll(x0[2])
                          : x0[1] ? 127u - (uint32_t)__builtin_clzll(x0[1]) : 63u - (uint32_t)__builtin_clzll(x0[0] | 1ull);
        const uint32_t b1 = x1[3] ? 255u - (uint32_t)__builtin_clzll(x1[3]) : x1[2] ? 191u - (uint32_t)__builtin_clzll(x1[2])
                          : x1[1] ? 127u - (uint32_t)__builtin_clzll(x1[1]) : 63u - (uint32_t)__builtin_clzll(x1[0] | 1ull);
        if (idx < max_pairs) {
          demi_dpor_pair p; p.branch = (uint8_t)b0; p.later = (uint8_t)l; p.earlier = (uint8_t)e0; p.pad = 0;
          po[idx] = p;
        }
        idx++;
        if (two) {
          if (idx < max_pairs) {
            demi_dpor_pair p; p.branch = (uint8_t)b1; p.later = (uint8_t)l; p.earlier = (uint8_t)e1; p.pad = 0;
            po[idx] = p;
          }
          idx++;
        }
      }
    }
  }
  return total;
}

__global__ __launch_bounds__(K3_WAVES * 64) void k3_dpor(const K3Args args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tables t;
  unsigned char* wave_base = tables_load(t, smem, args.model, args.ext, args.n_ext, (1u << args.model->n_actors) - 1);     // (n_actors <= 16)
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const LaneMem mem = lane_mem_carve(wave_base + (size_t)wave * lane_mem_wave_bytes(t.A, true), t.A, true, lane,
                                     args.spill, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                                     (size_t)gridDim.x * blockDim.x);
  uint64_t* const st = mem.st;
  uint64_t* const kp = reinterpret_cast<uint64_t*>(wave_base + (size_t)(blockDim.x >> 6) * lane_mem_wave_bytes(t.A, true) +
                                                   (size_t)wave * k3_key_wave_bytes(PEND_HOT)) + lane;      // [slot * 64]
  const uint32_t A = t.A, NE = t.E, PMAX = args.p_max;
  const uint32_t max_messages = args.max_messages ? args.max_messages : 0x7FFFFFFFu;

  bool active = false, fresh = false;
  uint64_t sched = 0, hash = 0;
  uint64_t app_rng = 0;     // Instrumenter().seededRandom, restarted with every interleaving (DEMI_OP_RND)
  demi_dpor_trace_entry* tr = nullptr;
  const demi_dpor_trace_entry* pf = nullptr;
  uint32_t pfx = 0, pfx_len = 0;
  uint32_t it_branch = 0, it_earlier = 0;      // device-resident next trace: position i is pf[i] up to the branch, then
  auto pf_at = [&](uint32_t i) -> const demi_dpor_trace_entry& {      // pf[i] or pf[i + 1]: `earlier` is left out
    return pf[(args.items && i > it_branch && i >= it_earlier) ? i + 1 : i];
  };
  uint32_t n_pend = 0, next_seq = 0, parent = 0, parent_depth = 0, cur_root = 0, qperiod = 0, next_qperiod = 0;
  uint64_t parent_key = DPOR_ROOT_KEY;       // key of trace[parent]: what the messages produced now descend from
  uint32_t marker_ext = 0, qmarker_ext = 0, isolated = 0, flags = 0, count = 0, deliveries = 0;
  tmask_t rep = 0;          // registered repeating timers (bit rcv * 4 + timer index)
  uint32_t blocked = 0;     // crashed actors (DEMI_OP_CRASH): skipped by getMatchingMessage (:478, 518) and getPendingEvent (:455)
  uint32_t n_trace = 0, ext_idx = 0;
  bool awaiting = false, marker_pending = false;
  uint64_t b_next = 0, b_end = 0;
  bool exhausted = false;
  // checkpoints: this interleaving's arena id, the trace length of its last record, "a step left no trace entry"
  const bool snap_on = args.snap != nullptr && args.items != nullptr;
  uint32_t my_id = 0, last_snap_c = 0;
  bool asym = false;

#ifdef DEMI_K3_PHASES
  uint64_t ph_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_iters = 0, ph_active = 0;
#define K3_NOW(V) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V) : : "memory")
#define K3_MARK(I) do { uint64_t now_; K3_NOW(now_); ph_t[I] += now_ - ph_last; ph_last = now_; } while (0)
  uint64_t ph_last; K3_NOW(ph_last);
#else
#define K3_MARK(I) do {} while (0)
#endif
#define K3_ABORT (DEMI_OVF_ANY | DEMI_V_TRACE_OVF | DEMI_V_SELFMSG)
#define TIMER_BIT(RCV, TYPE) ((tmask_t)1 << ((RCV) * DEMI_MAX_TIMER_TYPES + (t.meta[(TYPE)] >> 8)))

  // event_produced + getMessage: the node is a child of the current parentEvent; enqueued unless
  // the depth bound is hit (:832-838)
  auto produce = [&](word_t word) {
    if (args.depth_bound && parent_depth + 1 >= args.depth_bound) return;
    if (flags & DEMI_OVF_ANY) return;
    if (n_pend >= PMAX) { flags |= DEMI_V_PENDING_OVF; return; }
    const uint32_t side = parent | (qperiod << 8) | (next_seq << 16);
    if (n_pend < PEND_HOT) {          // (the usual case under ONE residency test: word, side word and key into LDS)
      mem.pend[n_pend * 64] = word;
      mem.pend_aux[n_pend * 64] = side;
      kp[n_pend * 64] = (parent_key ^ (uint64_t)word) * DPOR_PRIME;
    } else {
      pend_store(mem, n_pend, word);
      aux_store(mem, n_pend, side);
    }
    next_seq++;
    n_pend++;
  };
  // the node key of the pending message in slot k (word cw, side word aux)
  auto key_at = [&](uint32_t k, word_t cw, uint32_t aux) -> uint64_t {
    return k < PEND_HOT ? kp[k * 64] : (tr[aux & 0xFF].key ^ (uint64_t)cw) * DPOR_PRIME;
  };
  // swap-remove of slot k (the last slot's message moves into it)
  auto pend_remove_at = [&](uint32_t k) {
    const uint32_t last = n_pend - 1;
    if (last < PEND_HOT) {            // (the whole set is LDS-resident - the usual case - under ONE test; k <= last)
      mem.pend[k * 64] = mem.pend[last * 64];
      mem.pend_aux[k * 64] = mem.pend_aux[last * 64];
      kp[k * 64] = kp[last * 64];
    } else {
      const word_t lw = pend_load(mem, last);
      const uint32_t la = aux_load(mem, last);
      if (k < PEND_HOT && k != last) kp[k * 64] = key_at(last, lw, la);
      pend_store(mem, k, lw);
      aux_store(mem, k, la);
    }
    n_pend--;
  };
  // (a wide table's trace entry reports the low half of the 64-bit message word: type, dst, src, p0 - include/demi_gpu.h)
  uint32_t pushed_depth = 0;
  // (depth_hint >= 0: the entry's depth is known - a delivery matched against the prefix is the same node as the prefix entry, and a
  // node IS its causal path - and the producer's entry need not be read back)
  auto trace_push = [&](uint64_t key, word_t word, uint32_t par, uint32_t qp, uint32_t kind, int depth_hint = -1) -> int {
    if (n_trace >= DEMI_DPOR_MAX_TRACE) { flags |= DEMI_V_TRACE_OVF; return -1; }
    demi_dpor_trace_entry e;
    e.key = key; e.word = (uint32_t)word; e.parent = (uint8_t)par; e.qperiod = (uint8_t)qp;
    e.depth = (uint8_t)(n_trace == 0 ? 0 : depth_hint >= 0 ? (uint32_t)depth_hint : tr[par].depth + 1);
    e.kind = (uint8_t)kind;
    tr[n_trace] = e;
    pushed_depth = e.depth;                    // (the caller's setParentEvent: not read back from the trace)
    return (int)n_trace++;
  };
  // runExternal (:684-721)
  auto run_external = [&]() {
    bool await = false;
    while (ext_idx < NE && !await) {
      const uint64_t ev = t.trace[ext_idx];
      const uint32_t kind = (uint32_t)ev & 0xFF, a = (uint32_t)(ev >> 8) & 0xFF;
      if (kind == DEMI_EV_START) isolated &= ~(1u << a);
#ifdef DEMI_JIT_NPAY
      else if (kind == DEMI_EV_SEND)      // (more than two payload fields: the Send's whole area, behind the events - demi_ext_payload_areas)
        produce(msg_word_area((uint32_t)(ev >> 24) & 0xFF, DL, a, args.ext[EXT_AREA_OFFSET + ext_idx] & 0xFFFFFFFFFFFFull));
#else
      else if (kind == DEMI_EV_SEND)
        produce(msg_word((uint32_t)(ev >> 24) & 0xFF, DL, a,
                         ((uint32_t)(ev >> 32) & 0xFF) | (WIDE_TU ? ((uint32_t)(ev >> 48) & 0xFF) << 8 : 0u),
                         ((uint32_t)(ev >> 40) & 0xFF) | (WIDE_TU ? ((uint32_t)(ev >> 56) & 0xFF) << 8 : 0u)));
#endif
      else if (kind == DEMI_EV_WAIT_QUIESCENCE) { marker_pending = true; marker_ext = ext_idx; await = true; }
      ext_idx++;
    }
  };

  for (;;) {
    {
      const uint64_t idle = __ballot(!active && lane < args.lanes_per_wave);
      if (idle != 0 && !exhausted) {
        const uint32_t want = (uint32_t)__popcll(idle);
        const uint64_t have = b_end - b_next;
        uint64_t got = 0;
        if (have < want) {
          if (lane == 0) got = atomicAdd(args.work_counter, (unsigned long long)args.lanes_per_wave);
          got = __shfl(got, 0);
        }
        if (!active && lane < args.lanes_per_wave) {
          const uint32_t rank = (uint32_t)__popcll(idle & ((1ULL << lane) - 1));
          const uint64_t my = (rank < have) ? (b_next + rank) : (got + (rank - have));
          if (my < args.n) { sched = my; active = true; fresh = true; }
        }
        if (have < want) { b_next = got + (want - have); b_end = got + args.lanes_per_wave; }
        else b_next += want;
        if (b_next >= args.n) exhausted = true;
      }
      if (__ballot(active) == 0) break;
    }
    K3_MARK(0);
#ifdef DEMI_K3_PHASES
    ph_iters++; ph_active += __popcll(__ballot(active));
#endif

    word_t w = 0;
    bool deliver = false, finish = false;
    if (active) {
      if (fresh) {
        fresh = false;
        tr = args.traces + sched * DEMI_DPOR_MAX_TRACE;
        if (args.items) {
          const DporItem it = args.items[sched];
          const bool first = it.src == 0xFFFFFFFFu;
          pf = args.arena + (size_t)(first ? 0u : it.src) * DEMI_DPOR_MAX_TRACE;
          it_branch = it.branch; it_earlier = it.earlier;
          pfx_len = first ? 0u : (uint32_t)it.later;
        } else {
          pf = args.prefixes + sched * (uint64_t)args.stride;
          pfx_len = args.prefix_len[sched];
        }
        // ---- start from a record of the parent's (or of an ancestor's) state, if there is one at or below the branch point
        const unsigned char* rec = nullptr;
        asym = false; last_snap_c = 0;
        if (snap_on) {
          my_id = args.first_id + (uint32_t)sched;
          const DporItem it = args.items[sched];
          uint32_t* const mine = my_id < args.snap_ids ? args.snap_owner + (size_t)my_id * 16 : nullptr;
          uint32_t have = 0;                       // records 0 .. have - 1 of the parent are at or below the branch point
          const uint32_t* theirs = nullptr;
          if (it.src != 0xFFFFFFFFu && it.src < args.snap_ids) {
            theirs = args.snap_owner + (size_t)it.src * 16;
            have = ((uint32_t)it.branch + 1u) / K3_SNAP_EVERY;
            if (have > K3_SNAP_SLOTS) have = K3_SNAP_SLOTS;
          }
          uint32_t use = 0;                        // the deepest of them that exists: record use - 1
          for (uint32_t k = have; k > 0 && !use; k--) if (theirs[k - 1] != K3_SNAP_NONE) use = k;
          if (mine) for (uint32_t k = 0; k < 16; k++) mine[k] = (k < use) ? theirs[k] : (k == 15 ? 0u : K3_SNAP_NONE);      // ([15]: statistics)
          if (use) {
            const uint32_t owner = theirs[use - 1];
            rec = args.snap + ((size_t)owner * K3_SNAP_SLOTS + (use - 1)) * args.snap_stride;
            if (reinterpret_cast<const K3SnapHdr*>(rec)->magic != K3_SNAP_MAGIC || reinterpret_cast<const K3SnapHdr*>(rec)->c != use * K3_SNAP_EVERY) rec = nullptr;
          }
        }
        if (rec) {
          const K3SnapHdr h = *reinterpret_cast<const K3SnapHdr*>(rec);
          hash = h.hash; app_rng = h.app_rng; parent_key = h.parent_key;
          n_pend = h.n_pend; flags = h.flags; next_seq = h.next_seq; count = h.count; deliveries = h.deliveries; ext_idx = h.ext_idx;
          parent = h.parent; parent_depth = h.parent_depth; cur_root = h.cur_root; qperiod = h.qperiod; next_qperiod = h.next_qperiod;
          marker_ext = h.marker_ext & 0xFFFFu; marker_pending = (h.marker_ext >> 16) != 0; qmarker_ext = h.qmarker_ext;
          isolated = h.isolated; rep = h.rep; blocked = h.blocked;
#ifdef DEMI_BIG
          rep |= (tmask_t)h.rep_hi << 32;
#endif
          awaiting = false;
          const unsigned long long* sw = reinterpret_cast<const unsigned long long*>(rec + sizeof(K3SnapHdr));
          for (uint32_t a = 0; a < A * ST_WORDS; a++) st[a * 64] = sw[a];
          const unsigned char* pe = rec + sizeof(K3SnapHdr) + 8u * A * ST_WORDS;
          for (uint32_t k = 0; k < n_pend; k++, pe += k3_snap_pend_bytes(WIDE_TU)) {
            const unsigned long long key = *reinterpret_cast<const unsigned long long*>(pe);
            const word_t word = *reinterpret_cast<const word_t*>(pe + 8);
            const uint32_t side = *reinterpret_cast<const uint32_t*>(pe + 8 + sizeof(word_t));
            if (k < PEND_HOT) { mem.pend[k * 64] = word; mem.pend_aux[k * 64] = side; kp[k * 64] = key; }
            else { pend_store(mem, k, word); aux_store(mem, k, side); }
          }
          if (my_id < args.snap_ids) args.snap_owner[(size_t)my_id * 16 + 15] = h.c;      // (statistics: the entries this interleaving did not execute)
          n_trace = h.c;                            // the trace in front of the record: the parent's entries (they are below the branch point)
          for (uint32_t i = 0; i < n_trace; i++) tr[i] = pf[i];
          pfx = n_trace;                            // ... and as many prefix positions are consumed
          last_snap_c = n_trace;
        } else {
        pfx = 0;
        hash = 0xCBF29CE484222325ULL;
        app_rng = jr_seed(0);
        isolated = (1u << A) - 1;      // maybeStartActors: every actor exists and is isolated (:666-679)
        for (uint32_t a = 0; a < A * ST_WORDS; a++) st[a * 64] = t.init[a];
        n_pend = 0; next_seq = 0; qperiod = 0; next_qperiod = 0; rep = 0; flags = 0; count = 0; deliveries = 0; blocked = 0;
        n_trace = 0; ext_idx = 0; awaiting = false; marker_pending = false;
        trace_push(DPOR_ROOT_KEY, 0, 0, 0, 0);   // currentTrace += getRootEvent (:336-343)
        parent = 0; parent_depth = 0; cur_root = 0; parent_key = DPOR_ROOT_KEY;
        run_external();
        }
        K3_MARK(1);
      }
      // ---- leave a record of this state (see K3Snap above): the first scheduling step at a multiple of 16 trace entries
      if (snap_on && !asym && !awaiting && !(flags & K3_ABORT) && n_trace != last_snap_c && n_trace >= K3_SNAP_EVERY &&
          (n_trace & (K3_SNAP_EVERY - 1)) == 0 && n_trace <= K3_SNAP_EVERY * K3_SNAP_SLOTS && my_id < args.snap_ids) {
        last_snap_c = n_trace;
        if (n_pend <= args.snap_pend_cap) {
          const uint32_t slot = n_trace / K3_SNAP_EVERY - 1;
          unsigned char* rec = args.snap + ((size_t)my_id * K3_SNAP_SLOTS + slot) * args.snap_stride;
          K3SnapHdr h;
          h.hash = hash; h.app_rng = app_rng; h.parent_key = parent_key;
          h.magic = K3_SNAP_MAGIC; h.c = n_trace; h.n_pend = n_pend; h.flags = flags; h.next_seq = next_seq; h.count = count;
          h.deliveries = deliveries; h.ext_idx = ext_idx; h.parent = parent; h.parent_depth = parent_depth; h.cur_root = cur_root;
          h.qperiod = qperiod; h.next_qperiod = next_qperiod; h.marker_ext = marker_ext | (marker_pending ? 1u << 16 : 0u);
          h.qmarker_ext = qmarker_ext; h.isolated = isolated; h.rep = (uint32_t)rep; h.blocked = blocked;
#ifdef DEMI_BIG
          h.rep_hi = (uint32_t)(rep >> 32); h.pad_big = 0;
#endif
          *reinterpret_cast<K3SnapHdr*>(rec) = h;
          unsigned long long* sw = reinterpret_cast<unsigned long long*>(rec + sizeof(K3SnapHdr));
          for (uint32_t a = 0; a < A * ST_WORDS; a++) sw[a] = st[a * 64];
          unsigned char* pe = rec + sizeof(K3SnapHdr) + 8u * A * ST_WORDS;
          for (uint32_t k = 0; k < n_pend; k++, pe += k3_snap_pend_bytes(WIDE_TU)) {
            const word_t word = pend_load(mem, k);
            const uint32_t side = aux_load(mem, k);
            *reinterpret_cast<unsigned long long*>(pe) = key_at(k, word, side);
            *reinterpret_cast<word_t*>(pe + 8) = word;
            *reinterpret_cast<uint32_t*>(pe + 8 + sizeof(word_t)) = side;
          }
          args.snap_owner[(size_t)my_id * 16 + slot] = my_id;
        }
      }
      if (flags & K3_ABORT) {
        finish = true;
      } else {
        // ------------------------------------------------------ schedule_new_message (:421-648)
        int chosen = -1, chosen_depth = -1;
        bool chose_marker = false, none = false;
        count++;                                         // messagesScheduledSoFar += 1 (:583)
        if (count > max_messages) none = true;           // (:584-586)
        if (!none && !awaiting) {
          // getMatchingMessage: skip root / id-0 heads (:363-372), then match the head by identity; with
          // prioritizePendingUponDivergence keep popping heads until one is pending (getNextMatchingMessage :537-550)
          do {
            // (one 16-byte read of the head: the kind test used to be a read of its own in front of it - two trips to the arena
            // in a row; only the root entry at position 0 is ever skipped)
            if (pfx >= pfx_len) break;
            demi_dpor_trace_entry want = pf_at(pfx);
            while (want.kind == 0) {
              pfx++;
              if (pfx >= pfx_len) break;
              want = pf_at(pfx);
            }
            if (pfx >= pfx_len) break;
            pfx++;
            if (want.kind == 2) {
              if (marker_pending && want.key == dpor_marker_key(marker_ext)) chose_marker = true;
            } else {
              // first which pending slots hold this message word (a branch-free pass: its LDS / scratch reads do not wait for
              // one another), then the identity test - producer's key, FIFO order - for those few
              // (round 4: the LDS-resident slots in ONE unrolled pass - every slot a read at a constant offset, a compare and a
              // constant bit, slots past n_pend masked off afterwards - instead of a loop with a residency test, a 64-bit
              // variable shift and a branch per slot: the scan was 40 % of a scheduling step; the slots in the HBM scratch follow)
              uint32_t best_seq = 0xFFFFFFFFu;
              uint64_t hit0 = 0, hit1 = 0;
              if (!((blocked >> w_dst(want.word)) & 1u)) {
#pragma unroll
                for (uint32_t c = 0; c < PEND_HOT; c += 8) {          // (eight slots at a time, as far as some lane's set reaches)
                  if (c < n_pend) {
#pragma unroll
                    for (uint32_t k = c; k < c + 8 && k < PEND_HOT; k++) {
                      const uint64_t h = ((uint32_t)mem.pend[k * 64] == want.word) ? 1ull : 0ull;   // (the entry holds the word's low half)
                      if (k < 64) hit0 |= h << k; else hit1 |= h << (k - 64);
                    }
                  }
                }
                if (n_pend < 64) hit0 &= (1ull << n_pend) - 1ull;
                if (n_pend <= 64) hit1 = 0; else if (n_pend < 128) hit1 &= (1ull << (n_pend - 64)) - 1ull;
                for (uint32_t k = PEND_HOT; k < n_pend; k++) {
                  const uint64_t h = ((uint32_t)pend_load(mem, k) == want.word) ? 1ull : 0ull;
                  if (k < 64) hit0 |= h << k; else hit1 |= h << (k - 64);
                }
              }
              while (hit0 | hit1) {
                const uint32_t k = hit0 ? (uint32_t)__builtin_ctzll(hit0) : 64u + (uint32_t)__builtin_ctzll(hit1);
                if (hit0) hit0 &= hit0 - 1; else hit1 &= hit1 - 1;
                const word_t cw = pend_load(mem, k);
                const uint32_t aux = aux_load(mem, k);
                const uint64_t key = key_at(k, cw, aux);
                if (key == want.key && (aux >> 16) < best_seq) { best_seq = aux >> 16; chosen = (int)k; chosen_depth = (int)want.depth; }
              }
            }
          } while (args.prioritize && chosen < 0 && !chose_marker);
        }
        K3_MARK(2);
        if (!none && chosen < 0 && !chose_marker) {
          // getPendingEvent (:452-472), iteration order pinned: (snd, rcv) ascending, FIFO inside
          uint32_t best = 0xFFFFFFFFu;
#pragma unroll
          for (uint32_t c = 0; c < PEND_HOT; c += 8) {           // (the LDS-resident slots, unrolled by eight: constant offsets, no residency test)
            if (c < n_pend) {
#pragma unroll
              for (uint32_t k = c; k < c + 8 && k < PEND_HOT; k++) {
                const word_t pw = mem.pend[k * 64];
                const uint32_t ord = (((w_src(pw) << 4) | w_dst(pw)) << 16) | (mem.pend_aux[k * 64] >> 16);
                const bool ok = k < n_pend && !((blocked >> w_dst(pw)) & 1u) && ord < best;      // !(blockedActors contains k._2) (:455)
                best = ok ? ord : best;
                chosen = ok ? (int)k : chosen;
              }
            }
          }
          for (uint32_t k = PEND_HOT; k < n_pend; k++) {         // (the slots in the HBM scratch)
            const word_t pw = pend_load(mem, k);
            const uint32_t ord = (((w_src(pw) << 4) | w_dst(pw)) << 16) | (aux_load(mem, k) >> 16);
            const bool ok = !((blocked >> w_dst(pw)) & 1u) && ord < best;
            best = ok ? ord : best;
            chosen = ok ? (int)k : chosen;
          }
          if (chosen < 0 && marker_pending) chose_marker = true;
          if (chosen < 0 && !chose_marker) none = true;
        }
        K3_MARK(3);
        if (chose_marker) {                              // awaitQuiescenceUpdate (:256-266)
          marker_pending = false; awaiting = true; next_qperiod = marker_ext + 1; qmarker_ext = marker_ext;
        } else if (!none) {
          word_t pw; uint32_t aux; uint64_t key;
          if ((uint32_t)chosen < PEND_HOT) {        // (one residency test for the three reads)
            pw = mem.pend[chosen * 64]; aux = mem.pend_aux[chosen * 64]; key = kp[chosen * 64];
          } else {
            pw = pend_load(mem, (uint32_t)chosen); aux = aux_load(mem, (uint32_t)chosen); key = key_at((uint32_t)chosen, pw, aux);
          }
          pend_remove_at((uint32_t)chosen);
          const uint32_t snd = w_src(pw), rcv = w_dst(pw);
          if ((snd < MAX_ACT && ((isolated >> snd) & 1)) || ((isolated >> rcv) & 1)) {
            if (snd == rcv) { flags |= DEMI_V_SELFMSG; finish = true; }   // (:631-633)
            // else: discarded, schedule again (:626-635)
            asym = true;                        // (a step without a trace entry: no records of this interleaving from here on)
          } else {
            const uint32_t par = aux & 0xFF;
            const int ti = trace_push(key, pw, par, (aux >> 8) & 0xFF, 1, chosen_depth);
            if (ti < 0) finish = true;
            else {
              parent = (uint32_t)ti; parent_depth = pushed_depth; parent_key = key;   // setParentEvent
              w = pw; deliver = true;
              deliveries++;
              hash_step(hash, w);
              const uint32_t type = w_type(w), meta = t.meta[type];
              if (((meta & 0xFF) == DEMI_MSG_TIMER) && (rep & ((tmask_t)1 << (rcv * DEMI_MAX_TIMER_TYPES + (meta >> 8)))))
                produce(msg_word(type, DL, rcv, 0, 0));   // retrigger: enqueue_timer = `!`
            }
          }
        } else {
          // ---------------------------------------------------- notify_quiescence (:855-942)
          if (awaiting) {
            awaiting = false;
            qperiod = next_qperiod; next_qperiod = 0;
            const int ti = trace_push(dpor_marker_key(qmarker_ext), 0, cur_root, qperiod, 2);
            if (ti < 0) finish = true;
            else {
              cur_root = (uint32_t)ti; parent = (uint32_t)ti; parent_depth = pushed_depth; parent_key = dpor_marker_key(qmarker_ext);
              run_external();
            }
          } else {
            finish = true;
          }
        }
      }
    }

    K3_MARK(4);
    uint32_t nfx = 0;
    if (deliver) nfx = DEMI_VM_RUN(t, mem, w, flags, app_rng);
    K3_MARK(5);
    if (deliver) {
      const uint32_t me = w_dst(w);
      for (uint32_t k = 0; k < nfx && !(flags & DEMI_OVF_ANY); k++) {
        const word_t fxw = mem.fxq[k * 64];
        const uint32_t fx = (uint32_t)fxw;
        const uint32_t op = fx & 31u, type = (fx >> 5) & 31u, target = fx_target(fx);
        if (op <= DEMI_OP_BCAST) {
          const bool bc = (op == DEMI_OP_BCAST);
          const uint32_t first = bc ? 0u : target, last = bc ? A : (target < A ? target + 1 : 0u);
          for (uint32_t r = first; r < last; r++) {
            if (bc && r == me) continue;
            produce(fx_msg_word(fxw, type, me, r));
          }
        } else if (op == DEMI_OP_CRASH) {
          blocked |= 1u << me;                 // actorCrashed (Instrumenter.scala:184-199)
        } else if (op == DEMI_OP_TCANCEL) {
          // notify_timer_cancel (:961-984): first of the (deadLetters, rcv) queue with this message
          rep &= ~TIMER_BIT(me, type);
          const word_t wantw = msg_word(type, DL, me, 0, 0);
          int best = -1;
          uint32_t best_seq = 0xFFFFFFFFu;
          // (which slots hold the word: one unrolled pass over the LDS-resident slots, as in getMatchingMessage; the side words
          // only of those - usually none or one)
          uint64_t hit0 = 0, hit1 = 0;
#pragma unroll
          for (uint32_t c = 0; c < PEND_HOT; c += 8) {
            if (c < n_pend) {
#pragma unroll
              for (uint32_t q = c; q < c + 8 && q < PEND_HOT; q++) {
                const uint64_t h = (mem.pend[q * 64] == wantw) ? 1ull : 0ull;
                if (q < 64) hit0 |= h << q; else hit1 |= h << (q - 64);
              }
            }
          }
          if (n_pend < 64) hit0 &= (1ull << n_pend) - 1ull;
          if (n_pend <= 64) hit1 = 0; else if (n_pend < 128) hit1 &= (1ull << (n_pend - 64)) - 1ull;
          for (uint32_t q = PEND_HOT; q < n_pend; q++) {
            const uint64_t h = (pend_load(mem, q) == wantw) ? 1ull : 0ull;
            if (q < 64) hit0 |= h << q; else hit1 |= h << (q - 64);
          }
          while (hit0 | hit1) {
            const uint32_t q = hit0 ? (uint32_t)__builtin_ctzll(hit0) : 64u + (uint32_t)__builtin_ctzll(hit1);
            if (hit0) hit0 &= hit0 - 1; else hit1 &= hit1 - 1;
            const uint32_t sq = aux_load(mem, q) >> 16;
            if (sq < best_seq) { best_seq = sq; best = (int)q; }
          }
          if (best >= 0) pend_remove_at((uint32_t)best);
        } else {
          const tmask_t bit = TIMER_BIT(me, type);
          if (!(rep & bit)) {
            if (op == DEMI_OP_TREP) rep |= bit;
            produce(msg_word(type, DL, me, 0, 0));
          }
        }
      }
    }

    K3_MARK(6);
    // ---------------------------------------------------------- finished interleavings
    const bool fin = active && finish;
    const bool aborted = (flags & K3_ABORT) != 0;
    // dpor()'s racing pairs (:1122-1139) of the finished traces are k3_analyze's, a kernel of its own after this one (round 4):
    // in here a wave analysed its finished traces one after the other, all lanes helping and every other interleaving of the wave
    // waiting - 25-39 % of the kernel - and the analysis' 9 KB of LDS per wave halved the number of resident waves
    K3_MARK(7);
    if (fin) {
      uint32_t viol = 0;
      if (!aborted) {   // checkInvariant (:394-418)
        const uint32_t fp = invariant_code(t, st, (1u << A) - 1, A, DEMI_INV_KIND_OF(t), t.inv_fa, t.inv_va, t.inv_fb);
        if (fp) {
          if (!args.looking_for_valid) viol = fp;
          else if (((fp ^ args.looking_for) & t.fp_mask) == 0) viol = args.looking_for;
        }
      }
      for (uint32_t a = 0; a < A * ST_WORDS; a++) hash_step(hash, st[a * 64]);
      uint4 v;
      args.n_pairs[sched] = 0;                 // (k3_analyze's to fill, with DEMI_V_PAIRS_OVF in the verdict)
      if (aborted) {
        v.x = flags & K3_ABORT; v.y = 0; v.z = 0; v.w = 0;
        args.trace_len[sched] = 0;
      } else {
        v.x = (viol ? DEMI_V_VIOLATION : 0u) |
              ((count > max_messages) ? DEMI_V_MAXMSG : 0u) | ((deliveries < 0xFFFFu ? deliveries : 0xFFFFu) << 16);
        v.y = viol; v.z = (uint32_t)hash; v.w = (uint32_t)(hash >> 32);
        args.trace_len[sched] = n_trace;
      }
      *reinterpret_cast<uint4*>(&args.out[sched]) = v;
      active = false;
    }
    K3_MARK(8);
  }
#ifdef DEMI_K3_PHASES
  if (lane == 0 && args.phase_out) {
    unsigned long long* o = args.phase_out + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 16;
    for (int i = 0; i < 9; i++) o[i] = ph_t[i];
    o[14] = ph_iters; o[15] = ph_active;
  }
#endif
#undef TIMER_BIT
}

#ifndef __HIPCC_RTC__      // (a generic kernel of the library: nothing in it depends on the table)
// dpor()'s pair loop for the n interleavings k3_dpor just finished: one wave per trace, every trace of the round at once (a
// 16 384-interleaving round is sixteen waves per SIMD: the analysis' dependent LDS reads hide behind one another instead of
// holding up a wave's simulators).  Writes pairs[s][..] and n_pairs[s], and DEMI_V_PAIRS_OVF into the verdict.
constexpr int K3A_WAVES = 4;
template <bool BIG>
__device__ __forceinline__ void k3_analyze_body(const K3Args& args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t s = (uint64_t)blockIdx.x * K3A_WAVES + wave;
  if (s >= args.n) return;
  unsigned char* const an = smem + (size_t)wave * K3_ANALYSIS_BYTES;
  uint64_t* const s_anc = reinterpret_cast<uint64_t*>(an);
  uint32_t* const s_meta = reinterpret_cast<uint32_t*>(an + DEMI_DPOR_MAX_TRACE * 32);
  const uint32_t fl = args.out[s].flags, n = args.trace_len[s];
  if ((fl & K3_ABORT) || n == 0) return;                   // (aborted: k3_dpor left n_pairs = 0)
  const uint32_t shared = args.shared_len ? args.shared_len[s]
                          : (args.items && args.items[s].src != 0xFFFFFFFFu) ? (uint32_t)args.items[s].branch + 1u : 0u;
  const uint32_t total = k3_racing_pairs<BIG>(args.traces + s * DEMI_DPOR_MAX_TRACE, n, s_meta, s_anc, args.pairs + s * (uint64_t)args.max_pairs,
                                         args.max_pairs, lane, shared);
  if (lane == 0) {
    args.n_pairs[s] = total < args.max_pairs ? total : args.max_pairs;
    if (total > args.max_pairs) args.out[s].flags = fl | DEMI_V_PAIRS_OVF;
  }
}
__global__ __launch_bounds__(K3A_WAVES * 64) void k3_analyze(const K3Args args) { k3_analyze_body<false>(args); }
__global__ __launch_bounds__(K3A_WAVES * 64) void k3_analyze_big(const K3Args args) { k3_analyze_body<true>(args); }      // traces of a table with more than 8 actors
#endif
#undef K3_ABORT

}  // namespace demi
