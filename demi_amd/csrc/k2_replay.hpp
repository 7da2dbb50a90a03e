// k2_replay.hpp — K2: one STSScheduler.test (no peek) per wavefront lane: DDMin's replay oracle.
//
// Restates STSScheduler.test / advanceReplay / schedule_new_message / event_produced /
// notify_timer_cancel (schedulers/STSScheduler.scala:199-310, 405-559, 643-776, 561-623, 828-855)
// and the projection of the original trace onto a candidate subsequence
// (EventTrace.subsequenceIntersection + filterSends, EventTrace.scala:290-452).
//
// lane = candidate subsequence of the external events (a 256-bit mask).  The recorded original
// execution is lowered once on the host to a flat array of "expected" events (8 bytes each) that
// the lanes of a wave walk together, one event per step (the lock-step loop; DEMI_K2_LOCKSTEP=0 selects the older loop with a
// cursor per lane); the per-candidate projection is evaluated on the fly:
//   * a recorded Spawn/Kill/Partition/UnPartition is kept iff it equals (by name) the head of the
//     candidate's remaining non-Send externals, and dropped once those are exhausted;
//   * an external MsgSend, and the MsgEvent with the same id, are kept iff their Send is in the mask;
//   * an expected MsgEvent is delivered iff a message with the same (snd, rcv, fingerprint) is
//     pending, otherwise it is ignored ("Ignoring message", STSScheduler.scala:528-529).
// Messages with equal (snd, rcv, fingerprint) are interchangeable, so the per-key FIFO of the
// reference is realised as "any pending entry with this word" + swap-remove.
//
// k2_replay<true> goes one step further: which messages can ever be delivered is known when the original execution is
// loaded - the distinct message words of its MsgEvents (a few hundred) - so the pending set of a lane is not an array
// that every expected delivery scans (O(pending) loads, most of them in the HBM spill at p_max = 128) but one byte
// counter per such word: "is a message with this fingerprint pending" is one read, a produced message finds its
// counter through a workgroup-shared hash of the words, and a message nobody will ever ask for only counts against the
// capacity.  The lowered original trace is staged in LDS as well.  Same verdicts, bit for bit.  Where the counters live:
//   K2_FP_LDS   [word][lane] bytes in LDS: shortest dependent chain per delivery, but ~17 KB per wave for a 400-delivery
//               execution, i.e. one workgroup per CU - the choice for a DDMin frontier (10^2 - 10^4 candidates), whose
//               launch time is the serial chain of one replay and not throughput;
//   K2_FP_HBM   [word][global lane] bytes in an HBM scratch, incremented / decremented with fire-and-forget 32-bit
//               atomics on the dword that holds four lanes' counters, read with one load per expected delivery (the
//               same one round trip per delivery as K1's random pick): 4.6 KB of LDS per wave, 16-20 waves per CU -
//               the choice when there are more candidates than the LDS variant can hold at once.
//   K2_FP_WAVE  (round 4) ONE candidate per wave, its counters one byte per word in LDS (no lane dimension: ~5 KB per wave,
//               5 workgroups per CU).  Where DDMin lives - a frontier of 10^2 .. 10^4 candidates - the launch time is the
//               serial chain of one replay, and 80 % of the expected events of a candidate do nothing (their message was never
//               sent, their Send was pruned): the 64 lanes LOOK AHEAD together - lane i tests event pos + i against the
//               candidate's mask, its cursor head, its counters - and the walk jumps to the first event that acts on the
//               state (everything before it provably does nothing under the current state, and the state only changes when
//               an event acts); the skipped "Ignoring message" events are counted by a popcount.  Only the acting events are
//               stepped through one by one, on lane 0, by the same code as the lock-step walk.
#pragma once

#include "sim_core.hpp"

#ifndef DEMI_VM_RUN   // the table interpreter, unless a specialised build supplies the compiled handlers (jit.hpp)
#define DEMI_VM_RUN vm_run
#endif

namespace demi {

// expected event, 8 bytes: kind | a<<8 | b<<16 | type<<24 | p0<<32 | p1<<40 | ext<<48 | slot<<56
//   SPAWN/KILL: a = actor;  (UN)PARTITION: a, b;  MSG_SEND of an external: b = rcv, ext = Send index
//   MSG_EVENT: a = snd, b = rcv, ext = index of the Send that enqueued it (255 = internal / timer)
//   only in the lowering for filter_absents != 0: MSG_SEND of an actor (ext = 255): a = snd, b = rcv, slot = the bit of the lane's
//   pruned-sends mask that stands for this message until its MSG_EVENT (which carries the same slot; 255 = none) has passed
struct K2Args {
  const DevModel* model;
  const uint64_t* ext;      // original external events [n_ext]
  uint32_t n_ext;
  uint32_t exists;
  const uint64_t* expected; // lowered original trace [n_exp]
  uint32_t n_exp;
  uint32_t p_max, looking_for;
  uint32_t lanes_per_wave;  // candidates a wave replays at once (1..64): a DDMin frontier is small next to the chip, and a wave's
                            // step costs the union of its lanes' paths - spread thin, every candidate walks only its own
  uint32_t filter_absents;  // demi_filter_absents: EventTrace.filterKnownAbsentInternals as the last stage of the projection
  const uint64_t* masks;    // [n][4]; null = every external kept
  const uint32_t* skip;     // [n] index (in `expected`) of one MSG_EVENT removed from the trace, or null
  uint8_t* kept;            // [n][n_exp] (pre-zeroed) 1 where the expected event took effect, or null
  uint64_t n;
  demi_verdict* out;
  unsigned long long* work_counter;
  uint32_t* spill;
  // k2_replay<true>: the deliverable message words of the loaded execution
  const uint16_t* exp_fp;   // [n_exp] word id of a MSG_EVENT's message / of the message an external MSG_SEND enqueues (0xFFFF: none)
  const uint64_t* fp_hash;  // [fp_hash_mask + 1] open addressing: word | (id + 1) << 32, 0 = empty
  uint32_t fp_hash_mask;
  uint32_t n_fp;            // word ids 0 .. n_fp - 1
  uint8_t* fp_counts;       // K2_FP_HBM: [n_fp][resident lanes] counters
  uint32_t lockstep;        // 1: the lanes of a wave walk the expected events together (k2 lock-step loop below)
  unsigned long long* phase_out;   // -DDEMI_K2_PHASES builds only (tools/k2_phases.sh): [waves][16] cycle totals per phase
  const uint64_t* exp_area; // DEMI_MODEL_WIDE tables only: [n_exp] the 48-bit payload area of the expected event's message
                            // (demi_rec_event p0 | p1 << 16 | p_hi << 32; the 8-byte expected event has room for two bytes)
};

constexpr int K2_WAVES = 4;
constexpr uint32_t K2_FP_NONE = 0xFFFFu;
enum : int { K2_SCAN = 0, K2_FP_LDS = 1, K2_FP_HBM = 2, K2_FP_WAVE = 3 };

// LDS of k2_replay<true>: tables | expected (8 B each) | exp_fp (2 B each) | word hash | per wave: states, effect queue, counters
__host__ __device__ inline size_t k2_fp_shared_bytes(uint32_t n_exp, uint32_t hash_slots) {
  return (((size_t)n_exp * 8 + (size_t)n_exp * 2 + 15) & ~(size_t)15) + (size_t)hash_slots * 8;
}
// (mode: K2_FP_LDS counters [word][lane], K2_FP_WAVE one byte per word, K2_FP_HBM none in LDS)
__host__ __device__ inline size_t k2_fp_wave_bytes(uint32_t n_actors, uint32_t n_fp, int mode) {
  const size_t counters = mode == K2_FP_LDS ? (size_t)n_fp * 64 : mode == K2_FP_WAVE ? (size_t)n_fp : 0;
  return (size_t)n_actors * 64 * 8 + (size_t)DEMI_FX_CAP * 64 * 4 + ((counters + 15) & ~(size_t)15);
}
__host__ __device__ inline size_t k2_fp_lds_bytes(uint32_t code_len, uint32_t n_ext, uint32_t n_hs, uint32_t n_actors,
                                                  uint32_t n_exp, uint32_t hash_slots, uint32_t n_fp, int mode) {
  return tables_lds_bytes(code_len, n_ext, n_hs) + k2_fp_shared_bytes(n_exp, hash_slots) +
         K2_WAVES * k2_fp_wave_bytes(n_actors, n_fp, mode);
}

__host__ __device__ inline size_t k2_lds_bytes(uint32_t code_len, uint32_t n_ext, uint32_t n_hs, uint32_t n_actors, bool wide = WIDE_TU,
                                               uint32_t hot = PEND_HOT, uint32_t arr_words = ARR_WORDS, bool big = BIG_TU) {
  return tables_lds_bytes(code_len, n_ext, n_hs, wide, arr_words, big) + K2_WAVES * lane_mem_wave_bytes(n_actors, false, hot, wide, DEMI_FX_CAP, arr_words);
}

template <int MODE>
__device__ __forceinline__ void k2_replay_body(const K2Args& args) {
  constexpr bool FP = MODE != K2_SCAN;
  constexpr bool WAVE = MODE == K2_FP_WAVE;       // one candidate per wave, cooperative look-ahead
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tables t;
  unsigned char* wave_base = tables_load(t, smem, args.model, args.ext, args.n_ext, args.exists);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t NX = args.n_exp;
  const uint64_t* expected = args.expected;
  const uint16_t* exp_fp = nullptr;
  const uint64_t* fp_hash = nullptr;
  uint8_t* cnt = nullptr;                 // FP: this lane's counters, cnt[id * cnt_stride]
  size_t cnt_stride = 64;
  LaneMem mem;
  if (FP) {
    // stage the lowered trace, the word ids and the word hash once per workgroup
    uint64_t* s_exp = reinterpret_cast<uint64_t*>(wave_base);
    uint16_t* s_fp = reinterpret_cast<uint16_t*>(s_exp + NX);
    uint64_t* s_hash = reinterpret_cast<uint64_t*>(wave_base + (((size_t)NX * 10 + 15) & ~(size_t)15));
    for (uint32_t i = threadIdx.x; i < NX; i += blockDim.x) { s_exp[i] = args.expected[i]; s_fp[i] = args.exp_fp[i]; }
    for (uint32_t i = threadIdx.x; i <= args.fp_hash_mask; i += blockDim.x) s_hash[i] = args.fp_hash[i];
    __syncthreads();
    expected = s_exp; exp_fp = s_fp; fp_hash = s_hash;
    unsigned char* wb = wave_base + k2_fp_shared_bytes(NX, args.fp_hash_mask + 1) +
                        (size_t)wave * k2_fp_wave_bytes(t.A, args.n_fp, MODE);
    mem.st = reinterpret_cast<uint64_t*>(wb) + lane;
    mem.fxq = reinterpret_cast<word_t*>(wb + (size_t)t.A * 64 * 8) + lane;    // (the counter variants never run a wide table)
    mem.pend = nullptr; mem.pend_aux = nullptr; mem.spill = nullptr; mem.spill_aux = nullptr; mem.spill_stride = 0; mem.spill_lane = 0; mem.hot = 0;
    if (MODE == K2_FP_LDS) cnt = wb + (size_t)t.A * 64 * 8 + (size_t)DEMI_FX_CAP * 64 * 4 + lane;
    else if (WAVE) { cnt = wb + (size_t)t.A * 64 * 8 + (size_t)DEMI_FX_CAP * 64 * 4; cnt_stride = 1; }    // (every lane sees the wave's one candidate)
    else { cnt_stride = (size_t)gridDim.x * blockDim.x; cnt = args.fp_counts + (size_t)blockIdx.x * blockDim.x + threadIdx.x; }
  } else {
    mem = lane_mem_carve(wave_base + (size_t)wave * lane_mem_wave_bytes(t.A, false), t.A, false, lane,
                         args.spill, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
  }
  uint64_t* const st = mem.st;
  const uint32_t A = t.A, NE = t.E, exists = t.exists, PMAX = args.p_max;
  // (every lambda below is force-inlined: one that stays a call keeps the variables it captures - the candidate mask, the
  // cursor, the counters' base - in the stack frame, i.e. in scratch memory, and the walk then waits for HBM at every step)
  // counter of word id `f` of this lane: read; +1 / -1 (K2_FP_HBM: a 32-bit atomic on the dword shared with three other
  // lanes - their bytes are untouched as a count never leaves 0..255 - with no return value, so nothing waits for it)
  // (the HBM read goes to the L2, where the atomics are performed: an L1 line could predate this lane's own atomic)
  auto cnt_get = [&](uint32_t f) __attribute__((always_inline)) -> uint32_t {
    const uint8_t* c = cnt + (size_t)f * cnt_stride;
    if (MODE == K2_FP_HBM) {
      const uintptr_t ad = reinterpret_cast<uintptr_t>(c);
      const uint32_t w32 = __hip_atomic_load(reinterpret_cast<const uint32_t*>(ad & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return (w32 >> (8u * (uint32_t)(ad & 3u))) & 0xFFu;
    }
    return *c;
  };
  auto cnt_add = [&](uint32_t f, bool up) __attribute__((always_inline)) {
    uint8_t* c = cnt + (size_t)f * cnt_stride;
    if (MODE == K2_FP_HBM) {
      const uintptr_t ad = reinterpret_cast<uintptr_t>(c);
      uint32_t* w32 = reinterpret_cast<uint32_t*>(ad & ~(uintptr_t)3);
      const uint32_t one = 1u << (8u * (uint32_t)(ad & 3u));
      if (up) atomicAdd(w32, one); else atomicSub(w32, one);
    } else {
      // (LDS: the same trick - an add on the dword that holds four lanes' counters, without a return value, instead of a byte
      // read-modify-write whose read the lane would have to wait for)
      const uint32_t ad = (uint32_t)reinterpret_cast<uintptr_t>(c);
      uint32_t* w32 = reinterpret_cast<uint32_t*>(c - (ad & 3u));
      const uint32_t one = 1u << (8u * (ad & 3u));
      (void)__hip_atomic_fetch_add(w32, up ? one : (0u - one), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  // the word id of a message produced at run time (FP): a probe or two of the workgroup's hash
  auto fp_of = [&](uint32_t word) __attribute__((always_inline)) -> uint32_t {
    uint32_t i = (word * 0x9E3779B1u) >> 7 & args.fp_hash_mask;
    for (;;) {
      const uint64_t e = fp_hash[i];
      if (e == 0) return K2_FP_NONE;
      if ((uint32_t)e == word) return (uint32_t)(e >> 32) - 1u;
      i = (i + 1) & args.fp_hash_mask;
    }
  };

  // lane 0's 64-bit value on every lane
  auto bcast64 = [&](uint64_t x) __attribute__((always_inline)) -> uint64_t {
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) << 32);
  };

  bool active = false, fresh = false;
  uint64_t sched = 0, hash = 0;
  uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;   // candidate mask
  uint64_t app_rng = 0;                      // Instrumenter().seededRandom, restarted with every replay (DEMI_OP_RND)
  uint32_t idx = 0, cur = 0, n_pend = 0, count = 0, ignored = 0, flags = 0, skip = 0xFFFFFFFFu;
  tmask_t rep = 0;                           // registered repeating timers (bit rcv * 4 + timer index)
  Net net;
  net.inaccessible = 0; net.killed = 0; pairs_clear(net.partitioned);
  uint32_t blocked = 0;     // crashed actors (DEMI_OP_CRASH): an expected delivery to one is not "pending" (STSScheduler.scala:392-402)
  // EventTrace.filterKnownAbsentInternals (EventTrace.scala:458-534) on the fly.  actorToAlive is exists & ~inaccessible (a
  // kept SpawnEvent sets it, a kept KillEvent clears it, default false; deadLetters always alive).  actorsToPartitioned as the
  // reference wrote it - PartitionEvent((a,b)) -> false, UnPartitionEvent((a,b)) -> true under the ordered key (:523-528) - is
  // fk_part (LITERAL); CORRECTED reads net.partitioned in both directions.  prunedMessageSends is a mask over the slots the
  // lowering gave the in-flight internal messages.
  const uint32_t FK = args.filter_absents;
  PairSet fk_part;
  pairs_clear(fk_part);
  uint64_t fk_pruned0 = 0, fk_pruned1 = 0;
  auto fk_cut = [&](uint32_t s_, uint32_t r_) __attribute__((always_inline)) -> bool {
    if (s_ >= MAX_ACT || r_ >= MAX_ACT) return false;
#ifdef DEMI_BIG
    if (FK == DEMI_FILTER_ABSENTS_LITERAL) return pairs_get(fk_part, s_, r_);
    return pairs_get(net.partitioned, s_, r_) | pairs_get(net.partitioned, r_, s_);
#else
    if (FK == DEMI_FILTER_ABSENTS_LITERAL) return (fk_part.w >> (s_ * 8 + r_)) & 1ull;
    return ((net.partitioned.w >> (s_ * 8 + r_)) | (net.partitioned.w >> (r_ * 8 + s_))) & 1ull;
#endif
  };
  auto fk_alive = [&](uint32_t who) __attribute__((always_inline)) -> bool { return who >= MAX_ACT || (((exists & ~net.inaccessible) >> who) & 1u); };
  uint64_t tq = 0;
  uint32_t n_tq = 0;
  // K2_FP_WAVE: the look-ahead window (lane i: expected event win_base + i): class 0 network event / 1 actor's MsgSend (filter
  // lowering) / 2 external MsgSend / 3 MsgEvent / 4 none; the candidate-static test; the word id (class 0: the cursor-head word
  // it must meet); a | b << 8 | slot << 16
  uint32_t win_base = 0xFFFFFF00u, w_kind = 4u, w_fp = 0, w_ab = 0;
  bool w_static = false;
  uint64_t b_next = 0, b_end = 0;
  bool exhausted = false;

// bit I of the candidate mask.  Selected by VALUE: written as a choice between the four variables, the compiler turns it into a
// table of their addresses, which pins the mask, the cursor and the kernel arguments in the stack frame (scratch memory) - every
// step of the walk then waits for two memory round trips (round 3: 240 bytes of scratch per lane, 2 500 cycles per event)
#define IN_MASK(I) ((uint32_t)((((((I) >> 6) & 3u) == 0u ? m0 : 0ull) | ((((I) >> 6) & 3u) == 1u ? m1 : 0ull) | \
                                ((((I) >> 6) & 3u) == 2u ? m2 : 0ull) | ((((I) >> 6) & 3u) == 3u ? m3 : 0ull)) >> ((I) & 63u)) & 1u)
#define TIMER_BIT(RCV, TYPE) ((tmask_t)1 << ((RCV) * DEMI_MAX_TIMER_TYPES + (t.meta[(TYPE)] >> 8)))
// the message word of expected event E at index I, from SRC to DST (a wide table's payload area comes from exp_area)
#ifdef DEMI_WIDE
#define EXP_WORD(E, I, SRC, DST) msg_word_area((uint32_t)((E) >> 24) & 0xFF, (SRC), (DST), args.exp_area[(I)])
#else
#define EXP_WORD(E, I, SRC, DST) msg_word((uint32_t)((E) >> 24) & 0xFF, (SRC), (DST), (uint32_t)((E) >> 32) & 0xFFu, (uint32_t)((E) >> 40) & 0xFFu)
#endif
// FP: FPID is the word's id if the caller knows it, else it is looked up; a word without an id only takes up capacity
#define PEND_APPEND_ID(WORD, FPID)                                     \
  do {                                                                 \
    if (n_pend >= PMAX) { flags |= DEMI_V_PENDING_OVF; }               \
    else if (FP) {                                                     \
      const uint32_t id_ = (FPID);                                     \
      if (id_ != K2_FP_NONE) cnt_add(id_, true);                       \
      n_pend++;                                                        \
    } else { pend_store(mem, n_pend, (WORD)); n_pend++; }              \
  } while (0)
#define PEND_APPEND(WORD) PEND_APPEND_ID(WORD, FP ? fp_of((uint32_t)(WORD)) : K2_FP_NONE)

  // cursor over the candidate's non-Send, non-WaitQuiescence externals (subsequenceIntersection :299-304)
  auto cur_skip = [&]() __attribute__((always_inline)) {
    while (cur < NE) {
      const uint32_t kind = (uint32_t)t.trace[cur] & 0xFF;
      if (IN_MASK(cur) && kind != DEMI_EV_SEND && kind != DEMI_EV_WAIT_QUIESCENCE) break;
      cur++;
    }
  };
  // STSScheduler.enqueue_timer = handle_timer: straight into messagesToSend (no parking)
  auto handle_timer = [&](uint32_t rcv, uint32_t type) __attribute__((always_inline)) {
    if (n_tq >= DEMI_TQ_CAP) { flags |= DEMI_V_QUEUE_OVF; return; }
    tq |= (uint64_t)tq_pack(rcv, type, BIG_TU ? (t.meta[type] >> 8) : 0u) << (8 * n_tq);
    n_tq++;
  };

  // ------------------------------------------------------------ lock-step walk
  // Every lane of the wave looks at the SAME expected event in every step.  With a cursor per lane (the loop further
  // down) a lane whose candidate lost an ancestor of the next recorded deliveries skips a run of events while the lanes
  // that found theirs wait: 3.6 of 64 lanes active per VALU instruction on 2^20 random candidates, 1.65 M wave
  // instructions per 64 candidates (profiles/r02_ddmin_counters.txt).  In lock step the event is decoded once per wave
  // (scalar), the cheap "does it apply to my candidate" test runs full-width, and when some lane delivers, all of them
  // deliver the same message to the same receiver: one handler per step instead of up to eight.  A wave takes
  // lanes_per_wave candidates, walks the trace once, writes their verdicts and claims the next batch.  Per lane the
  // sequence of operations is exactly the one of the per-lane loop (same verdicts, kept marks and flags; the GPU suite runs
  // every K2 test in all three counter modes on it).  Measured on the ddmin record: 2^20 candidates 61.5 -> 10.8 ms,
  // 65 536: 7.2 -> 1.8 ms, 4 096: 1.33 -> 1.20 ms, 256: 0.81 -> 0.72 ms.
#ifdef DEMI_K2_PHASES
  uint64_t ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_ev = 0;
#define K2_NOW(V) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V) : : "memory")
#define K2_MARK(I) do { uint64_t now_; K2_NOW(now_); ph_t[I] += now_ - ph_last; ph_last = now_; } while (0)
  uint64_t ph_last; K2_NOW(ph_last);
#else
#define K2_MARK(I) do {} while (0)
#endif
  if (args.lockstep) {
    for (;;) {
      uint64_t base = 0;
      if (lane == 0) base = atomicAdd(args.work_counter, (unsigned long long)args.lanes_per_wave);
      base = __shfl(base, 0);
      if (base >= args.n) break;
      sched = base + lane;
      const bool mine = lane < args.lanes_per_wave && sched < args.n;
      active = mine;                                   // still replaying (no capacity abort so far)
      if (mine) {
        if (args.masks) {
          const uint64_t* mk = args.masks + sched * 4;
          m0 = mk[0]; m1 = mk[1]; m2 = mk[2]; m3 = mk[3];
        } else {
          m0 = m1 = m2 = m3 = ~0ull;
        }
        skip = args.skip ? args.skip[sched] : 0xFFFFFFFFu;
        hash = 0xCBF29CE484222325ULL;
        app_rng = jr_seed(0);
        net.inaccessible = exists; net.killed = 0; pairs_clear(net.partitioned);
        for (uint32_t a = 0; a < A * ST_WORDS; a++) st[a * 64] = t.init[a];
        if (FP && !WAVE) for (uint32_t f = 0; f < args.n_fp; f++) cnt[(size_t)f * cnt_stride] = 0;
        cur = 0; n_pend = 0; count = 0; ignored = 0; flags = 0; rep = 0; tq = 0; n_tq = 0; blocked = 0;
        pairs_clear(fk_part); fk_pruned0 = 0; fk_pruned1 = 0;
        cur_skip();
      }
      if (WAVE) {
        // the wave's one candidate (lane 0's): its counters zeroed by all lanes, its mask known to all of them
        for (uint32_t f = lane; f < args.n_fp; f += 64) cnt[f] = 0;
        m0 = bcast64(m0); m1 = bcast64(m1); m2 = bcast64(m2); m3 = bcast64(m3);
        win_base = 0xFFFFFF00u;                 // (no window yet: its static part depends on the candidate's mask)
        __builtin_amdgcn_wave_barrier();       // (the zeroed counters before lane 0's first increment: LDS operations of a wave complete in order)
      }
      K2_MARK(0);
      // the next event and its word id are requested one step ahead: with one wave per SIMD there is nothing else to hide
      // the LDS round trip behind, and the walk is a chain of them
      uint64_t ev_next = NX ? expected[0] : 0ull;
      uint32_t fp_next = (FP && NX) ? (uint32_t)exp_fp[0] : 0u;
      for (idx = 1; idx <= NX; idx++) {                // (idx - 1 is the event's index, as in the per-lane loop)
        K2_MARK(1);
#ifdef DEMI_K2_PHASES
        ph_ev++;
#endif
        if (__ballot(active) == 0) break;
        if (WAVE) {
          // ---- cooperative look-ahead over a WINDOW of 64 expected events (lane i holds event win_base + i).  What does not
          // depend on the replay's state is decoded once per window: the event's class, whether the candidate's mask / removed
          // delivery keeps it in the projected trace, the word id of its message.  What does depend on it - is the message
          // pending, is the receiver blocked, does a network event equal the cursor head - is re-evaluated after every event that
          // acted: one counter read and a handful of instructions per lane.  An event ACTS when the step below would change the
          // state for it; a MsgEvent of the projected trace whose message is not pending (or whose receiver is blocked) is
          // IGNORED (:528-529): it only counts.  Nothing before the first acting event changes anything the tests read, so all
          // 64 tests are the ones the sequential walk would have made.
          const uint32_t pos = idx - 1;
          if (pos >= win_base + 64u || pos < win_base) {
            win_base = pos;
            const uint32_t j = win_base + lane;
            const uint32_t skip0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)skip);
            w_kind = 4u; w_static = false; w_fp = 0; w_ab = 0;
            if (j < NX) {
              const uint64_t e_ = expected[j];
              const uint32_t kind_ = (uint32_t)e_ & 0xFF, a_ = (uint32_t)(e_ >> 8) & 0xFF, b_ = (uint32_t)(e_ >> 16) & 0xFF;
              const uint32_t ext_ = (uint32_t)(e_ >> 48) & 0xFF;
              w_ab = a_ | (b_ << 8) | ((uint32_t)(e_ >> 56) << 16);
              if (kind_ <= DEMI_REC_UNPARTITION) {
                // (the external-event kind this record must meet at the cursor head, with its actors: compared as one word)
                const uint32_t want_kind = (kind_ == DEMI_REC_SPAWN) ? DEMI_EV_START : (kind_ == DEMI_REC_KILL) ? DEMI_EV_KILL
                                         : (kind_ == DEMI_REC_PARTITION) ? DEMI_EV_PARTITION : DEMI_EV_UNPARTITION;
                w_kind = 0u; w_fp = want_kind | (a_ << 8) | ((kind_ >= DEMI_REC_PARTITION ? b_ : 0u) << 16);
              } else if (kind_ == DEMI_REC_MSG_SEND && ext_ == 255) {
                w_kind = 1u;
              } else if (kind_ == DEMI_REC_MSG_SEND) {
                w_kind = 2u; w_static = IN_MASK(ext_) && ((exists >> b_) & 1);
              } else {
                w_kind = 3u; w_static = j != skip0 && (ext_ == 255 || IN_MASK(ext_)); w_fp = (uint32_t)exp_fp[j];
              }
            }
          }
          const uint32_t cur0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur);
          const uint32_t blocked0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)blocked);
          // (the cursor head as the word a network event of the window carries: kind | a << 8 | b << 16, b only for partitions)
          uint32_t head = 0xFFFFFFFFu;
          if (cur0 < NE) {
            const uint64_t x = t.trace[cur0];
            const uint32_t xk = (uint32_t)x & 0xFF;
            head = xk | (((uint32_t)(x >> 8) & 0xFF) << 8) | ((xk >= DEMI_EV_PARTITION ? (uint32_t)(x >> 16) & 0xFF : 0u) << 16);
          }
          bool in_trace = w_static;
          if (FK) {       // (the filter's state, broadcast where control flow is wave-uniform: FK is a kernel argument)
            const uint32_t inacc0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)net.inaccessible);
            const uint64_t pr0 = bcast64(fk_pruned0), pr1 = bcast64(fk_pruned1);
            const uint32_t a_ = w_ab & 0xFF, b_ = (w_ab >> 8) & 0xFF, slot = w_ab >> 16;
            const bool sent = slot == 255u || !(((slot & 64u) ? pr1 : pr0) >> (slot & 63u) & 1ull);
            const bool alive = b_ >= MAX_ACT || (((exists & ~inacc0) >> b_) & 1u);
            bool cut = false;
#ifdef DEMI_BIG     // (compiled, never launched: a big table is a wide table and replays with the scanning kernel)
            cut = fk_cut(a_, b_);
#else
            const uint64_t partn0 = bcast64(net.partitioned.w), part0 = bcast64(fk_part.w);
            if (a_ < DEMI_MAX_ACTORS && b_ < DEMI_MAX_ACTORS)
              cut = FK == DEMI_FILTER_ABSENTS_LITERAL ? ((part0 >> (a_ * 8 + b_)) & 1ull)
                                                      : (((partn0 >> (a_ * 8 + b_)) | (partn0 >> (b_ * 8 + a_))) & 1ull);
#endif
            in_trace = in_trace && alive && !cut && sent;
          }
          const bool is_ev = w_kind == 3u;
          const bool pending = is_ev && in_trace && !((blocked0 >> ((w_ab >> 8) & 0xFF)) & 1u) && cnt[is_ev ? w_fp : 0u] != 0;
          const bool acts = (w_kind == 0u && w_fp == head) || (w_kind == 1u && FK != 0) || (w_kind == 2u && w_static) || pending;
          const bool ign = is_ev && in_trace && !pending;
          const uint32_t off = pos - win_base;                       // the window's events before `pos` are behind the walk
          const uint64_t ahead = ~0ull << off;
          const uint64_t acting = __ballot(acts) & ahead, ignoring = __ballot(ign) & ahead;
          if (acting == 0) {                                         // nothing in the rest of the window acts
            if (lane == 0 && active) ignored += (uint32_t)__popcll(ignoring);
            idx = win_base + 64u;                                    // (the loop's idx++ makes the next position win_base + 64)
            continue;
          }
          const uint32_t first = (uint32_t)__builtin_ctzll(acting);
          if (lane == 0 && active) ignored += (uint32_t)__popcll(ignoring & ((1ull << first) - 1ull));
          idx = win_base + first + 1u;
        }
        const uint64_t ev = WAVE ? expected[idx - 1] : ev_next;
        const uint32_t fp_cur = WAVE ? (uint32_t)exp_fp[idx - 1] : fp_next;
        if (!WAVE && idx < NX) { ev_next = expected[idx]; if (FP) fp_next = (uint32_t)exp_fp[idx]; }
        // the event is the same for every lane: keep it in scalar registers, so that its kind, the handler it selects
        // and the receiver are wave-uniform for the compiler too
        const uint64_t e = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ev) |
                           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ev >> 32)) << 32);
        const uint32_t kind = (uint32_t)e & 0xFF, a = (uint32_t)(e >> 8) & 0xFF, b = (uint32_t)(e >> 16) & 0xFF;
        const uint32_t ext = (uint32_t)(e >> 48) & 0xFF;
        if (kind <= DEMI_REC_UNPARTITION) {
          if (active && cur < NE) {
            const uint64_t x = t.trace[cur];
            const uint32_t xk = (uint32_t)x & 0xFF, xa = (uint32_t)(x >> 8) & 0xFF, xb = (uint32_t)(x >> 16) & 0xFF;
            const bool two = kind >= DEMI_REC_PARTITION;
            const uint32_t want_kind = (kind == DEMI_REC_SPAWN) ? DEMI_EV_START : (kind == DEMI_REC_KILL) ? DEMI_EV_KILL
                                     : (kind == DEMI_REC_PARTITION) ? DEMI_EV_PARTITION : DEMI_EV_UNPARTITION;
            if (xk == want_kind && xa == a && (!two || xb == b)) {
              cur++;
              cur_skip();
              if (args.kept) args.kept[sched * NX + idx - 1] = 1;
              if (kind == DEMI_REC_SPAWN) { net.inaccessible &= ~(1u << a); net.killed &= ~(1u << a); blocked &= ~(1u << a); }
              else if (kind == DEMI_REC_KILL) { net.killed |= 1u << a; net.inaccessible |= 1u << a; }
              else if (kind == DEMI_REC_PARTITION) { pairs_put(net.partitioned, a, b, true); pairs_put(fk_part, a, b, false); }
              else { pairs_put(net.partitioned, a, b, false); pairs_put(fk_part, a, b, true); }
            }
          }
          continue;
        }
        if (kind == DEMI_REC_MSG_SEND && ext == 255) {
          if (active) {
            const uint32_t slot = (uint32_t)(e >> 56);
            const bool pruned = FK && !(fk_alive(a) && !fk_cut(a, b));
            const uint64_t bit = 1ull << (slot & 63u);
            if (slot & 64u) fk_pruned1 = pruned ? (fk_pruned1 | bit) : (fk_pruned1 & ~bit);
            else fk_pruned0 = pruned ? (fk_pruned0 | bit) : (fk_pruned0 & ~bit);
          }
          continue;
        }
        if (kind == DEMI_REC_MSG_SEND) {
          if (active && IN_MASK(ext) && ((exists >> b) & 1)) {
            PEND_APPEND_ID(EXP_WORD(e, idx - 1, DL, b),
                           FP ? fp_cur : K2_FP_NONE);
            if (args.kept && !(flags & DEMI_OVF_ANY)) args.kept[sched * NX + idx - 1] = 1;
            if (flags & DEMI_OVF_ANY) active = false;
          }
          continue;
        }
        // ---- MSG_EVENT: which lanes deliver it
        const word_t want = EXP_WORD(e, idx - 1, a, b);
        bool deliver = false;
        if (active) {
          do {
            if (idx - 1 == skip) break;                  // the delivery this candidate removes
            if (ext != 255 && !IN_MASK(ext)) break;      // pruned together with its Send (filterSends)
            if (FK) {
              const uint32_t slot = (uint32_t)(e >> 56);
              const bool sent = slot == 255u || !(((slot & 64u) ? fk_pruned1 : fk_pruned0) >> (slot & 63u) & 1ull);
              if (!(fk_alive(b) && !fk_cut(a, b) && sent)) break;
            }
            if ((blocked >> b) & 1u) { ignored++; break; }
            if (FP) {
              const uint32_t f = fp_cur;
              if (cnt_get(f) == 0) { ignored++; break; }       // "Ignoring message" (:528-529)
              cnt_add(f, false);
            } else {
              uint32_t k = 0;
              for (; k < n_pend; k++)
                if (pend_load(mem, k) == want) break;
              if (k == n_pend) { ignored++; break; }
              pend_store(mem, k, pend_load(mem, n_pend - 1));
            }
            n_pend--;
            if (args.kept) args.kept[sched * NX + idx - 1] = 1;
            deliver = true;
          } while (0);
        }
        K2_MARK(2);
        if (__ballot(deliver) == 0) continue;            // nobody's candidate has this message pending
        const word_t w = want;
        const uint32_t type = w_type(w), me = w_dst(w);
        if (deliver) {
          count++;
          hash_step(hash, w);
          const uint32_t meta = t.meta[type];
          if (((meta & 0xFF) == DEMI_MSG_TIMER) && (rep & ((tmask_t)1 << (me * DEMI_MAX_TIMER_TYPES + (meta >> 8)))))
            handle_timer(me, type);
          if (flags & DEMI_OVF_ANY) { deliver = false; active = false; }
        }
        K2_MARK(3);
        uint32_t nfx = 0;
        if (deliver) nfx = DEMI_VM_RUN(t, mem, w, flags, app_rng);
        K2_MARK(4);
        if (deliver) {
          for (uint32_t k = 0; k < nfx && !(flags & DEMI_OVF_ANY); k++) {
            const word_t fxw = mem.fxq[k * 64];
            const uint32_t fx = (uint32_t)fxw;
            const uint32_t op = fx & 31u, ftype = (fx >> 5) & 31u, target = fx_target(fx);
            if (op <= DEMI_OP_BCAST) {
              const bool bc = (op == DEMI_OP_BCAST);
              const uint32_t first = bc ? 0u : target, last = bc ? A : (target < A ? target + 1 : 0u);
              for (uint32_t r = first; r < last; r++) {
                if ((bc && r == me) || !((exists >> r) & 1)) continue;
                if (!crosses_partition(net, me, r)) PEND_APPEND(fx_msg_word(fxw, ftype, me, r));
              }
            } else if (op == DEMI_OP_CRASH) {
              blocked |= 1u << me;
            } else if (op == DEMI_OP_TCANCEL) {
              rep &= ~TIMER_BIT(me, ftype);
              const uint32_t wantt = tq_pack(me, ftype, BIG_TU ? (t.meta[ftype] >> 8) : 0u);
              bool found = false;
              for (uint32_t q = 0; q < n_tq; q++) {
                if (((uint32_t)(tq >> (8 * q)) & 0xFF) == wantt) {
                  const uint64_t lowm = (q == 0) ? 0ull : (~0ull >> (64 - 8 * q));
                  tq = (tq & lowm) | ((tq >> 8) & ~lowm);
                  n_tq--; found = true; break;
                }
              }
              if (!found) {
                const word_t wantw = msg_word(ftype, DL, me, 0, 0);
                if (FP) {
                  const uint32_t f = fp_of((uint32_t)wantw);
                  if (cnt_get(f)) { cnt_add(f, false); n_pend--; }
                } else {
                  for (uint32_t q = 0; q < n_pend; q++) {
                    if (pend_load(mem, q) == wantw) {
                      pend_store(mem, q, pend_load(mem, n_pend - 1));
                      n_pend--; break;
                    }
                  }
                }
              }
            } else {
              const tmask_t bit = TIMER_BIT(me, ftype);
              if (!(rep & bit)) {
                if (op == DEMI_OP_TREP) rep |= bit;
                handle_timer(me, ftype);
              }
            }
          }
          for (uint32_t k = 0; k < n_tq && !(flags & DEMI_OVF_ANY); k++) {
            const uint32_t bt = (uint32_t)(tq >> (8 * k)) & 0xFF, rcv = tq_rcv(bt), ttype = tq_type(bt);
            if (!((net.inaccessible >> rcv) & 1)) PEND_APPEND(msg_word(ttype, DL, rcv, 0, 0));
          }
          tq = 0; n_tq = 0;
          if (flags & DEMI_OVF_ANY) active = false;
        }
        K2_MARK(5);
      }
      if (mine) {
        uint32_t viol = 0;
        if (!(flags & DEMI_OVF_ANY)) {
          const uint32_t fp = invariant_code(t, st, exists, A, DEMI_INV_KIND_OF(t), t.inv_fa, t.inv_va, t.inv_fb);
          if (fp && (((fp ^ args.looking_for) & t.fp_mask) == 0)) viol = args.looking_for;
        }
        for (uint32_t a = 0; a < A * ST_WORDS; a++) hash_step(hash, st[a * 64]);
        uint4 v;
        if (flags & DEMI_OVF_ANY) {
          v.x = flags & DEMI_OVF_ANY; v.y = 0; v.z = 0; v.w = 0;
        } else {
          v.x = (viol ? DEMI_V_VIOLATION : 0u) | (ignored ? DEMI_V_DIVERGED : 0u) | ((count < 0xFFFFu ? count : 0xFFFFu) << 16);
          v.y = viol; v.z = (uint32_t)hash; v.w = (uint32_t)(hash >> 32);
        }
        *reinterpret_cast<uint4*>(&args.out[sched]) = v;
      }
      K2_MARK(6);
    }
#ifdef DEMI_K2_PHASES
    if (lane == 0 && args.phase_out) {
      unsigned long long* o = args.phase_out + ((size_t)blockIdx.x * K2_WAVES + wave) * 16;
      for (int i = 0; i < 7; i++) o[i] = ph_t[i];
      o[15] = ph_ev;
    }
#endif
    return;
  }

  for (;;) {
    // ---------------------------------------------------------- refill (same protocol as K1)
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

    word_t w = 0;
    bool deliver = false, finish = false;
    if (active) {
      if (fresh) {
        fresh = false;
        if (args.masks) {
          const uint64_t* mk = args.masks + sched * 4;
          m0 = mk[0]; m1 = mk[1]; m2 = mk[2]; m3 = mk[3];
        } else {
          m0 = m1 = m2 = m3 = ~0ull;
        }
        skip = args.skip ? args.skip[sched] : 0xFFFFFFFFu;
        hash = 0xCBF29CE484222325ULL;
        app_rng = jr_seed(0);
        net.inaccessible = exists; net.killed = 0; pairs_clear(net.partitioned);
        for (uint32_t a = 0; a < A * ST_WORDS; a++) st[a * 64] = t.init[a];
        if (FP) for (uint32_t f = 0; f < args.n_fp; f++) cnt[(size_t)f * cnt_stride] = 0;
        idx = 0; cur = 0; n_pend = 0; count = 0; ignored = 0; flags = 0; rep = 0; tq = 0; n_tq = 0; blocked = 0;
        pairs_clear(fk_part); fk_pruned0 = 0; fk_pruned1 = 0;
        cur_skip();
      }
      // -------------------------------------------------------- advanceReplay (:405-559)
      while (idx < NX && !(flags & DEMI_OVF_ANY)) {
        const uint64_t e = expected[idx];
        idx++;
        const uint32_t kind = (uint32_t)e & 0xFF, a = (uint32_t)(e >> 8) & 0xFF, b = (uint32_t)(e >> 16) & 0xFF;
        const uint32_t ext = (uint32_t)(e >> 48) & 0xFF;
        if (kind <= DEMI_REC_UNPARTITION) {
          // kept iff it equals the cursor head by name; dropped once the cursor is exhausted
          if (cur >= NE) continue;
          const uint64_t x = t.trace[cur];
          const uint32_t xk = (uint32_t)x & 0xFF, xa = (uint32_t)(x >> 8) & 0xFF, xb = (uint32_t)(x >> 16) & 0xFF;
          const bool two = kind >= DEMI_REC_PARTITION;
          // demi_rec_kind SPAWN,KILL,PARTITION,UNPARTITION <-> demi_ext_kind START,KILL,PARTITION,UNPARTITION
          const uint32_t want_kind = (kind == DEMI_REC_SPAWN) ? DEMI_EV_START : (kind == DEMI_REC_KILL) ? DEMI_EV_KILL
                                   : (kind == DEMI_REC_PARTITION) ? DEMI_EV_PARTITION : DEMI_EV_UNPARTITION;
          if (xk != want_kind || xa != a || (two && xb != b)) continue;
          cur++;
          cur_skip();
          if (args.kept) args.kept[sched * NX + idx - 1] = 1;
          if (kind == DEMI_REC_SPAWN) { net.inaccessible &= ~(1u << a); net.killed &= ~(1u << a); blocked &= ~(1u << a); }
          else if (kind == DEMI_REC_KILL) { net.killed |= 1u << a; net.inaccessible |= 1u << a; }
          else if (kind == DEMI_REC_PARTITION) { pairs_put(net.partitioned, a, b, true); pairs_put(fk_part, a, b, false); }
          else { pairs_put(net.partitioned, a, b, false); pairs_put(fk_part, a, b, true); }
        } else if (kind == DEMI_REC_MSG_SEND && ext == 255) {
          // an actor's MsgSend (only lowered for the filter): `if (messageSendable(snd, rcv)) result += event else
          // prunedMessageSends += id`; the slot is reused, so a sendable one clears the bit
          const uint32_t slot = (uint32_t)(e >> 56);
          const bool pruned = FK && !(fk_alive(a) && !fk_cut(a, b));
          const uint64_t bit = 1ull << (slot & 63u);
          if (slot & 64u) fk_pruned1 = pruned ? (fk_pruned1 | bit) : (fk_pruned1 & ~bit);
          else fk_pruned0 = pruned ? (fk_pruned0 | bit) : (fk_pruned0 & ~bit);
        } else if (kind == DEMI_REC_MSG_SEND) {
          // external MsgSend -> enqueue_message (:509-511) unless its Send was pruned
          if (IN_MASK(ext) && ((exists >> b) & 1)) {
            PEND_APPEND_ID(EXP_WORD(e, idx - 1, DL, b),
                           FP ? (uint32_t)exp_fp[idx - 1] : K2_FP_NONE);
            if (args.kept && !(flags & DEMI_OVF_ANY)) args.kept[sched * NX + idx - 1] = 1;
          }
        } else {  // MSG_EVENT
          if (idx - 1 == skip) continue;               // the delivery this candidate removes (OneAtATimeRemoval.scala:57-124)
          if (ext != 255 && !IN_MASK(ext)) continue;   // pruned together with its Send (filterSends)
          if (FK) {                                    // messageDeliverable(snd, rcv, id), else not part of the projected trace
            const uint32_t slot = (uint32_t)(e >> 56);
            const bool sent = slot == 255u || !(((slot & 64u) ? fk_pruned1 : fk_pruned0) >> (slot & 63u) & 1ull);
            if (!(fk_alive(b) && !fk_cut(a, b) && sent)) continue;
          }
          if ((blocked >> b) & 1u) { ignored++; continue; }   // the destination is blocked: not deliverable (:392-402), ignored
          const word_t want = EXP_WORD(e, idx - 1, a, b);
          if (FP) {
            const uint32_t f = exp_fp[idx - 1];
            if (cnt_get(f) == 0) { ignored++; continue; }     // "Ignoring message" (:528-529)
            cnt_add(f, false);
          } else {
            uint32_t k = 0;
            for (; k < n_pend; k++)
              if (pend_load(mem, k) == want) break;
            if (k == n_pend) { ignored++; continue; }     // "Ignoring message" (:528-529)
            pend_store(mem, k, pend_load(mem, n_pend - 1));
          }
          n_pend--;
          if (args.kept) args.kept[sched * NX + idx - 1] = 1;
          w = want;
          deliver = true;
          break;
        }
      }
      if (!deliver) finish = true;
      if (deliver) {
        count++;
        hash_step(hash, w);
        // Instrumenter retrigger of a repeating timer (Instrumenter.scala:1008-1016)
        const uint32_t type = w_type(w), me = w_dst(w);
        const uint32_t meta = t.meta[type];
        if (((meta & 0xFF) == DEMI_MSG_TIMER) && (rep & ((tmask_t)1 << (me * DEMI_MAX_TIMER_TYPES + (meta >> 8)))))
          handle_timer(me, type);
        if (flags & DEMI_OVF_ANY) { deliver = false; finish = true; }
      }
    }

    uint32_t nfx = 0;
    if (deliver) nfx = DEMI_VM_RUN(t, mem, w, flags, app_rng);

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
            if ((bc && r == me) || !((exists >> r) & 1)) continue;
            if (!crosses_partition(net, me, r)) PEND_APPEND(fx_msg_word(fxw, type, me, r));
          }
        } else if (op == DEMI_OP_CRASH) {
          blocked |= 1u << me;                 // actorCrashed (Instrumenter.scala:184-199)
        } else if (op == DEMI_OP_TCANCEL) {
          // notify_timer_cancel (:828-855): messagesToSend first, then the (deadLetters, rcv) queue
          rep &= ~TIMER_BIT(me, type);
          const uint32_t want = tq_pack(me, type, BIG_TU ? (t.meta[type] >> 8) : 0u);
          bool found = false;
          for (uint32_t q = 0; q < n_tq; q++) {
            if (((uint32_t)(tq >> (8 * q)) & 0xFF) == want) {
              const uint64_t lowm = (q == 0) ? 0ull : (~0ull >> (64 - 8 * q));
              tq = (tq & lowm) | ((tq >> 8) & ~lowm);
              n_tq--; found = true; break;
            }
          }
          if (!found) {
            const word_t wantw = msg_word(type, DL, me, 0, 0);
            if (FP) {
              const uint32_t f = fp_of((uint32_t)wantw);         // every timer word has an id
              if (cnt_get(f)) { cnt_add(f, false); n_pend--; }
            } else {
              for (uint32_t q = 0; q < n_pend; q++) {
                if (pend_load(mem, q) == wantw) {
                  pend_store(mem, q, pend_load(mem, n_pend - 1));
                  n_pend--; break;
                }
              }
            }
          }
        } else {
          const tmask_t bit = TIMER_BIT(me, type);
          if (!(rep & bit)) {
            if (op == DEMI_OP_TREP) rep |= bit;
            handle_timer(me, type);
          }
        }
      }
      // schedule_new_message starts with send_external_messages (:655): timers become pending now,
      // unless the receiver is inaccessible (crosses_partition(deadLetters, rcv))
      for (uint32_t k = 0; k < n_tq && !(flags & DEMI_OVF_ANY); k++) {
        const uint32_t bt = (uint32_t)(tq >> (8 * k)) & 0xFF, rcv = tq_rcv(bt), type = tq_type(bt);
        if (!((net.inaccessible >> rcv) & 1)) PEND_APPEND(msg_word(type, DL, rcv, 0, 0));
      }
      tq = 0; n_tq = 0;
      if (flags & DEMI_OVF_ANY) finish = true;
    }

    if (active && finish) {
      // the invariant on the final state; verdict = fingerprint.matches(target) (:278-300)
      uint32_t viol = 0;
      if (!(flags & DEMI_OVF_ANY)) {
        const uint32_t fp = invariant_code(t, st, exists, A, DEMI_INV_KIND_OF(t), t.inv_fa, t.inv_va, t.inv_fb);
        if (fp && (((fp ^ args.looking_for) & t.fp_mask) == 0)) viol = args.looking_for;
      }
      for (uint32_t a = 0; a < A * ST_WORDS; a++) hash_step(hash, st[a * 64]);
      uint4 v;
      if (flags & DEMI_OVF_ANY) {
        v.x = flags & DEMI_OVF_ANY; v.y = 0; v.z = 0; v.w = 0;
      } else {
        v.x = (viol ? DEMI_V_VIOLATION : 0u) | (ignored ? DEMI_V_DIVERGED : 0u) | ((count < 0xFFFFu ? count : 0xFFFFu) << 16);
        v.y = viol; v.z = (uint32_t)hash; v.w = (uint32_t)(hash >> 32);
      }
      *reinterpret_cast<uint4*>(&args.out[sched]) = v;
      active = false;
    }
  }
#undef IN_MASK
#undef EXP_WORD
#undef TIMER_BIT
#undef PEND_APPEND
#undef PEND_APPEND_ID
}

__global__ __launch_bounds__(K2_WAVES * 64) void k2_replay(const K2Args args) { k2_replay_body<K2_SCAN>(args); }
__global__ __launch_bounds__(K2_WAVES * 64) void k2_replay_fp(const K2Args args) { k2_replay_body<K2_FP_LDS>(args); }
__global__ __launch_bounds__(K2_WAVES * 64) void k2_replay_fp_hbm(const K2Args args) { k2_replay_body<K2_FP_HBM>(args); }
__global__ __launch_bounds__(K2_WAVES * 64) void k2_replay_fp_wave(const K2Args args) { k2_replay_body<K2_FP_WAVE>(args); }

}  // namespace demi
