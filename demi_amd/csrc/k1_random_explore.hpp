// k1_random_explore.hpp — K1: one RandomScheduler execution per wavefront lane.
//
// Restates, per lane, RandomScheduler.explore / schedule_new_message / event_produced
// (schedulers/RandomScheduler.scala:234-272, 352-485, 274-321), FullyRandom + RandomizedHashSet
// (RandomScheduler.scala:635-697, schedulers/Util.scala:110-185), the ExternalEventInjector /
// EventOrchestrator trace driver (schedulers/ExternalEventInjector.scala:306-365, 382-441, 541-580;
// schedulers/EventOrchestrator.scala:132-189) and the Instrumenter's timer bookkeeping
// (Instrumenter.scala:159-168, 1008-1016, 1145-1200) over a table-encoded transition function.
//
// Execution shape: each loop iteration every live lane performs one scheduling step (guards,
// flush, random pick, swap-remove), runs the picked message's handler rows (branch-free
// interpreter, sim_core.hpp) and applies the recorded effects.  A lane whose execution ends writes
// its 16-byte verdict and immediately refills with the next schedule index (wave ballot + prefix
// count; 64-index batches are claimed from one global counter), so variable execution lengths do
// not idle lanes until the very tail of the launch.
#pragma once

#include "sim_core.hpp"

// The handler-row engine: the table interpreter, or the native code generated from the loaded table when the
// kernel is compiled by demi_model_specialize (jit.hpp defines DEMI_VM_RUN before including this file).
#ifndef DEMI_VM_RUN
#define DEMI_VM_RUN vm_run
#endif

namespace demi {

struct K1Args {
  const DevModel* model;
  const uint64_t* trace;    // demi_ext_event[n_ev] as 8-byte words
  uint32_t n_ev;
  uint32_t exists;          // actors created by populateActorSystem
  uint64_t seed_base;
  const uint64_t* seeds;    // optional explicit seeds
  uint64_t n;
  uint32_t max_messages, interval, looking_for_valid, looking_for;
  demi_verdict* out;
  unsigned long long* work_counter;  // zeroed before every launch
  uint32_t p_max;           // pending-set capacity of the spec (<= DEMI_MAX_PENDING)
  uint32_t* spill;          // HBM scratch for pending slots >= K1_HOT (and the SrcDstFIFO arrays): k1_spill_words_per_lane
  demi_rec_event* rec_out;  // REC only: [n][rec_cap]
  uint32_t* rec_count;      // REC only: [n]
  uint32_t rec_cap;
  uint32_t n_batches;       // WaitQuiescence events + 1
  unsigned long long* phase_out;   // -DDEMI_K1_PHASES builds only: [waves][16] cycle totals per phase
  uint32_t epi;             // CARRY variants: executions per RandomScheduler instance (demi_limits.executions_per_instance)
  uint32_t rec_shared;      // CARRY + REC: every execution of the instance records into rec_out[0 ..] (the last one stays)
};
// MULTI variants (demi_random_ddmin): workgroup = one CANDIDATE subsequence of the trace.  Their kernels take this struct; every
// other variant keeps K1Args as it was (same kernel-argument segment, same instructions as the measured kernels)
struct K1MultiArgs : K1Args {
  const uint64_t* cand_masks;   // [n_cand][4]: bit i = external event i of `trace` is part of the candidate
  uint32_t n_cand;
  uint32_t epc;                 // executions per candidate: execution k of every candidate uses seed_base + k (or seeds[k])
  uint32_t* cand_flags;         // [n_cand], zeroed by the host: bit 0 = some execution violated, bit 1 = some execution aborted on a capacity
  uint32_t populate_all;        // 1 = `exists` as given; 0 = the actors the CANDIDATE Start()s (what trace_load derives)
  uint32_t lanes_per_wave;      // 0 = 64; see SPREAD below: the first lanes_per_wave lanes of every wave run executions
};
// SPREAD variant (round 6): a launch far smaller than the chip on as many waves as it has executions.  A wave's iteration costs what
// the UNION of its lanes' paths costs (~780 vector instructions with 56 lanes delivering, ~240 for a lane alone: DESIGN section 4
// K1), so a launch of 100 executions is faster as 100 waves of one lane than as two full waves.  Only the first lanes_per_wave
// lanes of a wave ever claim work; the others idle through the wave's cooperative steps.  Its own struct and its own template
// parameter: every other variant keeps its kernel-argument segment and its instructions.
struct K1SpreadArgs : K1Args {
  uint32_t lanes_per_wave;      // 1 .. 64
};
template <bool MULTI, bool SPREAD = false> struct K1ArgsOf { typedef K1Args type; };
template <> struct K1ArgsOf<true, false> { typedef K1MultiArgs type; };
template <> struct K1ArgsOf<false, true> { typedef K1SpreadArgs type; };

enum : int { PH_IDLE = 0, PH_INJECT = 1, PH_DISPATCH = 2, PH_FINISH = 3 };

constexpr int K1_WAVES = 4;          // waves per workgroup
// LDS-resident pending slots per lane in K1.  The interpreter is VALU-bound and wants residency (32: 12 waves/CU).  The
// specialised build issues as many scalar as vector instructions: it gains from more waves per CU (VALU and SALU of
// different waves issue together) and loses on every `slot < hot ? LDS : scratch` branch that splits a wave, so the host
// gives it none at all: every pending slot then lives in the coalesced [slot][lane] HBM scratch.  That is the fastest
// setting and the one with the most HBM traffic (DESIGN.md section 4 has the table; measured on raft5, ms per 2^20
// schedules: 0 -> 4.84, 2 -> 5.41, 4 -> 5.35, 8 -> 5.62, 16 -> 6.02, 24 / 32 -> 6.6).
// (the generic kernel: 28 - with the timer directory and the reach words of round 3 in LDS, 32 resident slots would leave
// two workgroups per CU where 28 leave three)
#ifndef DEMI_K1_HOT
#define DEMI_K1_HOT 28
#endif
constexpr uint32_t K1_HOT = DEMI_K1_HOT;
constexpr int K1_BATCH = 64;         // schedule indices claimed per atomic

// K1 keeps two more workgroup-shared tables derived from the trace: the network state after every
// injection batch (inject_until_quiescence is schedule-independent: the events are applied in trace
// order whatever the interleaving, so the state after batch j is a function of j alone) and the
// message word of every Send (0 = not a deliverable Send).
// batch words: end index, inaccessible, killed, partitioned lo/hi, sends (offset | count << 16), actors Start()ed, reach lo/hi
// reach: byte `snd` = the created actors a message sent by `snd` reaches, i.e. NOT crosses_partition(snd, .)
// (EventOrchestrator.scala:345-351) under the batch's network state - what every SEND / BCAST of the batch's deliveries asks
// (the BIG layout: words 3 and 4 unused - only the recording variant reads partitions, and it walks the events itself - and the
// reach matrix is 16 rows of 16 bits, words 7 .. 14, read from LDS where the 8 x 8 one travels in a register pair)
__host__ __device__ constexpr uint32_t k1_batch_words(bool big) { return big ? 15u : 9u; }
constexpr uint32_t K1_BATCH_WORDS = k1_batch_words(BIG_TU);
__host__ __device__ inline size_t k1_extra_lds_bytes(uint32_t n_ev, uint32_t n_batches, bool wide = WIDE_TU, bool big = BIG_TU) {
  return ((size_t)n_batches * k1_batch_words(big) * 4 + 2 * (size_t)n_ev * (wide ? 8 : 4) + 15) & ~(size_t)15;
}
// SrcDstFIFO (RandomScheduler.scala:702-909) keeps the actor-to-actor messages apart from the timers / externals:
// one array in arrival order (a pair's queue is the sub-sequence with that (src, dst)); its NORM_HOT first slots in
// LDS, the rest in the HBM spill after the timers-and-externals arrays; srcDsts as a byte list (src * 8 + dst).
constexpr uint32_t NORM_HOT = 16;
// (a pair is listed while it has a queued message: at most p_max <= DEMI_MAX_PENDING of them at a time - what bounds the list
// of a table with more than 11 actors)
__host__ __device__ inline uint32_t k1_pair_words(uint32_t n_actors) {
  const uint32_t pairs = n_actors * n_actors < DEMI_MAX_PENDING ? n_actors * n_actors : DEMI_MAX_PENDING;
  return (pairs + 3) / 4;
}
// (message words are 8 bytes in a wide build; the ids beside them - recording variant - and the pair list stay 32-bit)
__host__ __device__ inline size_t k1_fifo_wave_bytes(uint32_t n_actors, bool rec, bool wide = WIDE_TU) {
  return ((size_t)NORM_HOT * ((wide ? 2 : 1) + (rec ? 1 : 0)) + k1_pair_words(n_actors)) * 64 * 4;
}
// HBM scratch words per simulator lane: pending slots beyond the LDS-resident ones, every array of the variant
__host__ __device__ inline size_t k1_spill_words_per_lane(bool rec, bool fifo, uint32_t hot = K1_HOT) {
  return (size_t)((DEMI_MAX_PENDING - hot) + (fifo ? (DEMI_MAX_PENDING - NORM_HOT) : 0)) * (rec ? 2 : 1);
}
// The timer directory: one byte per (actor, timer type) and lane = the pending slot that holds the ONLY pending copy of that
// timer message (TD_NONE: no copy pending, TD_MANY: there were several at some point - the cancel path then scans).  A
// cancellable.cancel() (notify_timer_cancel -> FullyRandom.remove, RandomScheduler.scala:525-534, 653-664) removes the
// first copy in array order; with the directory that is one byte read instead of probing the pending set, which lives in HBM
// for the specialised kernel.  Four entries share a word; word j of a lane is at [j][lane].
constexpr uint32_t TD_NONE = 0xFFu, TD_MANY = 0xFEu;
__host__ __device__ inline uint32_t k1_tdir_words(uint32_t n_actors, uint32_t n_timer_types) { return (n_actors * n_timer_types + 3) / 4; }
__host__ __device__ inline size_t k1_tdir_wave_bytes(uint32_t n_actors, uint32_t n_timer_types) {
  return (size_t)k1_tdir_words(n_actors, n_timer_types) * 64 * 4;
}
// effect-queue entries of this translation unit's K1: the SEND / BCAST slots of the table's effect schedule, or the whole queue
#ifdef DEMI_JIT_FXQ_SLOTS
constexpr uint32_t K1_FXQ_SLOTS = DEMI_JIT_FXQ_SLOTS;
#else
constexpr uint32_t K1_FXQ_SLOTS = DEMI_FX_CAP;
#endif
// REBIN (round 4): the deliveries of a workgroup's 256 simulators are re-dealt over its lanes by handler class in every
// iteration (see the kernel).  Its LDS: the scheduler state a delivery reads and writes, one 32-bit word per item and
// simulator ([word][workgroup lane]); per class a list of the simulators that deliver a message of that class in this
// iteration (one byte each: the lane within the workgroup); two sets of class counters and "somebody is still running" flags
// (iterations alternate between them, so that one is cleared while the other is in use).
constexpr uint32_t K1_RB_MAX_CLASSES = 16;
__host__ __device__ inline uint32_t k1_rb_classes(uint32_t n_msg_types) { return n_msg_types < K1_RB_MAX_CLASSES ? n_msg_types : K1_RB_MAX_CLASSES; }
// exchanged words per simulator: message word, packed counters, last pending word, tq, resend (2 each), just, rep
// [+ blocked when the table can crash an actor] [+ the application's generator when it has a RND row]
__host__ __device__ constexpr uint32_t k1_rb_words(bool wide, bool crashes, bool rnd) {
  return (wide ? 2u : 1u) * 2u + 1u + 4u + 2u + (crashes ? 1u : 0u) + (rnd ? 2u : 0u);
}
__host__ __device__ inline size_t k1_rb_bytes(uint32_t rb_words, uint32_t rb_classes) {
  return rb_words == 0 ? 0 : (size_t)rb_words * K1_WAVES * 64 * 4 + (size_t)rb_classes * K1_WAVES * 64 + (2 * K1_RB_MAX_CLASSES + 4) * 4;
}
template <bool REC, bool FIFO>
__host__ __device__ inline size_t k1_lds_bytes(uint32_t code_len, uint32_t n_ev, uint32_t n_hs, uint32_t n_actors,
                                               uint32_t n_batches, uint32_t n_timer_types, uint32_t hot = K1_HOT, bool wide = WIDE_TU,
                                               uint32_t fxq_slots = K1_FXQ_SLOTS, uint32_t arr_words = ARR_WORDS,
                                               uint32_t rb_words = 0, uint32_t rb_classes = 0, bool big = BIG_TU) {
  return tables_lds_bytes(code_len, n_ev, n_hs, wide, arr_words, big) + k1_extra_lds_bytes(n_ev, n_batches, wide, big) +
         K1_WAVES * (lane_mem_wave_bytes(n_actors, REC, hot, wide, fxq_slots, arr_words) + (FIFO ? k1_fifo_wave_bytes(n_actors, REC, wide) : 0) +
                     k1_tdir_wave_bytes(n_actors, n_timer_types)) + k1_rb_bytes(rb_words, rb_classes);
}

#ifdef DEMI_K1_MIN_WAVES_PER_EU    // experiment knob of the specialised build: ask for more waves per SIMD (fewer VGPRs)
#define K1_LAUNCH_BOUNDS __launch_bounds__(K1_WAVES * 64, DEMI_K1_MIN_WAVES_PER_EU)
#else
#define K1_LAUNCH_BOUNDS __launch_bounds__(K1_WAVES * 64)
#endif
// CARRY: the unit of work is a RandomScheduler INSTANCE that runs args.epi executions one after the other without reseeding
// its generator(s) (explore() with max_executions > 1, RandomScheduler.scala:248-269: reset_all_state :575-595 clears the
// pending set but keeps the FullyRandom / SrcDstFIFO objects and therefore their Random).  Work index = instance, verdict
// index = instance * epi + execution; lookingFor only applies to the first execution (:586 sets it to None); the instance
// stops at its first violating execution (:257-261) - the verdicts behind it stay as the host zeroed them.  The executions of
// an instance are a sequential chain on one lane; instances run in parallel like the executions of the default mode.
//
// REBIN: the deliveries of the workgroup's 256 simulators are re-dealt over its lanes by handler class, every iteration.
// Lanes sitting in eight different handlers with six effect bodies between them is what a wave pays for in the plain kernel
// (19.5 of 64 lanes active per vector instruction; with 64 copies of ONE execution per wave the same kernel needs 1.57 ms
// instead of 4.0: tools/r4_k1_ceiling.py).  The scheduling step - guards, flush, the random pick - stays with the OWNER lane
// (it is the same code for every simulator).  Then every owner that picked a message posts it: the message word and the part
// of its scheduler state a delivery reads or writes (pending count and last word, the two timer queues, justScheduledTimers,
// the registered repeating timers, the flags) go to LDS, its workgroup lane goes to the list of the message's class (an LDS
// counter per class hands out the positions).  After a barrier lane i of the workgroup takes item i of the concatenated
// lists: it runs the handler rows and applies the effects on the OWNER's state - actor states, effect queue and timer
// directory are LDS columns, the pending set a column of the HBM scratch, all addressable by any lane - so that the lanes of
// a wave now run (almost) one handler.  A second barrier, and every owner takes its state back.  Which lane runs a
// delivery changes nothing about it: verdicts are bit-identical with the plain kernel (tests run both).
//
// MULTI (round 5, demi_random_ddmin = RunnerUtils.randomDDMin's oracle, RunnerUtils.scala:601-623): the launch evaluates a
// FRONTIER of DDMin candidates, each against `epc` random interleavings.  A candidate is a subsequence of the external trace and
// everything K1 derives from the trace is workgroup-shared (the trace itself, the per-batch network states, the Send words), so
// a workgroup IS a candidate (ceil(epc / blockDim) workgroups when epc exceeds one workgroup): thread 0 compacts the trace by
// the candidate's mask in LDS before it builds the batch table, lane k runs execution k - no refill, a lane has one execution -
// and a violating execution sets the candidate's flag.  Verdict cand * epc + k is what the plain kernel returns for
// trace_load(candidate's events) and seed_base + k.
template <bool REC, bool FIFO = false, bool CARRY = false, bool REBIN = false, bool MULTI = false, bool SPREAD = false>
__global__ K1_LAUNCH_BOUNDS void k1_random_explore(const typename K1ArgsOf<MULTI, SPREAD>::type args) {
  static_assert(!SPREAD || (!REC && !CARRY && !REBIN && !MULTI), "the spread variant is the plain per-execution kernel");
  static_assert(!REBIN || (!REC && !FIFO), "the re-binned kernel exists for the non-recording FullyRandom variant");
  static_assert(!MULTI || (!REC && !CARRY && !REBIN), "a frontier of candidates runs the non-recording, per-execution-seed kernel");
  static_assert(!BIG_TU || !REBIN, "the re-binned kernel packs an 8 x 8 reach row: tables of up to 8 actors");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tables t;
  unsigned char* extra = tables_load(t, smem, args.model, args.trace, args.n_ev, args.exists);
  // MULTI: which candidate this workgroup evaluates, and which of its executions this lane runs
  uint32_t m_cand = 0, m_exec = 0;
  if constexpr (MULTI) {
    // (lanes_per_wave < 64: a candidate's executions on the first lanes of more waves - SPREAD above; 64: lane k = execution k)
    const uint32_t m_lpw = args.lanes_per_wave ? args.lanes_per_wave : 64u;
    const uint32_t m_eff = (blockDim.x >> 6) * m_lpw;                    // executions per workgroup
    const uint32_t m_wgpc = (args.epc + m_eff - 1) / m_eff;
    m_cand = blockIdx.x / m_wgpc;
    m_exec = (threadIdx.x & 63u) < m_lpw ? (blockIdx.x % m_wgpc) * m_eff + (threadIdx.x >> 6) * m_lpw + (threadIdx.x & 63u) : 0xFFFFFFFFu;
    // the candidate's events: counted by every thread (t.E is a per-thread value), compacted in place by thread 0 below
    const uint64_t* cm = args.cand_masks + 4 * (size_t)m_cand;
    uint32_t e_cnt = 0;
    for (uint32_t wi = 0; wi < 4; wi++) {
      uint64_t mw = cm[wi];
      if (args.n_ev < 64 * (wi + 1)) mw &= args.n_ev > 64 * wi ? ((1ull << (args.n_ev - 64 * wi)) - 1ull) : 0ull;
      e_cnt += (uint32_t)__popcll(mw);
    }
    if (threadIdx.x == 0) {
      uint64_t* tr = const_cast<uint64_t*>(t.trace);
      uint32_t j = 0;
      for (uint32_t i = 0; i < args.n_ev; i++)
        if ((cm[i >> 6] >> (i & 63)) & 1ull) {
#ifdef DEMI_JIT_NPAY
          reinterpret_cast<word_t*>(extra)[j] = (word_t)i;      // (the event's index in the loaded trace: where its payload area is; s_sendw's place, read back below)
#endif
          tr[j++] = tr[i];
        }
    }
    t.E = e_cnt;
    __syncthreads();
    if (!args.populate_all) {                 // populateActorSystem creates the actors the trace Start()s (ExternalEventInjector.scala:371-378)
      uint32_t ex = 0;
      for (uint32_t i = 0; i < t.E; i++) {
        const uint64_t ev = t.trace[i];
        if (((uint32_t)ev & 0xFF) == DEMI_EV_START) ex |= 1u << ((uint32_t)(ev >> 8) & 0xFF);
      }
      t.exists = ex;
    }
  }
  // (the word arrays first: 8-byte words in a wide build, and `extra` is 16-byte aligned)
  word_t* const s_sendw = reinterpret_cast<word_t*>(extra);
  word_t* const s_bsend = s_sendw + args.n_ev;       // the deliverable Send words, batch after batch, compacted
  uint32_t* const s_batch = reinterpret_cast<uint32_t*>(s_bsend + args.n_ev);
  unsigned char* wave_base = extra + k1_extra_lds_bytes(args.n_ev, args.n_batches);
  if (threadIdx.x == 0) {
    // EventOrchestrator.inject_until_quiescence (:132-189) once per workgroup, for every batch
    uint32_t inacc = t.exists, killed = 0, b_no = 0, n_bs = 0, bs_lo = 0, started = 0;
#ifdef DEMI_BIG
    // the 16 x 16 layout: partitions as one 16-bit row per sender; the batch's reach rows go straight into its words 7 .. 14
    uint32_t prow[MAX_ACT];
    for (uint32_t i = 0; i < MAX_ACT; i++) prow[i] = 0;
    auto put_reach = [&](uint32_t* o, uint32_t inacc_, uint32_t killed_) {
      for (uint32_t k = 0; k < MAX_ACT / 2; k++) o[7 + k] = 0;
      for (uint32_t snd = 0; snd < MAX_ACT; snd++) {
        uint32_t col = 0;
        for (uint32_t r = 0; r < MAX_ACT; r++) col |= ((prow[r] >> snd) & 1u) << r;
        uint32_t cut = prow[snd] | col | inacc_ | (((inacc_ >> snd) & 1u) ? 0xFFFFu : 0u);
        if (!((killed_ >> snd) & 1u)) cut &= ~(1u << snd);        // snd == rcv && !killed: never crosses
        o[7 + (snd >> 1)] |= (t.exists & ~cut & 0xFFFFu) << (16 * (snd & 1u));
      }
    };
    if (t.E == 0) {
      s_batch[0] = 0; s_batch[1] = inacc; s_batch[2] = 0; s_batch[3] = 0; s_batch[4] = 0; s_batch[5] = 0; s_batch[6] = 0;
      put_reach(s_batch, inacc, 0);
    }
#else
    uint64_t part = 0;
    auto reach_of = [&](uint32_t inacc_, uint32_t killed_, uint64_t part_) -> uint64_t {
      uint64_t r = 0;
      for (uint32_t snd = 0; snd < DEMI_MAX_ACTORS; snd++) {
        const uint32_t row = (uint32_t)(part_ >> (snd * 8)) & 0xFFu;
        const uint32_t col = (uint32_t)((((part_ >> snd) & 0x0101010101010101ULL) * 0x0102040810204080ULL) >> 56);
        uint32_t cut = row | col | inacc_ | (((inacc_ >> snd) & 1u) ? 0xFFu : 0u);
        if (!((killed_ >> snd) & 1u)) cut &= ~(1u << snd);        // snd == rcv && !killed: never crosses
        r |= (uint64_t)(t.exists & ~cut & 0xFFu) << (8 * snd);
      }
      return r;
    };
    if (t.E == 0) {
      const uint64_t r0 = reach_of(inacc, 0, 0);
      s_batch[0] = 0; s_batch[1] = inacc; s_batch[2] = 0; s_batch[3] = 0; s_batch[4] = 0; s_batch[5] = 0; s_batch[6] = 0;
      s_batch[7] = (uint32_t)r0; s_batch[8] = (uint32_t)(r0 >> 32);
    }
#endif
    for (uint32_t i = 0; i < t.E; i++) {
      const uint64_t ev = t.trace[i];
      const uint32_t kind = (uint32_t)ev & 0xFF, a = (uint32_t)(ev >> 8) & 0xFF, b = (uint32_t)(ev >> 16) & 0xFF;
      // (a wide table's Sends carry 16-bit payloads: demi_ext_event.p0_hi / p1_hi, zero otherwise)
      const uint32_t ep0 = ((uint32_t)(ev >> 32) & 0xFF) | (WIDE_TU ? ((uint32_t)(ev >> 48) & 0xFF) << 8 : 0u);
      const uint32_t ep1 = ((uint32_t)(ev >> 40) & 0xFF) | (WIDE_TU ? ((uint32_t)(ev >> 56) & 0xFF) << 8 : 0u);
#ifdef DEMI_JIT_NPAY
      // (a table whose messages have more than two fields: the Send's whole payload area, staged by demi_ext_payload_areas or
      // made of P0 / P1 by the load - behind the events in the same array)
      const uint32_t orig = MULTI ? (uint32_t)s_sendw[i] : i;
      const word_t sw = (kind == DEMI_EV_SEND && ((t.exists >> a) & 1))
                            ? msg_word_area((uint32_t)(ev >> 24) & 0xFF, DL, a, args.trace[EXT_AREA_OFFSET + orig] & 0xFFFFFFFFFFFFull)
                            : (word_t)0;
      (void)ep0; (void)ep1;
#else
      const word_t sw = (kind == DEMI_EV_SEND && ((t.exists >> a) & 1))
                            ? msg_word((uint32_t)(ev >> 24) & 0xFF, DL, a, ep0, ep1)
                            : (word_t)0;
#endif
      s_sendw[i] = sw;
      if (sw != 0) s_bsend[n_bs++] = sw;
      if (kind == DEMI_EV_START) { inacc &= ~(1u << a); killed &= ~(1u << a); started |= 1u << a; }
      else if (kind == DEMI_EV_KILL) { killed |= 1u << a; inacc |= 1u << a; }
#ifdef DEMI_BIG
      else if (kind == DEMI_EV_PARTITION) prow[a & (MAX_ACT - 1u)] |= 1u << b;
      else if (kind == DEMI_EV_UNPARTITION) prow[a & (MAX_ACT - 1u)] &= ~(1u << b);
#else
      else if (kind == DEMI_EV_PARTITION) part |= 1ULL << (a * 8 + b);
      else if (kind == DEMI_EV_UNPARTITION) part &= ~(1ULL << (a * 8 + b));
#endif
      if (kind == DEMI_EV_WAIT_QUIESCENCE || i + 1 == t.E) {
        uint32_t* o = s_batch + (size_t)b_no * K1_BATCH_WORDS;
#ifdef DEMI_BIG
        o[0] = i + 1; o[1] = inacc; o[2] = killed; o[3] = 0; o[4] = 0;
#else
        o[0] = i + 1; o[1] = inacc; o[2] = killed; o[3] = (uint32_t)part; o[4] = (uint32_t)(part >> 32);
#endif
        o[5] = bs_lo | ((n_bs - bs_lo) << 16);
        o[6] = started;
#ifdef DEMI_BIG
        put_reach(o, inacc, killed);
#else
        { const uint64_t r = reach_of(inacc, killed, part); o[7] = (uint32_t)r; o[8] = (uint32_t)(r >> 32); }
#endif
        bs_lo = n_bs; started = 0;
        b_no++;
      }
    }
  }
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (REBIN: `mem`, `st` and `tdir` are the simulator's this lane is WORKING on - its own outside the delivery part)
  LaneMem mem = lane_mem_carve(wave_base + (size_t)wave * lane_mem_wave_bytes(t.A, REC, K1_HOT, WIDE_TU, K1_FXQ_SLOTS), t.A, REC, lane, args.spill,
                               (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, K1_HOT);
  const LaneMem own_mem = mem;
  uint64_t* st = mem.st;
  const uint32_t PMAX = args.p_max;
  // SrcDstFIFO arrays of this lane (FIFO builds only)
  word_t* f_norm = nullptr, *f_spill = nullptr;
  uint32_t* f_norm_aux = nullptr, *f_pairs = nullptr, *f_spill_aux = nullptr;
  if (FIFO) {
    unsigned char* fb = wave_base + (size_t)K1_WAVES * lane_mem_wave_bytes(t.A, REC, K1_HOT, WIDE_TU, K1_FXQ_SLOTS) + (size_t)wave * k1_fifo_wave_bytes(t.A, REC);
    f_norm = reinterpret_cast<word_t*>(fb) + lane;
    uint32_t* const after = reinterpret_cast<uint32_t*>(fb + (size_t)NORM_HOT * 64 * sizeof(word_t)) + lane;
    if (REC) f_norm_aux = after;
    f_pairs = after + (REC ? (size_t)NORM_HOT * 64 : 0);
    // the scratch behind the timers-and-externals arrays (lane_mem_carve: their words, then - recording variant - their ids)
    const size_t lanes = (size_t)gridDim.x * blockDim.x, gl = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned char* const fs = reinterpret_cast<unsigned char*>(args.spill) + spill_words(lanes, K1_HOT) * (sizeof(word_t) + (REC ? 4 : 0));
    f_spill = reinterpret_cast<word_t*>(fs) + gl;
    if (REC) f_spill_aux = reinterpret_cast<uint32_t*>(fs + lanes * (DEMI_MAX_PENDING - NORM_HOT) * sizeof(word_t)) + gl;
  }
  const uint32_t f_stride = (uint32_t)((size_t)gridDim.x * blockDim.x);
  auto norm_load = [&](uint32_t slot) -> word_t {
    return slot < NORM_HOT ? f_norm[slot * 64] : f_spill[(size_t)(slot - NORM_HOT) * f_stride];
  };
  auto norm_store = [&](uint32_t slot, word_t v) {
    if (slot < NORM_HOT) f_norm[slot * 64] = v; else f_spill[(size_t)(slot - NORM_HOT) * f_stride] = v;
  };
  auto norm_aux_load = [&](uint32_t slot) -> uint32_t {
    return slot < NORM_HOT ? f_norm_aux[slot * 64] : f_spill_aux[(size_t)(slot - NORM_HOT) * f_stride];
  };
  auto norm_aux_store = [&](uint32_t slot, uint32_t v) {
    if (slot < NORM_HOT) f_norm_aux[slot * 64] = v; else f_spill_aux[(size_t)(slot - NORM_HOT) * f_stride] = v;
  };
  auto pair_get = [&](uint32_t i) -> uint32_t { return (f_pairs[(i >> 2) * 64] >> (8 * (i & 3))) & 0xFFu; };
  auto pair_set = [&](uint32_t i, uint32_t v) {
    uint32_t* q = f_pairs + (size_t)(i >> 2) * 64;
    const uint32_t sh = 8 * (i & 3);
    *q = (*q & ~(0xFFu << sh)) | (v << sh);
  };

  // a specialised build (jit.hpp) knows the model's constants at compile time
#ifdef DEMI_JIT_A
  const uint32_t A = DEMI_JIT_A;
  const uint32_t NTT = DEMI_JIT_NTT, timer_types = DEMI_JIT_TIMER_TYPES;
  const uint64_t tix_packed = DEMI_JIT_TIX_PACKED;
#else
  const uint32_t A = t.A;
  const uint32_t NTT = t.n_timer_types, timer_types = t.timer_types;
  const uint64_t tix_packed = t.tix_packed;
#endif
  // the timer directory of this lane (k1_tdir_words above): entry e = rcv * NTT + timer index is byte (e & 3) of word e >> 2
  unsigned char* const tdir_base = wave_base + (size_t)K1_WAVES * (lane_mem_wave_bytes(t.A, REC, K1_HOT, WIDE_TU, K1_FXQ_SLOTS) + (FIFO ? k1_fifo_wave_bytes(t.A, REC) : 0));
  unsigned char* tdir = tdir_base + (size_t)wave * k1_tdir_wave_bytes(t.A, t.n_timer_types) + (size_t)lane * 4;
  unsigned char* const own_tdir = tdir;
  // REBIN's LDS (k1_rb_bytes): exchanged words [word][workgroup lane], class lists [class][position], counters and flags
#ifdef DEMI_JIT_NO_RND
  constexpr bool APP_RND = false;
#else
  constexpr bool APP_RND = true;
#endif
#ifdef DEMI_JIT_NO_CRASH
  constexpr bool RB_CRASHES = false;
#else
  constexpr bool RB_CRASHES = true;
#endif
  constexpr uint32_t RB_WG = K1_WAVES * 64, RB_WORDS = k1_rb_words(WIDE_TU, RB_CRASHES, APP_RND);
#if defined(DEMI_JIT_NT) && !defined(DEMI_K1_RB_CLASS)
  constexpr uint32_t RB_C = DEMI_JIT_NT < K1_RB_MAX_CLASSES ? DEMI_JIT_NT : K1_RB_MAX_CLASSES;
#elif defined(DEMI_K1_RB_CLASS)
  constexpr uint32_t RB_C = K1_RB_MAX_CLASSES;
#else
  const uint32_t RB_C = k1_rb_classes(t.NT);
#endif
  uint32_t* const rb_x = reinterpret_cast<uint32_t*>(tdir_base + (size_t)K1_WAVES * k1_tdir_wave_bytes(t.A, t.n_timer_types));
  unsigned char* const rb_list = reinterpret_cast<unsigned char*>(rb_x + (size_t)RB_WORDS * RB_WG);
  uint32_t* const rb_cnt = reinterpret_cast<uint32_t*>(rb_list + (size_t)RB_C * RB_WG);      // [2][K1_RB_MAX_CLASSES], then flag[2]
  uint32_t* const rb_flag = rb_cnt + 2 * K1_RB_MAX_CLASSES;
  uint32_t rb_it = 0;                 // parity of the iteration (workgroup-uniform)
  bool wave_live = true;              // some lane of this wave still has (or may get) an execution
  if (REBIN) {
    if (threadIdx.x < 2 * K1_RB_MAX_CLASSES + 2) rb_cnt[threadIdx.x] = 0;
    __syncthreads();
  }
  auto td_ptr = [&](uint32_t e) -> unsigned char* { return tdir + ((e >> 2) << 8) + (e & 3u); };
  // (two bits per message type: one 32-bit shift when the table has at most 16 message types, which a specialised build knows)
#ifdef DEMI_JIT_NT
  constexpr bool TIX32 = DEMI_JIT_NT <= 16u;
#else
  constexpr bool TIX32 = false;
#endif
  auto tix_of = [&](uint32_t type) -> uint32_t {
    return TIX32 ? (((uint32_t)tix_packed >> (2 * type)) & 3u) : ((uint32_t)(tix_packed >> (2 * type)) & 3u);
  };
  // is `pw` a timer message (sender deadLetters, TIMER-class type; externals share the sender), and which entry is its
  auto timer_entry = [&](word_t pw, uint32_t& e) -> bool {
    const uint32_t ty = w_type(pw);
    e = w_dst(pw) * NTT + tix_of(ty);
    return w_src(pw) == DL && ((timer_types >> ty) & 1u);
  };
  const uint32_t E = t.E, exists = t.exists;
  const uint32_t max_messages = args.max_messages ? args.max_messages : 0x7FFFFFFFu;
  const uint32_t interval = args.interval;

  // ---- per-lane simulator state
  int ph = PH_IDLE;
  bool fresh = false;
  uint64_t sched = 0, rng = 0, hash = 0;
  uint64_t inst = 0, inst_end = 0;    // CARRY: this lane's instance and the end of its verdict range
  uint32_t exec_no = 0;               // CARRY: number of the running execution within the instance
  uint64_t app_rng = 0;               // Instrumenter().seededRandom: scala.util.Random(0), new with every execution (DEMI_OP_RND)
  uint32_t n_pend = 0, count = 0, cnt_mod = 0, tidx = 0, inj_lo = 0, inj_hi = 0, batch_no = 0;
  uint32_t fl_off = 0, fl_cnt = 0;    // !REC: the injected batch's Sends as a range of s_bsend, flushed by the whole wave
  // SrcDstFIFO: n_pend counts timersAndExternals, n_norm the actor-to-actor messages, n_pairs = srcDsts.size,
  // pairmask bit (src * 8 + dst) = that pair has a queue; te_rng is timersAndExternals' own generator
  uint32_t n_norm = 0, n_pairs = 0;
  uint64_t pairmask = 0, te_rng = 0;
  Net net;
  net.inaccessible = 0; net.killed = 0; pairs_clear(net.partitioned);
  uint64_t reach = 0;                 // !REC: the batch's reach matrix (K1_BATCH_WORDS above; BIG: read from the batch table instead)
  uint64_t tq = 0, resend = 0;        // messagesToSend timers / timersToResend: 1 byte each (demi_device.hpp tq_pack)
  uint32_t n_tq = 0, n_resend = 0;
  tmask_t just = 0, rep = 0;          // justScheduledTimers / registered repeating timers (bit rcv*4+tidx)
  // the created actors a message sent by `snd` reaches under the current batch's network state
  auto reach_row = [&](uint32_t snd) -> uint32_t {
#ifdef DEMI_BIG
    return (s_batch[(size_t)(batch_no - 1u) * K1_BATCH_WORDS + 7u + (snd >> 1)] >> (16u * (snd & 1u))) & 0xFFFFu;
#else
    return (uint32_t)(reach >> (snd * 8)) & 0xFFu;
#endif
  };
  // does pair `pr` (src * MAX_ACT + dst) have a SrcDstFIFO queue: a bit of pairmask; BIG (256 pairs): a scan of srcDsts
  auto pair_has = [&](uint32_t pr) -> bool {
#ifdef DEMI_BIG
    bool h = false;
    for (uint32_t i = 0; i < n_pairs; i++) h |= pair_get(i) == pr;
    return h;
#else
    return (pairmask >> pr) & 1ull;
#endif
  };
  uint32_t viol = 0, flags = 0;
  uint32_t blocked = 0;               // Instrumenter().blockedActors: actors that crashed (DEMI_OP_CRASH) and were not Start()ed since
  // a specialised build knows whether the table has a CRASH row at all (jit: DEMI_JIT_NO_CRASH): without one `blocked` is
  // provably 0 and the find_non_blocked_message path below is not even compiled
#ifdef DEMI_JIT_NO_CRASH
  constexpr bool CRASHES = false;
#else
  constexpr bool CRASHES = true;
#endif
  uint32_t hits = 0;                  // invariant "hit" mask of the actors (demi_device.hpp invariant_hit), kept up to date per delivery
  // The word in the LAST pending slot, kept in a register: every swap-remove needs it, and the pending set of the specialised
  // kernel lives in HBM.  An append knows it for free; a removal issues the load of the new last word right away and nobody
  // waits for it before the next removal, so the round trip is off the path of a delivery (DEMI_K1_NO_LASTW: load on demand).
  word_t lastw = 0;
#ifdef DEMI_K1_NO_LASTW
#define LASTW() pend_load(mem, n_pend - 1)
#define LASTW_SET(W) do {} while (0)
#define LASTW_RELOAD() do {} while (0)
#else
#define LASTW() lastw
#define LASTW_SET(W) do { lastw = (W); } while (0)
#define LASTW_RELOAD() do { if (n_pend != 0) lastw = pend_load(mem, n_pend - 1); } while (0)
#endif
  uint32_t next_id = 1, n_rec = 0;    // REC only
  demi_rec_event* rec = nullptr;

  // wave-uniform work cursor: [b_next, b_end) is the unassigned rest of the last claimed batch
  uint64_t b_next = 0, b_end = 0;
  bool exhausted = false;

#define REC_PUSH(KIND, SND, RCV, TYPE, AREA, FL, EXT, ID)  /* AREA: demi_rec_event's p0 | p1 << 16 | p_hi << 32 */ \
  do {                                                                                        \
    if (REC) {                                                                                \
      if (n_rec < args.rec_cap) {                                                             \
        demi_rec_event e_;                                                                    \
        e_.kind = (uint8_t)(KIND); e_.snd = (uint8_t)(SND); e_.rcv = (uint8_t)(RCV);          \
        const uint64_t ar_ = (uint64_t)(AREA);                                                \
        e_.msg_type = (uint8_t)(TYPE); e_.p0 = (uint16_t)ar_; e_.p1 = (uint16_t)(ar_ >> 16);  \
        e_.flags = (uint8_t)(FL); e_.ext_idx = (uint8_t)(EXT); e_.p_hi = (uint16_t)(ar_ >> 32); e_.id = (ID); \
        rec[n_rec] = e_;                                                                      \
      }                                                                                       \
      n_rec++;                                                                                \
    }                                                                                         \
  } while (0)

// TD_ENTRY: the timer directory entry of a timer message (>= 0: it now has a copy in slot n_pend), -1 for anything else
#define PEND_APPEND(WORD, ID, TD_ENTRY)                               \
  do {                                                                \
    if (n_pend + n_norm >= PMAX) { flags |= DEMI_V_PENDING_OVF; }     \
    else {                                                            \
      pend_store(mem, n_pend, (WORD));                                \
      LASTW_SET(WORD);                                                \
      if (REC) aux_store(mem, n_pend, (ID));                          \
      if ((TD_ENTRY) >= 0) {                                          \
        unsigned char* const p_ = td_ptr((uint32_t)(TD_ENTRY));       \
        *p_ = (unsigned char)((*p_ == TD_NONE) ? n_pend : TD_MANY);   \
      }                                                               \
      n_pend++;                                                       \
    }                                                                 \
  } while (0)

// SrcDstFIFO.+= for an actor-to-actor message (:791-811): append to the pair's queue, creating the queue (and its
// srcDsts entry) when the pair has none
#define NORM_APPEND(WORD, ID)                                         \
  do {                                                                \
    if (n_pend + n_norm >= PMAX) { flags |= DEMI_V_PENDING_OVF; }     \
    else {                                                            \
      const word_t w_ = (WORD);                                       \
      const uint32_t pr_ = w_src(w_) * MAX_ACT + w_dst(w_);           \
      norm_store(n_norm, w_);                                         \
      if (REC) norm_aux_store(n_norm, (ID));                          \
      if (!pair_has(pr_)) { pair_set(n_pairs, pr_); n_pairs++; if (!BIG_TU) pairmask |= 1ull << pr_; } \
      n_norm++;                                                       \
    }                                                                 \
  } while (0)

#define TIMER_BIT(RCV, TYPE) ((tmask_t)1 << ((RCV) * DEMI_MAX_TIMER_TYPES + tix_of(TYPE)))

  // RandomScheduler.enqueue_timer (:549-559) -> handle_timer (ExternalEventInjector.scala:282-297)
  auto enqueue_timer = [&](uint32_t rcv, uint32_t type, tmask_t tbit) {      // tbit = TIMER_BIT(rcv, type)
    const uint64_t b = (uint64_t)tq_pack(rcv, type, BIG_TU ? tix_of(type) : 0u);
    if (just & tbit) {
      if (n_resend >= DEMI_RESEND_CAP) { flags |= DEMI_V_QUEUE_OVF; return; }
      resend |= b << (8 * n_resend);
      n_resend++;
    } else {
      if (n_tq >= DEMI_TQ_CAP) { flags |= DEMI_V_QUEUE_OVF; return; }
      tq |= b << (8 * n_tq);
      n_tq++;
    }
  };

  // RandomizedHashSet.remove (Util.scala:146-163): the last element moves into the hole
  // (lw = the word in the last slot: callers load it together with the words they inspect, one round trip to the
  // pending set instead of two when it lives in HBM)
  // rw = the word that leaves slot idx.  The timer directory follows: a removed timer whose entry points at idx has no
  // copy left (an entry only points somewhere while that copy is the single one), a timer moved out of the last slot
  // takes its entry along.
  auto pend_remove = [&](uint32_t idx, word_t rw, word_t lw) {
    const uint32_t last = n_pend - 1;
    pend_store(mem, idx, lw);
    if (REC) aux_store(mem, idx, aux_load(mem, last));
    uint32_t e;
    if (timer_entry(rw, e)) { unsigned char* const p = td_ptr(e); if (*p == idx) *p = (unsigned char)TD_NONE; }
    if (idx != last && timer_entry(lw, e)) { unsigned char* const p = td_ptr(e); if (*p == last) *p = (unsigned char)idx; }
    n_pend = last;
    LASTW_RELOAD();          // (after the store above: when idx is the new last slot the load returns lw, same lane, same address)
  };
  // the directory from scratch (after the blocked-actor path has permuted the array)
  auto tdir_rebuild = [&]() {
    for (uint32_t j = 0; j < k1_tdir_words(A, NTT); j++) *reinterpret_cast<uint32_t*>(tdir + (j << 8)) = 0xFFFFFFFFu;
    for (uint32_t q = 0; q < n_pend; q++) {
      uint32_t e;
      if (timer_entry(pend_load(mem, q), e)) { unsigned char* const p = td_ptr(e); *p = (unsigned char)((*p == TD_NONE) ? q : TD_MANY); }
    }
  };

  // The invariant's hit mask is maintained incrementally (one actor changes per delivery), so a check - every
  // `interval` deliveries of every lane, i.e. in almost every iteration of the wave for SOME lane - is a test on a register
  // and only reads the states when two actors hit.
#ifdef DEMI_JIT_A
  const uint32_t inv_kind = DEMI_JIT_INV_KIND, inv_fa = DEMI_JIT_INV_FA, inv_va = DEMI_JIT_INV_VA, inv_fb = DEMI_JIT_INV_FB,
                 fp_mask = DEMI_JIT_FP_MASK;
#else
  const uint32_t inv_kind = t.inv_kind, inv_fa = t.inv_fa, inv_va = t.inv_va, inv_fb = t.inv_fb, fp_mask = t.fp_mask;
#endif
  auto check_invariant = [&]() -> uint32_t {
    if (inv_kind & DEMI_INV_PEERS) {      // the program reads other actors (DEMI_OP_PEER): every actor's hit, from the states as they are
      hits = 0;
      for (uint32_t a = 0; a < A; a++) if ((exists >> a) & 1u) hits |= invariant_hit_at(t, st, a, inv_kind, inv_fa, inv_va) << a;
    }
    const uint32_t fp = invariant_from_hits(t, st, hits & exists, A, inv_kind, inv_fb);
    if (!fp) return 0u;
    if (args.looking_for_valid && (!CARRY || exec_no == 0)) return (((fp ^ args.looking_for) & fp_mask) == 0) ? args.looking_for : 0u;
    return fp;
  };

#ifdef DEMI_K1_PHASES
  uint64_t ph_t[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_iters = 0, ph_active = 0;
  // volatile asm with a memory clobber: keeps loads/stores and control flow on their side of the mark
#define PH_NOW(V) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(V) : : "memory")
#define PH_MARK(I) do { uint64_t now_; PH_NOW(now_); ph_t[I] += now_ - ph_last; ph_last = now_; } while (0)
  uint64_t ph_last; PH_NOW(ph_last);
#else
#define PH_MARK(I) do {} while (0)
#endif
  // Service iterations.  Ending an execution (verdict), claiming the next one and injecting a batch of external events are
  // things a lane does a handful of times per execution, but with 64 lanes some lane needs one of them in almost every
  // iteration, and the wave pays for the whole block each time.  A lane that reaches such a point therefore WAITS (it is inert:
  // its execution is its own) until the wave's next service iteration: when SVC_LANES lanes are waiting, SVC_PERIOD iterations
  // after the last one, or when nobody is left delivering.  The other iterations skip these blocks as a whole (a scalar
  // branch).  DEMI_K1_SVC_PERIOD 0 = every iteration is a service iteration (the order of events of a lane is unchanged either
  // way, so are the verdicts).
#ifndef DEMI_K1_SVC_PERIOD
#define DEMI_K1_SVC_PERIOD 0
#endif
#ifndef DEMI_K1_SVC_LANES
#define DEMI_K1_SVC_LANES 16
#endif
  const uint64_t n_units = CARRY ? (args.n + args.epi - 1) / args.epi : args.n;    // work units: executions, or instances
  uint32_t svc_age = 0;
  for (;;) {
    bool service = true;
    if (DEMI_K1_SVC_PERIOD != 0) {
      const uint64_t waiting = __ballot(ph != PH_DISPATCH);
      service = (uint32_t)__popcll(waiting) >= (uint32_t)DEMI_K1_SVC_LANES || svc_age >= (uint32_t)DEMI_K1_SVC_PERIOD || waiting == ~0ull;
      svc_age = service ? 0u : svc_age + 1u;
    }
    if (service) {
      // ------------------------------------------------------------ verdict
      if (ph == PH_FINISH) {
        // explore(): `if (messagesScheduledSoFar <= maxMessages) checkIfBugFound` (:256-262, 156-180)
        if (!(flags & (DEMI_OVF_ANY | DEMI_V_MAXMSG)) && !viol) viol = check_invariant();
        for (uint32_t i = 0; i < A * ST_WORDS; i++) hash_step(hash, st[i * 64]);
        uint4 v;
        if (flags & DEMI_OVF_ANY) {
          v.x = flags & DEMI_OVF_ANY; v.y = 0; v.z = 0; v.w = 0;
        } else {
          v.x = (flags & 0xFF) | (viol ? DEMI_V_VIOLATION : 0u) | ((tidx & 0xFF) << 8) | ((count < 0xFFFFu ? count : 0xFFFFu) << 16);
          v.y = viol; v.z = (uint32_t)hash; v.w = (uint32_t)(hash >> 32);
        }
        *reinterpret_cast<uint4*>(&args.out[sched]) = v;
        if (REC) args.rec_count[sched] = n_rec;
        if constexpr (MULTI) {
          const uint32_t cf = ((v.x & DEMI_V_VIOLATION) ? 1u : 0u) | ((v.x & DEMI_OVF_ANY) ? 2u : 0u);
          if (cf) atomicOr(&args.cand_flags[m_cand], cf);
        }
        // reset the simulator for the next schedule
        ph = PH_IDLE;
        if (CARRY && !(v.x & DEMI_V_VIOLATION) && sched + 1 < inst_end) {
          // the instance's next execution: reset_all_state + execute_trace again, the generators as they are
          sched++; exec_no++;
          ph = PH_INJECT; fresh = true;
        }
        n_pend = 0; count = 0; cnt_mod = 0; tidx = 0; inj_lo = 0; inj_hi = 0; batch_no = 0;
        tq = 0; resend = 0; n_tq = 0; n_resend = 0; just = 0; rep = 0; viol = 0; flags = 0; hash = 0;
        n_norm = 0; n_pairs = 0; pairmask = 0;
      }
      PH_MARK(10);
      // ------------------------------------------------------------ refill idle lanes
      if constexpr (MULTI) {
        // one execution per lane: execution m_exec of this workgroup's candidate, claimed in the first iteration
        if (!exhausted) {
          exhausted = true;
          if (m_exec < args.epc && m_cand < args.n_cand) { ph = PH_INJECT; fresh = true; sched = (uint64_t)m_cand * args.epc + m_exec; }
        }
        if (__ballot(ph != PH_IDLE) == 0) break;
      } else {
        // (SPREAD: only the first lanes_per_wave lanes take work, and a claim is as many indices)
        uint32_t claim = K1_BATCH;
        bool may_claim = true;
        if constexpr (SPREAD) { claim = args.lanes_per_wave; may_claim = lane < args.lanes_per_wave; }
        const uint64_t idle = __ballot(ph == PH_IDLE && may_claim);
        if (idle != 0 && !exhausted) {                     // wave-uniform
          const uint32_t want = (uint32_t)__popcll(idle);
          const uint64_t have = b_end - b_next;
          uint64_t got = 0;
          if (have < want) {
            // one claim always suffices: at most 64 (SPREAD: lanes_per_wave) lanes ask and a batch holds as many indices
            if (lane == 0) got = atomicAdd(args.work_counter, (unsigned long long)claim);
            got = __shfl(got, 0);
          }
          if (ph == PH_IDLE && may_claim) {
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ULL << lane) - 1));
            const uint64_t my = (rank < have) ? (b_next + rank) : (got + (rank - have));
            if (my < n_units) {
              ph = PH_INJECT; fresh = true;
              if (CARRY) {
                inst = my; exec_no = 0; sched = my * (uint64_t)args.epi;
                inst_end = sched + args.epi < args.n ? sched + args.epi : args.n;
              } else sched = my;
            }
          }
          if (have < want) { b_next = got + (want - have); b_end = got + claim; }
          else b_next += want;
          if (b_next >= n_units) exhausted = true;
        }
        if (!REBIN) { if (__ballot(ph != PH_IDLE) == 0) break; }          // nothing running and nothing left to claim
        else wave_live = __ballot(ph != PH_IDLE) != 0;     // (REBIN: its lanes keep working for the other waves until all are done)
      }
      PH_MARK(0);

      // ------------------------------------------------------------ (re)initialise + inject
      if (ph == PH_INJECT) {
        if (fresh) {
          fresh = false;
          // new execution: `new FullyRandom(seed)`; populateActorSystem isolates every created actor
          // (ExternalEventInjector.scala:371-378)
          if (!CARRY || exec_no == 0) {
            const uint64_t unit = MULTI ? (uint64_t)m_exec : CARRY ? inst : sched;
            const uint64_t seed = args.seeds ? args.seeds[unit] : args.seed_base + unit;
            rng = jr_seed(seed);
            te_rng = rng;                // SrcDstFIFO: both generators are `new Random(seed)`
          }
          app_rng = jr_seed(0);
          hash = 0xCBF29CE484222325ULL;
          net.inaccessible = exists; net.killed = 0; pairs_clear(net.partitioned);
          blocked = 0;
          hits = 0;
          for (uint32_t j = 0; j < k1_tdir_words(A, NTT); j++) *reinterpret_cast<uint32_t*>(tdir + (j << 8)) = 0xFFFFFFFFu;   // no timer pending
          for (uint32_t i = 0; i < A * ST_WORDS; i++) st[i * 64] = t.init[i];
          for (uint32_t a = 0; a < A; a++) hits |= invariant_hit_at(t, st, a, inv_kind, inv_fa, inv_va) << a;
          if (REC) { rec = args.rec_out + (args.rec_shared ? 0ull : sched * (uint64_t)args.rec_cap); n_rec = 0; next_id = 1; }
          batch_no = 0;
        }
        // EventOrchestrator.inject_until_quiescence (:132-189).  Send events are not materialised:
        // messagesToSend's external part is the index range [inj_lo, inj_hi) of the trace.
        inj_lo = tidx;
        if (!REC) {
          // the batch's effect on the network state is a table lookup (computed once per workgroup)
          const uint32_t* bt = s_batch + (size_t)batch_no * K1_BATCH_WORDS;
          batch_no++;
          tidx = bt[0];
#ifdef DEMI_BIG
          net.inaccessible = bt[1]; net.killed = bt[2];      // (partitions and reach: the batch table's, see reach_row)
#else
          net.inaccessible = bt[1]; net.killed = bt[2]; net.partitioned.w = (uint64_t)bt[3] | ((uint64_t)bt[4] << 32);
          reach = (uint64_t)bt[7] | ((uint64_t)bt[8] << 32);
#endif
          fl_off = bt[5] & 0xFFFFu; fl_cnt = bt[5] >> 16;
          blocked &= ~bt[6];            // trigger_start: "allow scheduler to send messages to it again" (EventOrchestrator.scala:224-227)
        }
        bool loop = REC;     // the recording variant walks the events to emit their records
        while (loop && tidx < E) {
          const uint64_t ev = t.trace[tidx];
          const uint32_t kind = (uint32_t)ev & 0xFF, a = (uint32_t)(ev >> 8) & 0xFF, b = (uint32_t)(ev >> 16) & 0xFF;
          if (kind == DEMI_EV_START) {
            REC_PUSH(DEMI_REC_SPAWN, 0, a, 0, 0, 0, tidx, 0);
            net.inaccessible &= ~(1u << a); net.killed &= ~(1u << a); blocked &= ~(1u << a);
          } else if (kind == DEMI_EV_KILL) {
            REC_PUSH(DEMI_REC_KILL, 0, a, 0, 0, 0, tidx, 0);
            net.killed |= 1u << a; net.inaccessible |= 1u << a;
          } else if (kind == DEMI_EV_PARTITION) {
            REC_PUSH(DEMI_REC_PARTITION, a, b, 0, 0, 0, tidx, 0);
            pairs_put(net.partitioned, a, b, true);
          } else if (kind == DEMI_EV_UNPARTITION) {
            REC_PUSH(DEMI_REC_UNPARTITION, a, b, 0, 0, 0, tidx, 0);
            pairs_put(net.partitioned, a, b, false);
          } else if (kind == DEMI_EV_WAIT_QUIESCENCE) {
            REC_PUSH(DEMI_REC_BEGIN_WAIT_QUIESCENCE, 0, 0, 0, 0, 0, tidx, 0);
            loop = false;
          }
          tidx++;
        }
        inj_hi = tidx;
        ph = PH_DISPATCH;
      }
    }   // service

    PH_MARK(1);
    // ------------------------------------------------------------ one scheduling step
    word_t w = 0;              // the message picked by this step
    bool deliver = false;
    const bool disp = (ph == PH_DISPATCH);
    bool none = false, step = false;
    if (disp) {
      if (viol) {
        none = true;                                            // :354-360
      } else if (count > max_messages) {                        // :369-373 finish_early
        flags |= DEMI_V_MAXMSG; tidx = E; none = true;
      } else {
        if (interval > 0 && cnt_mod == 0 && count != 0) {       // :376-394 (lastCheckpoint == 0)
          viol = check_invariant();
          if (viol) none = true;
        }
        step = !none;
      }
    }
    PH_MARK(2);
    // send_external_messages (:424): injected Sends first (no partition check, :298-308).  The Sends of an
    // injection batch are the same words for every schedule, so the wave appends them together: for each lane
    // that has a batch to flush, lane i stores the batch's i-th word into that lane's pending slot n_pend + i.
    if (!REC) {
      uint64_t need = __ballot(step && fl_cnt != 0);
      bool spilled = false;
      while (need) {
        const int src = __builtin_ctzll(need);
        need &= need - 1;
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)fl_cnt, src);
        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)fl_off, src);
        const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)n_pend, src);
        const uint32_t other = (uint32_t)__builtin_amdgcn_readlane((int)n_norm, src);
        for (uint32_t i = lane; i < cnt; i += 64) {
          const uint32_t slot = base + i;
          if (slot + other >= PMAX) break;
          const word_t sw = s_bsend[off + i];
          if (slot < K1_HOT) (mem.pend - lane + src)[slot * 64] = sw;
          else { *spill_at(mem.spill, __umul24(slot - K1_HOT, mem.spill_stride) + (mem.spill_lane - lane + (uint32_t)src)) = sw; spilled = true; }
        }
      }
      // No fence after writing other lanes' spill slots: the owner lane belongs to this same wave, a wave's vector
      // memory instructions are issued in order, and the memory pipeline keeps accesses to one address in that order
      // (the same guarantee a lane relies on when it re-reads its own store).  The fence waited for every outstanding
      // store of the wave in almost every iteration: 4.52 -> 4.41 ms per 2^20 schedules.  DEMI_K1_FLUSH_FENCE restores it.
#ifdef DEMI_K1_FLUSH_FENCE
      if (__ballot(spilled) != 0) __threadfence_block();
#else
      // what stays is the COMPILER's side of that order: a wavefront-scope release fence constrains the reordering of the
      // stores above against the owner lane's later loads and costs no s_waitcnt on this target
      if (__ballot(spilled) != 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#endif
      if (step && fl_cnt != 0) {
        if (n_pend + n_norm + fl_cnt > PMAX) { flags |= DEMI_V_PENDING_OVF; n_pend = PMAX - n_norm; }
        else { n_pend += fl_cnt; LASTW_SET(s_bsend[fl_off + fl_cnt - 1]); }     // (an overflowing execution ends here)
        fl_cnt = 0;
      }
    }
    if (step) {
      if (REC) {
        for (uint32_t i = inj_lo; i < inj_hi; i++) {
          const word_t sw = s_sendw[i];
          if (sw != 0) {
            const uint32_t id = next_id; next_id++;
            PEND_APPEND(sw, id, -1);
            REC_PUSH(DEMI_REC_MSG_SEND, DL, w_dst(sw), w_type(sw), w_area(sw), 1, i, id);
          }
        }
        inj_lo = inj_hi;
      }
      // ... then timers: internal messages from deadLetters, dropped when the receiver is
      // inaccessible (crosses_partition(deadLetters, rcv), :287-297)
      for (uint32_t k = 0; k < n_tq; k++) {
        const uint32_t bt = (uint32_t)(tq >> (8 * k)) & 0xFF, rcv = tq_rcv(bt), type = tq_type(bt);
        const uint32_t id = next_id; if (REC) next_id++;
        const bool drop = (net.inaccessible >> rcv) & 1;
        if (!drop) PEND_APPEND(msg_word(type, DL, rcv, 0, 0), id, (int32_t)(rcv * NTT + tix_of(type)));
        REC_PUSH(DEMI_REC_MSG_SEND, DL, rcv, type, 0, 2 | (drop ? 4 : 0), 255, id);
      }
      tq = 0; n_tq = 0;
      if ((flags & DEMI_OVF_ANY) || n_pend + n_norm == 0) none = true;
    }
    PH_MARK(3);
    if (disp) {
      uint32_t wid = 0;
      // SrcDstFIFO.dequeue (:764-774) of pair number pi of srcDsts: the head of its queue; the gap closes (arrival order is the
      // FIFO order) and the pair leaves srcDsts when its queue is empty
      auto fifo_dequeue = [&](uint32_t pi) {
        const uint32_t pr = pair_get(pi);
        uint32_t k = 0;
        word_t cur = norm_load(0);
        while (k + 1 < n_norm && w_src(cur) * MAX_ACT + w_dst(cur) != pr) { k++; cur = norm_load(k); }   // queue.head
        w = cur;
        if (REC) wid = norm_aux_load(k);
        bool more = false;
        for (uint32_t j = k; j + 1 < n_norm; j++) {
          const word_t nx = norm_load(j + 1);
          more |= (w_src(nx) * MAX_ACT + w_dst(nx) == pr);
          norm_store(j, nx);
          if (REC) norm_aux_store(j, norm_aux_load(j + 1));
        }
        n_norm--;
        if (!more) {                                     // srcDstToMessages -= srcDst; srcDsts.remove(idx)
          for (uint32_t j = pi; j + 1 < n_pairs; j++) pair_set(j, pair_get(j + 1));
          n_pairs--;
          if (!BIG_TU) pairmask &= ~(1ull << pr);
        }
      };
      bool picked = false;          // the message was already chosen (and removed) by the blocked-actor path
      if (CRASHES && !none && blocked != 0) {
        // Some actor crashed: Util.find_non_blocked_message (Util.scala:470-489).  Draw until the receiver is not blocked;
        // what was drawn for a blocked actor is set aside and re-appended afterwards in draw order - which permutes
        // arr, so it is replayed literally.  Rare (only executions with a crashed actor come here), hence simple.
        auto find_non_blocked = [&](uint64_t& g) -> bool {
          const uint32_t n0 = n_pend;
          uint32_t k = 0;
          bool found = false;
          while (n_pend > 0) {
            const uint32_t i = jr_next_int(g, n_pend, t.magic);
            const word_t cw = pend_load(mem, i);
            const uint32_t cid = REC ? aux_load(mem, i) : 0u;
            pend_remove(i, cw, LASTW());
            if ((blocked >> w_dst(cw)) & 1u) {           // set aside in the slot this removal just freed
              pend_store(mem, n0 - 1 - k, cw);
              if (REC) aux_store(mem, n0 - 1 - k, cid);
              k++;
              continue;
            }
            w = cw; wid = cid; found = true;
            break;
          }
          // the rejected ones sit in slots n0 - 1 down to n0 - k in draw order: reverse them in place, then close the gap
          // the accepted element left, so that arr = remaining ++ rejected (collection ++= blocked)
          if (k > 1) {
            for (uint32_t lo = n0 - k, hi = n0 - 1; lo < hi; lo++, hi--) {
              const word_t x = pend_load(mem, lo), y = pend_load(mem, hi);
              pend_store(mem, lo, y); pend_store(mem, hi, x);
              if (REC) { const uint32_t ax = aux_load(mem, lo), ay = aux_load(mem, hi); aux_store(mem, lo, ay); aux_store(mem, hi, ax); }
            }
          }
          if (found)
            for (uint32_t j = 0; j < k; j++) {
              pend_store(mem, n0 - k - 1 + j, pend_load(mem, n0 - k + j));
              if (REC) aux_store(mem, n0 - k - 1 + j, aux_load(mem, n0 - k + j));
            }
          n_pend = n0 - (found ? 1u : 0u);
          LASTW_RELOAD();
          tdir_rebuild();                                // which slot holds which timer message, from scratch
          return found;
        };
        bool found = false;
        if (!FIFO) {
          found = find_non_blocked(rng);
        } else {
          // SrcDstFIFO.getNonBlockedMessage (:716-760) with blocked receivers
          bool open_pair = false;                        // a pair queue whose receiver is not blocked
          for (uint32_t i = 0; i < n_pairs; i++) open_pair |= !((blocked >> (pair_get(i) & (MAX_ACT - 1u))) & 1u);
          if (!open_pair) {
            found = find_non_blocked(te_rng);            // (:717-729) "only timers left"
          } else {
            if (jr_next_int(rng, n_pend + n_norm, t.magic) < n_pend) found = find_non_blocked(te_rng);
            if (!found) {
              uint32_t pi = jr_next_int(rng, n_pairs, t.magic);
              while ((blocked >> (pair_get(pi) & (MAX_ACT - 1u))) & 1u) pi = jr_next_int(rng, n_pairs, t.magic);
              fifo_dequeue(pi);
              found = true;
            }
          }
        }
        picked = found;
        if (!found) none = true;
      }
      if (!none) {
        if (!picked) {
          bool from_te = true;
          uint32_t idx = 0;
          if (!FIFO) {
            // FullyRandom.removeRandomElement -> RandomizedHashSet: nextInt(arr.length), swap with last
            idx = jr_next_int(rng, n_pend, t.magic);
          } else if (n_pairs == 0) {
            // SrcDstFIFO.getNonBlockedMessage (:716-729): only timers / externals left
            idx = jr_next_int(te_rng, n_pend, t.magic);
          } else {
            // (:731-759) a timer / external with probability |timersAndExternals| / |allMessages|, else a random pair's head
            from_te = jr_next_int(rng, n_pend + n_norm, t.magic) < n_pend;
            if (from_te) idx = jr_next_int(te_rng, n_pend, t.magic);
          }
          if (from_te) {
            w = pend_load(mem, idx);
            if (REC) wid = aux_load(mem, idx);
            pend_remove(idx, w, LASTW());
          } else {
            fifo_dequeue(jr_next_int(rng, n_pairs, t.magic));
          }
        }
        count++;
        cnt_mod++; if (cnt_mod == interval) cnt_mod = 0;
        const uint32_t type = w_type(w), me = w_dst(w);
        REC_PUSH(DEMI_REC_MSG_EVENT, w_src(w), me, type, w_area(w), 0, 255, wid);
        hash_step(hash, w);
        // updateRepeatingTimer (:405-421) and the Instrumenter's retrigger (Instrumenter.scala:1008-1016)
        const tmask_t tbit = TIMER_BIT(me, type);
        const bool is_rep = ((timer_types >> type) & 1u) && (rep & tbit);
        if (is_rep) {
          just |= tbit;
          enqueue_timer(me, type, tbit);      // parked in timersToResend (it is in justScheduledTimers)
        } else {
          // timersToResend re-enter messagesToSend in order (RandomScheduler.scala:416-419): one shifted OR
          if (n_resend != 0) {
            if (n_tq + n_resend > DEMI_TQ_CAP) flags |= DEMI_V_QUEUE_OVF;
            else { tq |= resend << (8 * n_tq); n_tq += n_resend; }
          }
          resend = 0; n_resend = 0; just = 0;
        }
        deliver = !(flags & DEMI_OVF_ANY);
        if (!deliver) ph = PH_FINISH;
      } else {
        // quiescence: notify_quiescence (:487-500) / handle_quiescence (ExternalEventInjector.scala:541-580)
        if ((flags & DEMI_OVF_ANY) || viol || tidx >= E) {
          ph = PH_FINISH;
        } else {
          REC_PUSH(DEMI_REC_QUIESCENCE, 0, 0, 0, 0, 0, 255, 0);
          ph = PH_INJECT;
        }
      }
    }

    PH_MARK(4);
    // ------------------------------------------------------------ REBIN: post the delivery, take somebody's
    uint32_t rb_me = 0xFFu;           // the receiver of the message THIS lane's simulator posted (0xFF: none)
    uint32_t rb_owner = 0, rb_hit = 0;
    if (REBIN) {
      const uint32_t tid = threadIdx.x;
      uint32_t* const xo = rb_x + tid;
      uint32_t* const cnt = rb_cnt + rb_it * K1_RB_MAX_CLASSES;
      // every lane parks its scheduler state in its own column (the registers are about to hold another simulator's)
      uint32_t k = 0;
      {
        const uint32_t me_ = w_dst(w);
        xo[k++ * RB_WG] = (flags & 0xFu) | (n_pend << 4) | (n_tq << 12) | (n_resend << 16) | (((uint32_t)(reach >> (me_ * 8)) & 0xFFu) << 20);
        xo[k++ * RB_WG] = (uint32_t)lastw; if (WIDE_TU) xo[k++ * RB_WG] = (uint32_t)((uint64_t)lastw >> 32);
        xo[k++ * RB_WG] = (uint32_t)tq; xo[k++ * RB_WG] = (uint32_t)(tq >> 32);
        xo[k++ * RB_WG] = (uint32_t)resend; xo[k++ * RB_WG] = (uint32_t)(resend >> 32);
        xo[k++ * RB_WG] = just; xo[k++ * RB_WG] = rep;
        if (RB_CRASHES) xo[k++ * RB_WG] = blocked;
        if (APP_RND) { xo[k++ * RB_WG] = (uint32_t)app_rng; xo[k++ * RB_WG] = (uint32_t)(app_rng >> 32); }
        if (deliver) {
          xo[k++ * RB_WG] = (uint32_t)w; if (WIDE_TU) xo[k++ * RB_WG] = (uint32_t)((uint64_t)w >> 32);
          // (any mapping is correct; the default groups by handler.  DEMI_K1_RB_CLASS(message word, low word of the receiver's
          // state) lets an experiment build try another one)
#ifdef DEMI_K1_RB_CLASS
          const uint32_t cls = (uint32_t)(DEMI_K1_RB_CLASS(w, (uint32_t)st[(ST_WORDS * me_) * 64])) & (K1_RB_MAX_CLASSES - 1u);
#else
          const uint32_t cls = w_type(w) & (K1_RB_MAX_CLASSES - 1u);
#endif
          const uint32_t pos = atomicAdd(&cnt[cls < RB_C ? cls : 0u], 1u);
          rb_list[(cls < RB_C ? cls : 0u) * RB_WG + pos] = (unsigned char)tid;
          rb_me = me_;
        }
      }
      if (lane == 0 && wave_live) rb_flag[rb_it] = 1u;
      __syncthreads();
      if (rb_flag[rb_it] == 0u) break;                                        // every wave is done (workgroup-uniform)
      // (the other set of counters / the other flag: last read before the previous iteration's second barrier, next written
      // after this iteration's)
      if (tid < K1_RB_MAX_CLASSES) rb_cnt[(rb_it ^ 1u) * K1_RB_MAX_CLASSES + tid] = 0u;
      if (tid == K1_RB_MAX_CLASSES) rb_flag[rb_it ^ 1u] = 0u;
      // item `tid` of the concatenated class lists
      uint32_t acc = 0, base = 0, csel = 0;
      for (uint32_t c = 0; c < RB_C; c++) {
        const uint32_t n_c = cnt[c];
        if (tid >= acc) { csel = c; base = acc; }
        acc += n_c;
      }
      deliver = tid < acc;
      if (deliver) {
        const uint32_t o = rb_list[csel * RB_WG + (tid - base)];
        rb_owner = o;
        const uint32_t ow = o >> 6, ol = o & 63u;
        mem = lane_mem_carve(wave_base + (size_t)ow * lane_mem_wave_bytes(t.A, REC, K1_HOT, WIDE_TU, K1_FXQ_SLOTS), t.A, REC, ol, args.spill,
                             (size_t)blockIdx.x * blockDim.x + o, (size_t)gridDim.x * blockDim.x, K1_HOT);
        st = mem.st;
        tdir = tdir_base + (size_t)ow * k1_tdir_wave_bytes(t.A, t.n_timer_types) + (size_t)ol * 4;
        const uint32_t* const xi = rb_x + o;
        uint32_t j = 0;
        const uint32_t pk = xi[j++ * RB_WG];
        flags = pk & 0xFu; n_pend = (pk >> 4) & 0xFFu; n_tq = (pk >> 12) & 0xFu; n_resend = (pk >> 16) & 0xFu;
        { uint64_t lw = xi[j++ * RB_WG]; if (WIDE_TU) lw |= (uint64_t)xi[j++ * RB_WG] << 32; lastw = (word_t)lw; }
        tq = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2;
        resend = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2;
        just = xi[j++ * RB_WG]; rep = xi[j++ * RB_WG];
        if (RB_CRASHES) blocked = xi[j++ * RB_WG];
        if (APP_RND) { app_rng = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2; }
        { uint64_t mw = xi[j++ * RB_WG]; if (WIDE_TU) mw |= (uint64_t)xi[j++ * RB_WG] << 32; w = (word_t)mw; }
        reach = (uint64_t)((pk >> 20) & 0xFFu) << (w_dst(w) * 8);              // (the receiver's row is all a delivery asks)
      }
    }
    // ------------------------------------------------------------ the receiver's handler rows
    uint32_t nfx = 0;
    if (deliver) nfx = DEMI_VM_RUN(t, mem, w, flags, app_rng);
    if (deliver) {      // the receiver's new state decides its bit of the invariant's hit mask
      const uint32_t me_ = w_dst(w);
      if (!(inv_kind & DEMI_INV_PEERS)) {     // (with PEER rows the mask is rebuilt at every check: check_invariant)
        rb_hit = invariant_hit_at(t, st, me_, inv_kind, inv_fa, inv_va);
        if (!REBIN) hits = (hits & ~(1u << me_)) | (rb_hit << me_);
      }
    }
    PH_MARK(5);
#ifdef DEMI_K1_PHASES
    ph_iters++; ph_active += __popcll(__ballot(deliver));
#endif

    // ------------------------------------------------------------ apply the recorded effects
    const uint32_t me = w_dst(w);
    // receivers of a SEND / BCAST effect that are not cut off: crosses_partition(me, .) for all receivers at once (row and
    // column of the ordered-pair matrix, the column gathered by a multiply; inaccessible receivers; isolated sender)
    auto send_targets = [&](uint32_t fx) -> uint32_t {    // (the low word of the effect: op, type, target)
      const bool bc = ((fx & 31u) == DEMI_OP_BCAST);
      const uint32_t target = fx_target(fx);
      const uint32_t tm = bc ? ~(1u << me) : (1u << target);       // (target FX_NOBODY: outside the matrix row)
      return tm & reach_row(me);                                   // created, and not cut off from `me` (reach_of above)
    };
    // event_produced for internal messages (:287-297): dropped at send time when crosses_partition, else appended
    auto apply_send = [&](word_t fxw) {
      const uint32_t fx = (uint32_t)fxw;
      const uint32_t type = (fx >> 5) & 31u;
      if (REC) {
        const bool bc = ((fx & 31u) == DEMI_OP_BCAST);
        const uint32_t target = fx_target(fx);
        const uint32_t first = bc ? 0u : target, last = bc ? A : (target < A ? target + 1 : 0u);
        for (uint32_t r = first; r < last; r++) {
          if ((bc && r == me) || !((exists >> r) & 1)) continue;
          const uint32_t id = next_id; next_id++;
          const bool drop = crosses_partition(net, me, r);
          if (!drop) {
            if (FIFO) NORM_APPEND(fx_msg_word(fxw, type, me, r), id);
            else PEND_APPEND(fx_msg_word(fxw, type, me, r), id, -1);
          }
          REC_PUSH(DEMI_REC_MSG_SEND, me, r, type, fx_area(fxw), drop ? 4 : 0, 255, id);
        }
      } else {
        uint32_t tm = send_targets(fx);
        const word_t base = fx_msg_word(fxw, type, me, 0);
        while (tm) {
          const uint32_t r = (uint32_t)__builtin_ctz(tm);
          tm &= tm - 1;
          if (FIFO) NORM_APPEND(base | (r << 5), 0u);
          else PEND_APPEND(base | (r << 5), 0u, -1);
        }
      }
    };
    // cancelTimer (Instrumenter.scala:159-168) -> notify_timer_cancel (:525-534)
    auto apply_cancel = [&](uint32_t type, tmask_t tbit, uint32_t tix) {      // tbit = TIMER_BIT(me, type), tix = its timer index
      rep &= ~tbit;
      const uint32_t want = tq_pack(me, type, tix);
      // handle_timer_cancel: messagesToSend first.  The first of its n_tq bytes equal to `want`, all eight
      // compared at once (zero-byte test on tq ^ want...want; its lowest hit is exact)
      bool found = false;
      if (n_tq != 0) {
        const uint64_t x = tq ^ (0x0101010101010101ull * (uint64_t)want);
        uint64_t z = (x - 0x0101010101010101ull) & ~x & 0x8080808080808080ull;
        z &= (n_tq >= 8) ? ~0ull : ((1ull << (8 * n_tq)) - 1ull);
        if (z != 0) {
          const uint32_t q = (uint32_t)__builtin_ctzll(z) >> 3;
          const uint64_t lowm = (1ull << (8 * q)) - 1ull;
          tq = (tq & lowm) | ((tq >> 8) & ~lowm);
          n_tq--; found = true;
        }
      }
      if (!found) {
        // FullyRandom.remove (:653-664): first match in arr order, then swap-remove.  The timer directory knows the slot
        // when the message has exactly one pending copy (the usual case: a timer is cancelled and set again), and that
        // there is nothing to remove when it has none - no access to the pending set (HBM in the specialised kernel)
        // beyond the word of the last slot, which the swap-remove needs anyway.
        unsigned char* const p = td_ptr(me * NTT + tix);
        const uint32_t d = *p;
        if (d != TD_NONE) {
          const word_t wantw = msg_word(type, DL, me, 0, 0);
          if (d != TD_MANY) {
            pend_remove(d, wantw, LASTW());
          } else {
            // several copies were pending at some point: scan.  The scan also counts the copies, so the entry becomes exact
            // again when at most one is left after this removal.
            uint32_t first = 0xFFFFFFFFu, second = 0xFFFFFFFFu, copies = 0;
            for (uint32_t q = 0; q < n_pend; q++)
              if (pend_load(mem, q) == wantw) {
                if (copies == 0) first = q; else if (copies == 1) second = q;
                copies++;
              }
            if (copies != 0) {
              const uint32_t last = n_pend - 1;
              pend_remove(first, wantw, LASTW());   // (the entry is TD_MANY: pend_remove leaves it alone)
              if (copies == 2) *p = (unsigned char)(second == last ? first : second);
            }
            if (copies <= 1) *p = (unsigned char)TD_NONE;
          }
        }
      }
    };
    // TSET / TREP: registerCancellable + handleTick (Instrumenter.scala:1145-1200)
    auto apply_timer_set = [&](bool repeating, uint32_t type, tmask_t bit) {      // bit = TIMER_BIT(me, type)
      if (!(rep & bit)) {               // else "Non-unique timer" (:1154-1157)
        if (repeating) rep |= bit;
        enqueue_timer(me, type, bit);
      }
    };
    if (deliver) {      // every effect row in program order
#ifdef DEMI_JIT_FX_SCHED
      // A table compiled with an effect schedule (jit.hpp fx_schedule): every effect row has a fixed slot whose class -
      // send, or (timer op, timer type) - is a compile-time constant, slots increase along every path of every handler
      // (program order), and `nfx` is the mask of the slots this delivery filled: one straight pass, each body once.
#define DEMI_FX_SLOT(J, KIND, OP, TYPE, TIDX, Q)                                                                      \
      if (((nfx >> (J)) & 1u) && !(flags & DEMI_OVF_ANY)) {                                                           \
        if ((KIND) == 0u) apply_send(mem.fxq[(Q) * 64]);                     /* (Q: the slot's entry of the effect queue) */ \
        else if ((KIND) == 1u) apply_cancel((TYPE), (tmask_t)1 << (me * DEMI_MAX_TIMER_TYPES + (TIDX)), (TIDX));               \
        else if ((KIND) == 2u) apply_timer_set((OP) == DEMI_OP_TREP, (TYPE), (tmask_t)1 << (me * DEMI_MAX_TIMER_TYPES + (TIDX))); \
        else if (CRASHES) blocked |= 1u << me;                                                                        \
      }                                                                                                               \
      PH_MARK((KIND) == 0u ? 6 : (KIND) == 1u ? 7 : 8);      /* (after the slot, where the wave has reconverged: every lane's clock) */
      DEMI_JIT_FX_APPLY
#undef DEMI_FX_SLOT
#else
      for (uint32_t k = 0; k < nfx && !(flags & DEMI_OVF_ANY); k++) {
        const word_t fxw = mem.fxq[k * 64];
        const uint32_t fx = (uint32_t)fxw;
        const uint32_t op = fx & 31u, type = (fx >> 5) & 31u;
        PH_MARK(9);
        if (op <= DEMI_OP_BCAST) { apply_send(fxw); PH_MARK(6); }
        else if (op == DEMI_OP_CRASH) { if (CRASHES) blocked |= 1u << me; }   // actorCrashed (Instrumenter.scala:184-199)
        else if (op == DEMI_OP_TCANCEL) { apply_cancel(type, TIMER_BIT(me, type), tix_of(type)); PH_MARK(7); }
        else { apply_timer_set(op == DEMI_OP_TREP, type, TIMER_BIT(me, type)); PH_MARK(8); }
      }
#endif
      if (!REBIN && (flags & DEMI_OVF_ANY)) ph = PH_FINISH;
    }
    // ------------------------------------------------------------ REBIN: hand the state back, take one's own
    if (REBIN) {
      if (deliver) {
        uint32_t* const xb = rb_x + rb_owner;
        uint32_t k = 0;
        xb[k++ * RB_WG] = (flags & 0xFu) | (n_pend << 4) | (n_tq << 12) | (n_resend << 16) | (rb_hit << 28);
        xb[k++ * RB_WG] = (uint32_t)lastw; if (WIDE_TU) xb[k++ * RB_WG] = (uint32_t)((uint64_t)lastw >> 32);
        xb[k++ * RB_WG] = (uint32_t)tq; xb[k++ * RB_WG] = (uint32_t)(tq >> 32);
        xb[k++ * RB_WG] = (uint32_t)resend; xb[k++ * RB_WG] = (uint32_t)(resend >> 32);
        xb[k++ * RB_WG] = just; xb[k++ * RB_WG] = rep;
        if (RB_CRASHES) xb[k++ * RB_WG] = blocked;
        if (APP_RND) { xb[k++ * RB_WG] = (uint32_t)app_rng; xb[k++ * RB_WG] = (uint32_t)(app_rng >> 32); }
      }
      __syncthreads();
      {
        const uint32_t* const xi = rb_x + threadIdx.x;
        uint32_t j = 0;
        const uint32_t pk = xi[j++ * RB_WG];
        flags = pk & 0xFu; n_pend = (pk >> 4) & 0xFFu; n_tq = (pk >> 12) & 0xFu; n_resend = (pk >> 16) & 0xFu;
        { uint64_t lw = xi[j++ * RB_WG]; if (WIDE_TU) lw |= (uint64_t)xi[j++ * RB_WG] << 32; lastw = (word_t)lw; }
        tq = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2;
        resend = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2;
        just = xi[j++ * RB_WG]; rep = xi[j++ * RB_WG];
        if (RB_CRASHES) blocked = xi[j++ * RB_WG];
        if (APP_RND) { app_rng = xi[j * RB_WG] | ((uint64_t)xi[(j + 1) * RB_WG] << 32); j += 2; }
        if (rb_me != 0xFFu) {
          hits = (hits & ~(1u << rb_me)) | (((pk >> 28) & 1u) << rb_me);
          if (flags & DEMI_OVF_ANY) ph = PH_FINISH;
        }
      }
      mem = own_mem; st = own_mem.st; tdir = own_tdir;
      if (batch_no != 0) {            // its batch's reach matrix again (the register held the worked-on simulator's row)
        const uint32_t* bt = s_batch + (size_t)(batch_no - 1) * K1_BATCH_WORDS;
        reach = (uint64_t)bt[7] | ((uint64_t)bt[8] << 32);
      }
      rb_it ^= 1u;
    }

    PH_MARK(9);
  }
#ifdef DEMI_K1_PHASES
  PH_MARK(11);
  if (lane == 0 && args.phase_out) {
    unsigned long long* o = args.phase_out + ((size_t)blockIdx.x * K1_WAVES + wave) * 16;
    for (int i = 0; i < 12; i++) o[i] = ph_t[i];
    o[14] = ph_iters; o[15] = ph_active;
  }
#endif
#undef PH_MARK
#undef REC_PUSH
#undef PEND_APPEND
#undef NORM_APPEND
#undef LASTW
#undef LASTW_SET
#undef LASTW_RELOAD
#undef TIMER_BIT
}

}  // namespace demi
