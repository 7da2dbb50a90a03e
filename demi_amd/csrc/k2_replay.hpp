// k2_replay.hpp — K2: one STSScheduler.test (no peek) per wavefront lane: DDMin's replay oracle.
//
// Restates STSScheduler.test / advanceReplay / schedule_new_message / event_produced /
// notify_timer_cancel (schedulers/STSScheduler.scala:199-310, 405-559, 643-776, 561-623, 828-855)
// and the projection of the original trace onto a candidate subsequence
// (EventTrace.subsequenceIntersection + filterSends, EventTrace.scala:290-452).
//
// lane = candidate subsequence of the external events (a 256-bit mask).  The recorded original
// execution is lowered once on the host to a flat array of "expected" events (8 bytes each) that
// every lane walks with its own cursor; the per-candidate projection is evaluated on the fly:
//   * a recorded Spawn/Kill/Partition/UnPartition is kept iff it equals (by name) the head of the
//     candidate's remaining non-Send externals, and dropped once those are exhausted;
//   * an external MsgSend, and the MsgEvent with the same id, are kept iff their Send is in the mask;
//   * an expected MsgEvent is delivered iff a message with the same (snd, rcv, fingerprint) is
//     pending, otherwise it is ignored ("Ignoring message", STSScheduler.scala:528-529).
// Messages with equal (snd, rcv, fingerprint) are interchangeable, so the per-key FIFO of the
// reference is realised as "any pending entry with this word" + swap-remove.
#pragma once

#include "sim_core.hpp"

#ifndef DEMI_VM_RUN   // the table interpreter, unless a specialised build supplies the compiled handlers (jit.hpp)
#define DEMI_VM_RUN vm_run
#endif

namespace demi {

// expected event, 8 bytes: kind | a<<8 | b<<16 | type<<24 | p0<<32 | p1<<40 | ext<<48 | flags<<56
//   SPAWN/KILL: a = actor;  (UN)PARTITION: a, b;  MSG_SEND (external only): b = rcv, ext = Send index
//   MSG_EVENT: a = snd, b = rcv, ext = index of the Send that enqueued it (255 = internal / timer)
struct K2Args {
  const DevModel* model;
  const uint64_t* ext;      // original external events [n_ext]
  uint32_t n_ext;
  uint32_t exists;
  const uint64_t* expected; // lowered original trace [n_exp]
  uint32_t n_exp;
  uint32_t p_max, looking_for;
  const uint64_t* masks;    // [n][4]; null = every external kept
  const uint32_t* skip;     // [n] index (in `expected`) of one MSG_EVENT removed from the trace, or null
  uint8_t* kept;            // [n][n_exp] (pre-zeroed) 1 where the expected event took effect, or null
  uint64_t n;
  demi_verdict* out;
  unsigned long long* work_counter;
  uint32_t* spill;
};

constexpr int K2_WAVES = 4;

__host__ __device__ inline size_t k2_lds_bytes(uint32_t code_len, uint32_t n_ext, uint32_t n_hs, uint32_t n_actors) {
  return tables_lds_bytes(code_len, n_ext, n_hs) + K2_WAVES * lane_mem_wave_bytes(n_actors, false);
}

__global__ __launch_bounds__(K2_WAVES * 64) void k2_replay(const K2Args args) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Tables t;
  unsigned char* wave_base = tables_load(t, smem, args.model, args.ext, args.n_ext, args.exists);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const LaneMem mem = lane_mem_carve(wave_base + (size_t)wave * lane_mem_wave_bytes(t.A, false), t.A, false, lane,
                                     args.spill, (size_t)blockIdx.x * blockDim.x + threadIdx.x,
                                     (size_t)gridDim.x * blockDim.x);
  uint64_t* const st = mem.st;
  const uint32_t A = t.A, NE = t.E, exists = t.exists, PMAX = args.p_max, NX = args.n_exp;
  const uint64_t* __restrict__ expected = args.expected;

  bool active = false, fresh = false;
  uint64_t sched = 0, hash = 0;
  uint64_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;   // candidate mask
  uint32_t idx = 0, cur = 0, n_pend = 0, count = 0, ignored = 0, flags = 0, rep = 0, skip = 0xFFFFFFFFu;
  Net net = {0, 0, 0};
  uint64_t tq = 0;
  uint32_t n_tq = 0;
  uint64_t b_next = 0, b_end = 0;
  bool exhausted = false;

#define IN_MASK(I) ((uint32_t)((((I) & 128u) ? (((I) & 64u) ? m3 : m2) : (((I) & 64u) ? m1 : m0)) >> ((I) & 63u)) & 1u)
#define TIMER_BIT(RCV, TYPE) (1u << ((RCV) * DEMI_MAX_TIMER_TYPES + (t.meta[(TYPE)] >> 8)))
#define PEND_APPEND(WORD)                                              \
  do {                                                                 \
    if (n_pend >= PMAX) { flags |= DEMI_V_PENDING_OVF; }               \
    else { pend_store(mem, n_pend, (WORD)); n_pend++; }                \
  } while (0)

  // cursor over the candidate's non-Send, non-WaitQuiescence externals (subsequenceIntersection :299-304)
  auto cur_skip = [&]() {
    while (cur < NE) {
      const uint32_t kind = (uint32_t)t.trace[cur] & 0xFF;
      if (IN_MASK(cur) && kind != DEMI_EV_SEND && kind != DEMI_EV_WAIT_QUIESCENCE) break;
      cur++;
    }
  };
  // STSScheduler.enqueue_timer = handle_timer: straight into messagesToSend (no parking)
  auto handle_timer = [&](uint32_t rcv, uint32_t type) {
    if (n_tq >= DEMI_TQ_CAP) { flags |= DEMI_V_QUEUE_OVF; return; }
    tq |= (uint64_t)((rcv << 5) | type) << (8 * n_tq);
    n_tq++;
  };

  for (;;) {
    // ---------------------------------------------------------- refill (same protocol as K1)
    {
      const uint64_t idle = __ballot(!active);
      if (idle != 0 && !exhausted) {
        const uint32_t want = (uint32_t)__popcll(idle);
        const uint64_t have = b_end - b_next;
        uint64_t got = 0;
        if (have < want) {
          if (lane == 0) got = atomicAdd(args.work_counter, 64ull);
          got = __shfl(got, 0);
        }
        if (!active) {
          const uint32_t rank = (uint32_t)__popcll(idle & ((1ULL << lane) - 1));
          const uint64_t my = (rank < have) ? (b_next + rank) : (got + (rank - have));
          if (my < args.n) { sched = my; active = true; fresh = true; }
        }
        if (have < want) { b_next = got + (want - have); b_end = got + 64; }
        else b_next += want;
        if (b_next >= args.n) exhausted = true;
      }
      if (__ballot(active) == 0) break;
    }

    uint32_t w = 0;
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
        net.inaccessible = exists; net.killed = 0; net.partitioned = 0;
        for (uint32_t a = 0; a < A; a++) st[a * 64] = t.init[a];
        idx = 0; cur = 0; n_pend = 0; count = 0; ignored = 0; flags = 0; rep = 0; tq = 0; n_tq = 0;
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
          if (kind == DEMI_REC_SPAWN) { net.inaccessible &= ~(1u << a); net.killed &= ~(1u << a); }
          else if (kind == DEMI_REC_KILL) { net.killed |= 1u << a; net.inaccessible |= 1u << a; }
          else if (kind == DEMI_REC_PARTITION) net.partitioned |= 1ULL << (a * 8 + b);
          else net.partitioned &= ~(1ULL << (a * 8 + b));
        } else if (kind == DEMI_REC_MSG_SEND) {
          // external MsgSend -> enqueue_message (:509-511) unless its Send was pruned
          if (IN_MASK(ext) && ((exists >> b) & 1)) {
            PEND_APPEND(msg_word((uint32_t)(e >> 24) & 0xFF, DEMI_DEADLETTERS, b, (uint32_t)(e >> 32) & 0xFF,
                                 (uint32_t)(e >> 40) & 0xFF));
            if (args.kept && !(flags & DEMI_OVF_ANY)) args.kept[sched * NX + idx - 1] = 1;
          }
        } else {  // MSG_EVENT
          if (idx - 1 == skip) continue;               // the delivery this candidate removes (OneAtATimeRemoval.scala:57-124)
          if (ext != 255 && !IN_MASK(ext)) continue;   // pruned together with its Send (filterSends)
          const uint32_t want = msg_word((uint32_t)(e >> 24) & 0xFF, a, b, (uint32_t)(e >> 32) & 0xFF,
                                         (uint32_t)(e >> 40) & 0xFF);
          uint32_t k = 0;
          for (; k < n_pend; k++)
            if (pend_load(mem, k) == want) break;
          if (k == n_pend) { ignored++; continue; }     // "Ignoring message" (:528-529)
          pend_store(mem, k, pend_load(mem, n_pend - 1));
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
        if (((meta & 0xFF) == DEMI_MSG_TIMER) && (rep & (1u << (me * DEMI_MAX_TIMER_TYPES + (meta >> 8)))))
          handle_timer(me, type);
        if (flags & DEMI_OVF_ANY) { deliver = false; finish = true; }
      }
    }

    uint32_t nfx = 0;
    if (deliver) nfx = DEMI_VM_RUN(t, mem, w, flags);

    if (deliver) {
      const uint32_t me = w_dst(w);
      for (uint32_t k = 0; k < nfx && !(flags & DEMI_OVF_ANY); k++) {
        const uint32_t fx = mem.fxq[k * 64];
        const uint32_t op = fx & 31u, type = (fx >> 5) & 31u, target = (fx >> 10) & 15u, p0 = (fx >> 14) & 0xFFu,
                       p1 = (fx >> 22) & 0xFFu;
        if (op <= DEMI_OP_BCAST) {
          const bool bc = (op == DEMI_OP_BCAST);
          const uint32_t first = bc ? 0u : target, last = bc ? A : (target < A ? target + 1 : 0u);
          for (uint32_t r = first; r < last; r++) {
            if ((bc && r == me) || !((exists >> r) & 1)) continue;
            if (!crosses_partition(net, me, r)) PEND_APPEND(msg_word(type, me, r, p0, p1));
          }
        } else if (op == DEMI_OP_TCANCEL) {
          // notify_timer_cancel (:828-855): messagesToSend first, then the (deadLetters, rcv) queue
          rep &= ~TIMER_BIT(me, type);
          const uint32_t want = (me << 5) | type;
          bool found = false;
          for (uint32_t q = 0; q < n_tq; q++) {
            if (((uint32_t)(tq >> (8 * q)) & 0xFF) == want) {
              const uint64_t lowm = (q == 0) ? 0ull : (~0ull >> (64 - 8 * q));
              tq = (tq & lowm) | ((tq >> 8) & ~lowm);
              n_tq--; found = true; break;
            }
          }
          if (!found) {
            const uint32_t wantw = msg_word(type, DEMI_DEADLETTERS, me, 0, 0);
            for (uint32_t q = 0; q < n_pend; q++) {
              if (pend_load(mem, q) == wantw) {
                pend_store(mem, q, pend_load(mem, n_pend - 1));
                n_pend--; break;
              }
            }
          }
        } else {
          const uint32_t bit = TIMER_BIT(me, type);
          if (!(rep & bit)) {
            if (op == DEMI_OP_TREP) rep |= bit;
            handle_timer(me, type);
          }
        }
      }
      // schedule_new_message starts with send_external_messages (:655): timers become pending now,
      // unless the receiver is inaccessible (crosses_partition(deadLetters, rcv))
      for (uint32_t k = 0; k < n_tq && !(flags & DEMI_OVF_ANY); k++) {
        const uint32_t bt = (uint32_t)(tq >> (8 * k)) & 0xFF, rcv = bt >> 5, type = bt & 31;
        if (!((net.inaccessible >> rcv) & 1)) PEND_APPEND(msg_word(type, DEMI_DEADLETTERS, rcv, 0, 0));
      }
      tq = 0; n_tq = 0;
      if (flags & DEMI_OVF_ANY) finish = true;
    }

    if (active && finish) {
      // the invariant on the final state; verdict = fingerprint.matches(target) (:278-300)
      uint32_t viol = 0;
      if (!(flags & DEMI_OVF_ANY)) {
        const uint32_t fp = invariant_code(args.model, st, exists, A, t.inv_kind, t.inv_fa, t.inv_va, t.inv_fb);
        if (fp && (((fp ^ args.looking_for) & t.fp_mask) == 0)) viol = args.looking_for;
      }
      for (uint32_t a = 0; a < A; a++) hash_step(hash, st[a * 64]);
      uint4 v;
      if (flags & DEMI_OVF_ANY) {
        v.x = flags & DEMI_OVF_ANY; v.y = 0; v.z = 0; v.w = 0;
      } else {
        v.x = (viol ? DEMI_V_VIOLATION : 0u) | (ignored ? DEMI_V_DIVERGED : 0u) | ((count & 0xFFFF) << 16);
        v.y = viol; v.z = (uint32_t)hash; v.w = (uint32_t)(hash >> 32);
      }
      *reinterpret_cast<uint4*>(&args.out[sched]) = v;
      active = false;
    }
  }
#undef IN_MASK
#undef TIMER_BIT
#undef PEND_APPEND
}

}  // namespace demi
