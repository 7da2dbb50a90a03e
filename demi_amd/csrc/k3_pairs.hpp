// k3_pairs.hpp — dpor()'s bookkeeping on the device: the ExploredTacker (AuxilaryTypes.scala:209-246) as a
// device-resident hash table over pairs of node keys (one 64-byte entry per unordered pair, a side per orientation), and the enqueue decision of dpor() (:1068-1070, 1134) and
// getNext()'s skip (:1153-1157) taken there, so that of the ~10^3 racing pairs of an interleaving only the backtrack
// points that can still be dequeued live ever leave the GPU.
//
// What is decided here, per round (one launch of k3_dpor = the interleavings popped together):
//   mark    every backtrack point popped for this round: setExplored(branch, (later, earlier)) (:1170-1172);
//   insert  every racing pair (earlier, later) of the round: setExplored(branch, (earlier, later)) (:1068-1070); a pair that
//           becomes explored while backtrack points that flip INTO it are queued is reported to the host ("kill"), whose
//           queue then skips them exactly as getNext() would; and every pair proposes itself as this round's candidate for
//           its flipped pair (later, earlier): highest branch first, then creation order - DefaultBacktrackOrdering's
//           dequeue order, so the candidate is the one point of the round that getNext() would reach first;
//   decide  a pair's backtrack point is emitted iff its flipped pair is unexplored after the round (else getNext() would
//           skip it: the explored set only grows and nothing is dequeued during a round), it is the round's candidate
//           for that flipped pair (else the candidate is dequeued before it and explores the pair), and no point of an
//           earlier round with the same flipped pair and an equal or higher branch is queued (that one is dequeued first).
// Every point dropped here is one the reference would enqueue and later skip; the points it dequeues live, and their
// order, are unchanged (tests: the exploration equals the host-side bookkeeping's, round by round).
#pragma once

#include "demi_device.hpp"
#include "dpor_types.hpp"

namespace demi {

struct PairEntry {           // 64 bytes, one per UNORDERED pair of node keys: everything dpor() does with a racing pair touches
                             // the ordered pair and its flip, so both live in one line.  Side 0 is (lo, hi), side 1 is (hi, lo).
  unsigned long long lo, hi; // lo < hi; lo == 0 = empty slot (node keys are FNV chains, never 0)
  unsigned long long cand[2];   // per side: this round's candidate for the points flipping INTO it: round | branch + 1 | ~ordinal
  uint32_t state[2];         // per side: bit 31 explored; bits 0..8: 1 + highest branch of a queued point flipping into it
  unsigned long long pop[2]; // per side: the dequeue (k3_queue.hpp) that last saw an unexplored candidate with this flipped pair, and
                             // the lowest such candidate: dequeue number << 32 | ~candidate index (atomicMax)
  uint32_t pad[2];
};
static_assert(sizeof(PairEntry) == 64, "one explored-pair entry is one 64-byte line");
constexpr uint32_t PE_EXPLORED = 0x80000000u, PE_QMASK = 0x1FFu;

__device__ __forceinline__ uint64_t pair_hash(uint64_t a, uint64_t b) {
  uint64_t h = (a * 0x9E3779B97F4A7C15ULL) ^ (b * 0xC2B2AE3D27D4EB4FULL) ^ (a >> 29);
  h ^= h >> 32; h *= 0xD6E8FEB86659FD93ULL; h ^= h >> 32;
  return h;
}

// find or insert the entry of {a, b}; returns slot * 2 + side of the ORDERED pair (a, b) - `ref ^ 1` is its flip - or
// 0xFFFFFFFF when the table is full.  An inserter claims the slot by CAS on `lo` and publishes `hi` right after, in the
// same iteration; a reader that sees the claim but not yet `hi` simply repeats the iteration (no spinning inside a
// divergent branch: the publishing lane may be in the same wave).
__device__ inline uint32_t pair_slot(PairEntry* tab, uint32_t mask, uint64_t a, uint64_t b) {
  const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
  const uint32_t side = a < b ? 0u : 1u;
  uint32_t i = (uint32_t)pair_hash(lo, hi) & mask;
  for (uint32_t probes = 0; probes < 4096;) {
    PairEntry* e = tab + i;
    // Most lookups of a round find a pair that an earlier interleaving inserted: an ordinary (cacheable) load of the key
    // answers those from the CU's own cache.  Keys are written once and never change, so a match is final; anything else
    // (empty, half-published, another key - possibly a stale line) takes the coherent path below.
    {
      const ulonglong2 k = *reinterpret_cast<const ulonglong2*>(&e->lo);
      if (k.x == lo && k.y == hi) return i * 2 + side;
    }
    unsigned long long el = __atomic_load_n(&e->lo, __ATOMIC_RELAXED);
    if (el == 0) {
      const unsigned long long seen = atomicCAS(&e->lo, 0ull, (unsigned long long)lo);
      if (seen == 0) { __atomic_store_n(&e->hi, (unsigned long long)hi, __ATOMIC_RELEASE); return i * 2 + side; }
      el = seen;
    }
    if (el == lo) {
      const unsigned long long eh = __atomic_load_n(&e->hi, __ATOMIC_ACQUIRE);
      if (eh == hi) return i * 2 + side;
      if (eh == 0) { probes++; continue; }      // claimed, key not published yet: look again
    }
    i = (i + 1) & mask;
    probes++;
  }
  return 0xFFFFFFFFu;
}

struct K3PairArgs {
  PairEntry* table;
  uint32_t mask;                       // table size - 1
  const demi_dpor_trace_entry* arena;  // [ids][DEMI_DPOR_MAX_TRACE]
  const DporItem* items;               // [n] the round's dequeued points
  uint32_t base_id;                    // arena id of item 0's interleaving
  uint32_t n;
  const demi_dpor_pair* pairs;         // [n][max_pairs] racing pairs of the round's interleavings
  const uint32_t* n_pairs;             // [n]
  uint32_t max_pairs;
  uint32_t round;                      // 1, 2, ...: stamps the candidates
  uint32_t* pair_slot_of;              // [n][max_pairs] slot of the flipped pair (insert -> decide)
  DporPoint* points; uint32_t points_cap;
  DporKill* kills; uint32_t kills_cap;
  unsigned long long* counters;        // [0] points, [1] kills, [2] table-full errors, [3] pair records
  // multi-GPU rounds: every rank runs a contiguous block of the round's items and owns the table entries of the pairs
  // with dpor_pair_owner(...) == rank; racing pairs travel as records (all-gathered), see k3_pairs_records
  uint32_t rank, world;
  uint32_t item_base;                  // index in the round of this rank's first item (ordinals are global)
  DporPairRec* recs;                   // k3_pairs_records out: [recs_cap]; insert_rec / decide_rec in: [world][recs_stride]
  uint32_t recs_cap, recs_stride;
  const unsigned long long* rec_counts;   // [world] valid records per rank segment
  // the parent filter of insert (see k3_pairs_insert); complete == nullptr: every reported pair is probed
  const uint32_t* arena_len;           // [ids]
  uint8_t* complete;                   // [ids]: invariant (I) of dpor_host.hpp's ParentFilter holds for that interleaving
  const demi_verdict* verdicts;        // [n] the round's verdicts (DEMI_V_PAIRS_OVF)
  // the device-resident backtrack queue (k3_queue.hpp)
  // insert -> decide: the pairs that reached the table, compacted per interleaving (any order): surv_k[it][j] = the pair's index,
  // pair_slot_of[it][j] = the slot of its flipped pair, j < n_surv[it]
  uint16_t* surv_k;                    // [n][max_pairs]
  uint32_t* n_surv;                    // [n]
  unsigned long long* keep_bits;       // [n][max_pairs / 64]: decide's verdict per racing pair (bit = its backtrack point is emitted)
  uint32_t* item_points;               // [n]: points emitted per interleaving
  // diagnostic (DEMI_K3_INSERT_PROBE): a launch of k3_pairs_insert that stops after the index build (1), after the filter (2) or after
  // the survivors' table loads (3) and changes nothing - the host times it beside the real launch of the same round
  uint32_t dry;
  unsigned long long* dbg;             // diagnostic: [0] survivors, [1] not at home, [2] atomicOr, [3] atomicMax (null: not counted)
};

__device__ __forceinline__ unsigned long long cand_pack(uint32_t round, uint32_t branch, unsigned long long ordinal) {
  // round < 2^24, branch + 1 < 2^9, ordinal < 2^31 (the host keeps n * max_pairs below that)
  return ((unsigned long long)round << 40) | ((unsigned long long)(branch + 1) << 31) | (0x7FFFFFFFull - ordinal);
}

// occupied entries of the table (diagnostics, and what sizes it: DESIGN section 4 K3)
__global__ __launch_bounds__(256) void k3_pairs_count(const PairEntry* tab, uint32_t mask, unsigned long long* out) {
  unsigned long long n = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i <= mask; i += (size_t)gridDim.x * blockDim.x) n += tab[i].lo != 0;
  if (n) atomicAdd(out, n);
}

// mark: the dequeued points' flipped pairs are explored (getNext, :1170-1172)
__global__ __launch_bounds__(256) void k3_pairs_mark(const K3PairArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const DporItem it = a.items[i];
  if (it.src == 0xFFFFFFFFu) return;
  const demi_dpor_trace_entry* T = a.arena + (size_t)it.src * DEMI_DPOR_MAX_TRACE;
  if (a.world > 1 && dpor_pair_owner(T[it.later].key, T[it.earlier].key, a.world) != a.rank) return;   // another rank's entry
  const uint32_t s = pair_slot(a.table, a.mask, T[it.later].key, T[it.earlier].key);
  if (s == 0xFFFFFFFFu) { atomicAdd(&a.counters[2], 1ull); return; }
  atomicOr(&a.table[s >> 1].state[s & 1], PE_EXPLORED);
}

// insert: one workgroup per finished interleaving, its threads stride over the racing pairs.
//
// The parent filter (round 5).  Thousands of sibling interleavings of a round report the SAME racing pairs - same node keys -
// and so did the interleaving they all descend from: config 5 probed the table 6 x 10^8 times for 5 x 10^6 distinct pairs.  What
// the PARENT of an interleaving C provably applied already is dropped here before it costs a probe: dpor_host.hpp's ParentFilter
// rule (both events also occur in the parent P - the interleaving whose racing pair produced C's next trace -, with keys unique
// on both sides and equal quiescent periods, in the same order, and the branch event sits at least as deep in P), evaluated
// against P's trace in the arena through a hash of its node keys in LDS (the same code as k3_ref_filter's rule (a)).  P's pairs
// were inserted in an earlier round - C's backtrack point was CREATED by that insert - with a branch at least as deep, so C's
// instance sets an explored bit that is set, proposes a candidate that cannot be emitted (the flipped pair is explored by now or
// carries a queued mark above C's branch) and, being unable to win, cannot displace an instance that could: a no-op.  The rule
// needs invariant (I) for P (every racing pair of P applied with a branch at least as deep): `complete`, kept per arena id -
// false once a pair list was truncated (DEMI_V_PAIRS_OVF) anywhere up the chain of parents; nothing is dropped then.
constexpr uint32_t PF_SLOTS = 512;        // 2 x DEMI_DPOR_MAX_TRACE (a trace has at most 256 distinct keys)
__device__ __forceinline__ uint32_t pf_key_hash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ULL) >> 55) & (PF_SLOTS - 1); }

// the LDS a workgroup needs for the filter (9.5 KB), and its construction: idx[i] = where event i of this interleaving sits in its
// parent (same key, unique on both sides, same quiescent period), or -1.  Returns whether the filter applies (uniform over the
// workgroup).  Written for any workgroup size (one wave per interleaving was measured: slower, see k3_pair_threads in demi_gpu.hip).
struct ParentIndex {
  unsigned long long pkey[DEMI_DPOR_MAX_TRACE], okey[DEMI_DPOR_MAX_TRACE];
  uint32_t pslot[PF_SLOTS], oslot[PF_SLOTS];      // 0xFFFFFFFF = empty, else an event index
  short idx[DEMI_DPOR_MAX_TRACE];
  uint8_t pdup[DEMI_DPOR_MAX_TRACE], odup[DEMI_DPOR_MAX_TRACE], pq[DEMI_DPOR_MAX_TRACE], oq[DEMI_DPOR_MAX_TRACE];
};
// it: the interleaving's index within a.pairs / a.n_pairs / a.verdicts (this rank's block); arena id a.base_id + it; its
// backtrack point a.items[a.item_base + it].  Also records `complete` for the interleaving (thread 0).
// The loads are issued as early and as unconditionally as their addresses allow - a workgroup's time is the chain of its
// dependent round trips to HBM: the item (1), then the parent's flag, its length and its keys together (2; whole rows are read,
// entries past a trace's length are never looked at), the own keys beside the item.
__device__ inline bool parent_index_build(const K3PairArgs& a, uint32_t it, uint32_t np, const demi_dpor_trace_entry* T, ParentIndex& S) {
  const uint32_t t = threadIdx.x, nt = blockDim.x;
  if (!a.complete) return false;
  const DporItem item = a.items[a.item_base + it];
  const uint32_t n_tr_raw = a.arena_len[a.base_id + it];
  for (uint32_t i = t; i < DEMI_DPOR_MAX_TRACE; i += nt) { const demi_dpor_trace_entry e = T[i]; S.okey[i] = e.key; S.oq[i] = e.qperiod; }
  for (uint32_t i = t; i < PF_SLOTS; i += nt) { S.pslot[i] = 0xFFFFFFFFu; S.oslot[i] = 0xFFFFFFFFu; }
  for (uint32_t i = t; i < DEMI_DPOR_MAX_TRACE; i += nt) { S.pdup[i] = 0; S.odup[i] = 0; S.idx[i] = -1; }
  const bool has_src = item.src != 0xFFFFFFFFu;
  uint32_t comp = 0, n_par_raw = 0;
  if (has_src) {
    const demi_dpor_trace_entry* TP = a.arena + (size_t)item.src * DEMI_DPOR_MAX_TRACE;
    comp = a.complete[item.src];
    n_par_raw = a.arena_len[item.src];
    if (np != 0)
      for (uint32_t i = t; i < DEMI_DPOR_MAX_TRACE; i += nt) { const demi_dpor_trace_entry e = TP[i]; S.pkey[i] = e.key; S.pq[i] = e.qperiod; }
  }
  const bool par = has_src && comp != 0;
  if (t == 0) a.complete[a.base_id + it] = (!(a.verdicts[it].flags & DEMI_V_PAIRS_OVF) && (!has_src || par)) ? 1 : 0;
  if (!par || np == 0) return false;
  const uint32_t n_tr = min(n_tr_raw, (uint32_t)DEMI_DPOR_MAX_TRACE), n_par = min(n_par_raw, (uint32_t)DEMI_DPOR_MAX_TRACE);
  __syncthreads();
  for (uint32_t i = t; i < n_par; i += nt) {       // equal keys mark each other as duplicates (collapsed siblings)
    const unsigned long long k = S.pkey[i];
    for (uint32_t h = pf_key_hash(k);; h = (h + 1) & (PF_SLOTS - 1)) {
      const uint32_t old = atomicCAS(&S.pslot[h], 0xFFFFFFFFu, i);
      if (old == 0xFFFFFFFFu) break;
      if (S.pkey[old] == k) { S.pdup[old] = 1; S.pdup[i] = 1; break; }
    }
  }
  for (uint32_t i = t; i < n_tr; i += nt) {
    const unsigned long long k = S.okey[i];
    for (uint32_t h = pf_key_hash(k);; h = (h + 1) & (PF_SLOTS - 1)) {
      const uint32_t old = atomicCAS(&S.oslot[h], 0xFFFFFFFFu, i);
      if (old == 0xFFFFFFFFu) break;
      if (S.okey[old] == k) { S.odup[old] = 1; S.odup[i] = 1; break; }
    }
  }
  __syncthreads();
  for (uint32_t i = t; i < n_tr; i += nt) {        // same key, unique on both sides, same quiescent period
    if (S.odup[i]) continue;
    const unsigned long long k = S.okey[i];
    for (uint32_t h = pf_key_hash(k);; h = (h + 1) & (PF_SLOTS - 1)) {
      const uint32_t j = S.pslot[h];
      if (j == 0xFFFFFFFFu) break;
      if (S.pkey[j] == k) { if (!S.pdup[j] && S.pq[j] == S.oq[i]) S.idx[i] = (short)j; break; }
    }
  }
  __syncthreads();
  return true;
}
__device__ __forceinline__ bool parent_applied(const ParentIndex& S, const demi_dpor_pair& p) {
  const int ie = S.idx[p.earlier], il = S.idx[p.later], ib = S.idx[p.branch];
  return ie >= 0 && il >= 0 && ie < il && ib >= (int)p.branch;
}

__global__ __launch_bounds__(256) void k3_pairs_insert(const K3PairArgs a) {
  __shared__ ParentIndex S;
  __shared__ uint16_t s_surv[4096];        // the pairs the filter leaves (their indices), in any order
  __shared__ uint32_t s_n;
  const uint32_t it = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  const demi_dpor_trace_entry* T = a.arena + (size_t)(a.base_id + it) * DEMI_DPOR_MAX_TRACE;
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  const uint32_t np = min(a.n_pairs[it], a.max_pairs);
  if (t == 0) s_n = 0;
  // (the first step's pairs are asked for before the index is built: their address depends on nothing)
  demi_dpor_pair first[8];
#pragma unroll
  for (uint32_t j = 0; j < 8; j++) { const uint32_t k = j * nt + t; if (k < a.max_pairs) first[j] = P[k]; }
  const bool par = parent_index_build(a, it, np, T, S);
  __syncthreads();
  if (a.dry == 1) return;
  uint32_t c_nh = 0, c_or = 0, c_max = 0;
  uint32_t done = 0;                        // survivors written so far (max_pairs may exceed the list: 4096 pairs at a time)
  for (uint32_t c0 = 0; c0 < np; c0 += 4096) {
  const uint32_t c1 = min(c0 + 4096u, np);
  // ---- the filter: four pairs per thread and step, their loads issued together (a step is one round trip to HBM, and the
  // steps of a thread wait for one another)
  if (par) {
    for (uint32_t base = c0; base < c1; base += 4 * nt) {
      demi_dpor_pair p[4];
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t k = base + j * nt + t;
        if (base == 0) p[j] = first[j]; else if (base == 4 * nt) p[j] = first[4 + j]; else if (k < c1) p[j] = P[k];
      }
#pragma unroll
      for (uint32_t j = 0; j < 4; j++) {
        const uint32_t k = base + j * nt + t;
        if (k < c1 && !parent_applied(S, p[j])) s_surv[atomicAdd(&s_n, 1u)] = (uint16_t)(k - c0);
      }
    }
  } else {
    for (uint32_t k = c0 + t; k < c1; k += nt) s_surv[k - c0] = (uint16_t)(k - c0);
    if (t == 0) s_n = c1 - c0;
  }
  __syncthreads();
  const uint32_t ns = s_n;
  if (a.dry == 2) return;
  // ---- the survivors' table entries
  for (uint32_t j = t; j < ns; j += nt) {
    const uint32_t k = c0 + s_surv[j];
    const demi_dpor_pair p = P[k];
    const uint64_t ke = par ? S.okey[p.earlier] : T[p.earlier].key, kl = par ? S.okey[p.later] : T[p.later].key;   // (the index holds the own keys)
    // The entry is usually there already (a sibling inserted it) and at its home position: key, candidates and states of that
    // position are read TOGETHER - three loads of one 64-byte line in flight instead of three round trips in a row - and used
    // when the key matches; anything else takes pair_slot()'s path.
    const uint64_t lo_ = ke < kl ? ke : kl, hi_ = ke < kl ? kl : ke;
    const uint32_t home = (uint32_t)pair_hash(lo_, hi_) & a.mask;
    const PairEntry* const eh = a.table + home;
    const ulonglong2 hk = *reinterpret_cast<const ulonglong2*>(&eh->lo);
    const ulonglong2 hc = *reinterpret_cast<const ulonglong2*>(&eh->cand[0]);
    const uint2 hs = *reinterpret_cast<const uint2*>(&eh->state[0]);
    const bool at_home = hk.x == lo_ && hk.y == hi_;
    if (a.dry == 3) { if (hk.x == 1 && hc.x == 1 && hs.x == 1) atomicAdd(&a.counters[2], 1ull); continue; }
    const uint32_t s1 = at_home ? home * 2 + (ke < kl ? 0u : 1u) : pair_slot(a.table, a.mask, ke, kl);   // (earlier, later); its flip is the other side
    const uint32_t s2 = s1 == 0xFFFFFFFFu ? s1 : (s1 ^ 1u);
    c_nh += !at_home;
    a.surv_k[(size_t)it * a.max_pairs + done + j] = (uint16_t)k;
    a.pair_slot_of[(size_t)it * a.max_pairs + done + j] = s2;
    if (s1 == 0xFFFFFFFFu) { atomicAdd(&a.counters[2], 1ull); continue; }
    // Thousands of interleavings of a round report the same few pairs, and a read-modify-write on one address is serialised
    // where it executes.  Both updates are monotone (the explored bit is only ever set, the candidate only ever raised), so
    // an ordinary load that already shows the result makes the atomic redundant; a stale line only means one atomic more.
    PairEntry* const e1 = a.table + (s1 >> 1);
    const uint32_t seen_state = at_home ? ((s1 & 1) ? hs.y : hs.x) : e1->state[s1 & 1];
    const unsigned long long seen_cand = at_home ? ((s2 & 1) ? hc.y : hc.x) : e1->cand[s2 & 1];
    if (!(seen_state & PE_EXPLORED)) {
      c_or++;
      const uint32_t old = atomicOr(&e1->state[s1 & 1], PE_EXPLORED);      // setExplored(branch, (earlier, later))
      if (a.kills && !(old & PE_EXPLORED) && (old & PE_QMASK)) {           // queued points flip into this pair: dead now
        const unsigned long long q = atomicAdd(&a.counters[1], 1ull);      // (only a HOST queue is told: the device queue reads the table)
        if (q < a.kills_cap) { DporKill kk; kk.a = ke; kk.b = kl; a.kills[q] = kk; }
      }
    }
    const unsigned long long mine = cand_pack(a.round, p.branch, (unsigned long long)it * a.max_pairs + k);
    c_max += seen_cand < mine;
    if (seen_cand < mine) atomicMax(&e1->cand[s2 & 1], mine);            // (s2 is the other side of the same entry)
  }
  done += ns;
  __syncthreads();
  if (t == 0) s_n = 0;
  __syncthreads();
  }
  if (a.dry) return;
  if (a.dbg) {
    if (c_nh) atomicAdd(&a.dbg[1], (unsigned long long)c_nh);
    if (c_or) atomicAdd(&a.dbg[2], (unsigned long long)c_or);
    if (c_max) atomicAdd(&a.dbg[3], (unsigned long long)c_max);
    if (t == 0) atomicAdd(&a.dbg[0], (unsigned long long)done);
  }
  // (The statistics of the device-queue rounds - pairs reported, pairs the filter dropped - are sums over n_pairs / n_surv that
  // k3_q_scan takes while it scans the round anyway.  They used to be two atomicAdds per workgroup on two fixed addresses: 131 072
  // same-address atomics of a 65 536-wide round, executed one after the other wherever atomics execute - 1.5 of the launch's
  // 1.9 ms, profiles/r06_insert_probe.txt.)
  if (t == 0) a.n_surv[it] = done;
}

// decide: which pairs' backtrack points can still be dequeued live
__global__ __launch_bounds__(256) void k3_pairs_decide(const K3PairArgs a) {
  const uint32_t it = blockIdx.x;
  const demi_dpor_trace_entry* T = a.arena + (size_t)(a.base_id + it) * DEMI_DPOR_MAX_TRACE;
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  const uint32_t ns = a.n_surv[it];
  for (uint32_t j = threadIdx.x; j < ns; j += blockDim.x) {
    const uint32_t s2 = a.pair_slot_of[(size_t)it * a.max_pairs + j];
    if (s2 == 0xFFFFFFFFu) continue;
    const uint32_t k = a.surv_k[(size_t)it * a.max_pairs + j];
    const demi_dpor_pair p = P[k];
    PairEntry* e = a.table + (s2 >> 1);
    const uint32_t sd = s2 & 1;
    const unsigned long long ord = (unsigned long long)it * a.max_pairs + k;
    const uint32_t st = e->state[sd];
    if (st & PE_EXPLORED) continue;                                        // getNext would skip it (:1153-1157)
    if (e->cand[sd] != cand_pack(a.round, p.branch, ord)) continue;        // another point of this round is dequeued first
    if ((st & PE_QMASK) > p.branch) continue;                              // so is a queued point of an earlier round
    e->state[sd] = (st & ~PE_QMASK) | ((uint32_t)p.branch + 1);            // (the only writer of this side this round)
    const unsigned long long q = atomicAdd(&a.counters[0], 1ull);
    if (q < a.points_cap) {
      DporPoint o;
      o.flip_a = T[p.later].key; o.flip_b = T[p.earlier].key; o.ordinal = ord; o.src = a.base_id + it;
      o.branch = p.branch; o.later = p.later; o.earlier = p.earlier; o.pad = 0; o.pad2 = 0;
      a.points[q] = o;
    }
  }
}

// decide for the device-resident queue (k3_queue.hpp): the same verdict per racing pair, left as one bit per pair (and the
// count per interleaving) instead of a point pushed through an atomic counter - k3_q_scan / k3_q_emit then write the round's
// points in creation order.  max_pairs <= 4096.
__global__ __launch_bounds__(256) void k3_pairs_decide_q(const K3PairArgs a) {
  __shared__ unsigned long long s_keep[64];
  __shared__ uint32_t s_cnt;
  const uint32_t it = blockIdx.x, t = threadIdx.x;
  const uint32_t words = (a.max_pairs + 63) / 64;
  if (t < 64) s_keep[t] = 0;
  if (t == 0) s_cnt = 0;
  __syncthreads();
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  const uint32_t ns = a.n_surv[it];
  for (uint32_t j = t; j < ns; j += blockDim.x) {
    const uint32_t s2 = a.pair_slot_of[(size_t)it * a.max_pairs + j];
    if (s2 == 0xFFFFFFFFu) continue;
    const uint32_t k = a.surv_k[(size_t)it * a.max_pairs + j];
    const demi_dpor_pair p = P[k];
    PairEntry* e = a.table + (s2 >> 1);
    const uint32_t sd = s2 & 1;
    const uint32_t st = e->state[sd];
    if (st & PE_EXPLORED) continue;                                        // getNext would skip it (:1153-1157)
    if (e->cand[sd] != cand_pack(a.round, p.branch, (unsigned long long)it * a.max_pairs + k)) continue;
    if ((st & PE_QMASK) > p.branch) continue;                              // a queued point of an earlier round is dequeued first
    e->state[sd] = (st & ~PE_QMASK) | ((uint32_t)p.branch + 1);            // (the only writer of this side this round)
    atomicOr(&s_keep[k >> 6], 1ull << (k & 63));
    atomicAdd(&s_cnt, 1u);
  }
  __syncthreads();
  if (t < words) a.keep_bits[(size_t)it * words + t] = s_keep[t];
  if (t == 0) a.item_points[it] = s_cnt;
}

}  // namespace demi

namespace demi {

// ------------------------------------------------------------------ multi-GPU rounds: racing pairs as records
// this rank's racing pairs of the round, compacted (any order: the ordinal travels with the record)
// (what the interleaving's parent provably applied is not a record at all: the parent filter of k3_pairs_insert, here before
// the pairs cost an all-gather - every rank holds every trace and every `complete` flag, both gathered in place)
__global__ __launch_bounds__(256) void k3_pairs_records(const K3PairArgs a) {
  __shared__ ParentIndex S;
  const uint32_t it = blockIdx.x;
  const demi_dpor_trace_entry* T = a.arena + (size_t)(a.base_id + it) * DEMI_DPOR_MAX_TRACE;
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  const uint32_t np = a.n_pairs[it];
  const bool par = parent_index_build(a, it, np, T, S);
  for (uint32_t k = threadIdx.x; k < np; k += blockDim.x) {
    const demi_dpor_pair p = P[k];
    if (par && parent_applied(S, p)) continue;
    const unsigned long long q = atomicAdd(&a.counters[3], 1ull);
    if (q >= a.recs_cap) continue;
    DporPairRec r;
    r.ke = T[p.earlier].key; r.kl = T[p.later].key;
    r.ordinal = (a.item_base + it) * a.max_pairs + k; r.src = a.base_id + it;
    r.branch = p.branch; r.later = p.later; r.earlier = p.earlier; r.pad = 0; r.pad2 = 0;
    a.recs[q] = r;
  }
}

// insert / decide over the gathered records [world][recs_stride], each rank only for the pairs it owns
__global__ __launch_bounds__(256) void k3_pairs_insert_rec(const K3PairArgs a) {
  const uint64_t total = (uint64_t)a.world * a.recs_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    a.pair_slot_of[i] = 0xFFFFFFFFu;
    if ((i % a.recs_stride) >= a.rec_counts[i / a.recs_stride]) continue;
    const DporPairRec r = a.recs[i];
    if (dpor_pair_owner(r.ke, r.kl, a.world) != a.rank) continue;
    const uint32_t s1 = pair_slot(a.table, a.mask, r.ke, r.kl);
    if (s1 == 0xFFFFFFFFu) { atomicAdd(&a.counters[2], 1ull); continue; }
    const uint32_t s2 = s1 ^ 1u;
    a.pair_slot_of[i] = s2;
    PairEntry* const e1 = a.table + (s1 >> 1);                             // (ordinary loads first: see k3_pairs_insert)
    if (!(e1->state[s1 & 1] & PE_EXPLORED)) {
      const uint32_t old = atomicOr(&e1->state[s1 & 1], PE_EXPLORED);
      if (!(old & PE_EXPLORED) && (old & PE_QMASK)) {
        const unsigned long long q = atomicAdd(&a.counters[1], 1ull);
        if (q < a.kills_cap) { DporKill kk; kk.a = r.ke; kk.b = r.kl; a.kills[q] = kk; }
      }
    }
    const unsigned long long mine = cand_pack(a.round, r.branch, r.ordinal);
    if (e1->cand[s2 & 1] < mine) atomicMax(&e1->cand[s2 & 1], mine);
  }
}

__global__ __launch_bounds__(256) void k3_pairs_decide_rec(const K3PairArgs a) {
  const uint64_t total = (uint64_t)a.world * a.recs_stride;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t s2 = a.pair_slot_of[i];
    if (s2 == 0xFFFFFFFFu) continue;
    const DporPairRec r = a.recs[i];
    PairEntry* e = a.table + (s2 >> 1);
    const uint32_t sd = s2 & 1;
    const uint32_t st = e->state[sd];
    if (st & PE_EXPLORED) continue;
    if (e->cand[sd] != cand_pack(a.round, r.branch, r.ordinal)) continue;
    if ((st & PE_QMASK) > r.branch) continue;
    e->state[sd] = (st & ~PE_QMASK) | ((uint32_t)r.branch + 1);
    const unsigned long long q = atomicAdd(&a.counters[0], 1ull);
    if (q < a.points_cap) {
      DporPoint o;
      o.flip_a = r.kl; o.flip_b = r.ke; o.ordinal = r.ordinal; o.src = r.src;
      o.branch = r.branch; o.later = r.later; o.earlier = r.earlier; o.pad = 0; o.pad2 = 0;
      a.points[q] = o;
    }
  }
}

}  // namespace demi

namespace demi {

// ------------------------------------------------------------------ REFERENCE order: the commit filter
// (dpor_host.hpp explore_reference_resident.)  The commit absorbs an interleaving's racing pairs one interleaving at a time on
// the host; of the ~10^3 pairs an interleaving reports only those that can still change the commit's state need to get
// there.  One workgroup per interleaving drops, in pair order,
//   (a) what its PARENT provably applied - ParentFilter's rule (dpor_host.hpp): both events also occur in the parent, with
//       unique keys on both sides and equal quiescent periods, in the same order, and the branch event sits at least as deep
//       there - evaluated against the parent's trace in the arena through a hash of its node keys in LDS;
//   (b) what is a no-op under the device's copy of the commit's explored-pair table (a SNAPSHOT: the host sends the entries
//       that changed before every launch): (ke, kl) explored and its flip explored or marked above the pair's branch.  The
//       state only grows, so a no-op under an older state is a no-op when the commit reaches the pair.
// The survivors are written as 24-byte records, contiguous per interleaving and in pair order.
struct RefRecDev { unsigned long long ke, kl; uint8_t branch, later, earlier, pad; uint32_t pad2; };        // = demi_host::RefRec
struct RefDeltaDev { unsigned long long lo, hi; uint32_t state[2]; };                                       // = demi_host::RefDelta

struct K3RefArgs {
  PairEntry* real_table; uint32_t real_mask;          // the commit's table as of the last launch
  const RefDeltaDev* deltas; uint32_t n_deltas;
  const demi_dpor_trace_entry* arena;
  const uint32_t* arena_len;
  const DporItem* items; const uint8_t* use_parent;   // [n]
  uint32_t base_id, n;
  const demi_dpor_pair* pairs; const uint32_t* n_pairs; uint32_t max_pairs;
  RefRecDev* recs; unsigned long long recs_cap;
  unsigned long long* rec_off; uint32_t* rec_cnt;     // [n]: where interleaving i's records start (rec_base + its offset in recs) / how many
  unsigned long long rec_base;
  unsigned long long* counters;                       // [0] records wanted (may exceed recs_cap: the host grows and re-runs), [1] table full
};

// A reference-order launch's fixed-size results - the verdicts, the round's and the filter's counters, the record counts - written
// straight into the launch's pinned host area by ONE kernel behind the filter (they used to be four copies, each a dispatch of its
// own on a stream where every dispatch waits for the one before it).
struct K3RefResults {
  const demi_verdict* verdicts; const unsigned long long* round_counters; const unsigned long long* filter_counters; const uint32_t* rec_cnt;
  uint32_t n;
  demi_verdict* h_verdicts; unsigned long long* h_counters /* [0..3] the round's, [4..5] the filter's */; uint32_t* h_rec_cnt;
};
__global__ __launch_bounds__(256) void k3_ref_results(const K3RefResults a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < a.n) {
    *reinterpret_cast<uint4*>(&a.h_verdicts[i]) = *reinterpret_cast<const uint4*>(&a.verdicts[i]);
    a.h_rec_cnt[i] = a.rec_cnt[i];
  }
  if (i < 4) a.h_counters[i] = a.round_counters[i];
  else if (i < 6) a.h_counters[i] = a.filter_counters[i - 4];
}

// A state of the commit's table only grows AS A NUMBER: the explored bit is the top bit and is never cleared, the queued mark
// below it is only replaced by a higher one (dpor_host.hpp RefBook::absorb).  So a change is merged with a maximum, and the
// order in which batches of changes arrive - a launch's on its stream, a record fetch's on another - does not matter: an
// entry never goes back to an older state.
__device__ __forceinline__ void ref_state_merge(PairEntry* e, const RefDeltaDev& d) {
  atomicMax(&e->state[0], d.state[0]);
  atomicMax(&e->state[1], d.state[1]);
}

// the commit's table: entries are written by k3_ref_apply and by k3_ref_fetch's own blocks, and read by both filters
__global__ __launch_bounds__(256) void k3_ref_apply(const K3RefArgs a) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_deltas) return;
  const RefDeltaDev d = a.deltas[i];
  const uint32_t s = pair_slot(a.real_table, a.real_mask, d.lo, d.hi);        // lo < hi: side 0
  if (s == 0xFFFFFFFFu) { atomicAdd(&a.counters[1], 1ull); return; }
  ref_state_merge(a.real_table + (s >> 1), d);
}

// read-only lookup: the states of (a, b) and of its flip, 0 / 0 when the pair has no entry
__device__ inline void pair_states(const PairEntry* tab, uint32_t mask, uint64_t a, uint64_t b, uint32_t& fwd, uint32_t& rev) {
  const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
  const uint32_t side = a < b ? 0u : 1u;
  fwd = 0; rev = 0;
  uint32_t i = (uint32_t)pair_hash(lo, hi) & mask;
  for (uint32_t probes = 0; probes < 4096; probes++, i = (i + 1) & mask) {
    const PairEntry* e = tab + i;
    const unsigned long long el = e->lo;
    if (el == 0) return;
    if (el == lo && e->hi == hi) { fwd = e->state[side]; rev = e->state[side ^ 1u]; return; }
  }
}

constexpr uint32_t REF_SLOTS = 1024;       // > 2 x DEMI_DPOR_MAX_TRACE

__device__ __forceinline__ uint32_t ref_key_hash(uint64_t k) { return (uint32_t)((k * 0x9E3779B97F4A7C15ULL) >> 54) & (REF_SLOTS - 1); }

__global__ __launch_bounds__(256) void k3_ref_filter(const K3RefArgs a) {
  __shared__ unsigned long long s_pkey[DEMI_DPOR_MAX_TRACE], s_okey[DEMI_DPOR_MAX_TRACE];
  __shared__ uint32_t s_pslot[REF_SLOTS], s_oslot[REF_SLOTS];      // 0xFFFFFFFF = empty, else an event index
  __shared__ uint32_t s_pdup[DEMI_DPOR_MAX_TRACE], s_odup[DEMI_DPOR_MAX_TRACE];
  __shared__ int s_idx[DEMI_DPOR_MAX_TRACE];
  __shared__ uint8_t s_pq[DEMI_DPOR_MAX_TRACE];
  __shared__ uint32_t s_keep[128], s_pre[129];                     // keep bits of up to 4096 pairs; exclusive prefix per word
  __shared__ unsigned long long s_off;
  const uint32_t it = blockIdx.x, t = threadIdx.x;
  const demi_dpor_trace_entry* T = a.arena + (size_t)(a.base_id + it) * DEMI_DPOR_MAX_TRACE;
  const uint32_t n_tr = min(a.arena_len[a.base_id + it], (uint32_t)DEMI_DPOR_MAX_TRACE);
  const DporItem item = a.items[it];
  const bool par = a.use_parent[it] != 0 && item.src != 0xFFFFFFFFu;
  const demi_dpor_trace_entry* TP = par ? a.arena + (size_t)item.src * DEMI_DPOR_MAX_TRACE : nullptr;
  const uint32_t n_par = par ? min(a.arena_len[item.src], (uint32_t)DEMI_DPOR_MAX_TRACE) : 0u;
  for (uint32_t i = t; i < REF_SLOTS; i += blockDim.x) { s_pslot[i] = 0xFFFFFFFFu; s_oslot[i] = 0xFFFFFFFFu; }
  for (uint32_t i = t; i < 128; i += blockDim.x) s_keep[i] = 0;
  if (t < DEMI_DPOR_MAX_TRACE) {
    s_pdup[t] = 0; s_odup[t] = 0; s_idx[t] = -1;
    if (t < n_par) { s_pkey[t] = TP[t].key; s_pq[t] = TP[t].qperiod; }
    if (t < n_tr) s_okey[t] = T[t].key;
  }
  __syncthreads();
  // hashes of the parent's and of the own node keys; equal keys mark each other as duplicates (collapsed siblings)
  if (t < n_par) {
    const unsigned long long k = s_pkey[t];
    for (uint32_t h = ref_key_hash(k);; h = (h + 1) & (REF_SLOTS - 1)) {
      const uint32_t old = atomicCAS(&s_pslot[h], 0xFFFFFFFFu, t);
      if (old == 0xFFFFFFFFu) break;
      if (s_pkey[old] == k) { s_pdup[old] = 1; s_pdup[t] = 1; break; }
    }
  }
  if (par && t < n_tr) {
    const unsigned long long k = s_okey[t];
    for (uint32_t h = ref_key_hash(k);; h = (h + 1) & (REF_SLOTS - 1)) {
      const uint32_t old = atomicCAS(&s_oslot[h], 0xFFFFFFFFu, t);
      if (old == 0xFFFFFFFFu) break;
      if (s_okey[old] == k) { s_odup[old] = 1; s_odup[t] = 1; break; }
    }
  }
  __syncthreads();
  // where the own event t sits in the parent: same key, unique on both sides, same quiescent period
  if (par && t < n_tr && !s_odup[t]) {
    const unsigned long long k = s_okey[t];
    for (uint32_t h = ref_key_hash(k);; h = (h + 1) & (REF_SLOTS - 1)) {
      const uint32_t j = s_pslot[h];
      if (j == 0xFFFFFFFFu) break;
      if (s_pkey[j] == k) { if (!s_pdup[j] && s_pq[j] == T[t].qperiod) s_idx[t] = (int)j; break; }
    }
  }
  __syncthreads();
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  const uint32_t np = min(a.n_pairs[it], a.max_pairs);
  for (uint32_t k = t; k < np; k += blockDim.x) {
    const demi_dpor_pair p = P[k];
    const int ie = s_idx[p.earlier], il = s_idx[p.later], ib = s_idx[p.branch];
    if (par && ie >= 0 && il >= 0 && ie < il && ib >= (int)p.branch) continue;           // (a) the parent applied it
    uint32_t sf, sr;
    pair_states(a.real_table, a.real_mask, s_okey[p.earlier], s_okey[p.later], sf, sr);
    if ((sf & PE_EXPLORED) && ((sr & PE_EXPLORED) || (sr & PE_QMASK) > p.branch)) continue;   // (b) a no-op for the commit
    atomicOr(&s_keep[k >> 5], 1u << (k & 31));
  }
  __syncthreads();
  if (t == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < 128; w++) { s_pre[w] = tot; tot += (uint32_t)__popc(s_keep[w]); }
    s_pre[128] = tot;
    s_off = atomicAdd(&a.counters[0], (unsigned long long)tot);
    a.rec_off[it] = a.rec_base + s_off;
    a.rec_cnt[it] = tot;
  }
  __syncthreads();
  const unsigned long long off = s_off;
  if (off + s_pre[128] > a.recs_cap) return;              // the host sees counters[0] > recs_cap, grows the buffer and runs this again
  for (uint32_t k = t; k < np; k += blockDim.x) {
    const uint32_t w = s_keep[k >> 5];
    if (!((w >> (k & 31)) & 1u)) continue;
    const demi_dpor_pair p = P[k];
    RefRecDev r;
    r.ke = s_okey[p.earlier]; r.kl = s_okey[p.later];
    r.branch = p.branch; r.later = p.later; r.earlier = p.earlier; r.pad = 0; r.pad2 = 0;
    a.recs[off + s_pre[k >> 5] + (uint32_t)__popc(w & ((1u << (k & 31)) - 1u))] = r;
  }
}

// ------------------------------------------------------------------ REFERENCE order: the commit fetches records
// (dpor_host.hpp explore_reference_resident, dev.ref_fetch.)  The records k3_ref_filter kept stay in a pool on the device.
// When the commit reaches an interleaving it asks for that one's records and for those of the interleavings its queue will
// most likely hand out next; rule (b) is applied AGAIN, under the table as the commit has made it by now (the host sends the
// entries that changed first).  The same monotonicity argument holds, so what is dropped here is a no-op for the commit too -
// and by now that is most of what a wide launch had to keep.  One workgroup per interleaving; survivors in pair order.
// out / out_off / out_cnt are pinned host memory mapped into the device's address space: the launch writes its answer across
// PCIe itself, so a fetch costs two launches (k3_ref_apply in front of this one) and one synchronisation, no copies; a request
// of up to FETCH_INLINE_IDS interleavings travels in the kernel's arguments, a longer one is read from pinned memory.
// Measured and not kept (round 5, profiles/r05_call14_fetch_one_launch_ab.txt, r05_call15_reference_eager_rounds_ab.txt): the
// table's changes applied by this kernel's own blocks - with a barrier between applying and filtering the fetch's device time
// goes from 13.7 to 19.0 ms (the barrier's atomics cross the chip), without one to 12.1 ms but a block then filters under a
// table that lacks other blocks' changes and 11 % more records cross PCIe -; the answer's last block raising a flag the host
// polls instead of an event (13.7 against 13.5 ms).
constexpr uint32_t FETCH_INLINE_IDS = 160;

struct K3FetchArgs {
  const PairEntry* real_table; uint32_t real_mask;
  const uint32_t* ids; uint32_t m;                          // arena ids of the interleavings asked for (null: ids_inline)
  const RefRecDev* pool;                                    // every record kept so far
  const unsigned long long* pool_off; const uint32_t* pool_cnt;   // per arena id
  RefRecDev* out;                                           // room for the sum of pool_cnt[ids]
  unsigned long long* out_off; uint32_t* out_cnt;           // [m]
  unsigned long long* counter; unsigned long long counter_base;   // [0] running total of records handed out, [1] k3_ref_apply's
                                                                  // "table full" count (device memory)
  unsigned long long* table_full_out;                       // counter[1], passed on to the host
  uint32_t ids_inline[FETCH_INLINE_IDS];
};

__global__ __launch_bounds__(256) void k3_ref_fetch(const K3FetchArgs a) {
  __shared__ uint32_t s_keep[128], s_pre[129];
  __shared__ unsigned long long s_off;
  const uint32_t j = blockIdx.x, t = threadIdx.x;
  if (j == 0 && t == 0) *a.table_full_out = a.counter[1];
  const uint32_t id = a.ids ? a.ids[j] : a.ids_inline[j];
  const RefRecDev* R = a.pool + a.pool_off[id];
  const uint32_t n = min(a.pool_cnt[id], 4096u);
  const uint32_t words = (n + 31) >> 5;                     // (an interleaving holds ~10^2 records: a few words of keep bits)
  for (uint32_t i = t; i < words; i += blockDim.x) s_keep[i] = 0;
  __syncthreads();
  for (uint32_t k = t; k < n; k += blockDim.x) {
    uint32_t sf, sr;
    pair_states(a.real_table, a.real_mask, R[k].ke, R[k].kl, sf, sr);
    if ((sf & PE_EXPLORED) && ((sr & PE_EXPLORED) || (sr & PE_QMASK) > R[k].branch)) continue;     // a no-op for the commit by now
    atomicOr(&s_keep[k >> 5], 1u << (k & 31));
  }
  __syncthreads();
  if (t == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < words; w++) { s_pre[w] = tot; tot += (uint32_t)__popc(s_keep[w]); }
    s_off = atomicAdd(a.counter, (unsigned long long)tot) - a.counter_base;
    a.out_off[j] = s_off;
    a.out_cnt[j] = tot;
  }
  __syncthreads();
  const unsigned long long off = s_off;
  for (uint32_t k = t; k < n; k += blockDim.x) {
    const uint32_t w = s_keep[k >> 5];
    if (!((w >> (k & 31)) & 1u)) continue;
    a.out[off + s_pre[k >> 5] + (uint32_t)__popc(w & ((1u << (k & 31)) - 1u))] = R[k];
  }
}


}  // namespace demi
