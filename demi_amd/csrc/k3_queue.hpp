// k3_queue.hpp — DPORwHeuristics' backtrack queue on the device (round 5; single-rank ROUNDS order).
//
// Until round 4 the queue lived on the host (dpor_host.hpp explore_rounds_resident): per round every backtrack point that could
// still be dequeued (40 bytes) and every pair that became explored under a queued point (16 bytes) crossed PCIe, the host
// radix-sorted the points into 256 FIFO buckets and kept a hash set of dead flipped pairs - for config 5 that was half the wall
// time of the exploration (197 MB device-to-host, 66 synchronous rounds, the device idle while the host commits).  But the host's
// `dead` set is a copy of what the device's explored-pair table already knows - getNext()'s skip (DPORwHeuristics.scala:1153-1157)
// is `isExplored(flipped pair)`, and both the dequeue (:1170-1172) and dpor() (:1068-1070) mark pairs in that table - so the queue
// can stay where its inputs and its consumer are:
//
//   * decide (k3_pairs.hpp) no longer emits points through an atomic counter: it leaves one keep BIT per racing pair, and
//     k3_q_scan / k3_q_emit compact the kept pairs in creation order (interleaving by interleaving, pair by pair - what the host's
//     sort by ordinal produced);
//   * k3_q_hist / k3_q_offsets / k3_q_scatter are one stable counting-sort pass over the 8-bit branch index, deepest branch first
//     (DefaultBacktrackOrdering, BacktrackOrdering.scala:58-69), into the next free segment of a pool that holds every queued
//     point: a segment is 256 runs, one per branch, each in creation order.  The host only learns the 256 run lengths;
//   * the host keeps, per branch, the FIFO of runs (pool offset, remaining) - PriorityQueue order with ties in creation order is
//     "deepest branch first, oldest run first, run order" - and for a dequeue hands the device the next ranges of that order;
//   * k3_q_probe / k3_q_take are getNext() for a whole round: every candidate of the ranges looks its flipped pair up in the
//     explored-pair table; a candidate is LIVE iff the pair is not explored and no earlier candidate of this dequeue has the same
//     flipped pair (the first one marks it explored, :1170-1172, the others would be skipped); the first `want` live candidates
//     become the round's items and mark their pairs; what precedes the last one taken is consumed (the dead ones for good), the
//     rest stays queued.  The host gets two numbers back.
// Same dequeue order, same rounds, same verdict sequence as the host queue (tests: both paths on the same explorations).
#pragma once

#include "k3_pairs.hpp"

namespace demi {

struct QPoint {                          // a queued backtrack point, 24 bytes
  unsigned long long flip_a, flip_b;     // (later key, earlier key): the pair getNext() tests with isExplored
  uint32_t src;                          // arena id of the interleaving that found it
  uint8_t branch, later, earlier, pad;
};

struct QRange { unsigned long long start; uint32_t count, prefix; };   // pool[start .. start + count): candidates prefix .. prefix + count

constexpr uint32_t Q_TILE = 2048;        // elements a wave sorts (k3_q_hist / k3_q_scatter)

struct K3QueueArgs {
  // ---- emit + sort (after decide)
  const demi_dpor_trace_entry* arena;
  const demi_dpor_pair* pairs;           // [n][max_pairs]
  const uint32_t* n_pairs;               // [n]
  const uint32_t* n_surv;                // [n] pairs the parent filter left (k3_pairs_insert), or nullptr: k3_q_scan's statistics
  const unsigned long long* keep_bits;   // [n][max_pairs / 64]
  uint32_t* item_points;                 // [n] in: points per interleaving; k3_q_scan turns it into the exclusive prefix within its tile of 1 024
  uint32_t* tile_sum;                    // [ceil(n / 1024)] the tiles' totals, then (k3_q_scan_tiles) their offsets
  uint32_t n, max_pairs, base_id;
  QPoint* staging;                       // [staging_cap] the round's points in creation order
  uint32_t staging_cap;
  uint32_t* tile_hist;                   // [tiles][256]
  uint32_t* digit_start;                 // [256] where the run of branch d starts within the segment (deepest branch first)
  QPoint* pool;                          // the queue's pool; the round's segment starts at pool_fill
  unsigned long long pool_fill;
  // [0] points of the round, [1] candidates taken by the dequeue, [2] candidates consumed, [3] table-full errors; [4 .. 260) run lengths per branch
  unsigned long long* out;
  // ---- dequeue
  PairEntry* table; uint32_t mask;
  const QRange* ranges; uint32_t n_ranges;
  uint32_t n_cand;                       // candidates of this dequeue (the ranges' total)
  uint32_t want;                         // live candidates to take
  uint32_t round;                        // stamps the dequeue (1, 2, ...: every k3_q_probe launch has its own)
  uint32_t* cand_slot;                   // [n_cand] table slot * 2 + side of each candidate's flipped pair
  uint32_t* blk_live;                    // [(n_cand + 255) / 256] live candidates per workgroup of k3_q_live
  DporItem* items;                       // out: the taken candidates, in dequeue order
};

__device__ __forceinline__ unsigned long long q_stamp(uint32_t round, uint32_t j) {
  return ((unsigned long long)round << 32) | (unsigned long long)(0xFFFFFFFFu - j);      // max = this dequeue, lowest candidate index
}

// exclusive prefix of item_points[0 .. n), in two steps: k3_q_scan - one workgroup per tile of 1 024 interleavings - turns its tile's
// counts into prefixes WITHIN the tile (a scan within each wave, the sixteen wave sums through LDS) and leaves the tile's total;
// k3_q_scan_tiles - one wavefront - turns the totals into the tiles' offsets and writes the round's total to out[0].  k3_q_emit adds
// the two.  (One workgroup walking the tiles one after the other was 72 us of a 65 536-wide round.)  k3_q_scan also takes the round's
// statistics: out[1] += racing pairs reported, out[3] += pairs the parent filter dropped (n_surv != nullptr: the device-queue rounds).
constexpr uint32_t Q_SCAN_TILE = 1024;
__global__ __launch_bounds__(1024) void k3_q_scan(const K3QueueArgs a) {
  __shared__ uint32_t s_w[16];
  __shared__ unsigned long long s_rep, s_drop;
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  if (t == 0) { s_rep = 0; s_drop = 0; }
  const uint32_t i = blockIdx.x * Q_SCAN_TILE + t;
  const uint32_t v = i < a.n ? a.item_points[i] : 0u;
  unsigned long long rep = 0, drop = 0;
  if (a.n_surv && i < a.n) { const uint32_t np = min(a.n_pairs[i], a.max_pairs), ns = a.n_surv[i]; rep = np; drop = np > ns ? np - ns : 0u; }
  uint32_t incl = v;
  for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d); if (lane >= d) incl += x; }
  if (lane == 63) s_w[wave] = incl;
  __syncthreads();
  uint32_t before = 0, tile = 0;
  for (uint32_t w = 0; w < 16; w++) { const uint32_t c = s_w[w]; before += w < wave ? c : 0u; tile += c; }
  if (i < a.n) a.item_points[i] = before + incl - v;
  if (rep) atomicAdd(&s_rep, rep);
  if (drop) atomicAdd(&s_drop, drop);
  __syncthreads();
  if (t == 0) {
    a.tile_sum[blockIdx.x] = tile;
    if (a.n_surv) { if (s_rep) atomicAdd(&a.out[1], s_rep); if (s_drop) atomicAdd(&a.out[3], s_drop); }
  }
}
// tile_sum[0 .. tiles) -> exclusive offsets in place, the total to out[0].  One wavefront: lane l owns a contiguous chunk of tiles.
__global__ __launch_bounds__(64) void k3_q_scan_tiles(const K3QueueArgs a) {
  const uint32_t lane = threadIdx.x;
  const uint32_t tiles = (a.n + Q_SCAN_TILE - 1) / Q_SCAN_TILE;
  const uint32_t per = (tiles + 63) / 64;
  const uint32_t lo = min(lane * per, tiles), hi = min(lo + per, tiles);
  uint32_t sum = 0;
  for (uint32_t b = lo; b < hi; b++) sum += a.tile_sum[b];
  uint32_t incl = sum;
  for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(incl, d); if (lane >= d) incl += x; }
  uint32_t run = incl - sum;
  for (uint32_t b = lo; b < hi; b++) { const uint32_t v = a.tile_sum[b]; a.tile_sum[b] = run; run += v; }
  if (lane == 63) a.out[0] = incl;
}

// the kept pairs of interleaving blockIdx.x, in pair order, to staging[item_points[it] ..)
__global__ __launch_bounds__(256) void k3_q_emit(const K3QueueArgs a) {
  __shared__ uint32_t s_pre[65];
  const uint32_t it = blockIdx.x, t = threadIdx.x;
  const uint32_t words = (a.max_pairs + 63) / 64;
  const unsigned long long* kb = a.keep_bits + (size_t)it * words;
  const uint32_t np = a.n_pairs[it];
  const uint32_t nw = (np + 63) / 64;
  if (t == 0) {
    uint32_t tot = 0;
    for (uint32_t w = 0; w < nw; w++) { s_pre[w] = tot; tot += (uint32_t)__popcll(kb[w]); }
    s_pre[64] = tot;
  }
  __syncthreads();
  if (s_pre[64] == 0) return;
  const uint32_t off = a.item_points[it] + a.tile_sum[it / Q_SCAN_TILE];   // (within the tile + the tile's offset: k3_q_scan, k3_q_scan_tiles)
  if ((unsigned long long)off + s_pre[64] > a.staging_cap) return;          // (the host sees out[0] > staging_cap and reports it)
  const demi_dpor_trace_entry* T = a.arena + (size_t)(a.base_id + it) * DEMI_DPOR_MAX_TRACE;
  const demi_dpor_pair* P = a.pairs + (size_t)it * a.max_pairs;
  for (uint32_t k = t; k < np; k += blockDim.x) {
    const unsigned long long w = kb[k >> 6];
    if (!((w >> (k & 63)) & 1ull)) continue;
    const demi_dpor_pair p = P[k];
    QPoint q;
    q.flip_a = T[p.later].key; q.flip_b = T[p.earlier].key; q.src = a.base_id + it;
    q.branch = p.branch; q.later = p.later; q.earlier = p.earlier; q.pad = 0;
    a.staging[off + s_pre[k >> 6] + (uint32_t)__popcll(w & ((1ull << (k & 63)) - 1ull))] = q;
  }
}

// per tile of Q_TILE staged points: how many of each branch.  One wave per tile.
__global__ __launch_bounds__(64) void k3_q_hist(const K3QueueArgs a) {
  __shared__ uint32_t s_h[256];
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  const uint32_t total = (uint32_t)a.out[0];
  if (tile * Q_TILE >= total || total > a.staging_cap) return;       // (the grid is sized for the largest round)
  for (uint32_t i = lane; i < 256; i += 64) s_h[i] = 0;
  __syncthreads();
  const uint32_t lo = tile * Q_TILE, hi = min(lo + Q_TILE, total);
  for (uint32_t i = lo + lane; i < hi; i += 64) atomicAdd(&s_h[a.staging[i].branch], 1u);
  __syncthreads();
  for (uint32_t i = lane; i < 256; i += 64) a.tile_hist[(size_t)tile * 256 + i] = s_h[i];
}

// thread d: the tiles' counts of branch d -> exclusive prefix over the tiles, the run length to out[4 + d]; then the runs' starts
// within the segment, deepest branch first.  One workgroup of 256.
__global__ __launch_bounds__(256) void k3_q_offsets(const K3QueueArgs a) {
  __shared__ uint32_t s_tot[256];
  const uint32_t d = threadIdx.x;
  // (the host launches before it knows the round's point count; more points than the staging area holds: the host reports it)
  const uint32_t tiles = a.out[0] > a.staging_cap ? 0u : ((uint32_t)a.out[0] + Q_TILE - 1) / Q_TILE;
  uint32_t run = 0;
  for (uint32_t tl = 0; tl < tiles; tl++) {
    const uint32_t c = a.tile_hist[(size_t)tl * 256 + d];
    a.tile_hist[(size_t)tl * 256 + d] = run;
    run += c;
  }
  s_tot[d] = run;
  a.out[4 + d] = run;
  __syncthreads();
  if (d == 0) {
    uint32_t acc = 0;
    for (int b = 255; b >= 0; b--) { a.digit_start[b] = acc; acc += s_tot[b]; }
  }
}

// stable scatter of a tile into the pool segment: dest = run start of the branch + the tile's base within the run + the rank
// among the tile's earlier points of that branch.  One wave per tile, 64 points at a time: the lanes that hold the same
// branch find each other with eight ballots.
__global__ __launch_bounds__(64) void k3_q_scatter(const K3QueueArgs a) {
  __shared__ uint32_t s_run[256];
  const uint32_t tile = blockIdx.x, lane = threadIdx.x;
  const uint32_t total = (uint32_t)a.out[0];
  if (tile * Q_TILE >= total || total > a.staging_cap) return;
  for (uint32_t i = lane; i < 256; i += 64) s_run[i] = a.digit_start[i] + a.tile_hist[(size_t)tile * 256 + i];
  __syncthreads();
  const uint32_t lo = tile * Q_TILE, hi = min(lo + Q_TILE, total);
  for (uint32_t base = lo; base < hi; base += 64) {
    const uint32_t i = base + lane;
    const bool on = i < hi;
    QPoint q;
    if (on) q = a.staging[i];
    const uint32_t d = on ? q.branch : 0u;
    unsigned long long peers = __ballot(on);
    for (uint32_t b = 0; b < 8; b++) {
      const unsigned long long m = __ballot(on && ((d >> b) & 1u));
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    uint32_t dst = 0;
    if (on) dst = s_run[d] + rank;
    __syncthreads();                       // (one wave: orders the reads of s_run above against the updates below)
    if (on && rank == 0) s_run[d] += (uint32_t)__popcll(peers);
    __syncthreads();
    if (on) a.pool[a.pool_fill + dst] = q;
  }
}

// ------------------------------------------------------------------ getNext() for a round
// candidate j of the dequeue: the j-th point of the ranges
__device__ __forceinline__ unsigned long long q_cand_index(const QRange* r, uint32_t n_ranges, uint32_t j) {
  uint32_t lo = 0, hi = n_ranges;           // the last range with prefix <= j
  while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (r[mid].prefix <= j) lo = mid; else hi = mid; }
  return r[lo].start + (j - r[lo].prefix);
}

__global__ __launch_bounds__(256) void k3_q_probe(const K3QueueArgs a) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= a.n_cand) return;
  const QPoint q = a.pool[q_cand_index(a.ranges, a.n_ranges, j)];
  const uint32_t s = pair_slot(a.table, a.mask, q.flip_a, q.flip_b);
  a.cand_slot[j] = s;
  if (s == 0xFFFFFFFFu) { atomicAdd(&a.out[3], 1ull); return; }
  PairEntry* e = a.table + (s >> 1);
  if (__atomic_load_n(&e->state[s & 1], __ATOMIC_RELAXED) & PE_EXPLORED) return;
  atomicMax(&e->pop[s & 1], q_stamp(a.round, j));
}

// which candidates are LIVE: not explored, and the first of this dequeue with its flipped pair.  Two dependent reads of the
// explored-pair table per candidate: done by the whole grid, the verdict left in bit 31 of cand_slot (slots are below 2^28), and
// the live candidates of every workgroup counted (blk_live), so that k3_q_take's workgroups know their ranks without a walk.
constexpr uint32_t Q_LIVE = 0x80000000u;
__global__ __launch_bounds__(256) void k3_q_live(const K3QueueArgs a) {
  __shared__ uint32_t s_w[4];
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  bool live = false;
  if (j < a.n_cand) {
    const uint32_t s = a.cand_slot[j];
    if (s == 0xFFFFFFFFu) a.cand_slot[j] = 0x7FFFFFFFu;                  // (table full: reported by k3_q_probe; never live)
    else {
      const PairEntry* e = a.table + (s >> 1);
      live = !(__atomic_load_n(&e->state[s & 1], __ATOMIC_RELAXED) & PE_EXPLORED) &&
             __atomic_load_n(&e->pop[s & 1], __ATOMIC_RELAXED) == q_stamp(a.round, j);
      if (live) a.cand_slot[j] = s | Q_LIVE;
    }
  }
  const unsigned long long m = __ballot(live);
  if ((threadIdx.x & 63u) == 0) s_w[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) a.blk_live[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// One workgroup walks the candidates in order: the first `want` live ones are taken.  out[1] = taken, out[2] = consumed (the
// index after the last one taken, or n_cand).
// compaction of the pool (round 6): range r's points pool[start .. start + count) move to dst[prefix .. prefix + count) - the live
// runs of the host's FIFOs, in the order the host lists them; one workgroup per range.
__global__ __launch_bounds__(256) void k3_q_compact(const QRange* ranges, uint32_t n_ranges, const QPoint* src, QPoint* dst, unsigned long long dst_base) {
  const uint32_t r = blockIdx.x;
  if (r >= n_ranges) return;
  const QRange g = ranges[r];
  for (uint32_t i = threadIdx.x; i < g.count; i += blockDim.x) dst[dst_base + g.prefix + i] = src[g.start + i];
}

// (same grid as k3_q_live)  The candidates in order: the first `want` live ones are taken.  A workgroup's first rank is the sum of
// the workgroups' counts before it.  out[1] = taken, out[2] = consumed (the index after the last one taken, or n_cand).
__global__ __launch_bounds__(256) void k3_q_take(const K3QueueArgs a) {
  __shared__ uint32_t s_w[4], s_sum[4];
  const uint32_t t = threadIdx.x, lane = t & 63u, wave = t >> 6;
  uint32_t part = 0;
  for (uint32_t b = t; b < blockIdx.x; b += 256) part += a.blk_live[b];
  for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t x = __shfl_up(part, d); if (lane >= d) part += x; }      // inclusive, per wave
  if (lane == 63) s_sum[wave] = part;
  const uint32_t j = blockIdx.x * 256 + t;
  bool live = false;
  uint32_t s = 0;
  if (j < a.n_cand) {
    const uint32_t cs = a.cand_slot[j];
    live = (cs & Q_LIVE) != 0;
    s = cs & ~Q_LIVE;
  }
  const unsigned long long m = __ballot(live);
  if (lane == 0) s_w[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  const uint32_t first = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
  uint32_t before = 0, mine = 0;
  for (uint32_t w = 0; w < 4; w++) { const uint32_t c = s_w[w]; before += w < wave ? c : 0u; mine += c; }
  const uint32_t rank = first + before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  if (live && rank < a.want) {
    const QPoint q = a.pool[q_cand_index(a.ranges, a.n_ranges, j)];
    DporItem it;
    it.src = q.src; it.branch = q.branch; it.later = q.later; it.earlier = q.earlier; it.pad = 0;
    a.items[rank] = it;
    atomicOr(&a.table[s >> 1].state[s & 1], PE_EXPLORED);        // setExplored(maxIndex, (e1, e2)) (:1170-1172)
    if (rank + 1 == a.want) a.out[2] = j + 1;
  }
  if (t == 0 && blockIdx.x + 1 == gridDim.x) {
    const uint32_t total = first + mine;
    a.out[1] = min(total, a.want);
    if (total < a.want) a.out[2] = a.n_cand;
  }
}

}  // namespace demi
