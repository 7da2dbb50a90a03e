// dpor_host.hpp — host-side bookkeeping of the DPOR exploration (no device code): the backtrack
// priority queue with DefaultBacktrackOrdering (BacktrackOrdering.scala:58-69), the ExploredTacker
// (AuxilaryTypes.scala:209-246), dpor()'s enqueue (:1068-1070, 1134) and getNext() (:1142-1185).
//
// A raft5 interleaving yields ~2 k racing pairs, so this bookkeeping, not the kernel, bounds the
// exploration rate.  What keeps it cheap and exactly equal to the one-at-a-time loop:
//  * every operation on a racing pair touches only the entries (a, b) and (b, a), so the state is sharded by
//    the unordered pair {a, b}; a round's pairs are bucketed by shard (in parallel, by contiguous ranges of
//    interleavings) and each shard is then processed by one thread in global pair order;
//  * the explored set is an open-addressing table (no allocation per insert);
//  * DefaultBacktrackOrdering only compares the branch index (< 256) and PriorityQueue ties are pinned to
//    creation order, so the queue is 256 FIFO buckets per shard (chunks from a per-shard pool: no malloc
//    traffic); the global pop takes, in the highest non-empty branch, the front with the smallest ordinal;
//  * getNext() skips a popped point whose flipped pair is explored (:1153-1157) and the explored set only
//    grows, so a point that can never be popped live is dropped early: when its flipped pair is already
//    explored, when a queued point of the same flipped pair precedes it in pop order (branch >= and created
//    earlier: that one is popped first and explores the pair), or when it reaches its shard's front dead.
#pragma once

#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#include "../../include/demi_gpu.h"

namespace demi_host {

using Trace = std::vector<demi_dpor_trace_entry>;

struct BtPoint {          // one entry of the backTrack queue (DPORwHeuristics.BacktrackKey), 16 bytes
  uint64_t seq;           // global ordinal of the racing pair that created it
  uint32_t trace_id;      // the interleaving that found it (its trace supplies the keys and the next trace)
  uint8_t branch, later, earlier, pad;
};

// (a, b) -> 32-bit value; open addressing, linear probing; key (0, 0) is the empty slot (node keys are FNV
// hash chains: never 0, 0)
class FlatPairMap {
 public:
  FlatPairMap() { resize(1u << 12); }
  const uint32_t* find(uint64_t a, uint64_t b) const {
    for (size_t i = slot(a, b);; i = (i + 1) & mask_) {
      const Entry& k = tab_[i];
      if (k.a == a && k.b == b) return &k.val;
      if (k.a == 0 && k.b == 0) return nullptr;
    }
  }
  // find or insert (value 0); the reference is valid until the next at()
  uint32_t& at(uint64_t a, uint64_t b) {
    if ((n_ + 1) * 5 > (mask_ + 1) * 3) grow();
    for (size_t i = slot(a, b);; i = (i + 1) & mask_) {
      Entry& k = tab_[i];
      if (k.a == a && k.b == b) return k.val;
      if (k.a == 0 && k.b == 0) { k.a = a; k.b = b; n_++; return k.val; }
    }
  }
  size_t size() const { return n_; }

 private:
  struct Entry { uint64_t a, b; uint32_t val, pad; };
  size_t slot(uint64_t a, uint64_t b) const {
    return (size_t)(((a * 0x9E3779B97F4A7C15ULL) ^ (b * 0xC2B2AE3D27D4EB4FULL) ^ (a >> 29)) >> 7) & mask_;
  }
  void resize(size_t cap) { tab_.assign(cap, Entry{0, 0, 0, 0}); mask_ = cap - 1; n_ = 0; }
  void grow() {
    std::vector<Entry> old;
    old.swap(tab_);
    resize((mask_ + 1) * 2);
    for (const Entry& k : old)
      if (k.a || k.b) at(k.a, k.b) = k.val;
  }
  std::vector<Entry> tab_;
  size_t mask_ = 0, n_ = 0;
};

// FIFO of BtPoints in 4 KB chunks taken from (and returned to) a pool owned by the shard
struct Chunk {
  Chunk* next;
  uint32_t head, tail;
  BtPoint item[255];
};
class ChunkPool {
 public:
  Chunk* get() {
    if (!free_) {
      slabs_.emplace_back(new Chunk[SLAB]);
      Chunk* c = slabs_.back().get();
      for (size_t i = 0; i < SLAB; i++) { c[i].next = free_; free_ = &c[i]; }
    }
    Chunk* c = free_;
    free_ = c->next;
    c->next = nullptr; c->head = c->tail = 0;
    return c;
  }
  void put(Chunk* c) { c->next = free_; free_ = c; }

 private:
  static constexpr size_t SLAB = 256;
  std::vector<std::unique_ptr<Chunk[]>> slabs_;
  Chunk* free_ = nullptr;
};
struct Fifo {
  Chunk *head = nullptr, *tail = nullptr;
  bool empty() const { return head == nullptr; }
  const BtPoint& front() const { return head->item[head->head]; }
  void push_back(ChunkPool& pool, const BtPoint& p) {
    if (!tail || tail->tail == 255) {
      Chunk* c = pool.get();
      if (tail) tail->next = c; else head = c;
      tail = c;
    }
    tail->item[tail->tail++] = p;
  }
  void pop_front(ChunkPool& pool) {
    Chunk* c = head;
    if (++c->head == c->tail) {
      head = c->next;
      if (!head) tail = nullptr;
      pool.put(c);
    }
  }
};

class DporBook {
 public:
  explicit DporBook(bool track_history, unsigned n_shards = 64) : track_(track_history), shards_(n_shards) {
    unsigned hw = std::thread::hardware_concurrency();
    threads_ = hw ? (hw > 32 ? 32u : hw) : 4u;
    if (threads_ > n_shards) threads_ = n_shards;
    pieces_.resize((size_t)threads_ * n_shards);
  }

  // dpor() for one round: interleaving i has trace tr[i * MAX_TRACE .. +tl[i]) and pairs pr[i * max_pairs .. +np[i])
  void absorb(const demi_dpor_trace_entry* tr, const uint32_t* tl, const demi_dpor_pair* pr, const uint32_t* np, size_t n,
              uint32_t max_pairs) {
    // trace ids and global pair ordinals (creation order = interleaving order, then pair order)
    std::vector<uint32_t> tid(n);
    std::vector<uint64_t> base(n);
    for (size_t i = 0; i < n; i++) {
      base[i] = seq_;
      seq_ += np[i];
      tid[i] = 0;
      if (np[i]) {
        const demi_dpor_trace_entry* t = tr + i * DEMI_DPOR_MAX_TRACE;
        traces_.push_back(std::make_shared<Trace>(t, t + tl[i]));
        tid[i] = (uint32_t)(traces_.size() - 1);
      }
    }
    const size_t S = shards_.size();
    // phase 1: thread t buckets the pairs of its contiguous range of interleavings by shard
    auto distribute = [&](unsigned t) {
      const size_t lo = n * t / threads_, hi = n * (t + 1) / threads_;
      for (size_t s = 0; s < S; s++) pieces_[t * S + s].clear();
      for (size_t i = lo; i < hi; i++) {
        const demi_dpor_trace_entry* tt = tr + i * DEMI_DPOR_MAX_TRACE;
        const demi_dpor_pair* pp = pr + i * (size_t)max_pairs;
        for (uint32_t k = 0; k < np[i]; k++) {
          const uint64_t ke = tt[pp[k].earlier].key, kl = tt[pp[k].later].key;
          pieces_[t * S + shard_of(ke, kl)].push_back(Piece{ke, kl, BtPoint{base[i] + k, tid[i], pp[k].branch, pp[k].later, pp[k].earlier, 0}});
        }
      }
    };
    // phase 2: thread t processes its shards; the pieces of a shard are read in thread (= interleaving) order
    auto process = [&](unsigned t) {
      for (size_t s = t; s < S; s += threads_) {
        Shard& sh = shards_[s];
        sh.front_valid = false;                                   // newly explored pairs may kill the front point
        for (unsigned src = 0; src < threads_; src++) {
          for (const Piece& pc : pieces_[src * S + s]) {
            const BtPoint& p = pc.p;
            if (track_) {
              sh.map.at(pc.ke, pc.kl) |= EXPLORED;               // setExplored(branchI, (earlier, later)) (:1068-1070)
              uint32_t& flipped = sh.map.at(pc.kl, pc.ke);
              if (flipped & EXPLORED) continue;                   // getNext would skip it (:1153-1157)
              if ((flipped & QUEUED_MASK) > p.branch) continue;   // a queued point of this pair pops before it
              flipped = (flipped & ~QUEUED_MASK) | ((uint32_t)p.branch + 1);
            }
            sh.bucket[p.branch].push_back(sh.pool, p);
            if ((int)p.branch > sh.top) sh.top = (int)p.branch;
            sh.queued++;
          }
        }
      }
    };
    run(distribute);
    run(process);
  }

  // getNext (:1142-1162) + the next trace `trace.take(maxIndex + 1) ++ needToReplay` (:1054-1057, 1180)
  bool get_next(Trace& out) {
    int best = -1;
    for (size_t s = 0; s < shards_.size(); s++) {
      Shard& sh = shards_[s];
      if (!sh.front_valid) settle(sh);
      if (sh.top < 0) continue;
      if (best < 0 || sh.top > shards_[best].top || (sh.top == shards_[best].top && sh.front_seq < shards_[best].front_seq))
        best = (int)s;
    }
    if (best < 0) return false;
    Shard& sh = shards_[best];
    const BtPoint p = sh.bucket[sh.top].front();
    sh.bucket[sh.top].pop_front(sh.pool);
    sh.queued--;
    sh.front_valid = false;
    const Trace& src = *traces_[p.trace_id];
    if (track_) sh.map.at(src[p.later].key, src[p.earlier].key) |= EXPLORED;   // setExplored(maxIndex, (e1, e2)) (:1170-1172)
    out.assign(src.begin(), src.begin() + p.branch + 1);
    for (int k = (int)p.branch + 1; k <= (int)p.later; k++)
      if (k != (int)p.earlier) out.push_back(src[k]);
    return true;
  }

  bool empty() const { return queue_len() == 0; }
  uint64_t queue_len() const {
    uint64_t n = 0;
    for (auto& s : shards_) n += s.queued;
    return n;
  }

 private:
  static constexpr uint32_t EXPLORED = 0x80000000u;   // the pair is in the ExploredTacker
  static constexpr uint32_t QUEUED_MASK = 0x1FFu;     // 1 + highest branch of a queued point that flips INTO this pair
  struct Piece { uint64_t ke, kl; BtPoint p; };
  struct Shard {
    FlatPairMap map;                         // ExploredTacker (+ queued marks) restricted to this shard's pairs
    ChunkPool pool;
    Fifo bucket[256];                        // backTrack, one FIFO per branch index
    int top = -1;                            // highest non-empty bucket once settled
    uint64_t queued = 0;
    bool front_valid = false;                // top / front_seq describe a live (unexplored) point
    uint64_t front_seq = 0;
  };
  // drop dead points from the front of the shard's queue until a live one (or nothing) is at the front
  void settle(Shard& sh) {
    while (sh.top >= 0) {
      Fifo& b = sh.bucket[sh.top];
      if (b.empty()) { sh.top--; continue; }
      const BtPoint& p = b.front();
      if (track_) {
        const Trace& src = *traces_[p.trace_id];
        const uint32_t* v = sh.map.find(src[p.later].key, src[p.earlier].key);
        if (v && (*v & EXPLORED)) { b.pop_front(sh.pool); sh.queued--; continue; }
      }
      sh.front_seq = p.seq;
      break;
    }
    sh.front_valid = true;
  }
  size_t shard_of(uint64_t a, uint64_t b) const {      // unordered pair: (a, b) and (b, a) share a shard
    const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
    return (size_t)(((lo * 0x9E3779B97F4A7C15ULL) ^ (hi * 0xC2B2AE3D27D4EB4FULL)) >> 40) % shards_.size();
  }
  template <class F>
  void run(F&& f) {
    if (threads_ <= 1) { f(0u); return; }
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < threads_; t++) pool.emplace_back(f, t);
    f(0u);
    for (auto& th : pool) th.join();
  }
  bool track_;
  unsigned threads_;
  uint64_t seq_ = 0;
  std::vector<Shard> shards_;
  std::vector<std::vector<Piece>> pieces_;              // [thread][shard] buckets of the current round
  std::vector<std::shared_ptr<Trace>> traces_;
};

// The exploration loop of DPORwHeuristics.test (:1193-1242) in rounds.
//   run(prefixes, prefix_len, stride, n, verdicts, trace_len, n_pairs)  executes one batch of next-traces and returns the
//       per-interleaving scalars (demi_dpor_batch's meaning; the GPU in the library, anything with the same contract in
//       a test harness);
//   fetch(lo, cnt, traces, pairs)  delivers the traces ([cnt][DEMI_DPOR_MAX_TRACE]) and racing pairs ([cnt][max_pairs])
//       of interleavings lo .. lo + cnt of the last batch.
// The bulky results are consumed chunk by chunk (bookkeeping is order-preserving, so a round absorbed in chunks is the
// round absorbed at once): the staging buffers stay small, pinned and allocated once.
// seconds (optional): [0] run, [1] fetch + absorb, [2] get_next.
#ifndef DEMI_EXPLORE_CHUNK
#define DEMI_EXPLORE_CHUNK 4096
#endif
constexpr size_t EXPLORE_CHUNK = DEMI_EXPLORE_CHUNK;

// A buffer from `alloc` / `release` (the library passes pinned host memory so the device-to-host copies run at full
// PCIe rate; malloc / free elsewhere), grown geometrically.
struct RawBuf {
  void* (*alloc)(size_t);
  void (*release)(void*);
  void* p = nullptr;
  size_t cap = 0;
  RawBuf(void* (*a)(size_t), void (*r)(void*)) : alloc(a), release(r) {}
  ~RawBuf() { if (p) release(p); }
  RawBuf(const RawBuf&) = delete;
  RawBuf& operator=(const RawBuf&) = delete;
  void* reserve(size_t bytes) {
    if (bytes > cap) {
      if (p) release(p);
      p = alloc(bytes);
      cap = p ? bytes : 0;
    }
    return p;
  }
};

template <class Run, class Fetch>
int explore_loop(Run&& run, Fetch&& fetch, uint32_t max_pairs, const demi_dpor_search* srch, demi_verdict* out_verdicts,
                 uint32_t* out_prefix_len, uint32_t* out_rounds, demi_dpor_trace_entry* first_violation_trace,
                 uint32_t* first_violation_len, demi_dpor_stats* stats, double* seconds,
                 RawBuf* trace_buf = nullptr, RawBuf* pair_buf = nullptr) {
  // staging buffers: the caller's (the library keeps pinned ones across calls) or malloc'ed ones for this call
  RawBuf own_tr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf own_pr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf& tr_buf = trace_buf ? *trace_buf : own_tr;
  RawBuf& pr_buf = pair_buf ? *pair_buf : own_pr;
  auto* tr = static_cast<demi_dpor_trace_entry*>(tr_buf.reserve(sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * EXPLORE_CHUNK));
  auto* pr = static_cast<demi_dpor_pair*>(pr_buf.reserve(sizeof(demi_dpor_pair) * (size_t)(max_pairs ? max_pairs : 1) * EXPLORE_CHUNK));
  if (!tr || !pr) return DEMI_ERR_INVALID_ARG;
  DporBook book(srch->track_history != 0);
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

  std::vector<Trace> frontier(1);         // first run: nextTrace is empty
  std::vector<demi_dpor_trace_entry> pf;
  std::vector<uint32_t> pl, tl, np;
  std::vector<demi_verdict> vd;
  bool exhausted = false;
  while (!frontier.empty()) {
    const size_t n = frontier.size();
    size_t stride = 1;
    for (auto& f : frontier) stride = f.size() > stride ? f.size() : stride;
    pf.resize(n * stride);             // rows are read up to their prefix length only: the padding is never looked at
    pl.resize(n); tl.resize(n); np.resize(n); vd.resize(n);
    for (size_t i = 0; i < n; i++) {
      pl[i] = (uint32_t)frontier[i].size();
      if (pl[i]) memcpy(&pf[i * stride], frontier[i].data(), sizeof(demi_dpor_trace_entry) * pl[i]);
    }
    double t0 = now();
    int rc = run(pf.data(), pl.data(), (uint32_t)stride, (uint64_t)n, vd.data(), tl.data(), np.data());
    if (rc) return rc;
    double t1 = now();
    if (out_rounds) out_rounds[stats->launches] = (uint32_t)n;
    stats->launches++;
    bool found = false;
    size_t first_here = n;             // position in this round of the overall first violation, if it is in this round
    for (size_t i = 0; i < n; i++) {
      const uint64_t idx = stats->interleavings++;
      out_verdicts[idx] = vd[i];
      out_prefix_len[idx] = pl[i];
      if (vd[i].flags & DEMI_V_VIOLATION) {
        stats->violations++;
        found = true;
        if (stats->first_violation == ~0ull) { stats->first_violation = idx; first_here = i; }
      }
    }
    // dpor(): bookkeeping for the racing pairs of the round (:1122-1139), chunk by chunk, sharded over host threads
    for (size_t lo = 0; lo < n; lo += EXPLORE_CHUNK) {
      const size_t cnt = n - lo < EXPLORE_CHUNK ? n - lo : EXPLORE_CHUNK;
      rc = fetch(lo, cnt, tr, pr);
      if (rc) return rc;
      if (first_here >= lo && first_here < lo + cnt) {
        const size_t k = first_here - lo;
        if (first_violation_trace) memcpy(first_violation_trace, &tr[k * DEMI_DPOR_MAX_TRACE], sizeof(demi_dpor_trace_entry) * tl[first_here]);
        if (first_violation_len) *first_violation_len = tl[first_here];
      }
      book.absorb(tr, tl.data() + lo, pr, np.data() + lo, cnt, max_pairs);
    }
    double t2 = now();
    frontier.clear();
    if (srch->stop_if_violation && found) break;
    if (stats->interleavings >= srch->max_interleavings) break;
    // getNext (:1142-1162) for up to `batch` points
    while (frontier.size() < srch->batch && stats->interleavings + frontier.size() < srch->max_interleavings) {
      Trace nxt;
      if (!book.get_next(nxt)) break;
      frontier.push_back(std::move(nxt));
    }
    if (frontier.empty() && book.empty()) exhausted = true;
    if (seconds) { seconds[0] += t1 - t0; seconds[1] += t2 - t1; seconds[2] += now() - t2; }
  }
  stats->queue_len = book.queue_len();
  stats->exhausted = exhausted ? 1u : 0u;
  return 0;
}

}  // namespace demi_host
