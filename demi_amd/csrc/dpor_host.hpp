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
#include "knobs.hpp"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <queue>
#include <algorithm>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/demi_gpu.h"
#include "dpor_types.hpp"

namespace demi_host {

using Trace = std::vector<demi_dpor_trace_entry>;

struct BtPoint {          // one entry of the backTrack queue (DPORwHeuristics.BacktrackKey), 16 bytes
  uint64_t seq;           // global ordinal of the racing pair that created it
  uint32_t trace_id;      // the interleaving that found it (its trace supplies the keys and the next trace)
  uint8_t branch, later, earlier, pad;
};

// Ordered pair (a, b) -> 32-bit value, with both orientations of a pair in ONE entry: everything dpor() and getNext() do
// with a racing pair touches (a, b) and its flip (b, a), so a pair costs one probe and one cache line.  Open addressing,
// linear probing, keyed by the unordered pair (lo, hi); key (0, 0) is the empty slot (node keys are FNV hash chains:
// never 0, 0).
class FlatPairMap {
 public:
  struct Ref { uint32_t* fwd; uint32_t* rev; };     // the values of (a, b) and of (b, a)
  FlatPairMap() { resize(1u << 12); }
  const uint32_t* find(uint64_t a, uint64_t b) const {
    const bool sw = b < a;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    for (size_t i = slot(lo, hi);; i = (i + 1) & mask_) {
      const Entry& k = tab_[i];
      if (k.lo == lo && k.hi == hi) return &k.val[sw];
      if (k.lo == 0 && k.hi == 0) return nullptr;
    }
  }
  // find or insert (values 0); the pointers are valid until the next at()
  Ref at(uint64_t a, uint64_t b) {
    if ((n_ + 1) * 5 > (mask_ + 1) * 3) grow();
    const bool sw = b < a;
    const uint64_t lo = sw ? b : a, hi = sw ? a : b;
    for (size_t i = slot(lo, hi);; i = (i + 1) & mask_) {
      Entry& k = tab_[i];
      if (k.lo == 0 && k.hi == 0) { k.lo = lo; k.hi = hi; n_++; }
      if (k.lo == lo && k.hi == hi) return Ref{&k.val[sw], &k.val[!sw]};
    }
  }
  size_t size() const { return n_; }
  // ask the memory system for the line find() / at() will look at first (the commit knows its next pairs in advance)
  void prefetch(uint64_t a, uint64_t b) const {
#ifndef DEMI_NO_PREFETCH
    const bool sw = b < a;
    const char* e = reinterpret_cast<const char*>(&tab_[slot(sw ? b : a, sw ? a : b)]);
    __builtin_prefetch(e);
    __builtin_prefetch(e + sizeof(Entry) - 1);       // (24-byte entries: one in three lies across two lines)
#else
    (void)a; (void)b;
#endif
  }

 private:
  struct Entry { uint64_t lo, hi; uint32_t val[2]; };
  size_t slot(uint64_t lo, uint64_t hi) const {
    return (size_t)(((lo * 0x9E3779B97F4A7C15ULL) ^ (hi * 0xC2B2AE3D27D4EB4FULL) ^ (lo >> 29)) >> 7) & mask_;
  }
  void resize(size_t cap) { tab_.assign(cap, Entry{0, 0, {0, 0}}); mask_ = cap - 1; n_ = 0; }
  void grow() {
    std::vector<Entry> old;
    old.swap(tab_);
    resize((mask_ + 1) * 2);
    for (const Entry& k : old)
      if (k.lo || k.hi) { const Ref r = at(k.lo, k.hi); *r.fwd = k.val[0]; *r.rev = k.val[1]; }
  }
  std::vector<Entry> tab_;
  size_t mask_ = 0, n_ = 0;
};

// A set of ORDERED pairs of node keys (the flipped pairs getNext() would skip): open addressing, linear probing, 16-byte
// entries; (0, 0) is the empty slot.  The resident loops insert a pair per dequeued point and per device-side kill - millions
// for a budgeted exploration - where a node-based set spends most of its time allocating (config 5: 0.29 s of 1.0 s).
class FlatPairSet {
 public:
  FlatPairSet() { resize(1u << 14); }
  // true if (a, b) was absent (it is present afterwards)
  bool insert(uint64_t a, uint64_t b) {
    if ((n_ + 1) * 5 > (mask_ + 1) * 3) grow();
    for (size_t i = slot(a, b);; i = (i + 1) & mask_) {
      Entry& k = tab_[i];
      if (k.a == 0 && k.b == 0) { k.a = a; k.b = b; n_++; return true; }
      if (k.a == a && k.b == b) return false;
    }
  }
  size_t size() const { return n_; }
  void prefetch(uint64_t a, uint64_t b) const { __builtin_prefetch(&tab_[slot(a, b)]); }

 private:
  struct Entry { uint64_t a, b; };
  size_t slot(uint64_t a, uint64_t b) const {
    return (size_t)(((a * 0x9E3779B97F4A7C15ULL) ^ (b * 0xC2B2AE3D27D4EB4FULL) ^ (a >> 29)) >> 7) & mask_;
  }
  void resize(size_t cap) { tab_.assign(cap, Entry{0, 0}); mask_ = cap - 1; n_ = 0; }
  void grow() {
    std::vector<Entry> old;
    old.swap(tab_);
    resize((mask_ + 1) * 2);
    for (const Entry& k : old) if (k.a || k.b) insert(k.a, k.b);
  }
  std::vector<Entry> tab_;
  size_t mask_ = 0, n_ = 0;
};

// the order of `n` 64-bit keys (indices, ascending by key, stable): LSD radix passes of 11 bits over the bits the largest key has
inline void radix_order(const uint64_t* key, size_t n, std::vector<uint32_t>& order, std::vector<uint32_t>& tmp) {
  order.resize(n); tmp.resize(n);
  uint64_t all = 0;
  for (size_t i = 0; i < n; i++) { order[i] = (uint32_t)i; all |= key[i]; }
  for (uint32_t shift = 0; shift < 64 && (all >> shift) != 0; shift += 11) {
    uint32_t cnt[2049] = {0};
    for (size_t i = 0; i < n; i++) cnt[((key[i] >> shift) & 2047u) + 1]++;
    for (int b = 0; b < 2048; b++) cnt[b + 1] += cnt[b];
    for (size_t i = 0; i < n; i++) { const uint32_t j = order[i]; tmp[cnt[(key[j] >> shift) & 2047u]++] = j; }
    order.swap(tmp);
  }
}

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

// One finished interleaving as dpor() sees it: its trace and its racing pairs.
struct Finished {
  const demi_dpor_trace_entry* trace;
  uint32_t trace_len;
  const demi_dpor_pair* pairs;
  uint32_t n_pairs;
  bool complete = true;    // every racing pair of the trace was applied by this or an earlier absorb (see ParentFilter)
};

class DporBook {
 public:
  // max_threads = 1: everything on the calling thread (the one-interleaving-at-a-time commit of the reference order)
  explicit DporBook(bool track_history, unsigned max_threads = 32, unsigned n_shards = 64) : track_(track_history), shards_(n_shards) {
    unsigned hw = std::thread::hardware_concurrency();
    threads_ = hw ? (hw > max_threads ? max_threads : hw) : (max_threads < 4u ? max_threads : 4u);
    if (threads_ > n_shards) threads_ = n_shards;
    if (threads_ < 1) threads_ = 1;
    pieces_.resize((size_t)threads_ * n_shards);
  }

  // dpor() for a run of finished interleavings, in the order given
  void absorb(const Finished* fin, size_t n) {
    // trace ids and global pair ordinals (creation order = interleaving order, then pair order)
    std::vector<uint32_t> tid(n);
    std::vector<uint64_t> base(n);
    for (size_t i = 0; i < n; i++) {
      base[i] = seq_;
      seq_ += fin[i].n_pairs;
      tid[i] = 0;
      if (fin[i].n_pairs) {
        traces_.emplace_back();
        traces_.back().t.assign(fin[i].trace, fin[i].trace + fin[i].trace_len);
        traces_.back().complete = fin[i].complete;
        tid[i] = (uint32_t)(traces_.size() - 1);
      }
    }
    const size_t S = shards_.size();
    // phase 1: thread t buckets the pairs of its contiguous range of interleavings by shard
    auto distribute = [&](unsigned t) {
      const size_t lo = n * t / threads_, hi = n * (t + 1) / threads_;
      for (size_t s = 0; s < S; s++) pieces_[t * S + s].clear();
      for (size_t i = lo; i < hi; i++) {
        const demi_dpor_trace_entry* tt = fin[i].trace;
        const demi_dpor_pair* pp = fin[i].pairs;
        for (uint32_t k = 0; k < fin[i].n_pairs; k++) {
          const uint64_t ke = tt[pp[k].earlier].key, kl = tt[pp[k].later].key;
          pieces_[t * S + shard_of(ke, kl)].push_back(Piece{ke, kl, BtPoint{base[i] + k, tid[i], pp[k].branch, pp[k].later, pp[k].earlier, 0}});
        }
      }
    };
    // what dpor() does with one racing pair, in the shard that owns it
    auto apply = [&](Shard& sh, uint64_t ke, uint64_t kl, const BtPoint& p) {
      if (track_) {
#ifdef DEMI_DPOR_PROFILE
        { const uint32_t* a_ = sh.map.find(ke, kl); const uint32_t* b_ = sh.map.find(kl, ke);
          prof_all++; if (a_ && (*a_ & EXPLORED)) { prof_fwd++; if (b_ && (*b_ & EXPLORED)) prof_both++; } }
#endif
        const FlatPairMap::Ref r = sh.map.at(ke, kl);
        *r.fwd |= EXPLORED;                                 // setExplored(branchI, (earlier, later)) (:1068-1070)
        uint32_t& flipped = *r.rev;
        if (flipped & EXPLORED) return;                     // getNext would skip it (:1153-1157)
        if ((flipped & QUEUED_MASK) > p.branch) return;     // a queued point of this pair pops before it
        flipped = (flipped & ~QUEUED_MASK) | ((uint32_t)p.branch + 1);
      }
      sh.bucket[p.branch].push_back(sh.pool, p);
      traces_[p.trace_id].refs.fetch_add(1, std::memory_order_relaxed);
      if ((int)p.branch > sh.top) sh.top = (int)p.branch;
      sh.queued++;
      sh.enqueued++;
    };
    // phase 2: thread t processes its shards; the pieces of a shard are read in thread (= interleaving) order
    auto process = [&](unsigned t) {
      for (size_t s = t; s < S; s += threads_) {
        Shard& sh = shards_[s];
        sh.front_valid = false;                                   // newly explored pairs may kill the front point
        for (unsigned src = 0; src < threads_; src++)
          for (const Piece& pc : pieces_[src * S + s]) apply(sh, pc.ke, pc.kl, pc.p);
      }
    };
    if (threads_ == 1) {
      // one thread (the reference order's commit): the pairs in the order given, no staging copy
      for (Shard& sh : shards_) sh.front_valid = false;
      for (size_t i = 0; i < n; i++) {
        const demi_dpor_trace_entry* tt = fin[i].trace;
        const demi_dpor_pair* pp = fin[i].pairs;
        for (uint32_t k = 0; k < fin[i].n_pairs; k++) {
          const uint64_t ke = tt[pp[k].earlier].key, kl = tt[pp[k].later].key;
          apply(S == 1 ? shards_[0] : shards_[shard_of(ke, kl)], ke, kl,
                BtPoint{base[i] + k, tid[i], pp[k].branch, pp[k].later, pp[k].earlier, 0});
        }
      }
    } else {
      run(distribute);
      run(process);
    }
    for (size_t i = 0; i < n; i++)                                // a trace no queued point refers to is not needed again
      if (fin[i].n_pairs && traces_[tid[i]].refs.load(std::memory_order_relaxed) == 0) Trace().swap(traces_[tid[i]].t);
  }

  // the contiguous layout of one fetched chunk: interleaving i has trace tr[i * MAX_TRACE .. +tl[i]) and pairs
  // pr[i * max_pairs .. +np[i])
  void absorb(const demi_dpor_trace_entry* tr, const uint32_t* tl, const demi_dpor_pair* pr, const uint32_t* np, size_t n,
              uint32_t max_pairs) {
    std::vector<Finished> fin(n);
    for (size_t i = 0; i < n; i++) fin[i] = Finished{tr + i * DEMI_DPOR_MAX_TRACE, tl[i], pr + i * (size_t)max_pairs, np[i]};
    absorb(fin.data(), n);
  }

  // getNext (:1142-1162) + the next trace `trace.take(maxIndex + 1) ++ needToReplay` (:1054-1057, 1180).  *shared = the
  // length of the take() part when its racing pairs need not be reported again (demi_gpu.h, demi_dpor_batch), else 0.
  // *parent (optional) = the whole trace of the interleaving that found the point, when all of its racing pairs have
  // been applied (else left empty): what ParentFilter needs.
  bool get_next(Trace& out, uint32_t* shared = nullptr, Trace* parent = nullptr) {
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
    const Trace& src = traces_[p.trace_id].t;
    if (track_) *sh.map.at(src[p.later].key, src[p.earlier].key).fwd |= EXPLORED;   // setExplored(maxIndex, (e1, e2)) (:1170-1172)
    out.assign(src.begin(), src.begin() + p.branch + 1);
    for (int k = (int)p.branch + 1; k <= (int)p.later; k++)
      if (k != (int)p.earlier) out.push_back(src[k]);
    if (shared) *shared = track_ ? (uint32_t)p.branch + 1 : 0u;
    if (parent) { if (track_ && traces_[p.trace_id].complete) *parent = src; else parent->clear(); }
    release(p.trace_id);
    return true;
  }

  uint64_t enqueued() const {          // backtrack points ever enqueued (demi_dpor_stats.backtrack_points)
    uint64_t n = 0;
    for (auto& s : shards_) n += s.enqueued;
    return n;
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
    uint64_t queued = 0, enqueued = 0;
    bool front_valid = false;                // top / front_seq describe a live (unexplored) point
    uint64_t front_seq = 0;
  };
  struct TraceRec {
    Trace t;                                 // emptied once no queued point refers to it
    std::atomic<uint32_t> refs{0};           // queued backtrack points found by this interleaving
    bool complete = true;
  };
  void release(uint32_t tid) {
    if (traces_[tid].refs.fetch_sub(1, std::memory_order_relaxed) == 1) Trace().swap(traces_[tid].t);
  }
  // drop dead points from the front of the shard's queue until a live one (or nothing) is at the front
  void settle(Shard& sh) {
    while (sh.top >= 0) {
      Fifo& b = sh.bucket[sh.top];
      if (b.empty()) { sh.top--; continue; }
      const BtPoint& p = b.front();
      if (track_) {
        const Trace& src = traces_[p.trace_id].t;
        const uint32_t* v = sh.map.find(src[p.later].key, src[p.earlier].key);
        if (v && (*v & EXPLORED)) { const uint32_t tid = p.trace_id; b.pop_front(sh.pool); sh.queued--; release(tid); continue; }
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
#ifdef DEMI_DPOR_PROFILE
 public:
  unsigned long long prof_all = 0, prof_fwd = 0, prof_both = 0;
 private:
#endif
  bool track_;
  unsigned threads_;
  uint64_t seq_ = 0;
  std::vector<Shard> shards_;
  std::vector<std::vector<Piece>> pieces_;              // [thread][shard] buckets of the current round
  std::deque<TraceRec> traces_;                         // (a deque: records keep their address while it grows)
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

// ------------------------------------------------------------------ ROUNDS order
template <class Run, class Fetch>
int explore_rounds(Run&& run, Fetch&& fetch, uint32_t max_pairs, const demi_dpor_search* srch, demi_verdict* out_verdicts,
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
  std::vector<uint32_t> shared(1, 0u);
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
    int rc = run(pf.data(), pl.data(), shared.data(), (uint32_t)stride, (uint64_t)n, vd.data(), tl.data(), np.data());
    if (rc) return rc;
    double t1 = now();
    if (out_rounds) out_rounds[stats->launches] = (uint32_t)n;
    stats->launches++;
    stats->executed += n;
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
    shared.clear();
    if (srch->stop_if_violation && found) break;
    if (stats->interleavings >= srch->max_interleavings) break;
    // getNext (:1142-1162) for up to `batch` points
    while (frontier.size() < srch->batch && stats->interleavings + frontier.size() < srch->max_interleavings) {
      Trace nxt;
      uint32_t sh = 0;
      if (!book.get_next(nxt, &sh)) break;
      frontier.push_back(std::move(nxt));
      shared.push_back(sh);
    }
    if (frontier.empty() && book.empty()) exhausted = true;
    if (seconds) { seconds[0] += t1 - t0; seconds[1] += t2 - t1; seconds[2] += now() - t2; }
  }
  stats->queue_len = book.queue_len();
  stats->backtrack_points = book.enqueued();
  stats->exhausted = exhausted ? 1u : 0u;
  return 0;
}

struct PairKeyHash {
  size_t operator()(const std::pair<uint64_t, uint64_t>& k) const {
    return (size_t)((k.first * 0x9E3779B97F4A7C15ULL) ^ (k.second * 0xC2B2AE3D27D4EB4FULL) ^ (k.first >> 29));
  }
};

// ------------------------------------------------------------------ other backtrack orderings, distance cap, initial trace
// DPORwHeuristics with a `backtrackHeuristic` other than DefaultBacktrackOrdering, with setMaxDistance and / or
// setInitialTrace (DPORwHeuristics.scala:63-90, 131-134, 211-213): what IncrementalDDMin's DPOR consultations use
// (ArvindDistanceOrdering, BacktrackOrdering.scala:99-173).  The shortcuts of explore_rounds do not hold here - a backtrack
// point whose flipped pair is already explored may not be dropped when it is created (getNext() looks at the queue's head
// before it pops: the head's distance against the cap), and the shared-prefix filter assumes the default priority - so this
// is the plain loop: one priority queue of (priority, creation order), every racing pair enqueued, explored flips skipped
// at pop time (:1153-1157), the queue kept when the head reaches the cap ("Tutto finito", :1144-1150).  Single-threaded; the
// interleavings still run `batch` at a time.  Same rounds, verdicts and prefix lengths as the Python mirror
// (demi_amd/dpor.py DPORwHeuristics.explore with ArvindDistanceOrdering / setMaxDistance / setInitialTrace).
struct OrderedSearch {
  uint32_t ordering = 0;                         // demi_dpor_ordering
  bool capped = false;                           // setMaxDistance called
  uint32_t max_distance = 0;
  std::unordered_map<uint64_t, uint32_t> original_index;   // ArvindDistanceOrdering.init: node key -> index in the original trace
  Trace initial;                                 // setInitialTrace: the first interleaving's next trace
};

// What an exploration leaves behind for the next one on the same instance (ResumableDPOR, IncrementalDeltaDebugging.scala:94-122:
// a later test() of the same DPORwHeuristics continues from its backtrack queue, DPORwHeuristics.scala:1219-1220): the queue
// - a point keeps the trace that found it -, the explored pairs, the creation counter.
struct OrderedState {
  // (distance, branch) compares as the reference's Ordered does - the GREATER one is dequeued first (:155-165 with scala's
  // max-PriorityQueue), ties in creation order
  struct Point { uint32_t distance, branch; uint64_t seq; uint64_t flip_a, flip_b; std::shared_ptr<Trace> trace; uint8_t later, earlier; };
  struct After {
    bool operator()(const Point& x, const Point& y) const {        // "x is dequeued after y"
      if (x.distance != y.distance) return x.distance < y.distance;
      if (x.branch != y.branch) return x.branch < y.branch;
      return x.seq > y.seq;
    }
  };
  std::priority_queue<Point, std::vector<Point>, After> queue;
  std::unordered_set<std::pair<uint64_t, uint64_t>, PairKeyHash> explored;
  uint64_t seq = 0;
  bool started = false;                          // an exploration has run on this state
};

// arvindDistance (BacktrackOrdering.scala:116-143) of the backtrack point (branch, later, earlier) of trace T: events of its
// path the original did not contain, plus misordered pairs among those it did.  The path: the causal chain root .. later
// (getCommonPrefix(later, later)), the events to replay, then (later, earlier) (:117-123).
inline uint32_t arvind_distance(const OrderedSearch& o, const demi_dpor_trace_entry* T, uint32_t branch, uint32_t later, uint32_t earlier) {
  uint32_t chain[DEMI_DPOR_MAX_TRACE + 1], nc = 0;
  for (uint32_t k = later;; k = T[k].parent) { chain[nc++] = k; if (k == 0 || nc > DEMI_DPOR_MAX_TRACE) break; }
  int32_t idx[2 * DEMI_DPOR_MAX_TRACE + 4];
  uint32_t n = 0;
  auto at = [&](uint32_t i) { auto it = o.original_index.find(T[i].key); idx[n++] = it == o.original_index.end() ? -1 : (int32_t)it->second; };
  for (uint32_t k = nc; k-- > 0;) at(chain[k]);
  for (uint32_t i = branch + 1; i <= later; i++) if (i != earlier) at(i);
  at(later); at(earlier);
  uint32_t distance = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (idx[i] < 0) { distance++; continue; }
    for (uint32_t j = 0; j < i; j++) if (idx[j] >= 0 && idx[j] > idx[i]) distance++;
  }
  return distance;
}

template <class Run, class Fetch>
int explore_rounds_ordered(Run&& run, Fetch&& fetch, uint32_t max_pairs, const demi_dpor_search* srch, const OrderedSearch& ord,
                           demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                           demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len, demi_dpor_stats* stats,
                           RawBuf* trace_buf = nullptr, RawBuf* pair_buf = nullptr, OrderedState* persistent = nullptr) {
  // persistent + srch->resume: continue from the queue the previous exploration on this state left (startFromBackTrackPoints)
  RawBuf own_tr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf own_pr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf& tr_buf = trace_buf ? *trace_buf : own_tr;
  RawBuf& pr_buf = pair_buf ? *pair_buf : own_pr;
  auto* tr = static_cast<demi_dpor_trace_entry*>(tr_buf.reserve(sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * EXPLORE_CHUNK));
  auto* pr = static_cast<demi_dpor_pair*>(pr_buf.reserve(sizeof(demi_dpor_pair) * (size_t)(max_pairs ? max_pairs : 1) * EXPLORE_CHUNK));
  if (!tr || !pr) return DEMI_ERR_INVALID_ARG;
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  const bool track = srch->track_history != 0;

  typedef OrderedState::Point Point;
  OrderedState local;
  OrderedState& S = persistent ? *persistent : local;
  const bool cont = persistent && srch->resume && S.started;       // a later test() of the same instance: its explored pairs stay
  if (!cont) S = OrderedState();                                  // a fresh instance
  const bool resume = cont && !S.queue.empty();                   // ... and it continues from its queue if there is one (:1219-1220)
  S.started = true;
  auto& queue = S.queue;
  auto& explored = S.explored;
  uint64_t& seq = S.seq;

  auto get_next = [&](Trace& out) -> bool {                 // getNext (:1142-1185)
    while (!queue.empty()) {
      const Point& head = queue.top();
      if (ord.capped && head.distance >= ord.max_distance) return false;      // the queue is kept
      const Point p = head;
      queue.pop();
      if (track && explored.count({p.flip_a, p.flip_b})) continue;
      if (track) explored.insert({p.flip_a, p.flip_b});
      const Trace& T = *p.trace;
      out.assign(T.begin(), T.begin() + p.branch + 1);
      for (uint32_t i = p.branch + 1; i <= p.later; i++) if (i != p.earlier) out.push_back(T[i]);
      return true;
    }
    return false;
  };

  std::vector<Trace> frontier;
  if (resume) {                                             // (:1219-1220: the first interleaving of a later test() is ONE dequeued point)
    Trace nxt;
    if (get_next(nxt)) frontier.push_back(std::move(nxt));
  } else {
    frontier.push_back(ord.initial);                        // first run: the initial trace, or empty
  }
  std::vector<uint32_t> shared(1, 0u);
  std::vector<demi_dpor_trace_entry> pf;
  std::vector<uint32_t> pl, tl, np;
  std::vector<demi_verdict> vd;
  bool stopped = false;
  while (!frontier.empty()) {
    const size_t n = frontier.size();
    size_t stride = 1;
    for (auto& f : frontier) stride = f.size() > stride ? f.size() : stride;
    pf.resize(n * stride);
    pl.resize(n); tl.resize(n); np.resize(n); vd.resize(n);
    shared.assign(n, 0u);                                   // (no shared-prefix filter: every racing pair is reported)
    for (size_t i = 0; i < n; i++) {
      pl[i] = (uint32_t)frontier[i].size();
      if (pl[i]) memcpy(&pf[i * stride], frontier[i].data(), sizeof(demi_dpor_trace_entry) * pl[i]);
    }
    int rc = run(pf.data(), pl.data(), shared.data(), (uint32_t)stride, (uint64_t)n, vd.data(), tl.data(), np.data());
    if (rc) return rc;
    if (out_rounds) out_rounds[stats->launches] = (uint32_t)n;
    stats->launches++;
    stats->executed += n;
    bool found = false;
    size_t first_here = n;
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
    for (size_t lo = 0; lo < n; lo += EXPLORE_CHUNK) {
      const size_t cnt = n - lo < EXPLORE_CHUNK ? n - lo : EXPLORE_CHUNK;
      rc = fetch(lo, cnt, tr, pr);
      if (rc) return rc;
      if (first_here >= lo && first_here < lo + cnt) {
        const size_t k = first_here - lo;
        if (first_violation_trace) memcpy(first_violation_trace, &tr[k * DEMI_DPOR_MAX_TRACE], sizeof(demi_dpor_trace_entry) * tl[first_here]);
        if (first_violation_len) *first_violation_len = tl[first_here];
      }
      // dpor()'s bookkeeping for each finished interleaving (:1122-1139): setExplored(branch, (earlier, later)), enqueue
      for (size_t k = 0; k < cnt; k++) {
        const demi_dpor_trace_entry* T = &tr[k * DEMI_DPOR_MAX_TRACE];
        const demi_dpor_pair* P = &pr[k * (size_t)(max_pairs ? max_pairs : 1)];
        if (np[lo + k] == 0) continue;
        auto keep = std::make_shared<Trace>(T, T + tl[lo + k]);
        for (uint32_t q = 0; q < np[lo + k]; q++) {
          const demi_dpor_pair p = P[q];
          const uint64_t ke = T[p.earlier].key, kl = T[p.later].key;
          if (track) explored.insert({ke, kl});
          const uint32_t dist = ord.ordering == DEMI_DPOR_ORDERING_ARVIND ? arvind_distance(ord, T, p.branch, p.later, p.earlier) : 0u;
          queue.push(Point{dist, p.branch, seq++, kl, ke, keep, p.later, p.earlier});
          stats->backtrack_points++;
        }
      }
    }
    frontier.clear();
    if (srch->stop_if_violation && found) { stopped = true; break; }
    if (stats->interleavings >= srch->max_interleavings) { stopped = true; break; }
    while (frontier.size() < srch->batch && stats->interleavings + frontier.size() < srch->max_interleavings) {
      Trace nxt;
      if (!get_next(nxt)) break;
      frontier.push_back(std::move(nxt));
    }
  }
  stats->queue_len = queue.size();
  stats->exhausted = (!stopped && queue.empty()) ? 1u : 0u;
  return 0;
}

// ------------------------------------------------------------------ REFERENCE order
// The sequence of interleavings of batch = 1 (DPORwHeuristics' own order), with the device running ahead.
//
// An interleaving is a pure function of its next trace, so its result can be computed at any time and by anyone.  Two
// books are kept: `spec`, a ROUNDS exploration of width `batch` that only serves to guess which next traces will be
// needed, and `real`, which pops ONE backtrack point at a time, takes that interleaving's result from a cache filled by
// the launches (key: the next trace's node keys + its shared length), absorbs it, and pops again - DPORwHeuristics' loop
// verbatim.  When `real` needs a result nobody has computed, the next launch runs it together with `spec`'s next round.
// What `spec` explores only changes how often that happens, never what `real` commits.
struct SpecResult {
  demi_verdict verdict;
  uint32_t prefix_len = 0;
  uint64_t key = 0;
  Trace trace;
  std::vector<demi_dpor_pair> pairs;
  size_t bytes() const { return sizeof(SpecResult) + trace.size() * sizeof(demi_dpor_trace_entry) + pairs.size() * sizeof(demi_dpor_pair); }
};

// `parent` = the trace of the interleaving whose racing pair made this next trace (empty: none / unknown).  It is part of the
// key because the result's pair list is filtered against it (ParentFilter).
inline uint64_t next_trace_key(const Trace& t, uint32_t shared, const Trace& parent) {
  uint64_t h = 0xCBF29CE484222325ULL ^ (uint64_t)t.size() ^ ((uint64_t)shared << 32);
  for (const demi_dpor_trace_entry& e : t) h = (h ^ e.key) * 0x100000001B3ULL + (h >> 47);
  for (const demi_dpor_trace_entry& e : parent) h = (h ^ e.key ^ ((uint64_t)e.qperiod << 56)) * 0x100000001B3ULL + (h >> 47);
  return h ? h : 1;
}

// Which racing pairs of an interleaving C need not be absorbed, because its PARENT P - the interleaving whose racing pair
// produced C's next trace, P.take(branch + 1) ++ replayed events - or one of P's ancestors has applied the same pair already.
//
// Invariant (I): when an interleaving T has been absorbed, every racing pair (e, l) of its trace, with branch index b_T,
// has been applied - by T or by an interleaving absorbed before it - as (key(e), key(l), b') with b' >= b_T.
// Applying (ke, kl, b) when (ke, kl, b') with b' >= b was applied before changes nothing: the explored bit of (ke, kl) is
// set, and the flipped pair is either explored or carries a queued mark >= b' + 1 > b; both only ever grow (absorb()).
//
// A racing pair is a property of its two nodes: both message deliveries, same receiver, no causal path from the earlier
// to the later one (all three read off the node keys, which ARE causal paths) and equal quiescent periods; its branch is
// the trace index of the last common ancestor of the two producers.  So if the nodes of a pair (e, l) of C also occur in
// P, in the same order and in the same quiescent periods, P's dpor() saw the same racing pair with b_P = P's index of the
// same ancestor node, and by (I) for P it was applied with a branch >= b_P.  If b_P >= b_C, C may drop the pair and (I)
// still holds for C.  P is absorbed before C is dequeued (C's backtrack point is created by absorbing P).  The device's
// shared-prefix filter is the special case l < shared (same indices in P and C).  Because equal (parent, message)
// children collapse into one key, the argument needs the three keys involved (e, l, the ancestor) to be unique in P and
// in C; a pair touching a duplicated key is kept.  (I) fails for an interleaving whose pair list was truncated
// (DEMI_V_PAIRS_OVF) or whose parent's did: get_next() then hands out no parent and nothing is dropped.
// Config 3: 109 M reported pairs -> 8.9 M absorbed.
class ParentFilter {
 public:
  explicit ParentFilter(const Trace& parent) {
    if (parent.empty() || parent.size() > DEMI_DPOR_MAX_TRACE) return;
    on_ = true;
    n_ = (uint32_t)parent.size();
    memset(slot_, 0xFF, sizeof slot_);
    for (uint32_t i = 0; i < n_; i++) {
      key_[i] = parent[i].key; qp_[i] = parent[i].qperiod; dup_[i] = 0;
      size_t h = hash(key_[i]);
      for (; slot_[h] != 0xFFFFu; h = (h + 1) & (SLOTS - 1))
        if (key_[slot_[h]] == key_[i]) { dup_[slot_[h]] = 1; dup_[i] = 1; }
      slot_[h] = (uint16_t)i;
    }
  }
  bool active() const { return on_; }
  // keeps the pairs that still have to be absorbed
  void filter(const demi_dpor_trace_entry* tr, uint32_t n_tr, const demi_dpor_pair* pr, uint32_t n_pr,
              std::vector<demi_dpor_pair>& out) const {
    if (n_tr > DEMI_DPOR_MAX_TRACE) { out.assign(pr, pr + n_pr); return; }
    // idx[i] = where C's event i sits in the parent (same key, same quiescent period, key unique on both sides), or -1
    int idx[DEMI_DPOR_MAX_TRACE];
    uint16_t own[SLOTS];
    memset(own, 0xFF, sizeof own);
    for (uint32_t i = 0; i < n_tr; i++) {
      idx[i] = index_of(tr[i]);
      size_t h = hash(tr[i].key);
      for (; own[h] != 0xFFFFu; h = (h + 1) & (SLOTS - 1))
        if (tr[own[h]].key == tr[i].key) { idx[own[h]] = -1; idx[i] = -1; }
      own[h] = (uint16_t)i;
    }
    uint32_t keep = 0;
    for (uint32_t k = 0; k < n_pr; k++) keep += !redundant(idx, pr[k]);
    out.clear();
    out.reserve(keep);
    for (uint32_t k = 0; k < n_pr; k++)
      if (!redundant(idx, pr[k])) out.push_back(pr[k]);
  }

 private:
  static constexpr size_t SLOTS = 1024;             // > 2 x DEMI_DPOR_MAX_TRACE
  static size_t hash(uint64_t k) { return (size_t)((k * 0x9E3779B97F4A7C15ULL) >> 54) & (SLOTS - 1); }
  int index_of(const demi_dpor_trace_entry& e) const {
    for (size_t h = hash(e.key);; h = (h + 1) & (SLOTS - 1)) {
      const uint16_t i = slot_[h];
      if (i == 0xFFFFu) return -1;
      if (key_[i] == e.key) return (!dup_[i] && qp_[i] == e.qperiod) ? (int)i : -1;
    }
  }
  static bool redundant(const int* idx, const demi_dpor_pair& p) {
    return idx[p.earlier] >= 0 && idx[p.later] >= 0 && idx[p.earlier] < idx[p.later] && idx[p.branch] >= (int)p.branch;
  }
  bool on_ = false;
  uint32_t n_ = 0;
  uint64_t key_[DEMI_DPOR_MAX_TRACE];
  uint8_t qp_[DEMI_DPOR_MAX_TRACE], dup_[DEMI_DPOR_MAX_TRACE];
  uint16_t slot_[SLOTS];
};

template <class Run, class Fetch>
int explore_reference_order(Run&& run, Fetch&& fetch, uint32_t max_pairs, const demi_dpor_search* srch,
                            demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                            demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                            demi_dpor_stats* stats, double* seconds, RawBuf* trace_buf = nullptr, RawBuf* pair_buf = nullptr) {
  RawBuf own_tr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf own_pr([](size_t b) { return malloc(b); }, [](void* q) { free(q); });
  RawBuf& tr_buf = trace_buf ? *trace_buf : own_tr;
  RawBuf& pr_buf = pair_buf ? *pair_buf : own_pr;
  auto* tr = static_cast<demi_dpor_trace_entry*>(tr_buf.reserve(sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * EXPLORE_CHUNK));
  auto* pr = static_cast<demi_dpor_pair*>(pr_buf.reserve(sizeof(demi_dpor_pair) * (size_t)(max_pairs ? max_pairs : 1) * EXPLORE_CHUNK));
  if (!tr || !pr) return DEMI_ERR_INVALID_ARG;
  const bool track = srch->track_history != 0;
  DporBook real(track, 1, 1), spec(track);   // the one-at-a-time book: one shard, so getNext() settles one queue front per pop instead of 64
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };

  // results computed and not yet committed; evicted oldest first beyond the memory cap (a needed one is then run again)
  std::unordered_map<uint64_t, std::unique_ptr<SpecResult>> cache;
  std::deque<uint64_t> age;
  size_t cache_bytes = 0;
  const size_t cache_cap = (size_t)(srch->cache_mb ? srch->cache_mb : 1024u) << 20;
  auto evict_to_cap = [&](uint64_t keep) {
    while (cache_bytes > cache_cap && !age.empty()) {
      const uint64_t k = age.front();
      age.pop_front();
      if (k == keep) { age.push_back(k); if (age.size() == 1) break; continue; }
      auto it = cache.find(k);
      if (it != cache.end()) { cache_bytes -= it->second->bytes(); cache.erase(it); }
    }
  };

#ifdef DEMI_DPOR_PROFILE
  double prof[3] = {0, 0, 0};
  unsigned long long prof_pairs = 0;
#endif
  Trace cur, cur_parent;             // the next trace `real` wants (first run: empty) and the trace it was derived from
  uint32_t cur_shared = 0;
  bool have_cur = true, exhausted = false, done = false;
  std::vector<Trace> spec_frontier(1), spec_parent(1);
  std::vector<uint32_t> spec_shared(1, 0u);
  bool spec_alive = true;

  std::vector<const Trace*> launch, launch_parent;   // next traces of this launch (and what each was derived from)
  std::vector<uint32_t> launch_shared;
  std::vector<uint64_t> launch_key;
  std::vector<demi_dpor_trace_entry> pf;
  std::vector<uint32_t> pl, tl, np;
  std::vector<demi_verdict> vd;
  std::vector<Finished> fin;
  std::vector<std::unique_ptr<SpecResult>> made;
  const bool no_parent_filter = demi_host::knob("DEMI_DPOR_NO_PARENT_FILTER") != nullptr;   // A/B and tests: absorb every reported pair
  unsigned long long stats_pairs_reported = 0, stats_pairs_kept = 0;

  while (!done) {
    // ---- commit, one interleaving at a time, as far as computed results reach
    double t0 = now();
    while (have_cur) {
      const uint64_t key = next_trace_key(cur, cur_shared, cur_parent);
      auto it = cache.find(key);
      if (it == cache.end()) break;
      const SpecResult& r = *it->second;
      const uint64_t idx = stats->interleavings++;
      out_verdicts[idx] = r.verdict;
      out_prefix_len[idx] = (uint32_t)cur.size();
      bool found = false;
      if (r.verdict.flags & DEMI_V_VIOLATION) {
        stats->violations++;
        found = true;
        if (stats->first_violation == ~0ull) {
          stats->first_violation = idx;
          if (first_violation_trace && !r.trace.empty()) memcpy(first_violation_trace, r.trace.data(), sizeof(demi_dpor_trace_entry) * r.trace.size());
          if (first_violation_len) *first_violation_len = (uint32_t)r.trace.size();
        }
      }
      // invariant (I) of ParentFilter holds for this interleaving if its own pair list is whole and (I) held for its parent
      const Finished f{r.trace.data(), (uint32_t)r.trace.size(), r.pairs.data(), (uint32_t)r.pairs.size(),
                       !(r.verdict.flags & DEMI_V_PAIRS_OVF) && (cur.empty() || !cur_parent.empty())};
#ifdef DEMI_DPOR_PROFILE
      const double p0 = now();
      prof_pairs += r.pairs.size();
#endif
      real.absorb(&f, 1);
#ifdef DEMI_DPOR_PROFILE
      const double p1 = now();
#endif
      cache_bytes -= r.bytes();
      cache.erase(it);
      if ((srch->stop_if_violation && found) || stats->interleavings >= srch->max_interleavings) { done = true; break; }
#ifdef DEMI_DPOR_PROFILE
      const double p2 = now();
#endif
      have_cur = real.get_next(cur, &cur_shared, &cur_parent);
#ifdef DEMI_DPOR_PROFILE
      const double p3 = now();
      prof[0] += p1 - p0; prof[1] += p2 - p1; prof[2] += p3 - p2;
#endif
      if (!have_cur) { exhausted = true; done = true; }
    }
    double t1 = now();
    if (seconds) seconds[2] += t1 - t0;
    if (done) break;

    // ---- one launch: what `real` is waiting for + the speculation's next round (minus what is computed already)
    stats->cache_misses++;
    launch.clear(); launch_parent.clear(); launch_shared.clear(); launch_key.clear();
    const uint64_t cur_key = next_trace_key(cur, cur_shared, cur_parent);
    launch.push_back(&cur); launch_parent.push_back(&cur_parent); launch_shared.push_back(cur_shared); launch_key.push_back(cur_key);
    std::vector<uint64_t> round_key(spec_frontier.size());
    {
      std::unordered_map<uint64_t, char> in_launch;
      in_launch[cur_key] = 1;
      for (size_t i = 0; i < spec_frontier.size(); i++) {
        const uint64_t k = next_trace_key(spec_frontier[i], spec_shared[i], spec_parent[i]);
        round_key[i] = k;
        if (cache.count(k) != 0 || !in_launch.emplace(k, 1).second) continue;
        launch.push_back(&spec_frontier[i]); launch_parent.push_back(&spec_parent[i]);
        launch_shared.push_back(spec_shared[i]); launch_key.push_back(k);
      }
    }
    const size_t n = launch.size();
    size_t stride = 1;
    for (const Trace* f : launch) stride = f->size() > stride ? f->size() : stride;
    pf.resize(n * stride);
    pl.resize(n); tl.resize(n); np.resize(n); vd.resize(n);
    for (size_t i = 0; i < n; i++) {
      pl[i] = (uint32_t)launch[i]->size();
      if (pl[i]) memcpy(&pf[i * stride], launch[i]->data(), sizeof(demi_dpor_trace_entry) * pl[i]);
    }
    int rc = run(pf.data(), pl.data(), launch_shared.data(), (uint32_t)stride, (uint64_t)n, vd.data(), tl.data(), np.data());
    if (rc) return rc;
    if (out_rounds && stats->launches < srch->max_interleavings) out_rounds[stats->launches] = (uint32_t)n;
    stats->launches++;
    stats->executed += n;
    double t2 = now();
    for (size_t lo = 0; lo < n; lo += EXPLORE_CHUNK) {
      const size_t cnt = n - lo < EXPLORE_CHUNK ? n - lo : EXPLORE_CHUNK;
      rc = fetch(lo, cnt, tr, pr);
      if (rc) return rc;
      // the results are built by several threads; each keeps only the racing pairs its parent has not applied already
      made.clear();
      made.resize(cnt);
      auto build = [&](unsigned t, unsigned nt) {
        for (size_t i = cnt * t / nt; i < cnt * (t + 1) / nt; i++) {
          std::unique_ptr<SpecResult> r(new SpecResult);
          r->verdict = vd[lo + i];
          r->prefix_len = pl[lo + i];
          r->key = launch_key[lo + i];
          const demi_dpor_trace_entry* tt = &tr[i * DEMI_DPOR_MAX_TRACE];
          const demi_dpor_pair* pp = &pr[i * (size_t)max_pairs];
          const uint32_t n_pp = np[lo + i] < max_pairs ? np[lo + i] : max_pairs;
          r->trace.assign(tt, tt + tl[lo + i]);
          const ParentFilter pf_(track && !no_parent_filter ? *launch_parent[lo + i] : Trace());
          if (pf_.active()) pf_.filter(tt, tl[lo + i], pp, n_pp, r->pairs);
          else r->pairs.assign(pp, pp + n_pp);
          made[i] = std::move(r);
        }
      };
      {
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt ? (nt > 32u ? 32u : nt) : 4u;
        if (cnt < 64) nt = 1;
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(build, t, nt);
        build(0u, nt);
        for (auto& th : pool) th.join();
      }
      for (size_t i = 0; i < cnt; i++) {
        stats_pairs_reported += np[lo + i];
        stats_pairs_kept += made[i]->pairs.size();
        cache_bytes += made[i]->bytes();
        age.push_back(made[i]->key);
        cache[made[i]->key] = std::move(made[i]);
      }
    }
    // the speculation absorbs its round in pop order (every member is in the cache now) and pops its next round
    if (spec_alive) {
      fin.clear();
      for (size_t i = 0; i < spec_frontier.size(); i++) {
        auto it = cache.find(round_key[i]);
        if (it == cache.end()) continue;               // (only if the cap is smaller than one round)
        const SpecResult& r = *it->second;
        fin.push_back(Finished{r.trace.data(), (uint32_t)r.trace.size(), r.pairs.data(), (uint32_t)r.pairs.size(),
                               !(r.verdict.flags & DEMI_V_PAIRS_OVF)});
      }
      spec.absorb(fin.data(), fin.size());
      spec_frontier.clear();
      spec_parent.clear();
      spec_shared.clear();
      while (spec_frontier.size() < srch->batch) {
        Trace nxt, par;
        uint32_t sh = 0;
        if (!spec.get_next(nxt, &sh, &par)) break;
        spec_frontier.push_back(std::move(nxt));
        spec_parent.push_back(std::move(par));
        spec_shared.push_back(sh);
      }
      if (spec_frontier.empty()) spec_alive = false;
    }
    evict_to_cap(cur_key);
    if (seconds) { seconds[0] += t2 - t1; seconds[1] += now() - t2; }
  }
  stats->queue_len = real.queue_len();
  stats->backtrack_points = real.enqueued();
  stats->exhausted = exhausted ? 1u : 0u;
  if (demi_host::knob("DEMI_DPOR_TIMING"))
    fprintf(stderr, "[reference order] racing pairs reported %llu, kept after the parent filter %llu\n", stats_pairs_reported, stats_pairs_kept);
#ifdef DEMI_DPOR_PROFILE
  fprintf(stderr, "[reference order commit] absorb %.3f s, cache erase %.3f s, get_next %.3f s, %llu pairs\n", prof[0], prof[1], prof[2],
          (unsigned long long)prof_pairs);
  fprintf(stderr, "[reference order commit] pairs %llu, forward orientation already explored %llu, both orientations explored %llu, table entries ?\n",
          real.prof_all, real.prof_fwd, real.prof_both);
#endif
  return 0;
}

// ------------------------------------------------------------------ ROUNDS order, bookkeeping on the device
// The same exploration as explore_rounds (track_history, DefaultBacktrackOrdering) when the ExploredTacker and dpor()'s
// enqueue decision live on the device (k3_pairs.hpp): per round the host sends the dequeued points (8 bytes each - the
// traces they refer to stayed in the device's arena) and gets back one verdict per interleaving, the backtrack points
// that can still be dequeued live, and the pairs that became explored while points flipping into them were queued.
// What is left here is the queue itself: 256 FIFO buckets by branch index and getNext()'s skip of explored pairs.
//   dev.round(items, n, round, base_id, verdicts, points, kills)   one launch; interleaving i gets arena id base_id + i
//   dev.ids_used(n)                                                arena ids a round of n items consumes (n; with several
//                                                                  GPUs the ranks' equal-sized blocks: world * ceil(n / world))
//   dev.fetch_trace(id, out, &len)                                 one finished trace (the first violation's)
// With several GPUs every rank runs this same loop on identical queues (SPMD): dev.round() returns the same verdicts,
// points and kills on every rank, whatever part of the round and of the explored-pair table that rank worked on.

// What the device-resident explorations remember of every interleaving they return (demi_dpor_explored): the backtrack point
// it was dequeued as and the arena row that holds the trace it executed.  8 + 4 bytes per interleaving.
struct ExploredLog {
  std::vector<demi::DporItem> item;
  std::vector<uint32_t> id;
  void clear() { item.clear(); id.clear(); }
  void add(const demi::DporItem& it, uint32_t arena_id) { item.push_back(it); id.push_back(arena_id); }
  size_t size() const { return id.size(); }
};

template <class Dev>
int explore_rounds_resident(Dev&& dev, const demi_dpor_search* srch, demi_verdict* out_verdicts, uint32_t* out_prefix_len,
                            uint32_t* out_rounds, demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                            demi_dpor_stats* stats, double* seconds, ExploredLog* log = nullptr) {
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  if (log) log->clear();
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  std::deque<demi::DporPoint> bucket[256];
  int top = -1;
  uint64_t queued = 0;
  FlatPairSet dead;                                                        // flipped pairs getNext() would skip
  std::vector<demi::DporItem> items(1, demi::DporItem{0xFFFFFFFFu, 0, 0, 0, 0});   // first run: nextTrace is empty
  std::vector<demi_verdict> vd;
  std::vector<demi::DporPoint> pts;
  std::vector<demi::DporKill> kills;
  std::vector<uint64_t> ord_key;
  std::vector<uint32_t> ord, ord_tmp;
  uint32_t base_id = 0, round = 0;
  uint64_t first_id = ~0ull;
  bool exhausted = false;
  while (!items.empty()) {
    const uint32_t n = (uint32_t)items.size();
    vd.resize(n);
    pts.clear(); kills.clear();
    round++;
    double t0 = now();
    int rc = dev.round(items.data(), n, round, base_id, vd.data(), pts, kills);
    if (rc) return rc;
    double t1 = now();
    if (out_rounds) out_rounds[stats->launches] = n;
    stats->launches++;
    stats->executed += n;
    bool found = false;
    for (uint32_t i = 0; i < n; i++) {
      const uint64_t idx = stats->interleavings++;
      out_verdicts[idx] = vd[i];
      out_prefix_len[idx] = items[i].src == 0xFFFFFFFFu ? 0u : (uint32_t)items[i].later;
      if (log) log->add(items[i], base_id + i);
      if (vd[i].flags & DEMI_V_VIOLATION) {
        stats->violations++;
        found = true;
        if (stats->first_violation == ~0ull) { stats->first_violation = idx; first_id = base_id + i; }
      }
    }
    base_id += dev.ids_used(n);
    {
      constexpr size_t AHEAD = 12;
      const size_t nk = kills.size();
      for (size_t i = 0; i < nk && i < AHEAD; i++) dead.prefetch(kills[i].a, kills[i].b);
      for (size_t i = 0; i < nk; i++) {
        if (i + AHEAD < nk) dead.prefetch(kills[i + AHEAD].a, kills[i + AHEAD].b);
        dead.insert(kills[i].a, kills[i].b);
      }
    }
    // creation order: the round's interleavings in pop order, then pair order (ordinals are unique within a round)
    ord_key.resize(pts.size());
    for (size_t i = 0; i < pts.size(); i++) ord_key[i] = pts[i].ordinal;
    radix_order(ord_key.data(), pts.size(), ord, ord_tmp);
    stats->backtrack_points += pts.size();
    for (size_t i = 0; i < pts.size(); i++) {
      const demi::DporPoint& p = pts[ord[i]];
      bucket[p.branch].push_back(p);
      if ((int)p.branch > top) top = (int)p.branch;
      queued++;
    }
    items.clear();
    double t2 = now();
    if (seconds) { seconds[0] += t1 - t0; seconds[1] += t2 - t1; }
    if (srch->stop_if_violation && found) break;
    if (stats->interleavings >= srch->max_interleavings) break;
    // getNext (:1142-1162): deepest branch first, creation order within a branch, explored pairs skipped
    while (items.size() < srch->batch && stats->interleavings + items.size() < srch->max_interleavings) {
      while (top >= 0 && bucket[top].empty()) top--;
      if (top < 0) break;
      const demi::DporPoint p = bucket[top].front();
      bucket[top].pop_front();
      queued--;
      if (!dead.insert(p.flip_a, p.flip_b)) continue;              // isExplored: skip; else setExplored (:1170-1172)
      items.push_back(demi::DporItem{p.src, p.branch, p.later, p.earlier, 0});
    }
    if (items.empty() && queued == 0) exhausted = true;
    if (seconds) seconds[2] += now() - t2;
  }
  if (first_id != ~0ull && first_violation_trace && first_violation_len) {
    int rc = dev.fetch_trace((uint32_t)first_id, first_violation_trace, first_violation_len);
    if (rc) return rc;
  }
  stats->queue_len = queued;
  stats->exhausted = exhausted ? 1u : 0u;
  return 0;
}

// ------------------------------------------------------------------ ROUNDS order, bookkeeping AND queue on the device
// explore_rounds_resident with the backtrack queue itself on the device (k3_queue.hpp; single rank).  What is left here: per branch
// the FIFO of runs (pool offset, points left) - the device sorts every round's points into one run per branch - and the order in
// which runs are handed to a dequeue: deepest branch first, oldest run first (DefaultBacktrackOrdering with PriorityQueue ties in
// creation order: the order of explore_rounds_resident's 256 buckets).
//   dev.q_round(items, n, round, base_id, pool_fill, verdicts, &points, run_len[256])
//       one launch: K3 for the items, dpor()'s insert / decide, the round's live points sorted into the pool segment at pool_fill
//   dev.q_pop(ranges, n_ranges, n_cand, want, dequeue_no, items_out, &taken, &consumed)
//       getNext() for up to `want` items among the candidates of `ranges` (in that order): the first `taken` live ones, `consumed`
//       candidates used up
struct QueueRange { unsigned long long start; uint32_t count, prefix; };      // = demi::QRange

template <class Dev>
int explore_rounds_devqueue(Dev&& dev, const demi_dpor_search* srch, demi_verdict* out_verdicts, uint32_t* out_prefix_len,
                            uint32_t* out_rounds, demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                            demi_dpor_stats* stats, double* seconds, ExploredLog* log = nullptr) {
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  if (log) log->clear();
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  struct Run { unsigned long long start; uint32_t left; };
  std::deque<Run> runs[256];
  int top = -1;
  uint64_t queued = 0, pool_fill = 0;
  std::vector<demi::DporItem> items(1, demi::DporItem{0xFFFFFFFFu, 0, 0, 0, 0});   // first run: nextTrace is empty
  std::vector<demi_verdict> vd;
  std::vector<QueueRange> ranges;
  uint64_t run_len[256];
  uint32_t base_id = 0, round = 0, dequeue_no = 0;
  uint64_t first_id = ~0ull;
  bool exhausted = false;
  constexpr size_t MAX_RANGES = 4096;
  // (pools below 2^22 points - 100 MB - are not worth a compaction; DEMI_K3_POOL_COMPACT_MIN: a test's way to a small threshold)
  // (with the knob set a compaction happens as soon as that many points are dead, whatever the live share)
  uint64_t compact_min = 1ull << 22;
  bool compact_eager = false;
  if (const char* e = demi_host::knob("DEMI_K3_POOL_COMPACT_MIN")) { const unsigned long long v = strtoull(e, nullptr, 10); if (v >= 1) { compact_min = v; compact_eager = true; } }
  while (!items.empty()) {
    const uint32_t n = (uint32_t)items.size();
    vd.resize(n);
    round++;
    double t0 = now();
    uint64_t points = 0;
    int rc = dev.q_round(items.data(), n, round, base_id, pool_fill, vd.data(), &points, run_len);
    if (rc) return rc;
    double t1 = now();
    if (out_rounds) out_rounds[stats->launches] = n;
    stats->launches++;
    stats->executed += n;
    bool found = false;
    for (uint32_t i = 0; i < n; i++) {
      const uint64_t idx = stats->interleavings++;
      out_verdicts[idx] = vd[i];
      out_prefix_len[idx] = items[i].src == 0xFFFFFFFFu ? 0u : (uint32_t)items[i].later;
      if (log) log->add(items[i], base_id + i);
      if (vd[i].flags & DEMI_V_VIOLATION) {
        stats->violations++;
        found = true;
        if (stats->first_violation == ~0ull) { stats->first_violation = idx; first_id = base_id + i; }
      }
    }
    base_id += n;
    // the round's segment: one run per branch, deepest first
    {
      uint64_t acc = 0;
      for (int b = 255; b >= 0; b--) {
        if (!run_len[b]) continue;
        runs[b].push_back(Run{pool_fill + acc, (uint32_t)run_len[b]});
        acc += run_len[b];
        if (b > top) top = b;
      }
      if (acc != points) return DEMI_ERR_DEVICE;          // (the device's own counts disagree)
      pool_fill += points;
      queued += points;
      stats->backtrack_points += points;
    }
    items.clear();
    double t2 = now();
    if (seconds) { seconds[0] += t1 - t0; seconds[1] += t2 - t1; }
    if (srch->stop_if_violation && found) break;
    if (stats->interleavings >= srch->max_interleavings) break;
    // getNext (:1142-1162) for up to `batch` points
    const uint64_t room = srch->max_interleavings - stats->interleavings;
    const uint32_t want = (uint32_t)(room < srch->batch ? room : srch->batch);
    while (items.size() < want && queued != 0) {
      const uint32_t need = want - (uint32_t)items.size();
      const uint64_t target = (uint64_t)need * 2 > 4096 ? (uint64_t)need * 2 : 4096;
      ranges.clear();
      uint32_t n_cand = 0;
      for (int b = top; b >= 0 && n_cand < target && ranges.size() < MAX_RANGES; b--)
        for (const Run& r : runs[b]) {
          if (n_cand >= target || ranges.size() >= MAX_RANGES) break;
          const uint32_t take = (uint64_t)r.left < target - n_cand ? r.left : (uint32_t)(target - n_cand);
          ranges.push_back(QueueRange{r.start, take, n_cand});
          n_cand += take;
        }
      if (n_cand == 0) break;
      uint32_t taken = 0, consumed = 0;
      const size_t have = items.size();
      items.resize(have + need);
      rc = dev.q_pop(ranges.data(), (uint32_t)ranges.size(), n_cand, need, ++dequeue_no, items.data() + have, &taken, &consumed);
      if (rc) return rc;
      if (taken > need || consumed > n_cand) return DEMI_ERR_DEVICE;
      items.resize(have + taken);
      // the consumed candidates leave the queue, in the order they were offered
      uint32_t left = consumed;
      while (left) {
        while (top >= 0 && runs[top].empty()) top--;
        Run& r = runs[top].front();
        const uint32_t c = r.left < left ? r.left : left;
        r.start += c; r.left -= c; left -= c;
        if (!r.left) runs[top].pop_front();
      }
      queued -= consumed;
      while (top >= 0 && runs[top].empty()) top--;
    }
    if (items.empty() && queued == 0) exhausted = true;
    // The pool only ever grew: 24 B for every point ever emitted (advisor, round 5).  Once more than half of a sizeable pool is
    // dequeued points, the live runs move to the front of a fresh one (deepest bucket first, FIFO order kept within a bucket)
    // and the runs are rebased.  A failed allocation leaves everything as it was.
    if (!items.empty() && pool_fill >= compact_min && (compact_eager ? pool_fill - queued >= compact_min : pool_fill - queued > queued)) {
      if (dev.q_compact_begin(queued) == 0) {
        uint64_t acc = 0;
        int crc = 0;
        ranges.clear();
        uint64_t block_base = 0;
        auto flush = [&]() {
          if (!ranges.empty() && !crc) crc = dev.q_compact_add(ranges.data(), (uint32_t)ranges.size(), block_base);
          ranges.clear();
          block_base = acc;
        };
        std::vector<std::pair<Run*, uint64_t>> moved;
        for (int b = top; b >= 0 && !crc; b--)
          for (Run& r : runs[b]) {
            if ((uint64_t)(acc - block_base) + r.left > 0xFFFFFFFFull || ranges.size() >= MAX_RANGES) flush();
            ranges.push_back(QueueRange{r.start, r.left, (uint32_t)(acc - block_base)});
            moved.emplace_back(&r, acc);
            acc += r.left;
          }
        flush();
        if (!crc && acc == queued) {
          crc = dev.q_compact_end();
          if (!crc) { for (auto& m : moved) m.first->start = m.second; pool_fill = queued; stats->fetches++; }
        } else dev.q_compact_abort();
        if (crc) return crc;
      }
    }
    if (seconds) seconds[2] += now() - t2;
  }
  if (first_id != ~0ull && first_violation_trace && first_violation_len) {
    int rc = dev.fetch_trace((uint32_t)first_id, first_violation_trace, first_violation_len);
    if (rc) return rc;
  }
  stats->queue_len = queued;
  stats->exhausted = exhausted ? 1u : 0u;
  return 0;
}

// ------------------------------------------------------------------ REFERENCE order, results resident on the device
// The same committed sequence as explore_reference_order (DPORwHeuristics' own one-at-a-time order), with what made that
// path slow taken out: every finished trace used to cross PCIe together with all of its racing pairs (1.3 GB for config 3),
// to be filtered and absorbed on the host.  Here
//   * traces stay in the device's arena; a backtrack point is {arena id of the interleaving that found it, branch, later,
//     earlier} and the device builds the next trace from it (as in the ROUNDS path), so the item IS the identity of an
//     interleaving: results are cached under it, no trace hashing;
//   * the SPECULATION - which interleavings to run before the commit asks for them - is the device-resident ROUNDS
//     exploration itself (explored-pair table, candidates and enqueue decision on the device, explore_rounds_resident's
//     queue here): it costs 3.5 MB of PCIe for the whole of config 3;
//   * the COMMIT (`real`: pop one point, absorb that interleaving's racing pairs, pop again - getNext() / dpor() verbatim)
//     only ever sees the racing pairs that can still change its state.  The device drops, per interleaving and in pair
//     order, (a) the pairs its parent provably applied (ParentFilter's rule, evaluated against the parent's trace in the
//     arena) and (b) the pairs that are no-ops under a SNAPSHOT of the commit's own explored-pair table, which the host
//     mirrors to the device as deltas before every launch.  (b) is sound because the state a no-op tests only grows: a pair
//     (ke, kl, b) is a no-op iff (ke, kl) is explored and its flip is explored or carries a queued mark above b; explored
//     bits are never cleared and a mark is only replaced by a higher one or superseded by the explored bit - so a pair that
//     is a no-op under an OLDER state of the table is a no-op when the commit reaches it.  What remains travels as 24-byte
//     records (keys + indices), in pair order.
// Committed sequence, verdicts, prefix lengths, first violation: those of batch = 1 (tests: the CPU harness restates the
// device rules sequentially under this same loop; the GPU suite holds config 3 against the committed golden sequence).
// 64-bit key -> V, open addressing over indices into chunks of values: an insertion allocates nothing per element (a chunk of
// 4096 values now and then) and the values keep their addresses (the commit holds pointers to results across insertions).
// Key 0 is the empty slot.
template <class V>
class FlatKeyMap {
 public:
  FlatKeyMap() { slots_.assign(1u << 12, Slot{0, 0}); mask_ = slots_.size() - 1; }
  V* find(uint64_t key) {
    for (size_t i = hash(key) & mask_;; i = (i + 1) & mask_) {
      const Slot& s = slots_[i];
      if (s.key == key) return &val(s.idx);
      if (s.key == 0) return nullptr;
    }
  }
  bool count(uint64_t key) { return find(key) != nullptr; }
  // the value of `key`, value-initialised first if absent (`inserted` says which)
  V& at(uint64_t key, bool& inserted) {
    if ((n_ + 1) * 2 > slots_.size()) grow();
    for (size_t i = hash(key) & mask_;; i = (i + 1) & mask_) {
      Slot& s = slots_[i];
      if (s.key == key) { inserted = false; return val(s.idx); }
      if (s.key == 0) {
        if ((n_ & (CHUNK - 1)) == 0 && n_ / CHUNK == chunks_.size()) chunks_.emplace_back(new V[CHUNK]());
        s.key = key; s.idx = (uint32_t)n_;
        inserted = true;
        return val(n_++);
      }
    }
  }
  V& at(uint64_t key) { bool ins; return at(key, ins); }
  void prefetch(uint64_t key) const { __builtin_prefetch(&slots_[hash(key) & mask_]); }
  size_t size() const { return n_; }

 private:
  static constexpr size_t CHUNK = 4096;
  struct Slot { uint64_t key; uint32_t idx; };
  static size_t hash(uint64_t k) { k *= 0x9E3779B97F4A7C15ULL; return (size_t)(k ^ (k >> 29)); }
  V& val(size_t i) { return chunks_[i / CHUNK][i % CHUNK]; }
  void grow() {
    std::vector<Slot> old;
    old.swap(slots_);
    slots_.assign(old.size() * 2, Slot{0, 0});
    mask_ = slots_.size() - 1;
    for (const Slot& o : old)
      if (o.key) { size_t i = hash(o.key) & mask_; while (slots_[i].key) i = (i + 1) & mask_; slots_[i] = o; }
  }
  std::vector<Slot> slots_;
  size_t mask_ = 0, n_ = 0;
  std::vector<std::unique_ptr<V[]>> chunks_;
};

struct RefRec {              // one racing pair the commit still has to absorb (device -> host), 24 bytes
  unsigned long long ke, kl;
  uint8_t branch, later, earlier, pad;
  uint32_t pad2;
};
struct RefDelta {            // one entry of the commit's explored-pair table that changed since the last launch (host -> device)
  unsigned long long lo, hi; // the unordered pair, lo < hi
  uint32_t state[2];         // side 0 = (lo, hi), side 1 = (hi, lo): bit 31 explored, bits 0..8 = 1 + highest queued branch
};

// The commit's bookkeeping over records: one queue (256 FIFO buckets by branch, creation order within one), one table.
class RefBook {
 public:
  struct Point { unsigned long long flip_a, flip_b; uint32_t src; uint8_t branch, later, earlier, pad; };
  // dpor() for one committed interleaving (arena id `src`): its surviving racing pairs in pair order
  void absorb(const RefRec* r, uint32_t n, uint32_t src) {
    front_valid_ = false;
    constexpr uint32_t AHEAD = 8;          // the table is far larger than the caches: the next records' lines are requested while this one is applied
    for (uint32_t k = 0; k < n && k < AHEAD; k++) map_.prefetch(r[k].ke, r[k].kl);
    for (uint32_t k = 0; k < n; k++) {
      if (k + AHEAD < n) map_.prefetch(r[k + AHEAD].ke, r[k + AHEAD].kl);
      const FlatPairMap::Ref e = map_.at(r[k].ke, r[k].kl);
      uint32_t* const side0 = r[k].ke < r[k].kl ? e.fwd : e.rev;      // (the entry's first value: where the dirty mark lives)
      if (!(*e.fwd & EXPLORED)) touch(side0, r[k].ke, r[k].kl);
      *e.fwd |= EXPLORED;                                  // setExplored(branchI, (earlier, later)) (:1068-1070)
      uint32_t& flipped = *e.rev;
      if (flipped & EXPLORED) continue;                    // getNext would skip it (:1153-1157)
      if ((flipped & QUEUED_MASK) > r[k].branch) continue; // a queued point of this pair pops before it
      flipped = (flipped & ~QUEUED_MASK) | ((uint32_t)r[k].branch + 1);
      touch(side0, r[k].ke, r[k].kl);
      bucket_[r[k].branch].push_back(Point{r[k].kl, r[k].ke, src, r[k].branch, r[k].later, r[k].earlier, 0});
      if ((int)r[k].branch > top_) top_ = (int)r[k].branch;
      queued_++; enqueued_++;
    }
  }
  // getNext (:1142-1162): deepest branch first, creation order within a branch, explored flips skipped; marks it explored
  bool get_next(Point& out) {
    while (top_ >= 0) {
      std::deque<Point>& b = bucket_[top_];
      if (b.empty()) { top_--; continue; }
      const Point p = b.front();
      b.pop_front();
      queued_--;
      if (b.size() > 3) map_.prefetch(b[3].flip_a, b[3].flip_b);          // (dead points are skipped in runs: the line of the one after next)
      const FlatPairMap::Ref e = map_.at(p.flip_a, p.flip_b);
      if (*e.fwd & EXPLORED) continue;                     // isExplored: skip
      *e.fwd |= EXPLORED;                                  // setExplored(maxIndex, (e1, e2)) (:1170-1172)
      touch(p.flip_a < p.flip_b ? e.fwd : e.rev, p.flip_a, p.flip_b);
      out = p;
      return true;
    }
    return false;
  }
  uint64_t queue_len() const { return queued_; }
  uint64_t enqueued() const { return enqueued_; }
  // up to `k` points in dequeue order that are live right now, without dequeuing them: what the commit will most likely ask
  // for next (a speculation: a point may still die before its turn, and new points may get ahead of it).  The table lines of
  // the points further down the bucket are requested while this one is looked at.
  template <class F>
  void peek(size_t k, F&& f) const {
    constexpr size_t AHEAD = 12;
    for (int b = top_; b >= 0 && k; b--) {
      const std::deque<Point>& q = bucket_[b];
      const size_t n = q.size();
      for (size_t j = 0; j < n && j < AHEAD; j++) map_.prefetch(q[j].flip_a, q[j].flip_b);
      for (size_t i = 0; i < n && k; i++) {
        if (i + AHEAD < n) map_.prefetch(q[i + AHEAD].flip_a, q[i + AHEAD].flip_b);
        const Point& p = q[i];
        const uint32_t* v = map_.find(p.flip_a, p.flip_b);
        if (v && (*v & EXPLORED)) continue;
        f(p);
        k--;
      }
    }
  }
  // the point get_next() would look at first (live or not): its result is what the commit most likely needs after this one
  const Point* front() {
    while (top_ >= 0 && bucket_[top_].empty()) top_--;
    return top_ >= 0 ? &bucket_[top_].front() : nullptr;
  }
  // the entries that changed since the last call, with their current states
  void take_deltas(std::vector<RefDelta>& out) {
    out.clear();
    out.reserve(dirty_.size());
    constexpr size_t AHEAD = 12;
    const size_t n = dirty_.size();
    for (size_t i = 0; i < n && i < AHEAD; i++) map_.prefetch(dirty_[i].first, dirty_[i].second);
    for (size_t i = 0; i < n; i++) {
      if (i + AHEAD < n) map_.prefetch(dirty_[i + AHEAD].first, dirty_[i + AHEAD].second);
      const std::pair<uint64_t, uint64_t>& k = dirty_[i];
      const FlatPairMap::Ref e = map_.at(k.first, k.second);      // (lo, hi): fwd = side 0
      *e.fwd &= ~DIRTY;
      out.push_back(RefDelta{k.first, k.second, {*e.fwd & ~DIRTY, *e.rev & ~DIRTY}});
    }
    dirty_.clear();
  }

 private:
  static constexpr uint32_t EXPLORED = 0x80000000u, QUEUED_MASK = 0x1FFu, DIRTY = 0x40000000u;   // DIRTY lives on side 0 only
  // (no second table lookup here: a lookup may grow the table and move the entry the caller is still pointing at)
  void touch(uint32_t* side0, uint64_t a, uint64_t b) {
    if (!(*side0 & DIRTY)) { *side0 |= DIRTY; dirty_.push_back({a < b ? a : b, a < b ? b : a}); }
  }
  FlatPairMap map_;
  std::deque<Point> bucket_[256];
  int top_ = -1;
  bool front_valid_ = false;
  uint64_t queued_ = 0, enqueued_ = 0;
  std::vector<std::pair<uint64_t, uint64_t>> dirty_;
};

// dev.round_ref_begin(items, use_parent, n, round, base_id, deltas, n_deltas, speculate) /
// dev.round_ref_end(verdicts, points, kills, rec_cnt, &spec_overflow) / dev.round_ref_abort()   (one launch in flight at a time):
//   one launch like dev.round() of the ROUNDS path (K3 + the speculation's mark / insert / decide), plus: the deltas applied to
//   the device's copy of the commit's table first, and afterwards the commit filter - interleaving i keeps rec_cnt[i] racing
//   pairs as records, in pair order, WITH THE DEVICE (keyed by its arena id base_id + i); use_parent[i] says whether its
//   parent's trace may be used for (a); speculate = false leaves the speculation's table and points out.  begin enqueues it and returns; end waits and hands the results over (spec_overflow:
//   the speculation's own table or point area is full - its points are dropped, everything else stands); abort waits and
//   drops them.
// dev.ref_fetch_begin(ids, m, deltas, n_deltas) / dev.ref_fetch_end(rec_off, rec_cnt, &recs)   (one fetch in flight at a time):
//   the commit is about to absorb the interleavings `ids` (arena ids, the first one right now, the others probably next):
//   their records, filtered AGAIN under the table as it is now (the deltas first) - rule (b) holds for any older state, and a
//   launch is 10^4 interleavings wide, most of whose pairs the commit has made no-ops by the time it reaches them: a
//   16 384-wide launch keeps 83 % of the pairs, the fetch a few per cent of those.  recs[rec_off[j] .. + rec_cnt[j]) are
//   interleaving ids[j]'s survivors, in pair order (memory owned by dev, valid until the exploration ends).
// how many interleavings of the commit's queue front a record fetch covers besides the one the commit stands at
inline size_t ref_fetch_width() {
  const char* e = demi_host::knob("DEMI_DPOR_FETCH_WIDTH");
  const long v = e ? atol(e) : 0;
  return v > 0 ? (size_t)v : 128u;
}

template <class Dev>
int explore_reference_resident(Dev&& dev, const demi_dpor_search* srch, demi_verdict* out_verdicts, uint32_t* out_prefix_len,
                               uint32_t* out_rounds, demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                               demi_dpor_stats* stats, double* seconds, ExploredLog* log = nullptr) {
  memset(stats, 0, sizeof *stats);
  stats->first_violation = ~0ull;
  if (first_violation_len) *first_violation_len = 0;
  if (log) log->clear();
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  auto key_of = [](const demi::DporItem& it) -> uint64_t {
    return ((uint64_t)it.src << 24) | ((uint64_t)it.branch << 16) | ((uint64_t)it.later << 8) | (uint64_t)it.earlier;
  };
  // (`seconds`, a diagnostic, is 8 doubles here: [0..2] as in the other loops, [3] naming the fetches (host), [4] the device's part of them, [5] the host's work after a launch, [6] before one)
  // fetched: its records have been ASKED for (a fetch names it); ready: they are here (nothing to fetch: both from the start)
  struct Result { uint32_t id; demi_verdict verdict; const RefRec* recs; uint32_t rec_cnt; bool fetched, ready; };
  FlatKeyMap<Result> results;                            // every interleaving run so far, by its item; its surviving racing
                                                         // pairs stay where the device's copy put them (dev owns that memory)
  std::vector<uint8_t> complete;                         // per arena id: invariant (I) of ParentFilter holds (see there)
  RefBook real;
  // the speculation's queue (explore_rounds_resident's)
  std::deque<demi::DporPoint> bucket[256];
  int top = -1;
  FlatPairSet dead;

  const demi::DporItem first{0xFFFFFFFFu, 0, 0, 0, 0};
  demi::DporItem cur = first;
  bool have_cur = true, exhausted = false, done = false;
  std::vector<demi::DporItem> spec_items(1, first), items;
  std::vector<uint8_t> use_parent;
  std::vector<demi_verdict> vd;
  std::vector<demi::DporPoint> pts;
  std::vector<demi::DporKill> kills;
  std::vector<uint64_t> rec_off;
  std::vector<uint32_t> rec_cnt;
  std::vector<RefDelta> deltas;
  std::vector<uint32_t> fetch_ids;
  std::vector<Result*> fetch_res, inflight_res, launch_res;
  std::vector<uint64_t> peeked;
  std::vector<demi::DporItem> peeked_items;
  bool inflight = false;                                  // a record fetch has been issued and not yet landed
  // Measured (round 5, config 3, profiles/r05_call4_reference_prefetch_ab.txt): with the next window's fetch in flight the wait
  // shrinks 23 -> 19 ms, the records fetched a window early are filtered under an older table and grow 33 -> 63 MB, the commit
  // 13 -> 17 ms: 59.7 against 59.8 ms.  A window of the commit (25 us) is shorter than a fetch's round trip (40 us: two launches,
  // a PCIe read of the request, the answer's writes, the event), so ONE fetch ahead cannot hide it and each further one costs more
  // stale records.  Off by default (DEMI_DPOR_PREFETCH=1 turns it on); the split into begin / end stays.
  const bool no_prefetch = demi_host::knob("DEMI_DPOR_PREFETCH") == nullptr;
  // Measured and not kept (round 5, profiles/r05_call15_reference_eager_rounds_ab.txt): the speculation's next round started the
  // moment the last one is back, on a stream of its own beside the record fetches.  The commit is depth-first - after a handful
  // of steps it stands at a descendant that only the speculation's LAST rounds produce - so it waits for the rounds one after
  // the other all the same (13 ms of launches either way), and the rounds' kernels and the fetches slow each other down
  // (kernels 11.3 -> 13.5 ms): 1.10·10⁶/s against 1.09-1.12.  A launch is started when the commit lacks a result, and waited for.
  uint32_t base_id = 0, round = 0, fl_base = 0;
  bool spec_off = false;                                  // the speculation has stopped for good (its table is full)
  uint64_t first_id = ~0ull;
  const size_t fetch_width = ref_fetch_width();

  // start a launch: the interleaving the commit stands at + its queue's front + the speculation's next round, minus what has
  // been run already.  Every item gets its entry in `results` right away (filled in when the launch is back): one table
  // operation per item, and an interleaving that is already there is not run again.
  auto begin_launch = [&]() -> int {
    const double tb = now();
    items.clear(); launch_res.clear();
    auto add = [&](const demi::DporItem& q) {
      bool fresh;
      Result& r = results.at(key_of(q), fresh);
      if (fresh) { items.push_back(q); launch_res.push_back(&r); }
    };
    add(cur);
    // the commit's own queue front: what it will most likely dequeue next (the speculation explores in rounds and runs out
    // long before the commit does; without this every later interleaving would be a launch of its own)
    peeked_items.clear();
    real.peek(srch->batch / 4 + 1, [&](const RefBook::Point& p) {
      const demi::DporItem q{p.src, p.branch, p.later, p.earlier, 0};
      results.prefetch(key_of(q));
      peeked_items.push_back(q);
    });
    for (const demi::DporItem& q : peeked_items) add(q);
    {
      constexpr size_t AHEAD = 12;
      const size_t ns = spec_items.size();
      for (size_t i = 0; i < ns && i < AHEAD; i++) results.prefetch(key_of(spec_items[i]));
      for (size_t i = 0; i < ns; i++) {
        if (i + AHEAD < ns) results.prefetch(key_of(spec_items[i + AHEAD]));
        add(spec_items[i]);
      }
      spec_items.clear();
    }
    const uint32_t n = (uint32_t)items.size();
    if (!n) return 0;
    use_parent.resize(n);
    for (uint32_t i = 0; i < n; i++) use_parent[i] = items[i].src != 0xFFFFFFFFu && items[i].src < complete.size() && complete[items[i].src];
    real.take_deltas(deltas);                          // (the table's changes so far travel with the launch)
    round++;
    if (seconds) seconds[6] += now() - tb;
    int rc = dev.round_ref_begin(items.data(), use_parent.data(), n, round, base_id, deltas.data(), (uint32_t)deltas.size(),
                                 !spec_off && stats->executed < srch->max_interleavings);
    if (rc) return rc;
    if (out_rounds && stats->launches < srch->max_interleavings) out_rounds[stats->launches] = n;
    stats->launches++;
    stats->executed += n;
    fl_base = base_id;
    base_id += dev.ids_used(n);
    return 0;
  };
  // wait for the launch in flight and take its results in
  auto land_launch = [&]() -> int {
    const double tl = now();
    const uint32_t n = (uint32_t)items.size();
    vd.resize(n); rec_cnt.resize(n);
    pts.clear(); kills.clear();
    bool spec_overflow = false;
    int rc = dev.round_ref_end(vd.data(), pts, kills, rec_cnt.data(), &spec_overflow);
    if (rc) return rc;
    if (spec_overflow) {                     // the speculation's own table (or a round's point area) is full: it guesses no further
      spec_off = true;
      for (std::deque<demi::DporPoint>& b : bucket) b.clear();
      top = -1;
    }
    const double t2 = now();
    if (complete.size() < (size_t)fl_base + n) complete.resize((size_t)fl_base + n, 0);
    for (uint32_t i = 0; i < n; i++) {
      *launch_res[i] = Result{fl_base + i, vd[i], nullptr, 0u, rec_cnt[i] == 0, rec_cnt[i] == 0};   // (nothing to fetch: as good as here)
      // (I) holds for this interleaving once it is absorbed iff its own pair list is whole and (I) held for its parent
      complete[(size_t)fl_base + i] = !(vd[i].flags & DEMI_V_PAIRS_OVF) && (items[i].src == 0xFFFFFFFFu || use_parent[i]);
    }
    // the speculation: this round's live points and kills into its queue, then its next round
    {
      constexpr size_t AHEAD = 12;
      const size_t nk = kills.size();
      for (size_t i = 0; i < nk && i < AHEAD; i++) dead.prefetch(kills[i].a, kills[i].b);
      for (size_t i = 0; i < nk; i++) {
        if (i + AHEAD < nk) dead.prefetch(kills[i + AHEAD].a, kills[i + AHEAD].b);
        dead.insert(kills[i].a, kills[i].b);
      }
    }
    std::sort(pts.begin(), pts.end(), [](const demi::DporPoint& x, const demi::DporPoint& y) { return x.ordinal < y.ordinal; });
    for (const demi::DporPoint& p : pts) {
      bucket[p.branch].push_back(p);
      if ((int)p.branch > top) top = (int)p.branch;
    }
    // (a budgeted exploration: once as many interleavings have been run as the commit may take, the speculation stops - what the
    // commit still lacks comes from its own queue's front)
    while (!spec_off && spec_items.size() < srch->batch && stats->executed + spec_items.size() < srch->max_interleavings) {
      while (top >= 0 && bucket[top].empty()) top--;
      if (top < 0) break;
      const demi::DporPoint p = bucket[top].front();
      bucket[top].pop_front();
      if (bucket[top].size() > 8) dead.prefetch(bucket[top][8].flip_a, bucket[top][8].flip_b);
      if (!dead.insert(p.flip_a, p.flip_b)) continue;
      spec_items.push_back(demi::DporItem{p.src, p.branch, p.later, p.earlier, 0});
    }
    if (seconds) { const double t3 = now(); seconds[0] += t2 - tl; seconds[1] += t3 - t2; seconds[5] += t3 - t2; }
    return 0;
  };

  while (!done) {
    // ---- commit, one interleaving at a time, as far as computed results reach
    double t0 = now();
    bool need_fetch = false;
    while (have_cur) {
      const Result* it = results.find(key_of(cur));
      if (!it) break;
      if (!it->ready) { need_fetch = true; break; }
      const Result& r = *it;
      const uint64_t idx = stats->interleavings++;
      out_verdicts[idx] = r.verdict;
      out_prefix_len[idx] = cur.src == 0xFFFFFFFFu ? 0u : (uint32_t)cur.later;
      if (log) log->add(cur, r.id);
      bool found = false;
      if (r.verdict.flags & DEMI_V_VIOLATION) {
        stats->violations++;
        found = true;
        if (stats->first_violation == ~0ull) { stats->first_violation = idx; first_id = r.id; }
      }
      if (const RefBook::Point* f = real.front())           // (most likely the next one: its result's line, while this one's records are absorbed)
        results.prefetch(key_of(demi::DporItem{f->src, f->branch, f->later, f->earlier, 0}));
      real.absorb(r.recs, r.rec_cnt, r.id);
      if ((srch->stop_if_violation && found) || stats->interleavings >= srch->max_interleavings) { done = true; break; }
      RefBook::Point p;
      have_cur = real.get_next(p);
      if (!have_cur) { exhausted = true; done = true; break; }
      cur = demi::DporItem{p.src, p.branch, p.later, p.earlier, 0};
    }
    double t1 = now();
    if (seconds) seconds[2] += t1 - t0;
    if (done) break;

    // A record fetch is two steps: issue() names the interleavings and hands the device the table's changes, land() waits for the
    // answer.  With DEMI_DPOR_PREFETCH the fetch for the NEXT window of the queue front is issued while the commit works through
    // this one (records fetched a window early are filtered under a table that is a window older - rule (b) holds for any older
    // state - so more of them cross PCIe); by default a fetch is waited for when it is issued (see no_prefetch above).
    double t_named = t1;
    auto issue = [&](bool with_cur) -> int {
      fetch_ids.clear(); fetch_res.clear();
      if (with_cur) {
        Result* r0 = results.find(key_of(cur));                // (FlatKeyMap: element addresses survive insertions)
        r0->fetched = true;
        fetch_ids.push_back(r0->id); fetch_res.push_back(r0);
      }
      // (with_cur: the window at the queue's front; else the window behind it - the front one has been asked for already)
      peeked.clear();
      real.peek(with_cur ? fetch_width : 2 * fetch_width, [&](const RefBook::Point& p) {
        const uint64_t k = key_of(demi::DporItem{p.src, p.branch, p.later, p.earlier, 0});
        results.prefetch(k);
        peeked.push_back(k);
      });
      for (const uint64_t k : peeked) {
        Result* it = results.find(k);
        if (!it || it->fetched) continue;
        it->fetched = true;
        fetch_ids.push_back(it->id); fetch_res.push_back(it);
      }
      if (fetch_ids.empty()) return 0;
      real.take_deltas(deltas);
      t_named = now();
      int rc = dev.ref_fetch_begin(fetch_ids.data(), (uint32_t)fetch_ids.size(), deltas.data(), (uint32_t)deltas.size());
      if (rc) return rc;
      inflight_res = fetch_res;
      inflight = true;
      return 0;
    };
    auto land = [&]() -> int {
      if (!inflight) return 0;
      const uint32_t m = (uint32_t)inflight_res.size();
      rec_off.resize(m); rec_cnt.resize(m);
      const RefRec* recs = nullptr;
      int rc = dev.ref_fetch_end(rec_off.data(), rec_cnt.data(), &recs);
      if (rc) return rc;
      for (uint32_t j = 0; j < m; j++) { inflight_res[j]->recs = recs + rec_off[j]; inflight_res[j]->rec_cnt = rec_cnt[j]; inflight_res[j]->ready = true; }
      inflight = false;
      stats->fetches++;
      return 0;
    };
    if (need_fetch) {
      // ---- the records of the interleaving the commit stands at: in the fetch that is in flight (the usual case), or asked for now
      int rc = land();
      if (rc) return rc;
      const Result& r0 = *results.find(key_of(cur));
      if (!r0.ready) {
        const double t_i0 = now();
        rc = issue(!r0.fetched);
        if (!rc) rc = land();
        if (rc) return rc;
        if (seconds) { seconds[3] += t_named - t_i0; seconds[4] += now() - t_named; }      // (diagnostic: naming the fetch / the device's part of it)
      }
      // ... and, while the commit absorbs them, those of the interleavings its queue will most likely hand out after this window
      if (!no_prefetch) { rc = issue(false); if (rc) return rc; }
      if (seconds) seconds[1] += now() - t1;
      continue;
    }
    // ---- one launch: what the commit is waiting for + its queue's front + the speculation's next round (minus what has been
    // run already)
    { int rc = land(); if (rc) return rc; }         // (the launch sends the table's changes too)
    stats->cache_misses++;
    int rc = begin_launch();
    if (!rc) rc = land_launch();
    if (rc) { dev.round_ref_abort(); return rc; }          // (nothing of a failed launch stays in flight)
  }
  if (inflight) {                                          // (an answer nobody needs any more: still waited for - it writes host memory)
    rec_off.resize(inflight_res.size()); rec_cnt.resize(inflight_res.size());
    const RefRec* recs = nullptr;
    int rc = dev.ref_fetch_end(rec_off.data(), rec_cnt.data(), &recs);
    if (rc) return rc;
    inflight = false;
  }
  if (first_id != ~0ull && first_violation_trace && first_violation_len) {
    int rc = dev.fetch_trace((uint32_t)first_id, first_violation_trace, first_violation_len);
    if (rc) return rc;
  }
  stats->queue_len = real.queue_len();
  stats->backtrack_points = real.enqueued();
  stats->exhausted = exhausted ? 1u : 0u;
  return 0;
}

template <class Run, class Fetch>
int explore_loop(Run&& run, Fetch&& fetch, uint32_t max_pairs, const demi_dpor_search* srch, demi_verdict* out_verdicts,
                 uint32_t* out_prefix_len, uint32_t* out_rounds, demi_dpor_trace_entry* first_violation_trace,
                 uint32_t* first_violation_len, demi_dpor_stats* stats, double* seconds,
                 RawBuf* trace_buf = nullptr, RawBuf* pair_buf = nullptr) {
  if (srch->order == DEMI_DPOR_ORDER_REFERENCE)
    return explore_reference_order(run, fetch, max_pairs, srch, out_verdicts, out_prefix_len, out_rounds, first_violation_trace,
                                   first_violation_len, stats, seconds, trace_buf, pair_buf);
  return explore_rounds(run, fetch, max_pairs, srch, out_verdicts, out_prefix_len, out_rounds, first_violation_trace,
                        first_violation_len, stats, seconds, trace_buf, pair_buf);
}

}  // namespace demi_host
