// dpor_host.hpp — host-side bookkeeping of the DPOR exploration (no device code): the backtrack
// priority queue with DefaultBacktrackOrdering (BacktrackOrdering.scala:58-69), the ExploredTacker
// (AuxilaryTypes.scala:209-246), dpor()'s enqueue (:1068-1070, 1134) and getNext() (:1142-1185).
//
// A raft5 interleaving yields ~1.7 k racing pairs, so this bookkeeping, not the kernel, bounds the
// exploration rate.  Three things keep it cheap and exactly equal to the one-at-a-time loop:
//  * every operation on a racing pair touches only the explored-set entries (a, b) and (b, a), so the
//    state is sharded by the unordered pair {a, b}; a round's pairs are bucketed by shard (in
//    parallel, by contiguous ranges of interleavings) and each shard is then processed by one thread
//    in global pair order;
//  * the explored set is an open-addressing table of 16-byte keys (no allocation per insert);
//  * DefaultBacktrackOrdering only compares the branch index (< 256) and PriorityQueue ties are pinned
//    to creation order, so the queue is 256 FIFO buckets per shard: push O(1); the global pop takes,
//    in the highest non-empty branch, the front with the smallest global pair ordinal.
#pragma once

#include <cstdint>
#include <cstring>
#include <deque>
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

// open addressing, linear probing; key (0, 0) is the empty slot (node keys are FNV hash chains: never 0, 0)
class FlatPairSet {
 public:
  FlatPairSet() { resize(1u << 12); }
  bool contains(uint64_t a, uint64_t b) const {
    for (size_t i = slot(a, b);; i = (i + 1) & mask_) {
      const Key& k = tab_[i];
      if (k.a == a && k.b == b) return true;
      if (k.a == 0 && k.b == 0) return false;
    }
  }
  void insert(uint64_t a, uint64_t b) {
    if ((n_ + 1) * 5 > (mask_ + 1) * 3) grow();
    for (size_t i = slot(a, b);; i = (i + 1) & mask_) {
      Key& k = tab_[i];
      if (k.a == a && k.b == b) return;
      if (k.a == 0 && k.b == 0) { k.a = a; k.b = b; n_++; return; }
    }
  }
  size_t size() const { return n_; }

 private:
  struct Key { uint64_t a, b; };
  size_t slot(uint64_t a, uint64_t b) const {
    return (size_t)(((a * 0x9E3779B97F4A7C15ULL) ^ (b * 0xC2B2AE3D27D4EB4FULL) ^ (a >> 29)) >> 7) & mask_;
  }
  void resize(size_t cap) { tab_.assign(cap, Key{0, 0}); mask_ = cap - 1; n_ = 0; }
  void grow() {
    std::vector<Key> old;
    old.swap(tab_);
    resize((mask_ + 1) * 2);
    for (const Key& k : old)
      if (k.a || k.b) insert(k.a, k.b);
  }
  std::vector<Key> tab_;
  size_t mask_ = 0, n_ = 0;
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
          pieces_[t * S + shard_of(ke, kl)].push_back(BtPoint{base[i] + k, tid[i], pp[k].branch, pp[k].later, pp[k].earlier, 0});
        }
      }
    };
    // phase 2: thread t processes its shards; the pieces of a shard are read in thread (= interleaving) order
    auto process = [&](unsigned t) {
      for (size_t s = t; s < S; s += threads_) {
        Shard& sh = shards_[s];
        for (unsigned src = 0; src < threads_; src++) {
          for (const BtPoint& p : pieces_[src * S + s]) {
            if (track_) {
              const Trace& trc = *traces_[p.trace_id];
              const uint64_t ke = trc[p.earlier].key, kl = trc[p.later].key;
              sh.explored.insert(ke, kl);                        // setExplored(branchI, (earlier, later))
              if (sh.explored.contains(kl, ke)) continue;         // getNext would skip it (:1153-1157)
            }
            sh.bucket[p.branch].push_back(p);
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
    for (;;) {
      int best = -1, best_branch = -1;
      uint64_t best_seq = 0;
      for (size_t s = 0; s < shards_.size(); s++) {
        Shard& sh = shards_[s];
        while (sh.top >= 0 && sh.bucket[sh.top].empty()) sh.top--;
        if (sh.top < 0) continue;
        const uint64_t sq = sh.bucket[sh.top].front().seq;
        if (sh.top > best_branch || (sh.top == best_branch && sq < best_seq)) { best = (int)s; best_branch = sh.top; best_seq = sq; }
      }
      if (best < 0) return false;
      Shard& sh = shards_[best];
      const BtPoint p = sh.bucket[best_branch].front();
      sh.bucket[best_branch].pop_front();
      sh.queued--;
      const Trace& src = *traces_[p.trace_id];
      if (track_) {
        const uint64_t ke = src[p.earlier].key, kl = src[p.later].key;
        if (sh.explored.contains(kl, ke)) continue;
        sh.explored.insert(kl, ke);                              // setExplored(maxIndex, (e1, e2)) (:1170-1172)
      }
      out.assign(src.begin(), src.begin() + p.branch + 1);
      for (int k = (int)p.branch + 1; k <= (int)p.later; k++)
        if (k != (int)p.earlier) out.push_back(src[k]);
      return true;
    }
  }

  bool empty() const { return queue_len() == 0; }
  uint64_t queue_len() const {
    uint64_t n = 0;
    for (auto& s : shards_) n += s.queued;
    return n;
  }

 private:
  struct Shard {
    FlatPairSet explored;                    // ExploredTacker restricted to this shard's pairs
    std::deque<BtPoint> bucket[256];         // backTrack, one FIFO per branch index
    int top = -1;
    uint64_t queued = 0;
  };
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
  std::vector<std::vector<BtPoint>> pieces_;            // [thread][shard] buckets of the current round
  std::vector<std::shared_ptr<Trace>> traces_;
};

}  // namespace demi_host
