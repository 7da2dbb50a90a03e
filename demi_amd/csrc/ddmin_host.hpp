// ddmin_host.hpp — DDMin (minification/DeltaDebugging.scala:7-110) natively on the host, around K2 launches.
//
// What runs here is host control: DDMin.minimize / ddmin2 (:27-109) over an EventDag of external events with its atomic
// events (minification/Util.scala:46-63, 161-304) and MinificationUtil.split_list (:9-37), as RunnerUtils.stsSchedDDMin
// (RunnerUtils.scala:642-707) sets them up.  ddmin2 is a sequential decision tree that consults its oracle once per node;
// one consultation is one STSScheduler.test, i.e. one K2 lane.  The tree is therefore evaluated SPECULATIVELY: at a
// consultation whose candidate has no verdict yet, every candidate the next levels could ask for (all outcomes) is
// enumerated, they are replayed in ONE launch, and the real path then walks through the cached verdicts.  The MCS and the
// sequence of consultations are those of the sequential algorithm (tests: the same loop over the CPU oracle against the Python
// mirror demi_amd/minification.py, and the GPU against both).
//
// This is the loop a JVM host would otherwise run around demi_replay_batch; in the Python mirror the enumeration of a few
// thousand candidates costs more than the launch that replays them.
#pragma once
#include "knobs.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "../../include/demi_gpu.h"

namespace demi_host {

struct Mask256 {
  uint64_t w[4] = {0, 0, 0, 0};
  bool get(uint32_t i) const { return (w[i >> 6] >> (i & 63)) & 1ull; }
  void set(uint32_t i) { w[i >> 6] |= 1ull << (i & 63); }
  void clear(uint32_t i) { w[i >> 6] &= ~(1ull << (i & 63)); }
  uint32_t count() const { return (uint32_t)(__builtin_popcountll(w[0]) + __builtin_popcountll(w[1]) + __builtin_popcountll(w[2]) + __builtin_popcountll(w[3])); }
  bool operator==(const Mask256& o) const { return w[0] == o.w[0] && w[1] == o.w[1] && w[2] == o.w[2] && w[3] == o.w[3]; }
  Mask256 operator|(const Mask256& o) const { Mask256 r; for (int k = 0; k < 4; k++) r.w[k] = w[k] | o.w[k]; return r; }
  bool disjoint(const Mask256& o) const { return !((w[0] & o.w[0]) | (w[1] & o.w[1]) | (w[2] & o.w[2]) | (w[3] & o.w[3])); }
};
struct Mask256Hash {
  size_t operator()(const Mask256& m) const {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (int k = 0; k < 4; k++) { h ^= m.w[k]; h *= 0x100000001B3ULL; h ^= h >> 29; }
    return (size_t)h;
  }
};

// candidate -> small value, open addressing (the frontier enumeration does a few lookups per candidate: no node allocations)
class MaskTable {
 public:
  explicit MaskTable(size_t cap = 1u << 12) { keys_.resize(cap); vals_.assign(cap, -1); }
  // value of m, -1 if absent
  int get(const Mask256& m) const {
    for (size_t i = Mask256Hash()(m) & (keys_.size() - 1);; i = (i + 1) & (keys_.size() - 1)) {
      if (vals_[i] < 0) return -1;
      if (keys_[i] == m) return vals_[i];
    }
  }
  // true if m was absent (and is now present with value v); an existing value is overwritten
  bool put(const Mask256& m, int v) {
    if ((n_ + 1) * 2 > keys_.size()) grow();
    for (size_t i = Mask256Hash()(m) & (keys_.size() - 1);; i = (i + 1) & (keys_.size() - 1)) {
      if (vals_[i] < 0) { keys_[i] = m; vals_[i] = (int8_t)v; n_++; return true; }
      if (keys_[i] == m) { vals_[i] = (int8_t)v; return false; }
    }
  }
  void clear() { std::fill(vals_.begin(), vals_.end(), (int8_t)-1); n_ = 0; }
  size_t size() const { return n_; }

 private:
  void grow() {
    std::vector<Mask256> k; std::vector<int8_t> v;
    k.swap(keys_); v.swap(vals_);
    keys_.resize(k.size() * 2); vals_.assign(k.size() * 2, -1); n_ = 0;
    for (size_t i = 0; i < k.size(); i++) if (v[i] >= 0) put(k[i], v[i]);
  }
  std::vector<Mask256> keys_;
  std::vector<int8_t> vals_;
  size_t n_ = 0;
};

// AtomicEvent (Util.scala:46-63): external events that are removed together, by index, first index first
typedef std::vector<uint8_t> Atom;

// UnmodifiedEventDag / EventDagView (Util.scala:161-304).  get_atomic_events (:197-265): explicit pairs first; a Kill pairs with the
// remembered Start of the same actor, an UnPartition with the remembered Partition of the same ordered pair; the rest are
// singletons; sorted by first index.  ddmin2 only ever asks for the atoms of a view that is a union of whole atoms of the view
// it started from (it removes atoms, never single events), and for such a view the pairing rule yields exactly the starting
// view's atoms that lie inside it.  So the atoms are computed ONCE, a view is a set of atom numbers, its atoms in order are
// its set bits, and split_list is a popcount (the CPU suite holds this loop against the Python mirror, which recomputes the
// atoms of every view as the reference does).
class DdminDag {
 public:
  DdminDag(const demi_ext_event* ext, uint32_t n, const uint8_t* conjoined) : ext_(ext, ext + n), conj_(n, 255) {
    if (conjoined) for (uint32_t i = 0; i < n; i++) conj_[i] = conjoined[i];
  }
  // the atoms of `given`; false where the reference throws / trips its assumption (Kill without Start, atoms not a partition)
  bool atoms(const Mask256& given, std::vector<Atom>& out) const {
    out.clear();
    const uint32_t n = (uint32_t)ext_.size();
    std::vector<uint8_t> done(n, 0);
    for (uint32_t e = 0; e < n; e++) {
      if (!given.get(e) || conj_[e] == 255 || done[e]) continue;
      const uint32_t o = conj_[e];
      if (o >= n || !given.get(o)) return false;
      out.push_back(Atom{(uint8_t)e, (uint8_t)o});
      done[e] = done[o] = 1;
    }
    int start_of[DEMI_MAX_ACTORS_BIG];
    int part_of[DEMI_MAX_ACTORS_BIG * DEMI_MAX_ACTORS_BIG];
    for (int& x : start_of) x = -1;
    for (int& x : part_of) x = -1;
    for (uint32_t e = 0; e < n; e++) {
      if (!given.get(e) || conj_[e] != 255) continue;
      const demi_ext_event& x = ext_[e];
      const uint32_t a = x.a % DEMI_MAX_ACTORS_BIG, b = x.b % DEMI_MAX_ACTORS_BIG;
      if (x.kind == DEMI_EV_KILL) {
        if (start_of[a] < 0) return false;                 // "Kill without preceding Start"
        out.push_back(Atom{(uint8_t)start_of[a], (uint8_t)e});
        start_of[a] = -1;
      } else if (x.kind == DEMI_EV_PARTITION) {
        part_of[a * DEMI_MAX_ACTORS_BIG + b] = (int)e;         // (a later Partition of the same pair overwrites: caught below)
      } else if (x.kind == DEMI_EV_START) {
        start_of[a] = (int)e;
      } else if (x.kind == DEMI_EV_UNPARTITION) {
        int& p = part_of[a * DEMI_MAX_ACTORS_BIG + b];
        if (p < 0) return false;                           // "UnPartition without preceding Partition"
        out.push_back(Atom{(uint8_t)p, (uint8_t)e});
        p = -1;
      } else {
        out.push_back(Atom{(uint8_t)e});
      }
    }
    for (int x : start_of) if (x >= 0) out.push_back(Atom{(uint8_t)x});
    for (int x : part_of) if (x >= 0) out.push_back(Atom{(uint8_t)x});
    uint32_t flat = 0;
    for (const Atom& a : out) flat += (uint32_t)a.size();
    if (flat != given.count()) return false;               // assume(atomics.flatten.length == given_events.length)
    std::sort(out.begin(), out.end(), [](const Atom& x, const Atom& y) { return x[0] < y[0]; });
    return true;
  }

 private:
  std::vector<demi_ext_event> ext_;
  std::vector<uint8_t> conj_;
};

// the k lowest set bits of v
inline Mask256 mask_first_k(const Mask256& v, uint32_t k) {
  Mask256 r;
  for (int w = 0; w < 4 && k; w++) {
    const uint32_t c = (uint32_t)__builtin_popcountll(v.w[w]);
    if (k >= c) { r.w[w] = v.w[w]; k -= c; continue; }
    uint64_t x = v.w[w], t = 0;
    for (; k; k--) { t |= x & (~x + 1); x &= x - 1; }
    r.w[w] = t;
  }
  return r;
}

struct DdminOutcome {
  Mask256 mcs;                                      // over the external events
  uint32_t consultations = 0, launches = 0, total_inputs_pruned = 0, verified = 0;
  uint64_t replays = 0;
  std::vector<uint32_t> batches;                    // candidates per launch
  std::vector<std::pair<Mask256, bool>> consulted;  // (candidate over the external events, passes) in consultation order
};

// test(masks [n][4], n, reproduced [n]) -> demi_status: reproduced[i] = 1 iff candidate i still triggers the violation.
// Inside, a view / candidate is a set of ATOM numbers (see DdminDag); masks over the external events only leave towards the
// oracle and the caller.
template <class TestBatch>
class SpeculativeDdmin {
 public:
  // sequential: no speculation and no verdict cache - every consultation of ddmin2 is one call of the oracle with that one
  // candidate, as the reference's DDMin does it.  For oracles whose consultation is expensive and has a memory of its own
  // (ResumableDPOR: a DPOR exploration per consultation, continued when the same subsequence is asked again; incddmin_host.hpp)
  SpeculativeDdmin(const DdminDag& dag, TestBatch& test, uint32_t depth, uint32_t max_candidates, bool sequential = false)
      : dag_(dag), test_(test), depth_(depth), budget_(max_candidates ? max_candidates : 4096u), sequential_(sequential) {}
  double oracle_s_ = 0;           // time spent inside the oracle (the launches); the rest of minimize() is this loop

  // DDMin.minimize (:27-46) on the events `view`, then verify_mcs (:48-51) if asked; returns a demi_status
  int minimize(const Mask256& view, bool check_unmodified, bool verify, DdminOutcome* out) {
    out_ = out;
    rc_ = DEMI_OK;
    std::vector<Atom> atoms;
    if (!dag_.atoms(view, atoms) || atoms.size() > 256) return DEMI_ERR_INVALID_TRACE;
    atom_events_.assign(atoms.size(), Mask256());
    singletons_ = true;
    for (size_t i = 0; i < atoms.size(); i++) {
      for (uint8_t e : atoms[i]) atom_events_[i].set(e);
      singletons_ = singletons_ && atoms[i].size() == 1;
    }
    if (singletons_) root_ = view;                 // every atom is one event: a set of atoms IS the set of events
    else for (size_t i = 0; i < atoms.size(); i++) root_.set((uint32_t)i);
    pending_check_ = check_unmodified;
    if (check_unmodified && depth_) {               // (with a fixed depth the check is a launch of its own, as in the mirror)
      rc_ = launch(std::vector<Mask256>(1, root_));
      if (rc_) return rc_;
      pending_check_ = false;
      if (cache_.get(root_) == 1) return DEMI_ERR_INVALID_ARG;   // "Unmodified trace does not trigger violation"
    }
    const Mask256 mcs = ddmin2(root_, Mask256());
    if (rc_) return rc_;
    if (pending_check_) {                           // (nothing was ever consulted: a single atom)
      rc_ = launch(std::vector<Mask256>(1, root_));
      if (rc_) return rc_;
      if (cache_.get(root_) == 1) return DEMI_ERR_INVALID_ARG;
    }
    out_->mcs = events_of(mcs);
    if (verify) {
      if (cache_.get(mcs) < 0 || depth_) {          // (a verdict the search already has is not replayed again)
        rc_ = launch(std::vector<Mask256>(1, mcs));
        if (rc_) return rc_;
      }
      out_->verified = cache_.get(mcs) == 0 ? 1u : 0u;
    }
    return rc_;
  }

 private:
  static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  Mask256 events_of(const Mask256& atoms) const {
    if (singletons_) return atoms;
    Mask256 r;
    for (int w = 0; w < 4; w++)
      for (uint64_t x = atoms.w[w]; x; x &= x - 1) {
        const Mask256& m = atom_events_[(size_t)w * 64 + (size_t)__builtin_ctzll(x)];
        for (int k = 0; k < 4; k++) r.w[k] |= m.w[k];
      }
    return r;
  }
  // MinificationUtil.split_list(atoms, 2) + `dag.remove_events(split)` + `.reverse` (DeltaDebugging.scala:80-83): splits[0] is the
  // dag without the SECOND half of its atoms (the first chunk gets the extra element), splits[1] the dag without the first half
  static void split(const Mask256& dag, uint32_t n, Mask256 splits[2]) {
    splits[0] = mask_first_k(dag, n / 2 + (n % 2));
    for (int k = 0; k < 4; k++) splits[1].w[k] = dag.w[k] & ~splits[0].w[k];
  }
  int launch(const std::vector<Mask256>& cands) {
    std::vector<Mask256> ev(cands.size());
    for (size_t i = 0; i < cands.size(); i++) ev[i] = events_of(cands[i]);
    std::vector<uint8_t> reproduced(cands.size(), 0);
    const double t = now_s();
    const int rc = test_(ev[0].w, (uint32_t)ev.size(), reproduced.data());
    oracle_s_ += now_s() - t;
    if (rc) return rc;
    out_->launches++; out_->replays += cands.size(); out_->batches.push_back((uint32_t)cands.size());
    for (size_t i = 0; i < cands.size(); i++) cache_.put(cands[i], reproduced[i] ? 0 : 1);
    return DEMI_OK;
  }

  // every candidate the next `depth` levels below (dag, remainder) could consult, whatever the outcomes (known verdicts prune)
  void frontier(const Mask256& dag, const Mask256& remainder, uint32_t depth, std::vector<Mask256>& out) {
    const uint32_t n = dag.count();
    if (n <= 1 || depth == 0) return;
    Mask256 splits[2];
    split(dag, n, splits);
    int known[2];
    for (int k = 0; k < 2; k++) {
      const Mask256 c = splits[k] | remainder;
      known[k] = cache_.get(c);                                          // 1 = passes, 0 = fails, -1 = unknown
      if (known[k] < 0 && seen_.put(c, 1)) out.push_back(c);
    }
    if (known[0] != 1) frontier(splits[0], remainder, depth - 1, out);                       // left may fail
    if (known[0] != 0 && known[1] != 1) frontier(splits[1], remainder, depth - 1, out);      // left passes, right may fail
    if (known[0] != 0 && known[1] != 0) {                                                          // both may pass: interference
      frontier(splits[0], splits[1] | remainder, depth - 1, out);
      frontier(splits[1], splits[0] | remainder, depth - 1, out);
    }
  }
  // The walk from the ROOT of the decision tree, level by level, where a step whose verdicts are known costs nothing: it
  // follows the path the algorithm has taken so far and fans out the UNKNOWN levels beyond it - wherever ddmin2 goes next, also
  // into the sibling subtrees it returns to (`right` after `left`), which a frontier below the current node cannot see.
  // expand() resolves one node as far as the known verdicts reach; where a verdict is missing it emits the node's candidates
  // and queues the nodes the possible outcomes lead to for the next level.
  struct Node { Mask256 dag, rem; };
  void expand(const Mask256& dag, const Mask256& remainder, std::vector<Mask256>& out, std::vector<Node>& next) {
    const uint32_t n = dag.count();
    if (n <= 1) return;
    Mask256 splits[2], c[2];
    split(dag, n, splits);
    int known[2];
    for (int k = 0; k < 2; k++) { c[k] = splits[k] | remainder; known[k] = cache_.get(c[k]); }
    const bool unknown = known[0] < 0 || (known[0] == 1 && known[1] < 0);     // this node needs a verdict that is not there yet
    if (known[0] < 0 && seen_.put(c[0], 1)) out.push_back(c[0]);
    if (known[0] != 0 && known[1] < 0 && seen_.put(c[1], 1)) out.push_back(c[1]);
    auto go = [&](const Mask256& d, const Mask256& r) { if (unknown) next.push_back(Node{d, r}); else expand(d, r, out, next); };
    if (known[0] != 1) go(splits[0], remainder);                                  // left may fail -> ddmin2(left, remainder)
    if (known[0] != 0 && known[1] != 1) go(splits[1], remainder);                 // left passes, right may fail
    if (known[0] != 0 && known[1] != 0) {                                         // both may pass: interference
      go(splits[0], splits[1] | remainder);
      go(splits[1], splits[0] | remainder);
    }
  }

  bool passes(const Mask256& cand) {
    if (rc_) return true;
    if (sequential_) {
      rc_ = launch(std::vector<Mask256>(1, cand));
      if (rc_) return true;
    } else if (cache_.get(cand) < 0) {
      std::vector<Mask256> cands;
      if (depth_) {
        seen_.clear();
        cands.push_back(cand); seen_.put(cand, 1);
        frontier(node_dag_, node_rem_, depth_, cands);
      } else {
        // as many unknown levels as fit the launch: a level more multiplies the candidates by up to four, and the launch's
        // time is the serial chain of one replay whatever its width (up to the chip's resident lanes)
        seen_.clear();
        cands.push_back(cand); seen_.put(cand, 1);
        if (pending_check_ && seen_.put(root_, 1)) cands.push_back(root_);        // (the unmodified trace rides along)
        std::vector<Node> open(1, Node{root_, Mask256()}), next;
        for (bool first = true; !open.empty(); first = false) {
          if (!first && cands.size() + 2 * open.size() > budget_) break;          // (a node asks for at most two candidates)
          next.clear();
          for (const Node& nd : open) expand(nd.dag, nd.rem, cands, next);
          open.swap(next);
        }
      }
      rc_ = launch(cands);
      if (rc_) return true;
      if (pending_check_ && !depth_) {
        pending_check_ = false;
        if (cache_.get(root_) == 1) { rc_ = DEMI_ERR_INVALID_ARG; return true; }     // "Unmodified trace does not trigger violation"
      }
    }
    const bool p = cache_.get(cand) == 1;
    out_->consultations++;
    out_->consulted.push_back({events_of(cand), p});
    return p;
  }

  // ddmin2 (:73-109)
  Mask256 ddmin2(const Mask256& dag, const Mask256& remainder) {
    const uint32_t n = dag.count();
    if (rc_ || n <= 1) return dag;
    Mask256 splits[2];
    split(dag, n, splits);
    for (int k = 0; k < 2; k++) {
      node_dag_ = dag; node_rem_ = remainder;
      if (!passes(splits[k] | remainder)) {
        out_->total_inputs_pruned += events_of(dag).count() - events_of(splits[k]).count();
        return ddmin2(splits[k], remainder);
      }
      if (rc_) return dag;
    }
    const Mask256 left = ddmin2(splits[0], splits[1] | remainder);
    const Mask256 right = ddmin2(splits[1], splits[0] | remainder);
    return left | right;
  }

  const DdminDag& dag_;
  TestBatch& test_;
  uint32_t depth_, budget_;
  bool sequential_ = false;
  std::vector<Mask256> atom_events_;             // the external events of atom i
  bool singletons_ = true, pending_check_ = false;
  MaskTable cache_;                              // candidate (atom numbers) -> 1 passes / 0 fails
  MaskTable seen_;                               // the candidates of the launch being put together
  Mask256 node_dag_, node_rem_, root_;
  DdminOutcome* out_ = nullptr;
  int rc_ = DEMI_OK;
};

// RunnerUtils.stsSchedDDMin (:642-707): WaitQuiescence stripped from the externals, DDMin over the replay oracle, the MCS
// verified.  out_consulted / out_passed (may be NULL) receive the first `cap` consultations.
template <class TestBatch>
int sts_sched_ddmin(const demi_ext_event* ext, uint32_t n_ext, const uint8_t* conjoined, const demi_ddmin_params* par, TestBatch&& test,
                    uint64_t out_mcs[4], uint64_t* out_consulted, uint8_t* out_passed, uint32_t cap, uint32_t* out_batches,
                    uint32_t batches_cap, demi_ddmin_stats* stats) {
  memset(stats, 0, sizeof *stats);
  DdminDag dag(ext, n_ext, conjoined);
  Mask256 view;
  for (uint32_t i = 0; i < n_ext; i++) if (ext[i].kind != DEMI_EV_WAIT_QUIESCENCE) view.set(i);
  DdminOutcome o;
  SpeculativeDdmin<typename std::remove_reference<TestBatch>::type> dd(dag, test, par->depth, par->max_candidates);
  const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  int rc = dd.minimize(view, par->check_unmodified != 0, par->verify_mcs != 0, &o);
  if (demi_host::knob("DEMI_DDMIN_TIMING")) {
    const double t1 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[ddmin] %u consultations, %u launches, %llu replays: oracle %.3f ms, host loop %.3f ms\n", o.consultations, o.launches,
            (unsigned long long)o.replays, dd.oracle_s_ * 1e3, (t1 - t0 - dd.oracle_s_) * 1e3);
  }
  stats->consultations = o.consultations; stats->launches = o.launches; stats->replays = o.replays;
  if (rc) return rc;
  for (int k = 0; k < 4; k++) out_mcs[k] = o.mcs.w[k];
  stats->mcs_len = o.mcs.count();
  stats->verified = o.verified;
  for (uint32_t i = 0; i < o.consulted.size() && i < cap; i++) {
    if (out_consulted) memcpy(out_consulted + 4 * (size_t)i, o.consulted[i].first.w, 32);
    if (out_passed) out_passed[i] = o.consulted[i].second ? 1 : 0;
  }
  for (uint32_t i = 0; i < o.batches.size() && i < batches_cap; i++) if (out_batches) out_batches[i] = o.batches[i];
  return DEMI_OK;
}

}  // namespace demi_host
