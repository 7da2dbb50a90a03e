// incddmin_host.hpp — IncrementalDDMin + ResumableDPOR (minification/IncrementalDeltaDebugging.scala:20-92, 94-122) and the
// driver around them, RunnerUtils.editDistanceDporDDMin (RunnerUtils.scala:810-879), natively on the host.
//
// Host control only: DDMin (ddmin_host.hpp, consulted sequentially - one oracle call per node of ddmin2, no verdict cache, as
// the reference) over an oracle whose every consultation is a bounded DPORwHeuristics exploration with ArvindDistanceOrdering,
// the original execution as the initial trace and a distance cap that doubles from pass to pass (0, 2, 4, ...).  ResumableDPOR
// keeps one DPORwHeuristics per subsequence of the external events: its backtrack queue and explored pairs (OrderedState,
// dpor_host.hpp) survive between consultations, so a subsequence asked again under a larger cap continues where it stopped
// (DPORwHeuristics.scala:1219-1220) instead of starting over.  The explorations themselves are `explore`: K3 launches in the
// library (demi_edit_distance_dpor_ddmin), the CPU oracle in the test harness (oracle/dpor_host_harness.cpp) - the same loop
// either way, held against the Python mirror demi_amd/incremental_ddmin.py consultation by consultation.
#pragma once

#include <memory>
#include <unordered_map>
#include <utility>
#include <vector>

#include "ddmin_host.hpp"
#include "dpor_host.hpp"

namespace demi_host {

struct IncDdminResult {
  Mask256 mcs;                                              // over the external events
  std::vector<std::pair<uint32_t, uint32_t>> distances;     // (distance cap of the pass, MCS size after it)
  uint64_t replays = 0;                                     // MinimizationStats.total_replays as IncrementalDDMin merges it (:33-41)
  uint64_t interleavings = 0;                               // every interleaving any consultation explored (incl. the checks)
  uint32_t consultations = 0;
  uint32_t instances = 0;                                   // DPORwHeuristics instances ResumableDPOR created
  int verified = -1;                                        // -1: the MCS removed nothing, not verified (:868-873); else 0 / 1
  std::vector<demi_dpor_trace_entry> violation_trace;       // the verifying exploration's violating interleaving
  std::vector<std::pair<Mask256, bool>> consulted;          // (subsequence, passes) of every DDMin consultation, pass after pass
  std::vector<uint32_t> consulted_distance;                 // the cap it was consulted under
};

// Explore: int(const std::vector<demi_ext_event>& externals, const demi_dpor_search& search, std::unique_ptr<OrderedState>& state,
//              demi_dpor_stats* stats, std::vector<demi_dpor_trace_entry>* first_violation_trace)
//   one demi_dpor_explore over `externals` with ArvindDistanceOrdering and the caller's initial trace, on / leaving `state`.
template <class Explore>
class ResumableDpor {
 public:
  ResumableDpor(const demi_ext_event* ext, uint32_t n_ext, uint32_t batch, uint32_t budget, Explore& explore)
      : ext_(ext, ext + n_ext), batch_(batch ? batch : 256u), budget_(budget ? budget : (1u << 16)), explore_(explore) {}
  void set_max_distance(uint32_t d) { distance_ = d; }
  uint32_t instances() const { return (uint32_t)inst_.size(); }
  uint64_t interleavings() const { return interleavings_; }

  // ResumableDPOR.test (:106-117) -> DPORwHeuristics.test (DPORwHeuristics.scala:1193-1242): *reproduced = Some(trace);
  // *replays = what the consultation adds to the caller's MinimizationStats
  int test(const Mask256& events, bool* reproduced, uint64_t* replays, const std::vector<demi_dpor_trace_entry>** trace) {
    Inst& in = inst_[events];
    *replays = 0;
    if (trace) *trace = &in.shortest;
    if (in.have_shortest) { *reproduced = true; return DEMI_OK; }             // stopIfViolationFound && shortestTraceSoFar != null
    std::vector<demi_ext_event> sub;
    for (uint32_t i = 0; i < ext_.size(); i++) if (events.get(i)) sub.push_back(ext_[i]);
    demi_dpor_search srch;
    memset(&srch, 0, sizeof srch);
    srch.batch = batch_; srch.max_interleavings = budget_; srch.stop_if_violation = 1; srch.track_history = 1;
    srch.order = DEMI_DPOR_ORDER_ROUNDS; srch.ordering = DEMI_DPOR_ORDERING_ARVIND; srch.max_distance_plus1 = distance_ + 1u;
    uint64_t total = 0;
    for (;;) {
      srch.resume = in.started ? 1u : 0u;
      demi_dpor_stats st;
      memset(&st, 0, sizeof st);
      std::vector<demi_dpor_trace_entry> vt;
      const int rc = explore_(sub, srch, in.state, &st, &vt);
      if (rc) return rc;
      in.started = true;
      total += st.interleavings;
      interleavings_ += st.interleavings;
      if (st.violations) { in.shortest.swap(vt); in.have_shortest = true; break; }
      // one call explores at most `budget` interleavings; the reference's test() ends when the queue is empty, its head is at
      // the cap, or a violation was found - a call that stopped BECAUSE of the budget is continued
      if (!(total && st.interleavings >= budget_ && st.queue_len > 0 && !st.exhausted)) break;
    }
    *replays = total;
    *reproduced = in.have_shortest;
    return DEMI_OK;
  }

 private:
  struct Inst { std::unique_ptr<OrderedState> state; bool started = false, have_shortest = false; std::vector<demi_dpor_trace_entry> shortest; };
  std::vector<demi_ext_event> ext_;
  uint32_t batch_, budget_, distance_ = 0;
  Explore& explore_;
  std::unordered_map<Mask256, Inst, Mask256Hash> inst_;
  uint64_t interleavings_ = 0;
};

// RunnerUtils.editDistanceDporDDMin (:810-879): IncrementalDDMin(ResumableDPOR(...), stopAtSize, maxMaxDistance).minimize over
// the Start / Send externals (convertToDPORTrace, DPORwHeuristics.scala:1279-1303: Kill / Partition / UnPartition are not part of
// the minimization; WaitQuiescence only with ignore_quiescence = 0), then verify_mcs when the MCS is smaller than the view.
template <class Explore>
int edit_distance_dpor_ddmin(const demi_ext_event* ext, uint32_t n_ext, const demi_incddmin_params* ip, Explore&& explore, IncDdminResult* out) {
  DdminDag dag(ext, n_ext, nullptr);
  Mask256 view;
  for (uint32_t i = 0; i < n_ext; i++)
    if (ext[i].kind == DEMI_EV_START || ext[i].kind == DEMI_EV_SEND || (ext[i].kind == DEMI_EV_WAIT_QUIESCENCE && !ip->ignore_quiescence)) view.set(i);
  ResumableDpor<typename std::remove_reference<Explore>::type> oracle(ext, n_ext, ip->batch, ip->budget, explore);
  uint32_t distance = 0;
  oracle.set_max_distance(distance);
  if (ip->check_unmodified) {                                                  // IncrementalDDMin.minimize (:48-54)
    bool rep = false; uint64_t r = 0;
    const int rc = oracle.test(view, &rep, &r, nullptr);
    if (rc) return rc;
    if (!rep) return DEMI_ERR_INVALID_ARG;                                     // "Unmodified trace does not trigger violation"
  }
  Mask256 cur = view;
  const uint32_t max_max = ip->max_max_distance ? ip->max_max_distance : 256u;
  while (distance < max_max && cur.count() > ip->stop_at_size) {              // (:58-74)
    uint64_t pass_replays = 0;
    auto test = [&](const uint64_t* masks, uint32_t n, uint8_t* reproduced) -> int {
      for (uint32_t i = 0; i < n; i++) {
        Mask256 m; for (int k = 0; k < 4; k++) m.w[k] = masks[4 * (size_t)i + k];
        bool rep = false; uint64_t r = 0;
        const int rc = oracle.test(m, &rep, &r, nullptr);
        if (rc) return rc;
        pass_replays += r;
        reproduced[i] = rep ? 1 : 0;
      }
      return DEMI_OK;
    };
    SpeculativeDdmin<decltype(test)> dd(dag, test, 0, 1, /*sequential=*/true);
    DdminOutcome o;
    const int rc = dd.minimize(cur, false, false, &o);
    if (rc) return rc;
    cur = o.mcs;
    out->replays += pass_replays;
    out->consultations += o.consultations;
    for (const auto& c : o.consulted) { out->consulted.push_back(c); out->consulted_distance.push_back(distance); }
    out->distances.push_back({distance, cur.count()});
    distance = distance == 0 ? 2u : distance << 1;
    oracle.set_max_distance(distance);
  }
  out->mcs = cur;
  out->verified = -1;
  if (ip->verify_mcs && cur.count() < view.count()) {                          // (:868-873) verify_mcs under the last cap set
    bool rep = false; uint64_t r = 0;
    const std::vector<demi_dpor_trace_entry>* tr = nullptr;
    const int rc = oracle.test(cur, &rep, &r, &tr);
    if (rc) return rc;
    out->verified = rep ? 1 : 0;
    if (rep && tr) out->violation_trace = *tr;
  }
  out->instances = oracle.instances();
  out->interleavings = oracle.interleavings();
  return DEMI_OK;
}

}  // namespace demi_host
