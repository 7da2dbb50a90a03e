// Test harness (CPU suite only): the native DPOR bookkeeping of demi_amd/csrc/dpor_host.hpp driven by the
// CPU oracle instead of the K3 kernel, so the queue / explored-set logic is covered without a GPU and its
// host time can be profiled.  Built by tests/test_dpor_cpu.py with g++, linked against oracle/_build/liboracle.so.
#include <cstring>
#include <thread>
#include <vector>

#include "../demi_amd/csrc/dpor_host.hpp"
#include "demi_oracle.h"

extern "C" int harness_dpor_explore(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                    const demi_dpor_params* par, const demi_dpor_search* srch, int n_threads,
                                    demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                    demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                    demi_dpor_stats* stats, double* seconds) {
  std::vector<demi_dpor_trace_entry> all_tr;
  std::vector<demi_dpor_pair> all_pr;
  auto run = [&](const demi_dpor_trace_entry* pf, const uint32_t* pl, const uint32_t* sh, uint32_t stride, uint64_t n,
                 demi_verdict* vd, uint32_t* tl, uint32_t* np) {
    all_tr.resize(n * DEMI_DPOR_MAX_TRACE);
    all_pr.resize(n * (size_t)par->max_pairs);
    auto work = [&](unsigned t) {
      std::vector<uint64_t> keys(DEMI_DPOR_MAX_TRACE);
      for (uint64_t i = t; i < n; i += (unsigned)n_threads) {
        for (uint32_t k = 0; k < pl[i]; k++) keys[k] = pf[i * stride + k].key;
        orc_dpor_execute(m, ext, n_ext, keys.data(), pl[i], sh ? sh[i] : 0u, par, &vd[i], &all_tr[i * DEMI_DPOR_MAX_TRACE], &tl[i],
                         &all_pr[i * (size_t)par->max_pairs], &np[i]);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, (unsigned)t);
    work(0u);
    for (auto& th : pool) th.join();
    return 0;
  };
  auto fetch = [&](size_t lo, size_t cnt, demi_dpor_trace_entry* tr, demi_dpor_pair* pr) {
    memcpy(tr, &all_tr[lo * DEMI_DPOR_MAX_TRACE], sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * cnt);
    memcpy(pr, &all_pr[lo * (size_t)par->max_pairs], sizeof(demi_dpor_pair) * (size_t)par->max_pairs * cnt);
    return 0;
  };
  return demi_host::explore_loop(run, fetch, par->max_pairs, srch, out_verdicts, out_prefix_len, out_rounds,
                                 first_violation_trace, first_violation_len, stats, seconds);
}
