// Test harness (CPU suite only): the native DPOR bookkeeping of demi_amd/csrc/dpor_host.hpp driven by the
// CPU oracle instead of the K3 kernel, so the queue / explored-set logic is covered without a GPU and its
// host time can be profiled.  Built by tests/test_dpor_cpu.py with g++, linked against oracle/_build/liboracle.so.
#include <thread>
#include <vector>

#include "../demi_amd/csrc/dpor_host.hpp"
#include "../oracle/demi_oracle.h"

extern "C" int harness_dpor_explore(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                    const demi_dpor_params* par, const demi_dpor_search* srch, int n_threads,
                                    demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                    demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                    demi_dpor_stats* stats, double* seconds) {
  auto run = [&](const demi_dpor_trace_entry* pf, const uint32_t* pl, uint32_t stride, uint64_t n, demi_verdict* vd,
                 demi_dpor_trace_entry* tr, uint32_t* tl, demi_dpor_pair* pr, uint32_t* np) {
    auto work = [&](unsigned t) {
      std::vector<uint64_t> keys(DEMI_DPOR_MAX_TRACE);
      for (uint64_t i = t; i < n; i += (unsigned)n_threads) {
        for (uint32_t k = 0; k < pl[i]; k++) keys[k] = pf[i * stride + k].key;
        orc_dpor_execute(m, ext, n_ext, keys.data(), pl[i], par, &vd[i], &tr[i * DEMI_DPOR_MAX_TRACE], &tl[i],
                         &pr[i * (size_t)par->max_pairs], &np[i]);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, (unsigned)t);
    work(0u);
    for (auto& th : pool) th.join();
    return 0;
  };
  return demi_host::explore_loop(run, par->max_pairs, srch, out_verdicts, out_prefix_len, out_rounds,
                                 first_violation_trace, first_violation_len, stats, seconds);
}
