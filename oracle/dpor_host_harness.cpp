// Test harness (CPU suite only): the native DPOR bookkeeping of demi_amd/csrc/dpor_host.hpp driven by the
// CPU oracle instead of the K3 kernel, so the queue / explored-set logic is covered without a GPU and its
// host time can be profiled.  Built by tests/test_dpor_cpu.py with g++, linked against oracle/_build/liboracle.so.
#include <cstring>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../demi_amd/csrc/dpor_host.hpp"
#include "../demi_amd/csrc/comm.hpp"
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

// ------------------------------------------------------------------------------------------------------------------
// The device-resident bookkeeping (demi_amd/csrc/k3_pairs.hpp: mark / insert / decide over a pair table, next traces
// read from an arena of finished traces) restated sequentially on the host, under the same host loop
// (explore_rounds_resident).  It is the CPU check of that loop and the reference the kernels are compared with.
namespace {
struct SimEntry { uint32_t state = 0; unsigned long long cand = 0; };
constexpr uint32_t SIM_EXPLORED = 0x80000000u, SIM_QMASK = 0x1FFu;
unsigned long long sim_cand(uint32_t round, uint32_t branch, unsigned long long ord) {
  return ((unsigned long long)round << 40) | ((unsigned long long)(branch + 1) << 31) | (0x7FFFFFFFull - ord);
}
struct SimDev {
  const demi_model* m; const demi_ext_event* ext; uint32_t n_ext; const demi_dpor_params* par; int n_threads = 1;
  std::vector<demi_host::Trace> arena;
  std::unordered_map<std::pair<uint64_t, uint64_t>, SimEntry, demi_host::PairKeyHash> table;
  std::vector<std::vector<demi_dpor_pair>> pairs;       // racing pairs of the last round's interleavings
  // the device's copy of the COMMIT's explored-pair table (REFERENCE order, explore_reference_resident): ordered pair -> state
  std::unordered_map<std::pair<uint64_t, uint64_t>, uint32_t, demi_host::PairKeyHash> real_tab;
  unsigned long long pairs_reported = 0, pairs_after_parent = 0, pairs_kept = 0;
  unsigned long long sum_branch = 0, sum_later = 0, n_items = 0, sum_len = 0;
  ~SimDev() { if (getenv("DEMI_HARNESS_STATS")) fprintf(stderr, "[harness] items %llu mean shared (branch + 1) %.1f mean prefix (later) %.1f\n", n_items, n_items ? (double)sum_branch / n_items : 0.0, n_items ? (double)sum_later / n_items : 0.0); }

  // ResidentDev::round_ref restated: the ROUNDS machinery for the speculation, then the commit filter per interleaving -
  // (a) ParentFilter against the parent's trace in the arena, (b) no-ops under the snapshot of the commit's table; the
  // survivors stay "with the device", per arena id
  void apply_deltas(const demi_host::RefDelta* deltas, uint32_t n_deltas) {
    for (uint32_t i = 0; i < n_deltas; i++) {
      uint32_t& f = real_tab[{deltas[i].lo, deltas[i].hi}];      // (merged with a maximum, as k3_pairs.hpp ref_state_merge does:
      uint32_t& r = real_tab[{deltas[i].hi, deltas[i].lo}];      //  a state only grows as a number)
      f = std::max(f, deltas[i].state[0]);
      r = std::max(r, deltas[i].state[1]);
    }
  }
  bool noop(uint64_t ke, uint64_t kl, uint32_t branch) const {
    auto f = real_tab.find({ke, kl});
    auto r = real_tab.find({kl, ke});
    const uint32_t sf = f == real_tab.end() ? 0u : f->second, sr = r == real_tab.end() ? 0u : r->second;
    return (sf & SIM_EXPLORED) && ((sr & SIM_EXPLORED) || (sr & SIM_QMASK) > branch);
  }
  int round_ref(const demi::DporItem* items, const uint8_t* use_parent, uint32_t n, uint32_t round_no, uint32_t base_id,
                const demi_host::RefDelta* deltas, uint32_t n_deltas, demi_verdict* vd, std::vector<demi::DporPoint>& pts,
                std::vector<demi::DporKill>& kills, uint32_t* rec_cnt) {
    apply_deltas(deltas, n_deltas);
    const int rc = round(items, n, round_no, base_id, vd, pts, kills);
    if (rc) return rc;
    if (held.size() < (size_t)base_id + n) held.resize((size_t)base_id + n);
    std::vector<demi_dpor_pair> kept;
    for (uint32_t i = 0; i < n; i++) {
      const demi_host::Trace& T = arena[(size_t)base_id + i];
      const std::vector<demi_dpor_pair>& P = pairs[i];
      pairs_reported += P.size();
      const demi_host::ParentFilter pf(use_parent[i] ? arena[items[i].src] : demi_host::Trace());
      if (pf.active()) pf.filter(T.data(), (uint32_t)T.size(), P.data(), (uint32_t)P.size(), kept);
      else kept.assign(P.begin(), P.end());
      pairs_after_parent += kept.size();
      std::vector<demi_host::RefRec>& H = held[(size_t)base_id + i];
      for (const demi_dpor_pair& p : kept) {
        const uint64_t ke = T[p.earlier].key, kl = T[p.later].key;
        if (noop(ke, kl, p.branch)) continue;                                      // a no-op for the commit
        H.push_back(demi_host::RefRec{ke, kl, p.branch, p.later, p.earlier, 0, 0});
      }
      rec_cnt[i] = (uint32_t)H.size();
      pairs_kept += rec_cnt[i];
    }
    return 0;
  }
  // the two-step form the host loop uses: the work happens at begin, the results are handed over at end
  struct PendRound { bool active = false; uint32_t n = 0; int rc = 0; std::vector<demi_verdict> vd; std::vector<demi::DporPoint> pts; std::vector<demi::DporKill> kills; std::vector<uint32_t> rec_cnt; } pend;
  int round_ref_begin(const demi::DporItem* items, const uint8_t* use_parent, uint32_t n, uint32_t round_no, uint32_t base_id,
                      const demi_host::RefDelta* deltas, uint32_t n_deltas, bool speculate) {
    if (pend.active) return DEMI_ERR_INVALID_ARG;
    (void)speculate;                                    // (always: this table never fills)
    pend.vd.assign(n, demi_verdict{}); pend.rec_cnt.assign(n, 0u); pend.pts.clear(); pend.kills.clear();
    pend.rc = round_ref(items, use_parent, n, round_no, base_id, deltas, n_deltas, pend.vd.data(), pend.pts, pend.kills, pend.rec_cnt.data());
    pend.active = true; pend.n = n;
    return 0;
  }
  int round_ref_end(demi_verdict* vd, std::vector<demi::DporPoint>& pts, std::vector<demi::DporKill>& kills, uint32_t* rec_cnt, bool* spec_overflow) {
    if (spec_overflow) *spec_overflow = false;          // (this table is a map: it never fills)
    if (!pend.active) return DEMI_ERR_INVALID_ARG;
    pend.active = false;
    if (pend.rc) return pend.rc;
    memcpy(vd, pend.vd.data(), sizeof(demi_verdict) * pend.n);
    memcpy(rec_cnt, pend.rec_cnt.data(), 4 * (size_t)pend.n);
    pts = pend.pts; kills = pend.kills;
    return 0;
  }
  void round_ref_abort() { pend.active = false; }
  // ResidentDev::ref_fetch restated: the held records of `ids`, filtered again under the table as of now
  int ref_fetch(const uint32_t* ids, uint32_t m, const demi_host::RefDelta* deltas, uint32_t n_deltas, uint64_t* rec_off,
                uint32_t* rec_cnt, const demi_host::RefRec** recs_out) {
    apply_deltas(deltas, n_deltas);
    rec_chunks.emplace_back();
    std::vector<demi_host::RefRec>& recs = rec_chunks.back();
    for (uint32_t j = 0; j < m; j++) {
      rec_off[j] = recs.size();
      for (const demi_host::RefRec& r : held[ids[j]]) if (!noop(r.ke, r.kl, r.branch)) recs.push_back(r);
      rec_cnt[j] = (uint32_t)(recs.size() - rec_off[j]);
      pairs_fetched += rec_cnt[j];
      std::vector<demi_host::RefRec>().swap(held[ids[j]]);
    }
    fetches++; fetched_ids += m;
    *recs_out = recs.data();
    return 0;
  }
  // the asynchronous form the host loop uses: the work happens at begin (the table as of THEN), the answer is handed over at end
  std::vector<uint64_t> pend_off;
  std::vector<uint32_t> pend_cnt;
  const demi_host::RefRec* pend_recs = nullptr;
  int ref_fetch_begin(const uint32_t* ids, uint32_t m, const demi_host::RefDelta* deltas, uint32_t n_deltas) {
    pend_off.assign(m, 0); pend_cnt.assign(m, 0);
    return ref_fetch(ids, m, deltas, n_deltas, pend_off.data(), pend_cnt.data(), &pend_recs);
  }
  int ref_fetch_end(uint64_t* rec_off, uint32_t* rec_cnt, const demi_host::RefRec** recs_out) {
    for (size_t j = 0; j < pend_off.size(); j++) { rec_off[j] = pend_off[j]; rec_cnt[j] = pend_cnt[j]; }
    *recs_out = pend_recs;
    return 0;
  }
  std::vector<std::vector<demi_host::RefRec>> held;           // per arena id: the first filter's survivors
  std::deque<std::vector<demi_host::RefRec>> rec_chunks;      // one per fetch, alive until the exploration ends
  unsigned long long pairs_fetched = 0, fetches = 0, fetched_ids = 0;

  int round(const demi::DporItem* items, uint32_t n, uint32_t round_no, uint32_t base_id, demi_verdict* vd,
            std::vector<demi::DporPoint>& pts, std::vector<demi::DporKill>& kills) {
    if (arena.size() < (size_t)base_id + n) arena.resize((size_t)base_id + n);
    const uint32_t mp = par->max_pairs;
    pairs.assign(n, std::vector<demi_dpor_pair>());
    // mark
    for (uint32_t i = 0; i < n; i++) {
      if (items[i].src == 0xFFFFFFFFu) continue;
      sum_branch += items[i].branch + 1u; sum_later += items[i].later; n_items++;
      const demi_host::Trace& T = arena[items[i].src];
      table[{T[items[i].later].key, T[items[i].earlier].key}].state |= SIM_EXPLORED;
    }
    // the interleavings (next trace = take(branch + 1) ++ replay, from the arena)
    auto work = [&](unsigned t) {
      std::vector<uint64_t> keys(DEMI_DPOR_MAX_TRACE);
      std::vector<demi_dpor_trace_entry> tr(DEMI_DPOR_MAX_TRACE);
      std::vector<demi_dpor_pair> pr(mp ? mp : 1);
      for (uint32_t i = t; i < n; i += (unsigned)n_threads) {
        uint32_t pl = 0, shared = 0;
        if (items[i].src != 0xFFFFFFFFu) {
          const demi_host::Trace& T = arena[items[i].src];
          for (uint32_t k = 0; k <= items[i].branch; k++) keys[pl++] = T[k].key;
          for (uint32_t k = items[i].branch + 1u; k <= items[i].later; k++) if (k != items[i].earlier) keys[pl++] = T[k].key;
          shared = items[i].branch + 1u;
        }
        uint32_t tl = 0, np = 0;
        orc_dpor_execute(m, ext, n_ext, keys.data(), pl, shared, par, &vd[i], tr.data(), &tl, pr.data(), &np);
        arena[(size_t)base_id + i].assign(tr.begin(), tr.begin() + tl);
        pairs[i].assign(pr.begin(), pr.begin() + np);
      }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; t++) pool.emplace_back(work, (unsigned)t);
    work(0u);
    for (auto& th : pool) th.join();
    // insert
    for (uint32_t i = 0; i < n; i++) {
      const demi_host::Trace& T = arena[(size_t)base_id + i];
      for (uint32_t k = 0; k < pairs[i].size(); k++) {
        const demi_dpor_pair p = pairs[i][k];
        const uint64_t ke = T[p.earlier].key, kl = T[p.later].key;
        SimEntry& e1 = table[{ke, kl}];
        if (!(e1.state & SIM_EXPLORED) && (e1.state & SIM_QMASK)) kills.push_back(demi::DporKill{ke, kl});
        e1.state |= SIM_EXPLORED;
        SimEntry& e2 = table[{kl, ke}];
        const unsigned long long c = sim_cand(round_no, p.branch, (unsigned long long)i * mp + k);
        if (c > e2.cand) e2.cand = c;
      }
    }
    // decide
    for (uint32_t i = 0; i < n; i++) {
      const demi_host::Trace& T = arena[(size_t)base_id + i];
      for (uint32_t k = 0; k < pairs[i].size(); k++) {
        const demi_dpor_pair p = pairs[i][k];
        SimEntry& e = table[{T[p.later].key, T[p.earlier].key}];
        const unsigned long long ord = (unsigned long long)i * mp + k;
        if (e.state & SIM_EXPLORED) continue;
        if (e.cand != sim_cand(round_no, p.branch, ord)) continue;
        if ((e.state & SIM_QMASK) > p.branch) continue;
        e.state = (e.state & ~SIM_QMASK) | ((uint32_t)p.branch + 1);
        pts.push_back(demi::DporPoint{T[p.later].key, T[p.earlier].key, ord, base_id + i, p.branch, p.later, p.earlier, 0, 0});
      }
    }
    return 0;
  }
  uint32_t ids_used(uint32_t n) const { return n; }
  int fetch_trace(uint32_t id, demi_dpor_trace_entry* out, uint32_t* len) {
    memcpy(out, arena[id].data(), sizeof(demi_dpor_trace_entry) * arena[id].size());
    *len = (uint32_t)arena[id].size();
    return 0;
  }
};

// The multi-GPU round (ResidentDev::round_sharded in demi_gpu.hip) restated for host "ranks": this rank runs its block of
// the round's items, all-gathers traces / verdicts / racing-pair records, owns the table entries with
// dpor_pair_owner == rank, and all-gathers what it decided.  Every rank must return the same verdicts, points and kills.
struct SimShardDev {
  const demi_model* m; const demi_ext_event* ext; uint32_t n_ext; const demi_dpor_params* par;
  demi_comm::Comm* comm;
  std::vector<demi_host::Trace> arena;          // replicated
  std::unordered_map<std::pair<uint64_t, uint64_t>, SimEntry, demi_host::PairKeyHash> table;     // this rank's shard

  uint32_t ids_used(uint32_t n) const { const uint32_t W = (uint32_t)comm->world; return W * ((n + W - 1) / W); }

  template <class T>
  std::vector<T> gather_var(const std::vector<T>& mine) {       // counts, then blocks padded to the largest
    const uint32_t W = (uint32_t)comm->world;
    unsigned long long c = mine.size();
    std::vector<unsigned long long> counts(W);
    comm->allgather(&c, counts.data(), sizeof c, nullptr);
    unsigned long long mx = 0;
    for (auto x : counts) mx = x > mx ? x : mx;
    std::vector<T> out;
    if (!mx) return out;
    std::vector<T> send(mx), recv((size_t)mx * W);
    std::copy(mine.begin(), mine.end(), send.begin());
    comm->allgather(send.data(), recv.data(), sizeof(T) * (size_t)mx, nullptr);
    for (uint32_t k = 0; k < W; k++) out.insert(out.end(), recv.begin() + (size_t)k * mx, recv.begin() + (size_t)k * mx + counts[k]);
    return out;
  }

  int round(const demi::DporItem* items, uint32_t n, uint32_t round_no, uint32_t base_id, demi_verdict* vd,
            std::vector<demi::DporPoint>& pts, std::vector<demi::DporKill>& kills) {
    const uint32_t W = (uint32_t)comm->world, r = (uint32_t)comm->rank, mp = par->max_pairs;
    const uint32_t blk = (n + W - 1) / W;
    const uint32_t lo = r * blk < n ? r * blk : n, hi = lo + blk < n ? lo + blk : n;
    if (arena.size() < (size_t)base_id + (size_t)W * blk) arena.resize((size_t)base_id + (size_t)W * blk);
    for (uint32_t i = 0; i < n; i++) {        // mark (owner only)
      if (items[i].src == 0xFFFFFFFFu) continue;
      const demi_host::Trace& T = arena[items[i].src];
      const uint64_t a = T[items[i].later].key, b = T[items[i].earlier].key;
      if (demi::dpor_pair_owner(a, b, W) == r) table[{a, b}].state |= SIM_EXPLORED;
    }
    // my block: interleavings, records
    std::vector<demi_dpor_trace_entry> my_tr((size_t)blk * DEMI_DPOR_MAX_TRACE);
    std::vector<uint32_t> my_tl(blk, 0);
    std::vector<demi_verdict> my_vd(blk);
    memset(my_vd.data(), 0, sizeof(demi_verdict) * blk);
    std::vector<demi::DporPairRec> my_recs;
    std::vector<uint64_t> keys(DEMI_DPOR_MAX_TRACE);
    std::vector<demi_dpor_pair> pr(mp ? mp : 1);
    for (uint32_t i = lo; i < hi; i++) {
      uint32_t pl = 0, shared = 0;
      if (items[i].src != 0xFFFFFFFFu) {
        const demi_host::Trace& T = arena[items[i].src];
        for (uint32_t k = 0; k <= items[i].branch; k++) keys[pl++] = T[k].key;
        for (uint32_t k = items[i].branch + 1u; k <= items[i].later; k++) if (k != items[i].earlier) keys[pl++] = T[k].key;
        shared = items[i].branch + 1u;
      }
      uint32_t np = 0;
      demi_dpor_trace_entry* tr = &my_tr[(size_t)(i - lo) * DEMI_DPOR_MAX_TRACE];
      orc_dpor_execute(m, ext, n_ext, keys.data(), pl, shared, par, &my_vd[i - lo], tr, &my_tl[i - lo], pr.data(), &np);
      for (uint32_t k = 0; k < np; k++)
        my_recs.push_back(demi::DporPairRec{tr[pr[k].earlier].key, tr[pr[k].later].key, i * mp + k, base_id + i, pr[k].branch, pr[k].later,
                                            pr[k].earlier, 0, 0});
    }
    // exchange: traces, lengths, verdicts (equal blocks), records (variable)
    std::vector<demi_dpor_trace_entry> all_tr((size_t)blk * W * DEMI_DPOR_MAX_TRACE);
    std::vector<uint32_t> all_tl((size_t)blk * W);
    std::vector<demi_verdict> all_vd((size_t)blk * W);
    comm->allgather(my_tr.data(), all_tr.data(), sizeof(demi_dpor_trace_entry) * DEMI_DPOR_MAX_TRACE * (size_t)blk, nullptr);
    comm->allgather(my_tl.data(), all_tl.data(), 4 * (size_t)blk, nullptr);
    comm->allgather(my_vd.data(), all_vd.data(), sizeof(demi_verdict) * (size_t)blk, nullptr);
    for (uint32_t i = 0; i < W * blk; i++)
      arena[(size_t)base_id + i].assign(&all_tr[(size_t)i * DEMI_DPOR_MAX_TRACE], &all_tr[(size_t)i * DEMI_DPOR_MAX_TRACE] + all_tl[i]);
    memcpy(vd, all_vd.data(), sizeof(demi_verdict) * n);
    std::vector<demi::DporPairRec> recs = gather_var(my_recs);
    // insert + decide for the pairs this rank owns
    std::vector<demi::DporPoint> my_pts;
    std::vector<demi::DporKill> my_kills;
    for (const demi::DporPairRec& x : recs) {
      if (demi::dpor_pair_owner(x.ke, x.kl, W) != r) continue;
      SimEntry& e1 = table[{x.ke, x.kl}];
      if (!(e1.state & SIM_EXPLORED) && (e1.state & SIM_QMASK)) my_kills.push_back(demi::DporKill{x.ke, x.kl});
      e1.state |= SIM_EXPLORED;
      SimEntry& e2 = table[{x.kl, x.ke}];
      const unsigned long long c = sim_cand(round_no, x.branch, x.ordinal);
      if (c > e2.cand) e2.cand = c;
    }
    for (const demi::DporPairRec& x : recs) {
      if (demi::dpor_pair_owner(x.ke, x.kl, W) != r) continue;
      SimEntry& e = table[{x.kl, x.ke}];
      if (e.state & SIM_EXPLORED) continue;
      if (e.cand != sim_cand(round_no, x.branch, x.ordinal)) continue;
      if ((e.state & SIM_QMASK) > x.branch) continue;
      e.state = (e.state & ~SIM_QMASK) | ((uint32_t)x.branch + 1);
      my_pts.push_back(demi::DporPoint{x.kl, x.ke, x.ordinal, x.src, x.branch, x.later, x.earlier, 0, 0});
    }
    pts = gather_var(my_pts);
    kills = gather_var(my_kills);
    return 0;
  }
  int fetch_trace(uint32_t id, demi_dpor_trace_entry* out, uint32_t* len) {
    memcpy(out, arena[id].data(), sizeof(demi_dpor_trace_entry) * arena[id].size());
    *len = (uint32_t)arena[id].size();
    return 0;
  }
};
}  // namespace

// `world` ranks as threads of this process (LocalComm): every rank runs the whole host loop; outputs are [world][...]
extern "C" int harness_dpor_explore_sharded(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                            const demi_dpor_params* par, const demi_dpor_search* srch, int world,
                                            demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                            demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                            demi_dpor_stats* stats) {
  demi_comm::LocalGroup group(world);
  std::vector<int> rcs((size_t)world, 0);
  const size_t cap = srch->max_interleavings;
  auto run = [&](int r) {
    demi_comm::LocalComm comm(&group, r);
    SimShardDev dev{m, ext, n_ext, par, &comm, {}, {}};
    rcs[(size_t)r] = demi_host::explore_rounds_resident(dev, srch, out_verdicts + (size_t)r * cap, out_prefix_len + (size_t)r * cap,
                                                        out_rounds + (size_t)r * cap, first_violation_trace + (size_t)r * DEMI_DPOR_MAX_TRACE,
                                                        first_violation_len + r, stats + r, nullptr);
  };
  std::vector<std::thread> pool;
  for (int r = 1; r < world; r++) pool.emplace_back(run, r);
  run(0);
  for (auto& th : pool) th.join();
  for (int rc : rcs) if (rc) return rc;
  return 0;
}

// REFERENCE order with the results resident (explore_reference_resident) over the restated device: the committed sequence must
// be the one of harness_dpor_explore with order = REFERENCE (and of batch = 1).  pair_counts (optional): racing pairs
// reported / left by the parent filter / left by the snapshot filter as well (= what would cross PCIe).
extern "C" int harness_dpor_explore_reference_resident(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                                       const demi_dpor_params* par, const demi_dpor_search* srch, int n_threads,
                                                       demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                                       demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                                       demi_dpor_stats* stats, double* seconds, uint64_t* pair_counts) {
  SimDev dev{m, ext, n_ext, par, n_threads};
  const int rc = demi_host::explore_reference_resident(dev, srch, out_verdicts, out_prefix_len, out_rounds, first_violation_trace,
                                                       first_violation_len, stats, seconds);
  if (pair_counts) { pair_counts[0] = dev.pairs_reported; pair_counts[1] = dev.pairs_after_parent; pair_counts[2] = dev.pairs_kept; pair_counts[3] = dev.pairs_fetched; pair_counts[4] = dev.fetches; pair_counts[5] = dev.fetched_ids; }
  return rc;
}

extern "C" int harness_dpor_explore_resident(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                             const demi_dpor_params* par, const demi_dpor_search* srch, int n_threads,
                                             demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                             demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                             demi_dpor_stats* stats, double* seconds, uint64_t* table_entries) {
  SimDev dev{m, ext, n_ext, par, n_threads};
  const int rc = demi_host::explore_rounds_resident(dev, srch, out_verdicts, out_prefix_len, out_rounds, first_violation_trace,
                                                    first_violation_len, stats, seconds);
  if (table_entries) *table_entries = dev.table.size();
  return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// DDMin in one call (demi_amd/csrc/ddmin_host.hpp, what demi_ddmin runs around K2 launches) over the CPU oracle's
// STSScheduler replays: the CPU check of that loop against the Python mirror (demi_amd/minification.py).
#include "../demi_amd/csrc/ddmin_host.hpp"
#include "../demi_amd/csrc/incddmin_host.hpp"

extern "C" int harness_ddmin(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_rec_event* rec, uint32_t n_rec,
                             const demi_limits* lim, const demi_ddmin_params* par, const uint8_t* conjoined, int n_threads,
                             uint64_t* out_mcs, uint64_t* out_consulted, uint8_t* out_passed, uint32_t cap, uint32_t* out_batches,
                             uint32_t batches_cap, demi_ddmin_stats* stats) {
  std::vector<demi_verdict> vd;
  auto test = [&](const uint64_t* masks, uint32_t n, uint8_t* reproduced) -> int {
    vd.resize(n);
    int rc = orc_sts_replay_batch(m, ext, n_ext, rec, n_rec, masks, n, lim, vd.data(), n_threads);
    if (rc) return rc;
    for (uint32_t i = 0; i < n; i++) {
      if (vd[i].flags & (DEMI_V_PENDING_OVF | DEMI_V_QUEUE_OVF)) return DEMI_ERR_CAPACITY;
      reproduced[i] = (vd[i].flags & DEMI_V_VIOLATION) ? 1 : 0;
    }
    return DEMI_OK;
  };
  return demi_host::sts_sched_ddmin(ext, n_ext, conjoined, par, test, out_mcs, out_consulted, out_passed, cap, out_batches, batches_cap, stats);
}


// DPORwHeuristics with ArvindDistanceOrdering / setMaxDistance / setInitialTrace (dpor_host.hpp explore_rounds_ordered, what
// demi_dpor_explore runs for them) over the CPU oracle's interleavings: the CPU check against the Python mirror.
extern "C" int harness_dpor_explore_ordered(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext, const demi_dpor_params* par,
                                            const demi_dpor_search* srch, const uint64_t* original_keys, uint32_t n_original,
                                            const demi_dpor_trace_entry* initial, uint32_t n_initial, int n_threads,
                                            demi_verdict* out_verdicts, uint32_t* out_prefix_len, uint32_t* out_rounds,
                                            demi_dpor_trace_entry* first_violation_trace, uint32_t* first_violation_len,
                                            demi_dpor_stats* stats, void* state) {
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
  demi_host::OrderedSearch ord;
  ord.ordering = srch->ordering;
  ord.capped = srch->max_distance_plus1 != 0;
  ord.max_distance = srch->max_distance_plus1 ? srch->max_distance_plus1 - 1u : 0u;
  for (uint32_t i = 0; i < n_original; i++) ord.original_index[original_keys[i]] = i;
  ord.initial.assign(initial, initial + n_initial);
  return demi_host::explore_rounds_ordered(run, fetch, par->max_pairs, srch, ord, out_verdicts, out_prefix_len, out_rounds,
                                           first_violation_trace, first_violation_len, stats, nullptr, nullptr,
                                           static_cast<demi_host::OrderedState*>(state));
}
// the state one DPORwHeuristics instance keeps between its test() calls (demi_ctx holds one per loaded trace)
extern "C" void* harness_ordered_state_new() { return new demi_host::OrderedState(); }
extern "C" void harness_ordered_state_free(void* p) { delete static_cast<demi_host::OrderedState*>(p); }

// RunnerUtils.editDistanceDporDDMin's native loop (incddmin_host.hpp: IncrementalDDMin over ResumableDPOR) with the CPU oracle's
// interleavings under every DPOR consultation: the CPU check of demi_edit_distance_dpor_ddmin against the Python mirror.
extern "C" int harness_edit_distance_dpor_ddmin(const demi_model* m, const demi_ext_event* ext, uint32_t n_ext,
                                                const demi_dpor_trace_entry* initial, uint32_t n_initial, const demi_dpor_params* par,
                                                const demi_incddmin_params* ip, int n_threads, uint64_t* out_mcs, uint64_t* out_consulted,
                                                uint8_t* out_passed, uint32_t* out_distance, uint32_t cap,
                                                demi_dpor_trace_entry* out_violation_trace, demi_incddmin_stats* stats) {
  std::vector<uint64_t> keys(n_initial);
  for (uint32_t i = 0; i < n_initial; i++) keys[i] = initial[i].key;
  const uint32_t budget = ip->budget ? ip->budget : (1u << 16);
  std::vector<demi_verdict> verdicts(budget);
  std::vector<uint32_t> plen(budget), rounds(budget);
  auto explore = [&](const std::vector<demi_ext_event>& sub, const demi_dpor_search& srch, std::unique_ptr<demi_host::OrderedState>& state,
                     demi_dpor_stats* st, std::vector<demi_dpor_trace_entry>* vt) -> int {
    if (!state) state.reset(new demi_host::OrderedState());
    vt->assign(DEMI_DPOR_MAX_TRACE, demi_dpor_trace_entry{});
    uint32_t vlen = 0;
    const int rc = harness_dpor_explore_ordered(m, sub.data(), (uint32_t)sub.size(), par, &srch, keys.data(), n_initial, initial, n_initial, n_threads,
                                                verdicts.data(), plen.data(), rounds.data(), vt->data(), &vlen, st, state.get());
    vt->resize(vlen);
    return rc;
  };
  demi_host::IncDdminResult r;
  const int rc = demi_host::edit_distance_dpor_ddmin(ext, n_ext, ip, explore, &r);
  if (rc) return rc;
  memset(stats, 0, sizeof *stats);
  for (int k = 0; k < 4; k++) out_mcs[k] = r.mcs.w[k];
  stats->replays = r.replays; stats->interleavings = r.interleavings; stats->consultations = r.consultations;
  stats->instances = r.instances; stats->passes = (uint32_t)r.distances.size(); stats->mcs_len = r.mcs.count();
  stats->verified = r.verified;
  for (size_t i = 0; i < r.distances.size() && i < 16; i++) { stats->pass_distance[i] = r.distances[i].first; stats->pass_mcs_len[i] = r.distances[i].second; }
  for (size_t i = 0; i < r.consulted.size() && i < cap; i++) {
    if (out_consulted) memcpy(out_consulted + 4 * i, r.consulted[i].first.w, 32);
    if (out_passed) out_passed[i] = r.consulted[i].second ? 1 : 0;
    if (out_distance) out_distance[i] = r.consulted_distance[i];
  }
  if (out_violation_trace && !r.violation_trace.empty()) {
    memcpy(out_violation_trace, r.violation_trace.data(), sizeof(demi_dpor_trace_entry) * r.violation_trace.size());
    stats->violation_len = (uint32_t)r.violation_trace.size();
  }
  return 0;
}

// the same loop around an arbitrary oracle (a callback): property tests of the DDMin host logic itself - atoms, splits, the
// frontier - against the Python mirror, independent of what a replay would say
extern "C" int harness_ddmin_callback(const demi_ext_event* ext, uint32_t n_ext, const demi_ddmin_params* par, const uint8_t* conjoined,
                                      int (*cb)(const uint64_t* masks, uint32_t n, uint8_t* reproduced),
                                      uint64_t* out_mcs, uint64_t* out_consulted, uint8_t* out_passed, uint32_t cap, uint32_t* out_batches,
                                      uint32_t batches_cap, demi_ddmin_stats* stats) {
  auto test = [&](const uint64_t* masks, uint32_t n, uint8_t* reproduced) -> int { return cb(masks, n, reproduced); };
  return demi_host::sts_sched_ddmin(ext, n_ext, conjoined, par, test, out_mcs, out_consulted, out_passed, cap, out_batches, batches_cap, stats);
}
