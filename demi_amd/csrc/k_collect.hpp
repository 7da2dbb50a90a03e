// k_collect.hpp — found-violation set extraction: stream the verdict array (16 B per schedule,
// coalesced uint4 loads), ballot the violating lanes, one atomic per wave, compacted 16-byte
// entries out.  HBM-bound: n * 16 bytes read, hits * 16 bytes written.
#pragma once

#include "demi_device.hpp"

namespace demi {

// flag_mask: which verdict flags select an entry (DEMI_V_VIOLATION for the found-violation set; with the overflow flags
// also the aborted executions, so that a driver can re-run them with a larger capacity).  count[0] = number of selected
// entries, count[1] (when first_index is set) = the lowest selected index.
// One atomic per WORKGROUP and pass, not per wave: with a few per cent of violating schedules nearly every wave of 64 has a
// hit, and 16 384 atomics on one address were what the kernel's 0.12 ms consisted of (16 MB of verdicts is a few microseconds
// of HBM time).  The waves' hit counts meet in LDS, wave 0's first lane claims the block's range, every hit writes at its
// rank inside it.
__global__ __launch_bounds__(256) void k_collect_violations(const demi_verdict* __restrict__ v, uint64_t n,
                                                            uint64_t index_base, demi_violation* __restrict__ out,
                                                            uint32_t cap, unsigned long long* __restrict__ count,
                                                            uint32_t flag_mask, unsigned long long* __restrict__ first_index) {
  __shared__ uint32_t s_cnt[4];
  __shared__ unsigned long long s_base, s_first[4];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // whole workgroups iterate together so that the barriers see every wave
  const uint64_t n_round = (n + 255) & ~255ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    uint4 x = make_uint4(0, 0, 0, 0);
    if (i < n) x = *reinterpret_cast<const uint4*>(&v[i]);
    const bool hit = (x.x & flag_mask) != 0;
    const uint64_t m = __ballot(hit);
    if (lane == 0) {
      s_cnt[wave] = (uint32_t)__popcll(m);
      s_first[wave] = m ? (unsigned long long)(index_base + (i - lane) + (uint64_t)__builtin_ctzll(m)) : ~0ull;
    }
    __syncthreads();
    const uint32_t c0 = s_cnt[0], c1 = s_cnt[1], c2 = s_cnt[2], c3 = s_cnt[3];
    const uint32_t total = c0 + c1 + c2 + c3;
    if (threadIdx.x == 0 && total) {
      s_base = atomicAdd(count, (unsigned long long)total);
      if (first_index) {
        unsigned long long f = s_first[0];
        f = s_first[1] < f ? s_first[1] : f; f = s_first[2] < f ? s_first[2] : f; f = s_first[3] < f ? s_first[3] : f;
        atomicMin(first_index, f);
      }
    }
    __syncthreads();
    if (hit) {
      const uint32_t before = (wave > 0 ? c0 : 0u) + (wave > 1 ? c1 : 0u) + (wave > 2 ? c2 : 0u);
      const uint64_t pos = s_base + before + (uint64_t)__popcll(m & ((1ULL << lane) - 1));
      if (pos < cap) {
        demi_violation e;
        e.index = index_base + i; e.fingerprint = x.y; e.flags = x.x;
        out[pos] = e;
      }
    }
    __syncthreads();          // (s_cnt / s_base are rewritten by the next pass)
  }
}

}  // namespace demi
