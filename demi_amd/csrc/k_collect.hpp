// k_collect.hpp — found-violation set extraction: stream the verdict array (16 B per schedule,
// coalesced uint4 loads), ballot the violating lanes, one atomic per wave, compacted 16-byte
// entries out.  HBM-bound: n * 16 bytes read, hits * 16 bytes written.
#pragma once

#include "demi_device.hpp"

namespace demi {

// flag_mask: which verdict flags select an entry (DEMI_V_VIOLATION for the found-violation set; with the overflow flags
// also the aborted executions, so that a driver can re-run them with a larger capacity).  count[0] = number of selected
// entries, count[1] (when first_index is set) = the lowest selected index.
__global__ __launch_bounds__(256) void k_collect_violations(const demi_verdict* __restrict__ v, uint64_t n,
                                                            uint64_t index_base, demi_violation* __restrict__ out,
                                                            uint32_t cap, unsigned long long* __restrict__ count,
                                                            uint32_t flag_mask, unsigned long long* __restrict__ first_index) {
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // whole waves iterate together so the ballot sees every lane of the wave
  const uint64_t n_round = (n + 63) & ~63ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
    uint4 x = make_uint4(0, 0, 0, 0);
    if (i < n) x = *reinterpret_cast<const uint4*>(&v[i]);
    const bool hit = (x.x & flag_mask) != 0;
    const uint64_t m = __ballot(hit);
    if (m == 0) continue;
    unsigned long long base = 0;
    if (lane == 0) {
      base = atomicAdd(count, (unsigned long long)__popcll(m));
      if (first_index) atomicMin(first_index, (unsigned long long)(index_base + (i - lane) + (uint64_t)__builtin_ctzll(m)));
    }
    base = __shfl(base, 0);
    if (hit) {
      const uint64_t pos = base + __popcll(m & ((1ULL << lane) - 1));
      if (pos < cap) {
        demi_violation e;
        e.index = index_base + i; e.fingerprint = x.y; e.flags = x.x;
        out[pos] = e;
      }
    }
  }
}

}  // namespace demi
