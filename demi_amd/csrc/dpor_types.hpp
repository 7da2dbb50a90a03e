// dpor_types.hpp — plain records exchanged between the host's backtrack queue (dpor_host.hpp) and the device-resident
// bookkeeping kernels (k3_pairs.hpp).  No device code: also compiled by the host-only test harness.
#pragma once

#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

namespace demi {

struct DporItem {            // one backtrack point dequeued for this round = one interleaving to run
  uint32_t src;              // arena id of the interleaving that found it (0xFFFFFFFF: the first run, empty next trace)
  uint8_t branch, later, earlier, pad;
};

struct DporPoint {           // a backtrack point that may still be dequeued live (device -> host), 40 bytes
  unsigned long long flip_a, flip_b;   // (later key, earlier key): the pair getNext() tests with isExplored
  unsigned long long ordinal;          // creation order within the round: item * max_pairs + pair index
  uint32_t src;                        // arena id of the finished interleaving that found it
  uint8_t branch, later, earlier, pad;
  uint32_t pad2;
};

struct DporKill { unsigned long long a, b; };   // this pair is explored now: queued points flipping into it are dead

// One racing pair as it travels between GPUs (multi-GPU rounds): everything insert / decide need, so that the rank that
// owns the pair's table entries does not need the trace it came from.  32 bytes.
struct DporPairRec {
  unsigned long long ke, kl;   // node keys of the earlier / later event
  uint32_t ordinal;            // creation order within the round: (item index in the round) * max_pairs + pair index
  uint32_t src;                // arena id of the interleaving that found it
  uint8_t branch, later, earlier, pad;
  uint32_t pad2;
};

// which rank owns the table entries of the unordered pair {a, b}: (a, b) and (b, a) live together
#if defined(__HIPCC__) || defined(__HIPCC_RTC__)
__host__ __device__
#endif
inline uint32_t dpor_pair_owner(unsigned long long a, unsigned long long b, uint32_t world) {
  const unsigned long long lo = a < b ? a : b, hi = a < b ? b : a;
  unsigned long long h = (lo * 0x9E3779B97F4A7C15ULL) ^ (hi * 0xC2B2AE3D27D4EB4FULL);
  h ^= h >> 31;
  return (uint32_t)(h % world);
}

}  // namespace demi
