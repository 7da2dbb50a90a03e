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

}  // namespace demi
