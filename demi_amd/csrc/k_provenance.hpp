// k_provenance.hpp — ProvenanceTracker.pruneConcurrentEvents (schedulers/Util.scala:267-376, RunnerUtils.scala:149-163)
// for a batch of delivery traces: one wavefront per trace, the happens-before relation as a bit matrix in LDS.
//
// trace: demi_dpor_trace_entry[n] as dpor_initial_trace / K3 produce it (index 0 = root, kind 1 = delivery, kind 2 =
// WaitQuiescence marker; `parent` = trace index of the delivery during which the message was sent; n <= 256).
// happens-before, first order (:283-299): every earlier receive on the same machine precedes a receive - the pair (u, u)
// included, as in the reference - and a receive precedes the messages sent while it was handled.  Then the transitive
// closure (:316-349).  An event is kept iff it strictly precedes (and does not follow) the last receive of at least one
// node named by the violation fingerprint (:355-375).
//
// The closure is not a fixed-point iteration here.  Every edge points forward in trace order, and the closure of "all
// earlier same-receiver events" equals the closure of "the previous same-receiver event", so row u of the closed matrix
// is  {u} | row[next receive of u's machine] | OR of the rows of the messages u sent.  Walking the trace backwards, a
// finished row is OR-ed into the (unfinished, earlier) rows of its previous same-machine receive and of its parent:
// n steps of 4 x 64-bit words, by the four first lanes of the wave.  LDS: 8 KB matrix + 0.8 KB index arrays per wave.
#pragma once

#include "demi_device.hpp"

namespace demi {

constexpr uint32_t PROV_MAX = DEMI_DPOR_MAX_TRACE;          // 256 events
constexpr uint32_t PROV_WORDS = PROV_MAX / 64;              // 4 words per row
constexpr int PROV_WAVES = 4;                               // traces per workgroup

struct ProvArgs {
  const demi_dpor_trace_entry* traces;   // [n][stride]
  const uint32_t* trace_len;             // [n]
  const uint32_t* affected;              // [n] bit a = actor a is named by the violation (ViolationFingerprint.affectedNodes)
  uint32_t stride;
  uint64_t n;
  uint64_t* keep;                        // [n][PROV_WORDS] bit u = event u is kept
};

__host__ __device__ inline size_t prov_wave_bytes() { return (size_t)PROV_MAX * PROV_WORDS * 8 + PROV_MAX * 3 + 9 * 4; }

// BIG: the traces of a table with more than 8 actors (a 4-bit receiver field in the entries' words, 16 machines + the root's)
template <bool BIG>
__device__ __forceinline__ void k_provenance_body(const ProvArgs& args) {
  constexpr uint32_t MACHINES = BIG ? DEMI_MAX_ACTORS_BIG : DEMI_MAX_ACTORS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t item = (uint64_t)blockIdx.x * PROV_WAVES + wave;
  if (item >= args.n) return;                                // (no workgroup barrier below: waves are independent)
  unsigned char* base = smem + (size_t)wave * ((prov_wave_bytes() + 15) & ~(size_t)15);
  uint64_t* M = reinterpret_cast<uint64_t*>(base);           // [PROV_MAX][PROV_WORDS]
  uint8_t* rcv = base + (size_t)PROV_MAX * PROV_WORDS * 8;   // receiver (0..7; BIG: 0..15), MACHINES = the root's own "machine", 255 = not a receive
  uint8_t* par = rcv + PROV_MAX;
  uint8_t* prev = par + PROV_MAX;                            // previous receive of the same machine (255: none)
  const uint32_t n = args.trace_len[item] < PROV_MAX ? args.trace_len[item] : PROV_MAX;
  const demi_dpor_trace_entry* tr = args.traces + item * (uint64_t)args.stride;

  for (uint32_t u = lane; u < PROV_MAX; u += 64) {
    uint32_t r = 255, p = 0;
    if (u < n) {
      const demi_dpor_trace_entry e = tr[u];
      if (u == 0) r = MACHINES;                              // the root is a MsgEvent("null", "null", null) (:283-285)
      else if (e.kind == 1) r = (e.word >> 5) & (MACHINES - 1u);
      p = e.parent;
    }
    rcv[u] = (uint8_t)r; par[u] = (uint8_t)p;
    for (uint32_t j = 0; j < PROV_WORDS; j++) M[u * PROV_WORDS + j] = 0;
  }
  // (the nine lanes below read what every lane has just written: same ordering point as further down - no instruction, it only
  // says so to the compiler, and to the lock-step emulator of the CPU suite, whose lanes really run one after the other)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // prev[u]: by one lane per machine (9 of them; BIG: 17), in trace order
  if (lane < MACHINES + 1u) {
    uint32_t last = 255;
    for (uint32_t u = 0; u < n; u++)
      if (rcv[u] == lane) { prev[u] = (uint8_t)last; last = u; }
  }
  for (uint32_t u = lane; u < n; u += 64) if (rcv[u] == 255) prev[u] = 255;
  // (a wave's LDS instructions complete in order; the compiler must not move them across these points)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // backward sweep: lanes 0..3 own one word of every row
  if (lane < PROV_WORDS) {
    for (uint32_t s = n; s-- > 0;) {
      if (rcv[s] == 255) continue;                           // a marker neither precedes nor follows anything
      uint64_t row = M[s * PROV_WORDS + lane];
      if ((s >> 6) == lane) row |= 1ull << (s & 63);         // (u, u)
      M[s * PROV_WORDS + lane] = row;
      const uint32_t q = prev[s];
      if (q != 255) M[q * PROV_WORDS + lane] |= row;
      const uint32_t p = par[s];
      if (s != 0 && p < s && rcv[p] != 255) M[p * PROV_WORDS + lane] |= row;     // "sends that result from the receive"
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

  // the last receive of every affected node
  uint32_t last_of[MACHINES];
  uint32_t n_last = 0;
  const uint32_t aff = args.affected[item];
  for (uint32_t a = 0; a < MACHINES; a++) {
    if (!((aff >> a) & 1u)) continue;
    uint32_t l = 0xFFFFu;
    for (uint32_t u = 0; u < n; u++) if (rcv[u] == a) l = u;       // (uniform over the wave; n <= 256)
    if (l != 0xFFFFu) last_of[n_last++] = l;
  }
  // kept iff NOT for every o: concurrent(o, u) or o happens-before u  <=>  exists o: u -> o and not o -> u
  for (uint32_t w = 0; w < PROV_WORDS; w++) {
    const uint32_t u = w * 64 + lane;
    bool keep = false;
    if (u < n) {
      for (uint32_t k = 0; k < n_last; k++) {
        const uint32_t o = last_of[k];
        const bool u_o = (M[u * PROV_WORDS + (o >> 6)] >> (o & 63)) & 1ull;
        const bool o_u = (M[o * PROV_WORDS + (u >> 6)] >> (u & 63)) & 1ull;
        keep |= u_o && !o_u;
      }
    }
    const uint64_t bits = __ballot(keep);
    if (lane == 0) args.keep[item * PROV_WORDS + w] = bits;
  }
}
__global__ __launch_bounds__(PROV_WAVES * 64) void k_provenance(const ProvArgs args) { k_provenance_body<false>(args); }
__global__ __launch_bounds__(PROV_WAVES * 64) void k_provenance_big(const ProvArgs args) { k_provenance_body<true>(args); }

}  // namespace demi
