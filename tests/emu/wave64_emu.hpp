// wave64_emu.hpp - TEST INFRASTRUCTURE (CPU suite only; the product never includes or loads this).
//
// A host stand-in for the few device-side constructs the kernels of demi_amd/csrc use, so that the UNMODIFIED kernel sources
// (and the host library around them) compile with g++ and run on the CPU under a lock-step emulation of the CDNA execution
// model: a workgroup is a set of fibers (one per work-item), a wavefront is 64 consecutive fibers, and every cross-lane
// operation (__ballot, readlane, __shfl, wave_barrier, __syncthreads) is a rendezvous of the lanes that are still running.
// That is exact for kernels whose cross-lane operations sit at wave-uniform points of the control flow - which is how the
// kernels here are written - and the runtime (w64rt.cpp) CHECKS it: if the running lanes of a wave wait at different source
// lines, the launch aborts with both lines instead of computing something a GPU would not.
//
// What this is for: the parity tests of the real kernel sources against oracle/ without a GPU (tests/test_emu_*_cpu.py).
// What it is not: a fallback.  Nothing under demi_amd/ knows it exists.
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

// ------------------------------------------------------------------ qualifiers
#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
// LDS: one workgroup runs at a time on an OS thread, all of its work-items are fibers of that thread
#define __shared__ thread_local

// ------------------------------------------------------------------ vector types
struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
struct alignas(16) ulonglong2 { unsigned long long x, y; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 v; v.x = x; v.y = y; return v; }

// ------------------------------------------------------------------ the runtime (w64rt.cpp)
struct w64_item {             // what a work-item knows about itself
  dim3 tid, bid, bdim, gdim;
};
enum : int { W64_BALLOT = 1, W64_READLANE, W64_READFIRST, W64_SHFL, W64_SHFL_UP, W64_WAVE_BARRIER, W64_SYNCTHREADS };
struct w64_xchg {             // result of a wave rendezvous: what every arriving lane deposited
  uint64_t live;              // lanes that took part
  const uint64_t* v;          // [64]
  unsigned lane;              // the caller's lane
};
extern "C" {
const w64_item* w64_self();
w64_xchg w64_exchange(uint64_t value, int kind, int line);
void w64_block_barrier(int line);
}
namespace demi { extern thread_local unsigned char smem[]; }     // the dynamic LDS of the running workgroup

#define threadIdx (w64_self()->tid)
#define blockIdx (w64_self()->bid)
#define blockDim (w64_self()->bdim)
#define gridDim (w64_self()->gdim)

// ------------------------------------------------------------------ cross-lane operations (macros: the source line is the site id)
static inline uint64_t w64_ballot_(bool p, int line) {
  const w64_xchg x = w64_exchange(p ? 1u : 0u, W64_BALLOT, line);
  uint64_t m = 0;
  for (unsigned l = 0; l < 64; l++) if (((x.live >> l) & 1u) && x.v[l]) m |= 1ull << l;
  return m;
}
template <class T> static inline T w64_bits_to_(uint64_t b) { T t; memcpy(&t, &b, sizeof t); return t; }
template <class T> static inline uint64_t w64_to_bits_(T t) { static_assert(sizeof(T) <= 8, "w64: value too wide"); uint64_t b = 0; memcpy(&b, &t, sizeof t); return b; }
static inline int w64_readlane_(int v, int src, int line) {
  const w64_xchg x = w64_exchange((uint32_t)v, W64_READLANE, line);
  return (int)(uint32_t)x.v[src & 63];        // (reading a lane that is not running returns what it last held: 0 here)
}
static inline int w64_readfirst_(int v, int line) {
  const w64_xchg x = w64_exchange((uint32_t)v, W64_READFIRST, line);
  return (int)(uint32_t)x.v[__builtin_ctzll(x.live)];
}
template <class T> static inline T w64_shfl_(T v, int src, int line) {
  const w64_xchg x = w64_exchange(w64_to_bits_(v), W64_SHFL, line);
  return w64_bits_to_<T>(x.v[src & 63]);
}
template <class T> static inline T w64_shfl_up_(T v, unsigned d, int line) {
  const w64_xchg x = w64_exchange(w64_to_bits_(v), W64_SHFL_UP, line);
  return x.lane >= d ? w64_bits_to_<T>(x.v[x.lane - d]) : v;
}
#define __ballot(P) w64_ballot_((P), __LINE__)
#define __builtin_amdgcn_readlane(V, L) w64_readlane_((V), (L), __LINE__)
#define __builtin_amdgcn_readfirstlane(V) w64_readfirst_((V), __LINE__)
#define __shfl(V, L) w64_shfl_((V), (L), __LINE__)
#define __shfl_up(V, D) w64_shfl_up_((V), (D), __LINE__)
// on the GPU only the compiler's side of the order (a wave runs in lock step); here the lanes really have to meet
#define __builtin_amdgcn_wave_barrier() ((void)w64_exchange(0, W64_WAVE_BARRIER, __LINE__))
#define __syncthreads() w64_block_barrier(__LINE__)
#define __builtin_amdgcn_fence(ORDER, SCOPE) __atomic_thread_fence(__ATOMIC_SEQ_CST)
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// ------------------------------------------------------------------ integer builtins
template <class T> static inline T min(T a, T b) { return b < a ? b : a; }      // (HIP's global min / max)
template <class T> static inline T max(T a, T b) { return a < b ? b : a; }
static inline uint32_t min(uint32_t a, int b) { return min(a, (uint32_t)b); }
static inline uint32_t max(uint32_t a, int b) { return max(a, (uint32_t)b); }
static inline unsigned long long wall_clock64() { return 0; }
static inline int __popcll(uint64_t x) { return __builtin_popcountll(x); }
static inline int __popc(uint32_t x) { return __builtin_popcount(x); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
static inline uint32_t __umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
// v_perm_b32: byte k of the result is picked by selector byte k from the 8 bytes {s0 (4..7), s1 (0..3)}
static inline uint32_t w64_perm_(uint32_t s0, uint32_t s1, uint32_t sel) {
  const uint64_t both = ((uint64_t)s0 << 32) | s1;
  uint32_t r = 0;
  for (int k = 0; k < 4; k++) {
    const uint32_t s = (sel >> (8 * k)) & 0xFFu;
    uint32_t b;
    if (s <= 7) b = (uint32_t)(both >> (8 * s)) & 0xFFu;
    else if (s == 8) b = ((s1 >> 15) & 1u) ? 0xFFu : 0u;
    else if (s == 9) b = ((s1 >> 31) & 1u) ? 0xFFu : 0u;
    else if (s == 10) b = ((s0 >> 15) & 1u) ? 0xFFu : 0u;
    else if (s == 11) b = ((s0 >> 31) & 1u) ? 0xFFu : 0u;
    else if (s == 12) b = 0u;
    else b = 0xFFu;
    r |= b << (8 * k);
  }
  return r;
}
#define __builtin_amdgcn_perm(A, B, S) w64_perm_((A), (B), (S))
// v_bfe_i32
static inline int w64_sbfe_(uint32_t src, uint32_t off, uint32_t width) {
  off &= 31u; width &= 31u;
  if (width == 0) return 0;
  const uint32_t x = (off + width >= 32) ? (src >> off) : ((src >> off) & ((1u << width) - 1u));
  const uint32_t w = (off + width >= 32) ? 32 - off : width;
  return (int)(x << (32 - w)) >> (32 - w);
}
#define __builtin_amdgcn_sbfe(S, O, W) w64_sbfe_((S), (O), (W))

// ------------------------------------------------------------------ atomics (workgroups of one launch may run on several OS threads)
template <class T, class U> static inline T atomicAdd(T* p, U v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicSub(T* p, U v) { return __atomic_fetch_sub(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicOr(T* p, U v) { return __atomic_fetch_or(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicAnd(T* p, U v) { return __atomic_fetch_and(p, (T)v, __ATOMIC_SEQ_CST); }
template <class T, class U> static inline T atomicMax(T* p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old < (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <class T, class U> static inline T atomicMin(T* p, U v) {
  T old = __atomic_load_n(p, __ATOMIC_SEQ_CST);
  while (old > (T)v && !__atomic_compare_exchange_n(p, &old, (T)v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
  return old;
}
template <class T, class U, class V> static inline T atomicCAS(T* p, U expect, V desired) {
  T e = (T)expect;
  __atomic_compare_exchange_n(p, &e, (T)desired, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return e;
}
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(P, ORDER, SCOPE) __atomic_load_n((P), __ATOMIC_SEQ_CST)
#define __hip_atomic_store(P, V, ORDER, SCOPE) __atomic_store_n((P), (V), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_add(P, V, ORDER, SCOPE) __atomic_fetch_add((P), (V), __ATOMIC_SEQ_CST)
#define __hip_atomic_fetch_or(P, V, ORDER, SCOPE) __atomic_fetch_or((P), (V), __ATOMIC_SEQ_CST)

// ------------------------------------------------------------------ gfx950 inline assembly (cycle counters of the diagnostic builds,
// the instruction-mix probes of k_probe.hpp): there is nothing to time here, the statements vanish and their outputs stay
// as they were.  (`asm volatile ( ... )` -> `asm ( ... )` -> nothing; every standard header is in before this point of a
// translation unit that uses the emulator, see hip/hip_runtime.h.)
#define W64_DROP_ASM 1
