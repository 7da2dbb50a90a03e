// k_probe.hpp — measurement helpers behind demi_device_probe / demi_calib_rw (bench.py, tools/profile_r5.sh).
// Not on the product path: they exist so that the bench line's clock and issue-rate figures are measured on the box
// the line is printed on, and so that the rocprofv3 FETCH_SIZE / WRITE_SIZE counters can be calibrated against a known
// byte count in K1's own access pattern (4 B per lane, one 256-byte row per wave and slot).
#pragma once

#include "demi_device.hpp"

namespace demi {

// out[wave] = {shader cycles (s_memtime), constant-rate ticks (wall_clock64: 100 MHz), integer VALU instructions issued}
// Every wave runs `iters` passes of 64 independent-by-8 integer VALU instructions (v_mad_u32_u24 / v_xor / v_add chains on
// 8 accumulators), so the issue rate is not limited by the latency of a single dependence chain.
__global__ __launch_bounds__(256) void k_probe_valu(unsigned long long* __restrict__ out, uint32_t iters, uint32_t seed) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u;
  unsigned long long t0, t1;
  const long long w0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {     // 8 x 8 = 64 VALU instructions per pass
      asm volatile("v_mad_u32_u24 %0, %0, 3, %1\n\tv_mad_u32_u24 %1, %1, 5, %2\n\tv_mad_u32_u24 %2, %2, 7, %3\n\t"
                   "v_mad_u32_u24 %3, %3, 9, %4\n\tv_xor_b32 %4, %4, %5\n\tv_add_u32 %5, %5, %6\n\t"
                   "v_add_u32 %6, %6, %7\n\tv_xor_b32 %7, %7, %0"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
  const long long w1 = wall_clock64();
  const uint32_t keep = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    o[0] = t1 - t0; o[1] = (unsigned long long)(w1 - w0); o[2] = (unsigned long long)iters * 64ull; o[3] = keep;
  }
}

// The same measurement for the scalar ALU (kind 1: 64 SALU instructions per pass on 4 scalar accumulators), for an even
// mix (kind 2: 64 VALU + 64 SALU per pass, alternating) and for the shape a divergent `if` compiles to (kind 3: 16 x
// {v_and, v_cmp, s_and_saveexec_b64, s_cbranch_execz, 2 body VALU, s_or_b64 exec} per pass): K1's compiled handlers issue
// two SALU instructions for every three VALU ones, so what a SIMD can issue per cycle of each kind - alone and together -
// bounds it.  o[2] counts VALU + SALU instructions.
__global__ __launch_bounds__(256) void k_probe_mix(unsigned long long* __restrict__ out, uint32_t iters, uint32_t seed, uint32_t kind) {
  uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
  uint32_t s0 = __builtin_amdgcn_readfirstlane(seed + blockIdx.x), s1 = s0 * 3u, s2 = s0 * 5u, s3 = s0 * 7u;
  unsigned long long t0, t1;
  const long long w0 = wall_clock64();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
  unsigned long long per_pass = 0;
  if (kind == 1) {
    per_pass = 64;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 8; k++)
        asm volatile("s_mul_i32 %0, %0, 3\n\ts_add_u32 %1, %1, %2\n\ts_xor_b32 %2, %2, %3\n\ts_add_u32 %3, %3, %0\n\t"
                     "s_lshl_b32 %0, %0, 1\n\ts_xor_b32 %1, %1, %3\n\ts_add_u32 %2, %2, %0\n\ts_xor_b32 %3, %3, %1"
                     : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
  } else if (kind == 2) {
    per_pass = 128;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++)
        asm volatile("v_mad_u32_u24 %0, %0, 3, %1\n\ts_add_u32 %4, %4, %5\n\tv_mad_u32_u24 %1, %1, 5, %2\n\ts_xor_b32 %5, %5, %6\n\t"
                     "v_xor_b32 %2, %2, %3\n\ts_add_u32 %6, %6, %7\n\tv_add_u32 %3, %3, %0\n\ts_xor_b32 %7, %7, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
  } else {
    per_pass = 16 * 7;
    for (uint32_t i = 0; i < iters; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) {
        uint32_t tmp;
        unsigned long long save;
        asm volatile("v_and_b32 %3, 1, %1\n\tv_cmp_ne_u32 vcc, 0, %3\n\ts_and_saveexec_b64 %4, vcc\n\ts_cbranch_execz L_probe_skip%=\n\t"
                     "v_mad_u32_u24 %1, %1, 5, %2\n\tv_xor_b32 %2, %2, %0\nL_probe_skip%=:\n\ts_or_b64 exec, exec, %4"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "=&v"(tmp), "=&s"(save) : : "vcc", "scc");
      }
      a0 = a0 * 3u + 1u;
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
  const long long w1 = wall_clock64();
  const uint32_t keep = a0 ^ a1 ^ a2 ^ a3 ^ s0 ^ s1 ^ s2 ^ s3;
  if ((threadIdx.x & 63) == 0) {
    unsigned long long* o = out + ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 4;
    o[0] = t1 - t0; o[1] = (unsigned long long)(w1 - w0); o[2] = (unsigned long long)iters * per_pass; o[3] = keep;
  }
}

// mode 0: every lane writes 4 B per row, rows of 256 B per wave (K1's [slot][lane] scratch pattern); mode 1: reads them
// back (sum to `sink`); mode 2 / 3: the same with 16 B per lane (the streaming pattern of the verdict array).
__global__ __launch_bounds__(256) void k_calib_rw(uint32_t* __restrict__ buf, uint64_t words, uint32_t mode, uint32_t* __restrict__ sink) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t acc = 0;
  if (mode == 0) { for (; i < words; i += stride) buf[i] = (uint32_t)i; }
  else if (mode == 1) { for (; i < words; i += stride) acc += buf[i]; }
  else if (mode == 2) { uint4* b4 = reinterpret_cast<uint4*>(buf); for (; i < words / 4; i += stride) b4[i] = make_uint4((uint32_t)i, 1, 2, 3); }
  else { const uint4* b4 = reinterpret_cast<const uint4*>(buf); for (; i < words / 4; i += stride) { const uint4 x = b4[i]; acc += x.x ^ x.y ^ x.z ^ x.w; } }
  if (acc == 0x12345678u) *sink = acc;
}

}  // namespace demi
