// a1: std::mt19937-compatible bulk stream generator, shared by mt19937.hip and sampler.hip
// (reference: `thread_local std::mt19937 gen`, fast_sampler/sample_cpu.hpp:11, seeded at
// fast_sampler/fast_sampler.cpp:994).  hipRAND/rocRAND's MT variants do not reproduce the single
// std::mt19937 stream, so the generator is written out here.
//
// Sequence view: x[0..623] is the seeded state, x[n] = x[n-227] ^ f(x[n-624], x[n-623]) for
// n >= 624, and engine output i is temper(x[624+i]).  With f(a,b) = U(a) ^ L(b),
//   U(a) = (a & 0x80000000) >> 1,   L(b) = ((b & 0x7fffffff) >> 1) ^ ((b & 1) ? 0x9908b0df : 0)
// the recurrence is GF(2)-linear: x = (S + F) x with S the 227-shift and F = U.shift624 + L.shift623.
// S and F commute, so x = (S + F)^2 x = (S^2 + F^2) x, and because U.U = 0:
//   x[n] = x[n-454] ^ G(x[n-1247]) ^ L(L(x[n-1246]))            for n >= 1248,
//   G(c) = U(L(c)) ^ L(U(c)) = ((c & 1) ? 0x40000000 : 0) ^ ((c & 0x80000000) >> 2).
// The minimum lag doubles to 454 words: one step produces 454 outputs, one per thread of a
// 512-thread workgroup (a lone wavefront is instruction-issue bound: ~1 us per step measured).
// The last 2048 words of x live in an LDS ring that is MIRRORED (x[n] stored at n&2047 and at
// (n&2047)+2048) so every tap is `base + immediate` without a wrap computation.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace spp {

constexpr int kMtThreads = 512;      // workgroup size of the generator kernels
constexpr int kMtRing = 2048;        // words of x kept (the LDS array is 2 * kMtRing: mirrored)
constexpr int kMtStep = 454;         // outputs per step of the doubled recurrence
constexpr int kMtSlack = 1280;       // a call may write up to this many outputs beyond `need`

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

__device__ __forceinline__ uint32_t mt_L(uint32_t b) {
  return ((b & 0x7fffffffu) >> 1) ^ ((b & 1u) ? 0x9908b0dfu : 0u);
}

// x: LDS array of 2*kMtRing words.  Seeds x[0..623] and produces outputs 0..623 (x[624..1247])
// with the base recurrence; outputs i in [skip, skip+cap) go to out[i - skip].
__device__ __forceinline__ void mt_block_seed(uint32_t* x, uint32_t seed, int64_t skip, int64_t cap, uint32_t* out) {
  const int t = threadIdx.x;
  if (t == 0) {
    uint32_t p = seed;  // std::mt19937::seed(value): x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i
    x[0] = p;
    x[kMtRing] = p;
    for (int i = 1; i < 624; ++i) {
      p = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
      x[i] = p;
      x[i + kMtRing] = p;
    }
  }
  __syncthreads();
  for (int base = 0; base < 624; base += 227) {
    const int cnt = (624 - base) < 227 ? (624 - base) : 227;
    if (t < cnt) {
      const int n = 624 + base + t;
      const uint32_t a = x[n - 624], b = x[n - 623], c = x[n - 227];
      const uint32_t v = c ^ ((a & 0x80000000u) >> 1) ^ mt_L(b);
      x[n] = v;
      x[n + kMtRing] = v;
      const int64_t i = base + t;
      if (i >= skip && i - skip < cap) out[i - skip] = mt_temper(v);
    }
    __syncthreads();
  }
}

// Continues the stream from output index `pos` (>= 624) until at least `need` outputs exist;
// returns the new position (uniform across the workgroup).
__device__ __forceinline__ int64_t mt_block_advance(uint32_t* x, int64_t pos, int64_t need, int64_t skip, int64_t cap,
                                                    uint32_t* out) {
  const int t = threadIdx.x;
  constexpr uint32_t M = kMtRing - 1;
  for (; pos < need; pos += kMtStep) {
    if (t < kMtStep) {
      const uint32_t b = ((uint32_t)(624 + pos) + (uint32_t)t) & M;  // ring slot of x[n]
      const uint32_t* xb = x + b + kMtRing;                          // mirrored: xb[-d] is x[n-d] for d <= 2048
      const uint32_t p = xb[-454];
      const uint32_t c = xb[-1247];
      const uint32_t d = xb[-1246];
      const uint32_t g = ((c & 1u) ? 0x40000000u : 0u) ^ ((c & 0x80000000u) >> 2);
      const uint32_t v = p ^ g ^ mt_L(mt_L(d));
      x[b] = v;
      x[b + kMtRing] = v;
      const int64_t i = pos + t;
      if (i >= skip && i - skip < cap) out[i - skip] = mt_temper(v);
    }
    __syncthreads();
  }
  return pos;
}

}  // namespace spp
