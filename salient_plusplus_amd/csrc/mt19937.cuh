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
// The minimum lag doubles to 454 words, so ONE 64-lane wavefront produces 454 outputs per step
// (<= 8 per lane) from a 2048-word LDS ring.  A single wavefront needs no s_barrier: its DS
// instructions execute in order, only the compiler has to be kept from reordering them
// (wavefront-scope fences) -- a workgroup barrier would also drain the global stores each step.
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

namespace spp {

constexpr int kMtRing = 2048;        // words of x kept in LDS
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

__device__ __forceinline__ void mt_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Called by ONE wavefront (64 lanes).  x: LDS ring of kMtRing words.
// Seeds the ring and produces outputs 0..623 with the base recurrence.
// Outputs i in [skip, skip+cap) are written to out[i - skip].
__device__ __forceinline__ void mt_wave_seed(uint32_t* x, uint32_t seed, int64_t skip, int64_t cap, uint32_t* out) {
  const int lane = threadIdx.x & 63;
  if (lane == 0) {
    uint32_t p = seed;  // std::mt19937::seed(value): x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i
    x[0] = p;
    for (int i = 1; i < 624; ++i) {
      p = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
      x[i] = p;
    }
  }
  mt_wave_sync();
  for (int base = 0; base < 624; base += 227) {
    const int cnt = (624 - base) < 227 ? (624 - base) : 227;
    uint32_t v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = lane + 64 * m;
      if (j < cnt) {
        const int n = 624 + base + j;
        const uint32_t a = x[n - 624], b = x[n - 623], c = x[n - 227];
        v[m] = c ^ ((a & 0x80000000u) >> 1) ^ mt_L(b);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = lane + 64 * m;
      if (j < cnt) {
        x[624 + base + j] = v[m];
        const int64_t i = base + j;
        if (i >= skip && i - skip < cap) out[i - skip] = mt_temper(v[m]);
      }
    }
    mt_wave_sync();
  }
}

// Continues the stream from output index `pos` (>= 624) until at least `need` outputs exist;
// returns the new position.  The ring holds x[n & (kMtRing-1)] for the last 2048 words.
__device__ __forceinline__ int64_t mt_wave_advance(uint32_t* x, int64_t pos, int64_t need, int64_t skip, int64_t cap,
                                                   uint32_t* out) {
  const int lane = threadIdx.x & 63;
  constexpr uint32_t M = kMtRing - 1;
  for (; pos < need; pos += kMtStep) {
    uint32_t v[8];
    const uint32_t n0 = (uint32_t)((624 + pos) & M);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = lane + 64 * m;
      if (j < kMtStep) {
        const uint32_t n = n0 + (uint32_t)j;
        const uint32_t p = x[(n - 454u) & M];
        const uint32_t c = x[(n - 1247u) & M];
        const uint32_t d = x[(n - 1246u) & M];
        const uint32_t g = ((c & 1u) ? 0x40000000u : 0u) ^ ((c & 0x80000000u) >> 2);
        v[m] = p ^ g ^ mt_L(mt_L(d));
      }
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const int j = lane + 64 * m;
      if (j < kMtStep) {
        x[(n0 + (uint32_t)j) & M] = v[m];
        const int64_t i = pos + j;
        if (i >= skip && i - skip < cap) out[i - skip] = mt_temper(v[m]);
      }
    }
    mt_wave_sync();
  }
  return pos;
}

}  // namespace spp
