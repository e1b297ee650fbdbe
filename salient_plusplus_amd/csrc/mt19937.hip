// a1: std::mt19937-compatible bulk stream generator (reference: fast_sampler/sample_cpu.hpp:11,
// seeded at fast_sampler.cpp:994).  hipRAND/rocRAND's MT variants do not reproduce the single
// std::mt19937 stream, so the generator is written out here.
//
// The state recurrence  x[n] = x[n-227] ^ twist(x[n-624], x[n-623])  has a minimum lag of 227
// words, so 227 outputs are data-parallel per step.  One 64-lane wavefront owns one stream; the
// last 1024 words of x live in an LDS ring (4 KiB), each lane produces <= 4 words per step.
// Output i of the engine is temper(x[624 + i]).
#include "spp_internal.h"

namespace spp {

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// Device routine shared with the sampler: the calling WAVEFRONT (blockDim.x == 64) writes raw
// outputs skip .. skip+n-1 of mt19937(seed) to out[0..n).  x is a 1024-word LDS ring.
__device__ void mt19937_wave_fill(uint32_t* x, uint32_t seed, int64_t skip, int64_t n, uint32_t* out) {
  const int lane = threadIdx.x;
  if (lane == 0) {
    // std::mt19937::seed(value): x[i] = 1812433253 * (x[i-1] ^ (x[i-1] >> 30)) + i
    uint32_t p = seed;
    x[0] = p;
    for (int i = 1; i < 624; ++i) {
      p = 1812433253u * (p ^ (p >> 30)) + (uint32_t)i;
      x[i] = p;
    }
  }
  __syncthreads();
  const int64_t total = skip + n;
  for (int64_t base = 0; base < total; base += 227) {
    uint32_t v[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = lane + 64 * m;
      if (j < 227) {
        const uint32_t nn = (uint32_t)((624 + base + j) & 1023);
        const uint32_t a = x[(nn - 624u) & 1023u];
        const uint32_t b = x[(nn - 623u) & 1023u];
        const uint32_t c = x[(nn - 227u) & 1023u];
        v[m] = c ^ mt_twist(a, b);
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int j = lane + 64 * m;
      if (j < 227) {
        x[(uint32_t)((624 + base + j) & 1023)] = v[m];
        const int64_t i = base + j;
        if (i >= skip && i < total) out[i - skip] = mt_temper(v[m]);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(64) void k_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out) {
  __shared__ uint32_t x[1024];
  mt19937_wave_fill(x, seed, skip, n, out);
}

}  // namespace spp

extern "C" spp_status spp_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out_dev, void* stream) {
  SPP_REQUIRE(n >= 0 && skip >= 0, "spp_mt19937_fill: n (%lld) and skip (%lld) must be >= 0", (long long)n,
              (long long)skip);
  if (n == 0) return SPP_OK;
  SPP_REQUIRE(out_dev != nullptr, "spp_mt19937_fill: out_dev is NULL");
  hipLaunchKernelGGL(spp::k_mt19937_fill, dim3(1), dim3(64), 0, spp::as_stream(stream), seed, skip, n, out_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
