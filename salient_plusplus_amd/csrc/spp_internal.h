// Internal helpers shared by the HIP translation units of libspp_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "spp.h"

namespace spp {

// thread-local last error message (spp_last_error)
void set_error(const char* fmt, ...);

#define SPP_HIP_TRY(expr)                                                              \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      ::spp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                       __LINE__);                                                      \
      return SPP_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

#define SPP_REQUIRE(cond, ...)         \
  do {                                 \
    if (!(cond)) {                     \
      ::spp::set_error(__VA_ARGS__);   \
      return SPP_ERR_INVALID;          \
    }                                  \
  } while (0)

#define SPP_TRY(expr)              \
  do {                             \
    spp_status s_ = (expr);        \
    if (s_ != SPP_OK) return s_;   \
  } while (0)

// live event timing (api.hip); kind is one of SPP_PROF_*
int prof_begin(int kind, hipStream_t st, int64_t units);
void prof_end(int kind, int idx, hipStream_t st);

// Per-device error word in pinned, device-visible host memory (spp_async_errors, include/spp.h):
// kernels raise SPP_AERR_* bits with a system-scope atomic, the host reads it without synchronising.
// Returns nullptr when it cannot be allocated (kernels then only clamp).
int32_t* async_err_word(int device);
int32_t* async_err_word_current();

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

constexpr int kWave = 64;  // CDNA4 wavefront

// ---------------------------------------------------------------------------
// device-side primitives
// ---------------------------------------------------------------------------

__device__ __forceinline__ void raise_async_error(int32_t* word, int32_t bit) {
  if (word) __hip_atomic_fetch_or(word, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// inclusive scan across one 64-lane wavefront
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    T o = __shfl_up(v, d, kWave);
    if (lane >= d) v += o;
  }
  return v;
}

// Exclusive scan over a workgroup of NT threads (NT multiple of 64, <= 1024).
// Returns the exclusive prefix of `v` for this thread; *total receives the workgroup sum.
// `lds` must hold NT/64 + 1 elements of T.  Contains two barriers.
template <typename T, int NT>
__device__ __forceinline__ T block_exclusive_scan(T v, T* lds, T* total) {
  constexpr int NW = NT / kWave;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = threadIdx.x / kWave;
  T inc = wave_inclusive_scan(v);
  if (lane == kWave - 1) lds[wid] = inc;
  __syncthreads();
  if (wid == 0) {
    T w = (lane < NW) ? lds[lane] : T(0);
    T winc = wave_inclusive_scan(w);
    if (lane < NW) lds[lane] = winc - w;  // exclusive offset of each wave
    if (lane == NW - 1) lds[NW] = winc;   // workgroup total
  }
  __syncthreads();
  T res = inc - v + lds[wid];
  *total = lds[NW];
  return res;
}

// ---- open-addressing node table: one 64-bit word per slot, (key << 32) | value ----
constexpr unsigned long long kEmptySlot = ~0ull;

__device__ __forceinline__ uint32_t hash_node(uint32_t k) {
  // Fibonacci hashing on the 32-bit node id
  return (uint32_t)((k * 0x9E3779B97F4A7C15ull) >> 32);
}

}  // namespace spp
