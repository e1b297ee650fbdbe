// a5 / K2: row gather  dst[i,:] = src[idx[i],:]  (reference serial_index,
// fast_sampler.cpp:238-279).  HBM-bound: per output row the algorithmic traffic is
// read row + write row + read index.  A group of LPR lanes (power of two) moves one row with
// VEC-byte accesses; consecutive groups take consecutive output rows so a wavefront's stores
// cover one contiguous span of dst, and every group keeps UNROLL independent rows in flight.
#include "spp_internal.h"

namespace spp {

template <int VEC> struct vec_of;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <> struct vec_of<16> { using type = u32x4; };
template <> struct vec_of<8> { using type = u32x2; };
template <> struct vec_of<4> { using type = uint32_t; };
template <> struct vec_of<2> { using type = uint16_t; };
template <> struct vec_of<1> { using type = uint8_t; };

constexpr int kGatherThreads = 256;
constexpr int kGatherUnroll = 4;

template <int VEC, typename IdxT>
__global__ __launch_bounds__(kGatherThreads) void k_gather_rows(const char* __restrict__ src,
                                                                 const IdxT* __restrict__ idx, int64_t n,
                                                                 int64_t row_bytes, int chunks, int lpr_log2,
                                                                 char* __restrict__ dst) {
  using V = typename vec_of<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int g = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & (lpr - 1);
  const int gpb = kGatherThreads >> lpr_log2;  // row groups per workgroup
  const int64_t rows_per_iter = (int64_t)gpb * kGatherUnroll;
  for (int64_t base = (int64_t)blockIdx.x * rows_per_iter; base < n; base += (int64_t)gridDim.x * rows_per_iter) {
    const V* s[kGatherUnroll];
    V* d[kGatherUnroll];
    bool ok[kGatherUnroll];
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) {
      const int64_t r = base + (int64_t)u * gpb + g;
      ok[u] = r < n;
      const int64_t sr = ok[u] ? (int64_t)idx[r] : 0;
      s[u] = reinterpret_cast<const V*>(src + sr * row_bytes);
      d[u] = reinterpret_cast<V*>(dst + r * row_bytes);
    }
    for (int c = l; c < chunks; c += lpr) {
      V v[kGatherUnroll];
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u)
        if (ok[u]) v[u] = s[u][c];
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u)
        if (ok[u]) __builtin_nontemporal_store(v[u], &d[u][c]);
    }
  }
}

template <typename IdxT>
static spp_status launch_gather(const void* src, int64_t row_bytes, const IdxT* idx, int64_t n, void* dst,
                                hipStream_t st) {
  if (n <= 0 || row_bytes <= 0) return SPP_OK;
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)row_bytes;
  int vec = 16;
  while (vec > 1 && (a % vec) != 0) vec >>= 1;
  const int chunks = (int)(row_bytes / vec);
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < chunks && lpr_log2 < 6) ++lpr_log2;
  const int gpb = kGatherThreads >> lpr_log2;
  const int64_t rows_per_iter = (int64_t)gpb * kGatherUnroll;
  int64_t grid = ceil_div(n, rows_per_iter);
  const int64_t max_grid = 256 * 16;  // 256 CUs x 16 workgroups: grid-stride beyond that
  if (grid > max_grid) grid = max_grid;
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  const int prof = prof_begin(SPP_PROF_GATHER, st, n);
#define SPP_LAUNCH_GATHER(V)                                                                              \
  hipLaunchKernelGGL((k_gather_rows<V, IdxT>), dim3((unsigned)grid), dim3(kGatherThreads), 0, st, s, idx, n, \
                     row_bytes, chunks, lpr_log2, d)
  switch (vec) {
    case 16: SPP_LAUNCH_GATHER(16); break;
    case 8: SPP_LAUNCH_GATHER(8); break;
    case 4: SPP_LAUNCH_GATHER(4); break;
    case 2: SPP_LAUNCH_GATHER(2); break;
    default: SPP_LAUNCH_GATHER(1); break;
  }
#undef SPP_LAUNCH_GATHER
  prof_end(SPP_PROF_GATHER, prof, st);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// used by sampler.hip (int32 node list of a slot)
spp_status gather_rows_i32(const void* src, int64_t row_bytes, const int32_t* idx, int64_t n, void* dst,
                           hipStream_t st) {
  return launch_gather<int32_t>(src, row_bytes, idx, n, dst, st);
}

// ---- to_row_major (reference fast_sampler.cpp:281-308): out[r*tc + c] = in[c*tr + r] ----
template <typename T>
__global__ __launch_bounds__(256) void k_to_row_major(const T* __restrict__ in, int64_t tr, int64_t tc,
                                                       T* __restrict__ out) {
  __shared__ T tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int64_t r0 = (int64_t)blockIdx.x * 32, c0 = (int64_t)blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {  // read along r (contiguous in the column-major input)
    const int64_t c = c0 + k, r = r0 + tx;
    if (c < tc && r < tr) tile[k][tx] = in[c * tr + r];
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {  // write along c (contiguous in the row-major output)
    const int64_t r = r0 + k, c = c0 + tx;
    if (r < tr && c < tc) out[r * tc + c] = tile[tx][k];
  }
}

}  // namespace spp

extern "C" spp_status spp_gather_rows(const void* src_dev, int64_t src_rows, int64_t row_bytes, const void* idx_dev,
                                      int idx_elem_bytes, int64_t n_idx, int64_t n_out, void* dst_dev,
                                      void* stream) {
  SPP_REQUIRE(row_bytes >= 0 && n_idx >= 0 && n_out >= 0, "spp_gather_rows: negative size");
  SPP_REQUIRE(idx_elem_bytes == 8 || idx_elem_bytes == 4, "spp_gather_rows: idx_elem_bytes must be 4 or 8, got %d",
              idx_elem_bytes);
  const int64_t n = n_idx < n_out ? n_idx : n_out;  // reference :253  min(idx.numel(), n)
  if (n == 0 || row_bytes == 0) return SPP_OK;
  SPP_REQUIRE(src_dev && idx_dev && dst_dev, "spp_gather_rows: NULL buffer");
  (void)src_rows;
  if (idx_elem_bytes == 8)
    return spp::launch_gather<int64_t>(src_dev, row_bytes, static_cast<const int64_t*>(idx_dev), n, dst_dev,
                                       spp::as_stream(stream));
  return spp::launch_gather<int32_t>(src_dev, row_bytes, static_cast<const int32_t*>(idx_dev), n, dst_dev,
                                     spp::as_stream(stream));
}

extern "C" spp_status spp_to_row_major(const void* src_dev, int64_t rows, int64_t cols, int elem_bytes, void* dst_dev,
                                       void* stream) {
  SPP_REQUIRE(rows >= 0 && cols >= 0, "spp_to_row_major: only support 2D tensors with non-negative sizes");
  if (rows == 0 || cols == 0) return SPP_OK;
  SPP_REQUIRE(src_dev && dst_dev, "spp_to_row_major: NULL buffer");
  dim3 grid((unsigned)spp::ceil_div(rows, 32), (unsigned)spp::ceil_div(cols, 32));
  hipStream_t st = spp::as_stream(stream);
  switch (elem_bytes) {
    case 8: hipLaunchKernelGGL(spp::k_to_row_major<uint64_t>, grid, dim3(256), 0, st, (const uint64_t*)src_dev, rows, cols, (uint64_t*)dst_dev); break;
    case 4: hipLaunchKernelGGL(spp::k_to_row_major<uint32_t>, grid, dim3(256), 0, st, (const uint32_t*)src_dev, rows, cols, (uint32_t*)dst_dev); break;
    case 2: hipLaunchKernelGGL(spp::k_to_row_major<uint16_t>, grid, dim3(256), 0, st, (const uint16_t*)src_dev, rows, cols, (uint16_t*)dst_dev); break;
    case 1: hipLaunchKernelGGL(spp::k_to_row_major<uint8_t>, grid, dim3(256), 0, st, (const uint8_t*)src_dev, rows, cols, (uint8_t*)dst_dev); break;
    default: spp::set_error("spp_to_row_major: unsupported element size %d", elem_bytes); return SPP_ERR_INVALID;
  }
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
