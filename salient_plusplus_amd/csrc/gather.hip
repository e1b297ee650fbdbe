// a5 / K2: row gather  dst[i,:] = src[idx[i],:]  (reference serial_index,
// fast_sampler.cpp:238-279).  HBM-bound: per output row the algorithmic traffic is
// read row + write row + read index.  A group of LPR lanes (power of two) moves one row with
// VEC-byte accesses; consecutive groups take consecutive output rows so a wavefront's stores
// cover one contiguous span of dst, and every group keeps UNROLL independent rows in flight.
#include "spp_internal.h"

#include "gather_body.hip.h"

namespace spp {

template <int VEC, typename IdxT, bool kNT>
__global__ __launch_bounds__(kGatherThreads) void k_gather_rows(const char* __restrict__ src, int64_t src_rows,
                                                                 const IdxT* __restrict__ idx, int64_t n,
                                                                 int64_t row_bytes, int chunks, int lpr_log2,
                                                                 char* __restrict__ dst, int64_t src_stride,
                                                                 int32_t* err) {
  gather_rows_checked_body<VEC, IdxT, kNT>(src, src_rows, idx, n, row_bytes, chunks, lpr_log2, dst, blockIdx.x,
                                           gridDim.x, src_stride, err, SPP_AERR_GATHER_INDEX);
}

template <typename IdxT>
static spp_status launch_gather(const void* src, int64_t src_rows, int64_t row_bytes, int64_t src_stride,
                                const IdxT* idx, int64_t n, void* dst, hipStream_t st) {
  if (n <= 0 || row_bytes <= 0) return SPP_OK;
  if (src_stride <= 0) src_stride = row_bytes;
  const GatherGeom gg = gather_geometry(src, dst, row_bytes, n, src_stride, /*allow_span=*/true);
  const int chunks = gg.chunks, lpr_log2 = gg.lpr_log2;
  const int64_t grid = gg.grid;
  const char* s = static_cast<const char*>(src);
  char* d = static_cast<char*>(dst);
  int32_t* err = async_err_word_current();
  const int prof = prof_begin(SPP_PROF_GATHER, st, n);
  static const bool nt = [] { const char* e = getenv("SPP_GATHER_NT"); return e ? atoi(e) != 0 : false; }();
#define SPP_LAUNCH_GATHER(V)                                                                                    \
  do {                                                                                                          \
    if (nt)                                                                                                     \
      hipLaunchKernelGGL((k_gather_rows<V, IdxT, true>), dim3((unsigned)grid), dim3(kGatherThreads), 0, st, s,  \
                         src_rows, idx, n, row_bytes, chunks, lpr_log2, d, src_stride, err);                    \
    else                                                                                                        \
      hipLaunchKernelGGL((k_gather_rows<V, IdxT, false>), dim3((unsigned)grid), dim3(kGatherThreads), 0, st, s, \
                         src_rows, idx, n, row_bytes, chunks, lpr_log2, d, src_stride, err);                    \
  } while (0)
  switch (gg.vec) {
    case kVecSpan: SPP_LAUNCH_GATHER(kVecSpan); break;
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

// dst[j,:] = the row at address addr[j] (row references, spp_mfg_out.row_addr)
template <int VEC>
__global__ __launch_bounds__(kGatherThreads) void k_gather_row_refs(const int64_t* __restrict__ addr, int64_t n,
                                                                     int64_t row_bytes, int chunks, int lpr_log2,
                                                                     char* __restrict__ dst) {
  move_rows_body<VEC, false>([=](int64_t r) { return addr[r]; },
                             [=](int64_t a) { return reinterpret_cast<const char*>((uintptr_t)a); }, n, row_bytes, chunks,
                             lpr_log2, dst, blockIdx.x, gridDim.x);
}

// used by sampler.hip (int32 node list of a slot)
spp_status gather_rows_i32(const void* src, int64_t src_rows, int64_t row_bytes, int64_t src_stride,
                           const int32_t* idx, int64_t n, void* dst, hipStream_t st) {
  return launch_gather<int32_t>(src, src_rows, row_bytes, src_stride, idx, n, dst, st);
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
  return spp_gather_rows_strided(src_dev, src_rows, row_bytes, row_bytes, idx_dev, idx_elem_bytes, n_idx, n_out,
                                 dst_dev, stream);
}

extern "C" spp_status spp_gather_rows_strided(const void* src_dev, int64_t src_rows, int64_t row_bytes,
                                              int64_t src_stride_bytes, const void* idx_dev, int idx_elem_bytes,
                                              int64_t n_idx, int64_t n_out, void* dst_dev, void* stream) {
  SPP_REQUIRE(row_bytes >= 0 && n_idx >= 0 && n_out >= 0, "spp_gather_rows: negative size");
  SPP_REQUIRE(src_stride_bytes == 0 || src_stride_bytes >= row_bytes,
              "spp_gather_rows: source stride %lld smaller than the row (%lld bytes)", (long long)src_stride_bytes,
              (long long)row_bytes);
  SPP_REQUIRE(idx_elem_bytes == 8 || idx_elem_bytes == 4, "spp_gather_rows: idx_elem_bytes must be 4 or 8, got %d",
              idx_elem_bytes);
  const int64_t n = n_idx < n_out ? n_idx : n_out;  // reference :253  min(idx.numel(), n)
  if (n == 0 || row_bytes == 0) return SPP_OK;
  SPP_REQUIRE(src_dev && idx_dev && dst_dev, "spp_gather_rows: NULL buffer");
  SPP_REQUIRE(src_rows > 0, "spp_gather_rows: %lld rows requested from an empty table (src_rows %lld)", (long long)n,
              (long long)src_rows);
  if (idx_elem_bytes == 8)
    return spp::launch_gather<int64_t>(src_dev, src_rows, row_bytes, src_stride_bytes,
                                       static_cast<const int64_t*>(idx_dev), n, dst_dev, spp::as_stream(stream));
  return spp::launch_gather<int32_t>(src_dev, src_rows, row_bytes, src_stride_bytes,
                                     static_cast<const int32_t*>(idx_dev), n, dst_dev, spp::as_stream(stream));
}

extern "C" spp_status spp_gather_row_refs(const int64_t* row_addr_dev, int64_t n, int64_t row_bytes, void* dst_dev,
                                          void* stream) {
  SPP_REQUIRE(n >= 0 && row_bytes >= 0, "spp_gather_row_refs: negative size");
  if (n == 0 || row_bytes == 0) return SPP_OK;
  SPP_REQUIRE(row_addr_dev && dst_dev, "spp_gather_row_refs: NULL buffer");
  // every referenced row is aligned to the largest power of two (<= 16) that divides row_bytes: rows of a table whose
  // stride is a multiple of it, dense rows of x_remote
  const spp::GatherGeom gg = spp::gather_geometry(nullptr, dst_dev, row_bytes, n, 0, /*allow_span=*/false);
  hipStream_t st = spp::as_stream(stream);
  char* d = static_cast<char*>(dst_dev);
#define SPP_LAUNCH_REFS(V)                                                                                       \
  hipLaunchKernelGGL(spp::k_gather_row_refs<V>, dim3((unsigned)gg.grid), dim3(spp::kGatherThreads), 0, st, row_addr_dev, n, \
                     row_bytes, gg.chunks, gg.lpr_log2, d)
  switch (gg.vec) {
    case 16: SPP_LAUNCH_REFS(16); break;
    case 8: SPP_LAUNCH_REFS(8); break;
    case 4: SPP_LAUNCH_REFS(4); break;
    case 2: SPP_LAUNCH_REFS(2); break;
    default: SPP_LAUNCH_REFS(1); break;
  }
#undef SPP_LAUNCH_REFS
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
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
