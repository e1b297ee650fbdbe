// f3: mean aggregation over an MFG hop's CSR -- the message passing of SAGEConv(aggr='mean')
// (reference: driver/models.py:19-56 uses torch_geometric.nn.SAGEConv on (x, x_target), adj_t):
//     out[t,:]     = (1 / max(deg t, 1)) * sum_{e in row t} x[col[e],:]               forward
//     grad_x[s,:] += grad_out[t,:] / max(deg t, 1)   for every edge (t, s)            backward
// HBM/L2 bound gather-reduce: LPR lanes share a target row and stride over the feature dimension in
// 16-B pieces; the first layer reads the batch's fp16 features directly (fp16 -> fp32 is exact, so
// this equals converting the whole matrix first, which the reference model does, minus one pass
// over 150 MB).  The linear layers stay library GEMMs.
#include "spp_internal.h"

#include <hip/hip_fp16.h>

namespace spp {

constexpr int kAggNT = 256;

struct f4 {
  float x, y, z, w;
};

__device__ __forceinline__ f4 load4(const float* p) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  return {v.x, v.y, v.z, v.w};
}
__device__ __forceinline__ f4 load4(const __half* p) {
  const uint2 raw = *reinterpret_cast<const uint2*>(p);
  const __half2 a = *reinterpret_cast<const __half2*>(&raw.x), b = *reinterpret_cast<const __half2*>(&raw.y);
  const float2 fa = __half22float2(a), fb = __half22float2(b);
  return {fa.x, fa.y, fb.x, fb.y};
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const __half* p) { return __half2float(*p); }

// VEC4: F % 4 == 0 and rows 16-B (fp32) / 8-B (fp16) aligned
template <typename Tin, bool VEC4>
__global__ __launch_bounds__(kAggNT) void k_csr_mean_fwd(const int64_t* __restrict__ rowptr,
                                                         const int64_t* __restrict__ col, int64_t T,
                                                         const Tin* __restrict__ x, int64_t x_stride, int64_t F,
                                                         int lpr_log2, float* __restrict__ out, int64_t out_stride) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  const float inv = 1.0f / (float)(e > b ? e - b : 1);
  if (VEC4) {
    for (int64_t c = (int64_t)lane * 4; c < F; c += (int64_t)lpr * 4) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      int64_t k = b;
      for (; k + 1 < e; k += 2) {  // two independent rows in flight
        const f4 v0 = load4(x + col[k] * x_stride + c), v1 = load4(x + col[k + 1] * x_stride + c);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      }
      if (k < e) {
        const f4 v0 = load4(x + col[k] * x_stride + c);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      }
      *reinterpret_cast<float4*>(out + t * out_stride + c) =
          make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
  } else {
    for (int64_t c = lane; c < F; c += lpr) {
      float acc = 0.f;
      for (int64_t k = b; k < e; ++k) acc += load1(x + col[k] * x_stride + c);
      out[t * out_stride + c] = acc * inv;
    }
  }
}

__global__ __launch_bounds__(kAggNT) void k_csr_mean_bwd(const int64_t* __restrict__ rowptr,
                                                         const int64_t* __restrict__ col, int64_t T,
                                                         const float* __restrict__ grad_out, int64_t go_stride,
                                                         int64_t F, int lpr_log2, float* __restrict__ grad_x) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  if (e <= b) return;
  const float inv = 1.0f / (float)(e - b);
  for (int64_t c = lane; c < F; c += lpr) {
    const float g = grad_out[t * go_stride + c] * inv;
    for (int64_t k = b; k < e; ++k) unsafeAtomicAdd(grad_x + col[k] * F + c, g);  // hardware fp32 atomic add
  }
}

static int lanes_log2(int64_t pieces) {
  int l = 0;
  while ((1 << l) < pieces && l < 6) ++l;
  return l;
}

}  // namespace spp

using namespace spp;

extern "C" spp_status spp_csr_mean_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                           const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                                           float* out_dev, int64_t out_stride_elems, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0, "spp_csr_mean_forward: negative size");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && out_dev, "spp_csr_mean_forward: NULL buffer");
  SPP_REQUIRE(x_stride_elems >= F, "spp_csr_mean_forward: row stride smaller than the row");
  if (out_stride_elems <= 0) out_stride_elems = F;
  SPP_REQUIRE(out_stride_elems >= F, "spp_csr_mean_forward: output stride smaller than the row");
  hipStream_t st = as_stream(stream);
  const int64_t esz = x_is_half ? 2 : 4;
  const bool vec = (F % 4 == 0) && ((x_stride_elems * esz) % (4 * esz) == 0) &&
                   (reinterpret_cast<uintptr_t>(x_dev) % (4 * esz) == 0) &&
                   (reinterpret_cast<uintptr_t>(out_dev) % 16 == 0) && (out_stride_elems % 4 == 0);
  const int lpr_log2 = lanes_log2(vec ? F / 4 : F);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
#define SPP_AGG(TIN, V)                                                                                          \
  hipLaunchKernelGGL((k_csr_mean_fwd<TIN, V>), dim3(grid), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, \
                     static_cast<const TIN*>(x_dev), x_stride_elems, F, lpr_log2, out_dev, out_stride_elems)
  if (x_is_half) {
    if (vec) SPP_AGG(__half, true); else SPP_AGG(__half, false);
  } else {
    if (vec) SPP_AGG(float, true); else SPP_AGG(float, false);
  }
#undef SPP_AGG
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_csr_mean_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                            const float* grad_out_dev, int64_t grad_out_stride_elems, int64_t F,
                                            float* grad_x_dev, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0, "spp_csr_mean_backward: negative size");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && grad_out_dev && grad_x_dev, "spp_csr_mean_backward: NULL buffer");
  if (grad_out_stride_elems <= 0) grad_out_stride_elems = F;
  SPP_REQUIRE(grad_out_stride_elems >= F, "spp_csr_mean_backward: gradient stride smaller than the row");
  const int lpr_log2 = lanes_log2(F);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
  hipLaunchKernelGGL(k_csr_mean_bwd, dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev, num_targets,
                     grad_out_dev, grad_out_stride_elems, F, lpr_log2, grad_x_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// ================================================================================================
// GATConv(heads=1) message passing over an MFG hop (reference: driver/models.py:195-231 uses
// torch_geometric.nn.GATConv(bias=False, heads=1) on ((x, x_target), adj_t)):
//     e_ij  = leaky_relu(a_src[j] + a_dst[i], slope)     for j in row i without its diagonal entry, plus j = i
//             (GATConv adds self loops with set_diag: exactly one (i, i) entry per target)
//     out_i = sum_j softmax_j(e_ij) * h[j,:]
// One pass over the neighbour rows with a running maximum (the rescaling trick of online softmax).
// ================================================================================================
namespace spp {

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

__global__ __launch_bounds__(kAggNT) void k_gat_fwd(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                    int64_t T, const float* __restrict__ h, int64_t F,
                                                    const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                    float slope, int lpr_log2, float* __restrict__ out,
                                                    float* __restrict__ row_max, float* __restrict__ row_sum) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  const float ad = a_dst[t];
  // every lane of the group walks the same edges, so the softmax statistics need no exchange
  float m = lrelu(a_src[t] + ad, slope);  // the self loop
  float ssum = 1.f;
  for (int64_t c0 = lane; c0 < F || c0 == lane; c0 += lpr) {  // at least one sweep even if F < lpr
    const bool has = c0 < F;
    float acc = has ? h[t * F + c0] : 0.f;
    float mm = lrelu(a_src[t] + ad, slope), ss = 1.f;
    for (int64_t k = b; k < e; ++k) {
      const int64_t j = col[k];
      if (j == t) continue;  // set_diag drops existing diagonal entries
      const float sc = lrelu(a_src[j] + ad, slope);
      if (sc > mm) {
        const float r = __expf(mm - sc);
        acc *= r;
        ss *= r;
        mm = sc;
      }
      const float w = __expf(sc - mm);
      ss += w;
      if (has) acc += w * h[j * F + c0];
    }
    if (has) out[t * F + c0] = acc / ss;
    m = mm;
    ssum = ss;
  }
  if (lane == 0) {
    row_max[t] = m;
    row_sum[t] = ssum;
  }
}

// grad_h[j,:] += a_ij * g_i;  grad_e_ij = a_ij * (g_i . h_j - g_i . out_i) * lrelu'(raw);  grad_a_src[j] += grad_e_ij;
// grad_a_dst[i] = sum_j grad_e_ij.  One wavefront-sized group per target reduces the dot products.
__global__ __launch_bounds__(kAggNT) void k_gat_bwd(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                    int64_t T, const float* __restrict__ h, int64_t F,
                                                    const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                    float slope, const float* __restrict__ out,
                                                    const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                    const float* __restrict__ g, float* __restrict__ grad_h,
                                                    float* __restrict__ grad_a_src, float* __restrict__ grad_a_dst) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) / kWave;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  const float ad = a_dst[t], m = row_max[t], inv_s = 1.f / row_sum[t];
  float go = 0.f;  // g_i . out_i
  for (int64_t c = lane; c < F; c += kWave) go += g[t * F + c] * out[t * F + c];
#pragma unroll
  for (int d = kWave / 2; d >= 1; d >>= 1) go += __shfl_xor(go, d, kWave);
  float gad = 0.f;
  for (int64_t k = b - 1; k < e; ++k) {  // k == b-1 stands for the self loop
    const int64_t j = (k < b) ? t : col[k];
    if (k >= b && j == t) continue;
    const float raw = a_src[j] + ad;
    const float a = __expf(lrelu(raw, slope) - m) * inv_s;
    float gh = 0.f;  // g_i . h_j
    for (int64_t c = lane; c < F; c += kWave) {
      const float gv = g[t * F + c];
      gh += gv * h[j * F + c];
      unsafeAtomicAdd(grad_h + j * F + c, a * gv);
    }
#pragma unroll
    for (int d = kWave / 2; d >= 1; d >>= 1) gh += __shfl_xor(gh, d, kWave);
    const float ge = a * (gh - go) * (raw > 0.f ? 1.f : slope);
    gad += ge;
    if (lane == 0) unsafeAtomicAdd(grad_a_src + j, ge);
  }
  if (lane == 0) grad_a_dst[t] = gad;
}

}  // namespace spp

extern "C" spp_status spp_gat_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                      const float* h_dev, int64_t F, const float* a_src_dev, const float* a_dst_dev,
                                      float negative_slope, float* out_dev, float* row_max_dev, float* row_sum_dev,
                                      void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0, "spp_gat_forward: negative size");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && h_dev && a_src_dev && a_dst_dev && out_dev && row_max_dev && row_sum_dev,
              "spp_gat_forward: NULL buffer");
  const int lpr_log2 = lanes_log2(F);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
  hipLaunchKernelGGL(k_gat_fwd, dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev, num_targets, h_dev,
                     F, a_src_dev, a_dst_dev, negative_slope, lpr_log2, out_dev, row_max_dev, row_sum_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_gat_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                       const float* h_dev, int64_t F, const float* a_src_dev, const float* a_dst_dev,
                                       float negative_slope, const float* out_dev, const float* row_max_dev,
                                       const float* row_sum_dev, const float* grad_out_dev, float* grad_h_dev,
                                       float* grad_a_src_dev, float* grad_a_dst_dev, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0, "spp_gat_backward: negative size");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && h_dev && a_src_dev && a_dst_dev && out_dev && row_max_dev && row_sum_dev && grad_out_dev &&
                  grad_h_dev && grad_a_src_dev && grad_a_dst_dev,
              "spp_gat_backward: NULL buffer");
  const unsigned grid = (unsigned)ceil_div(num_targets * kWave, kAggNT);
  hipLaunchKernelGGL(k_gat_bwd, dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev, num_targets, h_dev,
                     F, a_src_dev, a_dst_dev, negative_slope, out_dev, row_max_dev, row_sum_dev, grad_out_dev,
                     grad_h_dev, grad_a_src_dev, grad_a_dst_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
