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
#include <hipcub/hipcub.hpp>

#include <algorithm>

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

// ---- ReLU + dropout (driver/models.py:47-48: x = F.relu(x); x = F.dropout(x, p=0.5)) ----
// keep / drop from a counter-based generator: element i of the call with `seed` is kept iff
// hash(seed, i) < (1 - p) * 2^32; the output is relu(x) / (1 - p) where kept, 0 elsewhere.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
struct ActArgs {
  uint32_t keep_thr;
  float scale;
  uint64_t seed;
  int32_t training;
};
// the activation of the four elements 4*i4 .. 4*i4+3 of a dense array, exactly as k_relu_dropout_fwd computes it
__device__ __forceinline__ f4 relu_dropout4(f4 v, int64_t i4, const ActArgs& a) {
  if (!a.training) return {fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f)};
  const uint64_t r0 = mix64(a.seed + 2ull * (uint64_t)i4 * 0x9E3779B97F4A7C15ull);
  const uint64_t r1 = mix64(a.seed + (2ull * (uint64_t)i4 + 1ull) * 0x9E3779B97F4A7C15ull);
  f4 o;
  o.x = (v.x > 0.f && (uint32_t)r0 < a.keep_thr) ? v.x * a.scale : 0.f;
  o.y = (v.y > 0.f && (uint32_t)(r0 >> 32) < a.keep_thr) ? v.y * a.scale : 0.f;
  o.z = (v.z > 0.f && (uint32_t)r1 < a.keep_thr) ? v.z * a.scale : 0.f;
  o.w = (v.w > 0.f && (uint32_t)(r1 >> 32) < a.keep_thr) ? v.w * a.scale : 0.f;
  return o;
}

// VEC4: F % 4 == 0 and rows 16-B (fp32) / 8-B (fp16) aligned
// kAct (fp32, VEC4, dense rows: x_stride == F): x is a PRE-activation; relu + dropout are applied to every row
// as it is loaded -- the same values a k_relu_dropout_fwd pass over x would have produced (same generator,
// same element indices), without the pass.
// kTable (first layer, opt-in): x is the RESIDENT feature table and row j of the batch is x[nid[j]] (nid = the batch's
// n_id, int64) -- the batch's feature matrix is never written and re-read (242 MB each way at papers scale); the sum
// runs over the same rows in the same order as over a materialised x[n_id], so the operand is bit-identical.  An id
// outside [0, x_rows) reads row 0 (no fault).
// kRefs (first layer, opt-in, partitioned path): row j of the batch is the F elements at ADDRESS nid[j] (row references,
// spp_mfg_out.row_addr: local partition / VIP cache / received rows / a peer's partition); x and x_stride are unused.
template <typename Tin, bool VEC4, bool kAct = false, bool kTable = false, bool kRefs = false>
__global__ __launch_bounds__(kAggNT) void k_csr_mean_fwd(const int64_t* __restrict__ rowptr,
                                                         const int64_t* __restrict__ col, int64_t T,
                                                         const Tin* __restrict__ x, int64_t x_stride, int64_t F,
                                                         int lpr_log2, float* __restrict__ out, int64_t out_stride,
                                                         int concat_target, ActArgs act,
                                                         const int64_t* __restrict__ nid = nullptr, int64_t x_rows = 0) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  const float inv = 1.0f / (float)(e > b ? e - b : 1);
  auto row4 = [&](int64_t j, int64_t c) {  // four columns of row j (activated on load with kAct)
    if constexpr (kRefs) return load4(reinterpret_cast<const Tin*>((uintptr_t)nid[j]) + c);
    if constexpr (kTable) {
      const int64_t g = nid[j];
      j = (uint64_t)g < (uint64_t)x_rows ? g : 0;
    }
    f4 v = load4(x + j * x_stride + c);
    if constexpr (kAct) v = relu_dropout4(v, (j * x_stride + c) >> 2, act);
    return v;
  };
  if (VEC4) {
    for (int64_t c = (int64_t)lane * 4; c < F; c += (int64_t)lpr * 4) {
      if (concat_target) {  // [mean | x_target]: the target's own row (targets are the first rows of x), as fp32
        const f4 own = row4(t, c);
        *reinterpret_cast<float4*>(out + t * out_stride + F + c) = make_float4(own.x, own.y, own.z, own.w);
      }
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      int64_t k = b;
      for (; k + 1 < e; k += 2) {  // two independent rows in flight
        const f4 v0 = row4(col[k], c), v1 = row4(col[k + 1], c);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
      }
      if (k < e) {
        const f4 v0 = row4(col[k], c);
        acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
      }
      *reinterpret_cast<float4*>(out + t * out_stride + c) =
          make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
  } else {
    auto row_of = [&](int64_t j) {
      if constexpr (kTable) {
        const int64_t g = nid[j];
        j = (uint64_t)g < (uint64_t)x_rows ? g : 0;
      }
      return j;
    };
    for (int64_t c = lane; c < F; c += lpr) {
      if (concat_target) out[t * out_stride + F + c] = load1(x + row_of(t) * x_stride + c);
      float acc = 0.f;
      for (int64_t k = b; k < e; ++k) acc += load1(x + row_of(col[k]) * x_stride + c);
      out[t * out_stride + c] = acc * inv;
      static_assert(!kRefs || VEC4, "row references use the vector form");
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

// grad_x of the fused SAGE operand before the scatter of the mean's gradient: the first T source rows
// are the targets themselves and start from the gradient of the x_target half, the rest from zero
// (replaces a zero fill, the zero-padded gradient of the x[:T] slice and the add of the two).
__global__ __launch_bounds__(kAggNT) void k_grad_init(const float* __restrict__ grad_out, int64_t go_stride, int64_t T,
                                                      int64_t S, int64_t F, float* __restrict__ grad_x) {
  const int64_t n4 = S * F / 4;  // F % 4 == 0 (checked by the caller)
  for (int64_t i = (int64_t)blockIdx.x * kAggNT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kAggNT) {
    const int64_t srow = (i * 4) / F, c = (i * 4) - srow * F;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (srow < T) v = *reinterpret_cast<const float4*>(grad_out + srow * go_stride + F + c);
    reinterpret_cast<float4*>(grad_x)[i] = v;
  }
}

// ---- ReLU + dropout in one pass (the stand-alone form; the SAGE stack applies them on load, see kAct) ----
// The backward pass needs no mask: y > 0 exactly where x > 0 and the element was kept.
__global__ __launch_bounds__(kAggNT) void k_relu_dropout_fwd(const float* __restrict__ x, int64_t n, uint32_t keep_thr,
                                                             float scale, uint64_t seed, int training,
                                                             float* __restrict__ y) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kAggNT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kAggNT) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    if (training) {
      const uint64_t r0 = mix64(seed + 2ull * (uint64_t)i * 0x9E3779B97F4A7C15ull);
      const uint64_t r1 = mix64(seed + (2ull * (uint64_t)i + 1ull) * 0x9E3779B97F4A7C15ull);
      o.x = (v.x > 0.f && (uint32_t)r0 < keep_thr) ? v.x * scale : 0.f;
      o.y = (v.y > 0.f && (uint32_t)(r0 >> 32) < keep_thr) ? v.y * scale : 0.f;
      o.z = (v.z > 0.f && (uint32_t)r1 < keep_thr) ? v.z * scale : 0.f;
      o.w = (v.w > 0.f && (uint32_t)(r1 >> 32) < keep_thr) ? v.w * scale : 0.f;
    } else {
      o = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    }
    reinterpret_cast<float4*>(y)[i] = o;
  }
  // tail (n % 4 elements) by the first threads of block 0
  const int64_t t = n4 * 4 + threadIdx.x;
  if (blockIdx.x == 0 && t < n) {
    const float v = x[t];
    const uint64_t r = mix64(seed + (uint64_t)t * 0xD1B54A32D192ED03ull);
    y[t] = training ? ((v > 0.f && (uint32_t)r < keep_thr) ? v * scale : 0.f) : fmaxf(v, 0.f);
  }
}

__global__ __launch_bounds__(kAggNT) void k_relu_dropout_bwd(const float* __restrict__ g, const float* __restrict__ y,
                                                             int64_t n, float scale, float* __restrict__ gx) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kAggNT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kAggNT) {
    const float4 a = reinterpret_cast<const float4*>(g)[i], b = reinterpret_cast<const float4*>(y)[i];
    reinterpret_cast<float4*>(gx)[i] = make_float4(b.x > 0.f ? a.x * scale : 0.f, b.y > 0.f ? a.y * scale : 0.f,
                                                   b.z > 0.f ? a.z * scale : 0.f, b.w > 0.f ? a.w * scale : 0.f);
  }
  const int64_t t = n4 * 4 + threadIdx.x;
  if (blockIdx.x == 0 && t < n) gx[t] = y[t] > 0.f ? g[t] * scale : 0.f;
}

// the same backward from the PRE-activation z and the generator (no activated copy exists when the forward
// applied the activation on load): gx = g * scale where z > 0 and the element was kept
__global__ __launch_bounds__(kAggNT) void k_relu_dropout_bwd_pre(const float* __restrict__ g, const float* __restrict__ z,
                                                                 int64_t n, ActArgs act, float* __restrict__ gx) {
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * kAggNT + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kAggNT) {
    const float4 a = reinterpret_cast<const float4*>(g)[i], b = reinterpret_cast<const float4*>(z)[i];
    const f4 m = relu_dropout4(f4{b.x, b.y, b.z, b.w}, i, act);  // > 0 exactly where z > 0 and kept
    const float sc = act.training ? act.scale : 1.f;
    reinterpret_cast<float4*>(gx)[i] = make_float4(m.x > 0.f ? a.x * sc : 0.f, m.y > 0.f ? a.y * sc : 0.f,
                                                   m.z > 0.f ? a.z * sc : 0.f, m.w > 0.f ? a.w * sc : 0.f);
  }
}

// ---- backward of the fused operand by GATHER over the transposed hop ----------------------------
// 42 M fp32 atomics (163 k edges x 256 columns) run at the chip's atomic rate (~325 G/s: 140 us); the
// transposed adjacency (sources -> targets) costs three small integer passes per step and turns the
// backward into the same gather-reduce as the forward.
__global__ __launch_bounds__(kAggNT) void k_tr_count(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                     int64_t T, int32_t* __restrict__ cnt, float* __restrict__ inv) {
  const int64_t t = (int64_t)blockIdx.x * kAggNT + threadIdx.x;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  inv[t] = 1.0f / (float)(e > b ? e - b : 1);
  for (int64_t k = b; k < e; ++k) atomicAdd(&cnt[col[k]], 1);
}

__global__ __launch_bounds__(kAggNT) void k_tr_fill(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                    int64_t T, const int32_t* __restrict__ start,
                                                    int32_t* __restrict__ cursor, int32_t* __restrict__ tcol) {
  const int64_t t = (int64_t)blockIdx.x * kAggNT + threadIdx.x;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  for (int64_t k = b; k < e; ++k) {
    const int64_t s = col[k];
    tcol[start[s] + atomicAdd(&cursor[s], 1)] = (int32_t)t;
  }
}

// grad_x[s,:] = (s < T ? grad_out[s, F:2F] : 0) + sum over the targets t of s: grad_out[t, :F] / deg(t)
// kAct: grad_x is the gradient w.r.t. an ACTIVATED input whose pre-activation is z (dense [S, F]): the
// ReLU + dropout backward (k_relu_dropout_bwd_pre) is applied to the row before it is stored, instead of a
// separate read-modify-write pass over grad_x.
template <bool kAct>
__global__ __launch_bounds__(kAggNT) void k_operand_bwd_gather(const int32_t* __restrict__ start,
                                                               const int32_t* __restrict__ tcol,
                                                               const float* __restrict__ inv, int64_t T, int64_t S,
                                                               const float* __restrict__ g, int64_t go_stride, int64_t F,
                                                               int lpr_log2, float* __restrict__ grad_x,
                                                               const float* __restrict__ z, ActArgs act) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t srow = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (srow >= S) return;
  const int32_t b = start[srow], e = start[srow + 1];
  for (int64_t c = (int64_t)lane * 4; c < F; c += (int64_t)lpr * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (srow < T) acc = *reinterpret_cast<const float4*>(g + srow * go_stride + F + c);
    int32_t k = b;
    for (; k + 1 < e; k += 2) {  // two independent rows in flight
      const int32_t t0 = tcol[k], t1 = tcol[k + 1];
      const float w0 = inv[t0], w1 = inv[t1];
      const float4 v0 = *reinterpret_cast<const float4*>(g + (int64_t)t0 * go_stride + c);
      const float4 v1 = *reinterpret_cast<const float4*>(g + (int64_t)t1 * go_stride + c);
      acc.x += v0.x * w0 + v1.x * w1; acc.y += v0.y * w0 + v1.y * w1;
      acc.z += v0.z * w0 + v1.z * w1; acc.w += v0.w * w0 + v1.w * w1;
    }
    if (k < e) {
      const int32_t t0 = tcol[k];
      const float w0 = inv[t0];
      const float4 v0 = *reinterpret_cast<const float4*>(g + (int64_t)t0 * go_stride + c);
      acc.x += v0.x * w0; acc.y += v0.y * w0; acc.z += v0.z * w0; acc.w += v0.w * w0;
    }
    if constexpr (kAct) {
      const float4 zv = *reinterpret_cast<const float4*>(z + srow * F + c);
      const f4 m = relu_dropout4(f4{zv.x, zv.y, zv.z, zv.w}, (srow * F + c) >> 2, act);  // > 0: z > 0 and kept
      const float sc = act.training ? act.scale : 1.f;
      acc = make_float4(m.x > 0.f ? acc.x * sc : 0.f, m.y > 0.f ? acc.y * sc : 0.f, m.z > 0.f ? acc.z * sc : 0.f,
                        m.w > 0.f ? acc.w * sc : 0.f);
    }
    *reinterpret_cast<float4*>(grad_x + srow * F + c) = acc;
  }
}

static int lanes_log2(int64_t pieces) {
  int l = 0;
  while ((1 << l) < pieces && l < 6) ++l;
  return l;
}

}  // namespace spp

using namespace spp;

static spp_status mean_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                               const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                               float* out_dev, int64_t out_stride_elems, int concat_target, void* stream,
                               const int64_t* n_id_dev = nullptr, int64_t table_rows = 0);

extern "C" spp_status spp_csr_mean_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                           const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                                           float* out_dev, int64_t out_stride_elems, void* stream) {
  return mean_forward(rowptr_dev, col_dev, num_targets, x_dev, x_is_half, x_stride_elems, F, out_dev, out_stride_elems,
                      0, stream);
}

extern "C" spp_status spp_sage_operand_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                               const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                                               float* out_dev, int64_t out_stride_elems, void* stream) {
  SPP_REQUIRE(out_stride_elems >= 2 * F, "spp_sage_operand_forward: the operand [mean | x_target] needs 2F columns");
  return mean_forward(rowptr_dev, col_dev, num_targets, x_dev, x_is_half, x_stride_elems, F, out_dev, out_stride_elems,
                      1, stream);
}

extern "C" spp_status spp_sage_operand_forward_table(const int64_t* rowptr_dev, const int64_t* col_dev,
                                                     int64_t num_targets, const void* table_dev, int32_t table_is_half,
                                                     int64_t table_stride_elems, int64_t table_rows,
                                                     const int64_t* n_id_dev, int64_t F, float* out_dev,
                                                     int64_t out_stride_elems, void* stream) {
  SPP_REQUIRE(out_stride_elems >= 2 * F, "spp_sage_operand_forward_table: the operand [mean | x_target] needs 2F columns");
  SPP_REQUIRE(num_targets == 0 || (n_id_dev && table_dev && table_rows > 0),
              "spp_sage_operand_forward_table: needs the feature table and the batch's node ids");
  return mean_forward(rowptr_dev, col_dev, num_targets, table_dev, table_is_half, table_stride_elems, F, out_dev,
                      out_stride_elems, 1, stream, n_id_dev, table_rows);
}

static spp_status mean_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                               const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t F,
                               float* out_dev, int64_t out_stride_elems, int concat_target, void* stream,
                               const int64_t* n_id_dev, int64_t table_rows) {
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
#define SPP_AGG(TIN, V)                                                                                               \
  do {                                                                                                                \
    if (n_id_dev)                                                                                                     \
      hipLaunchKernelGGL((k_csr_mean_fwd<TIN, V, false, true>), dim3(grid), dim3(kAggNT), 0, st, rowptr_dev, col_dev, \
                         num_targets, static_cast<const TIN*>(x_dev), x_stride_elems, F, lpr_log2, out_dev,           \
                         out_stride_elems, concat_target, ActArgs{}, n_id_dev, table_rows);                           \
    else                                                                                                              \
      hipLaunchKernelGGL((k_csr_mean_fwd<TIN, V>), dim3(grid), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, \
                         static_cast<const TIN*>(x_dev), x_stride_elems, F, lpr_log2, out_dev, out_stride_elems,      \
                         concat_target, ActArgs{}, nullptr, (int64_t)0);                                              \
  } while (0)
  if (x_is_half) {
    if (vec) SPP_AGG(__half, true); else SPP_AGG(__half, false);
  } else {
    if (vec) SPP_AGG(float, true); else SPP_AGG(float, false);
  }
#undef SPP_AGG
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_sage_operand_forward_rows(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                    const int64_t* row_addr_dev, int32_t rows_are_half, int64_t F,
                                                    float* out_dev, int64_t out_stride_elems, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0, "spp_sage_operand_forward_rows: negative size");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && out_dev && row_addr_dev, "spp_sage_operand_forward_rows: NULL buffer");
  SPP_REQUIRE(out_stride_elems >= 2 * F, "spp_sage_operand_forward_rows: the operand [mean | x_target] needs 2F columns");
  SPP_REQUIRE(F % 4 == 0 && (reinterpret_cast<uintptr_t>(out_dev) % 16 == 0) && (out_stride_elems % 4 == 0),
              "spp_sage_operand_forward_rows: needs F %% 4 == 0 and a 16-byte aligned operand (F = %lld)", (long long)F);
  hipStream_t st = as_stream(stream);
  const int lpr_log2 = lanes_log2(F / 4);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
  if (rows_are_half)
    hipLaunchKernelGGL((k_csr_mean_fwd<__half, true, false, false, true>), dim3(grid), dim3(kAggNT), 0, st, rowptr_dev, col_dev,
                       num_targets, static_cast<const __half*>(nullptr), (int64_t)0, F, lpr_log2, out_dev, out_stride_elems, 1,
                       ActArgs{}, row_addr_dev, (int64_t)0);
  else
    hipLaunchKernelGGL((k_csr_mean_fwd<float, true, false, false, true>), dim3(grid), dim3(kAggNT), 0, st, rowptr_dev, col_dev,
                       num_targets, static_cast<const float*>(nullptr), (int64_t)0, F, lpr_log2, out_dev, out_stride_elems, 1,
                       ActArgs{}, row_addr_dev, (int64_t)0);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

static ActArgs act_args(float p, int32_t training, uint64_t seed) {
  const double keep = 1.0 - (double)p;
  ActArgs a{};
  a.keep_thr = keep >= 1.0 ? 0xffffffffu : (uint32_t)(keep * 4294967296.0);
  a.scale = (float)(1.0 / keep);
  a.seed = seed;
  a.training = training ? 1 : 0;
  return a;
}

// [mean | x_target] of relu_dropout(x) without materialising the activation (see kAct)
extern "C" spp_status spp_sage_operand_forward_act(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                   const float* x_dev, int64_t F, float* out_dev, int64_t out_stride_elems,
                                                   float p, int32_t training, uint64_t seed, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && F >= 0 && p >= 0.f && p < 1.f, "spp_sage_operand_forward_act: bad arguments");
  if (num_targets == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(rowptr_dev && x_dev && out_dev && out_stride_elems >= 2 * F && F % 4 == 0 && out_stride_elems % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(x_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(out_dev) % 16 == 0,
              "spp_sage_operand_forward_act: needs dense fp32 rows with F %% 4 == 0 and 16-byte aligned buffers");
  const int lpr_log2 = lanes_log2(F / 4);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
  hipLaunchKernelGGL((k_csr_mean_fwd<float, true, true>), dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev,
                     col_dev, num_targets, x_dev, F, F, lpr_log2, out_dev, out_stride_elems, 1, act_args(p, training, seed),
                     nullptr, (int64_t)0);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_relu_dropout_backward_pre(const float* grad_dev, const float* z_dev, int64_t n, float p,
                                                    int32_t training, uint64_t seed, float* grad_x_dev, void* stream) {
  SPP_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, "spp_relu_dropout_backward_pre: bad arguments");
  if (n == 0) return SPP_OK;
  SPP_REQUIRE(grad_dev && z_dev && grad_x_dev && n % 4 == 0 &&
                  (reinterpret_cast<uintptr_t>(grad_dev) | reinterpret_cast<uintptr_t>(z_dev) |
                   reinterpret_cast<uintptr_t>(grad_x_dev)) % 16 == 0,
              "spp_relu_dropout_backward_pre: needs n %% 4 == 0 and 16-byte aligned buffers");
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / 4, kAggNT), 256 * 32));
  hipLaunchKernelGGL(k_relu_dropout_bwd_pre, dim3(grid), dim3(kAggNT), 0, as_stream(stream), grad_dev, z_dev, n,
                     act_args(p, training, seed), grad_x_dev);
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

extern "C" spp_status spp_sage_operand_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                int64_t num_sources, const float* grad_out_dev,
                                                int64_t grad_out_stride_elems, int64_t F, float* grad_x_dev,
                                                void* stream) {
  SPP_REQUIRE(num_targets >= 0 && num_sources >= num_targets && F >= 0, "spp_sage_operand_backward: bad sizes");
  if (num_sources == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(grad_x_dev && (grad_out_dev || num_targets == 0), "spp_sage_operand_backward: NULL buffer");
  SPP_REQUIRE(F % 4 == 0 && grad_out_stride_elems >= 2 * F && grad_out_stride_elems % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(grad_out_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(grad_x_dev) % 16 == 0,
              "spp_sage_operand_backward: needs F %% 4 == 0 and 16-byte aligned rows");
  hipStream_t st = as_stream(stream);
  const int64_t n4 = num_sources * F / 4;
  const unsigned g0 = (unsigned)std::min<int64_t>(ceil_div(n4, kAggNT), 256 * 32);
  hipLaunchKernelGGL(k_grad_init, dim3(g0), dim3(kAggNT), 0, st, grad_out_dev, grad_out_stride_elems, num_targets,
                     num_sources, F, grad_x_dev);
  SPP_HIP_TRY(hipGetLastError());
  if (num_targets == 0) return SPP_OK;
  return spp_csr_mean_backward(rowptr_dev, col_dev, num_targets, grad_out_dev, grad_out_stride_elems, F, grad_x_dev,
                               stream);
}

extern "C" int64_t spp_sage_operand_backward_workspace_bytes(int64_t num_targets, int64_t num_sources,
                                                               int64_t num_edges) {
  size_t scan_tmp = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr,
                                         (int)(num_sources + 1));
  // cnt/cursor [S+1] | start [S+1] | tcol [E] | inv [T] | scan temporaries   (each 16-byte aligned)
  auto up = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  return up(4 * (num_sources + 1)) * 2 + up(4 * num_edges) + up(4 * num_targets) + up((int64_t)scan_tmp) + 64;
}

static spp_status operand_backward_gather(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                          int64_t num_sources, int64_t num_edges, const float* grad_out_dev,
                                          int64_t grad_out_stride_elems, int64_t F, float* grad_x_dev,
                                          void* workspace_dev, int64_t workspace_bytes, const float* z_pre_dev,
                                          ActArgs act, void* stream) {
  SPP_REQUIRE(num_targets >= 0 && num_sources >= num_targets && F >= 0 && num_edges >= 0,
              "spp_sage_operand_backward_gather: bad sizes");
  if (num_sources == 0 || F == 0) return SPP_OK;
  SPP_REQUIRE(num_sources < (1ll << 31) && num_edges < (1ll << 31), "spp_sage_operand_backward_gather: 32-bit indices");
  SPP_REQUIRE(grad_x_dev && workspace_dev && (grad_out_dev || num_targets == 0),
              "spp_sage_operand_backward_gather: NULL buffer");
  SPP_REQUIRE(F % 4 == 0 && grad_out_stride_elems >= 2 * F && grad_out_stride_elems % 4 == 0 &&
                  reinterpret_cast<uintptr_t>(grad_out_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(grad_x_dev) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(workspace_dev) % 16 == 0,
              "spp_sage_operand_backward_gather: needs F %% 4 == 0 and 16-byte aligned buffers");
  SPP_REQUIRE(workspace_bytes >= spp_sage_operand_backward_workspace_bytes(num_targets, num_sources, num_edges),
              "spp_sage_operand_backward_gather: workspace too small");
  hipStream_t st = as_stream(stream);
  auto up = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  char* w = static_cast<char*>(workspace_dev);
  int32_t* cnt = reinterpret_cast<int32_t*>(w);
  w += up(4 * (num_sources + 1));
  int32_t* start = reinterpret_cast<int32_t*>(w);
  w += up(4 * (num_sources + 1));
  int32_t* tcol = reinterpret_cast<int32_t*>(w);
  w += up(4 * num_edges);
  float* inv = reinterpret_cast<float*>(w);
  w += up(4 * num_targets);
  size_t scan_tmp = (size_t)(workspace_bytes - (w - static_cast<char*>(workspace_dev)));
  SPP_HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)(num_sources + 1), st));
  const unsigned gt = (unsigned)std::max<int64_t>(1, ceil_div(num_targets, kAggNT));
  if (num_targets > 0)
    hipLaunchKernelGGL(k_tr_count, dim3(gt), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, cnt, inv);
  SPP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w, scan_tmp, cnt, start, (int)(num_sources + 1), st));
  SPP_HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)(num_sources + 1), st));
  if (num_targets > 0)
    hipLaunchKernelGGL(k_tr_fill, dim3(gt), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, start, cnt, tcol);
  const int lpr_log2 = lanes_log2(F / 4);
  const unsigned grid = (unsigned)ceil_div(num_sources << lpr_log2, kAggNT);
  if (z_pre_dev)
    hipLaunchKernelGGL(k_operand_bwd_gather<true>, dim3(grid), dim3(kAggNT), 0, st, start, tcol, inv, num_targets,
                       num_sources, grad_out_dev, grad_out_stride_elems, F, lpr_log2, grad_x_dev, z_pre_dev, act);
  else
    hipLaunchKernelGGL(k_operand_bwd_gather<false>, dim3(grid), dim3(kAggNT), 0, st, start, tcol, inv, num_targets,
                       num_sources, grad_out_dev, grad_out_stride_elems, F, lpr_log2, grad_x_dev, nullptr, ActArgs{});
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_sage_operand_backward_gather(const int64_t* rowptr_dev, const int64_t* col_dev,
                                                       int64_t num_targets, int64_t num_sources, int64_t num_edges,
                                                       const float* grad_out_dev, int64_t grad_out_stride_elems,
                                                       int64_t F, float* grad_x_dev, void* workspace_dev,
                                                       int64_t workspace_bytes, void* stream) {
  return operand_backward_gather(rowptr_dev, col_dev, num_targets, num_sources, num_edges, grad_out_dev,
                                 grad_out_stride_elems, F, grad_x_dev, workspace_dev, workspace_bytes, nullptr, ActArgs{},
                                 stream);
}

// the same, followed in the same pass by the ReLU + dropout backward of spp_relu_dropout_backward_pre: grad_x
// becomes the gradient w.r.t. the PRE-activation z_pre_dev (dense fp32 [S, F]) of the rows the forward activated on load
extern "C" spp_status spp_sage_operand_backward_gather_act(const int64_t* rowptr_dev, const int64_t* col_dev,
                                                           int64_t num_targets, int64_t num_sources, int64_t num_edges,
                                                           const float* grad_out_dev, int64_t grad_out_stride_elems,
                                                           int64_t F, float* grad_x_dev, void* workspace_dev,
                                                           int64_t workspace_bytes, const float* z_pre_dev, float p,
                                                           int32_t training, uint64_t seed, void* stream) {
  SPP_REQUIRE(z_pre_dev && reinterpret_cast<uintptr_t>(z_pre_dev) % 16 == 0 && p >= 0.f && p < 1.f,
              "spp_sage_operand_backward_gather_act: NULL / unaligned pre-activation or bad p");
  return operand_backward_gather(rowptr_dev, col_dev, num_targets, num_sources, num_edges, grad_out_dev,
                                 grad_out_stride_elems, F, grad_x_dev, workspace_dev, workspace_bytes, z_pre_dev,
                                 act_args(p, training, seed), stream);
}

extern "C" spp_status spp_relu_dropout_forward(const float* x_dev, int64_t n, float p, int32_t training, uint64_t seed,
                                               float* y_dev, void* stream) {
  SPP_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, "spp_relu_dropout_forward: bad arguments");
  if (n == 0) return SPP_OK;
  SPP_REQUIRE(x_dev && y_dev && reinterpret_cast<uintptr_t>(x_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(y_dev) % 16 == 0,
              "spp_relu_dropout_forward: NULL or unaligned buffer");
  const double keep = 1.0 - (double)p;
  const uint32_t thr = keep >= 1.0 ? 0xffffffffu : (uint32_t)(keep * 4294967296.0);
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / 4, kAggNT), 256 * 32));
  hipLaunchKernelGGL(k_relu_dropout_fwd, dim3(grid), dim3(kAggNT), 0, as_stream(stream), x_dev, n, thr,
                     (float)(1.0 / keep), seed, training, y_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_relu_dropout_backward(const float* grad_dev, const float* y_dev, int64_t n, float scale,
                                                float* grad_x_dev, void* stream) {
  SPP_REQUIRE(n >= 0, "spp_relu_dropout_backward: negative size");
  if (n == 0) return SPP_OK;
  SPP_REQUIRE(grad_dev && y_dev && grad_x_dev, "spp_relu_dropout_backward: NULL buffer");
  SPP_REQUIRE((reinterpret_cast<uintptr_t>(grad_dev) | reinterpret_cast<uintptr_t>(y_dev) |
               reinterpret_cast<uintptr_t>(grad_x_dev)) % 16 == 0, "spp_relu_dropout_backward: unaligned buffer");
  const unsigned grid = (unsigned)std::max<int64_t>(1, std::min<int64_t>(ceil_div(n / 4, kAggNT), 256 * 32));
  hipLaunchKernelGGL(k_relu_dropout_bwd, dim3(grid), dim3(kAggNT), 0, as_stream(stream), grad_dev, y_dev, n, scale,
                     grad_x_dev);
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

// ================================================================================================
// GATConv, aggregate-then-project form.  GATConv computes h = W x for every SOURCE row and then
//   out_i = sum_j alpha_ij h_j,   alpha_ij = softmax_j(leaky_relu(att_src . h_j + att_dst . h_i)).
// Both uses of h are linear in x:  att_src . (W x_j) = x_j . (W^T att_src)  and  sum_j alpha_ij W x_j =
// W (sum_j alpha_ij x_j).  So the attention logits come from two K-vectors v_src = W^T att_src and
// v_dst = W^T att_dst, the aggregation runs over the RAW rows (128-wide fp16 in layer 1 instead of
// 256-wide fp32) and only the T aggregated target rows are projected: layer 1 of the papers-scale batch
// drops from a 947 k-row GEMM (62 GFLOP, and as much again for its weight gradient) to a 164 k-row one.
// ================================================================================================
__device__ __forceinline__ f4 load4v(const float* p) { return load4(p); }

// a_src[j] = x_j . v_src  (all S rows);  a_dst[j] = x_j . v_dst  (the first T rows: the targets)
// A group of lpr lanes takes kDotRows consecutive rows; their loads are issued back to back (row index clamped,
// not predicated: with one row per group and a predicate the kernel had ONE load in flight per lane and ran
// at a third of the HBM rate).
constexpr int kDotRows = 8;
template <typename Tin>
__global__ __launch_bounds__(kAggNT) void k_rowdot2(const Tin* __restrict__ x, int64_t x_stride, int64_t S, int64_t T,
                                                    int64_t K, const float* __restrict__ v_src,
                                                    const float* __restrict__ v_dst, int lpr_log2,
                                                    float* __restrict__ a_src, float* __restrict__ a_dst) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t j0 = (((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2) * kDotRows;
  if (j0 >= S) return;  // whole groups leave together (the shuffles below stay inside a group)
  float ds[kDotRows], dd[kDotRows];
#pragma unroll
  for (int u = 0; u < kDotRows; ++u) ds[u] = dd[u] = 0.f;
  for (int64_t c = (int64_t)lane * 4; c < K; c += (int64_t)lpr * 4) {
    f4 xv[kDotRows];
#pragma unroll
    for (int u = 0; u < kDotRows; ++u) {
      const int64_t j = j0 + u < S ? j0 + u : S - 1;
      xv[u] = load4(x + j * x_stride + c);
    }
    const f4 vs = load4v(v_src + c), vd = load4v(v_dst + c);
#pragma unroll
    for (int u = 0; u < kDotRows; ++u) {
      ds[u] += xv[u].x * vs.x + xv[u].y * vs.y + xv[u].z * vs.z + xv[u].w * vs.w;
      dd[u] += xv[u].x * vd.x + xv[u].y * vd.y + xv[u].z * vd.z + xv[u].w * vd.w;
    }
  }
#pragma unroll
  for (int u = 0; u < kDotRows; ++u) {
    for (int d = lpr >> 1; d >= 1; d >>= 1) {  // the lpr lanes of a row are consecutive lanes of one wavefront
      ds[u] += __shfl_xor(ds[u], d, kWave);
      dd[u] += __shfl_xor(dd[u], d, kWave);
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int u = 0; u < kDotRows; ++u) {
      const int64_t j = j0 + u;
      if (j < S) {
        a_src[j] = ds[u];
        if (j < T) a_dst[j] = dd[u];
      }
    }
  }
}

// out_src[c] += sum_j w_src[j] x[j,c] over all rows; out_dst[c] += sum_{j<T} w_dst[j] x[j,c]   (out zeroed by the caller)
// (1024 threads and 4096 rows per workgroup: every workgroup ends in K atomics onto the SAME K addresses, and with
// 256-thread / 1024-row workgroups those 924 x 256 same-address adds were most of the kernel's time)
constexpr int kColsumNT = 1024;
template <typename Tin>
__global__ __launch_bounds__(kColsumNT) void k_colsum2(const Tin* __restrict__ x, int64_t x_stride, int64_t S, int64_t T,
                                                       int64_t K, const float* __restrict__ w_src,
                                                       const float* __restrict__ w_dst, int64_t rows_per_wg,
                                                       float* __restrict__ out_src, float* __restrict__ out_dst) {
  __shared__ float red[2][kColsumNT][4];
  const int groups = (int)(K / 4);              // threads that share a row (K/4 <= kColsumNT)
  const int cg = threadIdx.x % groups;          // this thread's 4 columns
  const int rsub = threadIdx.x / groups, rstep = kColsumNT / groups;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_wg;
  const int64_t r1 = r0 + rows_per_wg < S ? r0 + rows_per_wg : S;
  f4 as = {0.f, 0.f, 0.f, 0.f}, ad = {0.f, 0.f, 0.f, 0.f};
  if (rsub < rstep) {
    int64_t j = r0 + rsub;
    const int64_t tlast = T > 0 ? T - 1 : 0;
    const float* __restrict__ wdp = T > 0 ? w_dst : w_src;  // (w_dst may be NULL without targets)
    for (; j + 3 * rstep < r1; j += 4 * rstep) {  // four independent rows in flight, no predicate on a load (eight: slower)
      f4 xv[4];
      float ws[4], wd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ju = j + u * rstep;
        xv[u] = load4(x + ju * x_stride + (int64_t)cg * 4);
        ws[u] = w_src[ju];
        wd[u] = wdp[ju < T ? ju : tlast];
        if (ju >= T) wd[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        as.x += ws[u] * xv[u].x; as.y += ws[u] * xv[u].y; as.z += ws[u] * xv[u].z; as.w += ws[u] * xv[u].w;
        ad.x += wd[u] * xv[u].x; ad.y += wd[u] * xv[u].y; ad.z += wd[u] * xv[u].z; ad.w += wd[u] * xv[u].w;
      }
    }
    for (; j < r1; j += rstep) {
      const f4 xv = load4(x + j * x_stride + (int64_t)cg * 4);
      const float ws = w_src[j];
      as.x += ws * xv.x; as.y += ws * xv.y; as.z += ws * xv.z; as.w += ws * xv.w;
      if (j < T) {
        const float wd = w_dst[j];
        ad.x += wd * xv.x; ad.y += wd * xv.y; ad.z += wd * xv.z; ad.w += wd * xv.w;
      }
    }
  }
  red[0][threadIdx.x][0] = as.x; red[0][threadIdx.x][1] = as.y; red[0][threadIdx.x][2] = as.z; red[0][threadIdx.x][3] = as.w;
  red[1][threadIdx.x][0] = ad.x; red[1][threadIdx.x][1] = ad.y; red[1][threadIdx.x][2] = ad.z; red[1][threadIdx.x][3] = ad.w;
  __syncthreads();
  if (threadIdx.x < groups) {  // one thread per column group sums the row sub-lanes, then one atomic per column
    float s4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int r = 0; r < rstep; ++r)
      for (int q = 0; q < 4; ++q) {
        s4[0][q] += red[0][r * groups + threadIdx.x][q];
        s4[1][q] += red[1][r * groups + threadIdx.x][q];
      }
    for (int q = 0; q < 4; ++q) {
      unsafeAtomicAdd(out_src + threadIdx.x * 4 + q, s4[0][q]);
      if (r0 < T) unsafeAtomicAdd(out_dst + threadIdx.x * 4 + q, s4[1][q]);
    }
  }
}

// z_i = sum_j alpha_ij x_j over row i (its diagonal entry dropped) plus the self loop; x rows fp16 or fp32,
// 4 columns per lane, online softmax (every lane of a row's group walks the same edges)
template <typename Tin>
__global__ __launch_bounds__(kAggNT) void k_gat_agg_fwd(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                        int64_t T, const Tin* __restrict__ x, int64_t x_stride, int64_t K,
                                                        const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                        float slope, int lpr_log2, float* __restrict__ z,
                                                        float* __restrict__ row_max, float* __restrict__ row_sum) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  const float ad = a_dst[t];
  const float self = lrelu(a_src[t] + ad, slope);
  for (int64_t c = (int64_t)lane * 4; c < K || c == (int64_t)lane * 4; c += (int64_t)lpr * 4) {
    const bool has = c < K;
    f4 acc = has ? load4(x + t * x_stride + c) : f4{0.f, 0.f, 0.f, 0.f};
    float mm = self, ss = 1.f;
    // kNb neighbours per round: their ids, then their logits and rows, each issued back to back (clamped
    // indices, no predicated load); the online softmax then runs over registers.  One neighbour at a time
    // was two dependent round trips per edge.
    constexpr int kNb = 4;
    const int64_t cc = has ? c : 0;
    for (int64_t k0 = b; k0 < e; k0 += kNb) {
      int64_t j[kNb];
#pragma unroll
      for (int u = 0; u < kNb; ++u) j[u] = col[k0 + u < e ? k0 + u : e - 1];
      float as[kNb];
      f4 xv[kNb];
#pragma unroll
      for (int u = 0; u < kNb; ++u) {
        as[u] = a_src[j[u]];
        xv[u] = load4(x + j[u] * x_stride + cc);
      }
#pragma unroll
      for (int u = 0; u < kNb; ++u) {
        if (k0 + u >= e || j[u] == t) continue;  // set_diag drops existing diagonal entries
        const float sc = lrelu(as[u] + ad, slope);
        if (sc > mm) {
          const float r = __expf(mm - sc);
          acc.x *= r; acc.y *= r; acc.z *= r; acc.w *= r;
          ss *= r;
          mm = sc;
        }
        const float w = __expf(sc - mm);
        ss += w;
        acc.x += w * xv[u].x; acc.y += w * xv[u].y; acc.z += w * xv[u].z; acc.w += w * xv[u].w;
      }
    }
    if (has) {
      const float inv = 1.f / ss;
      *reinterpret_cast<float4*>(z + t * K + c) = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
    }
    if (lane == 0 && c == 0) {
      row_max[t] = mm;
      row_sum[t] = ss;
    }
  }
}

// One wavefront per target i:  grad_e_ij = alpha_ij (g_i . x_j - g_i . z_i) lrelu'(raw);  grad_a_src[j] += grad_e_ij;
// grad_a_dst[i] = sum_j grad_e_ij;  grad_x[j,:] += alpha_ij g_i  (only when grad_x != NULL: the first layer's
// input needs no gradient, which saves E x K atomics)
// kVec: 4 columns per lane (no input gradient wanted: only dot products, 8/16-byte loads);
// otherwise one column per lane and step, so that the fp32 atomics of a group are contiguous
template <typename Tin, bool kVec>
__global__ __launch_bounds__(kAggNT) void k_gat_agg_bwd(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                        int64_t T, const Tin* __restrict__ x, int64_t x_stride, int64_t K,
                                                        const float* __restrict__ a_src, const float* __restrict__ a_dst,
                                                        float slope, const float* __restrict__ z,
                                                        const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                        const float* __restrict__ g, int lpt_log2, float* __restrict__ grad_x,
                                                        float* __restrict__ grad_a_src, float* __restrict__ grad_a_dst,
                                                        float* __restrict__ alpha_e, float* __restrict__ alpha_self) {
  // alpha_e / alpha_self (optional): the attention weight of every CSR entry (0 for a dropped diagonal entry) and of
  // every target's self loop -- what the input gradient by GATHER over the transposed hop needs (k_gat_gx_gather)
  // lpt = min(64, K rounded up to a power of two) lanes per target
  const int lpt = 1 << lpt_log2;
  const int lane = threadIdx.x & (lpt - 1);
  const int64_t t = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpt_log2;
  const bool live = t < T;
  const int64_t b = live ? rowptr[t] : 0, e = live ? rowptr[t + 1] : -1;
  const float ad = live ? a_dst[t] : 0.f, m = live ? row_max[t] : 0.f, inv_s = live ? 1.f / row_sum[t] : 0.f;
  auto group_sum = [&](float v) {
    for (int d = lpt >> 1; d >= 1; d >>= 1) v += __shfl_xor(v, d, kWave);
    return v;
  };
  // column c = lane + lpt * i: consecutive lanes touch consecutive columns, so a group's loads and --
  // what matters -- its fp32 atomics are contiguous 4 * lpt-byte segments (the atomic units run at full
  // rate on contiguous 256-byte wave instructions and ~17x slower on scattered ones)
  float go = 0.f;  // g_i . z_i
  if (live) {
    if (kVec) {
      for (int64_t c = (int64_t)lane * 4; c < K; c += (int64_t)lpt * 4) {
        const f4 gv = load4(g + t * K + c), zv = load4(z + t * K + c);
        go += gv.x * zv.x + gv.y * zv.y + gv.z * zv.z + gv.w * zv.w;
      }
    } else {
      for (int64_t c = lane; c < K; c += lpt) go += g[t * K + c] * z[t * K + c];
    }
  }
  go = group_sum(go);
  // the targets sharing a wavefront have different degrees: every group runs to the longest row of its
  // wavefront so that the shuffles stay convergent
  int64_t n = live ? e - b + 1 : 0;
  for (int d = lpt; d < kWave; d <<= 1) {
    const int64_t o = __shfl_xor((long long)n, d, kWave);
    n = o > n ? o : n;
  }
  float gad = 0.f;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t k = b - 1 + i;  // k == b-1 stands for the self loop
    const bool on = live && k < e;
    const int64_t j = !on ? 0 : ((k < b) ? t : col[k]);
    const bool use = on && !(k >= b && j == t);  // set_diag drops existing diagonal entries
    float gh = 0.f, a = 0.f, raw = 0.f;
    if (use) {
      raw = a_src[j] + ad;
      a = __expf(lrelu(raw, slope) - m) * inv_s;
      if (kVec) {
        for (int64_t c = (int64_t)lane * 4; c < K; c += (int64_t)lpt * 4) {
          const f4 gv = load4(g + t * K + c), xv = load4(x + j * x_stride + c);
          gh += gv.x * xv.x + gv.y * xv.y + gv.z * xv.z + gv.w * xv.w;
        }
      } else {
        for (int64_t c = lane; c < K; c += lpt) {
          const float gv = g[t * K + c];
          gh += gv * load1(x + j * x_stride + c);
          unsafeAtomicAdd(grad_x + j * K + c, a * gv);
        }
      }
    }
    gh = group_sum(gh);
    if (alpha_e && on && lane == 0) {
      if (k < b) alpha_self[t] = a;
      else alpha_e[k] = a;  // (a == 0 for a dropped diagonal entry)
    }
    if (use) {
      const float ge = a * (gh - go) * (raw > 0.f ? 1.f : slope);
      gad += ge;
      if (lane == 0) unsafeAtomicAdd(grad_a_src + j, ge);
    }
  }
  if (live && lane == 0) grad_a_dst[t] = gad;
}

// transposed hop with the CSR entry of every (source, target) pair kept: tedge[pos] = k, ttgt[pos] = t
__global__ __launch_bounds__(kAggNT) void k_tr_fill_e(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                      int64_t T, const int32_t* __restrict__ start,
                                                      int32_t* __restrict__ cursor, int32_t* __restrict__ ttgt,
                                                      int32_t* __restrict__ tedge) {
  const int64_t t = (int64_t)blockIdx.x * kAggNT + threadIdx.x;
  if (t >= T) return;
  const int64_t b = rowptr[t], e = rowptr[t + 1];
  for (int64_t k = b; k < e; ++k) {
    const int64_t s = col[k];
    const int32_t pos = start[s] + atomicAdd(&cursor[s], 1);
    ttgt[pos] = (int32_t)t;
    tedge[pos] = (int32_t)k;
  }
}

// Input gradient of the aggregate-then-project GAT layer by GATHER (no fp32 atomics, no zero fill, no separate rank-1
// passes):  grad_x[s,:] = sum over the targets t of s: alpha_ts grad_z[t,:]  (+ the self loop's alpha_ss grad_z[s,:], s < T)
//                         + grad_a_src[s] v_src  (+ grad_a_dst[s] v_dst, s < T)        -- a_src = x v_src, a_dst = x[:T] v_dst
__global__ __launch_bounds__(kAggNT) void k_gat_gx_gather(const int32_t* __restrict__ start, const int32_t* __restrict__ ttgt,
                                                          const int32_t* __restrict__ tedge, const float* __restrict__ alpha_e,
                                                          const float* __restrict__ alpha_self, int64_t T, int64_t S,
                                                          const float* __restrict__ g, int64_t K, int lpr_log2,
                                                          const float* __restrict__ grad_a_src,
                                                          const float* __restrict__ grad_a_dst,
                                                          const float* __restrict__ v_src, const float* __restrict__ v_dst,
                                                          float* __restrict__ grad_x) {
  const int lpr = 1 << lpr_log2;
  const int lane = threadIdx.x & (lpr - 1);
  const int64_t srow = ((int64_t)blockIdx.x * kAggNT + threadIdx.x) >> lpr_log2;
  if (srow >= S) return;
  const int32_t b = start[srow], e = start[srow + 1];
  const float gas = grad_a_src[srow];
  const float gad = srow < T ? grad_a_dst[srow] : 0.f;
  const float aself = srow < T ? alpha_self[srow] : 0.f;
  for (int64_t c = (int64_t)lane * 4; c < K; c += (int64_t)lpr * 4) {
    const float4 vs = *reinterpret_cast<const float4*>(v_src + c);
    float4 acc = make_float4(gas * vs.x, gas * vs.y, gas * vs.z, gas * vs.w);
    if (srow < T) {
      const float4 vd = *reinterpret_cast<const float4*>(v_dst + c);
      const float4 gs = *reinterpret_cast<const float4*>(g + srow * K + c);
      acc.x += gad * vd.x + aself * gs.x; acc.y += gad * vd.y + aself * gs.y;
      acc.z += gad * vd.z + aself * gs.z; acc.w += gad * vd.w + aself * gs.w;
    }
    int32_t k = b;
    for (; k + 1 < e; k += 2) {  // two independent rows in flight
      const int32_t t0 = ttgt[k], t1 = ttgt[k + 1];
      const float w0 = alpha_e[tedge[k]], w1 = alpha_e[tedge[k + 1]];
      const float4 v0 = *reinterpret_cast<const float4*>(g + (int64_t)t0 * K + c);
      const float4 v1 = *reinterpret_cast<const float4*>(g + (int64_t)t1 * K + c);
      acc.x += v0.x * w0 + v1.x * w1; acc.y += v0.y * w0 + v1.y * w1;
      acc.z += v0.z * w0 + v1.z * w1; acc.w += v0.w * w0 + v1.w * w1;
    }
    if (k < e) {
      const int32_t t0 = ttgt[k];
      const float w0 = alpha_e[tedge[k]];
      const float4 v0 = *reinterpret_cast<const float4*>(g + (int64_t)t0 * K + c);
      acc.x += v0.x * w0; acc.y += v0.y * w0; acc.z += v0.z * w0; acc.w += v0.w * w0;
    }
    *reinterpret_cast<float4*>(grad_x + srow * K + c) = acc;
  }
}

}  // namespace spp

static spp_status gat_check(int64_t K, const void* x, int64_t x_stride, int32_t is_half, const char* who) {
  const int64_t esz = is_half ? 2 : 4;
  SPP_REQUIRE(K > 0 && K % 4 == 0 && K / 4 <= spp::kAggNT && x_stride >= K && (x_stride * esz) % (4 * esz) == 0 &&
                  reinterpret_cast<uintptr_t>(x) % (4 * esz) == 0,
              "%s: needs K %% 4 == 0, K <= %d and rows aligned to 4 elements", who, 4 * spp::kAggNT);
  return SPP_OK;
}

extern "C" spp_status spp_gat_logits(const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t num_sources,
                                     int64_t num_targets, int64_t K, const float* v_src_dev, const float* v_dst_dev,
                                     float* a_src_dev, float* a_dst_dev, void* stream) {
  SPP_REQUIRE(num_sources >= num_targets && num_targets >= 0, "spp_gat_logits: bad sizes");
  if (num_sources == 0) return SPP_OK;
  SPP_TRY(gat_check(K, x_dev, x_stride_elems, x_is_half, "spp_gat_logits"));
  SPP_REQUIRE(v_src_dev && v_dst_dev && a_src_dev && (a_dst_dev || num_targets == 0) &&
                  reinterpret_cast<uintptr_t>(v_src_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(v_dst_dev) % 16 == 0,
              "spp_gat_logits: NULL or unaligned buffer");
  const int lpr_log2 = lanes_log2(K / 4);
  const unsigned grid = (unsigned)ceil_div(ceil_div(num_sources, kDotRows) << lpr_log2, kAggNT);
  if (x_is_half)
    hipLaunchKernelGGL(k_rowdot2<__half>, dim3(grid), dim3(kAggNT), 0, as_stream(stream), static_cast<const __half*>(x_dev),
                       x_stride_elems, num_sources, num_targets, K, v_src_dev, v_dst_dev, lpr_log2, a_src_dev, a_dst_dev);
  else
    hipLaunchKernelGGL(k_rowdot2<float>, dim3(grid), dim3(kAggNT), 0, as_stream(stream), static_cast<const float*>(x_dev),
                       x_stride_elems, num_sources, num_targets, K, v_src_dev, v_dst_dev, lpr_log2, a_src_dev, a_dst_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_gat_logits_backward(const void* x_dev, int32_t x_is_half, int64_t x_stride_elems,
                                              int64_t num_sources, int64_t num_targets, int64_t K,
                                              const float* grad_a_src_dev, const float* grad_a_dst_dev,
                                              float* grad_v_src_dev, float* grad_v_dst_dev, void* stream) {
  SPP_REQUIRE(num_sources >= num_targets && num_targets >= 0, "spp_gat_logits_backward: bad sizes");
  SPP_TRY(gat_check(K, x_dev, x_stride_elems, x_is_half, "spp_gat_logits_backward"));
  SPP_REQUIRE(grad_v_src_dev && grad_v_dst_dev, "spp_gat_logits_backward: NULL output");
  hipStream_t st = as_stream(stream);
  SPP_HIP_TRY(hipMemsetAsync(grad_v_src_dev, 0, sizeof(float) * (size_t)K, st));
  SPP_HIP_TRY(hipMemsetAsync(grad_v_dst_dev, 0, sizeof(float) * (size_t)K, st));
  if (num_sources == 0) return SPP_OK;
  SPP_REQUIRE(grad_a_src_dev && (grad_a_dst_dev || num_targets == 0), "spp_gat_logits_backward: NULL input");
  // 4 rows in flight per lane; K atomics onto the same K addresses per workgroup: about one workgroup per CU,
  // between 1024 and 4096 rows each
  const int64_t rows_per_wg = std::max<int64_t>(1024, std::min<int64_t>(4096, ceil_div(num_sources, 256)));
  const unsigned grid = (unsigned)ceil_div(num_sources, rows_per_wg);
  if (x_is_half)
    hipLaunchKernelGGL(k_colsum2<__half>, dim3(grid), dim3(kColsumNT), 0, st, static_cast<const __half*>(x_dev), x_stride_elems,
                       num_sources, num_targets, K, grad_a_src_dev, grad_a_dst_dev, rows_per_wg, grad_v_src_dev,
                       grad_v_dst_dev);
  else
    hipLaunchKernelGGL(k_colsum2<float>, dim3(grid), dim3(kColsumNT), 0, st, static_cast<const float*>(x_dev), x_stride_elems,
                       num_sources, num_targets, K, grad_a_src_dev, grad_a_dst_dev, rows_per_wg, grad_v_src_dev,
                       grad_v_dst_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_gat_aggregate_forward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                                const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                                float* z_dev, float* row_max_dev, float* row_sum_dev, void* stream) {
  SPP_REQUIRE(num_targets >= 0, "spp_gat_aggregate_forward: negative size");
  if (num_targets == 0) return SPP_OK;
  SPP_TRY(gat_check(K, x_dev, x_stride_elems, x_is_half, "spp_gat_aggregate_forward"));
  SPP_REQUIRE(rowptr_dev && a_src_dev && a_dst_dev && z_dev && row_max_dev && row_sum_dev &&
                  reinterpret_cast<uintptr_t>(z_dev) % 16 == 0, "spp_gat_aggregate_forward: NULL or unaligned buffer");
  const int lpr_log2 = lanes_log2(K / 4);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpr_log2, kAggNT);
  if (x_is_half)
    hipLaunchKernelGGL(k_gat_agg_fwd<__half>, dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev,
                       num_targets, static_cast<const __half*>(x_dev), x_stride_elems, K, a_src_dev, a_dst_dev,
                       negative_slope, lpr_log2, z_dev, row_max_dev, row_sum_dev);
  else
    hipLaunchKernelGGL(k_gat_agg_fwd<float>, dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev,
                       num_targets, static_cast<const float*>(x_dev), x_stride_elems, K, a_src_dev, a_dst_dev,
                       negative_slope, lpr_log2, z_dev, row_max_dev, row_sum_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

static spp_status gat_aggregate_backward_launch(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                                const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                                const float* z_dev, const float* row_max_dev, const float* row_sum_dev,
                                                const float* grad_z_dev, float* grad_x_dev, float* grad_a_src_dev,
                                                float* grad_a_dst_dev, float* alpha_e, float* alpha_self, void* stream);

extern "C" spp_status spp_gat_aggregate_backward(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                 const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                                 const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                                 const float* z_dev, const float* row_max_dev, const float* row_sum_dev,
                                                 const float* grad_z_dev, float* grad_x_dev /* NULL: not wanted */,
                                                 float* grad_a_src_dev, float* grad_a_dst_dev, void* stream) {
  SPP_REQUIRE(num_targets >= 0, "spp_gat_aggregate_backward: negative size");
  if (num_targets == 0) return SPP_OK;
  SPP_TRY(gat_check(K, x_dev, x_stride_elems, x_is_half, "spp_gat_aggregate_backward"));
  SPP_REQUIRE(rowptr_dev && a_src_dev && a_dst_dev && z_dev && row_max_dev && row_sum_dev && grad_z_dev &&
                  grad_a_src_dev && grad_a_dst_dev && reinterpret_cast<uintptr_t>(grad_z_dev) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(z_dev) % 16 == 0, "spp_gat_aggregate_backward: NULL or unaligned buffer");
  return gat_aggregate_backward_launch(rowptr_dev, col_dev, num_targets, x_dev, x_is_half, x_stride_elems, K, a_src_dev,
                                       a_dst_dev, negative_slope, z_dev, row_max_dev, row_sum_dev, grad_z_dev, grad_x_dev,
                                       grad_a_src_dev, grad_a_dst_dev, nullptr, nullptr, stream);
}

static spp_status gat_aggregate_backward_launch(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_targets,
                                                const void* x_dev, int32_t x_is_half, int64_t x_stride_elems, int64_t K,
                                                const float* a_src_dev, const float* a_dst_dev, float negative_slope,
                                                const float* z_dev, const float* row_max_dev, const float* row_sum_dev,
                                                const float* grad_z_dev, float* grad_x_dev, float* grad_a_src_dev,
                                                float* grad_a_dst_dev, float* alpha_e, float* alpha_self, void* stream) {
  const bool vec = grad_x_dev == nullptr;
  const int lpt_log2 = lanes_log2(vec ? K / 4 : K);
  const unsigned grid = (unsigned)ceil_div(num_targets << lpt_log2, kAggNT);
#define SPP_GAT_BWD(TIN, V)                                                                                            \
  hipLaunchKernelGGL((k_gat_agg_bwd<TIN, V>), dim3(grid), dim3(kAggNT), 0, as_stream(stream), rowptr_dev, col_dev,       \
                     num_targets, static_cast<const TIN*>(x_dev), x_stride_elems, K, a_src_dev, a_dst_dev,            \
                     negative_slope, z_dev, row_max_dev, row_sum_dev, grad_z_dev, lpt_log2, grad_x_dev, grad_a_src_dev, \
                     grad_a_dst_dev, alpha_e, alpha_self)
  if (x_is_half) {
    if (vec) SPP_GAT_BWD(__half, true); else SPP_GAT_BWD(__half, false);
  } else {
    if (vec) SPP_GAT_BWD(float, true); else SPP_GAT_BWD(float, false);
  }
#undef SPP_GAT_BWD
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int64_t spp_gat_aggregate_backward_gather_workspace_bytes(int64_t num_targets, int64_t num_sources,
                                                                     int64_t num_edges) {
  size_t scan_tmp = 0;
  (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp, (const int32_t*)nullptr, (int32_t*)nullptr,
                                         (int)(num_sources + 1));
  // cnt/cursor [S+1] | start [S+1] | ttgt [E] | tedge [E] | alpha_e [E] | alpha_self [T] | inv [T] (k_tr_count's by-product)
  // | scan temporaries   (each 16-byte aligned)
  auto up = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  return up(4 * (num_sources + 1)) * 2 + 3 * up(4 * num_edges) + 2 * up(4 * num_targets) + up((int64_t)scan_tmp) + 64;
}

// spp_gat_aggregate_backward with the input gradient by GATHER over the transposed hop: grad_x [S, K] is written
// completely, including the rank-1 terms of the logits (grad_a_src[s] v_src, grad_a_dst[s] v_dst for s < T) that
// the atomic form leaves to the caller.  grad_a_src_dev [S] is zeroed by the caller as before.
extern "C" spp_status spp_gat_aggregate_backward_gather(const int64_t* rowptr_dev, const int64_t* col_dev,
                                                        int64_t num_targets, int64_t num_sources, int64_t num_edges,
                                                        const void* x_dev, int32_t x_is_half, int64_t x_stride_elems,
                                                        int64_t K, const float* a_src_dev, const float* a_dst_dev,
                                                        float negative_slope, const float* z_dev,
                                                        const float* row_max_dev, const float* row_sum_dev,
                                                        const float* grad_z_dev, const float* v_src_dev,
                                                        const float* v_dst_dev, float* grad_x_dev, float* grad_a_src_dev,
                                                        float* grad_a_dst_dev, void* workspace_dev, int64_t workspace_bytes,
                                                        void* stream) {
  SPP_REQUIRE(num_targets >= 0 && num_sources >= num_targets && num_edges >= 0,
              "spp_gat_aggregate_backward_gather: bad sizes");
  if (num_sources == 0) return SPP_OK;
  SPP_TRY(gat_check(K, x_dev, x_stride_elems, x_is_half, "spp_gat_aggregate_backward_gather"));
  SPP_REQUIRE(num_sources < (1ll << 31) && num_edges < (1ll << 31), "spp_gat_aggregate_backward_gather: 32-bit indices");
  SPP_REQUIRE(rowptr_dev && a_src_dev && a_dst_dev && z_dev && row_max_dev && row_sum_dev && grad_z_dev && v_src_dev &&
                  v_dst_dev && grad_x_dev && grad_a_src_dev && grad_a_dst_dev && workspace_dev,
              "spp_gat_aggregate_backward_gather: NULL buffer");
  SPP_REQUIRE(reinterpret_cast<uintptr_t>(grad_z_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(z_dev) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(grad_x_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(v_src_dev) % 16 == 0 &&
                  reinterpret_cast<uintptr_t>(v_dst_dev) % 16 == 0 && reinterpret_cast<uintptr_t>(workspace_dev) % 16 == 0,
              "spp_gat_aggregate_backward_gather: buffers must be 16-byte aligned");
  SPP_REQUIRE(workspace_bytes >= spp_gat_aggregate_backward_gather_workspace_bytes(num_targets, num_sources, num_edges),
              "spp_gat_aggregate_backward_gather: workspace too small");
  hipStream_t st = as_stream(stream);
  auto up = [](int64_t v) { return (v + 15) & ~(int64_t)15; };
  char* w = static_cast<char*>(workspace_dev);
  int32_t* cnt = reinterpret_cast<int32_t*>(w);
  w += up(4 * (num_sources + 1));
  int32_t* start = reinterpret_cast<int32_t*>(w);
  w += up(4 * (num_sources + 1));
  int32_t* ttgt = reinterpret_cast<int32_t*>(w);
  w += up(4 * num_edges);
  int32_t* tedge = reinterpret_cast<int32_t*>(w);
  w += up(4 * num_edges);
  float* alpha_e = reinterpret_cast<float*>(w);
  w += up(4 * num_edges);
  float* alpha_self = reinterpret_cast<float*>(w);
  w += up(4 * num_targets);
  float* inv = reinterpret_cast<float*>(w);
  w += up(4 * num_targets);
  size_t scan_tmp = (size_t)(workspace_bytes - (w - static_cast<char*>(workspace_dev)));
  // the attention weights of every entry + grad_a_src / grad_a_dst (no input gradient by atomics)
  if (num_targets > 0)
    SPP_TRY(gat_aggregate_backward_launch(rowptr_dev, col_dev, num_targets, x_dev, x_is_half, x_stride_elems, K, a_src_dev,
                                          a_dst_dev, negative_slope, z_dev, row_max_dev, row_sum_dev, grad_z_dev, nullptr,
                                          grad_a_src_dev, grad_a_dst_dev, alpha_e, alpha_self, stream));
  // the transposed hop: count, scan, fill
  SPP_HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)(num_sources + 1), st));
  const unsigned gt = (unsigned)std::max<int64_t>(1, ceil_div(num_targets, kAggNT));
  if (num_targets > 0)
    hipLaunchKernelGGL(k_tr_count, dim3(gt), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, cnt, inv);
  SPP_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(w, scan_tmp, cnt, start, (int)(num_sources + 1), st));
  SPP_HIP_TRY(hipMemsetAsync(cnt, 0, 4 * (size_t)(num_sources + 1), st));
  if (num_targets > 0)
    hipLaunchKernelGGL(k_tr_fill_e, dim3(gt), dim3(kAggNT), 0, st, rowptr_dev, col_dev, num_targets, start, cnt, ttgt, tedge);
  const int lpr_log2 = lanes_log2(K / 4);
  const unsigned grid = (unsigned)ceil_div(num_sources << lpr_log2, kAggNT);
  hipLaunchKernelGGL(k_gat_gx_gather, dim3(grid), dim3(kAggNT), 0, st, start, ttgt, tedge, alpha_e, alpha_self, num_targets,
                     num_sources, grad_z_dev, K, lpr_log2, grad_a_src_dev, grad_a_dst_dev, v_src_dev, v_dst_dev, grad_x_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

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
