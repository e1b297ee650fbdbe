// f1: VIP analytic model on the GPU -- the probability that a vertex is touched by one mini-batch,
// propagated hop by hop over the CSR (reference: driver/drivers/ddp.py:135-239
// get_frequency_tensors_fast, the Taylor form the reference ships with):
//     p_0[v]     = batch_size / |train|            for v in this rank's training ids, else 0
//     q_h[u]     = min(1, fanout_h / deg(u)) * p_h[u]
//     p_{h+1}[v] = 1 - exp(-sum_{u in N(v)} q_h[u])          (segment_csr(..., 'add') over row v)
//     total[v]   = 1 - prod_h (1 - p_{h+1}[v])
// float64 throughout, like the reference.  HBM bound: per hop one sequential pass over `col` (8 B
// per edge) plus one random 8-B read of q per edge; the reference streams rowptr/col chunks from
// host memory and calls torch_scatter per chunk, here the topology is already resident.
#include "spp_internal.h"

namespace spp {

constexpr int kVipNT = 256;
constexpr int kVipLanes = 16;  // lanes cooperating on one row (average degree of the target graphs is 10-60)

__global__ __launch_bounds__(kVipNT) void k_vip_seed(const int64_t* __restrict__ train, int64_t n, double v,
                                                     int64_t num_nodes, double* __restrict__ p0) {
  const int64_t i = (int64_t)blockIdx.x * kVipNT + threadIdx.x;
  if (i >= n) return;
  const int64_t t = train[i];
  if (t >= 0 && t < num_nodes) p0[t] = v;
}

__global__ __launch_bounds__(kVipNT) void k_vip_weight(const int64_t* __restrict__ rowptr, int64_t num_nodes,
                                                       double fanout, const double* __restrict__ p,
                                                       double* __restrict__ q, double* __restrict__ prod,
                                                       int init_prod) {
  const int64_t u = (int64_t)blockIdx.x * kVipNT + threadIdx.x;
  if (u >= num_nodes) return;
  const double deg = (double)(rowptr[u + 1] - rowptr[u]);
  const double w = fmin(1.0, fanout / deg);  // deg == 0: fanout/0 = inf -> 1, as torch.minimum gives
  q[u] = w * p[u];
  if (init_prod) prod[u] = 1.0;
}

__global__ __launch_bounds__(kVipNT) void k_vip_hop(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ col,
                                                    int64_t num_nodes, const double* __restrict__ q,
                                                    double* __restrict__ p_next, double* __restrict__ prod) {
  const int lane = threadIdx.x & (kVipLanes - 1);
  const int64_t v = ((int64_t)blockIdx.x * kVipNT + threadIdx.x) / kVipLanes;
  double s = 0.0;
  if (v < num_nodes) {
    const int64_t b = rowptr[v], e = rowptr[v + 1];
    for (int64_t k = b + lane; k < e; k += kVipLanes) s += q[col[k]];
  }
#pragma unroll
  for (int d = kVipLanes / 2; d >= 1; d >>= 1) s += __shfl_xor(s, d, kVipLanes);
  if (v < num_nodes && lane == 0) {
    const double pn = 1.0 - exp(-s);
    p_next[v] = pn;
    prod[v] *= (1.0 - pn);
  }
}

__global__ __launch_bounds__(kVipNT) void k_vip_total(int64_t num_nodes, const double* __restrict__ prod,
                                                      double* __restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * kVipNT + threadIdx.x;
  if (v < num_nodes) out[v] = 1.0 - prod[v];
}

}  // namespace spp

using namespace spp;

extern "C" spp_status spp_vip_frequencies(const int64_t* rowptr_dev, const int64_t* col_dev, int64_t num_nodes,
                                          const int64_t* train_idx_dev, int64_t n_train, int64_t batch_size,
                                          const int64_t* fanouts_host, int32_t num_hops, double* out_dev,
                                          double* workspace_dev, void* stream) {
  SPP_REQUIRE(rowptr_dev && out_dev && workspace_dev && fanouts_host, "spp_vip_frequencies: NULL argument");
  SPP_REQUIRE(num_nodes > 0 && num_hops >= 1 && num_hops <= SPP_MAX_HOPS, "spp_vip_frequencies: bad sizes");
  SPP_REQUIRE(n_train > 0 && train_idx_dev, "spp_vip_frequencies: no training ids");
  SPP_REQUIRE(batch_size > 0, "spp_vip_frequencies: batch_size must be positive");
  hipStream_t st = as_stream(stream);
  double* p_a = workspace_dev;                // p_h
  double* p_b = workspace_dev + num_nodes;    // p_{h+1}
  double* q = workspace_dev + 2 * num_nodes;  // weighted p_h
  double* prod = out_dev;                     // running prod (1 - p_h), turned into the result at the end
  const unsigned gn = (unsigned)ceil_div(num_nodes, kVipNT);
  SPP_HIP_TRY(hipMemsetAsync(p_a, 0, sizeof(double) * (size_t)num_nodes, st));
  hipLaunchKernelGGL(k_vip_seed, dim3((unsigned)ceil_div(n_train, kVipNT)), dim3(kVipNT), 0, st, train_idx_dev,
                     n_train, (double)batch_size * 1.0 / (double)n_train, num_nodes, p_a);
  for (int h = 0; h < num_hops; ++h) {
    hipLaunchKernelGGL(k_vip_weight, dim3(gn), dim3(kVipNT), 0, st, rowptr_dev, num_nodes, (double)fanouts_host[h],
                       p_a, q, prod, h == 0 ? 1 : 0);
    hipLaunchKernelGGL(k_vip_hop, dim3((unsigned)ceil_div(num_nodes * kVipLanes, kVipNT)), dim3(kVipNT), 0, st,
                       rowptr_dev, col_dev, num_nodes, q, p_b, prod);
    double* t = p_a;
    p_a = p_b;
    p_b = t;
  }
  hipLaunchKernelGGL(k_vip_total, dim3(gn), dim3(kVipNT), 0, st, num_nodes, prod, out_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
