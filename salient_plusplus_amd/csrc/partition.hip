// a8-a11 / K3-K5: ownership bucketing of a batch's MFG node list, cache lookups and the fused
// final feature assembly.
//   RangePartitionBook   reference fast_sampler/range_partition_book.cpp:85-112
//   Cache                reference fast_sampler/range_partition_book.cpp:116-195
//   bucketing + perm     reference fast_sampler/fast_sampler.cpp:1031-1107 (no cache), :1108-1260 (cache)
//   assembly             reference fast_trainer/transferers.py:472-486
#include "spp_internal.h"

#include "partition_common.hip.h"

namespace spp {

constexpr int kPT = 256;
constexpr int kMaxBuckets = kPartBuckets;

__global__ __launch_bounds__(kPT) void k_nid2partid(Offsets o, const int64_t* __restrict__ nids, int64_t n,
                                                     int64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x;
  if (i < n) out[i] = owner_of(o, nids[i]);
}

// fast_cached_vertices_map[cached_vertices[i]] = i, later duplicates overwrite earlier ones
// (range_partition_book.cpp:154-158) -> atomicMax over the index.
__global__ __launch_bounds__(kPT) void k_cache_build(const int64_t* __restrict__ cv, int64_t n, int32_t* map,
                                                      int64_t len) {
  const int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x;
  if (i < n) {
    const int64_t v = cv[i];
    if (v >= 0 && v < len) atomicMax(&map[v], (int32_t)i);
  }
}

__global__ __launch_bounds__(kPT) void k_cache_lookup(const int32_t* __restrict__ map, int64_t len,
                                                       const int64_t* __restrict__ nids, int64_t n,
                                                       uint8_t* is_cached, int64_t* cache_nid) {
  const int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x;
  if (i >= n) return;
  const int64_t v = nids[i];
  const int32_t m = (v >= 0 && v < len) ? map[v] : -1;
  if (is_cached) is_cached[i] = m >= 0 ? 1 : 0;  // nid_is_cached (:161-183)
  if (cache_nid) cache_nid[i] = m;               // nid2cachenid (:185-195)
}

// ---- bucketing ---------------------------------------------------------------------------------
struct PartArgs {
  const int64_t* n_id;
  int64_t U;
  Offsets off;
  int32_t P, rank, use_cache;
  const int32_t* cache_map;
  int64_t cache_len;
  int64_t x_gpu_rows;
  uint8_t* bucket;     // [U]  bucket of every node; bit 7 = "local row living in host memory"
  int32_t* blk;        // [(P+2)][nblk] per-workgroup counts, bucket-major (scanned in place)
  int32_t nblk;
  int64_t* parts_out;
  int64_t* cached_out;
  int64_t* perm_out;
  int64_t* counts_out; // [P+2]
  int64_t* cpu_local_out;
};

__device__ __forceinline__ int32_t bucket_of(const PartArgs& a, int64_t v, bool& host_local) {
  host_local = false;
  const bool local = v >= a.off.v[a.rank] && v < a.off.v[a.rank + 1];  // nid_is_local (:105-107)
  if (local) host_local = (v - a.off.v[a.rank]) >= a.x_gpu_rows;         // fast_sampler.cpp:1046-1049
  if (!a.use_cache) return owner_of(a.off, v);                          // :1063
  if (local) return a.rank;                                             // :1216
  const int32_t m = (v >= 0 && v < a.cache_len) ? a.cache_map[v] : -1;
  if (m >= 0) return a.P;                                               // cache hits go last (:1243)
  return owner_of(a.off, v);                                            // :1202
}

__global__ __launch_bounds__(kPT) void k_part_hist(PartArgs a) {
  __shared__ int32_t cnt[kMaxBuckets];
  for (int k = threadIdx.x; k < a.P + 2; k += kPT) cnt[k] = 0;
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x;
  if (i < a.U) {
    bool hl;
    const int32_t b = bucket_of(a, a.n_id[i], hl);
    a.bucket[i] = (uint8_t)(b | (hl ? 0x80 : 0));
    atomicAdd(&cnt[b], 1);
    if (hl) atomicAdd(&cnt[a.P + 1], 1);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < a.P + 2; k += kPT) a.blk[(int64_t)k * a.nblk + blockIdx.x] = cnt[k];
}

__global__ __launch_bounds__(1024) void k_part_scan(PartArgs a) {
  __shared__ int32_t lds[1024 / kWave + 1];
  for (int32_t m = 0; m < a.P + 2; ++m) {
    int32_t* row = a.blk + (int64_t)m * a.nblk;
    int32_t carry = 0;
    for (int32_t base = 0; base < a.nblk; base += 1024) {
      const int32_t i = base + threadIdx.x;
      const int32_t v = (i < a.nblk) ? row[i] : 0;
      int32_t tot;
      const int32_t ex = block_exclusive_scan<int32_t, 1024>(v, lds, &tot);
      if (i < a.nblk) row[i] = carry + ex;
      carry += tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) a.counts_out[m] = carry;
  }
}

__global__ __launch_bounds__(kPT) void k_part_scatter(PartArgs a) {
  __shared__ int32_t wcnt[kPT / kWave][kMaxBuckets];
  __shared__ int64_t base[kMaxBuckets];
  const int wid = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  for (int k = threadIdx.x; k < (kPT / kWave) * kMaxBuckets; k += kPT) (&wcnt[0][0])[k] = 0;
  if (threadIdx.x == 0) {
    int64_t acc = 0;
    for (int m = 0; m <= a.P; ++m) {  // concat order: parts[0..P-1] then cache hits
      base[m] = acc;
      acc += a.counts_out[m];
    }
  }
  __syncthreads();
  const int64_t i = (int64_t)blockIdx.x * kPT + threadIdx.x;
  const bool valid = i < a.U;
  int32_t b = -1;
  bool hl = false;
  if (valid) {
    const uint8_t raw = a.bucket[i];
    b = raw & 0x7f;
    hl = (raw & 0x80) != 0;
  }
  // stable rank inside the wavefront among lanes of the same bucket
  int32_t rank_w = 0;
  unsigned long long todo = __ballot(valid);
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int32_t lb = __shfl(b, leader, kWave);
    const unsigned long long m = __ballot(valid && b == lb);
    if (valid && b == lb) {
      rank_w = __popcll(m & below);
      if (lane == leader) wcnt[wid][lb] = __popcll(m);
    }
    todo &= ~m;
  }
  const unsigned long long hm = __ballot(hl);
  const int32_t hrank_w = __popcll(hm & below);
  if (lane == 0) wcnt[wid][a.P + 1] = __popcll(hm);
  __syncthreads();
  if (!valid) return;
  int32_t pre = 0, hpre = 0;
  for (int w = 0; w < wid; ++w) {
    pre += wcnt[w][b];
    hpre += wcnt[w][a.P + 1];
  }
  const int64_t v = a.n_id[i];
  const int64_t pos = base[b] + a.blk[(int64_t)b * a.nblk + blockIdx.x] + pre + rank_w;
  a.perm_out[i] = pos;  // perm_partition_to_mfg (:1085 / :1246-1252)
  if (b < a.P) a.parts_out[pos] = v;
  else a.cached_out[pos - base[a.P]] = a.cache_map[v];  // nid2cachenid (:1256)
  if (hl && a.cpu_local_out) {
    const int64_t hp = a.blk[(int64_t)(a.P + 1) * a.nblk + blockIdx.x] + hpre + hrank_w;
    a.cpu_local_out[hp] = (v - a.off.v[a.rank]) - a.x_gpu_rows;  // :1048
  }
}

// ---- fused assembly ----------------------------------------------------------------------------
struct AsmArgs {
  const int64_t* n_id;
  const int64_t* perm;
  int64_t U;
  int32_t P, rank;
  int64_t seg_start[SPP_MAX_PARTS + 2];
  int64_t recv_base[SPP_MAX_PARTS + 1];
  int64_t rank_offset;
  const char* x_local;
  const char* recv;
  const char* cache_feats;
  const int64_t* cached_nids;
  int64_t row_bytes;
  int64_t x_local_stride, cache_stride;
  char* out;
  int64_t x_local_rows;
  int32_t* err;   // async error word (SPP_AERR_ASSEMBLE)
};

template <int VEC>
__global__ __launch_bounds__(kPT) void k_assemble(AsmArgs a, int chunks, int lpr_log2) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  using V = typename std::conditional<VEC == 16, u32x4,
            typename std::conditional<VEC == 8, u32x2,
            typename std::conditional<VEC == 4, uint32_t,
            typename std::conditional<VEC == 2, uint16_t, uint8_t>::type>::type>::type>::type;
  const int lpr = 1 << lpr_log2;
  const int g = threadIdx.x >> lpr_log2, l = threadIdx.x & (lpr - 1);
  const int gpb = kPT >> lpr_log2;
  for (int64_t r = (int64_t)blockIdx.x * gpb + g; r < a.U; r += (int64_t)gridDim.x * gpb) {
    const int64_t j = a.perm[r];
    int m = 0;
    while (m < a.P && j >= a.seg_start[m + 1]) ++m;
    const char* src;
    if (m == a.rank) {
      int64_t lr = a.n_id[r] - a.rank_offset;
      if ((uint64_t)lr >= (uint64_t)a.x_local_rows) {  // perm / n_id / partition book disagree
        raise_async_error(a.err, SPP_AERR_ASSEMBLE);
        lr = 0;
      }
      src = a.x_local + lr * a.x_local_stride;
    }
    else if (m == a.P) src = a.cache_feats + a.cached_nids[j - a.seg_start[a.P]] * a.cache_stride;
    else src = a.recv + (a.recv_base[m] + (j - a.seg_start[m])) * a.row_bytes;
    const V* s = reinterpret_cast<const V*>(src);
    V* d = reinterpret_cast<V*>(a.out + r * a.row_bytes);
    for (int c = l; c < chunks; c += lpr) d[c] = s[c];
  }
}

}  // namespace spp

using namespace spp;

static spp_status make_offsets(const int64_t* offsets_host, int32_t n_offsets, Offsets* o) {
  SPP_REQUIRE(offsets_host && n_offsets >= 2 && n_offsets <= SPP_MAX_PARTS + 1,
              "partition offsets: need 2..%d entries, got %d", SPP_MAX_PARTS + 1, n_offsets);
  o->n = n_offsets;
  for (int i = 0; i < n_offsets; ++i) o->v[i] = offsets_host[i];
  return SPP_OK;
}

extern "C" spp_status spp_nid2partid(const int64_t* offsets_host, int32_t n_offsets, const int64_t* nids_dev, int64_t n,
                                     int64_t* out_dev, void* stream) {
  Offsets o;
  SPP_TRY(make_offsets(offsets_host, n_offsets, &o));
  if (n <= 0) return SPP_OK;
  SPP_REQUIRE(nids_dev && out_dev, "spp_nid2partid: NULL buffer");
  hipLaunchKernelGGL(k_nid2partid, dim3((unsigned)ceil_div(n, kPT)), dim3(kPT), 0, as_stream(stream), o, nids_dev, n,
                     out_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_cache_build_map(const int64_t* cached_vertices_dev, int64_t n_cached, int32_t* cache_map_dev,
                                          int64_t cache_map_len, void* stream) {
  SPP_REQUIRE(cache_map_dev && cache_map_len > 0, "spp_cache_build_map: bad map");
  SPP_HIP_TRY(hipMemsetAsync(cache_map_dev, 0xFF, sizeof(int32_t) * (size_t)cache_map_len, as_stream(stream)));
  if (n_cached > 0) {
    SPP_REQUIRE(cached_vertices_dev, "spp_cache_build_map: NULL cached_vertices");
    hipLaunchKernelGGL(k_cache_build, dim3((unsigned)ceil_div(n_cached, kPT)), dim3(kPT), 0, as_stream(stream),
                       cached_vertices_dev, n_cached, cache_map_dev, cache_map_len);
    SPP_HIP_TRY(hipGetLastError());
  }
  return SPP_OK;
}

extern "C" spp_status spp_cache_lookup(const int32_t* cache_map_dev, int64_t cache_map_len, const int64_t* nids_dev,
                                       int64_t n, uint8_t* is_cached_dev, int64_t* cache_nid_dev, void* stream) {
  if (n <= 0) return SPP_OK;
  SPP_REQUIRE(cache_map_dev && nids_dev, "spp_cache_lookup: NULL buffer");
  hipLaunchKernelGGL(k_cache_lookup, dim3((unsigned)ceil_div(n, kPT)), dim3(kPT), 0, as_stream(stream), cache_map_dev,
                     cache_map_len, nids_dev, n, is_cached_dev, cache_nid_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" int64_t spp_partition_workspace_bytes(int64_t max_nodes) {
  if (max_nodes < 1) max_nodes = 1;
  const int64_t nblk = ceil_div(max_nodes, kPT);
  const int64_t bucket_bytes = (max_nodes + 255) / 256 * 256;
  return bucket_bytes + (int64_t)sizeof(int32_t) * nblk * kMaxBuckets;
}

extern "C" spp_status spp_partition_batch(const int64_t* n_id_dev, int64_t U, const int64_t* offsets_host, int32_t P,
                                          int32_t rank, int32_t use_cache, const int32_t* cache_map_dev,
                                          int64_t cache_map_len, int64_t x_gpu_rows, int64_t* parts_out_dev,
                                          int64_t* cached_out_dev, int64_t* perm_out_dev, int64_t* counts_out_dev,
                                          int64_t* cpu_local_out_dev, void* workspace_dev, int64_t workspace_bytes,
                                          void* stream) {
  PartArgs a{};
  SPP_TRY(make_offsets(offsets_host, P + 1, &a.off));
  SPP_REQUIRE(rank >= 0 && rank < P, "spp_partition_batch: rank %d out of [0,%d)", rank, P);
  SPP_REQUIRE(U >= 0, "spp_partition_batch: negative U");
  SPP_REQUIRE(counts_out_dev, "spp_partition_batch: counts_out_dev is NULL");
  SPP_REQUIRE(!use_cache || (cache_map_dev && cache_map_len > 0), "spp_partition_batch: use_cache without a cache map");
  hipStream_t st = as_stream(stream);
  if (U == 0) {
    SPP_HIP_TRY(hipMemsetAsync(counts_out_dev, 0, sizeof(int64_t) * (size_t)(P + 2), st));
    return SPP_OK;
  }
  SPP_REQUIRE(n_id_dev && parts_out_dev && cached_out_dev && perm_out_dev && workspace_dev,
              "spp_partition_batch: NULL buffer");
  SPP_REQUIRE(workspace_bytes >= spp_partition_workspace_bytes(U),
              "spp_partition_batch: workspace too small (%lld < %lld)", (long long)workspace_bytes,
              (long long)spp_partition_workspace_bytes(U));
  a.n_id = n_id_dev;
  a.U = U;
  a.P = P;
  a.rank = rank;
  a.use_cache = use_cache;
  a.cache_map = cache_map_dev;
  a.cache_len = cache_map_len;
  a.x_gpu_rows = x_gpu_rows;
  a.nblk = (int32_t)ceil_div(U, kPT);
  a.bucket = static_cast<uint8_t*>(workspace_dev);
  a.blk = reinterpret_cast<int32_t*>(static_cast<char*>(workspace_dev) + (U + 255) / 256 * 256);
  a.parts_out = parts_out_dev;
  a.cached_out = cached_out_dev;
  a.perm_out = perm_out_dev;
  a.counts_out = counts_out_dev;
  a.cpu_local_out = cpu_local_out_dev;
  hipLaunchKernelGGL(k_part_hist, dim3((unsigned)a.nblk), dim3(kPT), 0, st, a);
  hipLaunchKernelGGL(k_part_scan, dim3(1), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(k_part_scatter, dim3((unsigned)a.nblk), dim3(kPT), 0, st, a);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_assemble_features(const int64_t* n_id_dev, const int64_t* perm_dev, int64_t U,
                                            const int64_t* seg_start_host, int32_t P, int32_t rank, int64_t rank_offset,
                                            const void* x_local_dev, int64_t x_local_rows, const void* recv_dev,
                                            const void* cache_feats_dev, const int64_t* cached_nids_dev,
                                            int64_t row_bytes, int64_t x_local_stride_bytes,
                                            int64_t cache_stride_bytes, const int64_t* recv_base_host,
                                            void* x_out_dev, void* stream) {
  SPP_REQUIRE(P >= 1 && P <= SPP_MAX_PARTS && rank >= 0 && rank < P, "spp_assemble_features: bad P/rank");
  SPP_REQUIRE(seg_start_host, "spp_assemble_features: seg_start_host is NULL");
  if (U <= 0 || row_bytes <= 0) return SPP_OK;
  SPP_REQUIRE(n_id_dev && perm_dev && x_out_dev, "spp_assemble_features: NULL buffer");
  AsmArgs a{};
  a.n_id = n_id_dev;
  a.perm = perm_dev;
  a.U = U;
  a.P = P;
  a.rank = rank;
  int64_t rb = 0;
  for (int m = 0; m <= P + 1; ++m) a.seg_start[m] = seg_start_host[m];
  for (int m = 0; m < P; ++m) {
    a.recv_base[m] = recv_base_host ? recv_base_host[m] : rb;  // explicit: rows of a whole GROUP in one buffer
    if (m != rank) rb += seg_start_host[m + 1] - seg_start_host[m];
  }
  SPP_REQUIRE(seg_start_host[P + 1] == U, "spp_assemble_features: segments (%lld) do not cover U (%lld)",
              (long long)seg_start_host[P + 1], (long long)U);
  SPP_REQUIRE(x_local_dev || seg_start_host[rank + 1] == seg_start_host[rank], "spp_assemble_features: x_local is NULL");
  SPP_REQUIRE(recv_dev || rb == 0, "spp_assemble_features: recv is NULL");
  SPP_REQUIRE((cache_feats_dev && cached_nids_dev) || seg_start_host[P + 1] == seg_start_host[P],
              "spp_assemble_features: cache rows requested without a cache");
  SPP_REQUIRE(x_local_rows > 0 || seg_start_host[rank + 1] == seg_start_host[rank],
              "spp_assemble_features: local rows requested from an empty x_local");
  a.x_local_rows = x_local_rows;
  a.err = async_err_word_current();
  a.rank_offset = rank_offset;
  a.x_local = static_cast<const char*>(x_local_dev);
  a.recv = static_cast<const char*>(recv_dev);
  a.cache_feats = static_cast<const char*>(cache_feats_dev);
  a.cached_nids = cached_nids_dev;
  a.row_bytes = row_bytes;
  a.x_local_stride = x_local_stride_bytes > 0 ? x_local_stride_bytes : row_bytes;
  a.cache_stride = cache_stride_bytes > 0 ? cache_stride_bytes : row_bytes;
  SPP_REQUIRE(a.x_local_stride >= row_bytes && a.cache_stride >= row_bytes,
              "spp_assemble_features: row strides must be at least row_bytes");
  a.out = static_cast<char*>(x_out_dev);
  const uintptr_t al = reinterpret_cast<uintptr_t>(x_local_dev) | reinterpret_cast<uintptr_t>(recv_dev) |
                       reinterpret_cast<uintptr_t>(cache_feats_dev) | reinterpret_cast<uintptr_t>(x_out_dev) |
                       (uintptr_t)row_bytes | (uintptr_t)a.x_local_stride | (uintptr_t)a.cache_stride;
  int vec = 16;
  while (vec > 1 && (al % vec) != 0) vec >>= 1;
  const int chunks = (int)(row_bytes / vec);
  int lpr_log2 = 0;
  while ((1 << lpr_log2) < chunks && lpr_log2 < 6) ++lpr_log2;
  const int gpb = kPT >> lpr_log2;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(U, gpb), 256 * 32);
  hipStream_t st = as_stream(stream);
  const int prof = prof_begin(SPP_PROF_ASSEMBLE, st, U);
  switch (vec) {
    case 16: hipLaunchKernelGGL(k_assemble<16>, dim3(grid), dim3(kPT), 0, st, a, chunks, lpr_log2); break;
    case 8: hipLaunchKernelGGL(k_assemble<8>, dim3(grid), dim3(kPT), 0, st, a, chunks, lpr_log2); break;
    case 4: hipLaunchKernelGGL(k_assemble<4>, dim3(grid), dim3(kPT), 0, st, a, chunks, lpr_log2); break;
    case 2: hipLaunchKernelGGL(k_assemble<2>, dim3(grid), dim3(kPT), 0, st, a, chunks, lpr_log2); break;
    default: hipLaunchKernelGGL(k_assemble<1>, dim3(grid), dim3(kPT), 0, st, a, chunks, lpr_log2); break;
  }
  prof_end(SPP_PROF_ASSEMBLE, prof, st);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
