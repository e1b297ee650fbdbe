// Row-gather inner loop shared by k_gather_rows (gather.hip) and the fused k_deliver (sampler.hip).
// A group of LPR lanes (power of two) moves one row with VEC-byte accesses; consecutive groups take
// consecutive output rows so a wavefront's stores cover one contiguous span of dst, and every
// group keeps kGatherUnroll independent rows in flight.  `vblock`/`nvblocks` are the calling
// workgroup's index and the number of workgroups that share this gather (grid-stride).
#pragma once

#include <cstdint>
#include <cstdlib>

#include <hip/hip_runtime.h>

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
// stores of the gathered rows: non-temporal (default) or plain (-DSPP_GATHER_STORE_NT=0: measurement aid)
#ifndef SPP_GATHER_STORE_NT
#define SPP_GATHER_STORE_NT 1
#endif
template <typename T>
__device__ __forceinline__ void row_store(T v, T* p) {
#if SPP_GATHER_STORE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
#ifndef SPP_GATHER_UNROLL
#define SPP_GATHER_UNROLL 4
#endif
constexpr int kGatherUnroll = SPP_GATHER_UNROLL;

// workgroups per compute unit a row gather may put on the chip (spp_tune("gather_wg_per_cu"), SPP_GATHER_WG_PER_CU; api.hip)
int gather_wg_per_cu();

struct GatherGeom {
  int vec;        // bytes per lane access (16/8/4/2/1), or kVecSpan
  int chunks;     // row_bytes / vec (kVecSpan: 16-byte pieces of a row, the last one half used)
  int lpr_log2;   // lanes per row (log2)
  int64_t grid;   // workgroups wanted
};

// Rows of 16k + 8 bytes (200-byte rows: 100 fp16 features) out of a table whose row STRIDE is a multiple of 16:
// the row length alone used to force 8-byte accesses -- 25 of 32 lanes per row, two rows per wavefront instruction --
// although every source row starts 16-byte aligned and a wavefront's output rows form one contiguous span that starts
// 16-byte aligned whenever its first row is even.  kVecSpan: 16-byte loads (the last piece reads 8 bytes of the
// row's padding), the pieces regrouped across lanes (ds_bpermute) and the span stored as aligned 16-byte accesses.
constexpr int kVecSpan = 24;
bool gather_span_enabled();  // SPP_GATHER_SPAN=0 keeps the 8-byte form (api.hip)

static inline GatherGeom gather_geometry(const void* src, const void* dst, int64_t row_bytes, int64_t n,
                                         int64_t src_stride = 0, bool allow_span = false) {
  GatherGeom g{};
  const uintptr_t a = reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)row_bytes |
                      (uintptr_t)src_stride;
  g.vec = 16;
  while (g.vec > 1 && (a % g.vec) != 0) g.vec >>= 1;
  if (allow_span && g.vec == 8 && (row_bytes & 15) == 8 && row_bytes + 8 <= 32 * 16 &&
      ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)src_stride) & 15) == 0 &&
      src_stride >= row_bytes + 8 && gather_span_enabled())
    g.vec = kVecSpan;
  g.chunks = g.vec == kVecSpan ? (int)((row_bytes + 8) / 16) : (int)(row_bytes / g.vec);
  g.lpr_log2 = 0;
  while ((1 << g.lpr_log2) < g.chunks && g.lpr_log2 < 6) ++g.lpr_log2;
  const int gpb = kGatherThreads >> g.lpr_log2;
  const int64_t rows_per_iter = (int64_t)gpb * kGatherUnroll;
  g.grid = (n + rows_per_iter - 1) / rows_per_iter;
  const int64_t max_grid = (int64_t)256 * gather_wg_per_cu();  // 256 CUs x k workgroups: grid-stride beyond that
  if (g.grid > max_grid) g.grid = max_grid;
  if (g.grid < 1) g.grid = 1;
  return g;
}

// kNT: non-temporal loads of the source rows.  Measured on MI355X (770k random 200-B rows of a
// 490 MB table, sustained): plain loads 63 us, non-temporal loads 88-93 us -- the rows are not
// really read-once (hub rows repeat across batches and MALL/L2 catch them), so plain is the default;
// the stores stay non-temporal.
// move_rows_body: dst[r,:] = *ptr_of(key_of(r)) for r < n.  The source of an output row comes in two steps so
// that the loads of a round really are in flight together: key_of(r) is a pure load (an index, a source
// record), ptr_of(key) turns it into the row's address (range checks, selects).  A single functor doing
// both put control flow between the loads and the compiler waited for each before issuing the next.
template <int VEC, bool kNT, typename KeyFn, typename PtrFn>
__device__ __forceinline__ void move_rows_vec_body(KeyFn key_of, PtrFn ptr_of, int64_t n, int64_t row_bytes, int chunks,
                                                   int lpr_log2, char* __restrict__ dst, int64_t vblock, int64_t nvblocks) {
  using V = typename vec_of<VEC>::type;
  const int lpr = 1 << lpr_log2;
  const int g = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & (lpr - 1);
  const int gpb = kGatherThreads >> lpr_log2;  // row groups per workgroup
  const int64_t rows_per_iter = (int64_t)gpb * kGatherUnroll;
  // Software pipeline over the workgroup's grid-stride iterations: the keys of iteration k+1 are requested
  // right after the row loads of iteration k, so that an iteration costs ONE dependent round trip (rows) instead
  // of two (key, then row).  Loads return in order, so waiting for the rows leaves the younger key loads in flight.
  // Branch-free loads throughout: a row index past the end is clamped to the last row (loaded, never stored).
  // With `if (ok) v = load` the compiler put every load in its own exec-masked block behind an
  // s_waitcnt vmcnt(0) -- ONE load in flight per wavefront whatever the unroll.
  using K = decltype(key_of((int64_t)0));
  const int64_t stride = nvblocks * rows_per_iter;
  int64_t base = vblock * rows_per_iter;
  if (base >= n) return;
  // lpr is the power of two >= chunks: lanes past the row's last access load piece 0 (not predicated) and store nothing
  const bool lane_on = l < chunks;
  const int l0 = lane_on ? l : 0;
  K key_next[kGatherUnroll];
#pragma unroll
  for (int u = 0; u < kGatherUnroll; ++u) {
    const int64_t r = base + (int64_t)u * gpb + g;
    key_next[u] = key_of(r < n ? r : n - 1);
  }
  // Two bodies, chosen by WORKGROUP-UNIFORM conditions (a scalar branch): the full one -- every row of the
  // iteration exists and every lane of a group has a piece of the row -- stores unconditionally.  With a per-lane
  // predicate on the store the compiler sinks the row's LOAD into the predicated block as well (its only use),
  // behind an s_waitcnt vmcnt(0): that load then waits for everything in flight, the prefetched keys included.
  const bool dense_lanes = chunks == lpr;
  for (; base < n; base += stride) {
    const V* s[kGatherUnroll];
    V* d[kGatherUnroll];
    V v[kGatherUnroll];
    const int64_t nbase = base + stride;
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) {
      const int64_t r = base + (int64_t)u * gpb + g;
      d[u] = reinterpret_cast<V*>(dst + r * row_bytes);
      s[u] = reinterpret_cast<const V*>(ptr_of(key_next[u]));
    }
    if (dense_lanes && base + rows_per_iter <= n) {
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u) v[u] = kNT ? __builtin_nontemporal_load(&s[u][l]) : s[u][l];
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u) {  // next iteration's keys (clamped: past the end they are never used)
        const int64_t r = nbase + (int64_t)u * gpb + g;
        key_next[u] = key_of(r < n ? r : n - 1);
      }
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u) row_store(v[u], &d[u][l]);
      continue;
    }
    bool ok[kGatherUnroll];
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) ok[u] = base + (int64_t)u * gpb + g < n;
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) v[u] = kNT ? __builtin_nontemporal_load(&s[u][l0]) : s[u][l0];
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) {
      const int64_t r = nbase + (int64_t)u * gpb + g;
      key_next[u] = key_of(r < n ? r : n - 1);
    }
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u)
      if (ok[u] && lane_on) row_store(v[u], &d[u][l]);
    for (int c = l + lpr; c < chunks; c += lpr) {  // rows wider than one access per lane
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u) v[u] = kNT ? __builtin_nontemporal_load(&s[u][c]) : s[u][c];
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u)
        if (ok[u]) row_store(v[u], &d[u][c]);
    }
  }
}

// kVecSpan (see gather_geometry): row_bytes = 16k + 8, lpr = power of two >= k + 1 (<= 32) lanes per row, R = 64 / lpr
// (even) rows per wavefront and round -- consecutive output rows, the first one even, so the wavefront's span of
// R * row_bytes bytes starts 16-byte aligned and is a whole number C of 16-byte chunks (C <= 64).  Lane (row g, piece l)
// loads source bytes [16 l, 16 l + 16) of its row.  In an even row of the span a piece IS a chunk; in an odd row
// (start = 8 mod 16) its low half is the upper half of one chunk and its high half the lower half of the next.
// With P = the half of a piece that lands in the LOWER half of a chunk (even row: low, odd row: high) and Q = the
// other one, chunk c = {P of lane A(c), Q of lane B(c)}: four ds_bpermute per round, no LDS storage, no barrier.
template <bool kNT, typename KeyFn, typename PtrFn>
__device__ __forceinline__ void move_rows_span_body(KeyFn key_of, PtrFn ptr_of, int64_t n, int64_t row_bytes, int pieces,
                                                    int lpr_log2, char* __restrict__ dst, int64_t vblock,
                                                    int64_t nvblocks) {
  using K = decltype(key_of((int64_t)0));
  const int lpr = 1 << lpr_log2;
  const int g = threadIdx.x >> lpr_log2;
  const int l = threadIdx.x & (lpr - 1);
  const int gpb = kGatherThreads >> lpr_log2;
  const int lane = threadIdx.x & 63;
  const int rpw = 64 >> lpr_log2;                         // rows per wavefront and round
  const int g0 = (threadIdx.x >> 6) * rpw;                // first row group of this wavefront
  const int64_t rows_per_iter = (int64_t)gpb * kGatherUnroll;
  const int64_t stride = nvblocks * rows_per_iter;
  int64_t base = vblock * rows_per_iter;
  if (base >= n) return;
  // which lanes' halves make up chunk `lane` of the wavefront's span (loop invariant)
  const int rb = (int)row_bytes;
  const int nchunk = rpw * rb / 16;
  const int o = 16 * (lane < nchunk ? lane : 0);
  const int ra = o / rb, ba = o - ra * rb;
  const int rq = (o + 8) / rb, bq = (o + 8) - rq * rb;
  const int srcA = (ra << lpr_log2) + (ba >> 4), srcB = (rq << lpr_log2) + (bq >> 4);
  const bool odd = g & 1;
  const bool lane_on = l < pieces;
  const int l0 = lane_on ? l : 0;  // lanes past the row's last piece load piece 0 (not predicated); nothing refers to them
  K key_next[kGatherUnroll];
#pragma unroll
  for (int u = 0; u < kGatherUnroll; ++u) {
    const int64_t r = base + (int64_t)u * gpb + g;
    key_next[u] = key_of(r < n ? r : n - 1);
  }
  for (; base < n; base += stride) {
    const u32x4* s[kGatherUnroll];
    u32x4 v[kGatherUnroll];
    const int64_t nbase = base + stride;
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) s[u] = reinterpret_cast<const u32x4*>(ptr_of(key_next[u]));
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) v[u] = kNT ? __builtin_nontemporal_load(&s[u][l0]) : s[u][l0];
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) {  // next iteration's keys (clamped: past the end they are never used)
      const int64_t r = nbase + (int64_t)u * gpb + g;
      key_next[u] = key_of(r < n ? r : n - 1);
    }
    if (base + rows_per_iter <= n) {  // workgroup-uniform: every row of the iteration exists
#pragma unroll
      for (int u = 0; u < kGatherUnroll; ++u) {
        const uint32_t p0 = odd ? v[u].z : v[u].x, p1 = odd ? v[u].w : v[u].y;
        const uint32_t q0 = odd ? v[u].x : v[u].z, q1 = odd ? v[u].y : v[u].w;
        u32x4 c;
        c.x = __shfl(p0, srcA, 64);
        c.y = __shfl(p1, srcA, 64);
        c.z = __shfl(q0, srcB, 64);
        c.w = __shfl(q1, srcB, 64);
        char* span = dst + (base + (int64_t)u * gpb + g0) * row_bytes;
        if (lane < nchunk) row_store(c, reinterpret_cast<u32x4*>(span) + lane);
      }
      continue;
    }
    // last, partial iteration: every lane stores the two halves of its own piece
#pragma unroll
    for (int u = 0; u < kGatherUnroll; ++u) {
      const int64_t r = base + (int64_t)u * gpb + g;
      if (r < n && lane_on) {
        u32x2* d = reinterpret_cast<u32x2*>(dst + r * row_bytes + 16 * l);
        row_store(u32x2{v[u].x, v[u].y}, d);
        if (16 * l + 8 < rb) row_store(u32x2{v[u].z, v[u].w}, d + 1);
      }
    }
  }
}

template <int VEC, bool kNT, typename KeyFn, typename PtrFn>
__device__ __forceinline__ void move_rows_body(KeyFn key_of, PtrFn ptr_of, int64_t n, int64_t row_bytes, int chunks,
                                               int lpr_log2, char* __restrict__ dst, int64_t vblock, int64_t nvblocks) {
  if constexpr (VEC == kVecSpan)
    move_rows_span_body<kNT>(key_of, ptr_of, n, row_bytes, chunks, lpr_log2, dst, vblock, nvblocks);
  else
    move_rows_vec_body<VEC, kNT>(key_of, ptr_of, n, row_bytes, chunks, lpr_log2, dst, vblock, nvblocks);
}

template <int VEC, typename IdxT, bool kNT = false>
__device__ __forceinline__ void gather_rows_body(const char* __restrict__ src, const IdxT* __restrict__ idx,
                                                 int64_t n, int64_t row_bytes, int chunks, int lpr_log2,
                                                 char* __restrict__ dst, int64_t vblock, int64_t nvblocks,
                                                 int64_t src_stride) {
  if (n <= 0) return;
  move_rows_body<VEC, kNT>([=](int64_t r) { return idx[r]; }, [=](IdxT i) { return src + (int64_t)i * src_stride; }, n,
                           row_bytes, chunks, lpr_log2, dst, vblock, nvblocks);
}

// Same with caller-supplied (untrusted) indices: an index outside [0, src_rows) reads row 0 and raises
// SPP_AERR_GATHER_INDEX in `err` (the reference's serial_index would read out of bounds).
template <int VEC, typename IdxT, bool kNT = false>
__device__ __forceinline__ void gather_rows_checked_body(const char* __restrict__ src, int64_t src_rows,
                                                         const IdxT* __restrict__ idx, int64_t n, int64_t row_bytes,
                                                         int chunks, int lpr_log2, char* __restrict__ dst,
                                                         int64_t vblock, int64_t nvblocks, int64_t src_stride,
                                                         int32_t* err, int32_t err_bit) {
  if (n <= 0) return;
  move_rows_body<VEC, kNT>(
      [=](int64_t r) { return idx[r]; },
      [=](IdxT k) {
        int64_t i = (int64_t)k;
        if ((uint64_t)i >= (uint64_t)src_rows) {
          if (err) __hip_atomic_fetch_or(err, err_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          i = 0;
        }
        return src + i * src_stride;
      },
      n, row_bytes, chunks, lpr_log2, dst, vblock, nvblocks);
}

}  // namespace spp
