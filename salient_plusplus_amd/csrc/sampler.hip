// a2-a4 / K1: on-GPU multi-hop neighbour sampler, bit-exact with the sequential CPU algorithm of
// the reference (sample_adj, fast_sampler/sample_cpu.hpp:25-143, driven by multilayer_sample,
// fast_sampler/fast_sampler.cpp:191-236).
//
// What has to be reproduced:
//   * one std::mt19937 stream per batch, consumed in scan order: target i of hop h (only if
//     deg_i > fanout) uses draws  base_h + fanout * #{k < i : deg_k > fanout} ... (+fanout);
//   * Robert-Floyd picks per target (sample_cpu.hpp:97-110);
//   * local ids handed out in FIRST-SEEN order of the sequential scan (targets in order, picks in
//     emission order) through a running map shared by all hops (sample_cpu.hpp:50-60);
//   * each output row sorted by local id (sample_cpu.hpp:126).
//
// GPU formulation.  A launch processes a GROUP of up to 16 independent batches (blockIdx.y selects
// the batch slot) so that even the small first hops put >> 256 workgroups on the chip.  Per hop
// (T = nodes collected so far = targets; all counts stay on the device):
//   k_hop_count     lane/target : rowptr -> deg, row_start; per-workgroup sums of (#edges, #sampled)
//                                 (hop 0: done by k_seed_init)
//   k_hop_pick      lane/target : sums of the workgroups before its own -> out_rowptr[i], RNG offset
//                                 (the last workgroup records E_h, #sampled); Floyd picks staged in LDS;
//                                 col reads -> neighbour id of every edge position
//                                 (k_hop_scan: single-workgroup scan instead, generic / very large hops)
//   k_bucket_scatter tile/8k edges: tile bucket-sorted in LDS, (node, position) pairs written out as
//                                 coalesced runs into FIXED-CAPACITY bucket regions (no counting pass; what does
//                                 not fit goes to an overflow list); inv[p] = where edge p went
//   k_bucket_dedup  workgroup/bucket: LDS table of the bucket's known nodes + candidates ->
//                                 per edge (bucket order): final local id, or T + earliest position of a new node
//   k_hop_flag      4 edges/lane: results back to position order by reads (res[inv[p]]); bitmap of
//                                 first occurrences + per-word / per-block counts; the last workgroup
//                                 (ticket) scans the block counts -> number of new nodes
//                                 (k_hop_flag_tiled, round 5, behind k_bucket_scatter: the same per 8 k-edge scatter tile,
//                                 the results fetched in the tile's staging order -- runs instead of random words)
//   k_hop_rows      lane/target : local ids of the row (rank of a new node = prefix + popcount of the
//                                 bitmap), n_ids append at first occurrences, LDS rank-sort, out_col
//                                 (k_hop_rows_coalesced, round 5, fanout <= 28: the workgroup's run of positions through LDS)
// Round 5: on the small hops k_hop_pick<kFuse> does k_bucket_scatter's work for its own 256 targets' edges.
// The batch's mt19937 stream (mt19937.hip.h) is produced by k_rng_fill, one workgroup per batch, into
// one of two per-slot buffers: the Session generates it a whole group ahead on its own stream, so
// the ~0.5 ms serial recurrence never sits on the sampling critical path.
// The node table is radix-partitioned: 64-bit entries (key<<32 | value) live in per-bucket LDS tables
// for the duration of one k_bucket_dedup workgroup and, between hops, in compact per-bucket known
// lists in HBM; value < T means "final local id", value >= T means "T + position of the earliest
// edge that reaches this node in this hop".
//
// Hops with fanout < 0 (all neighbours) or fanout > 32 take a generic path (edge-parallel expand,
// hipcub segmented sort, one host sync per hop to size the launch); such samplers run one batch
// per launch.
#include "spp_internal.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <map>
#include <memory>
#include <mutex>
#include <tuple>

#include "gather_body.hip.h"
#include "partition_common.hip.h"

#ifndef SPP_TILE_NT
#define SPP_TILE_NT 1024
#endif
#include "mt19937.hip.h"
#include "sampler_internal.h"

namespace spp {

spp_status gather_rows_i32(const void* src, int64_t src_rows, int64_t row_bytes, int64_t src_stride,
                           const int32_t* idx, int64_t n, void* dst, hipStream_t st);

constexpr int kNT = 256;         // workgroup size of the per-target / per-edge kernels
constexpr int kFastMaxFanout = 32;
constexpr int kScanNT = 1024;

enum : int32_t { kErrEdgeCap = 1, kErrNodeCap = 2, kErrDrawCap = 4, kErrBucketCap = 8 };

// Geometry of the radix-partitioned dedup (fixed per sampler).
struct DedupGeom {
  int32_t nb_log2;   // log2(#buckets), buckets = top bits of hash_node(key)
  int32_t nb;        // #buckets (<= kMaxBuckets)
  int32_t kcap;      // capacity of one bucket's known-node list
};
constexpr int kMaxBucketsLog2 = 12;
constexpr int kMaxBuckets = 1 << kMaxBucketsLog2;
constexpr int kMaxFineLog2 = 6;              // a hop's bucket spans at most 2^6 fine buckets
constexpr int kMaxFinePerCoarse = 1 << kMaxFineLog2;
constexpr int kTileNT = SPP_TILE_NT;         // workgroup size of the two tile kernels (more waves per tile: latency bound)
constexpr int kScatterTile = 8192;           // edges k_bucket_scatter sorts in LDS per workgroup (48 KB + 8 B per bucket)
constexpr int kScatterEPT = kScatterTile / kTileNT;
static_assert(kScatterTile % kTileNT == 0 && kScatterTile <= 65536, "tile-local edge indices are 16 bit");
#ifndef SPP_DEDUP_SLOTS12
#define SPP_DEDUP_SLOTS12 3072  // 24 KB: six workgroups per compute unit, a quarter less to clear per bucket (4096: +3.5 % lone chain)
#endif
constexpr int kDedupRegs = 6;                // pairs per thread k_bucket_dedup keeps in registers (6 x 256 edges per bucket)
// Fixed-capacity bucket regions (round 4; a counting pass over the hop's edges used to size them exactly).  A hop
// with edge capacity pcap and nbk buckets gives every bucket room for twice its share plus 64 pairs; what does not fit
// (a hub reached by hundreds of edges of one hop) goes to an overflow list behind the regions, which only the
// workgroups of the buckets that overflowed walk.  region_for(cap) bounds nbk * bcap for every hop geometry.
__host__ __device__ inline int64_t region_for(int64_t edge_cap) { return 2 * edge_cap + 66 * (int64_t)(1 << 12); }
static inline int32_t bucket_cap(int64_t pcap, int64_t nbk) { return (int32_t)((2 * pcap + nbk - 1) / nbk) + 64; }
constexpr uint32_t kPending = 0x80000000u;  // known-list value = kPending | edge position of the previous hop

// Workgroup -> (batch of the group, block within the batch).  Every grouped kernel is launched as a
// 1-D grid of gx * n workgroups.  interleave = 1: consecutive workgroup ids go to consecutive BATCHES
// (batch = id % n).  The hardware deals workgroups round-robin over the 8 XCDs, so with n = 8 all
// workgroups of one batch run on ONE XCD and the batch's scratch arrays (a few MB per hop) are written
// and re-read through a single L2: scattered 4/8-byte stores merge into whole lines before they
// leave it and the random re-reads of the next kernel hit.  (Placement is a speed matter only: the
// results do not depend on it.)  interleave = 0 is the plain batch-major order.
struct GroupGrid {
  int32_t first_slot;
  int32_t n;          // batches in the group
  uint32_t gx;        // blocks per batch
  int32_t interleave;
};
#define SPP_GROUP_BLOCK(gg)                         \
  uint32_t bx_, by_;                                \
  if ((gg).interleave) {                            \
    by_ = blockIdx.x % (uint32_t)(gg).n;            \
    bx_ = blockIdx.x / (uint32_t)(gg).n;            \
  } else {                                          \
    by_ = blockIdx.x / (gg).gx;                     \
    bx_ = blockIdx.x - by_ * (gg).gx;               \
  }                                                 \
  (void)bx_

// Pointers that reach a kernel through SlotPtrs (loaded from memory) are generic to the compiler: it emits
// FLAT loads, which count on the LDS counter as well -- every wait for an LDS operation or a wavefront
// shuffle then also waits for every global load in flight.  G() names the address space, so that the hot
// kernels get global_load/global_store and their loads stay in flight across LDS work.
#define SPP_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ SPP_GLOBAL T* G(T* p) {
  return (SPP_GLOBAL T*)p;
}
template <typename T>
__device__ __forceinline__ const SPP_GLOBAL T* G(const T* p) {
  return (const SPP_GLOBAL T*)p;
}

// device-resident bookkeeping of one batch slot (copied to pinned host memory after sampling)
struct SlotState {
  int32_t cnt[SPP_MAX_HOPS + 1];   // cnt[0] = #seeds, cnt[h+1] = #nodes after hop h (processing order)
  int32_t E[SPP_MAX_HOPS];         // sampled edges of hop h
  int32_t nsmp[SPP_MAX_HOPS];      // targets with deg > fanout in hop h
  int64_t dbase[SPP_MAX_HOPS + 1]; // RNG draws consumed before hop h (relative to rng_skip)
  int32_t error;
  int32_t pad;
  const uint32_t* rng;             // this batch's mt19937 draws (slot buffer or the sampler's epoch arena)
  int32_t pcnt[kPartBuckets];      // ownership buckets of the node list (spp_partition_cfg): [0,P) owners, [P] cache hits
};

struct __attribute__((aligned(16))) RankWord {
  unsigned long long bits;
  uint32_t pre;
  uint32_t pad;
};

// a rank record through a global-address-space pointer (one 16-byte load)
typedef uint32_t rw_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ RankWord load_rank_word(const SPP_GLOBAL RankWord* p, uint32_t w) {
  const rw_u4 v = reinterpret_cast<const SPP_GLOBAL rw_u4*>(p)[w];
  RankWord r;
  r.bits = ((unsigned long long)v.y << 32) | v.x;
  r.pre = v.z;
  r.pad = 0u;
  return r;
}

struct SlotPtrs {
  int32_t* n_ids;
  uint8_t* dtag;       // [Ucap] min(degree, cap) of every node of the list (degree-tagged neighbour ids; NULL: off)
  int32_t* deg;
  int64_t* rowstart;
  int32_t* cval;       // neighbour node id of every edge position (later: local id, generic path)
  // edges regrouped by bucket: (node id << 32) | edge position.  Bucket b of a hop owns the fixed region
  // [b * bcap, (b + 1) * bcap); pairs past a bucket's capacity live in the overflow list behind all regions, at
  // [region, region + ovfc[h]) (region = region_for(edge capacity): the same for every hop)
  unsigned long long* bpairs;
  uint32_t* inv;       // where k_bucket_scatter put edge position p in bpairs (bucket order); with the tiled flag pass
                       // (round 5): where it put the pair in STAGING SLOT k of the position's 8 k-edge tile -- see tidx
  uint16_t* tidx;      // tile-local position of staging slot k (k_bucket_scatter with `tiled`, read by k_hop_flag_tiled)
  uint32_t* res;       // table value of every edge, indexed like bpairs (k_bucket_dedup)
  uint32_t* evals;     // final table value of every edge position: local id (< T) or T + first position
  // first occurrences of the hop (new nodes), by edge position: per 64 positions one 16-byte record
  // {bitmap word, set bits in the earlier words of the same 256-position block}, plus the exclusive
  // prefix of the blocks' totals.  rank(q) = fsum[q >> 8] + fwords[q >> 6].pre + popcount(bits below
  // q & 63): one 16-byte read (L2) and one of a table small enough for L1 replace a 4-byte rank per edge.
  RankWord* fwords;
  int32_t* fsum;
  unsigned long long* known;   // [nb][kcap] known nodes per bucket: (node id << 32) | local id (or kPending | pos)
  int32_t* kcount;     // [nb] entries in each known list
  int32_t* bfill;      // [nb] pairs each bucket received in the current hop, overflowed ones included (zero between
                       // hops: the bucket's k_bucket_dedup workgroup resets it)
  int32_t* ovfc;       // [SPP_MAX_HOPS] pairs of hop h that did not fit their bucket's region (zeroed per chain)
  uint32_t* rng[2];    // draws rng_skip .. of the batch stream (ping-pong: generated one group ahead)
  int32_t* bsum0;
  int32_t* bsum1;
  int32_t* ctr;        // "last workgroup" ticket counter (zero between launches)
  SlotState* st;
  int32_t* out_rowptr[SPP_MAX_HOPS];  // processing order
  int32_t* out_col[SPP_MAX_HOPS];
  // ownership bucketing (spp_partition_cfg; NULL when off)
  int32_t* parts;      // [Ucap] node ids grouped by owner, then (separately) ...
  int32_t* pcached;    // [Ucap] cache rows of the cache hits
  int32_t* pperm;      // [Ucap] perm_partition_to_mfg
  int2* psrc;          // [Ucap] where node i's feature row comes from: {bucket, row inside the bucket's source}
  uint8_t* pbucket;    // [Ucap] bucket of every node
  int32_t* pblk;       // [P+1][pnblk] per-workgroup bucket counts -> exclusive offsets
};

// per-launch description of a group of batches (passed by value)
struct GroupArgs {
  int32_t first_slot;
  int32_t n;
  GroupGrid grid;                  // how this launch's workgroups map to (batch, block)
  int32_t rng_buf;                 // slot ping-pong buffer holding the draws (ignored when rng[i] is set)
  const int64_t* seeds[kMaxGroup];
  int32_t n_seeds[kMaxGroup];
  uint32_t rng_seed[kMaxGroup];
  int64_t rng_skip[kMaxGroup];
  const uint32_t* rng[kMaxGroup];  // draws of batch i in the epoch arena (NULL: the slot's rng[rng_buf])
};

// ----------------------------------------------------------------------------------------------
// node table: radix-partitioned, LDS resident
// ----------------------------------------------------------------------------------------------
// A global hash table costs one random 64/128-B fabric transaction per probe, per atomic and per
// re-read for 8 useful bytes (measured: ~540 MB of L2<->fabric traffic per batch for ~21 MB of
// algorithmic bytes).  Instead the edges of a hop are regrouped by BUCKET (top bits of the node-id
// hash) with sequential traffic, and one workgroup per bucket resolves its keys in an LDS table that
// also holds the bucket's already-known nodes; only the per-edge result goes back to HBM.
__device__ __forceinline__ uint32_t bucket_of(uint32_t key, int32_t nb_log2) {
  return nb_log2 ? (hash_node(key) >> (32 - nb_log2)) : 0u;
}

// Slot of `key` in a table of NS slots (NS need not be a power of two: multiply-shift range reduction of a second
// multiplicative hash, independent of the bucket bits).
template <int NS>
__device__ __forceinline__ uint32_t lds_slot_of(uint32_t key) {
  return (uint32_t)(((unsigned long long)(key * 0x85EBCA6Bu) * (unsigned long long)NS) >> 32);
}

// Insert-or-update with min (kMax = false) / max (kMax = true) on the value; *ovf is set when the
// table is full.  LDS atomics of one workgroup are coherent, the pre-read only saves atomics: kPreRead = false goes
// straight to the compare-and-swap (one table operation instead of two for a key that is not in the table yet -- nearly
// every candidate of the big last hop).
template <bool kMax, int NS, bool kPreRead = true>
__device__ __forceinline__ void lds_upsert(unsigned long long* tab, uint32_t key, uint32_t val, int* ovf) {
  const unsigned long long entry = ((unsigned long long)key << 32) | val;
  uint32_t h = lds_slot_of<NS>(key);
  for (uint32_t probes = 0; probes < (uint32_t)NS; ++probes) {
    unsigned long long cur = kPreRead ? tab[h] : kEmptySlot;
    if (cur == kEmptySlot) {
      cur = atomicCAS(&tab[h], kEmptySlot, entry);
      if (cur == kEmptySlot) return;
    }
    if ((uint32_t)(cur >> 32) == key) {
      const uint32_t have = (uint32_t)cur;
      if (kMax) {
        if (have < val) atomicMax(&tab[h], entry);
      } else {
        if (have > val) atomicMin(&tab[h], entry);
      }
      return;
    }
    h = (h + 1 == (uint32_t)NS) ? 0u : h + 1;
  }
  *ovf = 1;
}

template <int NS>
__device__ __forceinline__ uint32_t lds_find(const unsigned long long* tab, uint32_t key) {
  uint32_t h = lds_slot_of<NS>(key);
  for (uint32_t probes = 0; probes < (uint32_t)NS; ++probes) {
    const unsigned long long cur = tab[h];
    if ((uint32_t)(cur >> 32) == key && cur != kEmptySlot) return (uint32_t)cur;
    if (cur == kEmptySlot) break;
    h = (h + 1 == (uint32_t)NS) ? 0u : h + 1;
  }
  return 0xffffffffu;
}

// the batch's mt19937 stream: draws [skip, skip + dcap) of mt19937(seed) -> rng[buf]
__global__ __launch_bounds__(kMtThreads) void k_rng_fill(const SlotPtrs* __restrict__ slots, GroupArgs ga,
                                                          int64_t dcap) {
  __shared__ uint32_t x[2 * kMtRing];
  const SlotPtrs& s = slots[ga.first_slot + blockIdx.y];
  const uint32_t seed = ga.rng_seed[blockIdx.y];
  const int64_t skip = ga.rng_skip[blockIdx.y];
  uint32_t* out = s.rng[ga.rng_buf];
  const int64_t cap = dcap + kMtSlack;
  mt_block_seed(x, seed, skip, cap, out);
  mt_block_advance(x, 624, skip + dcap, skip, cap, out);
}

// The streams of a whole epoch in one launch: stream b = draws [0, dcap) of mt19937(seeds[b]) at
// arena + b * stride.  A batch's generator seed is a function of its range end only
// (fast_sampler.cpp:994) and the range table repeats every epoch, so this runs once per range table
// and the result is kept by the sampler: 1178 batches x 4.3 MB = 5 GB at papers100M scale (288 GB HBM).
__global__ __launch_bounds__(kMtThreads) void k_rng_arena(uint32_t* __restrict__ arena, int64_t stride,
                                                           const uint32_t* __restrict__ seeds, int64_t dcap) {
  __shared__ uint32_t x[2 * kMtRing];
  uint32_t* out = arena + (int64_t)blockIdx.x * stride;
  const int64_t cap = dcap + kMtSlack;
  mt_block_seed(x, seeds[blockIdx.x], 0, cap, out);
  mt_block_advance(x, 624, dcap, 0, cap, out);
}

// One-off int32 copy of the neighbour array (node ids are < 2^31, fast_sampler.cpp:196-199 narrows them
// anyway): a sampled row spans half as many 128-B fetch granules, and the random neighbour reads of the
// last hop are the sampler's largest single source of HBM traffic.
__global__ __launch_bounds__(256) void k_narrow_col(const int64_t* __restrict__ col, int64_t nnz,
                                                    int32_t* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256)
    out[i] = (int32_t)col[i];
}

// Degree-tagged neighbour ids.  Node ids need idbits = ceil(log2 num_nodes) bits (27 for 111 M nodes); the spare top
// bits of every entry of the int32 neighbour array -- and of the row stubs copied from it -- carry
// min(degree of that neighbour, cap), cap = 2^(32 - idbits) - 1.  A hop's degree pass only decides "all neighbours or
// f picks" and counts edges, for which min(degree, cap) is exact while cap > f: the nodes a hop discovers then bring
// their own degree along, and the NEXT hop's degree pass is a sequential read of one byte per target instead of one
// random 128-byte line per target (25 MB of the chain's 143 MB per batch, 5.6 us of a 137 us step).  The exact degree
// and the row start a sampled row needs come out of the stub line k_hop_pick fetches anyway.  Built once per graph:
// a saturated-degree byte per node (sequential), then one random byte read per edge (0.2 s for 3.2 G edges).
__global__ __launch_bounds__(256) void k_build_deg8(const int64_t* __restrict__ rowptr, int64_t num_nodes,
                                                    uint8_t* __restrict__ out) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < num_nodes; v += (int64_t)gridDim.x * 256) {
    const int64_t d = rowptr[v + 1] - rowptr[v];
    out[v] = (uint8_t)(d > 255 ? 255 : (d < 0 ? 0 : d));
  }
}

__global__ __launch_bounds__(256) void k_narrow_col_tagged(const int64_t* __restrict__ col, int64_t nnz,
                                                           const uint8_t* __restrict__ deg8, int32_t idbits, uint32_t cap,
                                                           int32_t* __restrict__ out) {
  for (int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i0 < nnz; i0 += (int64_t)gridDim.x * 256 * 4) {
    uint32_t c[4], d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (uint32_t)col[i0 + u < nnz ? i0 + u : nnz - 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = deg8[c[u]];  // the four random byte reads in flight together
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u < nnz) out[i0 + u] = (int32_t)(c[u] | ((d[u] < cap ? d[u] : cap) << idbits));
  }
}

// Row stubs: one 128-byte, 128-byte-aligned record per node -- {degree, row start (lo, hi), the first 29
// neighbours} -- built once per graph.  A sampled row's degree AND (for most rows) every neighbour the
// picks can name then sit in ONE cache line at a computable address: k_hop_count / k_seed_init read the
// header instead of a rowptr pair, k_hop_pick reads the line instead of an unaligned 128-byte piece of the
// neighbour array (1.75 lines on average) -- only picks at positions >= 29 of longer rows still go to the
// neighbour array.  14 GB for the 111 M-node graph (HBM is 288 GB); skipped when memory is short.
constexpr int kStubInts = 32;
constexpr int kStubNbr = kStubInts - 3;
typedef int32_t stub4 __attribute__((ext_vector_type(4)));

template <typename ColT>
__global__ __launch_bounds__(256) void k_build_stubs(const int64_t* __restrict__ rowptr, const ColT* __restrict__ col,
                                                     int64_t num_nodes, stub4* __restrict__ out) {
  const int part = threadIdx.x & 7;
  for (int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); v < num_nodes; v += (int64_t)gridDim.x * 32) {
    const int64_t rs = rowptr[v];
    const int64_t d64 = rowptr[v + 1] - rs;
    const int32_t deg = d64 > 0x7fffffff ? 0x7fffffff : (int32_t)d64;
    int32_t e[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int idx = part * 4 + k - 3;  // neighbour position held by this word (header words: negative)
      e[k] = (idx >= 0 && idx < deg) ? (int32_t)col[rs + idx] : 0;
    }
    if (part == 0) {
      e[0] = deg;
      e[1] = (int32_t)(uint32_t)rs;
      e[2] = (int32_t)(rs >> 32);
    }
    out[v * 8 + part] = stub4{e[0], e[1], e[2], e[3]};
  }
}

// degree and row start of node v: from its stub's first 16 bytes, or from the rowptr pair
__device__ __forceinline__ void row_header(const int64_t* __restrict__ rowptr, const stub4* __restrict__ stubs, int32_t v,
                                           int32_t& deg, int64_t& rs) {
  if (stubs) {
    const stub4 hd = stubs[(int64_t)v * 8];
    deg = hd.x;
    rs = ((int64_t)hd.z << 32) | (uint32_t)hd.y;
  } else {
    rs = rowptr[v];
    deg = (int32_t)(rowptr[v + 1] - rs);
  }
}

__device__ __forceinline__ void target_counts(int32_t deg, int32_t f, int32_t replace, int32_t& cnt, int32_t& smp) {
  if (replace && f >= 0) {
    // with replacement (sample_cpu.hpp:74-82): f draws of gen() % deg whenever deg > 0
    smp = deg > 0 ? 1 : 0;
    cnt = smp ? f : 0;
    return;
  }
  // sample_cpu.hpp:67-73 (f < 0: all), :91-94 (deg <= f: all), :97-110 (Floyd: f picks)
  smp = (f >= 0 && deg > f) ? 1 : 0;
  cnt = smp ? f : (deg > 0 ? deg : 0);
}

// get_initial_sample_adj_hash_map (sample_cpu.hpp:13-19): n_id_map[n_ids[i]] = i, so a duplicated
// seed keeps its LAST position: every seed is appended to its bucket's known list and the LDS
// insert of known entries takes the max.
// Also the degree pass of hop 0 (k_hop_count with h = 0: the targets are the seeds themselves), so a
// chain starts with one launch instead of two.
__global__ __launch_bounds__(kNT) void k_seed_init(const SlotPtrs* __restrict__ slots, GroupArgs ga, DedupGeom g,
                                                    const int64_t* __restrict__ rowptr,
                                                    const stub4* __restrict__ stubs, int32_t f, int32_t replace,
                                                    uint32_t tag_cap) {
  SPP_GROUP_BLOCK(ga.grid);
  __shared__ int32_t lds[2][kNT / kWave + 1];
  const SlotPtrs& s = slots[ga.first_slot + by_];
  const int64_t* __restrict__ seeds = ga.seeds[by_];
  const int32_t n_seeds = ga.n_seeds[by_];
  const int i = bx_ * kNT + threadIdx.x;
  if ((int64_t)bx_ * kNT >= n_seeds && bx_ != 0) return;
  if (i == 0) {
    s.st->cnt[0] = n_seeds;
    s.st->dbase[0] = 0;
    s.st->error = 0;
    s.st->rng = ga.rng[by_] ? ga.rng[by_] : s.rng[ga.rng_buf];
  }
  int32_t cnt = 0, smp = 0;
  if (i < n_seeds) {
    const int32_t v = (int32_t)seeds[i];  // narrowing of fast_sampler.cpp:196-199
    int32_t deg;
    int64_t rs;
    row_header(rowptr, stubs, v, deg, rs);
    s.n_ids[i] = v;
    const uint32_t b = bucket_of((uint32_t)v, g.nb_log2);
    const int32_t j = atomicAdd(&s.kcount[b], 1);
    if (j < g.kcap) s.known[(int64_t)b * g.kcap + j] = ((unsigned long long)(uint32_t)v << 32) | (uint32_t)i;
    else atomicOr(&s.st->error, kErrBucketCap);
    s.deg[i] = deg;
    s.rowstart[i] = rs;
    if (tag_cap) s.dtag[i] = (uint8_t)((uint32_t)deg < tag_cap ? (uint32_t)deg : tag_cap);
    target_counts(deg, f, replace, cnt, smp);
  }
  int32_t tc, ts;
  block_exclusive_scan<int32_t, kNT>(cnt, lds[0], &tc);
  block_exclusive_scan<int32_t, kNT>(smp, lds[1], &ts);
  if (threadIdx.x == 0) {
    s.bsum0[bx_] = tc;
    s.bsum1[bx_] = ts;
  }
}

// ----------------------------------------------------------------------------------------------
// per-target degree pass
// ----------------------------------------------------------------------------------------------

// ---- "last workgroup finishes the job" ---------------------------------------------------------
// Producer workgroups publish partial results with device-scope atomics, drain them (s_waitcnt
// vmcnt(0)), take a ticket from a device-scope counter, and the workgroup that draws the last ticket
// reads the partials back with agent-scope atomic loads and finishes the job: the placement-independent
// 8-byte-granule hand-off of cdna_hip_programming.md G16 (atomics on both sides).  Only used where few
// workgroups take tickets (one per 1024 elements, k_hop_flag): with one ticket per 256 elements the
// same-address atomics and their round trip made k_hop_count / k_hop_flag slower than a separate
// single-workgroup scan launch (measured: sampling alone 76 -> 113 us/batch), so those keep k_hop_scan*.
__device__ __forceinline__ int32_t acquire_i32(const int32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// called by thread 0 once every wave of the workgroup has drained its atomics and passed a barrier
__device__ __forceinline__ bool take_ticket_is_last(int32_t* ctr, int32_t expected) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const int32_t t = __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool last = (t == expected - 1);
  if (last) __hip_atomic_store(ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return last;
}

// One WAVEFRONT per 256 targets (the unit k_hop_pick's workgroups work in): four targets per lane, their
// loads in flight together, and the unit's two sums by a wavefront reduction -- no LDS, no barrier.
// Two round trips: {state words, node ids, stored degrees} at clamped indices, then the row headers --
// unpredicated (lanes without a new target read node 0's header): a predicate on a load makes the
// compiler wait for it before it issues the next one.
__global__ __launch_bounds__(kNT) void k_hop_count(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                    const int64_t* __restrict__ rowptr,
                                                    const stub4* __restrict__ stubs, int32_t use_tags,
                                                    int32_t h, int32_t f, int32_t replace, int32_t tcap) {
  SPP_GROUP_BLOCK(gg);
  constexpr int kPer = kNT / kWave;  // targets per lane
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* n_ids = G(s.n_ids);
  SPP_GLOBAL int32_t* degp = G(s.deg);
  SPP_GLOBAL int64_t* rsp = G(s.rowstart);
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t unit = (int64_t)bx_ * (kNT / kWave) + threadIdx.x / kWave;  // index of the 256-target unit
  const int64_t i0 = unit * kNT;
  if (i0 >= tcap) return;
  const int32_t T = st->cnt[h];
  const int32_t Tprev = h > 0 ? st->cnt[h - 1] : 0;
  int32_t v[kPer], deg[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int64_t i = i0 + u * kWave + lane;
    const int64_t ic = i < tcap ? i : tcap - 1;
    v[u] = n_ids[ic];
    deg[u] = degp[ic];  // a target of the previous hop as well: its degree and row start are still in place
  }
  uint8_t tg[kPer];
  if (use_tags) {  // the nodes' own degree tags, written next to n_ids by the hop that discovered them: sequential
    const SPP_GLOBAL uint8_t* dtag = G(s.dtag);
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      const int64_t i = i0 + u * kWave + lane;
      tg[u] = dtag[i < tcap ? i : tcap - 1];
    }
  }
  if (i0 >= T) return;
  int64_t rs[kPer];
  int32_t nd[kPer];
  bool fresh[kPer];
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int64_t i = i0 + u * kWave + lane;
    fresh[u] = i < T && i >= Tprev;
    if (!fresh[u]) v[u] = 0;
  }
  if (use_tags) {  // the row start is not needed: k_hop_pick takes it (and the exact degree) from the stub line
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      nd[u] = tg[u];
      rs[u] = 0;
    }
  } else if (stubs) {
    stub4 hd[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) hd[u] = stubs[(int64_t)v[u] * 8];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      nd[u] = hd[u].x;
      rs[u] = ((int64_t)hd[u].z << 32) | (uint32_t)hd[u].y;
    }
  } else {
    int64_t re[kPer];
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
      rs[u] = rowptr[v[u]];
      re[u] = rowptr[v[u] + 1];
    }
#pragma unroll
    for (int u = 0; u < kPer; ++u) nd[u] = (int32_t)(re[u] - rs[u]);
  }
  int32_t cnt = 0, smp = 0;
#pragma unroll
  for (int u = 0; u < kPer; ++u) {
    const int64_t i = i0 + u * kWave + lane;
    if (fresh[u]) {
      deg[u] = nd[u];
      degp[i] = nd[u];
      if (!use_tags) rsp[i] = rs[u];
    }
    if (i < T) {
      int32_t c, m;
      target_counts(deg[u], f, replace, c, m);
      cnt += c;
      smp += m;
    }
  }
  cnt = wave_inclusive_scan(cnt);
  smp = wave_inclusive_scan(smp);
  if (lane == kWave - 1) {
    G(s.bsum0)[unit] = cnt;
    G(s.bsum1)[unit] = smp;
  }
}

// in-place exclusive scan of n workgroup sums by ONE workgroup; returns the total to every thread
__device__ int32_t scan_block_sums(int32_t* a, int32_t n, int32_t* lds) {
  int32_t carry = 0;
  for (int32_t base = 0; base < n; base += kScanNT) {
    const int32_t i = base + threadIdx.x;
    const int32_t v = (i < n) ? a[i] : 0;
    int32_t tot;
    const int32_t ex = block_exclusive_scan<int32_t, kScanNT>(v, lds, &tot);
    if (i < n) a[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  return carry;
}

__global__ __launch_bounds__(kScanNT) void k_hop_scan(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                       int32_t h, int32_t f, int32_t ecap, int64_t dcap) {
  SPP_GROUP_BLOCK(gg);
  __shared__ int32_t lds[kScanNT / kWave + 1];
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const int32_t T = s.st->cnt[h];
  const int32_t nblk = (T + kNT - 1) / kNT;
  const int32_t E = scan_block_sums(s.bsum0, nblk, lds);
  const int32_t S = scan_block_sums(s.bsum1, nblk, lds);
  if (threadIdx.x == 0) {
    s.st->E[h] = E;
    s.st->nsmp[h] = S;
    s.out_rowptr[h][T] = E;
    if (E > ecap) atomicOr(&s.st->error, kErrEdgeCap);
    if (s.st->dbase[h] + (int64_t)(f > 0 ? f : 0) * S > dcap) atomicOr(&s.st->error, kErrDrawCap);
  }
}

// ----------------------------------------------------------------------------------------------
// picks + col reads + node-table insert (fast path: 0 <= fanout <= 32)
// ----------------------------------------------------------------------------------------------
// kFuse (round 5; small hops, with self_prefix): the workgroup's edges -- 256 targets' worth, one contiguous run of edge
// positions -- are regrouped by bucket right here instead of in a k_bucket_scatter launch of their own.  With the 16 /
// 128 buckets of hops 1 / 2 a workgroup's 3840 / 2560 edges give bucket runs of hundreds / tens of pairs; the lone chain
// loses the first 1024-thread tile kernel of the hop (9.8 us alone, 60 us beside another chain: it waits for a compute
// unit with 16 free wave slots) and the re-read of cval.  FuseArgs: the scatter's geometry.
struct FuseArgs {
  int32_t cb_log2, bcap, region;
  uint32_t idmask;
  int32_t lds_off;   // first int of the staging area inside the dynamic LDS block (behind the picks' columns)
  int32_t tile_cap;  // edges a workgroup can hold: kNT * max(1, f)
};

template <bool kGeneric, typename ColT, bool kStub, bool kHdr = false, bool kFuse = false>
__global__ __launch_bounds__(kNT) void k_hop_pick(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                   const ColT* __restrict__ col, const stub4* __restrict__ stubs,
                                                   int32_t h, int32_t f,
                                                   int32_t replace, int32_t self_prefix, int32_t ecap, int64_t dcap,
                                                   int32_t tcap, uint32_t pick_mask, FuseArgs fa) {
  // pick_mask: applied to every neighbour entry read from the int32 array / the stubs -- all ones when the degree
  // tags travel on (or the entries carry none), the id mask when they are tagged but this sampler does not use them
  SPP_GROUP_BLOCK(gg);
  __shared__ int32_t lds_scan[2][kNT / kWave + 1];
  // Floyd picks of the row, one column per lane: f rows of kNT ints, sized by the launch (dynamic LDS) --
  // a static [32][kNT] array (32 KB) held the kernel to 20 waves per CU whatever the fanout
  extern __shared__ int32_t chosen_lds[];
  int32_t (*chosen)[kNT] = reinterpret_cast<int32_t (*)[kNT]>(chosen_lds);
  const SlotPtrs& s = slots[gg.first_slot + by_];
  SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* bsum0 = G(s.bsum0);
  const SPP_GLOBAL int32_t* bsum1 = G(s.bsum1);
  SPP_GLOBAL int32_t* out_rp = G(s.out_rowptr[h]);
  const int32_t i = bx_ * kNT + threadIdx.x;
  // ---- round trip 1: everything that does not depend on another load of this kernel, issued together.
  // The per-target arrays are read at a clamped index before T is known (they are sized for tcap).
  const int32_t ic = i < tcap ? i : tcap - 1;
  const int32_t T = st->cnt[h];
  const int32_t err0 = st->error;
  const int64_t dbase = st->dbase[h];
  const uint32_t* rng_gen = st->rng;
  // kHdr: s.deg holds min(degree, 255) (k_hop_count read the byte array) -- enough for the counts below; the exact
  // degree and the row start come out of the row's stub line once it has arrived
  int32_t deg = G(s.deg)[ic];
  int64_t rs = kHdr ? 0 : G(s.rowstart)[ic];
  int32_t vnode = kStub ? G(s.n_ids)[ic] : 0;
  // Offsets of this workgroup's targets: the sums of the workgroups before it.  With self_prefix the
  // workgroup adds up their per-workgroup sums (k_hop_count) itself -- at most a few loads per lane --
  // instead of a single-workgroup scan kernel between the two launches; the last workgroup records the
  // hop's totals.  (Without: bsum0/bsum1 were scanned in place by k_hop_scan.)
  int32_t a0 = 0, a1 = 0;
  if (self_prefix) {
    for (int k = threadIdx.x; k < (int)bx_; k += kNT) {
      a0 += bsum0[k];
      a1 += bsum1[k];
    }
  } else {
    a0 = bsum0[bx_];
    a1 = bsum1[bx_];
  }
  if ((int64_t)bx_ * kNT >= T && bx_ != 0) return;  // workgroup 0 always runs: it records the totals of an empty hop
  int32_t cnt = 0, smp = 0;
  if (i < T) {
    target_counts(deg, f, replace, cnt, smp);
  } else {
    deg = 0;
    rs = 0;
    vnode = 0;
  }
  // ---- round trip 2.  Cooperative neighbour reads: they need only the rows' nodes / starts and lengths,
  // not the picks, so the one long HBM miss of this kernel overlaps the offset sums, the draws and the Floyd steps.
  // One lane fetching its own picks would issue `cnt` scattered 4-byte loads -- `cnt` cache-line requests
  // for a row that spans one or two lines.  Instead 8 lanes read the row's first neighbours as one
  // contiguous 128-byte request: 8 rows per wavefront load instruction, 8 rounds for the wavefront's 64
  // rows.  With row stubs (kStub) the request is the node's stub -- one aligned line holding the first 29
  // neighbours after a 3-word header; without, the first 32 entries of the row in the int32 neighbour
  // array (unaligned: 1.75 lines on average).  (Rows are read even when a capacity error will discard
  // them: harmless.)
  constexpr int kSeg = kStub ? kStubNbr : 32;   // neighbour positions served from the staged line
  constexpr int kSegOff = kStub ? 3 : 0;        // word of the staged line holding position 0
  typedef int32_t i4 __attribute__((ext_vector_type(4)));
  typedef int32_t i4u __attribute__((ext_vector_type(4), aligned(4)));
  constexpr bool kCoop = !kGeneric && (kStub || sizeof(ColT) == 4);
  i4 seg[kCoop ? 8 : 1];
  if constexpr (kCoop) {
    const int lane = threadIdx.x & (kWave - 1), j = lane >> 3, part = lane & 7;
    const int32_t seglen = (deg < kSeg ? deg : kSeg) + (deg > 0 ? kSegOff : 0);  // words wanted; 0 for lanes without a target
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int src = r * 8 + j;
      const int32_t len_r = __shfl(seglen, src, kWave);
      if constexpr (kStub) {
        // unconditional (lanes without a target read node 0's stub): a stub is ONE line whatever the row's
        // length, and a conditional load made the compiler wait for each load before issuing the next
        const int32_t v_r = __shfl(vnode, src, kWave);
        seg[r] = stubs[(int64_t)v_r * 8 + part];
      } else {
        const int64_t rs_r = ((int64_t)__shfl((int32_t)(rs >> 32), src, kWave) << 32) |
                             (uint32_t)__shfl((int32_t)(uint32_t)rs, src, kWave);
        seg[r] = (part * 4 < len_r) ? *reinterpret_cast<const i4u*>(col + rs_r + part * 4) : i4{0, 0, 0, 0};
      }
    }
  }
  int32_t pre0, pre1, tot0, tot1;
  if (self_prefix) {
    block_exclusive_scan<int32_t, kNT>(a0, lds_scan[0], &pre0);
    block_exclusive_scan<int32_t, kNT>(a1, lds_scan[1], &pre1);
    __syncthreads();  // lds_scan is reused below
  } else {
    pre0 = a0;
    pre1 = a1;
  }
  const int32_t p0 = pre0 + block_exclusive_scan<int32_t, kNT>(cnt, lds_scan[0], &tot0);
  const int32_t r0 = pre1 + block_exclusive_scan<int32_t, kNT>(smp, lds_scan[1], &tot1);
  if (self_prefix && threadIdx.x == 0 && (int64_t)(bx_ + 1) * kNT >= T) {  // the last workgroup with targets
    const int32_t E = pre0 + tot0, S = pre1 + tot1;
    st->E[h] = E;
    st->nsmp[h] = S;
    out_rp[T] = E;
    if (E > ecap) atomicOr(&s.st->error, kErrEdgeCap);
    if (dbase + (int64_t)(f > 0 ? f : 0) * S > dcap) atomicOr(&s.st->error, kErrDrawCap);
  }
  // lanes without a target (or past a capacity error) stay until the end: the cooperative row reads
  // below use whole wavefronts
  bool live = i < T;
  if (live && self_prefix && (p0 + cnt > ecap || (smp && dbase + (int64_t)f * (r0 + 1) > dcap))) live = false;
  if (live) out_rp[i] = p0;
  if (err0) live = false;  // an earlier kernel of the chain failed: nothing to do
  if (!live) {
    cnt = 0;
    smp = 0;
    deg = 0;
  }
  const SPP_GLOBAL uint32_t* rng = G(rng_gen);
  if (smp) rng += dbase + (int64_t)f * r0;
  if (kGeneric) {
    // only the Floyd picks are produced here (into evals[p0..p0+f), free at this point);
    // expansion is edge-parallel
    if (smp) {
      int32_t* mine = reinterpret_cast<int32_t*>(s.evals) + p0;
      for (int32_t k = 0; k < f; ++k) {
        if (replace) {
          mine[k] = (int32_t)(rng[k] % (uint32_t)deg);
          continue;
        }
        const int32_t j = deg - f + k;
        const int32_t option = (int32_t)(rng[k] % (uint32_t)j);
        bool found = false;
        for (int32_t m = 0; m < k; ++m) found |= (mine[m] == option);
        mine[k] = found ? j : option;
      }
    }
    return;
  }
  const int tid = threadIdx.x;
  // kHdr: the first eight draws are requested BEFORE the row's header is taken out of its stub line (the first use
  // of the line, i.e. the wait for it): stub line, offset sums and draws then share one round trip again, as they did
  // when the exact degree came from the degree pass
  // (sixteen draws / far picks per round trip in the fused instances -- the small hops' fan-outs of 10 and 15 would then need
  // one round trip each instead of two -- measured nothing in round 6: 120 registers instead of 96, lone chain 34.5-35.0
  // against 34.5-34.6 us per batch)
  constexpr int kB = 8;
  uint32_t rfirst[kB];
  if constexpr (kHdr) {
#pragma unroll
    for (int u = 0; u < kB; ++u) rfirst[u] = rng[u < f ? u : (f > 0 ? f - 1 : 0)];  // (lanes that draw nothing read the stream's start)
  }
  if constexpr (kCoop && kHdr) {
    // header {degree, row start lo, hi} of lane L's row: words 0..2 of the piece lane (L & 7) * 8 holds in round L >> 3
    const int lane = threadIdx.x & (kWave - 1);
    const int src = (lane & 7) * 8;
    int32_t hd = 0, hlo = 0, hhi = 0;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int32_t d = __shfl(seg[r].x, src, kWave), lo = __shfl(seg[r].y, src, kWave), hi = __shfl(seg[r].z, src, kWave);
      if ((lane >> 3) == r) {
        hd = d;
        hlo = lo;
        hhi = hi;
      }
    }
    if (live) {
      deg = hd;
      rs = ((int64_t)hhi << 32) | (uint32_t)hlo;
    }
  }
  if (smp) {
    // Robert Floyd (sample_cpu.hpp:97-110): for j = deg-f .. deg-1: option = gen() % j;
    // winner = option unless already chosen, then j.  The draws of 8 steps are loaded before the first
    // of them is used: the steps depend on each other through `chosen`, the loads do not, and under
    // the delivery kernel's HBM load one global round trip per step was most of this kernel's time.
    for (int32_t k0 = 0; k0 < f; k0 += kB) {
      uint32_t r[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u)  // clamped, not predicated: the loads issue back to back
        r[u] = (kHdr && k0 == 0) ? rfirst[u] : rng[k0 + u < f ? k0 + u : f - 1];
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        const int32_t k = k0 + u;
        if (k >= f) break;
        if (replace) {  // sample_cpu.hpp:79-81
          chosen[k][tid] = (int32_t)(r[u] % (uint32_t)deg);
          continue;
        }
        const int32_t j = deg - f + k;
        const int32_t option = (int32_t)(r[u] % (uint32_t)j);
        bool found = false;
        for (int32_t m = 0; m < k; ++m) found |= (chosen[m][tid] == option);
        chosen[k][tid] = found ? j : option;
      }
    }
  }
  if constexpr (kCoop) {
    // The rows fetched at the top pass through a 1 KB LDS stage per wavefront, and each lane takes its
    // picks from there.  Picks at positions >= 32 (rows of higher degree) are read directly.
    __shared__ i4 stage[kNT / kWave][8][8 + 1];  // +1: rows start in different LDS banks
    const int lane = tid & (kWave - 1), wid = tid / kWave, j = lane >> 3, part = lane & 7;
    // far picks (position >= kSeg): direct reads, batched; the entry replaces the position in `chosen`, and a bit of
    // `farmask` (cnt <= 32) says so -- entries may carry a degree tag in their top bits, so their sign means nothing
    uint32_t farmask = 0u;
    for (int32_t k0 = 0; k0 < cnt; k0 += kB) {
      int32_t nb[kB];
      bool far[kB];
#pragma unroll
      for (int u = 0; u < kB; ++u) {
        const int32_t k = k0 + u;
        const int32_t w = (k < cnt) ? (smp ? chosen[k][tid] : k) : 0;
        far[u] = k < cnt && w >= kSeg;
        nb[u] = (int32_t)col[far[u] ? rs + w : 0];  // not predicated: lanes without a far pick share col[0]'s line
      }
#pragma unroll
      for (int u = 0; u < kB; ++u)
        if (far[u]) {
          chosen[k0 + u][tid] = (int32_t)((uint32_t)nb[u] & pick_mask);
          farmask |= 1u << (k0 + u);
        }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      stage[wid][j][part] = seg[r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if ((lane >> 3) == r) {
        const int32_t* row = reinterpret_cast<const int32_t*>(&stage[wid][lane & 7][0]);
        for (int32_t k = 0; k < cnt; ++k) {
          if ((farmask >> k) & 1u) continue;  // fetched above
          // unsampled rows take every position in order
          const int32_t c = smp ? chosen[k][tid] : k;
          chosen[k][tid] = (int32_t)((uint32_t)row[c + kSegOff] & pick_mask);  // now the neighbour entry
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    // (the same entries staged through LDS in position order and stored by consecutive lanes -- what k_hop_rows_coalesced does
    // for its arrays -- measured no gain here: 98 registers instead of 96 = four waves per SIMD instead of five; r05_ab_INDEX.md)
    SPP_GLOBAL int32_t* cv = G(s.cval) + p0;
    for (int32_t k = 0; k < cnt; ++k) cv[k] = chosen[k][tid];
    if constexpr (kFuse) {
      // ---- k_bucket_scatter's work for this workgroup's run of positions [pre0, pre0 + tot0) (see there) ----
      const int32_t nbk = 1 << fa.cb_log2;
      int32_t* cur = chosen_lds + fa.lds_off;     // [nbk] histogram -> running slot cursor of each bucket
      int32_t* delta = cur + nbk;                 // [nbk] (position inside the bucket's region of the first pair) - (its staging slot)
      uint32_t* snode = reinterpret_cast<uint32_t*>(delta + nbk);              // [tile_cap] node ids in bucket order
      uint16_t* sidx = reinterpret_cast<uint16_t*>(snode + fa.tile_cap);       // [tile_cap] tile-local edge index
      SPP_GLOBAL int32_t* bfill = G(s.bfill);
      SPP_GLOBAL uint32_t* inv = G(s.inv);
      SPP_GLOBAL unsigned long long* bpairs = G(s.bpairs);
      const int32_t li0 = p0 - pre0;              // tile-local index of this lane's first edge (lanes without edges: unused)
      for (int b = tid; b < nbk; b += kNT) cur[b] = 0;
      __syncthreads();
      for (int32_t k = 0; k < cnt; ++k)
        atomicAdd(&cur[bucket_of((uint32_t)chosen[k][tid] & fa.idmask, fa.cb_log2)], 1);
      __syncthreads();
      {
        const int per = (nbk + kNT - 1) / kNT;    // consecutive buckets per thread (<= kMaxBuckets / kNT)
        const int b0 = tid * per;
        int32_t c[kMaxBuckets / kNT], sum = 0;
#pragma unroll
        for (int k = 0; k < kMaxBuckets / kNT; ++k) {
          c[k] = (k < per && b0 + k < nbk) ? cur[b0 + k] : 0;
          sum += c[k];
        }
        int32_t tot;
        int32_t off = block_exclusive_scan<int32_t, kNT>(sum, lds_scan[0], &tot);
#pragma unroll
        for (int k = 0; k < kMaxBuckets / kNT; ++k) {
          if (k < per && b0 + k < nbk) {
            const int32_t g = c[k] ? __hip_atomic_fetch_add(bfill + b0 + k, c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            cur[b0 + k] = off;
            delta[b0 + k] = g - off;
            off += c[k];
          }
        }
      }
      __syncthreads();
      for (int32_t k = 0; k < cnt; ++k) {
        const uint32_t c = (uint32_t)chosen[k][tid] & fa.idmask;
        const uint32_t b = bucket_of(c, fa.cb_log2);
        const int32_t slot = atomicAdd(&cur[b], 1);
        const int32_t li = li0 + k;
        snode[slot] = c;
        sidx[slot] = (uint16_t)li;
        const int32_t rel = slot + delta[b];
        if (rel < fa.bcap) {
          inv[pre0 + li] = (uint32_t)((int32_t)b * fa.bcap + rel);
        } else {  // the bucket's region is full: the overflow list (one atomic per such pair; rare)
          const int32_t j = __hip_atomic_fetch_add(G(s.ovfc) + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          inv[pre0 + li] = (uint32_t)(fa.region + j);
          bpairs[(int64_t)fa.region + j] = ((unsigned long long)c << 32) | (uint32_t)(pre0 + li);
        }
      }
      __syncthreads();
      // the staged pairs: slots [0, n_stage) hold this workgroup's edges in bucket order
      int32_t n_stage = 0;
      {
        const int32_t last = nbk - 1;
        n_stage = cur[last];  // the last bucket's cursor ended at the total
      }
      for (int k = tid; k < n_stage; k += kNT) {
        const uint32_t c = snode[k];
        const int32_t b = (int32_t)bucket_of(c, fa.cb_log2);
        const int32_t rel = k + delta[b];
        if (rel < fa.bcap) bpairs[(int64_t)b * fa.bcap + rel] = ((unsigned long long)c << 32) | (uint32_t)(pre0 + sidx[k]);
      }
    }
  } else {
    // neighbour reads in batches of 8 held in registers: the loads of a batch are all issued before the
    // first use, so a lane has up to 8 independent HBM misses in flight instead of one per iteration
    for (int32_t k0 = 0; k0 < cnt; k0 += 8) {
      int32_t nb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int32_t k = k0 + u;
        if (k < cnt) nb[u] = (int32_t)col[rs + (smp ? chosen[k][tid] : k)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (k0 + u < cnt) G(s.cval)[p0 + k0 + u] = sizeof(ColT) == 4 ? (int32_t)((uint32_t)nb[u] & pick_mask) : nb[u];
    }
  }
}

// generic path: one lane per edge position, row found by binary search in out_rowptr
template <typename ColT>
__global__ __launch_bounds__(kNT) void k_hop_expand_generic(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                             const ColT* __restrict__ col, int32_t h, int32_t f,
                                                             int32_t replace, uint32_t idmask) {
  SPP_GROUP_BLOCK(gg);
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const int32_t T = s.st->cnt[h];
  const int32_t E = s.st->E[h];
  const int64_t p = (int64_t)bx_ * kNT + threadIdx.x;
  if (p >= E) return;
  const int32_t* rp = s.out_rowptr[h];
  int32_t lo = 0, hi = T;  // largest i with rp[i] <= p
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (rp[mid] <= p) lo = mid; else hi = mid;
  }
  const int32_t i = lo;
  const int32_t deg = s.deg[i];
  const int32_t k = (int32_t)p - rp[i];
  int32_t cnt_, smp_;
  target_counts(deg, f, replace, cnt_, smp_);
  const int32_t w = smp_ ? (int32_t)s.evals[p] : k;
  const int32_t c = (int32_t)col[s.rowstart[i] + w];
  s.cval[p] = sizeof(ColT) == 4 ? (int32_t)((uint32_t)c & idmask) : c;  // (the int32 copy may carry degree tags)
}

// ----------------------------------------------------------------------------------------------
// dedup: regroup the hop's edges by bucket, then one workgroup per bucket with an LDS table
// ----------------------------------------------------------------------------------------------
// Regroups a tile of kScatterTile edges by bucket.  Scattered 8-byte stores leave the L2 as one
// 32-byte write EACH (plain stores are written through; stores of different waves are never merged:
// measured 26 B of fabric writes per 8-byte pair), and a million small random writes per batch are what
// slowed the delivery kernel's streaming traffic most.  So the tile is first bucket-sorted in LDS and
// then written out in bucket order: consecutive lanes store consecutive pairs of one bucket's run.
// inv[p] (where edge p went) is stored in position order -- also coalesced -- so that the per-edge
// results of k_bucket_dedup can stay in bucket order and be fetched back by reads (k_hop_flag).
// Round 4: no counting pass ahead of it.  Bucket b owns the fixed region [b * bcap, (b + 1) * bcap) of bpairs; a
// tile reserves room for its run with one atomic on the bucket's fill count; pairs that land past the region's end go
// to the overflow list at [region, ...) one by one (rare: a node reached by hundreds of edges of one hop).
// tiled = 1 (round 5, with k_hop_flag_tiled): inv is written in the tile's STAGING order -- inv[base + k] = where staging
// slot k's pair went, consecutive inside a bucket run -- together with tidx[base + k] = the slot's tile-local position, so that
// the flag pass can fetch the dedup's results as the runs they were written in instead of one random 4-byte word per position.
__global__ __launch_bounds__(kTileNT) void k_bucket_scatter(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                         int32_t h, int32_t cb_log2, int32_t bcap, int32_t region,
                                                         int64_t pcap, uint32_t idmask, int32_t tiled) {
  SPP_GROUP_BLOCK(gg);
  extern __shared__ int32_t sc_lds[];
  const int32_t nbk = 1 << cb_log2;
  int32_t* cur = sc_lds;                // [nbk] tile histogram -> running slot cursor of each bucket
  int32_t* delta = sc_lds + nbk;        // [nbk] (position inside the bucket's region of the tile's first pair) - (its staging slot)
  uint32_t* snode = reinterpret_cast<uint32_t*>(sc_lds + 2 * nbk);               // [kScatterTile] node ids in bucket order
  uint16_t* sidx = reinterpret_cast<uint16_t*>(sc_lds + 2 * nbk + kScatterTile);  // [kScatterTile] tile-local edge index
  __shared__ int32_t lscan[kTileNT / kWave + 1];
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* cval = G(s.cval);
  SPP_GLOBAL int32_t* bfill = G(s.bfill);
  SPP_GLOBAL uint32_t* inv = G(s.inv);
  SPP_GLOBAL unsigned long long* bpairs = G(s.bpairs);
  const int64_t base = (int64_t)bx_ * kScatterTile;
  // one round trip: the state words and the tile's node ids (index clamped: E is not known yet)
  const int32_t E = st->E[h];
  const int32_t err0 = st->error;
  int32_t v[kScatterEPT];
#pragma unroll
  for (int u = 0; u < kScatterEPT; ++u) {
    const int64_t p = base + u * kTileNT + threadIdx.x;
    v[u] = cval[p < pcap ? p : pcap - 1];
  }
  if (base >= E || err0) return;
  bool ok[kScatterEPT];
#pragma unroll
  for (int u = 0; u < kScatterEPT; ++u) {
    ok[u] = base + u * kTileNT + threadIdx.x < E;
    v[u] = (int32_t)((uint32_t)v[u] & idmask);  // the node id without its degree tag
  }
  for (int b = threadIdx.x; b < nbk; b += kTileNT) cur[b] = 0;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kScatterEPT; ++u)
    if (ok[u]) atomicAdd(&cur[bucket_of((uint32_t)v[u], cb_log2)], 1);
  __syncthreads();
  // exclusive scan of the tile histogram (first slot of every bucket in the LDS staging order) and
  // one global reservation per non-empty bucket
  {
    const int per = (nbk + kTileNT - 1) / kTileNT;  // consecutive buckets per thread (1..4)
    const int b0 = threadIdx.x * per;
    int32_t c[kMaxBuckets / kTileNT], sum = 0;
#pragma unroll
    for (int k = 0; k < kMaxBuckets / kTileNT; ++k) {
      c[k] = (k < per && b0 + k < nbk) ? cur[b0 + k] : 0;
      sum += c[k];
    }
    int32_t tot;
    int32_t off = block_exclusive_scan<int32_t, kTileNT>(sum, lscan, &tot);
#pragma unroll
    for (int k = 0; k < kMaxBuckets / kTileNT; ++k) {
      if (k < per && b0 + k < nbk) {  // (usually per == 1: one reservation per thread)
        const int32_t g = c[k] ? __hip_atomic_fetch_add(bfill + b0 + k, c[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        cur[b0 + k] = off;
        delta[b0 + k] = g - off;
        off += c[k];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kScatterEPT; ++u) {
    if (!ok[u]) continue;
    const uint32_t c = (uint32_t)v[u];
    const uint32_t b = bucket_of(c, cb_log2);
    const int32_t slot = atomicAdd(&cur[b], 1);  // order inside a bucket is irrelevant (min / max are commutative)
    const int32_t li = u * kTileNT + threadIdx.x;
    snode[slot] = c;
    sidx[slot] = (uint16_t)li;
    const int32_t rel = slot + delta[b];  // position inside bucket b's region
    if (rel < bcap) {
      if (!tiled) inv[base + li] = (uint32_t)((int32_t)b * bcap + rel);
    } else {  // the bucket's region is full: the overflow list (one atomic per such pair; rare)
      const int32_t j = __hip_atomic_fetch_add(G(s.ovfc) + h, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      inv[base + (tiled ? slot : li)] = (uint32_t)(region + j);
      bpairs[(int64_t)region + j] = ((unsigned long long)c << 32) | (uint32_t)(base + li);
    }
  }
  __syncthreads();
  const int32_t n_tile = (int32_t)((E - base) < kScatterTile ? (E - base) : kScatterTile);
  SPP_GLOBAL uint16_t* tidx = G(s.tidx);
  for (int k = threadIdx.x; k < n_tile; k += kTileNT) {
    const uint32_t c = snode[k];
    const int32_t b = (int32_t)bucket_of(c, cb_log2);
    const int32_t rel = k + delta[b];
    if (rel < bcap) {
      bpairs[(int64_t)b * bcap + rel] = ((unsigned long long)c << 32) | (uint32_t)(base + sidx[k]);
      if (tiled) inv[base + k] = (uint32_t)(b * bcap + rel);
    }
    if (tiled) tidx[base + k] = sidx[k];
  }
}

// exclusive rank of edge position q among the hop's first occurrences (q must be one, or any position
// when only "first occurrences before q" is wanted)
__device__ __forceinline__ int32_t first_rank(const SlotPtrs& s, uint32_t q) {
  const uint32_t w = q >> 6;
  const RankWord rw = s.fwords[w];
  return s.fsum[q >> 8] + (int32_t)rw.pre + __popcll(rw.bits & ((1ull << (q & 63)) - 1ull));
}

// One workgroup per bucket: known nodes (ids from earlier hops; pending ones of the previous hop are
// resolved through its rank array first) and this hop's candidates meet in an LDS table.
//   value < T          : final local id of an already known node
//   value = T + p      : p is the earliest edge position of this hop reaching the node
// Every edge gets the final value of its node (evals[p]); the earliest edge of every new node appends
// (node, kPending | p) to the bucket's known list for the later hops.
// (amdgpu_waves_per_eu(6): 80 registers, six workgroups of four wavefronts per compute unit -- what the 24 KB table admits;
// at 81 registers the kernel drops to five and the hop's launch takes 10-60 % longer)
// amdgpu_waves_per_eu(6) for the small tables: 80 registers = six workgroups of four wavefronts per compute unit, what
// a 24 KB table admits; the allocator lands on 81 otherwise (five: the hop's launch then takes 10-60 % longer) and
// meets 80 with one 8-byte spill in the prologue.
// kFlat (round 6; launched when a hop's bucket spans MORE than four fine buckets: the small hops): the fine known lists are
// walked as ONE concatenation -- entry j of it belongs to the list whose prefix interval holds j -- instead of one wavefront
// per list, list after list: with 64 fine lists per bucket (hop 0 at papers scale) that was 16 lists per wavefront x two
// dependent round trips each (entries, then their rank records) = 32 round trips for a handful of entries; now three.
template <int NS, bool kFlat = false>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(NS <= 3072 ? 6 : 1))) void k_bucket_dedup(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                       int32_t hop_word, DedupGeom g, int32_t bcap, int32_t region) {
  SPP_GROUP_BLOCK(gg);
  // hop_word = h | cb_log2 << 8 | hop_flags << 16 (one scalar register instead of three: the kernel is at the edge of
  // both register files, and a spilled scalar costs a vector register -- 81 of them are five waves per SIMD instead of six)
  const int32_t h = hop_word & 0xff, cb_log2 = (hop_word >> 8) & 0xff, hop_flags = hop_word >> 16;
  const int32_t last_hop = hop_flags & 1;    // no later hop: the known lists are not extended
  const bool no_preread = hop_flags & 2;     // candidates go straight to the compare-and-swap (mostly new keys)
  __shared__ unsigned long long tab[NS];
  __shared__ int32_t fkc[kMaxFinePerCoarse];   // entries of each fine known list at entry
  __shared__ int32_t fnew[kMaxFinePerCoarse];  // nodes this hop appends to each
  __shared__ int ovf;
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL unsigned long long* bpairs = G(s.bpairs);
  SPP_GLOBAL unsigned long long* known = G(s.known);
  SPP_GLOBAL int32_t* kcount = G(s.kcount);
  const SPP_GLOBAL RankWord* fwords = G(s.fwords);
  const SPP_GLOBAL int32_t* fsum = G(s.fsum);
  SPP_GLOBAL uint32_t* res = G(s.res);
  // A hop's bucket is a run of 2^shift consecutive fine buckets (bucket ids are top bits of one hash):
  // small hops use few, well-filled workgroups instead of thousands that each set up an LDS table
  // for a handful of edges; the known-node lists stay per FINE bucket for the later, larger hops.
  const int32_t b = bx_;
  const int32_t shift = g.nb_log2 - cb_log2;
  const int32_t nf = 1 << shift;
  const int32_t fb0 = b << shift;
  // ---- round trip 1: the state words, the bucket's bounds and the fine lists' lengths, issued together
  const int32_t err0 = st->error;
  const uint32_t T = (uint32_t)st->cnt[h];
  const uint32_t Tprev = h > 0 ? (uint32_t)st->cnt[h - 1] : 0u;
  // the bucket's pairs: the first min(fill, bcap) entries of its region, plus -- when it overflowed -- its share of
  // the hop's overflow list (walked by this workgroup alone)
  // (workgroup-uniform values loaded through per-slot pointers arrive in vector registers: moved to scalar ones, the
  // kernel sits exactly at the 80 registers six waves per SIMD allow)
  const int32_t fill = __builtin_amdgcn_readfirstlane(G(s.bfill)[b]);
  const int32_t e0 = b * bcap, e1 = e0 + (fill < bcap ? fill : bcap);
  const bool spill = fill > bcap;  // block-uniform
  for (int i = threadIdx.x; i < nf; i += kNT) {
    fkc[i] = kcount[fb0 + i];
    fnew[i] = 0;
  }
  // (speculative, same round trip as the pairs below: the first 4 x 64 entries of this wavefront's first
  // fine list -- the list's length is not known yet, entries past it are dropped; kcap >= 256)
  const int lane = threadIdx.x & (kWave - 1);
  unsigned long long kpre[4];
  if constexpr (!kFlat) {
    const int lf0 = threadIdx.x / kWave;
    const SPP_GLOBAL unsigned long long* kl0 = known + (int64_t)(fb0 + (lf0 < nf ? lf0 : 0)) * g.kcap;
#pragma unroll
    for (int u = 0; u < 4; ++u) kpre[u] = kl0[lane + u * kWave];
  }
  if (err0) return;
  const bool work = e1 > e0;  // block-uniform
  // ---- round trip 2.  This hop's candidates: the first kDedupRegs * kNT pairs of the bucket stay in
  // registers between the insert pass and the lookup pass (a bucket holds ~1k edges: usually all of them).
  // Index clamped instead of a predicated load, so that the loads issue back to back.
  // (Round 5 tried these loads in round trip 1 -- a bucket's region starts at b * bcap whatever it holds: one dependent
  // round trip less per workgroup, no difference in the lone chain or the pipeline, and 4 MB more fetched per batch from
  // the unfilled half of every region; profiles/r05_ab_INDEX.md.)
  unsigned long long pr[kDedupRegs];
#pragma unroll
  for (int u = 0; u < kDedupRegs; ++u) {
    const int i = e0 + u * kNT + threadIdx.x;
    const unsigned long long v = bpairs[i < e1 ? i : 0];  // (an empty bucket has e1 == e0 <= i)
    pr[u] = i < e1 ? v : kEmptySlot;
  }
  if (work)
    for (int i = threadIdx.x; i < NS; i += kNT) tab[i] = kEmptySlot;
  if (threadIdx.x == 0) ovf = 0;
  __syncthreads();
  // known nodes: resolve the previous hop's pending ids (its first-occurrence bitmap is overwritten by
  // this hop's k_hop_flag, so this must happen now for EVERY list), then publish them in the LDS table
  if constexpr (kFlat) {
    __shared__ int32_t fpre[kMaxFinePerCoarse + 1];  // exclusive prefix of the lists' lengths
    if (threadIdx.x < kWave) {                       // nf <= 64: one wavefront scans
      const int32_t c = (int)threadIdx.x < nf ? fkc[threadIdx.x] : 0;
      const int32_t inc = wave_inclusive_scan(c);
      if ((int)threadIdx.x < nf) fpre[threadIdx.x + 1] = inc;
      if (threadIdx.x == 0) fpre[0] = 0;
    }
    __syncthreads();
    const int32_t K = fpre[nf];
    for (int j0 = threadIdx.x; j0 < K; j0 += 4 * kNT) {  // 4 entries per thread and round, loads batched
      unsigned long long e[4];
      uint32_t q[4];
      RankWord rw[4];
      int32_t fs[4];
      SPP_GLOBAL unsigned long long* at[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int32_t j = j0 + u * kNT;
        const int32_t jc = j < K ? j : K - 1;            // clamped, not predicated
        int32_t lo = 0, hi = nf;                         // the list whose interval [fpre[lf], fpre[lf + 1]) holds jc
        while (hi - lo > 1) {
          const int32_t mid = (lo + hi) >> 1;
          if (fpre[mid] <= jc) lo = mid; else hi = mid;
        }
        at[u] = known + (int64_t)(fb0 + lo) * g.kcap + (jc - fpre[lo]);
        const unsigned long long v = *at[u];
        e[u] = j < K ? v : kEmptySlot;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t val = (uint32_t)e[u];
        const bool pend = e[u] != kEmptySlot && (val & kPending);
        q[u] = pend ? (val & ~kPending) : 0u;
        rw[u] = load_rank_word(fwords, q[u] >> 6);
        fs[u] = fsum[q[u] >> 8];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (e[u] == kEmptySlot) continue;
        uint32_t val = (uint32_t)e[u];
        if (val & kPending) {
          val = Tprev + (uint32_t)(fs[u] + (int32_t)rw[u].pre + __popcll(rw[u].bits & ((1ull << (q[u] & 63)) - 1ull)));
          e[u] = (e[u] & 0xffffffff00000000ull) | val;
          *at[u] = e[u];
        }
        if (work) lds_upsert<true, NS, false>(tab, (uint32_t)(e[u] >> 32), val, &ovf);
      }
    }
  } else
  // one wavefront per fine list
  for (int lf = threadIdx.x / kWave; lf < nf; lf += kNT / kWave) {
    SPP_GLOBAL unsigned long long* kl = known + (int64_t)(fb0 + lf) * g.kcap;
    const int32_t kc = fkc[lf];
    const bool first_list = lf == (int)(threadIdx.x / kWave);
    for (int i0 = lane; i0 < kc; i0 += 4 * kWave) {  // 4 entries per lane and round, loads batched
      unsigned long long e[4];
      uint32_t q[4];
      RankWord rw[4];
      int32_t fs[4];
      if (first_list && i0 == lane) {  // fetched with the pairs
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = (i0 + u * kWave < kc) ? kpre[u] : kEmptySlot;
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int i = i0 + u * kWave;
          const unsigned long long v = kl[i < kc ? i : 0];  // kc > 0 here
          e[u] = i < kc ? v : kEmptySlot;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {  // rank records of the pending entries (others read record 0: no branch)
        const uint32_t val = (uint32_t)e[u];
        const bool pend = e[u] != kEmptySlot && (val & kPending);
        q[u] = pend ? (val & ~kPending) : 0u;
        rw[u] = load_rank_word(fwords, q[u] >> 6);
        fs[u] = fsum[q[u] >> 8];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (e[u] == kEmptySlot) continue;
        uint32_t val = (uint32_t)e[u];
        if (val & kPending) {
          val = Tprev + (uint32_t)(fs[u] + (int32_t)rw[u].pre + __popcll(rw[u].bits & ((1ull << (q[u] & 63)) - 1ull)));
          e[u] = (e[u] & 0xffffffff00000000ull) | val;
          kl[i0 + u * kWave] = e[u];
        }
        if (work) lds_upsert<true, NS, false>(tab, (uint32_t)(e[u] >> 32), val, &ovf);  // known nodes are distinct keys: no pre-read
      }
    }
  }
  if (!work) return;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kDedupRegs; ++u)
    if (pr[u] != kEmptySlot) {
      if (no_preread) lds_upsert<false, NS, false>(tab, (uint32_t)(pr[u] >> 32), T + (uint32_t)pr[u], &ovf);
      else lds_upsert<false, NS>(tab, (uint32_t)(pr[u] >> 32), T + (uint32_t)pr[u], &ovf);
    }
  for (int i0 = e0 + kDedupRegs * kNT + threadIdx.x; i0 < e1; i0 += 4 * kNT) {  // oversized bucket: the rest
    unsigned long long q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * kNT;
      const unsigned long long v = bpairs[i < e1 ? i : e0];
      q[u] = i < e1 ? v : kEmptySlot;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (q[u] != kEmptySlot) lds_upsert<false, NS, false>(tab, (uint32_t)(q[u] >> 32), T + (uint32_t)q[u], &ovf);
  }
  if (spill) {  // this bucket's pairs in the overflow list
    const int32_t n_ovf = G(s.ovfc)[h];
    for (int i = threadIdx.x; i < n_ovf; i += kNT) {
      const unsigned long long q = bpairs[(int64_t)region + i];
      if ((int32_t)bucket_of((uint32_t)(q >> 32), cb_log2) == b) lds_upsert<false, NS, false>(tab, (uint32_t)(q >> 32), T + (uint32_t)q, &ovf);
    }
  }
  __syncthreads();
  if (ovf) {
    if (threadIdx.x == 0) atomicOr(&s.st->error, kErrBucketCap);
    return;
  }
  auto resolve = [&](unsigned long long pair, int i) {
    const uint32_t key = (uint32_t)(pair >> 32), p = (uint32_t)pair;
    const uint32_t val = lds_find<NS>(tab, key);
    res[i] = val;  // bucket order: consecutive lanes, consecutive words (k_hop_flag brings it to position order)
    if (val == T + p && !last_hop) {  // first occurrence of a new node: append to its fine list (no later hop: skip)
      const int32_t lf = (int32_t)bucket_of(key, g.nb_log2) - fb0;
      const int j = fkc[lf] + atomicAdd(&fnew[lf], 1);
      if (j < g.kcap) known[(int64_t)(fb0 + lf) * g.kcap + j] = ((unsigned long long)key << 32) | kPending | p;
      else ovf = 1;
    }
  };
#pragma unroll
  for (int u = 0; u < kDedupRegs; ++u)
    if (pr[u] != kEmptySlot) resolve(pr[u], e0 + u * kNT + threadIdx.x);
  for (int i = e0 + kDedupRegs * kNT + threadIdx.x; i < e1; i += kNT) resolve(bpairs[i], i);
  if (spill) {
    const int32_t n_ovf = G(s.ovfc)[h];
    for (int i = threadIdx.x; i < n_ovf; i += kNT) {
      const unsigned long long q = bpairs[(int64_t)region + i];
      if ((int32_t)bucket_of((uint32_t)(q >> 32), cb_log2) == b) resolve(q, region + i);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) G(s.bfill)[b] = 0;  // the next hop's scatter counts from zero again
  if (threadIdx.x == 0 && ovf) atomicOr(&s.st->error, kErrBucketCap);
  for (int i = threadIdx.x; i < nf; i += kNT) {
    const int32_t room = g.kcap - fkc[i];
    kcount[fb0 + i] = fkc[i] + (fnew[i] < room ? fnew[i] : room);
  }
}

// ----------------------------------------------------------------------------------------------
// first-occurrence ranking
// ----------------------------------------------------------------------------------------------
constexpr int kFlagNT = kNT;                    // 256 threads: a 1024-thread workgroup needs 16 free wave slots on ONE CU at
                                                // once and waited for them behind the delivery kernel (36 us per batch in situ, 12 alone)
#ifndef SPP_FLAG_ROUNDS
#define SPP_FLAG_ROUNDS 4
#endif
constexpr int kFlagRounds = SPP_FLAG_ROUNDS;    // positions per thread; one round = one 256-position block = 4 bitmap words
constexpr int kFlagSpan = kFlagNT * kFlagRounds;  // positions per workgroup (one ticket each)
static_assert(kFlagNT == 256, "a round of the workgroup is one rank block");

// scan_block_sums for sums that other workgroups of the SAME launch stored with agent-scope stores,
// by a kFlagNT-thread workgroup
__device__ int32_t scan_block_sums_acquire(int32_t* a, int32_t n, int32_t* lds) {
  int32_t carry = 0;
  for (int32_t base = 0; base < n; base += kFlagNT) {
    const int32_t i = base + threadIdx.x;
    const int32_t v = (i < n) ? acquire_i32(&a[i]) : 0;
    int32_t tot;
    const int32_t ex = block_exclusive_scan<int32_t, kFlagNT>(v, lds, &tot);
    if (i < n) a[i] = carry + ex;
    carry += tot;
    __syncthreads();
  }
  return carry;
}

// Per edge position p: the table value of its node back in position order (reads of res through inv:
// a 4-byte random READ of a just-written array is served by L2 / Infinity Cache, a 4-byte random
// write would cost a 32-byte HBM write), the first-occurrence bitmap and its per-word / per-block
// counts.  The workgroup that finishes last (device-scope ticket) turns the block counts into the
// exclusive prefix and records the hop's node count -- no separate scan launch.
__global__ __launch_bounds__(kFlagNT) void k_hop_flag(const SlotPtrs* __restrict__ slots, GroupGrid gg, int32_t h,
                                                       int32_t f, int32_t ucap, int64_t pcap) {
  SPP_GROUP_BLOCK(gg);
  __shared__ int32_t wcnt[kFlagRounds][kFlagNT / kWave];
  __shared__ int32_t lscan[kFlagNT / kWave + 1];
  __shared__ int is_last;
  const SlotPtrs& s = slots[gg.first_slot + by_];
  SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL uint32_t* inv = G(s.inv);
  const SPP_GLOBAL uint32_t* res = G(s.res);
  SPP_GLOBAL uint32_t* evals = G(s.evals);
  const int64_t base = (int64_t)bx_ * kFlagSpan;
  // ---- round trip 1: state words and this workgroup's slice of inv (index clamped: E is not known yet)
  const int32_t err0 = st->error;
  const int32_t E = st->E[h];
  const uint32_t T = (uint32_t)st->cnt[h];
  uint32_t slot[kFlagRounds];
#pragma unroll
  for (int r = 0; r < kFlagRounds; ++r) {
    const int64_t p = base + r * kFlagNT + threadIdx.x;
    slot[r] = inv[p < pcap ? p : pcap - 1];
  }
  if (err0) {
    if (bx_ == 0 && threadIdx.x == 0) {  // keep the later hops' sizes defined
      st->cnt[h + 1] = (int32_t)T;
      st->dbase[h + 1] = st->dbase[h];
    }
    return;
  }
  const int32_t nwg = E > 0 ? (E + kFlagSpan - 1) / kFlagSpan : 1;  // workgroup 0 always takes part
  if ((int32_t)bx_ >= nwg) return;
  const int wid = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  // ---- round trip 2: the table values, through inv (positions past E read entry 0: no predicate on the load)
  uint32_t val[kFlagRounds];
#pragma unroll
  for (int r = 0; r < kFlagRounds; ++r) {
    const int64_t p = base + r * kFlagNT + threadIdx.x;
    val[r] = res[p < E ? slot[r] : 0u];
  }
  unsigned long long bits[kFlagRounds];
#pragma unroll
  for (int r = 0; r < kFlagRounds; ++r) {
    const int64_t p = base + r * kFlagNT + threadIdx.x;
    bool flag = false;
    if (p < E) {
      evals[p] = val[r];
      flag = (val[r] == T + (uint32_t)p);  // first occurrence of a node that is new in this hop
    }
    bits[r] = __ballot(flag);
    if (lane == 0) wcnt[r][wid] = __popcll(bits[r]);
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < kFlagRounds; ++r) {  // round r = rank block r of this workgroup, wave wid = its word wid
      int32_t pre = 0;
      for (int k = 0; k < wid; ++k) pre += wcnt[r][k];
      const int64_t blk = (int64_t)bx_ * kFlagRounds + r;
      s.fwords[blk * (kFlagNT / kWave) + wid] = RankWord{bits[r], (uint32_t)pre, 0u};
      if (wid == kFlagNT / kWave - 1)  // block total, published for the workgroup that will scan
        __hip_atomic_store(&s.fsum[blk], pre + wcnt[r][wid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its own stores before the barrier
  __syncthreads();
  if (threadIdx.x == 0) is_last = take_ticket_is_last(s.ctr, nwg) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  const int32_t nblk = (E + 255) / 256;
  const int32_t nnew = scan_block_sums_acquire(s.fsum, nblk, lscan);
  if (threadIdx.x == 0) {
    const int32_t U = st->cnt[h] + nnew;
    st->cnt[h + 1] = U;
    st->dbase[h + 1] = st->dbase[h] + (int64_t)(f > 0 ? f : 0) * st->nsmp[h];
    if (U > ucap) atomicOr(&s.st->error, kErrNodeCap);
  }
}

// k_hop_flag over the scatter's 8 k-edge tiles (round 5).  The dedup leaves its results in BUCKET order; k_hop_flag fetches them
// back as res[inv[p]] -- one random 4-byte word per edge position, 841 k of them per batch in the last hop.  But a tile's pairs
// of one bucket sit next to each other in res, and the scatter knows the order it staged them in: with `tiled` it writes inv in
// that staging order (and the slot's tile-local position, 2 bytes), so here consecutive lanes walk consecutive staging slots and
// read res as the runs it was written in (8 pairs = one 32-byte sector per bucket and tile at papers scale); the values pass
// through LDS to position order, and everything after that is k_hop_flag's: evals, the first-occurrence bitmap, the rank records
// and block counts, the last workgroup's scan.  One 256-thread workgroup per tile = 32 rank blocks of 256 positions.
constexpr int kFlagTileRounds = kScatterTile / kFlagNT;  // positions per thread
__global__ __launch_bounds__(kFlagNT) void k_hop_flag_tiled(const SlotPtrs* __restrict__ slots, GroupGrid gg, int32_t h,
                                                             int32_t f, int32_t ucap, int64_t pcap) {
  SPP_GROUP_BLOCK(gg);
  __shared__ uint32_t vals[kScatterTile];
  __shared__ unsigned long long wbits[kFlagTileRounds * (kFlagNT / kWave)];
  __shared__ int32_t wcnt[kFlagTileRounds * (kFlagNT / kWave)];
  __shared__ int32_t lscan[kFlagNT / kWave + 1];
  static_assert(kFlagTileRounds * (kFlagNT / kWave) <= kFlagNT, "one thread per bitmap word of the tile");
  __shared__ int is_last;
  const SlotPtrs& s = slots[gg.first_slot + by_];
  SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL uint32_t* inv = G(s.inv);
  const SPP_GLOBAL uint16_t* tidx = G(s.tidx);
  const SPP_GLOBAL uint32_t* res = G(s.res);
  SPP_GLOBAL uint32_t* evals = G(s.evals);
  const int64_t base = (int64_t)bx_ * kScatterTile;
  const int32_t err0 = st->error;
  const int32_t E = st->E[h];
  const uint32_t T = (uint32_t)st->cnt[h];
  if (err0) {
    if (bx_ == 0 && threadIdx.x == 0) {  // keep the later hops' sizes defined
      st->cnt[h + 1] = (int32_t)T;
      st->dbase[h + 1] = st->dbase[h];
    }
    return;
  }
  const int32_t nwg = E > 0 ? (E + kScatterTile - 1) / kScatterTile : 1;  // workgroup 0 always takes part
  if ((int32_t)bx_ >= nwg) return;
  const int32_t n_tile = (int32_t)((E - base) < kScatterTile ? (E - base) : kScatterTile);
  const int wid = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  // ---- staging order: where each slot's result is, then the result, in rounds of 8 slots per thread (loads batched)
  for (int32_t k0 = 0; k0 < n_tile; k0 += 8 * kFlagNT) {
    uint32_t loc[8], v[8];
    uint16_t pl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int32_t k = k0 + u * kFlagNT + (int32_t)threadIdx.x;
      const int64_t a = base + (k < n_tile ? k : 0);
      loc[u] = inv[a];
      pl[u] = tidx[a];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = res[loc[u]];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k0 + u * kFlagNT + (int32_t)threadIdx.x < n_tile) vals[pl[u]] = v[u];
  }
  __syncthreads();
  // ---- position order: round r = rank block r of the tile (256 positions = 4 bitmap words, one per wavefront)
  for (int r = 0; r < kFlagTileRounds; ++r) {
    const int64_t p = base + r * kFlagNT + threadIdx.x;
    bool flag = false;
    if (p < E) {
      const uint32_t val = vals[r * kFlagNT + threadIdx.x];
      evals[p] = val;
      flag = (val == T + (uint32_t)p);  // first occurrence of a node that is new in this hop
    }
    const unsigned long long bits = __ballot(flag);
    if (lane == 0) {
      wbits[r * (kFlagNT / kWave) + wid] = bits;
      wcnt[r * (kFlagNT / kWave) + wid] = __popcll(bits);
    }
  }
  __syncthreads();
  if (threadIdx.x < kFlagTileRounds * (kFlagNT / kWave)) {  // one thread per bitmap word of the tile
    const int w = threadIdx.x, blk_l = w / (kFlagNT / kWave), first = blk_l * (kFlagNT / kWave);
    if (base + (int64_t)blk_l * kFlagNT < E || blk_l == 0) {
      int32_t pre = 0;
      for (int k = first; k < w; ++k) pre += wcnt[k];
      const int64_t blk = (int64_t)bx_ * kFlagTileRounds + blk_l;
      s.fwords[blk * (kFlagNT / kWave) + (w - first)] = RankWord{wbits[w], (uint32_t)pre, 0u};
      if (w - first == kFlagNT / kWave - 1)  // block total, published for the workgroup that will scan
        __hip_atomic_store(&s.fsum[blk], pre + wcnt[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its own stores before the barrier
  __syncthreads();
  if (threadIdx.x == 0) is_last = take_ticket_is_last(s.ctr, nwg) ? 1 : 0;
  __syncthreads();
  if (!is_last) return;
  const int32_t nblk = (E + 255) / 256;
  const int32_t nnew = scan_block_sums_acquire(s.fsum, nblk, lscan);
  if (threadIdx.x == 0) {
    const int32_t U = st->cnt[h] + nnew;
    st->cnt[h + 1] = U;
    st->dbase[h + 1] = st->dbase[h] + (int64_t)(f > 0 ? f : 0) * st->nsmp[h];
    if (U > ucap) atomicOr(&s.st->error, kErrNodeCap);
  }
}

// local id behind a table value: final ids are below T; T + q names the node first reached at edge position q
__device__ __forceinline__ int32_t local_id_of(const SlotPtrs& s, uint32_t T, uint32_t v) {
  return (v < T) ? (int32_t)v : (int32_t)T + first_rank(s, v - T);
}

// fast path: one lane per target row.  Local ids of the row's edges (rank lookups for the nodes that
// are new in this hop), n_ids.push_back for the row's first occurrences (sample_cpu.hpp:50-60), rank
// sort of the <= 32 ids staged in LDS (sample_cpu.hpp:126).
__global__ __launch_bounds__(kNT) void k_hop_rows(const SlotPtrs* __restrict__ slots, GroupGrid gg, int32_t h,
                                                   uint32_t idmask, int32_t idbits, int32_t tcap, int64_t pcap) {
  SPP_GROUP_BLOCK(gg);
  extern __shared__ int32_t rows_lds[];  // [f][kNT]: the row's local ids, one column per lane (dynamic LDS)
  int32_t (*a)[kNT] = reinterpret_cast<int32_t (*)[kNT]>(rows_lds);
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* rp = G(s.out_rowptr[h]);
  const SPP_GLOBAL uint32_t* evals = G(s.evals);
  const SPP_GLOBAL int32_t* cval = G(s.cval);
  const SPP_GLOBAL RankWord* fwords = G(s.fwords);
  const SPP_GLOBAL int32_t* fsum = G(s.fsum);
  SPP_GLOBAL int32_t* n_ids = G(s.n_ids);
  SPP_GLOBAL uint8_t* dtag = G(s.dtag);
  const int32_t i = bx_ * kNT + threadIdx.x;
  // ---- round trip 1: state words and the row's bounds together (index clamped: T is not known yet; the row
  // pointer array holds tcap + 1 entries)
  const int32_t ic = i < tcap ? i : tcap - 1;
  const int32_t T = st->cnt[h];
  const int32_t err0 = st->error;
  const int32_t p0 = rp[ic];
  const int32_t p1 = rp[ic + 1];
  if (i >= T || err0) return;
  const int tid = threadIdx.x;
  const int32_t n = p1 - p0;
  // 8 edges at a time; each round's loads are all issued before any of them is used (clamped indices, not
  // predicated loads: a predicate makes the compiler wait for each load before it issues the next)
  for (int32_t k0 = 0; k0 < n; k0 += 8) {
    uint32_t v[8], q[8];
    int32_t c[8], fs[8];
    RankWord rw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // ---- round trip 2: table values and neighbour entries of the row
      const int64_t p = (k0 + u < n) ? (int64_t)p0 + k0 + u : (int64_t)p0;
      const int64_t pc = p < pcap ? p : pcap - 1;
      v[u] = evals[pc];
      c[u] = cval[pc];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {  // ---- round trip 3: rank records of the nodes that are new in this hop (others: record 0)
      const bool fresh = k0 + u < n && v[u] >= (uint32_t)T;
      q[u] = fresh ? v[u] - (uint32_t)T : 0u;
      fs[u] = fsum[q[u] >> 8];
      rw[u] = load_rank_word(fwords, q[u] >> 6);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k0 + u >= n) break;
      int32_t id = (int32_t)v[u];
      if (v[u] >= (uint32_t)T) {
        id = T + fs[u] + (int32_t)rw[u].pre + __popcll(rw[u].bits & ((1ull << (q[u] & 63)) - 1ull));
        if (q[u] == (uint32_t)(p0 + k0 + u)) {  // n_ids.push_back(c) at its first occurrence
          n_ids[id] = (int32_t)((uint32_t)c[u] & idmask);
          if (idbits < 32) dtag[id] = (uint8_t)((uint32_t)c[u] >> idbits);  // the node's degree tag travels with it
        }
      }
      a[k0 + u][tid] = id;
    }
  }
  SPP_GLOBAL int32_t* out = G(s.out_col[h]) + p0;
  for (int32_t k = 0; k < n; ++k) {
    const int32_t v = a[k][tid];
    int32_t rank = 0;
    for (int32_t j = 0; j < n; ++j) {
      const int32_t w = a[j][tid];
      rank += (w < v || (w == v && j < k)) ? 1 : 0;
    }
    out[rank] = v;  // std::sort of the row's local ids (sample_cpu.hpp:126)
  }
}

// k_hop_rows with its per-edge arrays moved through LDS in POSITION order (round 5).  A workgroup's 256 rows cover one
// contiguous run of edge positions [P0, P1); a lane reading its own row straight from HBM issues `f` loads whose 64
// lanes are 4 f bytes apart -- every load instruction touches 64 f / 16 sectors for 256 useful bytes -- and the sorted
// row is written back the same way.  Here the run's table values and neighbour entries are fetched by consecutive lanes
// (consecutive addresses), a lane takes its row out of LDS, the row's sorted local ids replace its table values in place
// (rows are disjoint), and the run goes out coalesced.  LDS: 8 bytes per edge of the run (<= 256 f edges).
__global__ __launch_bounds__(kNT) void k_hop_rows_coalesced(const SlotPtrs* __restrict__ slots, GroupGrid gg, int32_t h,
                                                             uint32_t idmask, int32_t idbits, int32_t tcap, int64_t pcap,
                                                             int32_t run_cap) {
  SPP_GROUP_BLOCK(gg);
  extern __shared__ int32_t rows_lds[];          // [run_cap] table values -> sorted local ids, [run_cap] neighbour entries
  uint32_t* lv = reinterpret_cast<uint32_t*>(rows_lds);
  int32_t* lc = rows_lds + run_cap;
  __shared__ int32_t run_lo, run_hi;
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* rp = G(s.out_rowptr[h]);
  const SPP_GLOBAL uint32_t* evals = G(s.evals);
  const SPP_GLOBAL int32_t* cval = G(s.cval);
  const SPP_GLOBAL RankWord* fwords = G(s.fwords);
  const SPP_GLOBAL int32_t* fsum = G(s.fsum);
  SPP_GLOBAL int32_t* n_ids = G(s.n_ids);
  SPP_GLOBAL uint8_t* dtag = G(s.dtag);
  const int tid = threadIdx.x;
  const int32_t i = bx_ * kNT + tid;
  // ---- round trip 1: state words and the row's bounds (index clamped: T is not known yet)
  const int32_t ic = i < tcap ? i : tcap - 1;
  const int32_t T = st->cnt[h];
  const int32_t err0 = st->error;
  const int32_t p0 = rp[ic];
  const int32_t p1 = rp[ic + 1];
  if ((int64_t)bx_ * kNT >= T || err0) return;   // (workgroup-uniform)
  const bool have = i < T;
  const int32_t n = have ? p1 - p0 : 0;
  if (tid == 0) run_lo = p0;
  if (i == T - 1 || (tid == kNT - 1 && have)) run_hi = p1;   // the last row of the workgroup
  __syncthreads();
  const int32_t P0 = run_lo, len = run_hi - run_lo;          // len <= run_cap: <= 256 rows of <= f edges
  // ---- round trip 2: the run's table values and neighbour entries, coalesced
  for (int32_t k = tid; k < len; k += kNT) {
    const int64_t p = (int64_t)P0 + k;
    lv[k] = evals[p < pcap ? p : pcap - 1];
    lc[k] = cval[p < pcap ? p : pcap - 1];
  }
  __syncthreads();
  const int32_t off = p0 - P0;
  // 8 edges at a time: the rank records of the nodes that are new in this hop (others: record 0), then the ids
  for (int32_t k0 = 0; k0 < n; k0 += 8) {
    uint32_t v[8], q[8];
    int32_t fs[8];
    RankWord rw[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      v[u] = (k0 + u < n) ? lv[off + k0 + u] : 0u;
      const bool fresh = k0 + u < n && v[u] >= (uint32_t)T;
      q[u] = fresh ? v[u] - (uint32_t)T : 0u;
      fs[u] = fsum[q[u] >> 8];
      rw[u] = load_rank_word(fwords, q[u] >> 6);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (k0 + u >= n) break;
      int32_t id = (int32_t)v[u];
      if (v[u] >= (uint32_t)T) {
        id = T + fs[u] + (int32_t)rw[u].pre + __popcll(rw[u].bits & ((1ull << (q[u] & 63)) - 1ull));
        if (q[u] == (uint32_t)(p0 + k0 + u)) {  // n_ids.push_back(c) at its first occurrence
          const int32_t c = lc[off + k0 + u];
          n_ids[id] = (int32_t)((uint32_t)c & idmask);
          if (idbits < 32) dtag[id] = (uint8_t)((uint32_t)c >> idbits);  // the node's degree tag travels with it
        }
      }
      lv[off + k0 + u] = (uint32_t)id;   // (the round's table values are in registers; later rounds' are further on)
    }
  }
  // rank sort of the row (sample_cpu.hpp:126) into its own positions of the run
  for (int32_t k = 0; k < n; ++k) {
    const int32_t vv = (int32_t)lv[off + k];
    int32_t rank = 0;
    for (int32_t m = 0; m < n; ++m) {
      const int32_t w = (int32_t)lv[off + m];
      rank += (w < vv || (w == vv && m < k)) ? 1 : 0;
    }
    lc[off + rank] = vv;   // (the neighbour entries have been consumed: their array takes the sorted row)
  }
  __syncthreads();
  SPP_GLOBAL int32_t* out = G(s.out_col[h]) + P0;
  for (int32_t k = tid; k < len; k += kNT) out[k] = lc[k];
}

// generic path: local id of every edge position (sorted afterwards by hipcub)
__global__ __launch_bounds__(kNT) void k_hop_lids_generic(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                           int32_t h) {
  SPP_GROUP_BLOCK(gg);
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const int32_t E = s.st->E[h];
  const int64_t p = (int64_t)bx_ * kNT + threadIdx.x;
  if (p >= E) return;
  const uint32_t T = (uint32_t)s.st->cnt[h];
  const uint32_t v = s.evals[p];
  const int32_t id = local_id_of(s, T, v);
  if (v == T + (uint32_t)p) s.n_ids[id] = s.cval[p];  // n_ids.push_back(c) at the node's first occurrence
  s.cval[p] = id;
}

// ----------------------------------------------------------------------------------------------
// ownership bucketing of the finished node list (worker distributed branch, fast_sampler.cpp:1017-1272)
// ----------------------------------------------------------------------------------------------
// Same three passes as the standalone spp_partition_batch (partition.hip), grouped over the batches
// of a launch and reading the slot's int32 node list, so that the bucket sizes travel to the host
// with the batch's other counts instead of costing a launch sequence + a blocking read per batch.
struct PartDev {
  Offsets off;
  int32_t P, rank, use_cache;
  const int32_t* cache_map;
  int64_t cache_len;
  int32_t nblk_cap;  // row pitch of pblk
  // membership bits of the cache map (bit v set: cache_map[v] >= 0), rebuilt from the map at every Session
  // start: N/8 bytes stay cache resident (14 MB for 111 M nodes), where a 4-byte lookup per remote node
  // in the 444 MB map was one HBM line each -- more lines per batch than the neighbour reads
  const uint32_t* cache_bits;
  // one bit per 64-node word of cache_bits (set: the word is not zero) -- N/512 bytes, L2 resident: with a cache
  // of ~1 % of the nodes about half of the 64-node words are empty, and their lookups stop here
  const uint32_t* cache_coarse;
};

__global__ __launch_bounds__(256) void k_cache_coarse(const unsigned long long* __restrict__ bits64, int64_t nwords,
                                                      unsigned long long* __restrict__ coarse64) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nc = (nwords + 63) / 64;
  for (int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave; c < nc; c += (int64_t)gridDim.x * (256 / kWave)) {
    const int64_t w = c * 64 + lane;
    const unsigned long long v = bits64[w < nwords ? w : nwords - 1];
    const unsigned long long mask = __ballot(w < nwords && v != 0ull);
    if (lane == 0) coarse64[c] = mask;
  }
}

__global__ __launch_bounds__(256) void k_cache_bits(const int32_t* __restrict__ map, int64_t len,
                                                    unsigned long long* __restrict__ bits64) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t nwords = (len + 63) / 64;
  for (int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) / kWave; w < nwords; w += (int64_t)gridDim.x * (256 / kWave)) {
    const int64_t i = w * 64 + lane;
    const int32_t m = map[i < len ? i : len - 1];
    const unsigned long long mask = __ballot(i < len && m >= 0);
    if (lane == 0) bits64[w] = mask;
  }
}

constexpr int kPartRounds = 4;                   // nodes per thread of k_gpart_hist / k_gpart_scatter
constexpr int kPartSpan = kNT * kPartRounds;     // nodes per workgroup (one row entry of pblk)

// bucket of node v when the cache map entry `m` of v is already in hand (part_bucket_of, partition_common.hip.h)
__device__ __forceinline__ int32_t part_bucket_with(const Offsets& off, int32_t P, int32_t rank, int32_t use_cache,
                                                    int32_t m, int64_t v) {
  if (!use_cache) return owner_of(off, v);
  if (v >= off.v[rank] && v < off.v[rank + 1]) return rank;
  if (m >= 0) return P;
  return owner_of(off, v);
}

__global__ __launch_bounds__(kNT) void k_gpart_hist(const SlotPtrs* __restrict__ slots, GroupGrid gg, int32_t H,
                                                    PartDev a, int32_t ucap) {
  SPP_GROUP_BLOCK(gg);
  __shared__ int32_t cnt[kPartBuckets];
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* n_ids = G(s.n_ids);
  SPP_GLOBAL uint8_t* pbucket = G(s.pbucket);
  // round trip 1: state words and the node ids (clamped index: U is not known yet)
  const int32_t err0 = st->error;
  const int32_t U0 = st->cnt[H];
  int32_t v[kPartRounds];
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    v[r] = n_ids[i < ucap ? i : ucap - 1];
  }
  const int32_t U = err0 ? 0 : U0;
  if ((int64_t)bx_ * kPartSpan >= U) return;
  for (int k = threadIdx.x; k <= a.P; k += kNT) cnt[k] = 0;
  // round trip 2: the cache map entries of the remote nodes (others read entry 0: no predicate on the load)
  int32_t cm[kPartRounds];
  uint32_t cw[kPartRounds];
  bool look_r[kPartRounds];
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    cw[r] = 0u;
    look_r[r] = false;
  }
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    if (i >= U) v[r] = -1;  // node ids are >= 0
    const int64_t vv = v[r];
    const bool look = a.use_cache && vv >= 0 && vv < a.cache_len && !(vv >= a.off.v[a.rank] && vv < a.off.v[a.rank + 1]);
    if (a.cache_bits) {
      look_r[r] = look;
      cw[r] = a.cache_coarse[look ? (vv >> 11) : 0];  // coarse level first (all four in flight)
    } else {  // no Session built the bits (stand-alone spp_sampler_sample): the map itself
      cm[r] = a.use_cache ? a.cache_map[look ? vv : 0] : -1;
      if (!look) cm[r] = -1;
    }
  }
  if (a.cache_bits) {  // fine level, only where the coarse bit is set (the others read word 0: no predicate on the load)
    uint32_t fw[kPartRounds];
#pragma unroll
    for (int r = 0; r < kPartRounds; ++r) {
      const int64_t vv = v[r];
      look_r[r] = look_r[r] && ((cw[r] >> ((vv >> 6) & 31)) & 1u);
      fw[r] = a.cache_bits[look_r[r] ? (vv >> 5) : 0];
    }
#pragma unroll
    for (int r = 0; r < kPartRounds; ++r) cm[r] = (look_r[r] && ((fw[r] >> (v[r] & 31)) & 1u)) ? 0 : -1;  // only the sign is used
  }
  __syncthreads();
  const int lane = threadIdx.x & (kWave - 1);
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    const bool valid = v[r] >= 0;
    int32_t b = -1;
    if (valid) {
      b = part_bucket_with(a.off, a.P, a.rank, a.use_cache, cm[r], (int64_t)v[r]);
      pbucket[i] = (uint8_t)b;
    }
    // one LDS atomic per (wavefront, bucket present in it) instead of one per node: most nodes of a
    // wavefront share an owner, and 64 lanes hammering one counter serialise
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int32_t lb = __shfl(b, leader, kWave);
      const unsigned long long m = __ballot(valid && b == lb);
      if (lane == leader) atomicAdd(&cnt[lb], __popcll(m));
      todo &= ~m;
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= a.P; k += kNT) G(s.pblk)[(int64_t)k * a.nblk_cap + bx_] = cnt[k];
}

// one wavefront per bucket: exclusive scan of the bucket's per-workgroup counts
__global__ __launch_bounds__(kScanNT) void k_gpart_scan(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                        int32_t H, PartDev a) {
  SPP_GROUP_BLOCK(gg);
  const SlotPtrs& s = slots[gg.first_slot + by_];
  SlotState* st = s.st;
  const int32_t U = st->error ? 0 : st->cnt[H];
  const int32_t nblk = (U + kPartSpan - 1) / kPartSpan;
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
  for (int32_t m = wid; m <= a.P; m += kScanNT / kWave) {
    int32_t* row = s.pblk + (int64_t)m * a.nblk_cap;
    int32_t carry = 0;
    for (int32_t base = 0; base < nblk; base += kWave) {
      const int32_t i = base + lane;
      const int32_t v = (i < nblk) ? row[i] : 0;
      const int32_t inc = wave_inclusive_scan(v);
      if (i < nblk) row[i] = carry + inc - v;
      carry += __shfl(inc, kWave - 1, kWave);
    }
    if (lane == 0) st->pcnt[m] = carry;
  }
  if (threadIdx.x == 0) st->pcnt[a.P + 1] = 0;  // every local row is HBM resident
}

__global__ __launch_bounds__(kNT) void k_gpart_scatter(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                       int32_t H, PartDev a, int32_t ucap) {
  SPP_GROUP_BLOCK(gg);
  constexpr int kW = kNT / kWave;
  __shared__ int32_t wcnt[kPartRounds * kW][kPartBuckets];  // [round][wavefront]: the order nodes appear in
  __shared__ int32_t base[kPartBuckets];
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  const SPP_GLOBAL int32_t* n_ids = G(s.n_ids);
  const SPP_GLOBAL uint8_t* pbucket = G(s.pbucket);
  const SPP_GLOBAL int32_t* pblk = G(s.pblk);
  // round trip 1: state words, bucket sizes (one lane each), this workgroup's nodes and their buckets
  // (clamped index: U is not known yet)
  const int32_t err0 = st->error;
  const int32_t U0 = st->cnt[H];
  const int32_t my_pcnt = (int)threadIdx.x <= a.P ? st->pcnt[threadIdx.x] : 0;
  int32_t b[kPartRounds], v[kPartRounds], rank_w[kPartRounds];
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    const int64_t ic = i < ucap ? i : ucap - 1;
    b[r] = (int32_t)pbucket[ic];
    v[r] = n_ids[ic];
  }
  const int32_t U = err0 ? 0 : U0;
  if ((int64_t)bx_ * kPartSpan >= U) return;
  const int wid = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
  for (int k = threadIdx.x; k < kPartRounds * kW * kPartBuckets; k += kNT) (&wcnt[0][0])[k] = 0;
  if ((int)threadIdx.x <= a.P) base[threadIdx.x] = my_pcnt;  // sizes now, offsets after the barrier
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    if (i >= U) b[r] = -1;
  }
  __syncthreads();
  if (threadIdx.x == 0) {  // concat order: parts[0..P-1] then cache hits (sizes -> exclusive offsets, in LDS)
    int32_t acc = 0;
    for (int m = 0; m <= a.P; ++m) {
      const int32_t c = base[m];
      base[m] = acc;
      acc += c;
    }
  }
  const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (kWave - lane));
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    // stable rank inside the wavefront among lanes of the same bucket
    const bool valid = b[r] >= 0;
    rank_w[r] = 0;
    unsigned long long todo = __ballot(valid);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int32_t lb = __shfl(b[r], leader, kWave);
      const unsigned long long m = __ballot(valid && b[r] == lb);
      if (valid && b[r] == lb) {
        rank_w[r] = __popcll(m & below);
        if (lane == leader) wcnt[r * kW + wid][lb] = __popcll(m);
      }
      todo &= ~m;
    }
  }
  __syncthreads();
  // round trip 2: this workgroup's offsets inside the buckets and the cache rows of the cache hits
  int32_t boff[kPartRounds], crow[kPartRounds];
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    const int32_t bb = b[r] >= 0 ? b[r] : 0;
    boff[r] = pblk[(int64_t)bb * a.nblk_cap + bx_];
    crow[r] = a.use_cache ? a.cache_map[b[r] == a.P ? v[r] : 0] : 0;
  }
#pragma unroll
  for (int r = 0; r < kPartRounds; ++r) {
    if (b[r] < 0) continue;
    const int64_t i = (int64_t)bx_ * kPartSpan + r * kNT + threadIdx.x;
    int32_t pre = 0;
    for (int w = 0; w < r * kW + wid; ++w) pre += wcnt[w][b[r]];
    const int32_t pos = base[b[r]] + boff[r] + pre + rank_w[r];
    G(s.pperm)[i] = pos;  // perm_partition_to_mfg (:1085 / :1246-1252)
    int32_t src_row;      // for the fused assembly (k_deliver): one record per node instead of pperm -> segment search -> id
    if (b[r] < a.P) {
      G(s.parts)[pos] = v[r];
      src_row = (b[r] == a.rank) ? (int32_t)((int64_t)v[r] - a.off.v[a.rank]) : pos - base[b[r]];
    } else {
      src_row = crow[r];  // nid2cachenid (:1256)
      G(s.pcached)[pos - base[a.P]] = src_row;
    }
    s.psrc[i] = int2{b[r], src_row};
  }
}

// ids requested from the peers, regrouped peer-major across the batches of a group (one send per
// peer instead of one per peer and batch)
__global__ __launch_bounds__(kNT) void k_pack_remote_ids(const SlotPtrs* __restrict__ slots, GroupGrid gg,
                                                         int32_t P, int32_t rank,
                                                         const int64_t* __restrict__ pack_base,
                                                         int32_t* __restrict__ out) {
  SPP_GROUP_BLOCK(gg);
  __shared__ int32_t seg_start[SPP_MAX_PARTS + 1];  // exclusive offsets of the owners' segments in `parts`
  const SlotPtrs& s = slots[gg.first_slot + by_];
  const SPP_GLOBAL SlotState* st = G(s.st);
  // the bucket sizes: one lane each, then a serial prefix in LDS (a per-thread loop over st->pcnt[] was up
  // to P dependent global loads for every thread)
  const int32_t err0 = st->error;
  if ((int)threadIdx.x < P) seg_start[threadIdx.x + 1] = st->pcnt[threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t acc = 0;
    seg_start[0] = 0;
    for (int m = 1; m <= P; ++m) {
      acc += seg_start[m];
      seg_start[m] = acc;
    }
  }
  __syncthreads();
  if (err0) return;
  const int32_t j = bx_ * kNT + threadIdx.x;
  if (j >= seg_start[P]) return;
  int32_t m = 0;
  while (m + 1 < P && j >= seg_start[m + 1]) ++m;
  if (m == rank) return;
  out[pack_base[(int64_t)by_ * P + m] + (j - seg_start[m])] = G(s.parts)[j];
}

// ----------------------------------------------------------------------------------------------
// export: widen the slot's int32 arrays into the caller's int64 tensors
// ----------------------------------------------------------------------------------------------
struct ExportSegs {
  int32_t n;
  const int32_t* src[2 * SPP_MAX_HOPS + 4];
  int64_t* dst[2 * SPP_MAX_HOPS + 4];
  int64_t start[2 * SPP_MAX_HOPS + 5];
};

// Segment by segment (uniform pointers: the segment's base addresses stay in scalar registers), four
// elements per thread and round with their loads issued together.  The earlier form searched the segment
// of every element and fetched its two pointers from the argument block -- three dependent loads before
// the element's own, one element per thread and round.
__device__ __forceinline__ void export_body(const ExportSegs& g, int64_t vblock, int64_t nvblocks) {
  const int64_t tid = vblock * kNT + threadIdx.x;
  const int64_t nthreads = nvblocks * kNT;
  for (int sgi = 0; sgi < g.n; ++sgi) {
    const int64_t len = g.start[sgi + 1] - g.start[sgi];
    const int32_t* __restrict__ src = g.src[sgi];
    int64_t* __restrict__ dst = g.dst[sgi];
    for (int64_t k0 = tid; k0 < len; k0 += 4 * nthreads) {
      int32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + u * nthreads;
        v[u] = src[k < len ? k : len - 1];  // clamped, not predicated
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t k = k0 + u * nthreads;
        if (k < len) dst[k] = (int64_t)v[u];
      }
    }
  }
}

__global__ __launch_bounds__(kNT) void k_export(ExportSegs g) { export_body(g, blockIdx.x, gridDim.x); }

// Fused delivery of one batch to the caller's tensors in ONE launch (the three separate launches
// cost ~40 us of host-side gaps per batch): workgroups [0, nb_x) gather the feature rows
// x[n_id[i],:], [nb_x, nb_x+nb_e) widen the MFG arrays to int64, the rest gather the label rows.
struct DeliverArgs {
  ExportSegs segs;
  // x = x_src[n_id[:U], :]
  const char* x_src;
  char* x_dst;
  int64_t x_rows, x_row_bytes, x_src_stride;
  int x_chunks, x_lpr_log2;
  int32_t nb_x, nb_e, nb_y;
  // y = y_src[n_id[:bs], :]
  const char* y_src;
  char* y_dst;
  int64_t y_rows, y_row_bytes;
  const int32_t* n_ids;
  // x assembled from {local partition, received rows, cache} instead of one table (native exchange)
  int32_t asm_on, P, rank, pad;
  int64_t rank_offset;
  const char* recv;
  const char* cache;
  int64_t cache_stride;
  const int2* psrc;
  // RCCL transport: row of `recv` where peer m's rows for this batch start.  P2P transport (p2p != 0): the ADDRESS of
  // node 0's row in peer m's partition (base - offsets[m] * stride), so that a remote row is src_base[m] + nid * p2p_stride
  int64_t recv_base[SPP_MAX_PARTS];
  int32_t p2p, nb_a, nb_r, pad2;
  int64_t p2p_stride;
  // row references (spp_mfg_out.row_addr): workgroups [nb_x, nb_x + nb_a) write every row's address instead of moving it
  // (nb_x == 0 then), [.., + nb_r) copy the rows received for this batch into xr_dst (segment m: xr_cnt[m] rows)
  int64_t* addr_out;
  char* xr_dst;
  int32_t xr_cnt[SPP_MAX_PARTS];
};

// The same for a whole GROUP of batches in ONE launch (spp_session_export_group).  A launch per batch left the
// delivery stream's hardware queue idle for ~25 us around every kernel (completion signal, event markers, dispatch
// ramp) -- at 105 us per delivery that, not the memory system, bounded the pipeline at ~0.14 ms per batch.
// Workgroups [start[i], start[i+1]) work on batch i; its DeliverArgs are read from HBM (uniform loads).
struct GroupBlocks {
  int32_t n;
  int32_t start[kMaxGroup + 1];
};

// where the feature row of MFG node r lives (the source the assembly would read)
__device__ __forceinline__ const char* deliver_row_address(const DeliverArgs& a, int64_t r, const int64_t* xr_off) {
  const int64_t nid = a.n_ids[r];
  if (!a.asm_on) return a.x_src + nid * a.x_src_stride;
  const int2 c = a.psrc[r];
  if (c.x == a.rank) return a.x_src + (int64_t)c.y * a.x_src_stride;
  if (c.x == a.P) return a.cache + (int64_t)c.y * a.cache_stride;
  if (a.p2p) return reinterpret_cast<const char*>(a.recv_base[c.x]) + nid * a.p2p_stride;
  return a.xr_dst + (xr_off[c.x] + c.y) * a.x_row_bytes;   // its copy in x_remote (segment-wise, below)
}

template <int VEC, bool kP2P>
__device__ __forceinline__ void deliver_body(const DeliverArgs& a, int b) {
  if (b >= a.nb_x && b < a.nb_x + a.nb_a) {
    // row references: one address per row, four rows per thread and round (psrc / n_ids loads of a round in flight together)
    __shared__ int64_t xr_off[SPP_MAX_PARTS + 1];
    if (threadIdx.x == 0) {
      int64_t acc = 0;
      for (int m = 0; m < a.P; ++m) {
        xr_off[m] = acc;
        acc += a.xr_cnt[m];
      }
    }
    __syncthreads();
    const int64_t nthreads = (int64_t)a.nb_a * kNT;
    for (int64_t r0 = (int64_t)(b - a.nb_x) * kNT + threadIdx.x; r0 < a.x_rows; r0 += 4 * nthreads) {
      const char* p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * nthreads;
        p[u] = deliver_row_address(a, r < a.x_rows ? r : a.x_rows - 1, xr_off);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t r = r0 + u * nthreads;
        if (r < a.x_rows) a.addr_out[r] = (int64_t)reinterpret_cast<uintptr_t>(p[u]);
      }
    }
    return;
  }
  if (b >= a.nb_x + a.nb_a && b < a.nb_x + a.nb_a + a.nb_r) {
    // the rows received for this batch, peer after peer: contiguous in the group's receive buffer, contiguous in x_remote
    const int64_t vb = b - a.nb_x - a.nb_a;
    int64_t done = 0;
    for (int m = 0; m < a.P; ++m) {
      const int64_t bytes = (int64_t)a.xr_cnt[m] * a.x_row_bytes;
      if (bytes == 0) continue;
      const char* src = a.recv + a.recv_base[m] * a.x_row_bytes;
      char* dst = a.xr_dst + done;
      done += bytes;
      if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)bytes) & 15) == 0) {
        const int64_t n16 = bytes >> 4;
        for (int64_t k = vb * kNT + threadIdx.x; k < n16; k += (int64_t)a.nb_r * kNT)
          row_store(reinterpret_cast<const u32x4*>(src)[k], reinterpret_cast<u32x4*>(dst) + k);
      } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | (uintptr_t)bytes) & 3) == 0) {
        const int64_t n4 = bytes >> 2;
        for (int64_t k = vb * kNT + threadIdx.x; k < n4; k += (int64_t)a.nb_r * kNT)
          reinterpret_cast<uint32_t*>(dst)[k] = reinterpret_cast<const uint32_t*>(src)[k];
      } else {
        for (int64_t k = vb * kNT + threadIdx.x; k < bytes; k += (int64_t)a.nb_r * kNT) dst[k] = src[k];
      }
    }
    return;
  }
  if (b >= a.nb_x) b -= a.nb_a + a.nb_r;
  if (b < a.nb_x) {
    if (!a.asm_on) {
#ifndef SPP_DELIVER_NT
#define SPP_DELIVER_NT false   // non-temporal LOADS of the source rows (measurement aid: profiles/r04_ab_INDEX.md)
#endif
      gather_rows_body<VEC, int32_t, SPP_DELIVER_NT>(a.x_src, a.n_ids, a.x_rows, a.x_row_bytes, a.x_chunks, a.x_lpr_log2, a.x_dst, b,
                                     a.nb_x, a.x_src_stride);
    } else if constexpr (kP2P) {
      // P2P transport: a remote row is read in its owner's partition (xGMI), at row (node id - offsets[m])
      // (the peers' bases through LDS: indexing the by-value argument block with a per-lane owner made the compiler keep a
      // private copy of it -- 1.4 KB of scratch per lane)
      __shared__ int64_t peer_base[SPP_MAX_PARTS];
      for (int m = 0; m < a.P; ++m)
        if ((int)threadIdx.x == m) peer_base[m] = a.recv_base[m];
      __syncthreads();
      const int2* __restrict__ psrc = a.psrc;
      const int32_t* __restrict__ nids = a.n_ids;
      const char* xl = a.x_src;
      const char* xc = a.cache;
      const int64_t xs = a.x_src_stride, cs = a.cache_stride, ps = a.p2p_stride;
      const int32_t rank = a.rank, P = a.P;
      move_rows_body<VEC, false>(
          [=](int64_t r) {
            const int2 c = psrc[r];
            return int4{c.x, c.y, nids[r], 0};
          },
          [=](int4 k) -> const char* {
            if (k.x == rank) return xl + (int64_t)k.y * xs;
            if (k.x == P) return xc + (int64_t)k.y * cs;
            return reinterpret_cast<const char*>(peer_base[k.x]) + (int64_t)k.z * ps;
          },
          a.x_rows, a.x_row_bytes, a.x_chunks, a.x_lpr_log2, a.x_dst, b, a.nb_x);
    } else {
      // combine (transferers.py:472-486) without the zeros + scatter + cat + permute passes
      move_rows_body<VEC, false>(
          [&](int64_t r) { return a.psrc[r]; },  // {bucket, row} written by k_gpart_scatter: one load before the row's own
          [&](int2 c) -> const char* {
            if (c.x == a.rank) return a.x_src + (int64_t)c.y * a.x_src_stride;
            if (c.x == a.P) return a.cache + (int64_t)c.y * a.cache_stride;
            return a.recv + (a.recv_base[c.x] + c.y) * a.x_row_bytes;
          },
          a.x_rows, a.x_row_bytes, a.x_chunks, a.x_lpr_log2, a.x_dst, b, a.nb_x);
    }
  } else if (b < a.nb_x + a.nb_e) {
    export_body(a.segs, b - a.nb_x, a.nb_e);
  } else {
    // label rows are tiny (batch_size x 8 B): one lane per row, byte loop
    const int64_t vb = b - a.nb_x - a.nb_e;
    for (int64_t r = vb * kNT + threadIdx.x; r < a.y_rows; r += (int64_t)a.nb_y * kNT) {
      const char* s = a.y_src + (int64_t)a.n_ids[r] * a.y_row_bytes;
      char* d = a.y_dst + r * a.y_row_bytes;
      if ((a.y_row_bytes & 7) == 0 && ((reinterpret_cast<uintptr_t>(s) | reinterpret_cast<uintptr_t>(d)) & 7) == 0) {
        for (int64_t k = 0; k < a.y_row_bytes; k += 8)
          *reinterpret_cast<uint64_t*>(d + k) = *reinterpret_cast<const uint64_t*>(s + k);
      } else {
        for (int64_t k = 0; k < a.y_row_bytes; ++k) d[k] = s[k];
      }
    }
  }
}

#ifndef SPP_DELIVER_WAVES
#define SPP_DELIVER_WAVES 0   // > 0: amdgpu_waves_per_eu for k_deliver (measurement aid: occupancy against registers)
#endif
#if SPP_DELIVER_WAVES > 0
#define SPP_DELIVER_ATTR __attribute__((amdgpu_waves_per_eu(SPP_DELIVER_WAVES, SPP_DELIVER_WAVES)))
#else
#define SPP_DELIVER_ATTR
#endif
// kP2P: the assembly reads remote rows in their owners' partitions (a kernel of its own: its wider source records must
// not cost the default delivery registers -- 76 = six waves per SIMD)
template <int VEC, bool kP2P = false>
__global__ __launch_bounds__(kGatherThreads) SPP_DELIVER_ATTR void k_deliver(DeliverArgs a) {
  static_assert(kGatherThreads == kNT, "one workgroup shape for all three parts");
  deliver_body<VEC, kP2P>(a, (int)blockIdx.x);
}

template <int VEC, bool kP2P = false>
__global__ __launch_bounds__(kGatherThreads) void k_deliver_group(const DeliverArgs* __restrict__ args, GroupBlocks gb) {
  int i = 0;
  while (i + 1 < gb.n && (int)blockIdx.x >= gb.start[i + 1]) ++i;  // uniform: a handful of scalar compares
  deliver_body<VEC, kP2P>(args[i], (int)blockIdx.x - gb.start[i]);
}

}  // namespace spp

// ================================================================================================
// host side
// ================================================================================================
using namespace spp;

// int32 copy of a neighbour array, shared by every sampler created over the same (pointer, nnz, device)
struct Col32 {
  int32_t* p = nullptr;
  int device = 0;
  int idbits = 32;       // < 32: the entries carry min(degree, cap) above bit idbits (k_narrow_col_tagged)
  uint32_t cap = 0;
  ~Col32() {
    if (p) {
      (void)hipSetDevice(device);
      (void)hipFree(p);
    }
  }
};
// Streams of the data path.  SPP_STREAM_PRIORITY=low puts them below the consumer's stream (torch's current stream has
// the default priority): the dispatcher then hands the model step's kernels out first and the data path takes what is left.
// SPP_SAMPLING_PRIORITY / SPP_DELIVERY_PRIORITY set the sampling streams / the delivery and exchange streams apart
// (low | high | a number); unset = SPP_STREAM_PRIORITY = the default priority.
static hipError_t create_data_stream(hipStream_t* st, bool sampling) {
  auto parse = [](const char* name) -> int {
    const char* e = getenv(name);
    if (!e || !*e) return 0;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return 0;
    if (!strcmp(e, "low")) return least;
    if (!strcmp(e, "high")) return greatest;
    return atoi(e);
  };
  static const int prio_all = parse("SPP_STREAM_PRIORITY");
  static const int prio_s = getenv("SPP_SAMPLING_PRIORITY") ? parse("SPP_SAMPLING_PRIORITY") : prio_all;
  static const int prio_d = getenv("SPP_DELIVERY_PRIORITY") ? parse("SPP_DELIVERY_PRIORITY") : prio_all;
  const int prio = sampling ? prio_s : prio_d;
  // Measurement aid (profiles/r05_ab_INDEX.md): SPP_DELIVERY_CU_MASK=n / SPP_SAMPLING_CU_MASK=n confine the delivery (and
  // exchange) stream / the sampling streams to n compute units.  SPP_CU_MASK_LAYOUT=interleaved (default): the first n bits
  // (the driver deals a multi-XCD part's mask bits round-robin over the XCDs); =blocked: the first n / 8 bits of every
  // 32-bit word.  Unset = the whole chip.
  static const int mask_d = getenv("SPP_DELIVERY_CU_MASK") ? atoi(getenv("SPP_DELIVERY_CU_MASK")) : 0;
  static const int mask_s = getenv("SPP_SAMPLING_CU_MASK") ? atoi(getenv("SPP_SAMPLING_CU_MASK")) : 0;
  const int ncu = sampling ? mask_s : mask_d;
  if (ncu > 0 && ncu < 256) {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const char* lay = getenv("SPP_CU_MASK_LAYOUT");
    if (lay && !strcmp(lay, "blocked")) {
      const int per_xcd = (ncu + 7) / 8;
      int left = ncu;
      for (int x = 0; x < 8 && left > 0; ++x)
        for (int k = 0; k < per_xcd && k < 32 && left > 0; ++k, --left) mask[x] |= 1u << k;
    } else {
      for (int k = 0; k < ncu; ++k) mask[k >> 5] |= 1u << (k & 31);
    }
    if (hipExtStreamCreateWithCUMask(st, 8, mask) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
  }
  return prio ? hipStreamCreateWithPriority(st, hipStreamNonBlocking, prio) : hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

static std::mutex g_col32_mu;
static std::map<std::tuple<const void*, int64_t, int, int64_t, int>, std::weak_ptr<Col32>> g_col32;  // (col, nnz, device, generation, tagged)

// row stubs of a graph (k_build_stubs), shared like the int32 neighbour array
struct RowStubs {
  stub4* p = nullptr;
  int device = 0;
  bool from_col32 = false;  // copied from the int32 neighbour array (tagged entries, if that array is tagged)
  ~RowStubs() {
    (void)hipSetDevice(device);
    if (p) (void)hipFree(p);
  }
};
// key: (rowptr, col, nnz, device, generation, the int32 array the entries were copied from or NULL)
static std::map<std::tuple<const void*, const void*, int64_t, int, int64_t, const void*>, std::weak_ptr<RowStubs>> g_stubs;  // guarded by g_col32_mu

struct SlotHost {
  SlotPtrs p{};
  hipEvent_t done = nullptr;        // own event object
  hipEvent_t exported = nullptr;    // recorded by the consumer after its copies out of this slot (session.hip)
  hipEvent_t wait_on = nullptr;     // event that marks this slot's batch complete (group leader's)
  SlotState* host_state = nullptr;  // into the pinned mirror array
  bool sampled = false;
  bool waited = false;
  int64_t ecap_dyn[SPP_MAX_HOPS];   // current capacity of out_col[h]
  int64_t etmp_cap = 0;             // capacity of the per-edge temporaries
  void* cub_tmp = nullptr;
  size_t cub_tmp_bytes = 0;
};

// spp_sampler_opts with every "automatic" resolved (environment read once, at spp_sampler_create)
struct ResolvedOpts {
  int col32 = 1;             // 0 / 1
  int deg_tags = 1;          // 0 / 1 (wanted; whether the ids leave room is decided when the array is built)
  int row_stubs = -1;        // 0 off, 1 always, -1 when an eighth of the free HBM holds them
  int rng_arena = -1;        // 0 per-group generation, 1 arena whenever it fits the budget, -1 also the free-HBM rule
  int64_t rng_arena_words = 0;  // budget
  int fuse = 1;              // 0 never, 1 rule, 2 wherever the kernel can
  int64_t fuse_max_edges = 262144;
  bool flag_tiled = true, rows_coalesced = true, dedup_preread = false;
};

// what hop h of the chain launches (fixed at creation: spp_sampler_get_info reports it)
struct HopPlan {
  bool self_prefix = false;  // k_hop_pick adds up the workgroup sums before its own (else k_hop_scan in between)
  bool fused = false;        // k_hop_pick<kFuse>: no k_bucket_scatter launch
  bool flag_tiled = false;   // k_hop_flag_tiled behind k_bucket_scatter
  bool rows_coal = false;    // k_hop_rows_coalesced
  unsigned row_lds = 0;      // dynamic LDS of k_hop_pick / k_hop_rows: max(f, 1) columns of kNT ints
  unsigned lds_fused = 0;    // dynamic LDS of the fused pick
  int64_t tile_cap = 0;
};

struct spp_sampler {
  spp_sampler_cfg cfg{};
  ResolvedOpts opt{};
  HopPlan plan[SPP_MAX_HOPS];
  int rng_arena_decision = -1;       // spp_sampler_info.rng_arena
  double col32_ms = 0, stubs_ms = 0; // wall clock this sampler spent building the shared tables
  hipEvent_t arena_t0 = nullptr, arena_t1 = nullptr;  // around the arena's generation launch
  double rng_arena_ms = 0;
  bool arena_timed = false;          // arena_t0/t1 hold a generation that has not been read yet
  int64_t tcap[SPP_MAX_HOPS + 1];   // node capacity before hop h (tcap[H] = Ucap)
  int64_t ecap[SPP_MAX_HOPS];
  int64_t dcap = 0;
  bool generic[SPP_MAX_HOPS];
  bool any_generic = false;
  DedupGeom geom{};
  int cb_log2[SPP_MAX_HOPS];        // bucket bits of hop h (<= geom.nb_log2; fewer for small hops)
  int lds_log2 = 12;                // LDS table of k_bucket_dedup (11: 2048 slots, 12: 3072 = 24 KB, 13: 64 KB, 14: 128 KB)
  int64_t bytes = 0;
  std::vector<SlotHost> slots;
  std::vector<void*> allocs;
  // Streams live as long as the sampler (HIP maps streams onto a handful of hardware queues in
  // creation order; creating them once, back to back, keeps the delivery stream and the sampling
  // streams on distinct queues for the whole run instead of re-rolling the mapping every epoch).
  hipStream_t deliver_stream = nullptr;
  hipStream_t comm_stream = nullptr;   // native exchange (created on first use)
  hipStream_t work_streams[kMaxWorkStreams] = {};
  SlotPtrs* d_slots = nullptr;       // device copy of every slot's pointer record
  SlotState* d_states = nullptr;     // contiguous device states
  SlotState* h_states = nullptr;     // pinned mirror
  int32_t* counts = nullptr;         // [slot][counts_per_slot]: kcount, bfill, ticket counter, ovfc (zeroed per batch, one memset)
  int64_t counts_per_slot = 0;       // 2 * nb + 1 + SPP_MAX_HOPS
  std::shared_ptr<Col32> col32_owner;  // int32 copy of cfg.col_dev, shared by the samplers of one graph
  int32_t* col32 = nullptr;          // = col32_owner->p (NULL: read the int64 array)
  std::shared_ptr<RowStubs> stubs_owner;
  const stub4* stubs = nullptr;      // = stubs_owner->p (NULL: degree from rowptr, neighbours from the array only)
  int idbits = 32;                   // bits of a node id inside an entry of col32 / the stubs (32: untagged)
  uint32_t idmask = 0xffffffffu;     // mask of those bits
  uint32_t tag_cap = 0;              // largest degree a tag holds (0: this sampler does not use the tags)
  bool use_tags = false;             // degree pass of hops >= 1 from the nodes' tags, k_hop_pick takes headers from the stubs
  // epoch arena of mt19937 streams (sampler_rng_arena): one stream per batch of the current range table
  uint32_t* rng_arena = nullptr;
  int64_t rng_arena_words = 0;       // allocated size
  int64_t rng_arena_stride = 0;      // words between consecutive batches' streams
  std::vector<uint32_t> rng_arena_seeds;  // seeds the arena currently holds (empty: nothing generated)
  uint32_t* rng_arena_seeds_dev = nullptr;
  int64_t rng_arena_seeds_cap = 0;
  hipEvent_t rng_arena_ready = nullptr;
  hipEvent_t inputs_ready = nullptr;  // sampler_inputs_event
  unsigned long long* cache_bits = nullptr;  // PartDev::cache_bits (owned)
  hipEvent_t cache_bits_ready = nullptr;
  std::unique_ptr<Worker> workers[2];  // persistent host threads lent to Sessions (sampler_worker)
  bool xcd_affinity = true;          // GroupGrid.interleave of the grouped launches (SPP_XCD_AFFINITY=0: batch-major ids)
  PartDev part{};                    // ownership bucketing (part.P == 0: off)
  DeliverArgs* dargs_host[kMaxSets] = {};  // per slot-set: pinned staging + device copy of a group delivery's arguments
  DeliverArgs* dargs_dev[kMaxSets] = {};
  XBuf xbuf[kMaxSets];               // exchange buffers per slot-set (session.hip), kept across Sessions
};

// layout of a slot's first-occurrence rank arrays for `cap` edge positions: [fbits | wpre | fsum] in one allocation
static inline int64_t rank_words(int64_t cap) { return cap / 64 + 64 + 4; }
static inline int64_t rank_blocks(int64_t cap) { return cap / 256 + 16 + 4; }
static inline size_t rank_off_fsum(int64_t cap) { return (size_t)(16 * rank_words(cap)); }
static inline size_t rank_bytes(int64_t cap) { return rank_off_fsum(cap) + (size_t)(4 * rank_blocks(cap)); }

static spp_status dev_alloc(spp_sampler* s, void** out, size_t bytes) {
  if (bytes == 0) bytes = 16;
  SPP_HIP_TRY(hipMalloc(out, bytes));
  s->allocs.push_back(*out);
  s->bytes += (int64_t)bytes;
  return SPP_OK;
}

static spp_status upload_slot(spp_sampler* s, int slot, hipStream_t st) {
  SPP_HIP_TRY(hipMemcpyAsync(s->d_slots + slot, &s->slots[slot].p, sizeof(SlotPtrs), hipMemcpyHostToDevice, st));
  SPP_HIP_TRY(hipStreamSynchronize(st));
  return SPP_OK;
}

extern "C" spp_status spp_sampler_create(const spp_sampler_cfg* cfg, spp_sampler** out) {
  SPP_REQUIRE(cfg && out, "spp_sampler_create: NULL argument");
  SPP_REQUIRE(cfg->num_hops >= 1 && cfg->num_hops <= SPP_MAX_HOPS, "spp_sampler_create: num_hops %d not in [1,%d]",
              cfg->num_hops, SPP_MAX_HOPS);
  SPP_REQUIRE(cfg->rowptr_dev && (cfg->col_dev || cfg->nnz == 0), "spp_sampler_create: NULL graph");
  SPP_REQUIRE(cfg->num_nodes > 0 && cfg->num_nodes < (1ll << 31), "spp_sampler_create: num_nodes %lld must be < 2^31",
              (long long)cfg->num_nodes);
  SPP_REQUIRE(cfg->max_batch > 0 && cfg->num_slots > 0, "spp_sampler_create: max_batch and num_slots must be > 0");
  const spp_partition_cfg& pc = cfg->part;
  SPP_REQUIRE(pc.num_parts >= 0 && pc.num_parts <= SPP_MAX_PARTS, "spp_sampler_create: num_parts %d not in [0,%d]",
              pc.num_parts, SPP_MAX_PARTS);
  if (pc.num_parts > 0) {
    SPP_REQUIRE(pc.rank >= 0 && pc.rank < pc.num_parts, "spp_sampler_create: rank %d out of [0,%d)", pc.rank,
                pc.num_parts);
    SPP_REQUIRE(!pc.use_cache || (pc.cache_map_dev && pc.cache_map_len > 0),
                "spp_sampler_create: use_cache without a cache map");
    for (int m = 0; m < pc.num_parts; ++m)
      SPP_REQUIRE(pc.offsets[m] <= pc.offsets[m + 1], "spp_sampler_create: partition offsets must be non-decreasing");
    SPP_REQUIRE(pc.offsets[0] <= 0 && pc.offsets[pc.num_parts] >= cfg->num_nodes,
                "spp_sampler_create: partition offsets [%lld,%lld) do not cover the %lld nodes",
                (long long)pc.offsets[0], (long long)pc.offsets[pc.num_parts], (long long)cfg->num_nodes);
  }
  int ndev = spp_device_count();
  SPP_REQUIRE(ndev > 0, "spp_sampler_create: no HIP device available (the on-GPU sampler has no CPU fallback)");
  SPP_HIP_TRY(hipSetDevice(cfg->device));

  auto* s = new spp_sampler();
  s->cfg = *cfg;
  if (const char* e = getenv("SPP_XCD_AFFINITY")) s->xcd_affinity = atoi(e) != 0;
  {
    // chain variants: a field of cfg->opts when set, else the environment variable, else the rule (include/spp.h)
    const spp_sampler_opts& o = cfg->opts;
    auto env_int = [](const char* name, long long dflt) -> long long {
      const char* e = getenv(name);
      return e && *e ? atoll(e) : dflt;
    };
    auto sw = [&](int32_t field, const char* name, int dflt) -> int {  // on/off switch -> 0 / 1
      if (field) return field > 0 ? 1 : 0;
      return env_int(name, dflt) != 0 ? 1 : 0;
    };
    ResolvedOpts& r = s->opt;
    r.col32 = sw(o.col32, "SPP_COL32", 1);
    r.deg_tags = sw(o.deg_tags, "SPP_DEG_TAGS", 1);
    if (o.row_stubs) r.row_stubs = o.row_stubs > 0 ? 1 : 0;
    else {
      const long long m = env_int("SPP_ROW_STUBS", -1);
      r.row_stubs = m < 0 ? -1 : (m > 0 ? 1 : 0);
    }
    const bool env_budget = getenv("SPP_RNG_ARENA_MB") != nullptr;
    const long long mb = o.rng_arena_mb > 0 ? o.rng_arena_mb : env_int("SPP_RNG_ARENA_MB", 16384);
    r.rng_arena_words = (mb < 0 ? 0 : mb) * (int64_t)(1 << 20) / 4;
    if (o.rng_arena) r.rng_arena = o.rng_arena > 0 ? 1 : 0;
    else r.rng_arena = (o.rng_arena_mb > 0 || env_budget) ? 1 : -1;   // an explicit budget replaces the free-HBM rule
    if (o.fuse_scatter) r.fuse = o.fuse_scatter < 0 ? 0 : std::min(o.fuse_scatter, 2);
    else r.fuse = (int)std::max<long long>(0, std::min<long long>(2, env_int("SPP_FUSE_SCATTER", 1)));
    r.fuse_max_edges = o.fuse_max_edges > 0 ? o.fuse_max_edges : env_int("SPP_FUSE_MAX_EDGES", 262144);
    r.flag_tiled = sw(o.flag_tiled, "SPP_FLAG_TILED", 1) != 0;
    r.rows_coalesced = sw(o.rows_coalesced, "SPP_ROWS_COALESCED", 1) != 0;
    r.dedup_preread = sw(o.dedup_preread, "SPP_DEDUP_PREREAD", 0) != 0;
  }
  const int H = cfg->num_hops;
  const int64_t node_bound = cfg->num_nodes + cfg->max_batch;  // distinct nodes + duplicated seeds
  s->tcap[0] = cfg->max_batch;
  int64_t etmp = 1;
  for (int h = 0; h < H; ++h) {
    const int64_t f = cfg->sizes[h];
    s->generic[h] = (f < 0 || f > kFastMaxFanout);
    s->any_generic |= s->generic[h];
    if (f >= 0) {
      s->ecap[h] = s->tcap[h] * f;
      s->tcap[h + 1] = std::min(s->tcap[h] * (1 + f), node_bound);
      s->dcap += f * s->tcap[h];
    } else {
      // all-neighbour hop: sized on demand (host sync per hop); start small
      s->ecap[h] = cfg->opts.initial_edge_cap > 0 ? cfg->opts.initial_edge_cap
                                                  : std::min<int64_t>(std::max<int64_t>(cfg->nnz, 1), 1 << 22);
      s->tcap[h + 1] = node_bound;
    }
    etmp = std::max(etmp, s->ecap[h]);
  }
  const int64_t ucap = s->tcap[H];
  if (ucap + etmp >= (1ll << 31) || region_for(etmp) + etmp >= (1ll << 31)) {
    set_error("spp_sampler_create: batch too large for 32-bit positions");
    delete s;
    return SPP_ERR_INVALID;
  }
  // dedup geometry: <= ~1.5k distinct nodes per bucket at the worst case so that a 3072-slot LDS
  // table (24 KB, 6 workgroups per CU) stays at most half full; larger tables only when the bucket
  // count is capped
  static const int64_t bucket_nodes = [] {  // worst-case distinct nodes per bucket the geometry aims for
    const char* e = getenv("SPP_DEDUP_BUCKET");
    const int64_t v = e ? atoll(e) : 1536;
    return v < 64 ? 64 : v;
  }();
  int nb_log2 = 0;
  while (nb_log2 < kMaxBucketsLog2 && (ucap >> nb_log2) > bucket_nodes) ++nb_log2;
  s->geom.nb_log2 = nb_log2;
  s->geom.nb = 1 << nb_log2;
  const int64_t per_bucket = (ucap + s->geom.nb - 1) / s->geom.nb;
  s->geom.kcap = (int32_t)std::min<int64_t>(per_bucket + per_bucket / 2 + 256, 0x7fffffff);
  // tier 12 is a 3072-slot table (SPP_DEDUP_SLOTS12): at most 1536 distinct nodes per bucket in the worst case, half full
  s->lds_log2 = per_bucket > 6144 ? 14 : (per_bucket > 1536 ? 13 : (per_bucket > 820 ? 12 : 11));
  for (int h = 0; h < H; ++h) {
    // as few buckets as keep the hop's worst-case node count per bucket within the LDS table's budget
    int c = std::max(0, nb_log2 - kMaxFineLog2);
    while (c < nb_log2 && (s->tcap[h + 1] >> c) > bucket_nodes) ++c;
    s->cb_log2[h] = c;
  }
  const int nb = s->geom.nb;
  int64_t tmax = 0;
  for (int h = 0; h < H; ++h) tmax = std::max(tmax, s->tcap[h]);
  const int64_t nblk_max = std::max(ceil_div(tmax, kNT), ceil_div(etmp, kNT)) + 1;
  const int nslots = cfg->num_slots;
  if (pc.num_parts > 0) {
    s->part.P = pc.num_parts;
    s->part.rank = pc.rank;
    s->part.use_cache = pc.use_cache ? 1 : 0;
    s->part.cache_map = pc.cache_map_dev;
    s->part.cache_len = pc.cache_map_len;
    s->part.off.n = pc.num_parts + 1;
    for (int m = 0; m <= pc.num_parts; ++m) s->part.off.v[m] = pc.offsets[m];
    s->part.nblk_cap = (int32_t)ceil_div(ucap, kNT);
  }

  s->slots.resize(nslots);
  spp_status rc = SPP_OK;
  void* v = nullptr;
  rc = dev_alloc(s, &v, sizeof(SlotPtrs) * (size_t)nslots);
  s->d_slots = static_cast<SlotPtrs*>(v);
  if (rc == SPP_OK) {
    rc = dev_alloc(s, &v, sizeof(SlotState) * (size_t)nslots);
    s->d_states = static_cast<SlotState*>(v);
  }
  s->counts_per_slot = 2 * (int64_t)nb + 1 + SPP_MAX_HOPS;
  if (rc == SPP_OK) {
    rc = dev_alloc(s, &v, sizeof(int32_t) * (size_t)s->counts_per_slot * (size_t)nslots);
    s->counts = static_cast<int32_t*>(v);
    if (rc == SPP_OK && hipMemset(s->counts, 0, sizeof(int32_t) * (size_t)s->counts_per_slot * (size_t)nslots) != hipSuccess) {
      set_error("spp_sampler_create: hipMemset failed");
      rc = SPP_ERR_HIP;
    }
  }
  if (rc == SPP_OK && hipHostMalloc((void**)&s->h_states, sizeof(SlotState) * (size_t)nslots, hipHostMallocDefault) !=
                          hipSuccess) {
    set_error("spp_sampler_create: hipHostMalloc failed");
    rc = SPP_ERR_HIP;
  }
  for (int i = 0; i < nslots && rc == SPP_OK; ++i) {
    SlotHost& sl = s->slots[i];
    SlotPtrs& p = sl.p;
#define A(ptr, type, count)                                                              \
  if (rc == SPP_OK) {                                                                    \
    void* v_ = nullptr;                                                                  \
    rc = dev_alloc(s, &v_, sizeof(type) * (size_t)(count));                              \
    ptr = static_cast<type*>(v_);                                                        \
  }
    A(p.n_ids, int32_t, ucap);
    A(p.dtag, uint8_t, ucap + 16);
    A(p.deg, int32_t, tmax);
    A(p.rowstart, int64_t, tmax);
    A(p.rng[0], uint32_t, s->dcap + kMtSlack);
    A(p.rng[1], uint32_t, s->dcap + kMtSlack);
    A(p.bsum0, int32_t, nblk_max);
    A(p.bsum1, int32_t, nblk_max);
    A(p.known, unsigned long long, (int64_t)nb * s->geom.kcap);
    for (int h = 0; h < H; ++h) {
      A(p.out_rowptr[h], int32_t, s->tcap[h] + 1);
      A(p.out_col[h], int32_t, s->ecap[h]);
      sl.ecap_dyn[h] = s->ecap[h];
    }
    if (s->part.P > 0) {
      A(p.parts, int32_t, ucap);
      A(p.pcached, int32_t, ucap);
      A(p.pperm, int32_t, ucap);
      A(p.psrc, int2, ucap);
      A(p.pbucket, uint8_t, ucap);
      A(p.pblk, int32_t, (int64_t)(s->part.P + 1) * s->part.nblk_cap);
    }
#undef A
    p.kcount = s->counts + (size_t)i * (size_t)s->counts_per_slot;
    p.bfill = p.kcount + nb;
    p.ctr = p.bfill + nb;
    p.ovfc = p.ctr + 1;
    p.st = s->d_states + i;
    sl.host_state = s->h_states + i;
    // per-edge temporaries are separately allocated so the generic path can grow them
    if (rc == SPP_OK) {
      sl.etmp_cap = etmp;
      hipError_t e = hipMalloc((void**)&p.cval, sizeof(int32_t) * (size_t)etmp);
      const size_t n_bucketed = (size_t)(region_for(etmp) + etmp);  // fixed-capacity bucket regions + overflow list
      if (e == hipSuccess) e = hipMalloc((void**)&p.bpairs, sizeof(unsigned long long) * n_bucketed);
      if (e == hipSuccess) e = hipMalloc((void**)&p.evals, sizeof(uint32_t) * (size_t)etmp);
      if (e == hipSuccess) e = hipMalloc((void**)&p.inv, sizeof(uint32_t) * (size_t)etmp);
      if (e == hipSuccess) e = hipMalloc((void**)&p.tidx, sizeof(uint16_t) * (size_t)etmp);
      if (e == hipSuccess) e = hipMalloc((void**)&p.res, sizeof(uint32_t) * n_bucketed);
      if (e == hipSuccess) e = hipMalloc((void**)&p.fwords, rank_bytes(etmp));
      if (e == hipSuccess) p.fsum = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(p.fwords) + rank_off_fsum(etmp));
      if (e != hipSuccess) {
        set_error("spp_sampler_create: hipMalloc of edge scratch failed: %s", hipGetErrorString(e));
        rc = SPP_ERR_HIP;
      }
      s->bytes += 14 * etmp + 12 * (int64_t)n_bucketed + (int64_t)rank_bytes(etmp);
    }
    if (rc == SPP_OK && (hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess ||
                         hipEventCreateWithFlags(&sl.exported, hipEventDisableTiming) != hipSuccess)) {
      set_error("spp_sampler_create: hipEventCreate failed");
      rc = SPP_ERR_HIP;
    }
  }
  if (rc == SPP_OK && cfg->nnz > 0 && s->opt.col32) {
    std::lock_guard<std::mutex> lk(g_col32_mu);
    // degree tags in the spare top bits (deg_tags off: plain ids), when at least 3 bits are spare
    int idbits = 1;
    while (idbits < 32 && ((int64_t)1 << idbits) < cfg->num_nodes) ++idbits;
    const bool want_tags = s->opt.deg_tags && idbits <= 29;
    const auto key = std::make_tuple((const void*)cfg->col_dev, cfg->nnz, (int)cfg->device, cfg->graph_generation,
                                     want_tags ? 1 : 0);
    std::shared_ptr<Col32> c = g_col32[key].lock();
    if (!c) {
      const auto t0 = std::chrono::steady_clock::now();
      c = std::make_shared<Col32>();
      c->device = cfg->device;
      // + slack: k_hop_pick reads rows in 16-byte pieces that may run past the last row's end
      if (hipMalloc((void**)&c->p, sizeof(int32_t) * ((size_t)cfg->nnz + 16)) != hipSuccess) {
        c->p = nullptr;
        set_error("spp_sampler_create: hipMalloc of the int32 neighbour array failed");
        rc = SPP_ERR_HIP;
      } else {
        uint8_t* deg8 = nullptr;
        if (want_tags && hipMalloc((void**)&deg8, (size_t)cfg->num_nodes) == hipSuccess) {
          const int tagbits = std::min(32 - idbits, 8);
          c->idbits = idbits;
          c->cap = (1u << tagbits) - 1u;
          hipLaunchKernelGGL(k_build_deg8, dim3(256 * 16), dim3(256), 0, nullptr, cfg->rowptr_dev, cfg->num_nodes, deg8);
          hipLaunchKernelGGL(k_narrow_col_tagged, dim3(256 * 32), dim3(256), 0, nullptr, cfg->col_dev, cfg->nnz, deg8,
                             idbits, c->cap, c->p);
        } else {
          (void)hipGetLastError();
          hipLaunchKernelGGL(k_narrow_col, dim3(256 * 16), dim3(256), 0, nullptr, cfg->col_dev, cfg->nnz, c->p);
        }
        if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
          set_error("spp_sampler_create: narrowing the neighbour array failed");
          rc = SPP_ERR_HIP;
        }
        if (deg8) (void)hipFree(deg8);
      }
      if (rc == SPP_OK) {
        g_col32[key] = c;
        s->col32_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    if (rc == SPP_OK) {
      s->col32_owner = c;
      s->col32 = c->p;
      s->idbits = c->idbits;
      s->idmask = c->idbits < 32 ? ((1u << c->idbits) - 1u) : 0xffffffffu;
      s->bytes += (int64_t)sizeof(int32_t) * cfg->nnz;
    }
  }
  // Row stubs (128 B per node): spp_sampler_opts.row_stubs / SPP_ROW_STUBS=0 off, =1 always; default: when they take at
  // most an eighth of the memory that is free right now (the caller still has tensors to place: the outputs of the
  // batches in flight, the model's activations).  Only the fast path uses them.
  if (rc == SPP_OK && cfg->nnz > 0 && cfg->num_nodes > 0 && !s->any_generic && s->opt.row_stubs != 0) {
    const int mode = s->opt.row_stubs;
    const size_t need = sizeof(int32_t) * kStubInts * (size_t)cfg->num_nodes;
    std::lock_guard<std::mutex> lk(g_col32_mu);
    const auto key = std::make_tuple((const void*)cfg->rowptr_dev, (const void*)cfg->col_dev, cfg->nnz, (int)cfg->device,
                                    cfg->graph_generation, (const void*)s->col32);
    std::shared_ptr<RowStubs> c = g_stubs[key].lock();
    if (!c) {
      size_t free_b = 0, total_b = 0;
      // (an EIGHTH since round 6: at S-mag on one GPU -- 187 GB of features -- a quarter admitted the 15.6 GB table, the
      // chain was not measurably faster with it, and torch's caching allocator, left with 4 GB, met retries of ~1 s)
      const bool room = mode > 0 || (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need <= free_b / 8);
      if (room) {
        const auto t0 = std::chrono::steady_clock::now();
        c = std::make_shared<RowStubs>();
        c->device = cfg->device;
        if (hipMalloc((void**)&c->p, need) != hipSuccess) {
          (void)hipGetLastError();
          c->p = nullptr;
          c.reset();  // no stubs: not an error
        } else {
          const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(cfg->num_nodes, 32), 256 * 64);
          if (s->col32)
            hipLaunchKernelGGL(k_build_stubs<int32_t>, dim3(grid), dim3(256), 0, nullptr, cfg->rowptr_dev, s->col32,
                               cfg->num_nodes, c->p);
          else
            hipLaunchKernelGGL(k_build_stubs<int64_t>, dim3(grid), dim3(256), 0, nullptr, cfg->rowptr_dev, cfg->col_dev,
                               cfg->num_nodes, c->p);
          c->from_col32 = s->col32 != nullptr;
          if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
            set_error("spp_sampler_create: building the row stubs failed");
            rc = SPP_ERR_HIP;
          } else {
            g_stubs[key] = c;
            s->stubs_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          }
        }
      }
    }
    if (rc == SPP_OK && c && c->p) {
      s->stubs_owner = c;
      s->stubs = c->p;
      s->bytes += (int64_t)need;
      // Degree tags are used when the stubs were copied from the tagged int32 array and every later hop's fanout is
      // below the tags' cap (then min(degree, cap) decides "all neighbours or f picks" exactly)
      bool fits = s->col32 && c->from_col32 && s->idbits < 32 && !cfg->replace;
      for (int h = 1; h < H && fits; ++h) fits = cfg->sizes[h] >= 0 && (uint64_t)cfg->sizes[h] < s->col32_owner->cap;
      if (fits) {
        s->use_tags = true;
        s->tag_cap = s->col32_owner->cap;
      }
    }
  }
  if (rc == SPP_OK) {
    // k_bucket_scatter stages a tile in up to 80 KB of dynamic LDS (gfx950: 160 KB per CU and per workgroup)
    const int need = (int)(sizeof(int32_t) * 2 * kMaxBuckets + (sizeof(uint32_t) + sizeof(uint16_t)) * kScatterTile);
    int have = 0;
    (void)hipDeviceGetAttribute(&have, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device);
    if (have < need + 1024 ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_scatter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            need) != hipSuccess) {
      set_error("spp_sampler_create: the device offers %d bytes of LDS per workgroup, k_bucket_scatter needs %d "
                "(this library targets gfx950)", have, need + 1024);
      (void)hipGetLastError();
      rc = SPP_ERR_HIP;
    }
  }
  if (rc == SPP_OK) {
    // what every hop launches (spp_sampler_get_info reports it)
    int lds_limit = 0;
    (void)hipDeviceGetAttribute(&lds_limit, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device);
    hipFuncAttributes fattr{};
    const void* fused_fn = reinterpret_cast<const void*>(&k_hop_pick<false, int32_t, true, true, true>);
    const int lds_static = hipFuncGetAttributes(&fattr, fused_fn) == hipSuccess ? (int)fattr.sharedSizeBytes : 8 * 1024;
    (void)hipGetLastError();
    unsigned lds_fused_max = 0;
    for (int h = 0; h < H; ++h) {
      HopPlan& pl = s->plan[h];
      const int32_t f = (int32_t)cfg->sizes[h];
      const int32_t fc = std::max<int32_t>(1, std::min<int32_t>(f, kFastMaxFanout));
      const int64_t gt = std::max<int64_t>(1, ceil_div(s->tcap[h], kNT));
      pl.row_lds = (unsigned)(sizeof(int32_t) * kNT * (size_t)fc);
      pl.self_prefix = !s->generic[h] && gt <= 2048;
      // Scatter folded into the pick kernel (kFuse) where a pick workgroup's edges give bucket runs of >= 8 pairs: the
      // small hops (a hop of more than fuse_max_edges edges per batch goes through k_bucket_scatter and the tiled flag
      // pass instead: hop 2 of [20,20,20], 430 k edges, 1.235 -> 1.219 ms per batch).  Its dynamic LDS -- the picks'
      // columns and the staging area -- stays within 64 KB (fanouts <= 25), and together with the kernel's static arrays
      // (a few KB more: up to ~69 KB in all) within what gfx950 gives one workgroup.
      const unsigned nbk = 1u << s->cb_log2[h];
      pl.tile_cap = (int64_t)kNT * fc;
      pl.lds_fused = pl.row_lds + (unsigned)(sizeof(int32_t) * 2 * nbk + (sizeof(uint32_t) + sizeof(uint16_t)) * (size_t)pl.tile_cap);
      pl.fused = !s->generic[h] && pl.self_prefix && s->col32 && s->stubs && s->use_tags && s->opt.fuse != 0 && f >= 1 &&
                 nbk <= (unsigned)kMaxBuckets && pl.lds_fused <= 64u * 1024u &&
                 (int64_t)pl.lds_fused + lds_static <= (int64_t)lds_limit &&
                 (s->opt.fuse >= 2 || (pl.tile_cap >= 8 * (int64_t)nbk && s->ecap[h] <= s->opt.fuse_max_edges));
      if (pl.fused) lds_fused_max = std::max(lds_fused_max, pl.lds_fused);
      // the flag pass over the scatter's tiles wherever the tile kernel runs on the fast path
      pl.flag_tiled = s->opt.flag_tiled && !pl.fused && !s->generic[h];
      // position-ordered staging of the rows' arrays while 8 bytes per edge of a workgroup's run fit 56 KB of LDS
      pl.rows_coal = !s->generic[h] && s->opt.rows_coalesced && f >= 1 && f <= 28;
    }
    if (lds_fused_max > 0) {
      // the kernel's dynamic-LDS ceiling is a per-process attribute: only ever raised
      static unsigned fused_lds_attr = 0;  // guarded by g_col32_mu
      std::lock_guard<std::mutex> lk(g_col32_mu);
      bool ok = lds_fused_max <= fused_lds_attr;
      if (!ok && hipFuncSetAttribute(fused_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fused_max) == hipSuccess) {
        fused_lds_attr = lds_fused_max;
        ok = true;
      }
      (void)hipGetLastError();
      if (!ok)  // (up to 64 KB a launch needs no attribute) beyond that those hops keep the tile kernel
        for (int h = 0; h < H; ++h)
          if (s->plan[h].fused && s->plan[h].lds_fused > 64u * 1024u) {
            s->plan[h].fused = false;
            s->plan[h].flag_tiled = s->opt.flag_tiled;
          }
    }
  }
  // work streams are created on first use, right after this one (only as many as slot-sets are used)
  if (rc == SPP_OK && create_data_stream(&s->deliver_stream, false) != hipSuccess) {
    set_error("spp_sampler_create: stream creation failed");
    rc = SPP_ERR_HIP;
  }
  if (rc == SPP_OK) {
    std::vector<SlotPtrs> tmp((size_t)nslots);
    for (int i = 0; i < nslots; ++i) tmp[(size_t)i] = s->slots[(size_t)i].p;
    if (hipMemcpy(s->d_slots, tmp.data(), sizeof(SlotPtrs) * (size_t)nslots, hipMemcpyHostToDevice) != hipSuccess) {
      set_error("spp_sampler_create: upload of the slot table failed");
      rc = SPP_ERR_HIP;
    }
  }
  if (rc != SPP_OK) {
    spp_sampler_destroy(s);
    return rc;
  }
  *out = s;
  return SPP_OK;
}

extern "C" void spp_sampler_destroy(spp_sampler* s) {
  if (!s) return;
  s->workers[0].reset();  // joins the host threads before anything they might touch goes away
  s->workers[1].reset();
  (void)hipSetDevice(s->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& sl : s->slots) {
    if (sl.done) (void)hipEventDestroy(sl.done);
    if (sl.exported) (void)hipEventDestroy(sl.exported);
    if (sl.p.cval) (void)hipFree(sl.p.cval);
    if (sl.p.bpairs) (void)hipFree(sl.p.bpairs);
    if (sl.p.evals) (void)hipFree(sl.p.evals);
    if (sl.p.inv) (void)hipFree(sl.p.inv);
    if (sl.p.tidx) (void)hipFree(sl.p.tidx);
    if (sl.p.res) (void)hipFree(sl.p.res);
    if (sl.p.fwords) (void)hipFree(sl.p.fwords);
    if (sl.cub_tmp) (void)hipFree(sl.cub_tmp);
  }
  if (s->rng_arena) (void)hipFree(s->rng_arena);
  if (s->rng_arena_seeds_dev) (void)hipFree(s->rng_arena_seeds_dev);
  if (s->rng_arena_ready) (void)hipEventDestroy(s->rng_arena_ready);
  if (s->arena_t0) (void)hipEventDestroy(s->arena_t0);
  if (s->arena_t1) (void)hipEventDestroy(s->arena_t1);
  if (s->inputs_ready) (void)hipEventDestroy(s->inputs_ready);
  if (s->cache_bits_ready) (void)hipEventDestroy(s->cache_bits_ready);
  if (s->cache_bits) (void)hipFree(s->cache_bits);
  if (s->h_states) (void)hipHostFree(s->h_states);
  if (s->deliver_stream) (void)hipStreamDestroy(s->deliver_stream);
  for (auto st : s->work_streams)
    if (st) (void)hipStreamDestroy(st);
  if (s->comm_stream) (void)hipStreamDestroy(s->comm_stream);
  for (auto& d : s->dargs_host)
    if (d) (void)hipHostFree(d);
  for (auto& d : s->dargs_dev)
    if (d) (void)hipFree(d);
  for (auto& xb : s->xbuf) {
    if (xb.cnt_dev) (void)hipFree(xb.cnt_dev);
    if (xb.cnt_host) (void)hipHostFree(xb.cnt_host);
    if (xb.cnt_ready) (void)hipEventDestroy(xb.cnt_ready);
    if (xb.rows_done) (void)hipEventDestroy(xb.rows_done);
    if (xb.send_ids) (void)hipFree(xb.send_ids);
    if (xb.recv_ids) (void)hipFree(xb.recv_ids);
    if (xb.send_rows) (void)hipFree(xb.send_rows);
    if (xb.recv_rows) (void)hipFree(xb.recv_rows);
  }
  for (void* a : s->allocs) (void)hipFree(a);
  delete s;
}

extern "C" int64_t spp_sampler_workspace_bytes(const spp_sampler* s) { return s ? s->bytes : 0; }

extern "C" void* spp_sampler_deliver_stream(spp_sampler* s) { return s ? (void*)s->deliver_stream : nullptr; }

extern "C" spp_status spp_sampler_get_cfg(const spp_sampler* s, spp_sampler_cfg* out) {
  SPP_REQUIRE(s && out, "spp_sampler_get_cfg: NULL argument");
  *out = s->cfg;
  return SPP_OK;
}

extern "C" spp_status spp_sampler_get_info(spp_sampler* s, spp_sampler_info* out) {
  SPP_REQUIRE(s && out, "spp_sampler_get_info: NULL argument");
  memset(out, 0, sizeof(*out));
  const int H = s->cfg.num_hops;
  out->col32 = s->col32 ? 1 : 0;
  out->deg_tags = s->use_tags ? 1 : 0;
  out->row_stubs = s->stubs ? 1 : 0;
  out->rng_arena = s->rng_arena_decision;
  out->idbits = s->idbits;
  out->tag_cap = (int32_t)s->tag_cap;
  out->num_hops = H;
  out->dedup_buckets_log2 = s->geom.nb_log2;
  out->dedup_table_slots = s->lds_log2 == 11 ? 2048 : (s->lds_log2 == 12 ? SPP_DEDUP_SLOTS12 : (s->lds_log2 == 13 ? 8192 : 16384));
  for (int h = 0; h < H; ++h) {
    out->generic[h] = s->generic[h] ? 1 : 0;
    out->fused_pick[h] = s->plan[h].fused ? 1 : 0;
    out->flag_tiled[h] = s->plan[h].flag_tiled ? 1 : 0;
    out->rows_coalesced[h] = s->plan[h].rows_coal ? 1 : 0;
    out->bucket_log2[h] = s->cb_log2[h];
  }
  if (s->arena_timed) {
    (void)hipSetDevice(s->cfg.device);
    float ms = 0.f;
    if (hipEventSynchronize(s->arena_t1) == hipSuccess && hipEventElapsedTime(&ms, s->arena_t0, s->arena_t1) == hipSuccess)
      s->rng_arena_ms = ms;
    (void)hipGetLastError();
    s->arena_timed = false;
  }
  out->col32_ms = s->col32_ms;
  out->row_stubs_ms = s->stubs_ms;
  out->rng_arena_ms = s->rng_arena_ms;
  out->col32_bytes = s->col32 ? (int64_t)sizeof(int32_t) * (s->cfg.nnz + 16) : 0;
  out->row_stubs_bytes = s->stubs ? (int64_t)sizeof(int32_t) * kStubInts * s->cfg.num_nodes : 0;
  out->rng_arena_bytes = 4 * s->rng_arena_words;
  out->rng_arena_batches = (int64_t)s->rng_arena_seeds.size();
  return SPP_OK;
}

static spp_status grow_edge_scratch(spp_sampler* s, int slot, int h, int64_t need, hipStream_t st) {
  // generic path only; the stream has been synchronised by the caller
  SlotHost& sl = s->slots[(size_t)slot];
  bool changed = false;
  if (need > sl.etmp_cap) {
    int64_t cap = std::max(need, sl.etmp_cap * 2);
    // the rank arrays are preserved: they still describe the previous hop, which k_bucket_dedup needs
    RankWord* old_fwords = sl.p.fwords;
    int32_t* old_fsum = sl.p.fsum;
    const int64_t old_cap = sl.etmp_cap;
    (void)hipFree(sl.p.cval); (void)hipFree(sl.p.bpairs); (void)hipFree(sl.p.evals);
    (void)hipFree(sl.p.inv); (void)hipFree(sl.p.res); (void)hipFree(sl.p.tidx);
    sl.p.tidx = nullptr;
    sl.p.cval = nullptr; sl.p.bpairs = nullptr; sl.p.evals = nullptr; sl.p.fwords = nullptr;
    sl.p.inv = nullptr; sl.p.res = nullptr;
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.cval, sizeof(int32_t) * (size_t)cap));
    SPP_REQUIRE(region_for(cap) + cap < (1ll << 31), "spp_sampler: a hop of %lld edges exceeds 32-bit bucket positions", (long long)need);
    const size_t n_bucketed = (size_t)(region_for(cap) + cap);
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.bpairs, sizeof(unsigned long long) * n_bucketed));
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.evals, sizeof(uint32_t) * (size_t)cap));
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.inv, sizeof(uint32_t) * (size_t)cap));
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.tidx, sizeof(uint16_t) * (size_t)cap));
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.res, sizeof(uint32_t) * n_bucketed));
    SPP_HIP_TRY(hipMalloc((void**)&sl.p.fwords, rank_bytes(cap)));
    sl.p.fsum = reinterpret_cast<int32_t*>(reinterpret_cast<char*>(sl.p.fwords) + rank_off_fsum(cap));
    SPP_HIP_TRY(hipMemcpy(sl.p.fwords, old_fwords, 16 * (size_t)rank_words(old_cap), hipMemcpyDeviceToDevice));
    SPP_HIP_TRY(hipMemcpy(sl.p.fsum, old_fsum, 4 * (size_t)rank_blocks(old_cap), hipMemcpyDeviceToDevice));
    (void)hipFree(old_fwords);
    s->bytes += 48 * (cap - sl.etmp_cap) + (int64_t)rank_bytes(cap) - (int64_t)rank_bytes(sl.etmp_cap);
    sl.etmp_cap = cap;
    changed = true;
  }
  if (need > sl.ecap_dyn[h]) {
    int64_t cap = std::max(need, sl.ecap_dyn[h] * 2);
    void* v = nullptr;
    SPP_HIP_TRY(hipMalloc(&v, sizeof(int32_t) * (size_t)cap));
    s->allocs.push_back(v);  // the old buffer is released at destroy
    sl.p.out_col[h] = static_cast<int32_t*>(v);
    s->bytes += 4 * cap;
    sl.ecap_dyn[h] = cap;
    changed = true;
  }
  if (changed) SPP_TRY(upload_slot(s, slot, st));
  return SPP_OK;
}

namespace spp {

int sampler_max_group(const spp_sampler* s) { return s->any_generic ? 1 : kMaxGroup; }

hipStream_t sampler_work_stream(spp_sampler* s, int i) {
  static const int n_streams = [] {
    const char* e = getenv("SPP_WORK_STREAMS");
    const int v = e ? atoi(e) : 2;  // two sampling streams measured best (1: -22 %, 3: queues get shared)
    return v < 1 ? 1 : (v > kMaxWorkStreams ? kMaxWorkStreams : v);
  }();
  hipStream_t& st = s->work_streams[i % n_streams];
  if (!st && create_data_stream(&st, true) != hipSuccess) st = nullptr;  // null stream as last resort
  return st;
}

// mt19937 streams of a group of batches into rng[buf] of their slots
spp_status sampler_launch_rng(spp_sampler* s, int first_slot, int n, int buf, const uint32_t* seeds,
                              const int64_t* skips, hipStream_t st) {
  if (s->dcap <= 0) return SPP_OK;
  GroupArgs ga{};
  ga.first_slot = first_slot;
  ga.n = n;
  ga.rng_buf = buf;
  for (int i = 0; i < n; ++i) {
    ga.rng_seed[i] = seeds[i];
    ga.rng_skip[i] = skips ? skips[i] : 0;
  }
  hipLaunchKernelGGL(k_rng_fill, dim3(1, (unsigned)n), dim3(kMtThreads), 0, st, s->d_slots, ga, s->dcap);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// Epoch arena: the mt19937 streams of ALL batches of an epoch (seeds[b] = spp_batch_seed(range b's end)),
// generated by one launch on `st` the first time this seed table is seen and kept until a session
// brings a different one.  Returns false (and leaves *base NULL) when the arena would exceed
// SPP_RNG_ARENA_MB (default 16384) or cannot be allocated: the caller then generates per group into
// the slots' ping-pong buffers as before.  *ready: event to order other streams after the generation.
spp_status sampler_rng_arena(spp_sampler* s, const uint32_t* seeds, int64_t nb, hipStream_t st, const uint32_t** base,
                             int64_t* stride, hipEvent_t* ready) {
  *base = nullptr;
  *stride = 0;
  *ready = nullptr;
  if (s->dcap <= 0 || nb <= 0) return SPP_OK;
  s->rng_arena_decision = 0;
  if (s->opt.rng_arena == 0) return SPP_OK;   // spp_sampler_opts.rng_arena off: per-group generation (k_rng_fill)
  const int64_t stride_w = (s->dcap + kMtSlack + 31) / 32 * 32;  // 128-B aligned streams
  const int64_t need = stride_w * nb;
  if (need > s->opt.rng_arena_words) return SPP_OK;
  if (need > s->rng_arena_words && s->opt.rng_arena < 0) {
    // no explicit budget: a new arena may take at most a quarter of the HBM that is free right now; otherwise the streams
    // are generated per group into the slots
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (size_t)need * 4 > free_b / 4) return SPP_OK;
  }
  const bool same = (int64_t)s->rng_arena_seeds.size() == nb && s->rng_arena_stride == stride_w &&
                    std::equal(seeds, seeds + nb, s->rng_arena_seeds.begin());
  if (!same) {
    s->rng_arena_seeds.clear();
    if (need > s->rng_arena_words) {
      if (s->rng_arena) {
        (void)hipFree(s->rng_arena);
        s->bytes -= 4 * s->rng_arena_words;
      }
      s->rng_arena = nullptr;
      s->rng_arena_words = 0;
      if (hipMalloc((void**)&s->rng_arena, sizeof(uint32_t) * (size_t)need) != hipSuccess) {
        (void)hipGetLastError();  // not fatal: fall back to per-group generation
        s->rng_arena = nullptr;
        return SPP_OK;
      }
      s->rng_arena_words = need;
      s->bytes += 4 * need;
    }
    if (nb > s->rng_arena_seeds_cap) {
      if (s->rng_arena_seeds_dev) (void)hipFree(s->rng_arena_seeds_dev);
      s->rng_arena_seeds_dev = nullptr;
      SPP_HIP_TRY(hipMalloc((void**)&s->rng_arena_seeds_dev, sizeof(uint32_t) * (size_t)nb));
      s->rng_arena_seeds_cap = nb;
    }
    if (!s->rng_arena_ready) SPP_HIP_TRY(hipEventCreateWithFlags(&s->rng_arena_ready, hipEventDisableTiming));
    SPP_HIP_TRY(hipMemcpyAsync(s->rng_arena_seeds_dev, seeds, sizeof(uint32_t) * (size_t)nb, hipMemcpyHostToDevice, st));
    SPP_HIP_TRY(hipStreamSynchronize(st));  // `seeds` is the caller's pageable memory
    if (!s->arena_t0) {
      SPP_HIP_TRY(hipEventCreate(&s->arena_t0));
      SPP_HIP_TRY(hipEventCreate(&s->arena_t1));
    }
    SPP_HIP_TRY(hipEventRecord(s->arena_t0, st));
    hipLaunchKernelGGL(k_rng_arena, dim3((unsigned)nb), dim3(kMtThreads), 0, st, s->rng_arena, stride_w,
                       s->rng_arena_seeds_dev, s->dcap);
    SPP_HIP_TRY(hipGetLastError());
    SPP_HIP_TRY(hipEventRecord(s->arena_t1, st));
    s->arena_timed = true;
    SPP_HIP_TRY(hipEventRecord(s->rng_arena_ready, st));
    s->rng_arena_stride = stride_w;
    s->rng_arena_seeds.assign(seeds, seeds + nb);
  }
  *base = s->rng_arena;
  *stride = stride_w;
  *ready = s->rng_arena_ready;
  s->rng_arena_decision = 1;
  return SPP_OK;
}

// sampling chain of a group of batches (RNG already in rng[buf]); records the group's completion event
spp_status sampler_launch_chain(spp_sampler* s, int first_slot, int n, int buf, const int64_t* const* seeds_dev,
                                const int64_t* n_seeds, hipStream_t st, const uint32_t* const* rng_streams) {
  SPP_REQUIRE(n >= 1 && n <= sampler_max_group(s), "sampler_launch_chain: group of %d batches not supported", n);
  SPP_REQUIRE(first_slot >= 0 && first_slot + n <= (int)s->slots.size(), "sampler_launch_chain: slots out of range");
  const int H = s->cfg.num_hops;
  const int64_t* rowptr = s->cfg.rowptr_dev;
  const int64_t* col = s->cfg.col_dev;
  const int32_t* col32 = s->col32;
  const stub4* stubs = s->stubs;
  const int32_t replace = s->cfg.replace ? 1 : 0;
  GroupArgs ga{};
  ga.first_slot = first_slot;
  ga.n = n;
  ga.rng_buf = buf;
  int64_t max_seeds = 1;
  for (int i = 0; i < n; ++i) {
    SPP_REQUIRE(n_seeds[i] >= 0 && n_seeds[i] <= s->cfg.max_batch, "spp_sampler: n_seeds %lld exceeds max_batch %lld",
                (long long)n_seeds[i], (long long)s->cfg.max_batch);
    SPP_REQUIRE(seeds_dev[i] || n_seeds[i] == 0, "spp_sampler: seeds_dev is NULL");
    ga.seeds[i] = seeds_dev[i];
    ga.n_seeds[i] = (int32_t)n_seeds[i];
    ga.rng[i] = rng_streams ? rng_streams[i] : nullptr;
    max_seeds = std::max(max_seeds, n_seeds[i]);
  }
  const unsigned gy = (unsigned)n;
  SlotHost& lead = s->slots[(size_t)first_slot];
  // batch-interleaved workgroup ids: one batch per XCD when the group has 8 batches (GroupGrid)
  const int32_t interleave = s->xcd_affinity ? 1 : 0;
  auto GG = [&](unsigned gx) { return GroupGrid{(int32_t)first_slot, (int32_t)n, gx, interleave}; };

  // Measurement aid (tools/ab_env.sh): SPP_WHATIF_DUP=count,pick,tiles,flag,rows launches the named (idempotent)
  // kernels twice, so that the step time's increase is that kernel's cost IN SITU.  Results are unchanged.
  static const struct Dup { int count, pick, flag, rows; } dup = [] {
    Dup d{1, 1, 1, 1};
    if (const char* e = getenv("SPP_WHATIF_DUP")) {
      if (strstr(e, "count")) d.count = 2;
      if (strstr(e, "pick")) d.pick = 2;
      if (strstr(e, "flag")) d.flag = 2;
      if (strstr(e, "rows")) d.rows = 2;
    }
    return d;
  }();
  const DedupGeom geom = s->geom;
  const int prof = prof_begin(SPP_PROF_CHAIN, st, n);
  // empty known lists / bucket counters of the group's slots (contiguous): 8*nb bytes per batch
  SPP_HIP_TRY(hipMemsetAsync(lead.p.kcount, 0, sizeof(int32_t) * (size_t)s->counts_per_slot * (size_t)n, st));
  const unsigned gseed = (unsigned)ceil_div(max_seeds, kNT);
  ga.grid = GG(gseed);
  hipLaunchKernelGGL(k_seed_init, dim3(gseed * gy), dim3(kNT), 0, st, s->d_slots, ga, geom, rowptr, stubs,
                     (int32_t)s->cfg.sizes[0], replace, s->tag_cap);  // includes hop 0's degree pass
  const uint32_t idmask = s->idmask;
  const uint32_t pick_mask = s->use_tags ? 0xffffffffu : idmask;
  const int32_t row_idbits = s->use_tags ? s->idbits : 32;
  for (int h = 0; h < H; ++h) {
    const int32_t f = (int32_t)s->cfg.sizes[h];
    const unsigned gt = (unsigned)std::max<int64_t>(1, ceil_div(s->tcap[h], kNT));
    const HopPlan& pl = s->plan[h];
    // per-lane row staging of k_hop_pick / k_hop_rows: max(f, 1) columns of kNT ints
    const unsigned row_lds = pl.row_lds;
    for (int rep = 0; h > 0 && rep < dup.count; ++rep) {
      const unsigned gc = (gt + kNT / kWave - 1) / (kNT / kWave);  // one wavefront per 256 targets
      hipLaunchKernelGGL(k_hop_count, dim3((gc) * gy), dim3(kNT), 0, st, s->d_slots, GG(gc), rowptr, stubs, s->use_tags ? 1 : 0, h, f, replace, (int32_t)s->tcap[h]);
    }
    // generic hops are sized after a host sync, so the device-side edge-capacity check is disabled
    const int32_t ecap_dev =
        s->generic[h] ? 0x7fffffff : (int32_t)std::min<int64_t>(lead.ecap_dyn[h], 0x7fffffff);
    // fast path: k_hop_pick adds up the workgroup sums before its own by itself (a few loads per lane up to
    // ~2k workgroups per batch); otherwise a single-workgroup scan launch in between
    const int32_t self_prefix = pl.self_prefix ? 1 : 0;
    if (!self_prefix)
      hipLaunchKernelGGL(k_hop_scan, dim3((1) * gy), dim3(kScanNT), 0, st, s->d_slots, GG(1), h, f, ecap_dev, s->dcap);
    const FuseArgs fa_plain{};
    unsigned ge;
    for (int rep = 0; !s->generic[h] && rep < dup.pick - 1; ++rep)
      if (col32 && stubs && s->use_tags)
        hipLaunchKernelGGL((k_hop_pick<false, int32_t, true, true>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col32, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
    // the hop's dedup geometry (needed by the fused pick already)
    const int32_t cb = s->cb_log2[h];
    const unsigned nbk = 1u << cb;
    const int64_t pcap = std::max<int64_t>(1, s->generic[h] ? 0 : s->ecap[h]);   // (generic hops: set below, after the host sync)
    // scatter folded into the pick kernel on the small hops, chosen at creation (HopPlan; spp_sampler_opts.fuse_scatter)
    const bool fused = pl.fused;
    const int64_t tile_cap = pl.tile_cap;
    const unsigned lds_fused = pl.lds_fused;
    if (!s->generic[h]) {
      if (fused) {
        // (first position of the overflow list: a sampler with a fused hop has no generic hop, its scratch never grows)
        FuseArgs fa{cb, bucket_cap(pcap, nbk), (int32_t)region_for(lead.etmp_cap), idmask, (int32_t)(row_lds / sizeof(int32_t)), (int32_t)tile_cap};
        const unsigned lds = lds_fused;
        hipLaunchKernelGGL((k_hop_pick<false, int32_t, true, true, true>), dim3((gt) * gy), dim3(kNT), lds, st, s->d_slots, GG(gt),
                           col32, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa);
      } else if (col32 && stubs && s->use_tags)
        hipLaunchKernelGGL((k_hop_pick<false, int32_t, true, true>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col32, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      else if (col32 && stubs)
        hipLaunchKernelGGL((k_hop_pick<false, int32_t, true>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col32, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      else if (col32)
        hipLaunchKernelGGL((k_hop_pick<false, int32_t, false>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col32, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      else if (stubs)
        hipLaunchKernelGGL((k_hop_pick<false, int64_t, true>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      else
        hipLaunchKernelGGL((k_hop_pick<false, int64_t, false>), dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt),
                           col, stubs, h, f, replace, self_prefix, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      ge = (unsigned)std::max<int64_t>(1, ceil_div(s->ecap[h], kNT));
    } else {
      // slow path (n == 1): the edge count is needed on the host to size launches and scratch
      SPP_HIP_TRY(hipMemcpyAsync(lead.host_state, lead.p.st, sizeof(SlotState), hipMemcpyDeviceToHost, st));
      SPP_HIP_TRY(hipStreamSynchronize(st));
      const int64_t E = lead.host_state->E[h];
      if (lead.host_state->error) break;
      SPP_TRY(grow_edge_scratch(s, first_slot, h, E, st));
      ge = (unsigned)std::max<int64_t>(1, ceil_div(E, kNT));
      hipLaunchKernelGGL((k_hop_pick<true, int64_t, false>), dim3((gt) * gy), dim3(kNT), sizeof(int32_t) * kNT, st, s->d_slots,
                         GG(gt), col, stubs, h, f, replace, 0, ecap_dev, s->dcap, (int32_t)s->tcap[h], pick_mask, fa_plain);
      if (col32)
        hipLaunchKernelGGL(k_hop_expand_generic<int32_t>, dim3((ge) * gy), dim3(kNT), 0, st, s->d_slots, GG(ge), col32,
                           h, f, replace, idmask);
      else
        hipLaunchKernelGGL(k_hop_expand_generic<int64_t>, dim3((ge) * gy), dim3(kNT), 0, st, s->d_slots, GG(ge), col,
                           h, f, replace, idmask);
    }
    // first position of the overflow list behind the bucket regions -- AFTER the generic branch, which may just have
    // grown the per-edge scratch (the regions are sized from the hop's edge count: an overflow list placed by the old
    // capacity would lie inside them)
    const int32_t region = (int32_t)region_for(lead.etmp_cap);
    // dedup: regroup into fixed-capacity bucket regions -> one workgroup per bucket with an LDS table
    // bit 0: the known lists are not read after the last hop; bit 1: the candidates skip the table pre-read (most edges of
    // a hop reach nodes that are new to the batch; spp_sampler_opts.dedup_preread keeps the pre-read)
    const bool dedup_preread = s->opt.dedup_preread;
    const int32_t last = ((h == H - 1) ? 1 : 0) | (!dedup_preread ? 2 : 0);
    const int32_t hop_word = h | (s->cb_log2[h] << 8) | (last << 16);
    // positions < pcap_h are inside the per-edge scratch arrays whatever E turns out to be
    const int64_t pcap_h = s->generic[h] ? std::max<int64_t>(1, lead.host_state->E[h]) : pcap;
    const unsigned gsc = (unsigned)std::max<int64_t>(1, ceil_div((int64_t)ge * kNT, kScatterTile));
    const unsigned sc_lds = (unsigned)(sizeof(int32_t) * 2 * nbk + (sizeof(uint32_t) + sizeof(uint16_t)) * kScatterTile);
    const int32_t bcap = bucket_cap(pcap_h, nbk);                   // pairs a bucket's region holds
    // the flag pass over the scatter's tiles (k_hop_flag_tiled) wherever the tile kernel runs on the fast path;
    // spp_sampler_opts.flag_tiled off: position-ordered inv and one random word per position
    const bool flag_tiled = pl.flag_tiled;
    if (!fused)
      hipLaunchKernelGGL(k_bucket_scatter, dim3((gsc) * gy), dim3(kTileNT), sc_lds, st, s->d_slots, GG(gsc), h, cb, bcap, region, pcap_h, idmask,
                         flag_tiled ? 1 : 0);
    // LDS table of k_bucket_dedup: 2048 / 4096 / 8192 / 16384 slots (any size works: multiply-shift slot index).
    // 3584 slots let five workgroups share a compute unit's LDS instead of four; measured: no difference
    // (lone chain 44-47 us per batch at 4096, 3584, 3072 and 2560 slots), so the roomier table stays.
    // Measurement aid (profiles/r06_ab_INDEX.md): SPP_WHATIF_DEDUP_LDS_PAD=<bytes> of unused dynamic LDS per dedup workgroup of
    // the LAST hop -- fewer of them fit a compute unit (24 KB: three instead of six), i.e. the kernel's wave-slot footprint
    // beside the delivery shrinks while its own duration grows.  Results are unchanged.
    static const unsigned dedup_pad = [] { const char* e = getenv("SPP_WHATIF_DEDUP_LDS_PAD"); return e ? (unsigned)atoi(e) : 0u; }();
    const unsigned dpad = (h == H - 1) ? dedup_pad : 0u;
    // (a hop whose bucket spans more than four fine buckets walks their known lists as one concatenation: kFlat)
    static const bool dedup_flat_on = [] { const char* e = getenv("SPP_DEDUP_FLAT"); return !e || atoi(e) != 0; }();
    const bool flat = dedup_flat_on && (geom.nb_log2 - cb) > 2;
    if (s->lds_log2 == 11 && flat)
      hipLaunchKernelGGL((k_bucket_dedup<2048, true>), dim3((nbk) * gy), dim3(kNT), dpad, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    else if (s->lds_log2 == 12 && flat)
      hipLaunchKernelGGL((k_bucket_dedup<SPP_DEDUP_SLOTS12, true>), dim3((nbk) * gy), dim3(kNT), dpad, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    else if (s->lds_log2 == 11)
      hipLaunchKernelGGL(k_bucket_dedup<2048>, dim3((nbk) * gy), dim3(kNT), dpad, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    else if (s->lds_log2 == 12)
      hipLaunchKernelGGL(k_bucket_dedup<SPP_DEDUP_SLOTS12>, dim3((nbk) * gy), dim3(kNT), dpad, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    else if (s->lds_log2 == 13)
      hipLaunchKernelGGL(k_bucket_dedup<8192>, dim3((nbk) * gy), dim3(kNT), 0, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    else
      hipLaunchKernelGGL(k_bucket_dedup<16384>, dim3((nbk) * gy), dim3(kNT), 0, st, s->d_slots, GG(nbk), hop_word, geom, bcap, region);
    const unsigned gflag = (unsigned)std::max<int64_t>(1, ceil_div((int64_t)ge * kNT, kFlagSpan));
    // (flag + rows as ONE pass with a decoupled look-back over per-tile status granules was built and measured in round 5:
    // bit-exact, and 289 us against 102 + 93 us per 16-batch launch of the last hop -- a tile holds its wave slots while it
    // waits for the slowest of its predecessors' gathers; profiles/r05_ab_INDEX.md, commit "Sampling chain experiments")
    for (int rep = 0; rep < dup.flag; ++rep) {
      if (flag_tiled)
        hipLaunchKernelGGL(k_hop_flag_tiled, dim3((gsc) * gy), dim3(kFlagNT), 0, st, s->d_slots, GG(gsc), h, f,
                           (int32_t)s->tcap[H], pcap_h);
      else
        hipLaunchKernelGGL(k_hop_flag, dim3((gflag) * gy), dim3(kFlagNT), 0, st, s->d_slots, GG(gflag), h, f,
                           (int32_t)s->tcap[H], pcap_h);
    }
    if (!s->generic[h]) {
      // position-ordered staging of the rows' arrays (k_hop_rows_coalesced) while 8 bytes per edge of a workgroup's run fit
      // 56 KB of LDS (f <= 28); spp_sampler_opts.rows_coalesced off: the lane-per-row loads
      const int32_t run_cap = (int32_t)(kNT * std::max<int32_t>(1, std::min<int32_t>(f, kFastMaxFanout)));
      for (int rep = 0; rep < dup.rows; ++rep) {
        if (pl.rows_coal)
          hipLaunchKernelGGL(k_hop_rows_coalesced, dim3((gt) * gy), dim3(kNT), (unsigned)(8 * run_cap), st, s->d_slots, GG(gt), h,
                             idmask, row_idbits, (int32_t)s->tcap[h], pcap_h, run_cap);
        else
          hipLaunchKernelGGL(k_hop_rows, dim3((gt) * gy), dim3(kNT), row_lds, st, s->d_slots, GG(gt), h, idmask, row_idbits,
                             (int32_t)s->tcap[h], pcap_h);
      }
    } else {
      const int64_t E = lead.host_state->E[h];
      const int32_t T = lead.host_state->cnt[h];
      if (E > 0) {
        hipLaunchKernelGGL(k_hop_lids_generic, dim3((ge) * gy), dim3(kNT), 0, st, s->d_slots, GG(ge), h);
        size_t need = 0;
        SPP_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, need, lead.p.cval, lead.p.out_col[h], (int)E, T,
                                                               lead.p.out_rowptr[h], lead.p.out_rowptr[h] + 1, 0, 32,
                                                               st));
        if (need > lead.cub_tmp_bytes) {
          SPP_HIP_TRY(hipStreamSynchronize(st));
          if (lead.cub_tmp) (void)hipFree(lead.cub_tmp);
          lead.cub_tmp = nullptr;
          SPP_HIP_TRY(hipMalloc(&lead.cub_tmp, need));
          lead.cub_tmp_bytes = need;
        }
        size_t have = lead.cub_tmp_bytes;
        SPP_HIP_TRY(hipcub::DeviceSegmentedRadixSort::SortKeys(lead.cub_tmp, have, lead.p.cval, lead.p.out_col[h],
                                                               (int)E, T, lead.p.out_rowptr[h],
                                                               lead.p.out_rowptr[h] + 1, 0, 32, st));
      }
    }
  }
  if (s->part.P > 0) {
    const unsigned gu = (unsigned)std::max<int64_t>(1, ceil_div(s->tcap[H], kPartSpan));
    hipLaunchKernelGGL(k_gpart_hist, dim3((gu) * gy), dim3(kNT), 0, st, s->d_slots, GG(gu), H, s->part, (int32_t)s->tcap[H]);
    hipLaunchKernelGGL(k_gpart_scan, dim3((1) * gy), dim3(kScanNT), 0, st, s->d_slots, GG(1), H, s->part);
    hipLaunchKernelGGL(k_gpart_scatter, dim3((gu) * gy), dim3(kNT), 0, st, s->d_slots, GG(gu), H, s->part, (int32_t)s->tcap[H]);
  }
  prof_end(SPP_PROF_CHAIN, prof, st);
  SPP_HIP_TRY(hipGetLastError());
  SPP_HIP_TRY(hipMemcpyAsync(lead.host_state, lead.p.st, sizeof(SlotState) * (size_t)n, hipMemcpyDeviceToHost, st));
  SPP_HIP_TRY(hipEventRecord(lead.done, st));
  for (int i = 0; i < n; ++i) {
    SlotHost& sl = s->slots[(size_t)(first_slot + i)];
    sl.wait_on = lead.done;
    sl.sampled = true;
    sl.waited = false;
  }
  return SPP_OK;
}

// Describes the delivery of the (waited) batch in `slot` -- MFG widening + x and y row gathers; any of mfg / x / y
// may be absent -- as the workgroup ranges of one launch.  *vec: widest access all of its buffers allow.
static spp_status fill_deliver_args(spp_sampler* s, int slot, const spp_mfg_out* mfg, const void* x_src,
                                    int64_t x_row_bytes, int64_t x_src_stride, void* x_dst, const void* y_src,
                                    int64_t y_row_bytes, int64_t y_rows, void* y_dst, const AssembleSrc* asrc,
                                    DeliverArgs& a, int* vec_out) {
  SlotHost& sl = s->slots[(size_t)slot];
  if (!sl.sampled || !sl.waited) {
    set_error("spp_session_export: slot %d must be sampled and waited first", slot);
    return SPP_ERR_STATE;
  }
  const SlotState* hs = sl.host_state;
  const int H = s->cfg.num_hops;
  const int64_t U = hs->cnt[H];
  a = DeliverArgs{};
  int n = 0;
  int64_t total = 0;
  auto add = [&](const int32_t* src, int64_t* dst, int64_t len) {
    if (len <= 0 || dst == nullptr) return;
    a.segs.src[n] = src;
    a.segs.dst[n] = dst;
    a.segs.start[n] = total;
    total += len;
    ++n;
  };
  // Measurement aid (WRONG RESULTS: the MFG tensors stay unwritten): SPP_WHATIF_NO_WIDEN=1 drops the int32 -> int64
  // widening from the delivery launch, which bounds what writing the final arrays from the chain could gain.
  static const bool whatif_no_widen = [] { const char* e = getenv("SPP_WHATIF_NO_WIDEN"); return e && atoi(e) != 0; }();
  if (mfg && !whatif_no_widen) {
    add(sl.p.n_ids, mfg->n_id, U);
    for (int k = 0; k < H; ++k) {
      const int h = H - 1 - k;
      add(sl.p.out_rowptr[h], mfg->rowptr[k], (int64_t)hs->cnt[h] + 1);
      add(sl.p.out_col[h], mfg->col[k], hs->E[h]);
    }
    if (s->part.P > 0) {
      int64_t owned = 0;
      for (int m = 0; m < s->part.P; ++m) owned += hs->pcnt[m];
      add(sl.p.parts, mfg->parts, owned);
      add(sl.p.pcached, mfg->cached, hs->pcnt[s->part.P]);
      add(sl.p.pperm, mfg->perm, U);
    }
  }
  a.segs.n = n;
  a.segs.start[n] = total;
  a.n_ids = sl.p.n_ids;
  a.nb_e = total > 0 ? (int32_t)std::min<int64_t>(ceil_div(total, kNT), 1024) : 0;
  int vec = 16;
  if (asrc) {
    SPP_REQUIRE(s->part.P > 0, "sampler_deliver: feature assembly needs ownership bucketing");
    x_src = asrc->x_local;
    x_src_stride = asrc->x_local_stride;
    a.cache_stride = asrc->cache_stride > 0 ? asrc->cache_stride : x_row_bytes;
    a.asm_on = 1;
    a.P = s->part.P;
    a.rank = s->part.rank;
    a.rank_offset = s->part.off.v[s->part.rank];
    a.recv = asrc->recv;
    a.cache = asrc->cache;
    a.psrc = sl.p.psrc;
    for (int m = 0; m < s->part.P; ++m) a.recv_base[m] = asrc->recv_base[m];
    if (asrc->p2p) {
      a.p2p = 1;
      a.p2p_stride = asrc->peer_stride > 0 ? asrc->peer_stride : x_row_bytes;
      for (int m = 0; m < s->part.P; ++m) {
        SPP_REQUIRE(m == s->part.rank || hs->pcnt[m] == 0 || asrc->peer[m], "sampler_deliver: no table of peer %d", m);
        a.recv_base[m] = (int64_t)reinterpret_cast<uintptr_t>(asrc->peer[m]) - s->part.off.v[m] * a.p2p_stride;
      }
    }
    SPP_REQUIRE(hs->pcnt[s->part.P] == 0 || asrc->cache, "sampler_deliver: cache hits without cache rows");
  }
  // row references instead of rows (spp_mfg_out.row_addr; the caller passes no x destination)
  if (mfg && mfg->row_addr && !x_dst && (x_src || asrc) && U > 0 && x_row_bytes > 0) {
    if (x_src_stride <= 0) x_src_stride = x_row_bytes;
    a.x_src = static_cast<const char*>(x_src);
    a.x_src_stride = x_src_stride;
    a.x_rows = U;
    a.x_row_bytes = x_row_bytes;
    a.addr_out = mfg->row_addr;
    a.nb_a = (int32_t)std::min<int64_t>(ceil_div(U, 4 * kNT), 512);
    if (asrc && !asrc->p2p) {
      int64_t remote = 0;
      for (int m = 0; m < s->part.P; ++m) {
        a.xr_cnt[m] = (m == s->part.rank) ? 0 : hs->pcnt[m];
        remote += a.xr_cnt[m];
      }
      if (remote > 0) {
        SPP_REQUIRE(mfg->x_remote, "sampler_deliver: row references of a batch with %lld received rows need x_remote",
                    (long long)remote);
        a.xr_dst = static_cast<char*>(mfg->x_remote);
        a.nb_r = (int32_t)std::min<int64_t>(ceil_div(remote * x_row_bytes, 16 * 4 * kNT), 1024);
      }
    }
  }
  if ((x_src || asrc) && x_dst && U > 0 && x_row_bytes > 0) {  // a rank may own no rows at all (x_local NULL)
    if (x_src_stride <= 0) x_src_stride = x_row_bytes;
    SPP_REQUIRE(x_src_stride >= x_row_bytes, "spp_session_export: source stride %lld smaller than the row (%lld bytes)",
                (long long)x_src_stride, (long long)x_row_bytes);
    uintptr_t align_probe = reinterpret_cast<uintptr_t>(x_src);
    if (asrc) {
      align_probe |= reinterpret_cast<uintptr_t>(asrc->recv) | reinterpret_cast<uintptr_t>(asrc->cache) |
                     (uintptr_t)a.cache_stride;
      if (asrc->p2p) {
        align_probe |= (uintptr_t)a.p2p_stride;
        for (int m = 0; m < s->part.P; ++m) align_probe |= reinterpret_cast<uintptr_t>(asrc->peer[m]);
      }
    }
    const GatherGeom gg = gather_geometry(reinterpret_cast<const void*>(align_probe), x_dst, x_row_bytes, U, x_src_stride,
                                          /*allow_span=*/!asrc);  // (received rows are dense: no padding to read into)
    a.x_src_stride = x_src_stride;
    vec = gg.vec;
    a.x_src = static_cast<const char*>(x_src);
    a.x_dst = static_cast<char*>(x_dst);
    a.x_rows = U;
    a.x_row_bytes = x_row_bytes;
    a.x_chunks = gg.chunks;
    a.x_lpr_log2 = gg.lpr_log2;
    a.nb_x = (int32_t)gg.grid;
  }
  const int64_t ny = std::min<int64_t>(y_rows, U);
  if (y_src && y_dst && ny > 0 && y_row_bytes > 0) {
    a.y_src = static_cast<const char*>(y_src);
    a.y_dst = static_cast<char*>(y_dst);
    a.y_rows = ny;
    a.y_row_bytes = y_row_bytes;
    a.nb_y = (int32_t)std::min<int64_t>(ceil_div(ny, kNT), 64);
  }
  *vec_out = vec;
  return SPP_OK;
}

// Fused delivery of the (waited) batch in `slot` on the caller's stream in one launch.
spp_status sampler_deliver(spp_sampler* s, int slot, const spp_mfg_out* mfg, const void* x_src, int64_t x_row_bytes,
                           int64_t x_src_stride, void* x_dst, const void* y_src, int64_t y_row_bytes, int64_t y_rows,
                           void* y_dst, const AssembleSrc* asrc, hipStream_t st) {
  DeliverArgs a;
  int vec = 1;
  SPP_TRY(fill_deliver_args(s, slot, mfg, x_src, x_row_bytes, x_src_stride, x_dst, y_src, y_row_bytes, y_rows, y_dst, asrc,
                            a, &vec));
  const unsigned grid = (unsigned)(a.nb_x + a.nb_a + a.nb_r + a.nb_e + a.nb_y);
  if (grid == 0) return SPP_OK;
  const int prof = prof_begin(SPP_PROF_GATHER, st, a.x_rows);
  if (a.p2p && a.nb_x > 0) {
    switch (vec) {
      case 16: hipLaunchKernelGGL((k_deliver<16, true>), dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 8: hipLaunchKernelGGL((k_deliver<8, true>), dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 4: hipLaunchKernelGGL((k_deliver<4, true>), dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 2: hipLaunchKernelGGL((k_deliver<2, true>), dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      default: hipLaunchKernelGGL((k_deliver<1, true>), dim3(grid), dim3(kGatherThreads), 0, st, a); break;
    }
  } else {
    switch (vec) {
      case kVecSpan: hipLaunchKernelGGL(k_deliver<kVecSpan>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 16: hipLaunchKernelGGL(k_deliver<16>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 8: hipLaunchKernelGGL(k_deliver<8>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 4: hipLaunchKernelGGL(k_deliver<4>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      case 2: hipLaunchKernelGGL(k_deliver<2>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
      default: hipLaunchKernelGGL(k_deliver<1>, dim3(grid), dim3(kGatherThreads), 0, st, a); break;
    }
  }
  prof_end(SPP_PROF_GATHER, prof, st);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// The same for the `n` (waited) batches in slots first_slot.. in ONE launch (k_deliver_group).  Their DeliverArgs
// travel through a pinned staging record of the slot-set `set` and one small copy on `st`: the set's previous
// delivery -- and with it the previous copy out of this staging record -- completed before the chain that filled
// these slots could start, so the record is free.
spp_status sampler_deliver_group(spp_sampler* s, int set, int first_slot, int n, const spp_group_out* outs,
                                 const void* x_src, int64_t x_row_bytes, int64_t x_src_stride, const void* y_src,
                                 int64_t y_row_bytes, const int64_t* y_rows, const AssembleSrc* asrc /*[n] or NULL*/,
                                 hipStream_t st) {
  SPP_REQUIRE(n >= 1 && n <= kMaxGroup && set >= 0 && set < kMaxSets, "sampler_deliver_group: bad group");
  if (!s->dargs_host[set]) {
    SPP_HIP_TRY(hipHostMalloc((void**)&s->dargs_host[set], sizeof(DeliverArgs) * kMaxGroup, hipHostMallocDefault));
    SPP_HIP_TRY(hipMalloc((void**)&s->dargs_dev[set], sizeof(DeliverArgs) * kMaxGroup));
    s->bytes += (int64_t)sizeof(DeliverArgs) * kMaxGroup;
  }
  DeliverArgs* ha = s->dargs_host[set];
  GroupBlocks gb{};
  gb.n = n;
  int vec = 16;
  int64_t rows = 0, blocks = 0;
  for (int i = 0; i < n; ++i) {
    int v = 16;
    SPP_TRY(fill_deliver_args(s, first_slot + i, &outs[i].mfg, x_src, x_row_bytes, x_src_stride, outs[i].x_out, y_src,
                              y_row_bytes, y_rows[i], outs[i].y_out, asrc ? &asrc[i] : nullptr, ha[i], &v));
    // (kVecSpan only when every batch of the group qualifies; otherwise such a batch runs in the 8-byte form)
    if (i == 0) vec = v;
    else if (vec != v) vec = std::min(vec == kVecSpan ? 8 : vec, v == kVecSpan ? 8 : v);
    gb.start[i] = (int32_t)blocks;
    blocks += ha[i].nb_x + ha[i].nb_a + ha[i].nb_r + ha[i].nb_e + ha[i].nb_y;
    rows += ha[i].x_rows;
  }
  if (blocks == 0) return SPP_OK;
  // Occupancy cap.  One launch covers the group, so without a cap its workgroups would hold every wave slot of the
  // chip for the whole ~0.8 ms and the sampling chains of the next groups -- whose tile kernels need 16 free wave
  // slots on ONE compute unit -- would starve behind it (the tile kernel of hop 1: 0.8 ms per launch).  The row
  // gather is a grid-stride loop: SPP_DELIVER_WG_PER_CU workgroups (4 wavefronts each) per compute unit.
  static const int wg_per_cu = [] {
    const char* e = getenv("SPP_DELIVER_WG_PER_CU");
    const int v = e ? atoi(e) : 6;
    return v < 1 ? 1 : v;
  }();
  const int64_t cap_x = std::max<int64_t>(1, (int64_t)256 * wg_per_cu / n);
  blocks = 0;
  for (int i = 0; i < n; ++i) {
    ha[i].nb_x = (int32_t)std::min<int64_t>(ha[i].nb_x, cap_x);
    ha[i].nb_e = (int32_t)std::min<int64_t>(ha[i].nb_e, 96);
    ha[i].nb_y = (int32_t)std::min<int64_t>(ha[i].nb_y, 4);
    ha[i].nb_a = (int32_t)std::min<int64_t>(ha[i].nb_a, 64);
    ha[i].nb_r = (int32_t)std::min<int64_t>(ha[i].nb_r, cap_x);
    gb.start[i] = (int32_t)blocks;
    blocks += ha[i].nb_x + ha[i].nb_a + ha[i].nb_r + ha[i].nb_e + ha[i].nb_y;
  }
  gb.start[n] = (int32_t)blocks;
  // a narrower access width than a batch was laid out for: its lanes-per-row geometry follows the common width
  for (int i = 0; i < n; ++i) {
    if (ha[i].nb_x == 0) continue;
    const int chunks = vec == kVecSpan ? ha[i].x_chunks : (int)(ha[i].x_row_bytes / vec);
    if (chunks != ha[i].x_chunks) {
      int lpr = 0;
      while ((1 << lpr) < chunks && lpr < 6) ++lpr;
      ha[i].x_chunks = chunks;
      ha[i].x_lpr_log2 = lpr;
    }
  }
  SPP_HIP_TRY(hipMemcpyAsync(s->dargs_dev[set], ha, sizeof(DeliverArgs) * (size_t)n, hipMemcpyHostToDevice, st));
  const int prof = prof_begin(SPP_PROF_GATHER, st, rows);
  const DeliverArgs* da = s->dargs_dev[set];
  if (asrc && asrc[0].p2p) {
    switch (vec) {
      case 16: hipLaunchKernelGGL((k_deliver_group<16, true>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 8: hipLaunchKernelGGL((k_deliver_group<8, true>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 4: hipLaunchKernelGGL((k_deliver_group<4, true>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 2: hipLaunchKernelGGL((k_deliver_group<2, true>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      default: hipLaunchKernelGGL((k_deliver_group<1, true>), dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
    }
  } else {
    switch (vec) {
      case kVecSpan: hipLaunchKernelGGL(k_deliver_group<kVecSpan>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 16: hipLaunchKernelGGL(k_deliver_group<16>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 8: hipLaunchKernelGGL(k_deliver_group<8>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 4: hipLaunchKernelGGL(k_deliver_group<4>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      case 2: hipLaunchKernelGGL(k_deliver_group<2>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
      default: hipLaunchKernelGGL(k_deliver_group<1>, dim3((unsigned)blocks), dim3(kGatherThreads), 0, st, da, gb); break;
    }
  }
  prof_end(SPP_PROF_GATHER, prof, st);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

void sampler_slot_parts(const spp_sampler* s, int slot, SlotParts* out) {
  const SlotHost& sl = s->slots[(size_t)slot];
  out->parts = sl.p.parts;
  out->pcnt = sl.host_state->pcnt;
  out->num_nodes = sl.host_state->cnt[s->cfg.num_hops];
  out->error = sl.host_state->error;
}

spp_status sampler_pack_remote_ids(spp_sampler* s, int first_slot, int n, const int64_t* pack_base_dev,
                                   int32_t* out_dev, hipStream_t st) {
  SPP_REQUIRE(s->part.P > 0, "sampler_pack_remote_ids: no ownership bucketing");
  SPP_REQUIRE(n >= 1 && first_slot >= 0 && first_slot + n <= (int)s->slots.size(), "sampler_pack_remote_ids: bad slots");
  const unsigned gx = (unsigned)std::max<int64_t>(1, ceil_div(s->tcap[s->cfg.num_hops], kNT));
  hipLaunchKernelGGL(k_pack_remote_ids, dim3(gx * (unsigned)n), dim3(kNT), 0, st, s->d_slots,
                     GroupGrid{(int32_t)first_slot, (int32_t)n, gx, s->xcd_affinity ? 1 : 0}, s->part.P, s->part.rank,
                     pack_base_dev, out_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

XBuf* sampler_xbuf(spp_sampler* s, int set) { return &s->xbuf[set % kMaxSets]; }

spp_status sampler_xbuf_grow(spp_sampler* s, void** buf, int64_t* cap, int64_t need, int64_t unit_bytes) {
  if (need <= *cap) return SPP_OK;
  const int64_t ncap = std::max(need + need / 2, *cap * 2);  // group sizes vary by a few %: do not grow twice
  void* v = nullptr;
  SPP_HIP_TRY(hipMalloc(&v, (size_t)(ncap * unit_bytes)));
  if (*buf) s->allocs.push_back(*buf);  // released by spp_sampler_destroy
  *buf = v;
  *cap = ncap;
  s->bytes += (ncap) * unit_bytes;
  return SPP_OK;
}

Worker* sampler_worker(spp_sampler* s, int which) {
  if (!s->workers[which]) s->workers[which].reset(new Worker());
  return s->workers[which].get();
}

spp_status sampler_xbuf_counts(spp_sampler* s, XBuf* xb, int64_t bytes) {
  (void)s;
  if (bytes > xb->cnt_bytes) {
    if (xb->cnt_dev) (void)hipFree(xb->cnt_dev);
    if (xb->cnt_host) (void)hipHostFree(xb->cnt_host);
    xb->cnt_dev = nullptr;
    xb->cnt_host = nullptr;
    xb->cnt_bytes = 0;
    SPP_HIP_TRY(hipMalloc((void**)&xb->cnt_dev, (size_t)bytes));
    SPP_HIP_TRY(hipHostMalloc((void**)&xb->cnt_host, (size_t)bytes, hipHostMallocDefault));
    xb->cnt_bytes = bytes;
  }
  if (!xb->cnt_ready) SPP_HIP_TRY(hipEventCreateWithFlags(&xb->cnt_ready, hipEventDisableTiming));
  if (!xb->rows_done) SPP_HIP_TRY(hipEventCreateWithFlags(&xb->rows_done, hipEventDisableTiming));
  return SPP_OK;
}

hipStream_t sampler_comm_stream(spp_sampler* s) {
  if (!s->comm_stream && create_data_stream(&s->comm_stream, false) != hipSuccess)
    s->comm_stream = nullptr;
  return s->comm_stream;
}

void sampler_poison_comm_stream(spp_sampler* s) { s->comm_stream = nullptr; }  // leaked on purpose

hipEvent_t sampler_export_event(spp_sampler* s, int slot) { return s->slots[(size_t)slot].exported; }

// Rebuilds the cache membership bits from the map on `st` (a Session calls this once, before its first
// chain: the map may have been rebuilt in place since the last Session) and returns the event that marks
// them ready; *ready == NULL: this sampler has no cache.
spp_status sampler_refresh_cache_bits(spp_sampler* s, hipStream_t st, hipEvent_t* ready) {
  *ready = nullptr;
  if (s->part.P <= 0 || !s->part.use_cache || !s->part.cache_map || s->part.cache_len <= 0) return SPP_OK;
  const int64_t nwords = (s->part.cache_len + 63) / 64;
  if (!s->cache_bits) {
    SPP_HIP_TRY(hipSetDevice(s->cfg.device));
    const int64_t ncoarse = (nwords + 63) / 64;  // the coarse level lives behind the fine one
    SPP_HIP_TRY(hipMalloc((void**)&s->cache_bits, sizeof(unsigned long long) * (size_t)(nwords + ncoarse)));
    s->bytes += (int64_t)sizeof(unsigned long long) * (nwords + ncoarse);
    s->part.cache_bits = reinterpret_cast<const uint32_t*>(s->cache_bits);
    s->part.cache_coarse = reinterpret_cast<const uint32_t*>(s->cache_bits + nwords);
  }
  if (!s->cache_bits_ready) SPP_HIP_TRY(hipEventCreateWithFlags(&s->cache_bits_ready, hipEventDisableTiming));
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(nwords, 256 / kWave), 256 * 16);
  hipLaunchKernelGGL(k_cache_bits, dim3(grid), dim3(256), 0, st, s->part.cache_map, s->part.cache_len, s->cache_bits);
  hipLaunchKernelGGL(k_cache_coarse, dim3((unsigned)std::min<int64_t>(ceil_div(ceil_div(nwords, 64), 256 / kWave), 1024)), dim3(256),
                     0, st, s->cache_bits, nwords, s->cache_bits + nwords);
  SPP_HIP_TRY(hipGetLastError());
  SPP_HIP_TRY(hipEventRecord(s->cache_bits_ready, st));
  *ready = s->cache_bits_ready;
  return SPP_OK;
}

hipEvent_t sampler_inputs_event(spp_sampler* s) {
  if (!s->inputs_ready && hipEventCreateWithFlags(&s->inputs_ready, hipEventDisableTiming) != hipSuccess)
    s->inputs_ready = nullptr;
  return s->inputs_ready;
}

hipEvent_t sampler_slot_event(const spp_sampler* s, int slot) {
  const SlotHost& sl = s->slots[(size_t)slot];
  return sl.sampled ? sl.wait_on : nullptr;
}

}  // namespace spp

extern "C" spp_status spp_sampler_sample(spp_sampler* s, int32_t slot, const int64_t* seeds_dev, int64_t n_seeds,
                                         uint32_t rng_seed, int64_t rng_skip, void* stream) {
  SPP_REQUIRE(s, "spp_sampler_sample: NULL sampler");
  SPP_REQUIRE(slot >= 0 && slot < (int32_t)s->slots.size(), "spp_sampler_sample: slot %d out of range", slot);
  SPP_REQUIRE(rng_skip >= 0, "spp_sampler_sample: rng_skip must be >= 0");
  hipStream_t st = as_stream(stream);
  SPP_TRY(sampler_launch_rng(s, slot, 1, 0, &rng_seed, &rng_skip, st));
  return sampler_launch_chain(s, slot, 1, 0, &seeds_dev, &n_seeds, st, nullptr);
}

static void fill_counts(const spp_sampler* s, const SlotState* hs, spp_mfg_counts* out) {
  const int H = s->cfg.num_hops;
  out->num_hops = H;
  out->num_seeds = hs->cnt[0];
  out->num_nodes = hs->cnt[H];
  out->draws = hs->dbase[H];
  for (int k = 0; k < H; ++k) {  // std::reverse(adjs) (fast_sampler.cpp:224)
    const int h = H - 1 - k;
    out->T[k] = hs->cnt[h];
    out->S[k] = hs->cnt[h + 1];
    out->E[k] = hs->E[h];
  }
  for (int m = 0; m < SPP_MAX_PARTS + 2; ++m) out->part_counts[m] = (m < s->part.P + 2 && s->part.P > 0) ? hs->pcnt[m] : 0;
}

extern "C" spp_status spp_sampler_wait(spp_sampler* s, int32_t slot, spp_mfg_counts* out) {
  SPP_REQUIRE(s && slot >= 0 && slot < (int32_t)s->slots.size(), "spp_sampler_wait: bad sampler/slot");
  SlotHost& sl = s->slots[slot];
  if (!sl.sampled) {
    set_error("spp_sampler_wait: slot %d has no batch in flight", slot);
    return SPP_ERR_STATE;
  }
  SPP_HIP_TRY(hipEventSynchronize(sl.wait_on));
  sl.waited = true;
  if (sl.host_state->error) {
    set_error("spp_sampler: batch exceeded the slot workspace (error mask %d: 1=edges 2=nodes 4=draws 8=dedup bucket)",
              sl.host_state->error);
    return SPP_ERR_CAPACITY;
  }
  if (out) fill_counts(s, sl.host_state, out);
  return SPP_OK;
}

extern "C" spp_status spp_sampler_export(spp_sampler* s, int32_t slot, const spp_mfg_out* out, void* stream) {
  SPP_REQUIRE(s && out && slot >= 0 && slot < (int32_t)s->slots.size(), "spp_sampler_export: bad argument");
  SlotHost& sl = s->slots[slot];
  if (!sl.sampled || !sl.waited) {
    set_error("spp_sampler_export: slot %d must be sampled and waited first", slot);
    return SPP_ERR_STATE;
  }
  const SlotState* hs = sl.host_state;
  const int H = s->cfg.num_hops;
  ExportSegs g{};
  int n = 0;
  int64_t total = 0;
  auto add = [&](const int32_t* src, int64_t* dst, int64_t len) {
    if (len <= 0 || dst == nullptr) return;
    g.src[n] = src;
    g.dst[n] = dst;
    g.start[n] = total;
    total += len;
    ++n;
  };
  add(sl.p.n_ids, out->n_id, hs->cnt[H]);
  for (int k = 0; k < H; ++k) {
    const int h = H - 1 - k;
    add(sl.p.out_rowptr[h], out->rowptr[k], (int64_t)hs->cnt[h] + 1);
    add(sl.p.out_col[h], out->col[k], hs->E[h]);
  }
  {
    const int64_t U = hs->cnt[H];
    if (s->part.P > 0) {
      int64_t owned = 0;
      for (int m = 0; m < s->part.P; ++m) owned += hs->pcnt[m];
      add(sl.p.parts, out->parts, owned);
      add(sl.p.pcached, out->cached, hs->pcnt[s->part.P]);
      add(sl.p.pperm, out->perm, U);
    }
  }
  g.n = n;
  g.start[n] = total;
  if (total == 0) return SPP_OK;
  const unsigned grid = (unsigned)std::min<int64_t>(ceil_div(total, kNT), 4096);
  hipLaunchKernelGGL(k_export, dim3(grid), dim3(kNT), 0, as_stream(stream), g);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

extern "C" spp_status spp_sampler_gather(spp_sampler* s, int32_t slot, const void* src_dev, int64_t src_rows,
                                         int64_t row_bytes, int64_t src_stride_bytes, int64_t n_rows, void* dst_dev,
                                         void* stream) {
  SPP_REQUIRE(s && slot >= 0 && slot < (int32_t)s->slots.size(), "spp_sampler_gather: bad sampler/slot");
  SlotHost& sl = s->slots[slot];
  if (!sl.sampled || !sl.waited) {
    set_error("spp_sampler_gather: slot %d must be sampled and waited first", slot);
    return SPP_ERR_STATE;
  }
  const int64_t U = sl.host_state->cnt[s->cfg.num_hops];
  const int64_t n = (n_rows < 0 || n_rows > U) ? U : n_rows;
  if (n == 0 || row_bytes == 0) return SPP_OK;
  SPP_REQUIRE(src_dev && dst_dev, "spp_sampler_gather: NULL buffer");
  SPP_REQUIRE(src_stride_bytes == 0 || src_stride_bytes >= row_bytes, "spp_sampler_gather: bad source stride");
  SPP_REQUIRE(src_rows > 0, "spp_sampler_gather: empty source table");
  return gather_rows_i32(src_dev, src_rows, row_bytes, src_stride_bytes, sl.p.n_ids, n, dst_dev, as_stream(stream));
}
