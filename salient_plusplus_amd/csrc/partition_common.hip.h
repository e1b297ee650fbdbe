// Shared by partition.hip (standalone spp_partition_batch) and sampler.hip (bucketing fused into the
// sampling chain): RangePartitionBook ownership (reference range_partition_book.cpp:98-107).
#pragma once

#include <cstdint>

#include <hip/hip_runtime.h>

#include "spp.h"

namespace spp {

constexpr int kPartBuckets = SPP_MAX_PARTS + 2;  // P partitions + cache-hit bucket + host-local counter

struct Offsets {
  int32_t n;  // P + 1
  int64_t v[SPP_MAX_PARTS + 1];
};

// searchsorted(offsets, nid, right=True) - 1   (range_partition_book.cpp:98-100)
__device__ __forceinline__ int32_t owner_of(const Offsets& o, int64_t v) {
  int32_t c = 0;
  for (int32_t k = 0; k < o.n; ++k) c += (o.v[k] <= v) ? 1 : 0;
  return c - 1;
}

// bucket of a node in the concatenation [parts[0..P-1], cache hits] of the distributed worker
// branch (fast_sampler.cpp:1031-1107 without cache, :1108-1260 with cache)
__device__ __forceinline__ int32_t part_bucket_of(const Offsets& off, int32_t P, int32_t rank, int32_t use_cache,
                                                  const int32_t* cache_map, int64_t cache_len, int64_t v) {
  if (!use_cache) return owner_of(off, v);                          // :1063
  if (v >= off.v[rank] && v < off.v[rank + 1]) return rank;         // nid_is_local, :1216
  const int32_t m = (v >= 0 && v < cache_len) ? cache_map[v] : -1;
  if (m >= 0) return P;                                             // cache hits go last (:1243)
  return owner_of(off, v);                                          // :1202
}

}  // namespace spp
