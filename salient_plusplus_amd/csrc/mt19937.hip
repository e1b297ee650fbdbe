// a1: spp_mt19937_fill -- raw outputs skip .. skip+n-1 of std::mt19937(seed) written to HBM by one
// workgroup (generator: mt19937.hip.h; reference: fast_sampler/sample_cpu.hpp:11, fast_sampler.cpp:994).
#include "spp_internal.h"

#include "mt19937.hip.h"

namespace spp {

__global__ __launch_bounds__(kMtThreads) void k_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out) {
  __shared__ uint32_t x[2 * kMtRing];
  mt_block_seed(x, seed, skip, n, out);
  mt_block_advance(x, 624, skip + n, skip, n, out);
}

}  // namespace spp

extern "C" spp_status spp_mt19937_fill(uint32_t seed, int64_t skip, int64_t n, uint32_t* out_dev, void* stream) {
  SPP_REQUIRE(n >= 0 && skip >= 0, "spp_mt19937_fill: n (%lld) and skip (%lld) must be >= 0", (long long)n,
              (long long)skip);
  if (n == 0) return SPP_OK;
  SPP_REQUIRE(out_dev != nullptr, "spp_mt19937_fill: out_dev is NULL");
  hipLaunchKernelGGL(spp::k_mt19937_fill, dim3(1), dim3(spp::kMtThreads), 0, spp::as_stream(stream), seed, skip, n, out_dev);
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}
