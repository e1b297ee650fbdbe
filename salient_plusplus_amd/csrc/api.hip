// libspp_hip.so: error plumbing + trivial entry points of include/spp.h.
#include "spp_internal.h"

#include <cstring>

namespace spp {
static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace spp

extern "C" {

int spp_abi_version(void) { return SPP_ABI_VERSION; }

const char* spp_last_error(void) { return spp::g_err; }

int spp_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    spp::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}

// gen.seed(pair.second * 17 + 5)  (reference fast_sampler.cpp:994; pair.second is int32)
uint32_t spp_batch_seed(int32_t stop) { return (uint32_t)(stop * 17 + 5); }

}  // extern "C"
