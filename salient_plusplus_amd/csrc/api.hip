// libspp_hip.so: error plumbing + trivial entry points of include/spp.h.
#include "spp_internal.h"

#include <cstdlib>
#include <cstring>

#include <atomic>
#include <mutex>
#include <vector>

namespace spp {
static thread_local char g_err[1024] = "";

// ---- live kernel timing with HIP events on the launching stream (bench.py roofline) ----
struct ProfRec {
  hipEvent_t a, b;
  int64_t units;
  bool ended;   // b has been recorded (a chain is timed from the launcher thread: a reader may come in between)
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static int g_prof_every = 1;                 // time every n-th launch of a kind (spp_profile_enable(n))
static int64_t g_prof_seen[SPP_PROF_KINDS] = {};
static std::vector<ProfRec> g_prof[SPP_PROF_KINDS];
static int g_prof_gen = 0;   // bumped by spp_profile_enable: a prof_end that belongs to an earlier recording is dropped

int prof_begin(int kind, hipStream_t st, int64_t units) {
  if (!g_prof_on) return -1;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if ((g_prof_seen[kind]++ % (kind == SPP_PROF_CHAIN ? 1 : g_prof_every)) != 0) return -1;
  if (g_prof[kind].size() >= (1u << 16) - 1) return -1;
  ProfRec r{};
  if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
  r.units = units;
  (void)hipEventRecord(r.a, st);
  g_prof[kind].push_back(r);
  return ((g_prof_gen & 0x7fff) << 16) | ((int)g_prof[kind].size() - 1);
}

void prof_end(int kind, int token, hipStream_t st) {
  if (token < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if ((token >> 16) != (g_prof_gen & 0x7fff)) return;  // the recording it was started in has been cleared since
  const int idx = token & 0xffff;
  if (idx < (int)g_prof[kind].size() && hipEventRecord(g_prof[kind][idx].b, st) == hipSuccess) g_prof[kind][idx].ended = true;
}

// ---- run-time tuning knobs (spp_tune) ----
static std::atomic<int> g_gather_wg_per_cu{0};   // 0: not read from the environment yet
int gather_wg_per_cu() {
  int v = g_gather_wg_per_cu.load(std::memory_order_relaxed);
  if (v <= 0) {
    const char* e = getenv("SPP_GATHER_WG_PER_CU");
    v = e ? atoi(e) : 16;
    if (v < 1) v = 1;
    g_gather_wg_per_cu.store(v, std::memory_order_relaxed);
  }
  return v;
}

static std::atomic<int> g_gather_span{-1};       // -1: not read from the environment yet
bool gather_span_enabled() {
  int v = g_gather_span.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("SPP_GATHER_SPAN");
    v = (e && atoi(e) == 0) ? 0 : 1;
    g_gather_span.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

// ---- asynchronously detected data errors (spp_async_errors) ----
static std::mutex g_aerr_mu;
static int32_t* g_aerr[64] = {};

int32_t* async_err_word(int device) {
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(g_aerr_mu);
  if (!g_aerr[device]) {
    int prev = 0;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    int32_t* w = nullptr;
    // coherent + mapped: one pointer valid on host and device, device writes visible without a flush
    if (hipHostMalloc((void**)&w, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipSetDevice(prev);
      return nullptr;
    }
    *w = 0;
    g_aerr[device] = w;
    (void)hipSetDevice(prev);
  }
  return g_aerr[device];
}

int32_t* async_err_word_current() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess) return nullptr;
  return async_err_word(d);
}

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

int spp_async_errors(int device, int clear) {
  int32_t* w = spp::async_err_word(device);
  if (!w) {
    spp::set_error("spp_async_errors: no error word for device %d", device);
    return SPP_ERR_HIP;
  }
  const int32_t v = clear ? __atomic_exchange_n(w, 0, __ATOMIC_ACQ_REL) : __atomic_load_n(w, __ATOMIC_ACQUIRE);
  return (int)(v & 0x7fffffff);
}

int spp_tune(const char* key, int value) {
  if (!key) return SPP_ERR_INVALID;
  if (!strcmp(key, "gather_wg_per_cu")) {
    const int prev = spp::gather_wg_per_cu();
    if (value > 0) spp::g_gather_wg_per_cu.store(value, std::memory_order_relaxed);
    return prev;
  }
  if (!strcmp(key, "gather_span")) {  // 16-byte span form of the row gather for rows of 16k + 8 bytes (value < 0: query)
    const int prev = spp::gather_span_enabled() ? 1 : 0;
    if (value >= 0) spp::g_gather_span.store(value ? 1 : 0, std::memory_order_relaxed);
    return prev;
  }
  spp::set_error("spp_tune: unknown knob '%s'", key);
  return SPP_ERR_INVALID;
}

void spp_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(spp::g_prof_mu);
  for (auto& v : spp::g_prof) {
    for (auto& r : v) {
      (void)hipEventDestroy(r.a);
      (void)hipEventDestroy(r.b);
    }
    v.clear();
  }
  ++spp::g_prof_gen;
  spp::g_prof_on = on != 0;
  spp::g_prof_every = on > 1 ? on : 1;
  for (auto& c : spp::g_prof_seen) c = 0;
}

spp_status spp_profile_read(int kind, double* total_ms, int64_t* launches, int64_t* units) {
  SPP_REQUIRE(kind >= 0 && kind < SPP_PROF_KINDS, "spp_profile_read: kind %d out of range", kind);
  std::lock_guard<std::mutex> lk(spp::g_prof_mu);
  double ms = 0;
  int64_t n = 0, u = 0;
  for (auto& r : spp::g_prof[kind]) {
    if (!r.ended) continue;  // still being enqueued by another thread: not part of this reading
    float t = 0;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) {
      (void)hipGetLastError();  // do not leave the failure behind for the caller's next HIP check
      continue;
    }
    ms += t;
    u += r.units;
    ++n;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (units) *units = u;
  return SPP_OK;
}

// ---- cross-process mappings for the P2P transport (include/spp.h) ----
spp_status spp_ipc_export(const void* ptr_dev, void* handle_out, int64_t* offset_out) {
  static_assert(sizeof(hipIpcMemHandle_t) <= SPP_IPC_HANDLE_BYTES, "SPP_IPC_HANDLE_BYTES too small");
  SPP_REQUIRE(ptr_dev && handle_out && offset_out, "spp_ipc_export: NULL argument");
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  SPP_HIP_TRY(hipMemGetAddressRange(&base, &size, const_cast<void*>(ptr_dev)));
  hipIpcMemHandle_t h;
  SPP_HIP_TRY(hipIpcGetMemHandle(&h, base));
  memset(handle_out, 0, SPP_IPC_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  *offset_out = (int64_t)(static_cast<const char*>(ptr_dev) - static_cast<const char*>(base));
  return SPP_OK;
}

spp_status spp_ipc_open(const void* handle, int32_t device, void** base_out) {
  SPP_REQUIRE(handle && base_out, "spp_ipc_open: NULL argument");
  SPP_HIP_TRY(hipSetDevice(device));
  hipIpcMemHandle_t h;
  memcpy(&h, handle, sizeof(h));
  SPP_HIP_TRY(hipIpcOpenMemHandle(base_out, h, hipIpcMemLazyEnablePeerAccess));
  return SPP_OK;
}

spp_status spp_ipc_close(void* base) {
  if (!base) return SPP_OK;
  SPP_HIP_TRY(hipIpcCloseMemHandle(base));
  return SPP_OK;
}

// gen.seed(pair.second * 17 + 5)  (reference fast_sampler.cpp:994; pair.second is int32)
uint32_t spp_batch_seed(int32_t stop) { return (uint32_t)(stop * 17 + 5); }

}  // extern "C"
