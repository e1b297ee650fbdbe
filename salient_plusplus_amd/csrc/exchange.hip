// spp_comm: the point-to-point transport under the native remote-feature exchange (session.hip).
//
// Two implementations of one small interface (all-gather + grouped send/recv on a HIP stream):
//   * RcclTransport  -- RCCL over xGMI, one process per GPU.  librccl.so.1 is resolved with dlopen
//     at run time so that the copy PyTorch has already mapped is the one used (two RCCL instances
//     in a process would each bring up their own proxy threads and channel buffers), and so that
//     libspp_hip.so still loads on a machine without RCCL.
//   * LocalTransport -- `world` ranks inside ONE process (one host thread per rank) that copy
//     device-to-device after a host rendezvous.  It exists to exercise the exchange logic of
//     session.hip on a single GPU (RCCL refuses two ranks on one device); not a product path.
//
// Replaces the reference's use of torch.distributed.all_to_all for the feature exchange
// (fast_trainer/transferers.py:521, :709, :757).
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include <rccl/rccl.h>  // types only: every function is looked up with dlsym

#include "exchange_internal.h"

namespace spp {

// ----------------------------------------------------------------------------------------------
// RCCL, late bound
// ----------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // RTLD_NOLOAD first: reuse the library the process already holds (PyTorch's bundled copy)
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
      const char* e = dlerror();
      api.err = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
      return;
    }
    api.lib = h;
    bool ok = true;
    auto sym = [&](const char* name) -> void* {
      void* p = dlsym(h, name);
      if (!p) {
        ok = false;
        api.err = std::string("librccl.so.1 lacks ") + name;
      }
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
    api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
    api.Send = reinterpret_cast<decltype(api.Send)>(sym("ncclSend"));
    api.Recv = reinterpret_cast<decltype(api.Recv)>(sym("ncclRecv"));
    api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!ok) api.lib = nullptr;
    else api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(dlsym(h, "ncclCommAbort"));
  });
  return &api;
}

#define SPP_NCCL_TRY(expr)                                                                 \
  do {                                                                                     \
    ncclResult_t r_ = (expr);                                                              \
    if (r_ != ncclSuccess) {                                                               \
      set_error("%s failed: %s", #expr, rccl_api()->GetErrorString ? rccl_api()->GetErrorString(r_) : "?"); \
      return SPP_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

class RcclTransport final : public Transport {
 public:
  RcclTransport(ncclComm_t c, int rank, int world) : comm_(c), rank_(rank), world_(world) {}
  ~RcclTransport() override {
    if (comm_) (void)rccl_api()->CommDestroy(comm_);
  }
  void abort() override {
    // ncclCommAbort frees the communicator: nothing may use comm_ afterwards
    ncclComm_t c = comm_;
    comm_ = nullptr;
    if (c) (void)(rccl_api()->CommAbort ? rccl_api()->CommAbort(c) : rccl_api()->CommDestroy(c));
  }
  int rank() const override { return rank_; }
  int world() const override { return world_; }
  spp_status all_gather(const void* send, void* recv, size_t bytes, hipStream_t st) override {
    SPP_REQUIRE(comm_, "exchange communicator was aborted");
    SPP_NCCL_TRY(rccl_api()->AllGather(send, recv, bytes, ncclInt8, comm_, st));
    return SPP_OK;
  }
  spp_status group_begin() override {
    SPP_NCCL_TRY(rccl_api()->GroupStart());
    return SPP_OK;
  }
  spp_status send(const void* p, size_t bytes, int peer, hipStream_t st) override {
    SPP_REQUIRE(comm_, "exchange communicator was aborted");
    SPP_NCCL_TRY(rccl_api()->Send(p, bytes, ncclInt8, peer, comm_, st));
    return SPP_OK;
  }
  spp_status recv(void* p, size_t bytes, int peer, hipStream_t st) override {
    SPP_REQUIRE(comm_, "exchange communicator was aborted");
    SPP_NCCL_TRY(rccl_api()->Recv(p, bytes, ncclInt8, peer, comm_, st));
    return SPP_OK;
  }
  spp_status group_end(hipStream_t) override {
    SPP_NCCL_TRY(rccl_api()->GroupEnd());
    return SPP_OK;
  }

 private:
  ncclComm_t comm_;
  int rank_, world_;
};

// ----------------------------------------------------------------------------------------------
// in-process ranks (testing aid)
// ----------------------------------------------------------------------------------------------
struct LocalOp {
  const void* src;
  void* dst;
  size_t bytes;
  int peer;
};

struct LocalWorld {
  int world = 0;
  std::mutex mu;
  std::condition_variable cv;
  int arrived = 0;
  int64_t generation = 0;
  bool broken = false;
  std::vector<std::vector<LocalOp>> sends, recvs;  // per rank, posted for the current collective
  std::vector<const void*> ag_src;                 // all-gather sources
  std::vector<hipEvent_t> ready, done;             // per rank
  int alive = 0;

  // reusable barrier; returns false when a rank has left (destroyed / aborted its communicator) or
  // failed to arrive within the exchange timeout (the world is then broken for everybody)
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (broken) return false;
    const int64_t gen = generation;
    if (++arrived == world) {
      arrived = 0;
      ++generation;
      cv.notify_all();
      return true;
    }
    const auto limit = std::chrono::duration<double>(exchange_timeout_s());
    if (!cv.wait_for(lk, limit, [&] { return generation != gen || broken; })) {
      broken = true;
      timed_out = true;
      cv.notify_all();
    }
    return !broken;
  }
  bool timed_out = false;
};

class LocalTransport final : public Transport {
 public:
  LocalTransport(std::shared_ptr<LocalWorld> w, int rank) : w_(std::move(w)), rank_(rank) {}
  ~LocalTransport() override {
    std::lock_guard<std::mutex> lk(w_->mu);
    w_->broken = true;  // peers still inside a collective give up instead of waiting forever
    w_->cv.notify_all();
    if (--w_->alive == 0) {
      for (auto e : w_->ready) (void)hipEventDestroy(e);
      for (auto e : w_->done) (void)hipEventDestroy(e);
    }
  }
  int rank() const override { return rank_; }
  int world() const override { return w_->world; }

  spp_status all_gather(const void* send, void* recv, size_t bytes, hipStream_t st) override {
    SPP_HIP_TRY(hipEventRecord(w_->ready[(size_t)rank_], st));
    w_->ag_src[(size_t)rank_] = send;
    if (!w_->barrier()) return gone();
    for (int m = 0; m < w_->world; ++m) {
      SPP_HIP_TRY(hipStreamWaitEvent(st, w_->ready[(size_t)m], 0));
      SPP_HIP_TRY(hipMemcpyAsync(static_cast<char*>(recv) + (size_t)m * bytes, w_->ag_src[(size_t)m], bytes,
                                 hipMemcpyDeviceToDevice, st));
    }
    return finish(st);
  }
  spp_status group_begin() override {
    w_->sends[(size_t)rank_].clear();
    w_->recvs[(size_t)rank_].clear();
    return SPP_OK;
  }
  spp_status send(const void* p, size_t bytes, int peer, hipStream_t) override {
    w_->sends[(size_t)rank_].push_back({p, nullptr, bytes, peer});
    return SPP_OK;
  }
  spp_status recv(void* p, size_t bytes, int peer, hipStream_t) override {
    w_->recvs[(size_t)rank_].push_back({nullptr, p, bytes, peer});
    return SPP_OK;
  }
  spp_status group_end(hipStream_t st) override {
    SPP_HIP_TRY(hipEventRecord(w_->ready[(size_t)rank_], st));
    if (!w_->barrier()) return gone();
    // the k-th receive from peer m pairs with m's k-th send to this rank
    std::vector<size_t> cursor((size_t)w_->world, 0);
    for (const LocalOp& r : w_->recvs[(size_t)rank_]) {
      const auto& ps = w_->sends[(size_t)r.peer];
      size_t& k = cursor[(size_t)r.peer];
      while (k < ps.size() && ps[k].peer != rank_) ++k;
      if (k == ps.size() || ps[k].bytes != r.bytes) {
        set_error("local transport: receive of %zu bytes from rank %d has no matching send", r.bytes, r.peer);
        return SPP_ERR_STATE;
      }
      SPP_HIP_TRY(hipStreamWaitEvent(st, w_->ready[(size_t)r.peer], 0));
      if (r.bytes) SPP_HIP_TRY(hipMemcpyAsync(r.dst, ps[k].src, r.bytes, hipMemcpyDeviceToDevice, st));
      ++k;
    }
    return finish(st);
  }

 private:
  // senders may reuse their buffers only after every receiver's copy: wait for all `done` events
  spp_status finish(hipStream_t st) {
    SPP_HIP_TRY(hipEventRecord(w_->done[(size_t)rank_], st));
    if (!w_->barrier()) return gone();
    for (int m = 0; m < w_->world; ++m)
      if (m != rank_) SPP_HIP_TRY(hipStreamWaitEvent(st, w_->done[(size_t)m], 0));
    if (!w_->barrier()) return gone();  // op lists / events are reused by the next collective
    return SPP_OK;
  }
  void abort() override {
    std::lock_guard<std::mutex> lk(w_->mu);
    w_->broken = true;
    w_->cv.notify_all();
  }

 private:
  spp_status gone() {
    if (w_->timed_out)
      set_error("local transport: rank %d waited %.0f s for its peers (a rank is missing, or the ranks disagree on the "
                "batch sequence)", rank_, exchange_timeout_s());
    else
      set_error("local transport: a peer rank left the communicator");
    return SPP_ERR_STATE;
  }
  std::shared_ptr<LocalWorld> w_;
  int rank_;
};

}  // namespace
}  // namespace spp

using namespace spp;

struct spp_comm {
  std::unique_ptr<Transport> t;
};

namespace spp {
Transport* comm_transport(spp_comm* c) { return c ? c->t.get() : nullptr; }
}  // namespace spp

extern "C" spp_status spp_comm_unique_id(void* out_id) {
  SPP_REQUIRE(out_id, "spp_comm_unique_id: NULL argument");
  static_assert(sizeof(ncclUniqueId) == SPP_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
  RcclApi* api = rccl_api();
  if (!api->lib) {
    set_error("spp_comm_unique_id: %s", api->err.c_str());
    return SPP_ERR_HIP;
  }
  ncclUniqueId id;
  SPP_NCCL_TRY(api->GetUniqueId(&id));
  std::memcpy(out_id, &id, sizeof(id));
  return SPP_OK;
}

extern "C" spp_status spp_comm_create(const void* id, int32_t rank, int32_t world, int32_t device, spp_comm** out) {
  SPP_REQUIRE(id && out, "spp_comm_create: NULL argument");
  SPP_REQUIRE(world >= 1 && world <= SPP_MAX_PARTS && rank >= 0 && rank < world,
              "spp_comm_create: rank %d / world %d out of range", rank, world);
  RcclApi* api = rccl_api();
  if (!api->lib) {
    set_error("spp_comm_create: %s", api->err.c_str());
    return SPP_ERR_HIP;
  }
  SPP_REQUIRE(spp_device_count() > 0, "spp_comm_create: no HIP device available");
  SPP_HIP_TRY(hipSetDevice(device));
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t comm = nullptr;
  SPP_NCCL_TRY(api->CommInitRank(&comm, world, uid, rank));
  auto* c = new spp_comm();
  c->t.reset(new RcclTransport(comm, rank, world));
  *out = c;
  return SPP_OK;
}

extern "C" spp_status spp_comm_create_local(int32_t world, int32_t device, spp_comm** out) {
  SPP_REQUIRE(out && world >= 1 && world <= SPP_MAX_PARTS, "spp_comm_create_local: bad argument");
  {  // a rehearsal transport, compiled into the product library for the test suite only: opt-in
    const char* e = getenv("SPP_ALLOW_LOCAL_COMM");
    SPP_REQUIRE(e && atoi(e) != 0,
                "spp_comm_create_local: the in-process transport is a test aid (ranks as threads of one process on one "
                "GPU); set SPP_ALLOW_LOCAL_COMM=1 to use it -- production ranks use spp_comm_create (RCCL)");
  }
  SPP_REQUIRE(spp_device_count() > 0, "spp_comm_create_local: no HIP device available");
  SPP_HIP_TRY(hipSetDevice(device));
  auto w = std::make_shared<LocalWorld>();
  w->world = world;
  w->alive = world;
  w->sends.resize((size_t)world);
  w->recvs.resize((size_t)world);
  w->ag_src.assign((size_t)world, nullptr);
  w->ready.assign((size_t)world, nullptr);
  w->done.assign((size_t)world, nullptr);
  for (int m = 0; m < world; ++m) {
    SPP_HIP_TRY(hipEventCreateWithFlags(&w->ready[(size_t)m], hipEventDisableTiming));
    SPP_HIP_TRY(hipEventCreateWithFlags(&w->done[(size_t)m], hipEventDisableTiming));
  }
  for (int m = 0; m < world; ++m) {
    auto* c = new spp_comm();
    c->t.reset(new LocalTransport(w, m));
    out[m] = c;
  }
  return SPP_OK;
}

extern "C" void spp_comm_destroy(spp_comm* c) { delete c; }
extern "C" int32_t spp_comm_rank(const spp_comm* c) { return c && c->t ? c->t->rank() : -1; }
extern "C" int32_t spp_comm_world(const spp_comm* c) { return c && c->t ? c->t->world() : 0; }
