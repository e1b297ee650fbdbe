// Internal (non-ABI) interface between sampler.hip and session.hip: grouped launches.
#pragma once

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "spp_internal.h"

namespace spp {

// A host thread that outlives Sessions (owned by the pooled sampler): a Session per epoch used to
// start and join its own launcher thread, and the HIP runtime's per-thread set-up / teardown showed up
// as a ~75 ms stall of whichever thread was inside a HIP call every few epochs.
class Worker {
 public:
  ~Worker() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      quit_ = true;
    }
    cv_.notify_all();
    if (th_.joinable()) th_.join();
  }
  // run `fn` on the worker thread (one task at a time; waits for the previous one to finish first)
  void run(std::function<void()> fn) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
    task_ = std::move(fn);
    busy_ = true;
    if (!th_.joinable()) th_ = std::thread([this] { loop(); });
    lk.unlock();
    cv_.notify_all();
  }
  void wait_idle() {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [this] { return !busy_; });
  }

 private:
  void loop() {
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
      cv_.wait(lk, [this] { return quit_ || (busy_ && task_); });
      if (quit_) return;
      std::function<void()> fn = std::move(task_);
      task_ = nullptr;
      lk.unlock();
      fn();
      lk.lock();
      busy_ = false;
      cv_.notify_all();
    }
  }
  std::thread th_;
  std::mutex mu_;
  std::condition_variable cv_;
  std::function<void()> task_;
  bool busy_ = false, quit_ = false;
};

// the sampler's two persistent host threads: 0 = chain launcher, 1 = exchange issuer
Worker* sampler_worker(spp_sampler* s, int which);

#ifndef SPP_MAX_GROUP
#define SPP_MAX_GROUP 16
#endif
constexpr int kMaxGroup = SPP_MAX_GROUP;  // batches one launch can process (the grids are flattened: GroupGrid)
constexpr int kMaxWorkStreams = 4;  // sampling streams owned by a sampler (slot-sets share them round-robin)
constexpr int kMaxSets = 8;         // slot-sets (groups sampled, exchanged or waiting for the consumer) in flight

// largest group the sampler supports (1 when a hop takes the generic path)
int sampler_max_group(const spp_sampler* s);

// the sampler's i-th sampling stream (persistent; i is taken modulo kMaxWorkStreams)
hipStream_t sampler_work_stream(spp_sampler* s, int i);

// Generate the mt19937 streams of `n` batches (slots first_slot..first_slot+n-1) into RNG buffer
// `buf` (0/1) of their slots: draws [skip, skip + Dcap) of mt19937(seed).
spp_status sampler_launch_rng(spp_sampler* s, int first_slot, int n, int buf, const uint32_t* seeds,
                              const int64_t* skips, hipStream_t st);

// Enqueue the sampling chain of `n` batches whose RNG streams are in buffer `buf`; records the
// group's completion event (spp_sampler_wait on any slot of the group waits for it).
// rng_streams: NULL, or per batch the device pointer of its draws in the epoch arena (then `buf` is unused).
spp_status sampler_launch_chain(spp_sampler* s, int first_slot, int n, int buf, const int64_t* const* seeds_dev,
                                const int64_t* n_seeds, hipStream_t st, const uint32_t* const* rng_streams);

// The mt19937 streams of a whole epoch, kept by the sampler across Sessions (see sampler.hip).
// *base == NULL on return: no arena (over budget / allocation failed) -> generate per group.
spp_status sampler_rng_arena(spp_sampler* s, const uint32_t* seeds, int64_t nb, hipStream_t st, const uint32_t** base,
                             int64_t* stride, hipEvent_t* ready);

// Where the feature rows of a distributed batch come from (native exchange, session.hip): row r of
// x is row pperm[r] of the virtual concatenation [owner 0 | ... | owner P-1 | cache hits]; the
// rank-th segment is read from x_local, the cache segment from `cache`, segment m from
// recv + recv_base[m] rows (rows received from peer m for THIS batch).
struct AssembleSrc {
  const char* x_local;
  const char* recv;      // dense rows
  const char* cache;
  int64_t x_local_stride, cache_stride;  // bytes between rows
  int64_t recv_base[SPP_MAX_PARTS];
  // P2P transport (spp_exchange_cfg.peer_x_dev): segment m != rank is read in rank m's own partition -- row
  // (node id - offsets[m]) of peer[m], rows peer_stride bytes apart; recv / recv_base are unused
  bool p2p = false;
  const char* peer[SPP_MAX_PARTS] = {};
  int64_t peer_stride = 0;
};

// Fused delivery of the waited batch in `slot` to caller buffers in one launch on `st`:
// MFG widening (mfg may be NULL), x = x_src[n_id,:] (or, with `asrc`, assembled from the local
// partition / received rows / cache), y = y_src[n_id[:y_rows],:].
spp_status sampler_deliver(spp_sampler* s, int slot, const spp_mfg_out* mfg, const void* x_src, int64_t x_row_bytes,
                           int64_t x_src_stride, void* x_dst, const void* y_src, int64_t y_row_bytes, int64_t y_rows,
                           void* y_dst, const AssembleSrc* asrc, hipStream_t st);

// The `n` waited batches of slot-set `set` (slots first_slot..) in ONE launch; outs[i] / y_rows[i] / asrc[i] belong to
// batch i (asrc NULL: rows come from the one table x_src).
spp_status sampler_deliver_group(spp_sampler* s, int set, int first_slot, int n, const spp_group_out* outs,
                                 const void* x_src, int64_t x_row_bytes, int64_t x_src_stride, const void* y_src,
                                 int64_t y_row_bytes, const int64_t* y_rows, const AssembleSrc* asrc, hipStream_t st);

// Ownership buckets of the batch in `slot` (valid once the group's completion event has been
// synchronised): device arrays and the host mirror of the bucket sizes.
struct SlotParts {
  const int32_t* parts;    // node ids grouped by owner (device)
  const int32_t* pcnt;     // host: [P+2] bucket sizes
  int32_t num_nodes;
  int32_t error;
};
void sampler_slot_parts(const spp_sampler* s, int slot, SlotParts* out);
// Copy the ids this rank requests from its peers, for the `n` batches in slots first_slot.., into
// one buffer laid out peer-major: out[pack_base[i*P + m] + k] = k-th id of batch i owned by m
// (m != rank).  pack_base_dev: int64[n*P] in HBM.
spp_status sampler_pack_remote_ids(spp_sampler* s, int first_slot, int n, const int64_t* pack_base_dev,
                                   int32_t* out_dev, hipStream_t st);
// Growable exchange buffers of one slot-set.  They belong to the sampler so that they outlive the
// per-epoch Sessions of a pooled sampler (growing them costs hipMallocs of hundreds of MB).
struct XBuf {
  int32_t* send_ids = nullptr;  // node ids this rank requests (peer-major, then batch)
  int64_t send_ids_cap = 0;
  int32_t* recv_ids = nullptr;  // node ids the peers request from this rank (peer-major, then batch)
  int64_t recv_ids_cap = 0;
  char* send_rows = nullptr;    // their rows, same order
  int64_t send_rows_cap = 0;    // rows
  char* recv_rows = nullptr;    // rows received for this rank's batches (peer-major, then batch)
  int64_t recv_rows_cap = 0;    // rows
  int64_t row_bytes = 0;        // row size the row buffers were sized for
  // request-count staging of the set's exchange and its two events (kept here so that a Session per
  // epoch neither pins host memory nor creates HIP objects)
  int64_t* cnt_dev = nullptr;
  int64_t* cnt_host = nullptr;  // pinned mirror
  int64_t cnt_bytes = 0;
  hipEvent_t cnt_ready = nullptr;
  hipEvent_t rows_done = nullptr;
};
// make sure the set's count staging holds `bytes` bytes and its events exist
spp_status sampler_xbuf_counts(spp_sampler* s, XBuf* xb, int64_t bytes);
// the sampler's persistent stream for the native exchange; `poisoned` = the previous one can never drain
// (aborted communicator): forget it and hand out a fresh one next time
hipStream_t sampler_comm_stream(spp_sampler* s);
void sampler_poison_comm_stream(spp_sampler* s);
XBuf* sampler_xbuf(spp_sampler* s, int set);
// grow *buf (capacity *cap, in units of unit_bytes) to at least `need`; the outgrown buffer is kept
// until the sampler is destroyed (kernels in flight may still read it)
spp_status sampler_xbuf_grow(spp_sampler* s, void** buf, int64_t* cap, int64_t need, int64_t unit_bytes);
// Persistent per-slot "the consumer's copies out of this slot are done" events and one "session inputs
// are ready" event, owned by the sampler: a Session per epoch that created and destroyed ~35 HIP events
// saw its teardown take 35-80 ms every few epochs (the runtime recycling its signal pool).
hipEvent_t sampler_export_event(spp_sampler* s, int slot);
hipEvent_t sampler_inputs_event(spp_sampler* s);

// cache membership bits of the ownership bucketing, rebuilt from the cache map on `st`; *ready is the event to
// order the sampling streams after (NULL: no cache)
spp_status sampler_refresh_cache_bits(spp_sampler* s, hipStream_t st, hipEvent_t* ready);
// completion event of the group `slot` belongs to (NULL when nothing was sampled into it)
hipEvent_t sampler_slot_event(const spp_sampler* s, int slot);

}  // namespace spp
