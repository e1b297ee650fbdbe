// a6/a7: Session runtime.  Replaces the reference's CPU worker pool (global ThreadPool,
// fast_sampler.cpp:398-453), per-epoch FastSamplerSession (:533-936), MPMC queues and the
// `items_in_queue` semaphore with batch slots kept in flight on HIP streams: producers are GPU
// streams, back-pressure is slot reuse, and the consumer only ever blocks on a HIP event.
// Python-visible semantics are preserved:
//   * batch ranges: fast_sampler.cpp:587-627 (plain / skip_nonfull / exact-count split);
//   * per-batch generator seed: gen.seed(range.second*17+5)  (fast_sampler.cpp:994);
//   * worker body (non-distributed): multilayer_sample -> x = x[n_id], y = y[n_id[:bs]]
//     (fast_sampler.cpp:998-1016);
//   * end of data is signalled by "no batch" (the reference returns None).
// Batches are delivered in index order (a valid completion order for the non-distributed path and
// the required order for the distributed one, fast_sampler.cpp:672-711).
//
// Scheduling.  Consecutive batches are sampled in GROUPS of G (one launch sequence per group, see
// sampler.hip); the `max_items_in_queue` slots form max_items/G slot-sets, each with its own HIP
// stream.  The mt19937 streams of a group are generated one slot-set generation AHEAD, at the tail
// of the same stream, into the idle half of a per-slot ping-pong buffer, so the serial generator
// never delays a sampling chain.  When the last batch of a group has been exported, its slot-set immediately
// starts the next pending group (ordered after the consumer's copies by events).
//
// Native feature exchange (distributed mode, spp_exchange_cfg).  A second session thread follows the
// launcher one stage behind: as soon as a group's chain has completed it reads the group's bucket
// sizes from the pinned state mirror and runs ONE exchange for the whole group on a stream of its
// own -- all-gather of the request counts (C1, transferers.py:757), grouped send/recv of the
// int32 node ids straight out of the slots (C2, :709), one gather of the requested rows out of the
// local partition (K5, :645-658), grouped send/recv of the rows (C3, :521).  The only host
// synchronisation is the one read of the gathered counts per group, off the consumer thread.
// spp_session_export orders the consumer's stream after the group's rows and assembles x in MFG
// order inside the delivery launch (sampler.hip k_deliver).
#include "spp_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <tuple>
#include <vector>

#include "exchange_internal.h"
#include "gather_body.hip.h"
#include "sampler_internal.h"

using namespace spp;

namespace spp {

// serve (transferers.py:645-658): send_rows[j,:] = x_local[ids[j] - rank_offset,:]
template <int VEC>
__global__ __launch_bounds__(kGatherThreads) void k_serve_rows(const char* __restrict__ x_local, int64_t x_rows,
                                                               const int32_t* __restrict__ ids, int64_t n,
                                                               int64_t rank_offset, int64_t row_bytes,
                                                               int64_t src_stride, int chunks, int lpr_log2,
                                                               char* __restrict__ out, int32_t* err) {
  move_rows_body<VEC, false>(
      [=](int64_t j) { return ids[j]; },
      [=](int32_t id) -> const char* {
        int64_t r = (int64_t)id - rank_offset;
        if ((uint64_t)r >= (uint64_t)x_rows) {
          // a peer only asks for rows this rank owns: anything else is a bucketing / partition-book
          // mismatch between the ranks -- serve row 0 (no fault) and report it (SPP_AERR_SERVE_ID)
          raise_async_error(err, SPP_AERR_SERVE_ID);
          r = 0;
        }
        return x_local + r * src_stride;
      },
      n, row_bytes, chunks, lpr_log2, out, blockIdx.x, gridDim.x);
}

}  // namespace spp

// exchange buffers of one slot-set
struct XSet {
  int64_t* cnt_dev = nullptr;   // [G*P] this rank's request counts, [G*P] pack_base, [world*G*P] everybody's counts
  int64_t* cnt_host = nullptr;  // pinned mirror, same layout
  XBuf* b = nullptr;            // growable id / row buffers, owned by the sampler (outlive the Session)
  hipEvent_t cnt_ready = nullptr;
  hipEvent_t rows_done = nullptr;
  bool rows_recorded = false;
  int64_t recv_base[kMaxGroup][SPP_MAX_PARTS];
};

struct spp_session {
  spp_session_cfg cfg{};
  spp_sampler* sampler = nullptr;
  bool owns_sampler = true;
  std::vector<std::pair<int32_t, int32_t>> ranges;
  int G = 1;                 // batches per group
  int num_sets = 1;          // slot-sets in flight
  int64_t num_groups = 0;    // groups in this epoch
  std::vector<hipStream_t> streams;      // one per slot-set (borrowed from the sampler)
  std::vector<hipEvent_t> export_done;   // per slot
  std::vector<char> export_recorded;     // per slot
  int64_t chain_launched = 0;            // groups whose sampling chain was launched (guarded by mu)
  int refill_lag = 0;                    // a freed slot-set is refilled this many groups later (see launcher_main)
  int64_t pending_consumed = -1;         // consumer thread only: groups fully consumed, not yet told to the launcher
  // Launcher thread: enqueues the ~40 kernel launches of a group's chain off the consumer thread
  // (they cost 0.2-0.5 ms of host time per group, which used to stall the consumer at every group
  // boundary).  It is the counterpart of the reference's worker threads, with the GPU doing the work.
  bool launcher_running = false;         // launcher_main is running on the sampler's worker thread 0
  std::mutex mu;
  std::condition_variable cv;
  int64_t groups_consumed = 0;           // groups fully consumed by the caller (guarded by mu)
  bool stop = false;
  spp_status launch_rc = SPP_OK;
  std::string launch_err;
  int64_t next_to_deliver = 0;           // batch index
  int32_t current_slot = -1;             // delivered by next(), not yet exported/recycled
  int64_t current_group = -1;            // delivered by next_group(), not yet (completely) exported
  int32_t group_member = 0;              // its members exported so far by per-batch spp_session_export calls
  int32_t open_slot = -1;                // last slot exported without an event of its own (mid-group), and its stream
  hipStream_t open_stream = nullptr;
  int64_t blocked_us = 0;
  int64_t blocked_occasions = 0;
  // native exchange (off when tr == nullptr)
  Transport* tr = nullptr;
  spp_exchange_cfg xcfg{};
  int P = 0, rank = 0;
  int64_t rank_offset = 0;
  hipStream_t comm_stream = nullptr;
  bool exchanger_running = false;        // exchanger_main is running on worker thread 1
  bool issue_on_consumer = false;        // exchanges are issued by the consumer thread at fixed program points
  int64_t exchange_launched = 0;         // groups whose exchange was enqueued (guarded by mu)
  spp_status exchange_rc = SPP_OK;
  std::string exchange_err;
  std::vector<XSet> xsets;
  std::atomic<int64_t> sent_bytes{0}, recv_bytes{0};
  bool comm_failed = false;              // an exchange timed out / a peer left: comm_stream may never drain
  // P2P transport (spp_exchange_cfg.peer_x_dev): no exchange at all -- the delivery reads remote rows in their owners' partitions
  bool p2p = false;
  const char* peer[SPP_MAX_PARTS] = {};
  int64_t peer_stride = 0;
  // epoch arena of mt19937 streams (sampler_rng_arena); NULL: per-group generation into the slots
  const uint32_t* rng_base = nullptr;
  int64_t rng_stride = 0;
};

// how long a rank waits for its peers inside an exchange before giving up with a message
// (SPP_EXCHANGE_TIMEOUT_S, default 300 s); read on every use so a caller may change it
namespace spp {
double exchange_timeout_s() {
  const char* e = getenv("SPP_EXCHANGE_TIMEOUT_S");
  const double v = e ? atof(e) : 300.0;
  return v > 0 ? v : 300.0;
}
}  // namespace spp

// Bounded wait for an event that completes only when every peer has taken part in a collective.
static spp_status wait_peers(spp_session* s, hipEvent_t ev, const char* what, long long g) {
  const double limit_s = exchange_timeout_s();
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return SPP_OK;
    if (q != hipErrorNotReady) SPP_HIP_TRY(q);
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s) {
      // The collective stays queued on comm_stream: abort the communicator so that its kernels leave
      // the device, and remember that the stream may never drain (teardown must not wait for it).
      s->comm_failed = true;
      if (s->tr) s->tr->abort();
      set_error("%s (group %lld): rank %d waited %.0f s for its peers (a rank is missing, or the ranks disagree on "
                "the batch sequence); the exchange communicator was aborted", what, g, s->rank, limit_s);
      return SPP_ERR_STATE;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
  }
}

// fast_sampler.cpp:587-627
static void build_ranges(const spp_session_cfg& c, std::vector<std::pair<int32_t, int32_t>>& out) {
  const int64_t n = c.n_idx;
  if (c.force_exact_num_batches) {
    const int64_t k = c.exact_num_batches;
    if (k <= 0) return;
    std::vector<uint64_t> bs((size_t)k);
    int64_t rem = n;
    const uint64_t avg = (uint64_t)(n / k) - 1;  // :595
    for (int64_t i = 0; i < k; ++i) {
      bs[(size_t)i] = avg;
      rem -= (int64_t)avg;
    }
    while (rem > 0) {  // :602-608 round-robin from index 0
      for (int64_t i = 0; i < k; ++i) {
        if (rem <= 0) break;
        bs[(size_t)i]++;
        rem--;
      }
    }
    uint64_t sum = 0;
    for (int64_t i = 0; i < k; ++i) {
      out.emplace_back((int32_t)sum, (int32_t)(sum + bs[(size_t)i]));
      sum += bs[(size_t)i];
    }
  } else {
    for (int64_t i = 0; i < n; i += c.batch_size) {  // :618-626
      const int64_t this_bs = std::min(n, i + c.batch_size) - i;
      if (c.skip_nonfull_batch && this_bs < c.batch_size) continue;
      out.emplace_back((int32_t)i, (int32_t)(i + this_bs));
    }
  }
}

static inline int group_len(const spp_session* s, int64_t g) {
  const int64_t nb = (int64_t)s->ranges.size();
  return (int)std::min<int64_t>(s->G, nb - g * s->G);
}

// mt19937 streams of group g into its RNG buffer, enqueued on the group's own slot-set stream
static spp_status launch_group_rng(spp_session* s, int64_t g) {
  const int set = (int)(g % s->num_sets);
  const int buf = (int)((g / s->num_sets) & 1);
  const int n = group_len(s, g);
  uint32_t seeds[kMaxGroup];
  for (int i = 0; i < n; ++i) seeds[i] = spp_batch_seed(s->ranges[(size_t)(g * s->G + i)].second);  // :994
  return sampler_launch_rng(s->sampler, set * s->G, n, buf, seeds, nullptr, s->streams[(size_t)set]);
}

// Stream of a slot-set, in order:  [rng(g) only for the set's first group] wait(exports of the
// previous group) -> chain(g) -> completion event -> rng(g + num_sets) into the other RNG buffer.
// The generator therefore runs in the stream's idle time while the consumer drains other groups;
// a separate RNG stream turned out to share a hardware queue with a chain and serialise behind it.
static spp_status launch_group_chain(spp_session* s, int64_t g) {
  const int set = (int)(g % s->num_sets);
  const int buf = (int)((g / s->num_sets) & 1);
  const int n = group_len(s, g);
  hipStream_t st = s->streams[(size_t)set];
  const int64_t* seeds[kMaxGroup];
  int64_t n_seeds[kMaxGroup];
  for (int i = 0; i < n; ++i) {
    const auto& r = s->ranges[(size_t)(g * s->G + i)];
    seeds[i] = s->cfg.idx_dev + r.first;
    n_seeds[i] = (int64_t)r.second - r.first;
  }
  const uint32_t* rng_streams[kMaxGroup];
  for (int i = 0; i < n; ++i) rng_streams[i] = s->rng_base ? s->rng_base + (g * s->G + i) * s->rng_stride : nullptr;
  if (!s->rng_base && g < s->num_sets) SPP_TRY(launch_group_rng(s, g));
  // the consumer's copies out of these slots (previous group of this slot-set) must be done
  for (int i = 0; i < s->G; ++i) {
    const size_t slot = (size_t)(set * s->G + i);
    if (s->export_recorded[slot]) {
      SPP_HIP_TRY(hipStreamWaitEvent(st, s->export_done[slot], 0));
      s->export_recorded[slot] = 0;
    }
  }
  // ... and the previous exchange out of these slots (its sends read the slots' id lists)
  if (s->tr && s->xsets[(size_t)set].rows_recorded) SPP_HIP_TRY(hipStreamWaitEvent(st, s->xsets[(size_t)set].rows_done, 0));
  SPP_TRY(sampler_launch_chain(s->sampler, set * s->G, n, buf, seeds, n_seeds, st, s->rng_base ? rng_streams : nullptr));
  if (!s->rng_base && g + s->num_sets < s->num_groups) SPP_TRY(launch_group_rng(s, g + s->num_sets));
  return SPP_OK;
}

// Launcher thread body: keep chains in flight for up to num_sets groups beyond the ones fully consumed.
// SPP_TRACE_LAUNCHER=1: host timestamps (us since the first) of "chain g enqueue begin/end" and of the consumer's
// "need group g" / "got group g", printed by spp_session_destroy -- who waits for whom
static bool trace_on() {
  static const bool on = getenv("SPP_TRACE_LAUNCHER") && atoi(getenv("SPP_TRACE_LAUNCHER")) != 0;
  return on;
}
static std::mutex g_trace_mu;
static std::vector<std::tuple<long long, char, long long>> g_trace;
static void trace_ev(char what, long long g) {
  if (!trace_on()) return;
  const long long t = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  std::lock_guard<std::mutex> lk(g_trace_mu);
  g_trace.emplace_back(t, what, g);
}
static void trace_dump() {
  if (!trace_on()) return;
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (g_trace.empty()) return;
  const long long t0 = std::get<0>(g_trace.front());
  for (auto& e : g_trace) fprintf(stderr, "[spp trace] %lld %c %lld\n", std::get<0>(e) - t0, std::get<1>(e), std::get<2>(e));
  g_trace.clear();
}

// how many chains may be in flight beyond the groups the consumer has finished: all slot-sets, minus the refill lag
static inline int64_t launch_window(const spp_session* s) { return (int64_t)s->num_sets - s->refill_lag; }

// Refill lag (round 4).  A chain enqueued the moment a slot-set is handed back waits, on the GPU, for that set's last
// delivery -- so it can only START once the deliveries before it are done, and a consumer that synchronises soon after
// (a short timed window, an epoch end, any host read) waits for the whole chain, alone on the GPU and latency bound
// (0.6-0.7 ms for 16 batches).  With a lag of one group the launcher refills the set that was handed back one group
// EARLIER: its deliveries completed long ago, the chain starts the moment it is enqueued -- which, for a consumer that
// runs ahead of the GPU, is while the deliveries of the current group are still queued -- and runs beside them.
static void launcher_main(spp_session* s) {
  (void)hipSetDevice(s->cfg.device);
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    s->cv.wait(lk, [s] {
      return s->stop || (s->launch_rc == SPP_OK && s->chain_launched < s->num_groups &&
                         s->chain_launched < s->groups_consumed + launch_window(s));
    });
    if (s->stop) return;
    const int64_t g = s->chain_launched;
    lk.unlock();
    trace_ev('L', g);
    const spp_status rc = launch_group_chain(s, g);  // touches only slots of group g's slot-set
    trace_ev('l', g);
    lk.lock();
    if (rc != SPP_OK) {
      s->launch_rc = rc;
      s->launch_err = spp_last_error();
    } else {
      s->chain_launched = g + 1;
    }
    s->cv.notify_all();
  }
}

// consumer side: a whole group has been consumed -> its slot-set may be reused
static void notify_group_consumed(spp_session* s, int64_t groups_fully_consumed) {
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->groups_consumed = groups_fully_consumed;
  }
  s->cv.notify_all();
}

// The launcher is told about a finished group when the consumer COMES BACK for more (the next spp_session_next* call),
// not at the export that finished it: a consumer that synchronises at a group boundary -- a timed window that ends there,
// an evaluation loop, an epoch end -- then does not find a freshly enqueued refill chain (0.6 ms alone on the GPU) in front
// of its synchronize; the chain is enqueued with its next request instead and runs beside the following deliveries.
static void defer_group_consumed(spp_session* s, int64_t groups_fully_consumed) { s->pending_consumed = groups_fully_consumed; }
static void flush_group_consumed(spp_session* s) {
  if (s->pending_consumed >= 0) {
    notify_group_consumed(s, s->pending_consumed);
    s->pending_consumed = -1;
  }
}

// consumer side: block until the chain of group g has been enqueued (its completion event exists)
static spp_status wait_group_launched(spp_session* s, int64_t g) {
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv.wait(lk, [s, g] { return s->chain_launched > g || s->launch_rc != SPP_OK; });
  if (s->chain_launched > g) return SPP_OK;
  set_error("%s", s->launch_err.c_str());
  return s->launch_rc;
}

// ---- native exchange -----------------------------------------------------------------------------
static spp_status launch_serve(spp_session* s, const int32_t* ids, int64_t n, char* out, hipStream_t st) {
  if (n <= 0) return SPP_OK;
  SPP_REQUIRE(s->xcfg.x_local_rows > 0, "exchange: peers request %lld rows but this rank owns none", (long long)n);
  int32_t* err = async_err_word(s->cfg.device);
  const GatherGeom gg = gather_geometry(s->xcfg.x_local_dev, out, s->xcfg.row_bytes, n, s->xcfg.x_local_stride_bytes,
                                        /*allow_span=*/true);
  const char* x = static_cast<const char*>(s->xcfg.x_local_dev);
#define SPP_SERVE(V)                                                                                              \
  hipLaunchKernelGGL(k_serve_rows<V>, dim3((unsigned)gg.grid), dim3(kGatherThreads), 0, st, x, s->xcfg.x_local_rows, \
                     ids, n, s->rank_offset, s->xcfg.row_bytes, s->xcfg.x_local_stride_bytes, gg.chunks, gg.lpr_log2, out, err)
  switch (gg.vec) {
    case kVecSpan: SPP_SERVE(kVecSpan); break;
    case 16: SPP_SERVE(16); break;
    case 8: SPP_SERVE(8); break;
    case 4: SPP_SERVE(4); break;
    case 2: SPP_SERVE(2); break;
    default: SPP_SERVE(1); break;
  }
#undef SPP_SERVE
  SPP_HIP_TRY(hipGetLastError());
  return SPP_OK;
}

// one exchange for all batches of group g (its chain has been launched)
static spp_status exchange_group(spp_session* s, int64_t g) {
  const int set = (int)(g % s->num_sets);
  const int n = group_len(s, g);
  const int P = s->P, R = s->rank, G = s->G;
  XSet& x = s->xsets[(size_t)set];
  hipStream_t st = s->comm_stream;
  Transport* tr = s->tr;
  const int64_t rb = s->xcfg.row_bytes;

  hipEvent_t chain_done = sampler_slot_event(s->sampler, set * G);
  SPP_REQUIRE(chain_done, "exchange: group %lld has no sampling chain in flight", (long long)g);
  SPP_HIP_TRY(hipEventSynchronize(chain_done));
  SlotParts sp[kMaxGroup];
  for (int i = 0; i < n; ++i) {
    sampler_slot_parts(s->sampler, set * G + i, &sp[i]);
    if (sp[i].error) {
      set_error("spp_sampler: batch exceeded the slot workspace (error mask %d)", sp[i].error);
      return SPP_ERR_CAPACITY;
    }
  }
  // C1: what this rank requests from every owner, per batch; everybody learns everybody's requests.
  // The same upload carries pack_base[i][m]: where batch i's ids for owner m go in the peer-major
  // send buffer -- and, since rows come back in request order, where its rows land in recv_rows.
  const size_t ge = (size_t)G * (size_t)P;
  int64_t* mine = x.cnt_host;            // [G*P] request counts
  int64_t* pbase = x.cnt_host + ge;      // [G*P] pack_base
  int64_t* all_host = x.cnt_host + 2 * ge;
  int64_t* all_dev = x.cnt_dev + 2 * ge;
  int64_t want_from[SPP_MAX_PARTS], in_base[SPP_MAX_PARTS];
  int64_t total_in = 0;
  for (size_t k = 0; k < 2 * ge; ++k) x.cnt_host[k] = 0;
  for (int m = 0; m < P; ++m) {
    in_base[m] = total_in;
    want_from[m] = 0;
    for (int i = 0; i < n; ++i) {
      const int64_t c = (m != R) ? sp[i].pcnt[m] : 0;
      mine[(size_t)i * P + m] = c;
      pbase[(size_t)i * P + m] = total_in;
      x.recv_base[i][m] = total_in;
      total_in += c;
      want_from[m] += c;
    }
  }
  SPP_HIP_TRY(hipMemcpyAsync(x.cnt_dev, x.cnt_host, 2 * ge * 8, hipMemcpyHostToDevice, st));
  SPP_TRY(tr->all_gather(x.cnt_dev, all_dev, ge * 8, st));
  SPP_HIP_TRY(hipMemcpyAsync(all_host, all_dev, ge * 8 * (size_t)P, hipMemcpyDeviceToHost, st));
  SPP_HIP_TRY(hipEventRecord(x.cnt_ready, st));
  // meanwhile: regroup the requested ids peer-major (needs only this rank's counts)
  XBuf& xb = *x.b;
  if (xb.row_bytes != rb) {  // the sampler was last used with another feature width: capacities are in rows
    xb.send_rows_cap = xb.send_rows_cap * xb.row_bytes / rb;
    xb.recv_rows_cap = xb.recv_rows_cap * xb.row_bytes / rb;
    xb.row_bytes = rb;
  }
  SPP_TRY(sampler_xbuf_grow(s->sampler, (void**)&xb.send_ids, &xb.send_ids_cap, total_in, 4));
  SPP_TRY(sampler_xbuf_grow(s->sampler, (void**)&xb.recv_rows, &xb.recv_rows_cap, total_in, rb));
  if (total_in > 0) SPP_TRY(sampler_pack_remote_ids(s->sampler, set * G, n, x.cnt_dev + ge, xb.send_ids, st));
  // wait for the gathered counts -- i.e. for every peer to reach this group.  A peer that never arrives
  // (crashed rank, unequal batch counts slipping past the creation-time check) would make this wait
  // forever: give up with a message after SPP_EXCHANGE_TIMEOUT_S seconds.
  SPP_TRY(wait_peers(s, x.cnt_ready, "exchange of request counts", (long long)g));
  // rows peer m wants from this rank (all batches of the group, in batch order)
  int64_t serve_for[SPP_MAX_PARTS], out_base[SPP_MAX_PARTS];
  int64_t total_req = 0;
  for (int m = 0; m < P; ++m) {
    out_base[m] = total_req;
    serve_for[m] = 0;
    if (m == R) continue;
    for (int i = 0; i < G; ++i) serve_for[m] += all_host[((size_t)m * G + i) * P + R];
    total_req += serve_for[m];
  }
  SPP_TRY(sampler_xbuf_grow(s->sampler, (void**)&xb.recv_ids, &xb.recv_ids_cap, total_req, 4));
  SPP_TRY(sampler_xbuf_grow(s->sampler, (void**)&xb.send_rows, &xb.send_rows_cap, total_req, rb));

  // C2: node ids, int32 -- one send and one receive per peer
  SPP_TRY(tr->group_begin());
  for (int m = 0; m < P; ++m) {
    if (m == R) continue;
    if (want_from[m] > 0) SPP_TRY(tr->send(xb.send_ids + in_base[m], (size_t)want_from[m] * 4, m, st));
    if (serve_for[m] > 0) SPP_TRY(tr->recv(xb.recv_ids + out_base[m], (size_t)serve_for[m] * 4, m, st));
  }
  SPP_TRY(tr->group_end(st));
  // K5: one gather of every requested row
  SPP_TRY(launch_serve(s, xb.recv_ids, total_req, xb.send_rows, st));
  // C3: the rows, again one send and one receive per peer
  SPP_TRY(tr->group_begin());
  for (int m = 0; m < P; ++m) {
    if (m == R) continue;
    if (serve_for[m] > 0) SPP_TRY(tr->send(xb.send_rows + out_base[m] * rb, (size_t)(serve_for[m] * rb), m, st));
    if (want_from[m] > 0) SPP_TRY(tr->recv(xb.recv_rows + in_base[m] * rb, (size_t)(want_from[m] * rb), m, st));
  }
  SPP_TRY(tr->group_end(st));
  const size_t cnt_elems = ge;
  SPP_HIP_TRY(hipEventRecord(x.rows_done, st));
  x.rows_recorded = true;
  s->sent_bytes += total_req * rb + total_in * 4 + (int64_t)cnt_elems * 8;
  s->recv_bytes += total_in * rb + total_req * 4 + (int64_t)cnt_elems * 8 * P;
  return SPP_OK;
}

static void exchanger_main(spp_session* s) {
  (void)hipSetDevice(s->cfg.device);
  for (int64_t g = 0; g < s->num_groups; ++g) {
    {
      std::unique_lock<std::mutex> lk(s->mu);
      s->cv.wait(lk, [s, g] { return s->stop || s->launch_rc != SPP_OK || s->chain_launched > g; });
      if (s->stop || s->chain_launched <= g) return;
    }
    const spp_status rc = exchange_group(s, g);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      if (rc != SPP_OK) {
        s->exchange_rc = rc;
        s->exchange_err = spp_last_error();
      } else {
        s->exchange_launched = g + 1;
      }
    }
    s->cv.notify_all();
    if (rc != SPP_OK) return;
  }
}

static spp_status wait_group_exchanged(spp_session* s, int64_t g) {
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv.wait(lk, [s, g] { return s->exchange_launched > g || s->exchange_rc != SPP_OK || s->launch_rc != SPP_OK; });
  if (s->exchange_launched > g) return SPP_OK;
  if (s->exchange_rc != SPP_OK) {
    set_error("%s", s->exchange_err.c_str());
    return s->exchange_rc;
  }
  set_error("%s", s->launch_err.c_str());
  return s->launch_rc;
}

static spp_status exchange_setup(spp_session* s, const spp_exchange_cfg* xc, const spp_partition_cfg& part) {
  const bool p2p = xc->peer_x_dev != nullptr;
  Transport* tr = p2p ? nullptr : comm_transport(xc->comm);
  SPP_REQUIRE(tr || p2p, "spp_session_create: exchange without a communicator");
  SPP_REQUIRE(part.num_parts > 0, "spp_session_create: the native exchange needs spp_partition_cfg");
  SPP_REQUIRE(p2p || (tr->world() == part.num_parts && tr->rank() == part.rank),
              "spp_session_create: communicator is rank %d of %d but the partition book says %d of %d", tr->rank(),
              tr->world(), part.rank, part.num_parts);
  SPP_REQUIRE(xc->row_bytes > 0 && (xc->x_local_dev || xc->x_local_rows == 0), "spp_session_create: bad x_local");
  SPP_REQUIRE(xc->x_local_rows >= part.offsets[part.rank + 1] - part.offsets[part.rank],
              "spp_session_create: x_local holds %lld rows, the partition owns %lld", (long long)xc->x_local_rows,
              (long long)(part.offsets[part.rank + 1] - part.offsets[part.rank]));
  SPP_REQUIRE(!part.use_cache || xc->cache_feats_dev || xc->cache_rows == 0,
              "spp_session_create: use_cache without cache rows");
  s->xcfg = *xc;
  s->issue_on_consumer = xc->issue_on_consumer != 0;
  if (s->xcfg.x_local_stride_bytes <= 0) s->xcfg.x_local_stride_bytes = xc->row_bytes;
  if (s->xcfg.cache_stride_bytes <= 0) s->xcfg.cache_stride_bytes = xc->row_bytes;
  SPP_REQUIRE(s->xcfg.x_local_stride_bytes >= xc->row_bytes && s->xcfg.cache_stride_bytes >= xc->row_bytes,
              "spp_session_create: row strides must be at least row_bytes");
  s->P = part.num_parts;
  s->rank = part.rank;
  s->rank_offset = part.offsets[part.rank];
  if (p2p) {
    // nothing to exchange: every remote row is read where it lives.  A rank that owns rows must be given their table.
    s->peer_stride = xc->peer_x_stride_bytes > 0 ? xc->peer_x_stride_bytes : s->xcfg.x_local_stride_bytes;
    SPP_REQUIRE(s->peer_stride >= xc->row_bytes, "spp_session_create: peer row stride smaller than the row");
    for (int m = 0; m < s->P; ++m) {
      s->peer[m] = static_cast<const char*>(xc->peer_x_dev[m]);
      SPP_REQUIRE(m == s->rank || s->peer[m] || part.offsets[m + 1] == part.offsets[m],
                  "spp_session_create: P2P transport without the table of rank %d", m);
    }
    s->p2p = true;
    return SPP_OK;
  }
  s->comm_stream = sampler_comm_stream(s->sampler);  // persistent, owned by the (pooled) sampler
  SPP_REQUIRE(s->comm_stream, "spp_session_create: no stream for the exchange");
  s->xsets.resize((size_t)s->num_sets);
  const size_t cnt_elems = (size_t)s->G * (size_t)s->P;
  for (size_t k = 0; k < s->xsets.size(); ++k) s->xsets[k].b = sampler_xbuf(s->sampler, (int)k);
  for (auto& x : s->xsets) {
    const size_t cnt_bytes = 8 * (cnt_elems * (size_t)(s->P + 2) + 2 * (size_t)s->P + 2);  // + the creation-time check
    SPP_TRY(sampler_xbuf_counts(s->sampler, x.b, (int64_t)cnt_bytes));
    x.cnt_dev = x.b->cnt_dev;
    x.cnt_host = x.b->cnt_host;
    x.cnt_ready = x.b->cnt_ready;
    x.rows_done = x.b->rows_done;
  }
  // Collective sanity check: the exchange is one collective sequence per group, so every rank must
  // run the same number of batches in groups of the same size (force_exact_num_batches in the
  // reference's distributed mode); a mismatch would otherwise show up as a hang.
  {
    XSet& x0 = s->xsets[0];
    x0.cnt_host[0] = (int64_t)s->ranges.size();
    x0.cnt_host[1] = s->G;
    SPP_HIP_TRY(hipMemcpyAsync(x0.cnt_dev, x0.cnt_host, 16, hipMemcpyHostToDevice, s->comm_stream));
    SPP_TRY(tr->all_gather(x0.cnt_dev, x0.cnt_dev + 2, 16, s->comm_stream));
    SPP_HIP_TRY(hipMemcpyAsync(x0.cnt_host + 2, x0.cnt_dev + 2, 16 * (size_t)s->P, hipMemcpyDeviceToHost, s->comm_stream));
    SPP_HIP_TRY(hipEventRecord(x0.cnt_ready, s->comm_stream));
    s->tr = tr;  // wait_peers aborts it on a timeout
    const spp_status wrc = wait_peers(s, x0.cnt_ready, "session creation (batch-count check)", -1);
    s->tr = nullptr;
    SPP_TRY(wrc);
    for (int m = 0; m < s->P; ++m)
      SPP_REQUIRE(x0.cnt_host[2 + 2 * m] == x0.cnt_host[0] && x0.cnt_host[3 + 2 * m] == x0.cnt_host[1],
                  "spp_session_create: rank %d runs %lld batches in groups of %lld, rank %d runs %lld in groups of %lld "
                  "(every rank must run the same number of batches)",
                  s->rank, (long long)x0.cnt_host[0], (long long)x0.cnt_host[1], m, (long long)x0.cnt_host[2 + 2 * m],
                  (long long)x0.cnt_host[3 + 2 * m]);
  }
  s->tr = tr;
  return SPP_OK;
}

static void exchange_teardown(spp_session* s) {
  if (s->comm_failed) {
    // The communicator was aborted after a timeout.  Give its queued work a moment to leave the
    // stream; if it does not, LEAK the stream, its events and buffers (hipFree / hipStreamDestroy would
    // wait for it forever) -- the caller is about to report the error and exit.
    bool drained = false;
    const auto t0 = std::chrono::steady_clock::now();
    while (s->comm_stream && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 5.0) {
      if (hipStreamQuery(s->comm_stream) != hipErrorNotReady) {
        drained = true;
        break;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (!drained) {
      sampler_poison_comm_stream(s->sampler);  // a later Session gets a fresh stream; this one is leaked
      s->comm_stream = nullptr;
      s->xsets.clear();
      return;
    }
  }
  // stream, count staging and events belong to the sampler: drain, keep
  if (s->comm_stream) (void)hipStreamSynchronize(s->comm_stream);
}

extern "C" spp_status spp_session_create(const spp_session_cfg* cfg, spp_session** out) {
  SPP_REQUIRE(cfg && out, "spp_session_create: NULL argument");
  SPP_REQUIRE(cfg->max_items_in_queue > 0, "max_items_in_queue (%d) must be positive", cfg->max_items_in_queue);
  SPP_REQUIRE(cfg->batch_size > 0 || cfg->force_exact_num_batches, "spp_session_create: batch_size must be > 0");
  SPP_REQUIRE(cfg->n_idx >= 0 && (cfg->idx_dev || cfg->n_idx == 0), "spp_session_create: bad idx");
  if (cfg->force_exact_num_batches)
    SPP_REQUIRE(cfg->exact_num_batches > 0 && cfg->n_idx / cfg->exact_num_batches >= 1,
                "spp_session_create: exact_num_batches (%lld) needs n_idx/exact_num_batches >= 1 (n_idx %lld)",
                (long long)cfg->exact_num_batches, (long long)cfg->n_idx);
  auto* s = new spp_session();
  s->cfg = *cfg;
  s->cfg.part = nullptr;  // caller-owned; only read here
  s->cfg.exchange = nullptr;
  spp_partition_cfg want{};
  if (cfg->part) want = *cfg->part;
  build_ranges(*cfg, s->ranges);
  int64_t max_batch = 1;
  for (auto& r : s->ranges) max_batch = std::max<int64_t>(max_batch, r.second - r.first);
  const int64_t nb = (int64_t)s->ranges.size();
  const int M = (int)std::max<int64_t>(1, std::min<int64_t>(cfg->max_items_in_queue, std::max<int64_t>(nb, 1)));

  bool generic = false;
  for (int h = 0; h < cfg->num_hops; ++h) generic |= (cfg->sizes[h] < 0 || cfg->sizes[h] > 32);
  // auto: four slot-sets of up to 16 batches when 32 or more slots are allowed (64 slots = 4 x 16 measured best on
  // MI355X at papers and products scale: the small hops of a chain are latency bound, so a launch over 16 batches
  // costs little more than one over 8 -- profiles/r03_ab_INDEX.md), two sets of up to 8 below that
  int G = cfg->group_size > 0 ? cfg->group_size : (M >= 32 ? std::min(16, M / 4) : std::max(1, std::min(8, M / 2)));
  G = std::min(G, std::min(M, kMaxGroup));
  if (generic) G = 1;
  int sets = std::max(1, std::min(M / G, kMaxSets));

  spp_status rc = SPP_OK;
  if (cfg->sampler) {
    spp_sampler_cfg have{};
    rc = spp_sampler_get_cfg(cfg->sampler, &have);
    bool ok = rc == SPP_OK && have.rowptr_dev == cfg->rowptr_dev && have.col_dev == cfg->col_dev &&
              have.num_hops == cfg->num_hops && have.max_batch >= max_batch && have.device == cfg->device &&
              have.replace == 0 && have.num_slots >= 1;
    for (int h = 0; ok && h < cfg->num_hops; ++h) ok = have.sizes[h] == cfg->sizes[h];
    ok = ok && have.part.num_parts == want.num_parts;
    if (ok && want.num_parts > 0) {
      ok = have.part.rank == want.rank && (have.part.use_cache != 0) == (want.use_cache != 0) &&
           (!want.use_cache ||
            (have.part.cache_map_dev == want.cache_map_dev && have.part.cache_map_len == want.cache_map_len));
      for (int m = 0; ok && m <= want.num_parts; ++m) ok = have.part.offsets[m] == want.offsets[m];
    }
    if (!ok) {
      set_error("spp_session_create: borrowed sampler is not compatible with this session's graph/fanouts/batch/partitioning");
      delete s;
      return SPP_ERR_INVALID;
    }
    s->sampler = cfg->sampler;
    s->owns_sampler = false;
    G = std::min(G, have.num_slots);
    sets = std::max(1, std::min(sets, have.num_slots / G));
  } else {
    spp_sampler_cfg sc{};
    sc.rowptr_dev = cfg->rowptr_dev;
    sc.col_dev = cfg->col_dev;
    sc.num_nodes = cfg->num_nodes;
    sc.nnz = cfg->nnz;
    sc.num_hops = cfg->num_hops;
    for (int h = 0; h < SPP_MAX_HOPS; ++h) sc.sizes[h] = cfg->sizes[h];
    sc.max_batch = max_batch;
    sc.num_slots = G * sets;
    sc.device = cfg->device;
    sc.part = want;
    rc = spp_sampler_create(&sc, &s->sampler);
    if (rc != SPP_OK) {
      delete s;
      return rc;
    }
  }
  s->G = G;
  s->num_sets = sets;
  {  // SPP_REFILL_LAG (groups; default 1): only with three or more slot-sets -- with two a lag would serialise.
     // Measured on S-papers (profiles/r04_ab_refill_lag.txt): 20-step windows 0.1363-0.1391 -> 0.1299-0.1305 ms per
     // step, 192-step windows 0.124-0.129 -> 0.122-0.124; a lag of 2, or a fifth slot-set for the lag, gives no more.
    const char* e = getenv("SPP_REFILL_LAG");
    const int lag = e ? atoi(e) : 1;
    s->refill_lag = (sets >= 3 && lag > 0) ? std::min(lag, sets - 2) : 0;
  }
  s->num_groups = (nb + G - 1) / G;

  s->streams.assign((size_t)sets, nullptr);
  for (int i = 0; i < sets; ++i) s->streams[(size_t)i] = sampler_work_stream(s->sampler, i);
  // the events live in the (pooled) sampler: a Session creates no HIP objects of its own
  s->export_done.assign((size_t)(sets * G), nullptr);
  s->export_recorded.assign((size_t)(sets * G), 0);
  for (size_t k = 0; k < s->export_done.size(); ++k) s->export_done[k] = sampler_export_event(s->sampler, (int)k);
  if (rc == SPP_OK && cfg->exchange) rc = exchange_setup(s, cfg->exchange, want);
  // Order every stream the session launches on after the producer of its device inputs (seed ids
  // written by a shuffle kernel still queued on the caller's stream, a cache map just built, ...).
  if (rc == SPP_OK && cfg->order_after_input_stream) {
    hipEvent_t inputs_ready = sampler_inputs_event(s->sampler);
    if (!inputs_ready || hipEventRecord(inputs_ready, as_stream(cfg->input_stream)) != hipSuccess) {
      set_error("spp_session_create: recording the input event failed");
      rc = SPP_ERR_HIP;
    }
    auto after_inputs = [&](hipStream_t st) {
      if (rc == SPP_OK && hipStreamWaitEvent(st, inputs_ready, 0) != hipSuccess) {
        set_error("spp_session_create: hipStreamWaitEvent failed");
        rc = SPP_ERR_HIP;
      }
    };
    for (auto st : s->streams) after_inputs(st);
    after_inputs(as_stream(spp_sampler_deliver_stream(s->sampler)));
    if (s->comm_stream) after_inputs(s->comm_stream);
  }
  // cache membership bits (ownership bucketing with a cache): rebuilt from the map at every Session start,
  // on the first sampling stream; the others wait for them
  if (rc == SPP_OK) {
    hipEvent_t bits_ready = nullptr;
    rc = sampler_refresh_cache_bits(s->sampler, s->streams[0], &bits_ready);
    if (rc == SPP_OK && bits_ready)
      for (size_t k = 1; k < s->streams.size(); ++k)
        if (hipStreamWaitEvent(s->streams[k], bits_ready, 0) != hipSuccess) {
          set_error("spp_session_create: hipStreamWaitEvent failed");
          rc = SPP_ERR_HIP;
        }
  }
  // mt19937 streams of the whole epoch: generated once per range table, kept by the (pooled) sampler
  if (rc == SPP_OK && nb > 0) {
    std::vector<uint32_t> seeds((size_t)nb);
    for (int64_t b = 0; b < nb; ++b) seeds[(size_t)b] = spp_batch_seed(s->ranges[(size_t)b].second);  // :994
    hipEvent_t arena_ready = nullptr;
    rc = sampler_rng_arena(s->sampler, seeds.data(), nb, s->streams[0], &s->rng_base, &s->rng_stride, &arena_ready);
    if (rc == SPP_OK && s->rng_base && arena_ready)
      for (size_t i = 1; i < s->streams.size() && rc == SPP_OK; ++i)
        if (hipStreamWaitEvent(s->streams[i], arena_ready, 0) != hipSuccess) {
          set_error("spp_session_create: hipStreamWaitEvent failed");
          rc = SPP_ERR_HIP;
        }
  }
  if (rc == SPP_OK) {
    // both run on host threads that belong to the (pooled) sampler and outlive this Session
    sampler_worker(s->sampler, 0)->run([s] { launcher_main(s); });  // primes the pipeline right away
    s->launcher_running = true;
    if (s->tr && !s->issue_on_consumer) {
      sampler_worker(s->sampler, 1)->run([s] { exchanger_main(s); });
      s->exchanger_running = true;
    }
    if (s->num_groups > 0) rc = wait_group_launched(s, 0);
  }
  if (rc != SPP_OK) {
    spp_session_destroy(s);
    return rc;
  }
  *out = s;
  return SPP_OK;
}

extern "C" void spp_session_destroy(spp_session* s) {
  trace_dump();
  if (!s) return;
  if (s->launcher_running || s->exchanger_running) {
    {
      std::lock_guard<std::mutex> lk(s->mu);
      s->stop = true;
    }
    s->cv.notify_all();
    if (s->launcher_running) sampler_worker(s->sampler, 0)->wait_idle();
    if (s->exchanger_running) sampler_worker(s->sampler, 1)->wait_idle();  // it leaves after the group in hand
  }
  (void)hipSetDevice(s->cfg.device);
  const bool had_comm_stream = s->comm_stream != nullptr;
  exchange_teardown(s);
  if (s->comm_failed && had_comm_stream && !s->comm_stream) {
    // the aborted exchange never drained: the sampling streams wait on its events, so nothing here
    // can be synchronised or freed without hanging -- leak the device side, the caller is failing
    delete s;
    return;
  }
  if (s->open_slot >= 0) {  // deliveries of a partly consumed group: they read the slots until they complete
    (void)hipStreamSynchronize(s->open_stream);
    s->open_slot = -1;
  }
  for (auto st : s->streams)
    if (st) (void)hipStreamSynchronize(st);
  // the consumer's last deliveries still read the slots and the exchange buffers: a (pooled) sampler
  // must be quiescent before another Session samples into it
  for (size_t k = 0; k < s->export_done.size(); ++k)
    if (s->export_done[k] && s->export_recorded[k]) (void)hipEventSynchronize(s->export_done[k]);
  if (s->sampler && s->owns_sampler) spp_sampler_destroy(s->sampler);
  delete s;
}

extern "C" int64_t spp_session_num_total_batches(const spp_session* s) { return s ? (int64_t)s->ranges.size() : 0; }
extern "C" int64_t spp_session_num_consumed_batches(const spp_session* s) { return s ? s->next_to_deliver : 0; }
extern "C" int64_t spp_session_blocked_us(const spp_session* s) { return s ? s->blocked_us : 0; }
extern "C" int64_t spp_session_blocked_occasions(const spp_session* s) { return s ? s->blocked_occasions : 0; }
extern "C" spp_sampler* spp_session_sampler(spp_session* s) { return s ? s->sampler : nullptr; }
extern "C" int32_t spp_session_group_size(const spp_session* s) { return s ? s->G : 0; }

extern "C" spp_status spp_session_batch_ranges(const spp_session* s, int32_t* out) {
  SPP_REQUIRE(s && out, "spp_session_batch_ranges: NULL argument");
  for (size_t i = 0; i < s->ranges.size(); ++i) {
    out[2 * i] = s->ranges[i].first;
    out[2 * i + 1] = s->ranges[i].second;
  }
  return SPP_OK;
}

// the batch in `current_slot` is done with (exported or dropped): recycle its slot-set once the
// whole group has been consumed
static spp_status retire_current(spp_session* s) {
  const int64_t b = s->next_to_deliver - 1;
  s->current_slot = -1;
  const int64_t g = b / s->G;
  const bool last_of_group = (b + 1 == (int64_t)s->ranges.size()) || ((b + 1) % s->G == 0);
  if (last_of_group) defer_group_consumed(s, g + 1);
  return SPP_OK;
}

// Consumer-issued exchanges (spp_exchange_cfg.issue_on_consumer): the exchange of group g is issued
// from spp_session_next at a fixed point of the program -- when the consumer reaches the middle of
// group g-1 (or needs group g right now) -- instead of from the session thread as soon as the chain
// completes.  Every rank then issues its exchanges at the same place relative to the collectives the
// caller issues on its own process group (e.g. DDP gradient all-reduces), which rules out the
// opposite-order queueing of two communicators' kernels; the price is less overlap.
static spp_status issue_exchanges_up_to(spp_session* s, int64_t last_group) {
  if (last_group >= s->num_groups) last_group = s->num_groups - 1;
  while (s->exchange_launched <= last_group) {
    const int64_t g = s->exchange_launched;
    SPP_TRY(wait_group_launched(s, g));
    const spp_status rc = exchange_group(s, g);
    if (rc != SPP_OK) {
      s->exchange_rc = rc;
      s->exchange_err = spp_last_error();
      return rc;
    }
    std::lock_guard<std::mutex> lk(s->mu);
    s->exchange_launched = g + 1;
  }
  return SPP_OK;
}

// row indices outside their table met by the exchange's kernels (spp_async_errors): reported once -- the bits are
// cleared as they are read, so that a later Session on this device (another sampler, another communicator) does
// not fail on a stale error of this one
static spp_status check_exchange_errors(spp_session* s) {
  int32_t* aw = async_err_word(s->cfg.device);
  int32_t bits = aw ? __atomic_load_n(aw, __ATOMIC_ACQUIRE) : 0;
  if (!(bits & (SPP_AERR_SERVE_ID | SPP_AERR_ASSEMBLE))) return SPP_OK;
  bits = __atomic_fetch_and(aw, ~(SPP_AERR_SERVE_ID | SPP_AERR_ASSEMBLE), __ATOMIC_ACQ_REL);
  if (!(bits & (SPP_AERR_SERVE_ID | SPP_AERR_ASSEMBLE))) return SPP_OK;
  set_error("feature exchange: a row outside its table was requested (async error mask %d: 2 = a peer asked this "
            "rank for a row it does not own, 4 = assembly source out of range) -- the ranks disagree on the "
            "partition book or the bucketing", bits);
  return SPP_ERR_STATE;
}

extern "C" int spp_session_next(spp_session* s, spp_batch_desc* out) {
  if (!s || !out) {
    set_error("spp_session_next: NULL argument");
    return SPP_ERR_INVALID;
  }
  if (s->current_group >= 0) {
    set_error("spp_session_next: a group returned by spp_session_next_group has not been exported");
    return SPP_ERR_STATE;
  }
  if (s->current_slot >= 0) {  // previous batch was never exported: drop it
    spp_status rc = retire_current(s);
    if (rc != SPP_OK) return rc;
  }
  flush_group_consumed(s);
  if (s->next_to_deliver == (int64_t)s->ranges.size()) return 0;  // blocking_get_batch -> None
  const int64_t b = s->next_to_deliver;
  const int64_t g = b / s->G;
  const int32_t slot = (int32_t)((g % s->num_sets) * s->G + b % s->G);
  const auto t0 = std::chrono::steady_clock::now();
  if (b % s->G == 0) trace_ev('N', g);
  spp_status rc = wait_group_launched(s, g);
  if (rc == SPP_OK && s->tr && s->issue_on_consumer)
    // own group now; the next one from mid-group on -- but only when a second slot-set exists: with one
    // set the launcher cannot start group g+1 before this consumer has finished group g
    rc = issue_exchanges_up_to(s, (s->num_sets >= 2 && (b % s->G) * 2 >= s->G) ? g + 1 : g);
  if (rc == SPP_OK) rc = spp_sampler_wait(s->sampler, slot, &out->counts);
  if (rc == SPP_OK && s->tr) rc = wait_group_exchanged(s, g);
  const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  if (us > 50) {  // the reference counts only waits that actually spun (fast_sampler.cpp:788-799)
    s->blocked_us += us;
    s->blocked_occasions++;
  }
  if (rc == SPP_OK && (s->tr || s->p2p)) rc = check_exchange_errors(s);
  if (rc != SPP_OK) return rc;
  if (b % s->G == 0) trace_ev('G', g);
  out->batch_index = b;
  out->start = s->ranges[(size_t)b].first;
  out->stop = s->ranges[(size_t)b].second;
  out->slot = slot;
  s->current_slot = slot;
  s->next_to_deliver++;
  return 1;
}

extern "C" int spp_session_try_next(spp_session* s, spp_batch_desc* out) {
  if (!s || !out) {
    set_error("spp_session_try_next: NULL argument");
    return SPP_ERR_INVALID;
  }
  if (s->current_slot < 0) flush_group_consumed(s);
  if (s->current_slot < 0 && s->next_to_deliver < (int64_t)s->ranges.size()) {
    // is the next batch ready?  (chain launched and complete; with the native exchange: issued)
    const int64_t g = s->next_to_deliver / s->G;
    const int32_t slot = (int32_t)((g % s->num_sets) * s->G + s->next_to_deliver % s->G);
    {
      std::lock_guard<std::mutex> lk(s->mu);
      if (s->launch_rc == SPP_OK && s->exchange_rc == SPP_OK &&
          (s->chain_launched <= g || (s->tr && !s->issue_on_consumer && s->exchange_launched <= g)))
        return 2;
    }
    hipEvent_t ev = sampler_slot_event(s->sampler, slot);
    if (ev && hipEventQuery(ev) == hipErrorNotReady) return 2;
  }
  return spp_session_next(s, out);  // ready (or an error to report / a batch to drop): does not block now
}

// where the rows of batch `member` of group g come from (native exchange / P2P transport)
static void fill_assemble_src(const spp_session* s, int64_t g, int member, AssembleSrc* src) {
  *src = AssembleSrc{};
  src->x_local = static_cast<const char*>(s->xcfg.x_local_dev);
  src->cache = static_cast<const char*>(s->xcfg.cache_feats_dev);
  src->x_local_stride = s->xcfg.x_local_stride_bytes;
  src->cache_stride = s->xcfg.cache_stride_bytes;
  if (s->p2p) {
    src->p2p = true;
    src->peer_stride = s->peer_stride;
    for (int m = 0; m < s->P; ++m) src->peer[m] = s->peer[m];
    return;
  }
  const XSet& x = s->xsets[(size_t)(g % s->num_sets)];
  src->recv = x.b->recv_rows;
  for (int m = 0; m < s->P; ++m) src->recv_base[m] = x.recv_base[member][m];
}

extern "C" spp_status spp_session_export(spp_session* s, const spp_mfg_out* mfg, const void* x_src_dev, int64_t x_rows,
                                         int64_t x_row_bytes, int64_t x_src_stride_bytes, void* x_out_dev,
                                         const void* y_src_dev, int64_t y_rows, int64_t y_row_bytes, void* y_out_dev,
                                         void* stream) {
  SPP_REQUIRE(s, "spp_session_export: NULL session");
  // Either the batch spp_session_next returned, or -- after spp_session_next_group -- the group's next member: the
  // members of a fetched group may be exported one launch each, in order, instead of by one spp_session_export_group
  const bool member = s->current_slot < 0 && s->current_group >= 0;
  if (s->current_slot < 0 && !member) {
    set_error("spp_session_export: no current batch (call spp_session_next or spp_session_next_group first)");
    return SPP_ERR_STATE;
  }
  const int64_t b = member ? s->current_group * s->G + s->group_member : s->next_to_deliver - 1;
  const int32_t slot = member ? (int32_t)((s->current_group % s->num_sets) * s->G + s->group_member) : s->current_slot;
  const int64_t bs = (int64_t)s->ranges[(size_t)b].second - s->ranges[(size_t)b].first;
  (void)x_rows;
  (void)y_rows;
  // one launch: MFG widening, x_s = serial_index(x, n_id) (fast_sampler.cpp:1006) and
  // y_s = serial_index(y, n_id, batch_size) (fast_sampler.cpp:1009)
  if (s->tr || s->p2p) {
    // x comes from the exchange: order the consumer after the group's rows and assemble in place
    // (P2P transport: nothing was exchanged -- remote rows are read in their owners' partitions)
    const int64_t g = b / s->G;
    AssembleSrc src{};
    fill_assemble_src(s, g, (int)(b % s->G), &src);
    if (s->tr) SPP_HIP_TRY(hipStreamWaitEvent(as_stream(stream), s->xsets[(size_t)(g % s->num_sets)].rows_done, 0));
    SPP_TRY(sampler_deliver(s->sampler, slot, mfg, nullptr, s->xcfg.row_bytes, 0, x_out_dev, y_src_dev, y_row_bytes,
                            bs, y_out_dev, &src, as_stream(stream)));
  } else {
    SPP_TRY(sampler_deliver(s->sampler, slot, mfg, x_src_dev, x_row_bytes, x_src_stride_bytes, x_out_dev, y_src_dev,
                            y_row_bytes, bs, y_out_dev, nullptr, as_stream(stream)));
  }
  // One marker per GROUP on the delivery stream, not one per batch: deliveries on one stream complete in order, so
  // the event behind the group's last delivery covers the earlier ones, and every marker between two ~100 us kernels
  // costs the hardware queue a few microseconds of idle time.  A batch exported on ANOTHER stream than its
  // predecessor closes the predecessor's run with an event of its own.
  hipStream_t st = as_stream(stream);
  const bool last_of_group = (b + 1 == (int64_t)s->ranges.size()) || ((b + 1) % s->G == 0);
  if (s->open_slot >= 0 && s->open_stream != st) {
    SPP_HIP_TRY(hipEventRecord(s->export_done[(size_t)s->open_slot], s->open_stream));
    s->export_recorded[(size_t)s->open_slot] = 1;
    s->open_slot = -1;
  }
  if (last_of_group) {
    SPP_HIP_TRY(hipEventRecord(s->export_done[(size_t)slot], st));
    s->export_recorded[(size_t)slot] = 1;
    s->open_slot = -1;
  } else {
    s->open_slot = slot;
    s->open_stream = st;
  }
  if (member) {
    if (++s->group_member == group_len(s, s->current_group)) {
      const int64_t g = s->current_group;
      s->current_group = -1;
      s->group_member = 0;
      defer_group_consumed(s, g + 1);
    }
    return SPP_OK;
  }
  return retire_current(s);
}

// ---- group-at-a-time consumption -------------------------------------------------------------------
extern "C" int spp_session_next_group(spp_session* s, int32_t block, spp_batch_desc* out, int32_t* n_out) {
  if (!s || !out || !n_out) {
    set_error("spp_session_next_group: NULL argument");
    return SPP_ERR_INVALID;
  }
  *n_out = 0;
  if (s->current_slot >= 0 || s->current_group >= 0) {
    set_error("spp_session_next_group: the previous batch / group has not been exported");
    return SPP_ERR_STATE;
  }
  const int64_t nb = (int64_t)s->ranges.size();
  flush_group_consumed(s);
  if (s->next_to_deliver == nb) return 0;
  if (s->next_to_deliver % s->G != 0) {
    set_error("spp_session_next_group: batch %lld is in the middle of a group (mix per-batch and group calls only "
              "at group boundaries)", (long long)s->next_to_deliver);
    return SPP_ERR_STATE;
  }
  const int64_t g = s->next_to_deliver / s->G;
  const int n = group_len(s, g);
  const int32_t slot0 = (int32_t)((g % s->num_sets) * s->G);
  if (!block) {
    {
      std::lock_guard<std::mutex> lk(s->mu);
      if (s->launch_rc == SPP_OK && s->exchange_rc == SPP_OK &&
          (s->chain_launched <= g || (s->tr && !s->issue_on_consumer && s->exchange_launched <= g)))
        return 2;
    }
    hipEvent_t ev = sampler_slot_event(s->sampler, slot0);
    if (ev && hipEventQuery(ev) == hipErrorNotReady) return 2;
  }
  const auto t0 = std::chrono::steady_clock::now();
  trace_ev('N', g);
  spp_status rc = wait_group_launched(s, g);
  if (rc == SPP_OK && s->tr && s->issue_on_consumer)
    // own group now and -- when a slot-set is to spare -- the next one, so that its transfers overlap the
    // consumption of this group; always at THIS program point (every rank issues the same sequence)
    rc = issue_exchanges_up_to(s, s->num_sets >= 3 ? g + 1 : g);
  for (int i = 0; i < n && rc == SPP_OK; ++i) rc = spp_sampler_wait(s->sampler, slot0 + i, &out[i].counts);
  if (rc == SPP_OK && s->tr) rc = wait_group_exchanged(s, g);
  const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  if (us > 50) {
    s->blocked_us += us;
    s->blocked_occasions++;
  }
  if (rc == SPP_OK && (s->tr || s->p2p)) rc = check_exchange_errors(s);
  if (rc != SPP_OK) return rc;
  trace_ev('G', g);
  for (int i = 0; i < n; ++i) {
    const int64_t b = g * s->G + i;
    out[i].batch_index = b;
    out[i].start = s->ranges[(size_t)b].first;
    out[i].stop = s->ranges[(size_t)b].second;
    out[i].slot = slot0 + i;
  }
  s->current_group = g;
  s->next_to_deliver += n;
  *n_out = n;
  return 1;
}

extern "C" spp_status spp_session_export_group(spp_session* s, int32_t n, const spp_group_out* outs, const void* x_src_dev,
                                               int64_t x_rows, int64_t x_row_bytes, int64_t x_src_stride_bytes,
                                               const void* y_src_dev, int64_t y_rows, int64_t y_row_bytes, void* stream) {
  SPP_REQUIRE(s && outs, "spp_session_export_group: NULL argument");
  if (s->current_group < 0) {
    set_error("spp_session_export_group: no current group (call spp_session_next_group first)");
    return SPP_ERR_STATE;
  }
  const int64_t g = s->current_group;
  SPP_REQUIRE(n == group_len(s, g), "spp_session_export_group: the group holds %d batches, %d given", group_len(s, g), n);
  if (s->group_member != 0) {
    set_error("spp_session_export_group: %d members of the group have been exported one by one already", s->group_member);
    return SPP_ERR_STATE;
  }
  (void)x_rows;
  (void)y_rows;
  const int set = (int)(g % s->num_sets);
  const int32_t slot0 = (int32_t)(set * s->G);
  int64_t bs[kMaxGroup];
  for (int i = 0; i < n; ++i) {
    const auto& r = s->ranges[(size_t)(g * s->G + i)];
    bs[i] = (int64_t)r.second - r.first;
  }
  hipStream_t st = as_stream(stream);
  if (s->tr || s->p2p) {
    if (s->tr) SPP_HIP_TRY(hipStreamWaitEvent(st, s->xsets[(size_t)set].rows_done, 0));
    std::vector<AssembleSrc> src((size_t)n);
    for (int i = 0; i < n; ++i) fill_assemble_src(s, g, i, &src[(size_t)i]);
    SPP_TRY(sampler_deliver_group(s->sampler, set, slot0, n, outs, nullptr, s->xcfg.row_bytes, 0, y_src_dev, y_row_bytes, bs,
                                  src.data(), st));
  } else {
    SPP_TRY(sampler_deliver_group(s->sampler, set, slot0, n, outs, x_src_dev, x_row_bytes, x_src_stride_bytes, y_src_dev,
                                  y_row_bytes, bs, nullptr, st));
  }
  // one event for the whole set: the chain that reuses it waits for this launch
  SPP_HIP_TRY(hipEventRecord(s->export_done[(size_t)slot0], st));
  s->export_recorded[(size_t)slot0] = 1;
  s->current_group = -1;
  defer_group_consumed(s, g + 1);
  return SPP_OK;
}

extern "C" spp_status spp_session_exchange_stats(const spp_session* s, int64_t* sent_bytes, int64_t* recv_bytes) {
  SPP_REQUIRE(s, "spp_session_exchange_stats: NULL session");
  if (sent_bytes) *sent_bytes = s->sent_bytes.load();
  if (recv_bytes) *recv_bytes = s->recv_bytes.load();
  return SPP_OK;
}

extern "C" spp_status spp_session_quiesce(spp_session* s) {
  SPP_REQUIRE(s, "spp_session_quiesce: NULL session");
  // (a deferred "group consumed" stays deferred: quiesce drains what HAS been enqueued, it does not start a refill)
  {
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [s] {
      const int64_t target = std::min<int64_t>(s->num_groups, s->groups_consumed + launch_window(s));
      const bool launcher_idle = s->launch_rc != SPP_OK || s->chain_launched >= target;
      const bool exchanger_idle = !s->tr || s->issue_on_consumer || s->exchange_rc != SPP_OK ||
                                  s->launch_rc != SPP_OK || s->exchange_launched >= target;
      return launcher_idle && exchanger_idle;
    });
  }
  SPP_HIP_TRY(hipSetDevice(s->cfg.device));
  for (auto st : s->streams)
    if (st) SPP_HIP_TRY(hipStreamSynchronize(st));
  if (s->comm_stream) SPP_HIP_TRY(hipStreamSynchronize(s->comm_stream));
  return SPP_OK;
}
