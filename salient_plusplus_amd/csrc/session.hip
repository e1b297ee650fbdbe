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
#include "spp_internal.h"

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "sampler_internal.h"

using namespace spp;

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
  // Launcher thread: enqueues the ~40 kernel launches of a group's chain off the consumer thread
  // (they cost 0.2-0.5 ms of host time per group, which used to stall the consumer at every group
  // boundary).  It is the counterpart of the reference's worker threads, with the GPU doing the work.
  std::thread launcher;
  std::mutex mu;
  std::condition_variable cv;
  int64_t groups_consumed = 0;           // groups fully consumed by the caller (guarded by mu)
  bool stop = false;
  spp_status launch_rc = SPP_OK;
  std::string launch_err;
  int64_t next_to_deliver = 0;           // batch index
  int32_t current_slot = -1;             // delivered by next(), not yet exported/recycled
  int64_t blocked_us = 0;
  int64_t blocked_occasions = 0;
};

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
  if (g < s->num_sets) SPP_TRY(launch_group_rng(s, g));
  // the consumer's copies out of these slots (previous group of this slot-set) must be done
  for (int i = 0; i < s->G; ++i) {
    const size_t slot = (size_t)(set * s->G + i);
    if (s->export_recorded[slot]) {
      SPP_HIP_TRY(hipStreamWaitEvent(st, s->export_done[slot], 0));
      s->export_recorded[slot] = 0;
    }
  }
  SPP_TRY(sampler_launch_chain(s->sampler, set * s->G, n, buf, seeds, n_seeds, st));
  if (g + s->num_sets < s->num_groups) SPP_TRY(launch_group_rng(s, g + s->num_sets));
  return SPP_OK;
}

// Launcher thread body: keep chains in flight for up to num_sets groups beyond the ones fully consumed.
static void launcher_main(spp_session* s) {
  (void)hipSetDevice(s->cfg.device);
  std::unique_lock<std::mutex> lk(s->mu);
  for (;;) {
    s->cv.wait(lk, [s] {
      return s->stop || (s->launch_rc == SPP_OK && s->chain_launched < s->num_groups &&
                         s->chain_launched < s->groups_consumed + s->num_sets);
    });
    if (s->stop) return;
    const int64_t g = s->chain_launched;
    lk.unlock();
    const spp_status rc = launch_group_chain(s, g);  // touches only slots of group g's slot-set
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

// consumer side: block until the chain of group g has been enqueued (its completion event exists)
static spp_status wait_group_launched(spp_session* s, int64_t g) {
  std::unique_lock<std::mutex> lk(s->mu);
  s->cv.wait(lk, [s, g] { return s->chain_launched > g || s->launch_rc != SPP_OK; });
  if (s->chain_launched > g) return SPP_OK;
  set_error("%s", s->launch_err.c_str());
  return s->launch_rc;
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
  spp_partition_cfg want{};
  if (cfg->part) want = *cfg->part;
  build_ranges(*cfg, s->ranges);
  int64_t max_batch = 1;
  for (auto& r : s->ranges) max_batch = std::max<int64_t>(max_batch, r.second - r.first);
  const int64_t nb = (int64_t)s->ranges.size();
  const int M = (int)std::max<int64_t>(1, std::min<int64_t>(cfg->max_items_in_queue, std::max<int64_t>(nb, 1)));

  bool generic = false;
  for (int h = 0; h < cfg->num_hops; ++h) generic |= (cfg->sizes[h] < 0 || cfg->sizes[h] > 32);
  // auto: two slot-sets of up to 8 batches (each set-stream then owns a hardware queue; measured
  // best on MI355X: 16 slots = 2 x 8), smaller groups only when fewer slots are allowed
  int G = cfg->group_size > 0 ? cfg->group_size : std::max(1, std::min(8, M / 2));
  G = std::min(G, std::min(M, kMaxGroup));
  if (generic) G = 1;
  int sets = std::max(1, std::min(M / G, kMaxWorkStreams));

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
  s->num_groups = (nb + G - 1) / G;

  auto mk_event = [&](hipEvent_t* ev) {
    if (rc == SPP_OK && hipEventCreateWithFlags(ev, hipEventDisableTiming) != hipSuccess) {
      set_error("spp_session_create: hipEventCreate failed");
      rc = SPP_ERR_HIP;
    }
  };
  s->streams.assign((size_t)sets, nullptr);
  for (int i = 0; i < sets; ++i) s->streams[(size_t)i] = sampler_work_stream(s->sampler, i);
  s->export_done.assign((size_t)(sets * G), nullptr);
  s->export_recorded.assign((size_t)(sets * G), 0);
  for (auto& e : s->export_done) mk_event(&e);
  if (rc == SPP_OK) {
    s->launcher = std::thread(launcher_main, s);  // primes the pipeline right away
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
  if (!s) return;
  if (s->launcher.joinable()) {
    {
      std::lock_guard<std::mutex> lk(s->mu);
      s->stop = true;
    }
    s->cv.notify_all();
    s->launcher.join();
  }
  (void)hipSetDevice(s->cfg.device);
  for (auto st : s->streams)
    if (st) (void)hipStreamSynchronize(st);
  if (s->sampler && s->owns_sampler) spp_sampler_destroy(s->sampler);
  for (auto ev : s->export_done)
    if (ev) (void)hipEventDestroy(ev);
  delete s;
}

extern "C" int64_t spp_session_num_total_batches(const spp_session* s) { return s ? (int64_t)s->ranges.size() : 0; }
extern "C" int64_t spp_session_num_consumed_batches(const spp_session* s) { return s ? s->next_to_deliver : 0; }
extern "C" int64_t spp_session_blocked_us(const spp_session* s) { return s ? s->blocked_us : 0; }
extern "C" int64_t spp_session_blocked_occasions(const spp_session* s) { return s ? s->blocked_occasions : 0; }
extern "C" spp_sampler* spp_session_sampler(spp_session* s) { return s ? s->sampler : nullptr; }

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
  if (last_of_group) notify_group_consumed(s, g + 1);
  return SPP_OK;
}

extern "C" int spp_session_next(spp_session* s, spp_batch_desc* out) {
  if (!s || !out) {
    set_error("spp_session_next: NULL argument");
    return SPP_ERR_INVALID;
  }
  if (s->current_slot >= 0) {  // previous batch was never exported: drop it
    spp_status rc = retire_current(s);
    if (rc != SPP_OK) return rc;
  }
  if (s->next_to_deliver == (int64_t)s->ranges.size()) return 0;  // blocking_get_batch -> None
  const int64_t b = s->next_to_deliver;
  const int64_t g = b / s->G;
  const int32_t slot = (int32_t)((g % s->num_sets) * s->G + b % s->G);
  const auto t0 = std::chrono::steady_clock::now();
  spp_status rc = wait_group_launched(s, g);
  if (rc == SPP_OK) rc = spp_sampler_wait(s->sampler, slot, &out->counts);
  const auto us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  if (us > 50) {  // the reference counts only waits that actually spun (fast_sampler.cpp:788-799)
    s->blocked_us += us;
    s->blocked_occasions++;
  }
  if (rc != SPP_OK) return rc;
  out->batch_index = b;
  out->start = s->ranges[(size_t)b].first;
  out->stop = s->ranges[(size_t)b].second;
  out->slot = slot;
  s->current_slot = slot;
  s->next_to_deliver++;
  return 1;
}

extern "C" spp_status spp_session_export(spp_session* s, const spp_mfg_out* mfg, const void* x_src_dev, int64_t x_rows,
                                         int64_t x_row_bytes, void* x_out_dev, const void* y_src_dev, int64_t y_rows,
                                         int64_t y_row_bytes, void* y_out_dev, void* stream) {
  SPP_REQUIRE(s, "spp_session_export: NULL session");
  if (s->current_slot < 0) {
    set_error("spp_session_export: no current batch (call spp_session_next first)");
    return SPP_ERR_STATE;
  }
  const int32_t slot = s->current_slot;
  const int64_t b = s->next_to_deliver - 1;
  const int64_t bs = (int64_t)s->ranges[(size_t)b].second - s->ranges[(size_t)b].first;
  (void)x_rows;
  (void)y_rows;
  // one launch: MFG widening, x_s = serial_index(x, n_id) (fast_sampler.cpp:1006) and
  // y_s = serial_index(y, n_id, batch_size) (fast_sampler.cpp:1009)
  SPP_TRY(sampler_deliver(s->sampler, slot, mfg, x_src_dev, x_row_bytes, x_out_dev, y_src_dev, y_row_bytes, bs,
                          y_out_dev, as_stream(stream)));
  SPP_HIP_TRY(hipEventRecord(s->export_done[(size_t)slot], as_stream(stream)));
  s->export_recorded[(size_t)slot] = 1;
  return retire_current(s);
}
