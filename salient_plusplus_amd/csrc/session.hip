// a6/a7: Session runtime.  Replaces the reference's CPU worker pool (global ThreadPool,
// fast_sampler.cpp:398-453), per-epoch FastSamplerSession (:533-936), MPMC queues and the
// `items_in_queue` semaphore with `max_items_in_queue` batch slots kept in flight on a few HIP
// streams: producers are GPU streams, back-pressure is slot reuse, and the consumer only ever
// blocks on a HIP event.  Python-visible semantics are preserved:
//   * batch ranges: fast_sampler.cpp:587-627 (plain / skip_nonfull / exact-count split);
//   * per-batch generator seed: gen.seed(range.second*17+5)  (fast_sampler.cpp:994);
//   * worker body (non-distributed): multilayer_sample -> x = x[n_id], y = y[n_id[:bs]]
//     (fast_sampler.cpp:998-1016);
//   * end of data is signalled by "no batch" (the reference returns None).
// Batches are delivered in index order (a valid completion order for the non-distributed path and
// the required order for the distributed one, fast_sampler.cpp:672-711).
#include "spp_internal.h"

#include <chrono>
#include <utility>
#include <vector>

using namespace spp;

struct spp_session {
  spp_session_cfg cfg{};
  spp_sampler* sampler = nullptr;
  bool owns_sampler = true;
  std::vector<std::pair<int32_t, int32_t>> ranges;
  std::vector<hipStream_t> streams;
  std::vector<hipEvent_t> export_done;  // per slot
  int32_t num_slots = 0;
  int64_t next_to_launch = 0;
  int64_t next_to_deliver = 0;
  int32_t current_slot = -1;  // delivered by next(), not yet exported/recycled
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

static spp_status launch_batch(spp_session* s, int64_t b, int32_t slot) {
  const auto& r = s->ranges[(size_t)b];
  hipStream_t st = s->streams[(size_t)slot % s->streams.size()];
  return spp_sampler_sample(s->sampler, slot, s->cfg.idx_dev + r.first, (int64_t)r.second - r.first,
                            spp_batch_seed(r.second), 0, st);
}

// hand `slot` back: the next pending batch starts sampling into it, ordered after `after` (an event
// recorded on the consumer's stream once its copies out of the slot were enqueued), if any
static spp_status recycle_slot(spp_session* s, int32_t slot, hipEvent_t after) {
  if (s->next_to_launch < (int64_t)s->ranges.size()) {
    hipStream_t st = s->streams[(size_t)slot % s->streams.size()];
    if (after) SPP_HIP_TRY(hipStreamWaitEvent(st, after, 0));
    SPP_TRY(launch_batch(s, s->next_to_launch, slot));
    s->next_to_launch++;
  }
  return SPP_OK;
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
  build_ranges(*cfg, s->ranges);
  int64_t max_batch = 1;
  for (auto& r : s->ranges) max_batch = std::max<int64_t>(max_batch, r.second - r.first);
  const int64_t nb = (int64_t)s->ranges.size();
  s->num_slots = (int32_t)std::max<int64_t>(1, std::min<int64_t>(cfg->max_items_in_queue, nb));

  spp_status rc = SPP_OK;
  if (cfg->sampler) {
    spp_sampler_cfg have{};
    rc = spp_sampler_get_cfg(cfg->sampler, &have);
    bool ok = rc == SPP_OK && have.rowptr_dev == cfg->rowptr_dev && have.col_dev == cfg->col_dev &&
              have.num_hops == cfg->num_hops && have.max_batch >= max_batch && have.device == cfg->device && have.replace == 0;
    for (int h = 0; ok && h < cfg->num_hops; ++h) ok = have.sizes[h] == cfg->sizes[h];
    if (!ok) {
      set_error("spp_session_create: borrowed sampler is not compatible with this session's graph/fanouts/batch");
      delete s;
      return SPP_ERR_INVALID;
    }
    s->sampler = cfg->sampler;
    s->owns_sampler = false;
    s->num_slots = std::min<int32_t>(s->num_slots, have.num_slots);
  } else {
    spp_sampler_cfg sc{};
    sc.rowptr_dev = cfg->rowptr_dev;
    sc.col_dev = cfg->col_dev;
    sc.num_nodes = cfg->num_nodes;
    sc.nnz = cfg->nnz;
    sc.num_hops = cfg->num_hops;
    for (int h = 0; h < SPP_MAX_HOPS; ++h) sc.sizes[h] = cfg->sizes[h];
    sc.max_batch = max_batch;
    sc.num_slots = s->num_slots;
    sc.device = cfg->device;
    rc = spp_sampler_create(&sc, &s->sampler);
    if (rc != SPP_OK) {
      delete s;
      return rc;
    }
  }
  int nstreams = cfg->num_streams > 0 ? cfg->num_streams : 4;
  nstreams = std::min(nstreams, (int)s->num_slots);
  for (int i = 0; i < nstreams && rc == SPP_OK; ++i) {
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
      set_error("spp_session_create: hipStreamCreate failed");
      rc = SPP_ERR_HIP;
    } else {
      s->streams.push_back(st);
    }
  }
  for (int i = 0; i < s->num_slots && rc == SPP_OK; ++i) {
    hipEvent_t ev = nullptr;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
      set_error("spp_session_create: hipEventCreate failed");
      rc = SPP_ERR_HIP;
    } else {
      s->export_done.push_back(ev);
    }
  }
  // prime the pipeline: the first num_slots batches start sampling now
  for (int i = 0; i < s->num_slots && rc == SPP_OK && s->next_to_launch < nb; ++i) {
    rc = launch_batch(s, s->next_to_launch, i);
    if (rc == SPP_OK) s->next_to_launch++;
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
  (void)hipSetDevice(s->cfg.device);
  for (auto st : s->streams) (void)hipStreamSynchronize(st);
  if (s->sampler && s->owns_sampler) spp_sampler_destroy(s->sampler);
  for (auto ev : s->export_done) (void)hipEventDestroy(ev);
  for (auto st : s->streams) (void)hipStreamDestroy(st);
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

extern "C" int spp_session_next(spp_session* s, spp_batch_desc* out) {
  if (!s || !out) {
    set_error("spp_session_next: NULL argument");
    return SPP_ERR_INVALID;
  }
  if (s->current_slot >= 0) {  // previous batch was never exported: drop it and reuse its slot
    spp_status rc = recycle_slot(s, s->current_slot, nullptr);
    s->current_slot = -1;
    if (rc != SPP_OK) return rc;
  }
  if (s->next_to_deliver == (int64_t)s->ranges.size()) return 0;  // blocking_get_batch -> None
  const int64_t b = s->next_to_deliver;
  const int32_t slot = (int32_t)(b % s->num_slots);
  const auto t0 = std::chrono::steady_clock::now();
  spp_status rc = spp_sampler_wait(s->sampler, slot, &out->counts);
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
  if (mfg) SPP_TRY(spp_sampler_export(s->sampler, slot, mfg, stream));
  if (x_src_dev && x_out_dev)  // x_s = serial_index(x, n_id)            (fast_sampler.cpp:1006)
    SPP_TRY(spp_sampler_gather(s->sampler, slot, x_src_dev, x_rows, x_row_bytes, -1, x_out_dev, stream));
  if (y_src_dev && y_out_dev)  // y_s = serial_index(y, n_id, batch_size) (fast_sampler.cpp:1009)
    SPP_TRY(spp_sampler_gather(s->sampler, slot, y_src_dev, y_rows, y_row_bytes, bs, y_out_dev, stream));
  SPP_HIP_TRY(hipEventRecord(s->export_done[(size_t)slot], as_stream(stream)));
  s->current_slot = -1;
  return recycle_slot(s, slot, s->export_done[(size_t)slot]);
}
