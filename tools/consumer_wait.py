"""Is the pipeline bound by the sampling chains or by the consumer?  (GPU box)
usage: python3 tools/consumer_wait.py [workload=S-papers] [batches=600]
One epoch slice through FastSampler -> DevicePrefetcher; prints, per batch: wall time, the time the consumer
spent blocked inside spp_session_next waiting for a sampled batch (spp_session_blocked_us counts waits > 50 us),
and the host time of one next() call when nothing has to be waited for (slots pre-filled, GPU idle)."""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

wl_name = sys.argv[1] if len(sys.argv) > 1 else "S-papers"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 600
dev = torch.device("cuda", 0)
wl = make_workload(wl_name, seed=1234, device=dev)
idx = wl.train_idx[:nb * wl.batch_size].contiguous()
cfg = FastSamplerConfig(x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=idx,
                        batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False,
                        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
                        exact_num_batches=nb, count_remote_frequency=False, use_cache=False)
slots = int(os.environ.get("SPP_MAX_SLOTS", "32"))
for rep in range(2):
    it = iter(FastSampler(4, slots, cfg))
    sess = it.session
    dp = DevicePrefetcher([dev], it)
    torch.cuda.synchronize()
    # host cost of next() with nothing to wait for: let the sampler fill its slots first
    time.sleep(0.05)
    t0 = time.perf_counter()
    for _ in range(slots):
        next(dp)
    t_host = (time.perf_counter() - t0) / slots
    torch.cuda.synchronize()
    b0 = sess.total_blocked_dur.total_seconds()
    o0 = sess.total_blocked_occasions
    prof = None
    if os.environ.get("PROFILE") == "1" and rep == 1:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    n = 0
    for _ in dp:
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("tottime").print_stats(22)
    blocked = sess.total_blocked_dur.total_seconds() - b0
    occ = sess.total_blocked_occasions - o0
    print(f"rep {rep}: {n} batches, {dt / n * 1e6:.1f} us per batch wall; consumer blocked waiting for sampling "
          f"{blocked / n * 1e6:.1f} us per batch ({occ} waits > 50 us); host time of an unblocked next(): {t_host * 1e6:.1f} us",
          flush=True)
