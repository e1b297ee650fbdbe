"""Where does the host spend its time per batch? (development aid)"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda", 0)
wl = make_workload(os.environ.get("WL", "S-products"), device=dev)
bs = wl.batch_size
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
    batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
    exact_num_batches=max(1, wl.train_idx.numel() // bs), count_remote_frequency=False, use_cache=False)
sampler = FastSampler(4, int(os.environ.get("SLOTS", "24")), cfg)


def epoch(prefetch=True):
    it = iter(sampler)
    devit = DevicePrefetcher([dev], it) if prefetch else None
    n = 0
    t0 = time.perf_counter()
    if prefetch:
        for (b,) in devit:
            n += 1
    else:
        for b in it:
            n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = it.get_stats()
    return n, dt, st


for prefetch in (True, False):
    epoch(prefetch)
    n, dt, st = epoch(prefetch)
    print(f"prefetch={prefetch}: {n} batches in {dt*1e3:.1f} ms = {dt/n*1e6:.0f} us/batch; session blocked "
          f"{st.total_blocked_dur.total_seconds()*1e3:.1f} ms in {st.total_blocked_occasions} waits", flush=True)

pr = cProfile.Profile()
pr.enable()
n, dt, st = epoch(True)
pr.disable()
ps = pstats.Stats(pr).sort_stats("cumulative")
ps.print_stats(18)
