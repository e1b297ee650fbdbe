"""Cost of an epoch boundary: wall time of E epochs of a short workload (S-arxiv: 88 batches per epoch),
split into Session creation, first batch, steady batches and teardown (development aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.shufflers import Shuffler  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

if os.environ.get("GC_OFF") == "1":
    import gc
    gc.disable()
dev = torch.device("cuda", 0)
wl = make_workload(os.environ.get("WL", "S-arxiv"), device=dev)
bs = wl.batch_size
cfg = FastSamplerConfig(
    x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx,
    batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False, distributed=False,
    partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
    exact_num_batches=max(1, wl.train_idx.numel() // bs), count_remote_frequency=False, use_cache=False)
# --- max duration per epoch of the pieces of one next() call (where does an occasional 30-70 ms stall sit?) ---
worst = {}


def _timed(cls, name):
    fn = getattr(cls, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = time.perf_counter() - t
            if d > worst.get(name, 0.0):
                worst[name] = d
    setattr(cls, name, wrap)


for _n in ("_next_desc", "_alloc_mfg", "_export", "close"):
    _timed(fs.Session, _n)
from salient_plusplus_amd.fast_trainer import samplers as _smp  # noqa: E402
_timed(_smp.PreparedBatch, "record_stream")
_orig_empty = torch.empty


def _empty(*a, **k):
    t = time.perf_counter()
    r = _orig_empty(*a, **k)
    d = time.perf_counter() - t
    if d > worst.get("torch.empty", 0.0):
        worst["torch.empty"] = d
    return r


torch.empty = _empty
sampler = FastSampler(4, 32, cfg)
shuffler = Shuffler(wl.train_idx)
for epoch in range(int(os.environ.get('EPOCHS', '24'))):
    t0 = time.perf_counter()
    shuffler.set_epoch(epoch)
    sampler.idx = shuffler.get_idx()
    t1 = time.perf_counter()
    it = iter(sampler)
    t2 = time.perf_counter()
    devit = DevicePrefetcher([dev], it)
    t3 = time.perf_counter()
    n = 0
    per = []
    tp = time.perf_counter()
    for (b,) in devit:
        n += 1
        tn = time.perf_counter()
        per.append(tn - tp)
        tp = tn
    t4 = time.perf_counter()
    big = sorted(((v, k) for k, v in enumerate(per)), reverse=True)[:6]
    print("   slowest next() calls (ms, batch):", [(round(v * 1e3, 2), k) for v, k in big],
          "blocked", it.get_stats().total_blocked_dur.total_seconds() * 1e3, "ms in", it.get_stats().total_blocked_occasions,
          "reserved MB", torch.cuda.memory_reserved() >> 20)
    del devit, it, b
    torch.cuda.synchronize()
    t5 = time.perf_counter()
    print("   worst piece (ms):", {k: round(v * 1e3, 2) for k, v in worst.items()})
    worst.clear()
    print(f"epoch {epoch}: shuffle {1e3*(t1-t0):.2f} ms, Session {1e3*(t2-t1):.2f}, first batch {1e3*(t3-t2):.2f}, "
          f"{n} batches {1e3*(t4-t3):.2f} ({1e6*(t4-t3)/n:.0f} us each), teardown {1e3*(t5-t4):.2f}  total {1e3*(t5-t0):.2f} ms",
          flush=True)
