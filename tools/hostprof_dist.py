"""Host-side profile of the distributed iterator with one rank (development aid)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda", 0)
wl = make_workload("S-products", device=dev)
bs = wl.batch_size
N = wl.num_nodes
cfg = FastSamplerConfig(
    x_cpu=torch.empty((0, wl.x.size(1)), dtype=wl.x.dtype), x_gpu=wl.x, y=wl.y.unsqueeze(-1), rowptr=wl.rowptr,
    col=wl.col, idx=wl.train_idx, batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False,
    distributed=True, partition_book=fs.RangePartitionBook(0, 1, torch.tensor([0, N])), cache=fs.Cache(),
    force_exact_num_batches=True, exact_num_batches=max(1, wl.train_idx.numel() // bs),
    count_remote_frequency=False, use_cache=False)
sampler = FastSampler(4, 16, cfg)


def epoch():
    n = 0
    t0 = time.perf_counter()
    for (b,) in DeviceDistributedPrefetcher([dev], iter(sampler), True):
        n += 1
    torch.cuda.synchronize()
    return n, time.perf_counter() - t0


epoch()
n, dt = epoch()
print(f"distributed world=1: {n} batches, {dt/n*1e6:.0f} us/batch", flush=True)
pr = cProfile.Profile()
pr.enable()
epoch()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
dist.destroy_process_group()
