"""What an epoch boundary costs on the host (GPU box): the CPU permutation of the training ids, its upload, the
gather of the ids, and the creation of the next Session.  usage: epoch_cost.py [workload=S-papers]"""
import os
import sys
import time
os.environ.setdefault("OMP_NUM_THREADS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from salient_plusplus_amd import fast_sampler as fs
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
from salient_plusplus_amd.fast_trainer.shufflers import Shuffler
from salient_plusplus_amd.synthetic import make_workload
dev = torch.device("cuda", 0)
wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "S-papers", seed=1234, device=dev)
sh = Shuffler(wl.train_idx)


def t(fn, n=5):
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        s = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - s)
    return best * 1e3, r


ms, order = t(lambda: sh._permutation())
print(f"torch.randperm({wl.train_idx.numel()}) on the CPU generator: {ms:.2f} ms")
ms, od = t(lambda: order.to(dev))
print(f"upload of the permutation: {ms:.2f} ms")
ms, idx = t(lambda: wl.train_idx[od])
print(f"gather of the ids: {ms:.2f} ms")
cfg = FastSamplerConfig(x_cpu=wl.x, x_gpu=torch.empty(0), y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=idx,
                        batch_size=wl.batch_size, sizes=wl.fanouts, skip_nonfull_batch=False, pin_memory=False,
                        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=True,
                        exact_num_batches=idx.numel() // wl.batch_size, count_remote_frequency=False, use_cache=False)
smp = FastSampler(4, 32, cfg)
it = iter(smp)
next(it)
del it


def new_session():
    it = iter(smp)
    b = next(it)
    del it
    return b


ms, _ = t(new_session, 4)
print(f"new Session + first batch (sampler pooled, arena kept): {ms:.2f} ms")
