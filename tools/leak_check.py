"""Free HBM before / after many epochs of the iterators (development aid: leaks in the pooled sampler,
exchange buffers or per-epoch Sessions would show as a steady decline)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher, DevicePrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = make_workload("S-arxiv", device=dev)
N, F = wl.num_nodes, wl.x.size(1)
dist_mode = len(sys.argv) > 1 and sys.argv[1] == "dist"
if dist_mode:
    import ctypes as C
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    token = (C.c_uint8 * nat.SPP_COMM_ID_BYTES)()
    nat.check(L.spp_comm_unique_id(token))
    h = C.c_void_p()
    nat.check(L.spp_comm_create(token, 0, 1, 0, C.byref(h)))
    fs.set_native_comm(fs.NativeComm(h, 0, 1))
cfg = FastSamplerConfig(
    x_cpu=torch.empty((0, F), dtype=wl.x.dtype) if dist_mode else wl.x, x_gpu=wl.x if dist_mode else torch.empty(0),
    y=wl.y.unsqueeze(-1), rowptr=wl.rowptr, col=wl.col, idx=wl.train_idx, batch_size=wl.batch_size, sizes=wl.fanouts,
    skip_nonfull_batch=False, pin_memory=False, distributed=dist_mode,
    partition_book=fs.RangePartitionBook(0, 1, torch.tensor([0, N])) if dist_mode else None, cache=fs.Cache(),
    force_exact_num_batches=True, exact_num_batches=wl.train_idx.numel() // wl.batch_size,
    count_remote_frequency=False, use_cache=False)
sampler = FastSampler(2, 32 if dist_mode else 16, cfg)
Pre = DeviceDistributedPrefetcher if dist_mode else DevicePrefetcher
for epoch in range(60):
    n = sum(1 for _ in Pre([dev], iter(sampler)))
    if epoch in (4, 9, 19, 39, 59):
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        print(f"epoch {epoch + 1}: {n} batches, free HBM {free / 2**30:.3f} GiB, torch reserved "
              f"{torch.cuda.memory_reserved() / 2**30:.3f} GiB", flush=True)
