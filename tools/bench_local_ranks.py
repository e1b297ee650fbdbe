"""P in-process ranks on ONE GPU through the native exchange (in-process transport) at headline batch
size: a rehearsal of the N>1 pipeline (stalls, buffer growth), not a performance number -- the
ranks share one GPU's HBM and one Python interpreter.  usage: bench_local_ranks.py [P] [batches]"""
import os
os.environ.setdefault("SPP_ALLOW_LOCAL_COMM", "1")   # rehearsal transport: opt-in
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = make_workload(os.environ.get("WL", "S-products"), device=dev)
N, F = wl.num_nodes, wl.x.size(1)
offsets = torch.linspace(0, N, P + 1).long()
offsets[-1] = N
comms = fs.NativeComm.local(P)
res = {}


def rank_main(r):
    torch.cuda.set_device(0)
    fs.set_native_comm(comms[r])
    lo, hi = int(offsets[r]), int(offsets[r + 1])
    deg = (wl.rowptr[1:] - wl.rowptr[:-1]).clone()
    deg[lo:hi] = -1
    cv = torch.topk(deg, int(0.1 * N / P)).indices.sort().values
    cache = fs.Cache(r, P, cv, wl.x[cv].contiguous())
    bs = wl.batch_size
    g = torch.Generator()
    g.manual_seed(r)
    idx = wl.train_idx[torch.randperm(wl.train_idx.numel(), generator=g).to(dev)][:NB * bs]
    cfg = FastSamplerConfig(
        x_cpu=torch.empty((0, F), dtype=wl.x.dtype), x_gpu=wl.x[lo:hi].contiguous(), y=wl.y.unsqueeze(-1),
        rowptr=wl.rowptr, col=wl.col, idx=idx, batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False,
        pin_memory=False, distributed=True, partition_book=fs.RangePartitionBook(r, P, offsets), cache=cache,
        force_exact_num_batches=True, exact_num_batches=NB, count_remote_frequency=False, use_cache=True)
    sampler = FastSampler(2, 32, cfg)
    for epoch in range(3):
        t0 = time.perf_counter()
        n = 0
        it = iter(sampler)
        marks = []
        for (b,) in DeviceDistributedPrefetcher([dev], it, True):
            n += 1
            marks.append(time.perf_counter())
        torch.cuda.synchronize()
        if r == 0 and os.environ.get("SERIES"):
            ms = torch.cuda.memory_stats()
            print("allocator: segments allocated", ms["segment.all.allocated"], "freed", ms["segment.all.freed"],
                  "retries", ms["num_alloc_retries"], "reserved GB", ms["reserved_bytes.all.current"] / 1e9, flush=True)
            d = [round((marks[i + 1] - marks[i]) * 1e6) for i in range(len(marks) - 1)]
            print(f"epoch {epoch} per-batch us:", d, flush=True)
        dt = time.perf_counter() - t0
        sent, recv = 0, 0
        res[(r, epoch)] = (n, dt, it.session.total_blocked_dur.total_seconds(), it.session.total_blocked_occasions)
        it.session.close()
    fs.set_native_comm(None)


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in ts:
    t.start()
for t in ts:
    t.join()
for k in sorted(res):
    n, dt, blk, occ = res[k]
    print(f"rank {k[0]} epoch {k[1]}: {n} batches, {dt/n*1e6:.0f} us/batch, session blocked {blk*1e3:.1f} ms in {occ} waits")
for c in comms:
    c.close()
