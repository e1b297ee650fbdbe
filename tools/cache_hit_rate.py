"""Composition of a rank's batches under the bench's N-way partition and VIP cache: local rows,
cache hits, rows to fetch (development aid).  usage: cache_hit_rate.py [P] [cache_frac] [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.vip_cache import rank_remote_vertices  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.10
name = sys.argv[3] if len(sys.argv) > 3 else "S-papers"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = make_workload(name, device=dev)
N, F = wl.num_nodes, wl.x.size(1)
offsets = torch.linspace(0, N, P + 1).long()
offsets[-1] = N
rank = 0
pb = fs.RangePartitionBook(rank, P, offsets)
federated = os.environ.get("SEEDS", "federated") == "federated"
seeds = wl.train_idx[wl.train_idx < int(offsets[1])] if federated else wl.train_idx   # rank 0's own partition
for strategy in ("vip", "degree-desc"):
    k = int(frac * N / P)
    if strategy == "vip":
        cv = rank_remote_vertices("vip", pb, N, k, rowptr=wl.rowptr, col=wl.col, train_idx=seeds,
                                  fanouts=wl.fanouts, batch_size=wl.batch_size)
    else:
        deg = (wl.rowptr[1:] - wl.rowptr[:-1]).clone()
        deg[int(offsets[rank]):int(offsets[rank + 1])] = -1
        cv = torch.topk(deg, k).indices
    cache = fs.Cache(rank, P, cv, torch.zeros((cv.numel(), F), dtype=wl.x.dtype, device=dev))
    bs = wl.batch_size
    cfg = FastSamplerConfig(
        x_cpu=torch.empty((0, F), dtype=wl.x.dtype), x_gpu=wl.x[:int(offsets[1])], y=wl.y.unsqueeze(-1),
        rowptr=wl.rowptr, col=wl.col, idx=seeds[:16 * bs], batch_size=bs, sizes=wl.fanouts,
        skip_nonfull_batch=False, pin_memory=False, distributed=True, partition_book=pb, cache=cache,
        force_exact_num_batches=True, exact_num_batches=16, count_remote_frequency=False, use_cache=True)
    tot = loc = hit = 0
    remote = []                  # per batch: the ids to fetch (all peers)
    for proto in iter(FastSampler(2, 16, cfg)):
        counts = [int(t.numel()) for t in proto.partition_nids]
        tot += sum(counts) + int(proto.cached_nids.numel())
        loc += counts[rank]
        hit += int(proto.cached_nids.numel())
        remote.append(torch.cat([t for m, t in enumerate(proto.partition_nids) if m != rank]))
    fetch = tot - loc - hit
    # rows requested more than once inside a group of 8 batches (what a per-group request list would save)
    dup = []
    for g0 in range(0, len(remote), 8):
        grp = torch.cat(remote[g0:g0 + 8])
        dup.append(1.0 - torch.unique(grp).numel() / max(1, grp.numel()))
    print(f"  duplicate requests inside a group of 8 batches: {sum(dup) / len(dup):.1%} of the rows to fetch", flush=True)
    print(f"{name} P={P} {'federated' if federated else 'global'} seeds, cache {frac:.0%} ({cv.numel()} rows, {strategy}): per batch {tot/16:.0f} nodes = "
          f"{loc/tot:.1%} local + {hit/tot:.1%} cache hits + {fetch/tot:.1%} to fetch "
          f"({fetch/16*F*2/1e6:.0f} MB of rows per batch)", flush=True)
