"""VIP cache construction on the GPU (reference driver/drivers/ddp.py:23-239 VIP analytic
model, :417-570 DDPDriver.create_vip_cache).

``create_vip_cache`` ranks the vertices owned by other ranks (strategy "vip": analytic access
probability, "degree": the reference's degree ordering, "simulation": counted remote accesses of
a dry-run epoch), fetches the feature rows of the top ``cache_size`` percent of N/P from their
owners with the same counts -> ids -> rows exchange as the reference (three synchronous
all_to_alls, done once) and returns the ``fast_sampler.Cache`` the Session consumes."""
import ctypes as C
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from .. import _native as nat
from .. import fast_sampler as fs


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


def vip_frequencies(rowptr: torch.Tensor, col: torch.Tensor, train_idx: torch.Tensor, fanouts: Sequence[int],
                    batch_size: int) -> torch.Tensor:
    """float64[N] on the GPU: probability that a vertex is touched by one mini-batch of this rank
    (ddp.py:135-239 get_frequency_tensors_fast)."""
    L = nat.load()
    nat.require_device()
    dev = torch.device("cuda", torch.cuda.current_device())
    rp = fs._resident.get(rowptr, torch.int64)
    cl = fs._resident.get(col, torch.int64)
    tr = train_idx.to(dev, torch.int64).contiguous()
    n = rp.numel() - 1
    out = torch.empty(n, dtype=torch.float64, device=dev)
    ws = torch.empty(3 * n, dtype=torch.float64, device=dev)
    fan = (C.c_int64 * len(fanouts))(*[int(f) for f in fanouts])
    nat.check(L.spp_vip_frequencies(_p(rp), _p(cl), n, _p(tr), tr.numel(), int(batch_size), fan, len(fanouts),
                                    _p(out), _p(ws), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def rank_remote_vertices(strategy: str, partition_book, num_nodes: int, num_to_cache: int, *, rowptr=None, col=None,
                         train_idx=None, fanouts=None, batch_size=None,
                         remote_vertices_ordered_by_freq: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The remote vertices to cache, best first (ddp.py:425-492).  int64 on the GPU."""
    dev = torch.device("cuda", torch.cuda.current_device())
    rank = int(partition_book.rank)
    offs = partition_book.partition_offsets.to(torch.int64)
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    ids = torch.arange(num_nodes, device=dev)
    external = (ids < lo) | (ids >= hi)                       # partition_ids != rank  (:434, :487)
    if strategy == "vip":
        freq = vip_frequencies(rowptr, col, train_idx, fanouts, batch_size)
        ext_f = freq[external]
        ext_ids = ids[external]
        k = min(int(torch.count_nonzero(ext_f).item()), int(num_to_cache))          # :437-438
        order = torch.argsort(ext_f, descending=True, stable=True)                  # :440 (stable: deterministic ties)
        return ext_ids[order[:k]]
    if strategy == "degree":
        rp = fs._resident.get(rowptr, torch.int64)
        deg = rp[1:] - rp[:-1]
        order = torch.argsort(deg[external], descending=False, stable=True)         # :489: ascending, as the reference
        return ids[external][order[:int(num_to_cache)]]
    if strategy == "simulation":
        v = remote_vertices_ordered_by_freq.to(dev)
        return v[:min(int(num_to_cache), v.numel())]                                # :476-477
    raise ValueError(f"invalid cache strategy {strategy!r}")


def fetch_cache_rows(partition_book, remote_vertices: torch.Tensor, x_local: torch.Tensor, group=None):
    """(cached_vertices, cached_features): the rows of `remote_vertices` fetched from their owners
    (ddp.py:497-553: counts, ids, rows -- three synchronous all_to_alls, once per run)."""
    L = nat.load()
    dev = remote_vertices.device
    P, rank = int(partition_book.world_size), int(partition_book.rank)
    owner = partition_book.nid2partid(remote_vertices)
    parts = [remote_vertices[owner == m] for m in range(P)]   # ranking order kept inside an owner (:500)
    assert parts[rank].numel() == 0, "local vertices must not be cached"            # :504
    cached_vertices = torch.cat(parts) if P > 1 else remote_vertices[:0]
    F = x_local.size(1)
    if P == 1:
        return cached_vertices, torch.empty((0, F), dtype=x_local.dtype, device=dev)
    send_counts = [int(p.numel()) for p in parts]
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(P, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc, group=group)                                     # META :517-519
    recv_counts = [int(v) for v in rc.tolist()]
    recv_ids = torch.empty(sum(recv_counts), dtype=torch.int64, device=dev)
    dist.all_to_all_single(recv_ids, cached_vertices, output_split_sizes=recv_counts,
                           input_split_sizes=send_counts, group=group)              # INDICES :522-524
    local = (recv_ids - int(partition_book.partition_offsets[rank])).contiguous()   # nid2localnid :530
    rows = torch.empty((local.numel(), F), dtype=x_local.dtype, device=dev)
    stride = int(x_local.stride(0)) * x_local.element_size() if x_local.size(0) > 1 else 0
    nat.check(L.spp_gather_rows_strided(_p(x_local), x_local.size(0), F * x_local.element_size(), stride, _p(local),
                                        8, local.numel(), local.numel(), _p(rows),
                                        C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    feats = torch.empty((sum(send_counts), F), dtype=x_local.dtype, device=dev)
    dist.all_to_all_single(feats, rows, output_split_sizes=send_counts, input_split_sizes=recv_counts,
                           group=group)                                             # FEATURES :527-547
    return cached_vertices, feats


def create_vip_cache(partition_book, num_nodes: int, x_local: torch.Tensor, cache_size: float, strategy: str = "vip",
                     *, rowptr=None, col=None, train_idx=None, fanouts=None, batch_size=None,
                     remote_vertices_ordered_by_freq=None, group=None) -> "fs.Cache":
    """DDPDriver.create_vip_cache (ddp.py:417-570).  cache_size is the replication factor in
    percent of N/P (:421); x_local holds this rank's rows in HBM."""
    P, rank = int(partition_book.world_size), int(partition_book.rank)
    num_to_cache = int(num_nodes / P * (cache_size / 100))                          # :421
    if x_local.device.type != "cuda":
        x_local = fs._resident.get_rows(x_local)
    remote = rank_remote_vertices(strategy, partition_book, num_nodes, num_to_cache, rowptr=rowptr, col=col,
                                  train_idx=train_idx, fanouts=fanouts, batch_size=batch_size,
                                  remote_vertices_ordered_by_freq=remote_vertices_ordered_by_freq)
    cached_vertices, cached_features = fetch_cache_rows(partition_book, remote, x_local, group)
    return fs.Cache(rank, P, cached_vertices, cached_features)                      # :558
