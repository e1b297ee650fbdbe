"""Seeded synthetic workloads at dataset scale (datasets are not available offline; SURVEY.md
section 8(d)): symmetric, coalesced CSR with skewed hubs, fp16 features, integer labels and
train ids.  Built on the GPU when one is present (the papers-scale graphs would take minutes on
the host), identical on every rank because everything derives from the seed."""
from typing import NamedTuple

import torch

WORKLOADS = {
    # name: (num_nodes, directed_edges_before_symmetrisation, feat_dim, num_train, fanouts, batch)
    "S-tiny": (20_000, 200_000, 32, 4_096, [15, 10, 5], 256),
    "S-arxiv": (169_343, 1_166_243, 128, 90_941, [15, 10, 5], 1024),
    "S-products": (2_449_029, 61_859_140, 100, 196_615, [15, 10, 5], 1024),
    # the same sizes with the planted 8-block locality of S-papers: the one-GPU rehearsal of the 8-rank exchange
    "S-products-local": (2_449_029, 61_859_140, 100, 196_615, [15, 10, 5], 1024),
    # ogbn-papers100M scale: 111 M nodes, ~3.2 G symmetric nnz (col 25.8 GB int64), F=128 fp16 (28.4 GB)
    "S-papers": (111_059_956, 1_615_685_872, 128, 1_207_179, [15, 10, 5], 1024),
    "S-papers-uniform": (111_059_956, 1_615_685_872, 128, 1_207_179, [15, 10, 5], 1024),
    # MAG240M paper-paper scale (BASELINE.json configs[4]): 121.7 M nodes, ~2.6 G symmetric nnz, F=768 fp16
    # = 187 GB of features: with the topology it still fits ONE MI355X (288 GB)
    "S-mag": (121_751_666, 1_297_748_926, 768, 1_112_392, [25, 15], 1024),
}

# Planted partition locality.  The multi-GPU configurations of BASELINE.json are METIS partitions of
# real graphs (few cut edges, vertices relabelled so that a partition is a contiguous id range,
# driver/dataset.py:299-353).  A uniformly random graph has no such partition to find -- 7/8 of all
# neighbours would be remote on 8 GPUs whatever the partitioner -- so the papers-scale stand-in plants
# one: the id space is cut into LOCALITY_BLOCKS contiguous blocks and an edge's second endpoint is
# drawn from the first endpoint's block with probability q (80 % intra-block edges = a 20 % edge cut
# at 8 parts, less at 4 and 2 since the contiguous range partitions are unions of blocks).
# "S-papers-uniform" is the same graph without it.  name -> (blocks, q)
LOCALITY = {"S-papers": (8, 0.8), "S-mag": (8, 0.8), "S-products-local": (8, 0.8)}


MAX_KEYS_PER_SORT = 1 << 30     # torch.unique / CUB take fewer than 2^31 keys per call


class Workload(NamedTuple):
    name: str
    rowptr: torch.Tensor      # int64[N+1]
    col: torch.Tensor         # int64[nnz]
    x: torch.Tensor           # fp16[N, F]
    y: torch.Tensor           # int64[N]
    train_idx: torch.Tensor   # int64[num_train]
    fanouts: list
    batch_size: int

    @property
    def num_nodes(self):
        return self.rowptr.numel() - 1


def make_graph(num_nodes: int, num_directed: int, seed: int, device, locality=None) -> tuple:
    """endpoints src = perm[floor(N*u^2)], dst = floor(N*v): a few very high degree hubs, long tail.
    `locality` = (blocks, q): with probability q, dst is drawn from src's block of N/blocks ids instead.

    The (row, col) keys are sorted and coalesced per ROW RANGE so that no single sort sees 2^31 keys
    (papers100M scale has 3.2 G of them); with one range this is exactly one global unique()."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N = num_nodes
    perm = torch.randperm(N, generator=g, device=device)
    n_ranges = max(1, -(-2 * num_directed // MAX_KEYS_PER_SORT))
    bounds = torch.tensor([(N * k) // n_ranges for k in range(n_ranges + 1)], device=device, dtype=torch.int64) * N
    pieces = [[] for _ in range(n_ranges)]
    CH = 1 << 25
    left = num_directed
    while left > 0:
        m = min(CH, left)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        v = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        src = perm[(u * u * N).long().clamp_(max=N - 1)]
        dst = (v * N).long().clamp_(max=N - 1)
        if locality is not None:
            blocks, q = locality
            w = torch.rand(m, generator=g, device=device, dtype=torch.float64)
            blk = (src * blocks) // N                                    # block of the first endpoint
            lo = (blk * N + blocks - 1) // blocks                        # first id of that block
            hi = ((blk + 1) * N + blocks - 1) // blocks
            inside = lo + (v * (hi - lo).double()).long().clamp_(min=0)
            dst = torch.where(w < q, torch.minimum(inside, hi - 1), dst)
        keep = src != dst
        src, dst = src[keep], dst[keep]
        if n_ranges == 1:
            pieces[0].append(src * N + dst)
            pieces[0].append(dst * N + src)          # symmetric
        else:
            key = torch.sort(torch.cat([src * N + dst, dst * N + src])).values
            cut = torch.searchsorted(key, bounds).tolist()
            for k in range(n_ranges):
                if cut[k + 1] > cut[k]:
                    pieces[k].append(key[cut[k]:cut[k + 1]].clone())
            del key
        left -= m
    cols, counts = [], []
    for k in range(n_ranges):
        key = torch.unique(torch.cat(pieces[k]))       # sorted + coalesced
        pieces[k] = None
        row = torch.div(key, N, rounding_mode="floor")
        cols.append(key - row * N)
        lo = (N * k) // n_ranges
        hi = (N * (k + 1)) // n_ranges
        counts.append(torch.bincount(row - lo, minlength=hi - lo))
        del key, row
    col = torch.cat(cols) if n_ranges > 1 else cols[0]
    del cols
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    torch.cumsum(torch.cat(counts) if n_ranges > 1 else counts[0], 0, out=rowptr[1:])
    return rowptr, col.contiguous()


def make_workload(name: str, seed: int = 1234, device=None) -> Workload:
    N, m, F, n_train, fanouts, bs = WORKLOADS[name]
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    rowptr, col = make_graph(N, m, seed, device, LOCALITY.get(name))
    if torch.device(device).type == "cuda":
        torch.cuda.empty_cache()      # make room for the feature matrix (187 GB at MAG240 scale)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    x = torch.empty((N, F), device=device, dtype=torch.float16)
    step = 1 << 23                                       # chunked: no fp32 copy of a papers-scale matrix
    for i in range(0, N, step):
        j = min(N, i + step)
        x[i:j] = torch.randn((j - i, F), generator=g, device=device, dtype=torch.float32).to(torch.float16)
    y = torch.randint(0, 47, (N,), generator=g, device=device, dtype=torch.int64)
    train_idx = torch.randperm(N, generator=g, device=device)[:n_train].contiguous()
    if torch.device(device).type == "cuda":
        torch.cuda.empty_cache()      # the sort temporaries of a papers-scale build are tens of GB
    return Workload(name, rowptr, col, x, y, train_idx, list(fanouts), bs)
