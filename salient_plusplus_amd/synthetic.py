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
}


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


def make_graph(num_nodes: int, num_directed: int, seed: int, device) -> tuple:
    """endpoints src = perm[floor(N*u^2)], dst = floor(N*v): a few very high degree hubs, long tail."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    N = num_nodes
    perm = torch.randperm(N, generator=g, device=device)
    chunks = []
    CH = 1 << 25
    left = num_directed
    while left > 0:
        m = min(CH, left)
        u = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        v = torch.rand(m, generator=g, device=device, dtype=torch.float64)
        src = perm[(u * u * N).long().clamp_(max=N - 1)]
        dst = (v * N).long().clamp_(max=N - 1)
        keep = src != dst
        src, dst = src[keep], dst[keep]
        chunks.append(src * N + dst)
        chunks.append(dst * N + src)          # symmetric
        left -= m
    key = torch.cat(chunks)
    del chunks
    key = torch.unique(key)                    # sorted + coalesced
    row = torch.div(key, N, rounding_mode="floor")
    col = key - row * N
    del key
    counts = torch.bincount(row, minlength=N)
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=rowptr[1:])
    return rowptr, col.contiguous()


def make_workload(name: str, seed: int = 1234, device=None) -> Workload:
    N, m, F, n_train, fanouts, bs = WORKLOADS[name]
    if device is None:
        device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")
    rowptr, col = make_graph(N, m, seed, device)
    g = torch.Generator(device=device)
    g.manual_seed(seed + 1)
    x = torch.randn((N, F), generator=g, device=device, dtype=torch.float32).to(torch.float16)
    y = torch.randint(0, 47, (N,), generator=g, device=device, dtype=torch.int64)
    train_idx = torch.randperm(N, generator=g, device=device)[:n_train].contiguous()
    return Workload(name, rowptr, col, x, y, train_idx, list(fanouts), bs)
