"""On-disk dataset formats of SALIENT++: the data either side of the hot path.

``FastDataset`` (driver/dataset.py:29-142) and ``DisjointPartFeatReorderedDataset``
(driver/dataset.py:145-427) with the reference's directory layouts, so that data prepared for the
reference loads here unchanged and vice versa:

    <root>/<name>/{name,x,y,rowptr,col,split_idx,meta_info}.pt                       FastDataset
    <root>/metis-reordered-k<P>/<name>/{num_parts,rowptr,col,split_idx,split_idx_parts,
        part_offsets,y,meta_info,name}.pt + x<r>.pt per partition                   partitioned

``reorder_and_save`` relabels the vertices so that every partition owns a contiguous id range,
hottest (highest access probability) first inside a partition, and permutes the CSR, features,
labels and splits accordingly (dataset.py:270-369).  The OGB download / PyG import paths of the
reference are out of scope (no network, no PyG here): datasets come from tensors
(``FastDataset.from_tensors``) or from disk.  ``adj_t()`` returns torch_sparse's SparseTensor when it
is installed and the CSR holder of ``fast_trainer.monkeypatch`` otherwise.
"""
from pathlib import Path
from typing import Any, Mapping, NamedTuple, Optional

import torch

from .fast_sampler import RangePartitionBook


def csr_permute_symmetric(rowptr: torch.Tensor, col: torch.Tensor, invperm: torch.Tensor):
    """Relabel rows and columns of a CSR with ``new = invperm[old]`` and return the coalesced CSR
    (rows ascending, columns ascending inside a row, duplicate entries merged) -- the nested helper of
    dataset.py:289-297 without the torch_sparse dependency.  Runs where the inputs live (GPU or CPU)."""
    n = rowptr.numel() - 1
    dev = col.device
    deg = rowptr[1:] - rowptr[:-1]
    rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
    inv = invperm.to(dev)
    key = inv[rows] * n + inv[col]                      # (new row, new col) as one sortable key; n^2 < 2^63
    key = torch.unique(key, sorted=True)                # sort + coalesce
    new_rows = torch.div(key, n, rounding_mode="floor")
    new_col = key - new_rows * n
    new_rowptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
    new_rowptr[1:] = torch.cumsum(torch.bincount(new_rows, minlength=n), 0)
    return new_rowptr, new_col


class _VertexOrder(NamedTuple):
    """A relabelling of the vertices: ``old_id[new]`` and ``new_id[old]`` (inverse permutations) and the
    first new id of every partition (``part_offsets``, P + 1 entries); all on one device."""
    old_id: torch.Tensor
    new_id: torch.Tensor
    part_offsets: torch.Tensor
    num_parts: int


def _vertex_order(partition_labels: torch.Tensor, probability_of_access: Optional[torch.Tensor], device) -> _VertexOrder:
    """Partitions in ascending order; inside a partition descending access probability, ties by old id.
    Two stable sorts (minor key first) instead of the reference's single float key
    ``2 * (P - 1 - label) + probability`` (dataset.py:309-320): the same order for probabilities in
    [0, 1], exact for any number of partitions, and reproducible when probabilities tie."""
    labels = partition_labels.to(device=device, dtype=torch.int64)
    n = labels.numel()
    P = int(labels.max()) + 1 if n else 0
    if probability_of_access is None:
        hot_first = torch.arange(n, device=device)
    else:
        prob = probability_of_access.to(device)
        if prob.dim() == 2:                                  # [P, N]: row p is valid for partition p's vertices
            prob = prob.gather(0, labels.unsqueeze(0)).squeeze(0)
        elif prob.dim() != 1:
            raise ValueError(f"probability_of_access must be 1-D or 2-D, got {prob.dim()}-D")
        hot_first = torch.sort(prob, descending=True, stable=True).indices
    by_part = torch.sort(labels[hot_first], stable=True).indices
    old_id = hot_first[by_part]
    new_id = torch.empty_like(old_id)
    new_id[old_id] = torch.arange(n, device=device)
    offsets = torch.zeros(P + 1, dtype=torch.int64, device=device)
    offsets[1:] = torch.cumsum(torch.bincount(labels, minlength=P), 0)
    return _VertexOrder(old_id, new_id, offsets, P)


def _bucket_splits(split_idx: Mapping[str, torch.Tensor], order: _VertexOrder):
    """{partition: {split: new ids of the split's vertices owned by the partition}} (dataset.py:327-343).
    A partition owns a contiguous range of new ids, so the owner is a search in ``part_offsets``."""
    dev = order.new_id.device
    out = {r: dict() for r in range(order.num_parts)}
    for name, ids in split_idx.items():
        relabelled = order.new_id[ids.to(dev)]
        owner = torch.searchsorted(order.part_offsets, relabelled, right=True) - 1
        grouped = relabelled[torch.sort(owner, stable=True).indices].cpu()
        cuts = torch.cumsum(torch.bincount(owner, minlength=order.num_parts), 0).tolist()
        lo = 0
        for r, hi in enumerate(cuts):
            out[r][name] = grouped[lo:hi].clone()
            lo = hi
    return out


def _take_rows(t: torch.Tensor, rows: torch.Tensor, device) -> torch.Tensor:
    """t[rows] on `device`; 2-D tables go through the HIP row gather (spp_gather_rows) when that is a GPU.
    Offline path: synchronises and raises when a row index was outside the table (the kernel clamps, it
    does not fault)."""
    device = torch.device(device)
    if device.type == "cuda" and t.dim() == 2 and t.stride(-1) == 1:
        from . import fast_sampler as fs
        out = fs.serial_index(t.to(device), rows.to(device))
        torch.cuda.current_stream(device).synchronize()
        bits = fs.async_errors(clear=True, device=device)
        if bits & 1:
            raise IndexError(f"row index outside the {t.size(0)}-row table (async error mask {bits})")
        return out
    return t.to(device)[rows.to(device)]


class FastDataset(NamedTuple):
    name: str
    x: torch.Tensor
    y: torch.Tensor
    rowptr: torch.Tensor
    col: torch.Tensor
    split_idx: Mapping[str, torch.Tensor]
    meta_info: Mapping[str, Any]

    @classmethod
    def from_tensors(cls, name, x, y, rowptr, col, split_idx, num_classes):
        y = y.squeeze()
        if y.is_floating_point():                       # dataset.py:74-76
            y = y.nan_to_num_(-1).long()
        return cls(name=name, x=x.to(torch.float16), y=y, rowptr=rowptr, col=col, split_idx=dict(split_idx),
                   meta_info={"num classes": int(num_classes)})

    def get_num_iterations(self, minibatch_size: int):
        return {n: max(1, int(self.split_idx[n].numel() / minibatch_size)) for n in ("train", "valid", "test")}

    @classmethod
    def from_path(cls, _path, name, skip_features=False):
        path = Path(_path).joinpath(name)
        if not (path.exists() and path.is_dir()):
            raise ValueError(f"dataset {name!r} does not exist at {path} (the OGB download path of the reference "
                             f"is not available here)")
        return cls.from_path_if_exists(_path, name, skip_features=skip_features)

    @classmethod
    def from_path_if_exists(cls, path, name, skip_features=False):
        path = Path(path).joinpath(name)
        assert path.exists() and path.is_dir()
        data = {field: torch.load(path.joinpath(field + ".pt"), weights_only=False)
                for field in cls._fields if not skip_features or (field != "y" and field != "x")}
        if not skip_features:
            data["y"] = data["y"].long()
            data["x"] = data["x"].to(torch.float16)
        else:
            data["y"] = torch.tensor([])
            data["x"] = torch.tensor([])
        assert data["name"] == name
        return cls(**data)

    def save(self, path):
        path = Path(path).joinpath(self.name)
        path.mkdir()
        for i, field in enumerate(self._fields):
            torch.save(self[i], path.joinpath(field + ".pt"))

    def adj_t(self):
        from .fast_trainer.monkeypatch import SparseTensor
        return SparseTensor(rowptr=self.rowptr, col=self.col, sparse_sizes=(self.num_nodes, self.num_nodes),
                            is_sorted=True, trust_data=True)

    @property
    def num_nodes(self):
        return self.rowptr.numel() - 1

    def share_memory_(self):
        for t in (self.x, self.y, self.rowptr, self.col, *self.split_idx.values()):
            t.share_memory_()

    @property
    def num_features(self):
        return self.x.size(1)

    @property
    def num_classes(self):
        return int(self.meta_info["num classes"] if "num classes" in self.meta_info else self.meta_info["num_classes"])


class DisjointPartFeatReorderedDataset(NamedTuple):
    """One rank's view of a feature-partitioned, vertex-reordered dataset (dataset.py:145-184)."""
    name: str
    rank: int
    num_parts: int
    x: torch.Tensor
    y: torch.Tensor
    rowptr: torch.Tensor
    col: torch.Tensor
    split_idx: Mapping[str, torch.Tensor]
    split_idx_parts: Mapping[int, Mapping[str, torch.Tensor]]
    part_offsets: torch.Tensor
    meta_info: Mapping[str, Any]

    @classmethod
    def from_path(cls, _path, name, rank):
        path = Path(_path).joinpath(name)
        if not (path.exists() and path.is_dir()):
            raise ValueError("ERROR dataset does not exist at specified path.")
        return cls.from_path_if_exists(_path, name, rank)

    @classmethod
    def from_path_if_exists(cls, path, name, rank):
        path = Path(path).joinpath(name)
        assert path.exists() and path.is_dir()
        some_fields = [f for f in cls._fields if f not in ("x", "rank")]
        data = {f: torch.load(path.joinpath(f + ".pt"), weights_only=False) for f in some_fields}
        data["y"] = data["y"].long()
        data["x"] = torch.load(path.joinpath("x" + str(rank) + ".pt"), weights_only=False).to(torch.float16)
        data["rank"] = rank
        data["num_parts"] = int(data["num_parts"])
        assert data["name"] == name
        return cls(**data)

    @classmethod
    def reorder_and_save(cls, dataset: FastDataset, partition_labels: torch.Tensor,
                         probability_of_access: Optional[torch.Tensor], dir: Path, device=None):
        """Write the partitioned, vertex-reordered form of `dataset` under
        ``dir/metis-reordered-k<P>/<name>/`` (what driver/dataset.py:270-369 produces) and return that
        directory.  New vertex ids: partition by partition, inside a partition by descending access
        probability (1-D: one value per vertex; 2-D [P, N]: row p applies to the vertices of partition p).

        Everything proportional to the graph runs on `device` (default: the current GPU): the vertex
        order, the symmetric CSR relabel + coalesce, the split relabel/bucketing, and the feature rows,
        which are gathered and written ONE PARTITION AT A TIME (the reordered feature matrix of a
        papers-scale dataset is never materialised as a whole)."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
        device = torch.device(device)
        order = _vertex_order(partition_labels, probability_of_access, device)
        P = order.num_parts
        target = Path(dir) / f"metis-reordered-k{P}" / dataset.name
        target.mkdir(parents=True, exist_ok=False)

        rowptr_new, col_new = csr_permute_symmetric(dataset.rowptr.to(device), dataset.col.to(device), order.new_id)
        files = {
            "num_parts": P,
            "rowptr": rowptr_new.cpu(),
            "col": col_new.cpu(),
            "split_idx": dict(),                                   # the reference leaves it empty (:326)
            "split_idx_parts": _bucket_splits(dataset.split_idx, order),
            "part_offsets": order.part_offsets.cpu(),
            "y": _take_rows(dataset.y, order.old_id, device).cpu(),
            "meta_info": dict(dataset.meta_info),
            "name": dataset.name,
        }
        del rowptr_new, col_new
        for stem, value in files.items():
            torch.save(value, target / f"{stem}.pt")
        bounds = files["part_offsets"].tolist()
        x_src = dataset.x.to(device)                               # uploaded once; gathered partition by partition
        for r in range(P):
            rows = _take_rows(x_src, order.old_id[bounds[r]:bounds[r + 1]], device)
            torch.save(rows.to(torch.float16).cpu(), target / f"x{r}.pt")
        return target

    def get_RangePartitionBook(self):
        return RangePartitionBook(self.rank, self.num_parts, self.part_offsets)

    def get_num_iterations(self, minibatch_size: int):
        """Equal iteration counts on every rank (dataset.py:374-392)."""
        out = {}
        for name in ("train", "valid", "test"):
            total = sum(int(self.split_idx_parts[i][name].numel()) for i in range(self.num_parts))
            out[name] = int(max(1, total // minibatch_size))
        return out

    @property
    def num_nodes(self):
        return self.rowptr.numel() - 1

    def adj_t(self):
        from .fast_trainer.monkeypatch import SparseTensor
        return SparseTensor(rowptr=self.rowptr, col=self.col, sparse_sizes=(self.num_nodes, self.num_nodes),
                            is_sorted=True, trust_data=True)

    def share_memory_(self):
        for t in (self.x, self.y, self.rowptr, self.col, *self.split_idx.values()):
            t.share_memory_()
        for v in self.split_idx_parts.values():
            for v2 in v.values():
                v2.share_memory_()

    @property
    def num_features(self):
        return self.x.size(1)

    @property
    def num_classes(self):
        return int(self.meta_info["num classes"] if "num classes" in self.meta_info else self.meta_info["num_classes"])
