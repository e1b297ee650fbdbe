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
        """dataset.py:270-369.  The heavy steps (CSR relabel + sort, feature permutation) run on
        `device` (default: the GPU when there is one).  Returns the directory written."""
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else "cpu"
        partition_labels = partition_labels.to(torch.int64)
        num_parts = int(partition_labels.max()) + 1
        sizes_partition = torch.bincount(partition_labels, minlength=num_parts)
        # ascending partition id globally, descending access probability inside a partition (:299-320)
        ordering_vals = 2 * (partition_labels.max() - partition_labels.float())
        if probability_of_access is not None:
            if probability_of_access.dim() == 1:
                ordering_vals += probability_of_access.to(ordering_vals.device)
            elif probability_of_access.dim() == 2:
                for part in range(probability_of_access.size(0)):
                    mask = partition_labels == part
                    ordering_vals[mask] += probability_of_access[part][mask].to(ordering_vals.dtype)
            else:
                print(f"*WARNING* Unexpected dimensionality of probability_of_access ({probability_of_access.dim()})")
        perm = ordering_vals.argsort(descending=True, stable=True)      # stable: reproducible ties
        invperm = perm.argsort()

        rowptr_p, col_p = csr_permute_symmetric(dataset.rowptr.to(device), dataset.col.to(device), invperm)
        rowptr_p, col_p = rowptr_p.cpu(), col_p.cpu()

        split_idx_p = dict()                                            # saved empty, as the reference does (:326)
        split_idx_parts = {r: dict() for r in range(num_parts)}
        for k, v in dataset.split_idx.items():
            partition_ids = partition_labels[v]
            local_part_size = torch.bincount(partition_ids, minlength=num_parts)
            local_part_offset = torch.cat((torch.tensor([0]), torch.cumsum(local_part_size, 0)))
            sorted_relabeled = invperm[v][partition_ids.argsort(stable=True)]
            for r in range(num_parts):
                split_idx_parts[r][k] = sorted_relabeled[local_part_offset[r]:local_part_offset[r + 1]]

        x_p = dataset.x[perm]
        y_p = dataset.y[perm]
        part_offsets_p = torch.cat((torch.tensor([0]), torch.cumsum(sizes_partition, 0)))

        prefix = Path(dir) / f"metis-reordered-k{num_parts}" / dataset.name
        prefix.mkdir(parents=True, exist_ok=False)
        torch.save(num_parts, prefix / "num_parts.pt")
        torch.save(rowptr_p, prefix / "rowptr.pt")
        torch.save(col_p, prefix / "col.pt")
        torch.save(split_idx_p, prefix / "split_idx.pt")
        torch.save(split_idx_parts, prefix / "split_idx_parts.pt")
        torch.save(part_offsets_p, prefix / "part_offsets.pt")
        torch.save(y_p, prefix / "y.pt")
        torch.save(dict(dataset.meta_info), prefix / "meta_info.pt")
        torch.save(dataset.name, prefix / "name.pt")
        for r in range(num_parts):
            torch.save(x_p[part_offsets_p[r]:part_offsets_p[r + 1]].to(torch.float16).clone(), prefix / f"x{r}.pt")
        return prefix

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
