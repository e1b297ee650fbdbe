"""``Adj`` / ``SparseTensor`` shims (reference: fast_trainer/monkeypatch.py:25-69).

When PyG and torch_sparse are importable their types are used and patched exactly like the
reference does (``Adj.pin_memory``, ``Adj.record_stream``, ``EdgeIndex.pin_memory``).  When they
are absent (this image), minimal stand-ins with the same attribute surface are provided so the
data path and the transferers work unchanged: ``Adj(adj_t, e_id, size)`` and a CSR holder with
``.to`` / ``.record_stream`` / ``.pin_memory`` / ``.csr()`` / ``.sparse_sizes()``.
"""
from typing import NamedTuple, Optional, Tuple

import torch

builtins = __builtins__ if isinstance(__builtins__, dict) else vars(__builtins__)
if 'profile' not in builtins:
    def profile(func):          # line_profiler no-op (monkeypatch.py:15-21)
        return func
else:
    profile = builtins['profile']

try:  # pragma: no cover - depends on the environment
    from torch_sparse import SparseTensor      # type: ignore
    HAVE_TORCH_SPARSE = True
except Exception:  # noqa: BLE001
    HAVE_TORCH_SPARSE = False

    class SparseTensor:                         # minimal CSR holder, API subset of torch_sparse.SparseTensor
        def __init__(self, rowptr=None, row=None, col=None, value=None, sparse_sizes=None,
                     is_sorted=False, trust_data=False):
            assert rowptr is not None and col is not None, "the fallback SparseTensor is CSR only"
            self._rowptr, self._col, self._value = rowptr, col, value
            self._sparse_sizes = tuple(int(s) for s in sparse_sizes)

        # torch_sparse surface used by the data path / simple models
        def csr(self):
            return self._rowptr, self._col, self._value

        def sparse_sizes(self) -> Tuple[int, int]:
            return self._sparse_sizes

        def sparse_size(self, dim: int) -> int:
            return self._sparse_sizes[dim]

        def size(self, dim: int) -> int:
            return self._sparse_sizes[dim]

        def nnz(self) -> int:
            return self._col.numel()

        @property
        def storage(self):
            return self

        def rowptr(self):
            return self._rowptr

        def col(self):
            return self._col

        def to(self, *args, **kwargs):
            v = self._value.to(*args, **kwargs) if self._value is not None else None
            return SparseTensor(rowptr=self._rowptr.to(*args, **kwargs), col=self._col.to(*args, **kwargs),
                                value=v, sparse_sizes=self._sparse_sizes, is_sorted=True, trust_data=True)

        def pin_memory(self):
            if self._rowptr.is_cuda:
                return self
            v = self._value.pin_memory() if self._value is not None else None
            return SparseTensor(rowptr=self._rowptr.pin_memory(), col=self._col.pin_memory(), value=v,
                                sparse_sizes=self._sparse_sizes, is_sorted=True, trust_data=True)

        def record_stream(self, stream):
            for t in (self._rowptr, self._col, self._value):
                if t is not None and t.is_cuda:
                    t.record_stream(stream)

        def to_torch_sparse_csr_tensor(self, dtype=torch.float32):
            vals = self._value if self._value is not None else \
                torch.ones(self._col.numel(), dtype=dtype, device=self._col.device)
            return torch.sparse_csr_tensor(self._rowptr, self._col, vals, size=self._sparse_sizes)

        def __repr__(self):
            return f"SparseTensor(csr, sparse_sizes={self._sparse_sizes}, nnz={self.nnz()})"

try:  # pragma: no cover - depends on the environment
    import torch_geometric                      # type: ignore
    if torch_geometric.__version__ < '2.0.0':
        from torch_geometric.data.sampler import Adj, EdgeIndex      # type: ignore
    else:
        from torch_geometric.loader.neighbor_sampler import Adj, EdgeIndex   # type: ignore
    HAVE_PYG = True
except Exception:  # noqa: BLE001
    HAVE_PYG = False

    class Adj(NamedTuple):
        adj_t: SparseTensor
        e_id: Optional[torch.Tensor]
        size: Tuple[int, int]

        def to(self, *args, **kwargs):
            adj_t = self.adj_t.to(*args, **kwargs)
            e_id = self.e_id.to(*args, **kwargs) if self.e_id is not None else None
            return Adj(adj_t, e_id, self.size)

    class EdgeIndex(NamedTuple):
        edge_index: torch.Tensor
        e_id: Optional[torch.Tensor]
        size: Tuple[int, int]

        def to(self, *args, **kwargs):
            edge_index = self.edge_index.to(*args, **kwargs)
            e_id = self.e_id.to(*args, **kwargs) if self.e_id is not None else None
            return EdgeIndex(edge_index, e_id, self.size)


def _sparse_record_stream(adj_t, stream):
    if HAVE_TORCH_SPARSE:
        st = adj_t.storage
        for name in ("_row", "_rowptr", "_col", "_value", "_rowcount", "_colptr", "_colcount",
                     "_csr2csc", "_csc2csr"):
            t = getattr(st, name, None)
            if t is not None and t.is_cuda:
                t.record_stream(stream)
    else:
        adj_t.record_stream(stream)


def _adj_pin_memory(self, *args, **kwargs):
    adj = self.adj_t.pin_memory(*args, **kwargs)
    e_id = self.e_id.pin_memory(*args, **kwargs) if self.e_id is not None and not self.e_id.is_cuda else self.e_id
    return type(self)(adj, e_id, self.size)


def _adj_record_stream(self, stream):
    _sparse_record_stream(self.adj_t, stream)
    if self.e_id is not None and self.e_id.is_cuda:
        self.e_id.record_stream(stream)


def _edge_index_pin_memory(self, *args, **kwargs):
    edge_index = self.edge_index.pin_memory(*args, **kwargs)
    e_id = self.e_id.pin_memory(*args, **kwargs) if self.e_id is not None else None
    return type(self)(edge_index, e_id, self.size)


Adj.pin_memory = _adj_pin_memory
Adj.record_stream = _adj_record_stream
EdgeIndex.pin_memory = _edge_index_pin_memory
