"""Batch records, sampler configuration and sampler iterables of the training façade.

This module offers the names the rest of SALIENT++ imports from ``fast_trainer.samplers`` -- the batch
records (``PreparedBatch``, ``ProtoBatch``, ``ProtoDistributedBatch``, ``NumpyProtoDistributedBatch``),
``Adj__from_fast_sampler``,
``FastSamplerConfig``, ``FastSampler`` / ``FastSamplerIter`` / ``FastPreSampler`` and the two statistics
records -- with the field sets and call signatures of the reference (fast_trainer/samplers.py:22-30,
:32-165, :213-268, :271-305, :331-423), on top of the MI355X ``fast_sampler`` module.

Every tensor of a batch already lives in HBM when an iterator yields it: moving a batch ``.to`` its
own device returns the same storage and ``record_stream`` is what makes hand-over between streams safe.
"""
import dataclasses
import datetime
import itertools
from abc import abstractmethod
from typing import Iterable, Iterator, List, NamedTuple, Optional, Sized

import torch

from .. import fast_sampler
from ..fast_sampler import Cache, RangePartitionBook
from .monkeypatch import Adj, SparseTensor


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _mark_in_use(stream, *tensors):
    """record_stream on every CUDA tensor given (None and host tensors are skipped)"""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)


def _moved(t, device, non_blocking):
    return None if t is None else t.to(device=device, non_blocking=non_blocking)


def Adj__from_fast_sampler(adj) -> Adj:
    """One hop as the native module returns it, ``(rowptr, col, e_id, (T, S))``, becomes the PyG-style
    ``Adj``: a CSR ``SparseTensor`` of T target rows over S source columns, the (empty) edge ids and
    ``size = (S, T)``."""
    row_pointers, columns, edge_ids, (n_targets, n_sources) = adj
    matrix = SparseTensor(rowptr=row_pointers, row=None, col=columns, value=None,
                          sparse_sizes=(n_targets, n_sources), is_sorted=True, trust_data=True)
    return Adj(matrix, edge_ids, (n_sources, n_targets))


def _hops(raw_adjs) -> List[Adj]:
    return [Adj__from_fast_sampler(h) for h in raw_adjs]


def _as_slice(bounds) -> slice:
    first, last = bounds
    return slice(first, last)


# --------------------------------------------------------------------------------------------
# batch records
# --------------------------------------------------------------------------------------------
class PreparedBatch(NamedTuple):
    """A batch ready for the model: features of all MFG nodes (targets first), labels of the seeds,
    the hops outermost first and the seed range it covers."""
    x: torch.Tensor
    y: Optional[torch.Tensor]
    adjs: List[Adj]
    idx_range: slice

    # -- constructors --
    @classmethod
    def from_fast_sampler(cls, prepared_sample):
        feats, labels, raw_adjs, bounds = prepared_sample
        return cls(feats, None if labels is None else labels.squeeze(), _hops(raw_adjs), _as_slice(bounds))

    @classmethod
    def from_proto_batch(cls, x, y, proto_batch: "ProtoBatch"):
        nodes = proto_batch.n_id
        labels = None if y is None else y[nodes[:proto_batch.batch_size]]
        return cls(x[nodes], labels, proto_batch.adjs, proto_batch.idx_range)

    # -- movement / stream safety --
    def to(self, device, non_blocking=False):
        return PreparedBatch(_moved(self.x, device, non_blocking), _moved(self.y, device, non_blocking),
                             [hop.to(device=device, non_blocking=non_blocking) for hop in self.adjs],
                             self.idx_range)

    def record_stream(self, stream):
        _mark_in_use(stream, self.x, self.y)
        for hop in self.adjs:
            hop.record_stream(stream)

    # -- sizes --
    @property
    def batch_size(self):
        return self.idx_range.stop - self.idx_range.start

    @property
    def num_total_nodes(self):
        return self.x.size(0)


class ProtoBatch(NamedTuple):
    """A sampled batch without features: MFG node ids, hops, seed range."""
    n_id: torch.Tensor
    adjs: List[Adj]
    idx_range: slice

    @classmethod
    def from_fast_sampler(cls, proto_sample):
        nodes, raw_adjs, bounds = proto_sample
        return cls(nodes, _hops(raw_adjs), _as_slice(bounds))

    @property
    def batch_size(self):
        return self.idx_range.stop - self.idx_range.start


class ProtoDistributedBatch(NamedTuple):
    """One rank's sampled batch before the feature exchange.  With ``feat(ids)`` the rows of the
    owner of ``ids``, the features in MFG order are
    ``cat([feat(partition_nids[0]), ..., feat(partition_nids[P-1]), cache[cached_nids]])[perm_partition_to_mfg]``;
    ``cached_nids`` index the cache's own rows.  The GPU session adds the MFG node ids, and -- when the
    exchange ran natively -- ``x``, the finished feature matrix."""
    partition_nids: List[torch.Tensor]
    sliced_cpu_features: torch.Tensor
    sliced_cpu_labels: torch.Tensor
    cached_nids: torch.Tensor
    perm_partition_to_mfg: torch.Tensor
    adjs: List[Adj]
    idx_range: slice
    n_id: Optional[torch.Tensor] = None
    x: Optional[torch.Tensor] = None
    partition_nids_flat: Optional[torch.Tensor] = None      # all partition_nids as one buffer, when they share one

    @classmethod
    def from_fast_sampler(cls, batch):
        if batch.sliced_cpu_features is None:
            raise AssertionError("the native batch carries no sliced_cpu_features")
        extras = {name: getattr(batch, name, None) for name in ("n_id", "x", "partition_nids_flat")}
        return cls(batch.partition_nids, batch.sliced_cpu_features, batch.sliced_cpu_labels, batch.cached_nids,
                   batch.perm_partition_to_mfg, _hops(batch.adjs), _as_slice(batch.idx_range), **extras)

    def to(self, device, stream=None, non_blocking=False, streams_to_sync=None, delay_feature_transfer=True):
        with torch.cuda.stream(stream):
            changed = dict(
                adjs=[hop.to(device, non_blocking=non_blocking) for hop in self.adjs],
                partition_nids=[ids.to(device, non_blocking=non_blocking) for ids in self.partition_nids],
                perm_partition_to_mfg=self.perm_partition_to_mfg.to(device, non_blocking=non_blocking))
            if not delay_feature_transfer:
                changed["sliced_cpu_features"] = self.sliced_cpu_features.to(device, non_blocking=non_blocking)
        return self._replace(**changed)

    def record_stream(self, stream):
        _mark_in_use(stream, *self.partition_nids, self.perm_partition_to_mfg, self.cached_nids, self.n_id, self.x)
        for hop in self.adjs:
            hop.record_stream(stream)

    @property
    def num_total_nodes(self):
        return self.perm_partition_to_mfg.size(0)

    @property
    def num_cached_nodes(self):
        return self.cached_nids.size(0)

    def get_num_local_nodes(self, local_rank):
        return self.partition_nids[local_rank].numel()

    def get_num_communicated_nodes(self, local_rank):
        return sum(ids.numel() for owner, ids in enumerate(self.partition_nids) if owner != local_rank)


class NumpyProtoDistributedBatch(NamedTuple):
    """A ProtoDistributedBatch as host arrays, for saving batch statistics to .npz and plotting them (reference
    fast_trainer/samplers.py:167-196: the same five fields; there ``from_proto_batch`` is switched off with an
    ``assert False`` and its only caller, DeviceDistributedPrefetcher.collect_data, never reaches it).  Here the
    conversion works: every tensor of the batch is copied out of HBM, so it is for diagnostics, not for the
    training loop.  Each hop becomes a ``scipy.sparse.csr_matrix`` of T target rows over S source columns with
    unit edge data."""
    partition_nids: list
    cache_specific_nids: object
    perm_partition_to_mfg: object
    adjs: list
    seed_indices: object

    @classmethod
    def from_proto_batch(cls, batch: "ProtoDistributedBatch", ids: torch.Tensor):
        import numpy as np
        import scipy.sparse

        def host(t):
            return t.detach().cpu().numpy()

        hops = []
        for hop in batch.adjs:
            rowptr, col, _ = hop.adj_t.csr()
            t, s = hop.adj_t.sparse_sizes()
            hops.append(scipy.sparse.csr_matrix((np.ones(col.numel(), dtype=np.float32), host(col), host(rowptr)), shape=(t, s)))
        return cls(partition_nids=[host(n) for n in batch.partition_nids], cache_specific_nids=host(batch.cached_nids),
                   perm_partition_to_mfg=host(batch.perm_partition_to_mfg), adjs=hops, seed_indices=host(ids[batch.idx_range]))


# --------------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------------
@dataclasses.dataclass
class FastSamplerConfig:
    """Everything a Session needs; the field names are those of the native ``Config``."""
    x_cpu: torch.Tensor
    x_gpu: torch.Tensor
    y: torch.Tensor
    rowptr: torch.Tensor
    col: torch.Tensor
    idx: torch.Tensor
    batch_size: int
    sizes: List[int]
    skip_nonfull_batch: bool
    pin_memory: bool
    distributed: bool
    partition_book: RangePartitionBook
    cache: Cache
    force_exact_num_batches: bool
    exact_num_batches: int
    count_remote_frequency: bool
    use_cache: bool

    def to_fast_sampler(self) -> fast_sampler.Config:
        native = fast_sampler.Config()
        for f in dataclasses.fields(self):
            # a single-GPU configuration has no partition book to hand over
            if f.name == "partition_book" and not self.distributed:
                continue
            setattr(native, f.name, getattr(self, f.name))
        return native

    def get_num_batches(self) -> int:
        """Number of batches a Session over this configuration will deliver."""
        if self.force_exact_num_batches:
            return self.exact_num_batches
        full, rest = divmod(self.idx.numel(), self.batch_size)
        return full + (1 if rest and not self.skip_nonfull_batch else 0)


# --------------------------------------------------------------------------------------------
# statistics records
# --------------------------------------------------------------------------------------------
class FastSamplerStats(NamedTuple):
    total_blocked_dur: datetime.timedelta
    total_blocked_occasions: int

    @classmethod
    def from_session(cls, session):
        return cls(session.total_blocked_dur, session.total_blocked_occasions)


class FastSamplerDistributedStats(NamedTuple):
    remote_frequency_tensor: torch.Tensor
    remote_vertices_ordered_by_freq: torch.Tensor

    @classmethod
    def from_session(cls, session):
        if session.num_consumed_batches != session.num_total_batches:
            raise AssertionError("remote-frequency statistics are read after the epoch has been consumed")
        session.reduce_multithreaded_frequency_counts()
        return cls(session.remote_frequency_tensor, session.remote_vertices_ordered_by_freq)


# --------------------------------------------------------------------------------------------
# iterables
# --------------------------------------------------------------------------------------------
class FastSamplerIter(Iterator[PreparedBatch]):
    """One epoch: owns the native Session (``.session``) and yields its batches in index order."""
    session: fast_sampler.Session

    def __init__(self, num_threads: int, max_items_in_queue: int, cfg: FastSamplerConfig, table_features: bool = False,
                 row_refs: bool = False):
        self.session = fast_sampler.Session(num_threads, max_items_in_queue, cfg.to_fast_sampler())
        if row_refs:
            # opt-in, partitioned sessions with the native exchange / P2P transport: PreparedBatch.x is
            # fast_sampler.RowRefs (where every row lives) and the delivery assembles nothing; models.SAGE aggregates its
            # first layer from the addresses, anything else calls .materialize()
            self.session.row_refs = True
        if table_features:
            # opt-in: PreparedBatch.x is fast_sampler.TableRows(resident table, n_id) and the delivery skips the feature
            # gather; models.SAGE aggregates its first layer straight from the table, anything else calls .materialize()
            self.session.table_features = True
        # this façade's records have fields for the assembled features and the MFG ids: the native
        # exchange need not also export the ownership buckets a reference-shaped record would read
        self.session.compact_native_records = True
        expected = cfg.get_num_batches()
        if self.session.num_total_batches != expected:
            raise AssertionError(f"session plans {self.session.num_total_batches} batches, the configuration {expected}")

    def __next__(self):
        distributed = self.session.config.distributed
        raw = self.session.blocking_get_batch_distributed() if distributed else self.session.blocking_get_batch()
        if raw is None:
            raise StopIteration
        return (ProtoDistributedBatch if distributed else PreparedBatch).from_fast_sampler(raw)

    def get_stats(self) -> FastSamplerStats:
        return FastSamplerStats.from_session(self.session)

    def get_distributed_stats(self) -> FastSamplerDistributedStats:
        return FastSamplerDistributedStats.from_session(self.session)


class ABCNeighborSampler(Iterable[PreparedBatch], Sized):
    """A sampler whose seed set can be replaced between epochs."""

    @property
    @abstractmethod
    def idx(self) -> torch.Tensor:
        ...

    @idx.setter
    @abstractmethod
    def idx(self, idx: torch.Tensor) -> None:
        ...


@dataclasses.dataclass
class FastSampler(ABCNeighborSampler):
    num_threads: int
    max_items_in_queue: int
    cfg: FastSamplerConfig
    # not in the reference (samplers.py:381-399 has the three fields above): see FastSamplerIter
    table_features: bool = False
    row_refs: bool = False

    def __iter__(self):
        return FastSamplerIter(self.num_threads, self.max_items_in_queue, self.cfg, self.table_features, self.row_refs)

    def __len__(self):
        return self.cfg.get_num_batches()

    # the epoch's seeds and the feature cache live in the configuration
    @property
    def idx(self):
        return self.cfg.idx

    @idx.setter
    def idx(self, idx: torch.Tensor) -> None:
        self.cfg.idx = idx

    @property
    def cache(self):
        return self.cfg.cache

    @cache.setter
    def cache(self, cache: Cache) -> None:
        self.cfg.cache = cache


@dataclasses.dataclass
class FastPreSampler(ABCNeighborSampler):
    """Samples the whole epoch up front with ``full_sample`` and replays it.  (The reference reads a
    non-existent ``cfg.x`` here, samplers.py:402-423; the features come from ``cfg.x_cpu``.)"""
    cfg: FastSamplerConfig

    def __iter__(self) -> Iterator[PreparedBatch]:
        c = self.cfg
        per_thread = fast_sampler.full_sample(c.x_cpu, c.y, c.rowptr, c.col, c.idx, c.batch_size, c.sizes,
                                              c.skip_nonfull_batch, c.pin_memory)
        return map(PreparedBatch.from_fast_sampler, itertools.chain.from_iterable(per_thread))

    def __len__(self):
        return self.cfg.get_num_batches()

    @property
    def idx(self):
        return self.cfg.idx

    @idx.setter
    def idx(self, idx: torch.Tensor) -> None:
        self.cfg.idx = idx
