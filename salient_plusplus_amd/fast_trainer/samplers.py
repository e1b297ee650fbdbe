"""Python façade over the MI355X ``fast_sampler`` module -- same public names and field sets as the
reference's fast_trainer/samplers.py (``FastSamplerConfig`` :271-305, ``FastSampler`` :372-399,
``FastSamplerIter`` :331-357, ``PreparedBatch`` :213-268, ``ProtoDistributedBatch`` :32-165,
``Adj__from_fast_sampler`` :22-30), so ``fast_trainer.train`` / the drivers run unchanged.

All tensors of a batch are already in HBM when the iterator yields them; ``.to(device)`` and
``.pin_memory()`` on them are therefore no-ops and ``record_stream`` is what keeps them safe
across streams.
"""
import datetime
import itertools
from abc import abstractmethod
from dataclasses import dataclass, fields
from typing import Iterable, Iterator, List, NamedTuple, Optional, Sized

import torch

from .. import fast_sampler
from ..fast_sampler import Cache, RangePartitionBook
from .monkeypatch import Adj, SparseTensor


def Adj__from_fast_sampler(adj) -> Adj:
    """(rowptr, col, e_id, (T, S)) -> Adj(SparseTensor[T x S] CSR, e_id, size=(S, T))  (samplers.py:22-30)."""
    rowptr, col, e_id, sparse_sizes = adj
    adj_t = SparseTensor(rowptr=rowptr, row=None, col=col, value=None, sparse_sizes=sparse_sizes,
                         is_sorted=True, trust_data=True)
    return Adj(adj_t, e_id, sparse_sizes[::-1])


class ProtoDistributedBatch(NamedTuple):
    """Sampled, not yet feature-complete batch of one rank (samplers.py:32-165):
    ``cat([feat(partition_nids[0]), ..., feat(partition_nids[P-1]), cache[cached_nids]])[perm]``
    is the feature matrix in MFG order."""
    partition_nids: List[torch.Tensor]
    sliced_cpu_features: torch.Tensor
    sliced_cpu_labels: torch.Tensor
    cached_nids: torch.Tensor          # indices INTO cache.cached_features
    perm_partition_to_mfg: torch.Tensor
    adjs: List[Adj]
    idx_range: slice
    n_id: Optional[torch.Tensor] = None    # MFG node ids (extra of the GPU path)
    x: Optional[torch.Tensor] = None       # native exchange: features already assembled in MFG order
    partition_nids_flat: Optional[torch.Tensor] = None   # cat(partition_nids) when they share one buffer

    @classmethod
    def from_fast_sampler(cls, batch):
        assert batch.sliced_cpu_features is not None
        start, stop = batch.idx_range
        return cls(partition_nids=batch.partition_nids,
                   sliced_cpu_features=batch.sliced_cpu_features,
                   sliced_cpu_labels=batch.sliced_cpu_labels,
                   cached_nids=batch.cached_nids,
                   perm_partition_to_mfg=batch.perm_partition_to_mfg,
                   adjs=[Adj__from_fast_sampler(a) for a in batch.adjs],
                   idx_range=slice(start, stop),
                   n_id=getattr(batch, "n_id", None),
                   x=getattr(batch, "x", None),
                   partition_nids_flat=getattr(batch, "partition_nids_flat", None))

    def record_stream(self, stream):
        for part in self.partition_nids:
            if part.is_cuda:
                part.record_stream(stream)
        for t in (self.perm_partition_to_mfg, self.cached_nids, self.n_id, self.x):
            if t is not None and t.is_cuda:
                t.record_stream(stream)
        for adj in self.adjs:
            adj.record_stream(stream)

    def to(self, device, stream=None, non_blocking=False, streams_to_sync=None, delay_feature_transfer=True):
        with torch.cuda.stream(stream):
            adjs = [adj.to(device, non_blocking=non_blocking) for adj in self.adjs]
            parts = [p.to(device, non_blocking=non_blocking) for p in self.partition_nids]
            perm = self.perm_partition_to_mfg.to(device, non_blocking=non_blocking)
            feats = self.sliced_cpu_features
            if not delay_feature_transfer:
                feats = feats.to(device, non_blocking=non_blocking)
        return self._replace(adjs=adjs, partition_nids=parts, perm_partition_to_mfg=perm,
                             sliced_cpu_features=feats)

    @property
    def num_total_nodes(self):
        return self.perm_partition_to_mfg.size(0)

    @property
    def num_cached_nodes(self):
        return self.cached_nids.size(0)

    def get_num_local_nodes(self, local_rank):
        return self.partition_nids[local_rank].numel()

    def get_num_communicated_nodes(self, local_rank):
        return sum(p.numel() for i, p in enumerate(self.partition_nids) if i != local_rank)


class ProtoBatch(NamedTuple):
    n_id: torch.Tensor
    adjs: List[Adj]
    idx_range: slice

    @classmethod
    def from_fast_sampler(cls, proto_sample):
        n_id, adjs, (start, stop) = proto_sample
        return cls(n_id=n_id, adjs=[Adj__from_fast_sampler(a) for a in adjs], idx_range=slice(start, stop))

    @property
    def batch_size(self):
        return self.idx_range.stop - self.idx_range.start


class PreparedBatch(NamedTuple):
    """(x [U,F] in MFG order, y, adjs outermost hop first, idx_range)  (samplers.py:213-268)."""
    x: torch.Tensor
    y: Optional[torch.Tensor]
    adjs: List[Adj]
    idx_range: slice

    @classmethod
    def from_proto_batch(cls, x, y, proto_batch: ProtoBatch):
        return cls(x=x[proto_batch.n_id],
                   y=y[proto_batch.n_id[:proto_batch.batch_size]] if y is not None else None,
                   adjs=proto_batch.adjs, idx_range=proto_batch.idx_range)

    @classmethod
    def from_fast_sampler(cls, prepared_sample):
        x, y, adjs, (start, stop) = prepared_sample
        return cls(x=x, y=y.squeeze() if y is not None else None,
                   adjs=[Adj__from_fast_sampler(a) for a in adjs], idx_range=slice(start, stop))

    def record_stream(self, stream):
        if self.x is not None and self.x.is_cuda:
            self.x.record_stream(stream)
        if self.y is not None and self.y.is_cuda:
            self.y.record_stream(stream)
        for adj in self.adjs:
            adj.record_stream(stream)

    def to(self, device, non_blocking=False):
        return PreparedBatch(
            x=self.x.to(device=device, non_blocking=non_blocking) if self.x is not None else None,
            y=self.y.to(device=device, non_blocking=non_blocking) if self.y is not None else None,
            adjs=[adj.to(device=device, non_blocking=non_blocking) for adj in self.adjs],
            idx_range=self.idx_range)

    @property
    def num_total_nodes(self):
        return self.x.size(0)

    @property
    def batch_size(self):
        return self.idx_range.stop - self.idx_range.start


@dataclass
class FastSamplerConfig:
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
        c = fast_sampler.Config()
        for field in fields(self):
            if not self.distributed and field.name == 'partition_book':
                continue
            setattr(c, field.name, getattr(self, field.name))
        return c

    def get_num_batches(self) -> int:
        if self.force_exact_num_batches:
            return self.exact_num_batches
        num_batches, r = divmod(self.idx.numel(), self.batch_size)
        if not self.skip_nonfull_batch and r > 0:
            num_batches += 1
        return num_batches


class FastSamplerStats(NamedTuple):
    total_blocked_dur: datetime.timedelta
    total_blocked_occasions: int

    @classmethod
    def from_session(cls, session):
        return cls(total_blocked_dur=session.total_blocked_dur,
                   total_blocked_occasions=session.total_blocked_occasions)


class FastSamplerDistributedStats(NamedTuple):
    remote_frequency_tensor: torch.Tensor
    remote_vertices_ordered_by_freq: torch.Tensor

    @classmethod
    def from_session(cls, session):
        assert session.num_consumed_batches == session.num_total_batches
        session.reduce_multithreaded_frequency_counts()
        return cls(remote_frequency_tensor=session.remote_frequency_tensor,
                   remote_vertices_ordered_by_freq=session.remote_vertices_ordered_by_freq)


class FastSamplerIter(Iterator[PreparedBatch]):
    session: fast_sampler.Session

    def __init__(self, num_threads: int, max_items_in_queue: int, cfg: FastSamplerConfig):
        ncfg = cfg.to_fast_sampler()
        self.session = fast_sampler.Session(num_threads, max_items_in_queue, ncfg)
        assert self.session.num_total_batches == cfg.get_num_batches()

    def __next__(self):
        if not self.session.config.distributed:
            sample = self.session.blocking_get_batch()
            if sample is None:
                raise StopIteration
            return PreparedBatch.from_fast_sampler(sample)
        sample = self.session.blocking_get_batch_distributed()
        if sample is None:
            raise StopIteration
        return ProtoDistributedBatch.from_fast_sampler(sample)

    def get_stats(self) -> FastSamplerStats:
        return FastSamplerStats.from_session(self.session)

    def get_distributed_stats(self) -> FastSamplerDistributedStats:
        return FastSamplerDistributedStats.from_session(self.session)


class ABCNeighborSampler(Iterable[PreparedBatch], Sized):
    @property
    @abstractmethod
    def idx(self) -> torch.Tensor:
        ...

    @idx.setter
    @abstractmethod
    def idx(self, idx: torch.Tensor) -> None:
        ...


@dataclass
class FastSampler(ABCNeighborSampler):
    num_threads: int
    max_items_in_queue: int
    cfg: FastSamplerConfig

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

    def __iter__(self):
        return FastSamplerIter(self.num_threads, self.max_items_in_queue, self.cfg)

    def __len__(self):
        return self.cfg.get_num_batches()


@dataclass
class FastPreSampler(ABCNeighborSampler):
    """Samples the whole epoch up front (samplers.py:402-423; the reference reads a non-existent
    ``cfg.x`` there -- here the features come from ``cfg.x_cpu``)."""
    cfg: FastSamplerConfig

    @property
    def idx(self):
        return self.cfg.idx

    @idx.setter
    def idx(self, idx: torch.Tensor) -> None:
        self.cfg.idx = idx

    def __iter__(self) -> Iterator[PreparedBatch]:
        cfg = self.cfg
        p = fast_sampler.full_sample(cfg.x_cpu, cfg.y, cfg.rowptr, cfg.col, cfg.idx, cfg.batch_size,
                                     cfg.sizes, cfg.skip_nonfull_batch, cfg.pin_memory)
        return (PreparedBatch.from_fast_sampler(s) for s in itertools.chain(*p))

    def __len__(self):
        return self.cfg.get_num_batches()
