"""-m gpu: the distributed boundary fed with the REFERENCE's batch record.

A SALIENT++ checkout that keeps its own fast_trainer/samplers.py wraps every native batch in a 7-field
NamedTuple (fast_trainer/samplers.py:32-88: partition_nids, sliced_cpu_features, sliced_cpu_labels,
cached_nids, perm_partition_to_mfg, adjs, idx_range) -- no `x`, `n_id` or `partition_nids_flat`.
`RefShapedProto` / `RefShapedIter` below are test-local stand-ins with exactly that shape and the way
the reference's FastSamplerIter builds them; they are driven through this repository's
DeviceDistributedPrefetcher on both transports (native exchange on in-process ranks; torch.distributed
with two processes on the one GPU), VIP cache on and off, and every delivered x is compared with
x_full[n_id] of the oracle."""
import os
import sys
import threading
from typing import List, NamedTuple

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

REF_FIELDS = ("partition_nids", "sliced_cpu_features", "sliced_cpu_labels", "cached_nids", "perm_partition_to_mfg",
              "adjs", "idx_range")


class RefShapedProto(NamedTuple):
    """the seven fields of the reference's ProtoDistributedBatch, nothing else"""
    partition_nids: List[torch.Tensor]
    sliced_cpu_features: torch.Tensor
    sliced_cpu_labels: torch.Tensor
    cached_nids: torch.Tensor
    perm_partition_to_mfg: torch.Tensor
    adjs: list
    idx_range: slice

    @classmethod
    def from_fast_sampler(cls, batch):
        from salient_plusplus_amd.fast_trainer.samplers import Adj__from_fast_sampler
        assert batch.sliced_cpu_features != None  # noqa: E711  (the reference's check, samplers.py:71)
        start, stop = batch.idx_range
        return cls(partition_nids=batch.partition_nids, sliced_cpu_features=batch.sliced_cpu_features,
                   sliced_cpu_labels=batch.sliced_cpu_labels, cached_nids=batch.cached_nids,
                   perm_partition_to_mfg=batch.perm_partition_to_mfg,
                   adjs=[Adj__from_fast_sampler(a) for a in batch.adjs], idx_range=slice(start, stop))

    def record_stream(self, stream):          # what samplers.py:94-101 touches
        for part in self.partition_nids:
            part.record_stream(stream)
        self.perm_partition_to_mfg.record_stream(stream)
        for adj in self.adjs:
            adj.record_stream(stream)


class RefShapedIter:
    """the reference's FastSamplerIter (samplers.py:331-357): owns the Session, wraps each batch"""

    def __init__(self, num_threads, max_items_in_queue, cfg):
        from salient_plusplus_amd import fast_sampler
        self.session = fast_sampler.Session(num_threads, max_items_in_queue, cfg.to_fast_sampler())
        assert self.session.num_total_batches == cfg.get_num_batches()

    def __iter__(self):
        return self

    def __next__(self):
        sample = self.session.blocking_get_batch_distributed()
        if sample is None:
            raise StopIteration
        return RefShapedProto.from_fast_sampler(sample)


def test_ref_shaped_record_has_exactly_the_reference_fields():
    assert RefShapedProto._fields == REF_FIELDS


def _check_record(proto, P, n_id_want, cache, use_cache):
    """what a reference-side consumer relies on: P buckets, cat(buckets ++ cache vertices)[perm] == n_id"""
    assert len(proto.partition_nids) == P
    assert proto.perm_partition_to_mfg.numel() == len(n_id_want)
    ids = list(proto.partition_nids)
    if use_cache:
        ids.append(cache.cached_vertices.to(proto.cached_nids.device)[proto.cached_nids])
    else:
        assert proto.cached_nids.numel() == 0
    got = torch.cat(ids)[proto.perm_partition_to_mfg]
    np.testing.assert_array_equal(got.cpu().numpy(), n_id_want)
    assert proto.sliced_cpu_features is not None and proto.sliced_cpu_features.size(0) == 0


def _native_rank(rank, P, comms, g, offsets, use_cache, nb, bs, errors, done):
    it = None
    try:
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        from test_gpu_native_exchange import SIZES, _check_batch, _rank_cfg
        torch.cuda.set_device(0)
        fs.set_native_comm(comms[rank])
        cfg, idx = _rank_cfg(g, rank, P, offsets, use_cache, nb, bs, fs)
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        dev = torch.device("cuda", 0)
        it = RefShapedIter(2, 8, cfg)
        assert it.session.native_exchange and not it.session.compact_native_records
        # the records themselves, as the reference façade would see them (peek through a tee on the iterator)
        seen = []
        real_next = it.__class__.__next__

        class Tee:
            session = it.session

            def __iter__(self):
                return self

            def __next__(self):
                p = real_next(it)
                seen.append(p)
                return p
        got = 0
        for (batch,) in DeviceDistributedPrefetcher([dev], Tee(), True):
            _check_batch(batch, got, ranges, g, idx, g["x"], orc)
            start, stop = int(ranges[got][0]), int(ranges[got][1])
            m = orc.sample_batch(g["rowptr"], g["col"], idx, start, stop, SIZES)
            _check_record(seen[got], P, m.n_id, cfg.cache, use_cache)
            got += 1
        assert got == nb
        it.session.close()
        done[rank] = True
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        comms[rank].close()
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.set_native_comm(None)


@pytest.mark.parametrize("use_cache", [False, True])
@pytest.mark.parametrize("issue", ["consumer", "thread"])
def test_reference_record_native_exchange(use_cache, issue, monkeypatch):
    from salient_plusplus_amd import fast_sampler as fs
    from test_gpu_native_exchange import _graph
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", issue)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    P, nb, bs = 2, 5, 16
    offsets = [0, 1400, n]
    comms = fs.NativeComm.local(P)
    errors, done = [], {}
    ts = [threading.Thread(target=_native_rank, args=(r, P, comms, g, offsets, use_cache, nb, bs, errors, done))
          for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"
    assert all(done.get(r) for r in range(P))


def _torch_worker(rank, port, use_cache, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from test_gpu_dist_2ranks import _staged_all_to_all_single
        dist.all_to_all_single = _staged_all_to_all_single(dist.all_to_all_single)
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        from test_gpu_native_exchange import _check_batch, _graph, _rank_cfg
        g = _graph()
        n = g["rowptr"].shape[0] - 1
        nb, bs = 5, 16
        cfg, idx = _rank_cfg(g, rank, 2, [0, 1400, n], use_cache, nb, bs, fs)
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        it = RefShapedIter(2, 8, cfg)
        assert not it.session.native_exchange        # gloo process group: the exchange stays in Python
        got = 0
        for (batch,) in DeviceDistributedPrefetcher([torch.device("cuda", 0)], it, True):
            _check_batch(batch, got, ranges, g, idx, g["x"], orc)
            got += 1
        assert got == nb
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(f"rank {rank}: {e}\n{traceback.format_exc()}")
        raise


@pytest.mark.parametrize("use_cache", [False, True])
def test_reference_record_torch_transport(use_cache):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29760 + int(use_cache)
    procs = [ctx.Process(target=_torch_worker, args=(r, port, use_cache, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    alive = [p for p in procs if p.is_alive()]
    for p in alive:
        p.kill()
    msgs = []
    while not q.empty():
        msgs.append(q.get())
    assert not alive, "rank(s) hung"
    assert all(p.exitcode == 0 for p in procs), "\n".join(msgs)
