"""Device iterators: same protocol as the reference's fast_trainer/transferers.py
(``DeviceIterator`` :20-29, ``DevicePrefetcher`` :890-970, ``DeviceTransferer`` :973-985,
``DeviceDistributedPrefetcher`` :33-887) -- ctor ``(devices, it, pipeline_on=True)``, attributes
``devices`` / ``device`` / ``it``, ``__next__() -> [PreparedBatch]`` raising ``StopIteration`` after
a device synchronise, ``print_stats()``, ``NUMBER_OF_SENT_BYTES``.

MI355X re-design of the distributed iterator.  The reference runs a 10-stage pipeline with three
list-form NCCL all_to_alls, two forced D2H syncs, host-side slicing of host-resident rows and a
zeros+scatter+cat+permute assembly (~4x the algorithmic feature bytes) PER BATCH.  Here every feature row of
the rank is HBM resident and the sampler already produced the ownership buckets on the GPU, so:

  * native transport (default with an NCCL/RCCL process group): the Session runs ONE exchange per GROUP of
    batches below the C ABI (all-gather of the request counts, grouped send/recv of int32 ids, one row gather,
    grouped send/recv of the rows; csrc/session.hip) and delivers the whole group -- MFG, labels and x assembled
    from {local partition, received rows, VIP cache} -- with one launch.  This iterator then only double-buffers:
    it hands out the batch fetched during the previous call and waits for that batch's group event;
  * torch.distributed transport (SPP_DIST_TRANSPORT=torch, non-NCCL backends, reference-shaped producers): the
    same exchange with `all_to_all_single`, also once per group of batches (counts [P x G], ids and rows
    peer-major across the group's batches, one fused assembly kernel per batch), software-pipelined two groups
    deep on one side stream: counts of group g+2, ids/serve/rows of group g+1, assembly of group g.
"""
import ctypes as C
import time
from collections import deque
from typing import Iterator, List, Optional

import torch
import torch.distributed as dist

from .samplers import NumpyProtoDistributedBatch, PreparedBatch, ProtoBatch, ProtoDistributedBatch
from .utils import runtime_stats_cuda

# nanoseconds per timer name, summed over the calls of aggregate_time (reference transferers.py:13-18: the hook its
# Timer objects report to; reset by DeviceDistributedPrefetcher.print_stats)
aggregate_time_results = dict()


def aggregate_time(result):
    """Timer callback: adds ``result.nanos`` to the running total kept under ``result.name``."""
    aggregate_time_results[result.name] = aggregate_time_results.get(result.name, 0) + result.nanos


class DeviceIterator(Iterator[List[PreparedBatch]]):
    """Abstract iterator returning one PreparedBatch per device (transferers.py:20-29)."""
    devices: List[torch.device]

    def __init__(self, devices):
        assert len(devices) > 0
        self.devices = devices
        self.device = self.devices[0]

    def print_stats(self):
        return


def _is_cuda(device) -> bool:
    return torch.device(device).type == "cuda"


def _ready_event(it):
    """the event after which the batch `it` returned last is complete (GPU Session: one per delivered group), or None"""
    return getattr(getattr(it, "session", None), "last_ready_event", None)


class _HipFeatureOps:
    """Feature movement of the distributed iterator on the HIP kernels (product path)."""

    def __init__(self):
        from .. import _native as nat
        self.nat = nat
        self.L = nat.load()
        nat.require_device()

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    @staticmethod
    def _row_stride(t):
        """bytes between rows (the resident tables are padded to the HBM fetch granule); 0 = dense"""
        return int(t.stride(0)) * t.element_size() if t is not None and t.dim() == 2 and t.size(0) > 1 else 0

    def gather_rows(self, x: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        out = torch.empty((idx.numel(), x.size(1)), dtype=x.dtype, device=x.device)
        self.nat.check(self.L.spp_gather_rows_strided(self._p(x), x.size(0), x.size(1) * x.element_size(),
                                                      self._row_stride(x), self._p(idx), 8, idx.numel(), idx.numel(),
                                                      self._p(out), self._stream()))
        return out

    def check_async(self):
        """raise if a kernel met a row index outside its table (include/spp.h spp_async_errors): a peer
        asked for rows this rank does not own, i.e. the ranks disagree on the partition book"""
        bits = self.L.spp_async_errors(torch.cuda.current_device(), 1)
        if bits > 0:
            raise RuntimeError(f"feature exchange: row index outside its table (async error mask {bits}: "
                               "1 = gather index, 2 = served id, 4 = assembly source)")

    def assemble(self, n_id, perm, seg_start: List[int], P: int, rank: int, rank_offset: int, x_local, recv,
                 cache_feats, cached_nids, recv_base: Optional[List[int]] = None) -> torch.Tensor:
        """recv_base[m]: row of `recv` where peer m's rows for THIS batch start (None: rows packed in
        partition order with the own segment absent)"""
        U = n_id.numel()
        out = torch.empty((U, x_local.size(1)), dtype=x_local.dtype, device=x_local.device)
        seg = (C.c_int64 * (P + 2))(*seg_start)
        base = (C.c_int64 * P)(*recv_base) if recv_base is not None else None
        self.nat.check(self.L.spp_assemble_features(
            self._p(n_id), self._p(perm), U, seg, P, rank, rank_offset, self._p(x_local), x_local.size(0),
            self._p(recv), self._p(cache_feats), self._p(cached_nids), x_local.size(1) * x_local.element_size(),
            self._row_stride(x_local), self._row_stride(cache_feats), base, self._p(out), self._stream()))
        return out


class _NoContext:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NO_CTX = _NoContext()


class _SideStream:
    """`with` context running on a side stream of a CUDA device, or inline on a CPU 'device'
    (the CPU form only exists so the exchange logic can be exercised with gloo in tests)."""

    def __init__(self, device, priority=0, stream=None):
        self.cuda = _is_cuda(device)
        if not self.cuda:
            self.stream = None
        else:
            self.stream = stream if stream is not None else torch.cuda.Stream(device, priority=priority)

    def __enter__(self):
        if self.cuda:
            self._ctx = torch.cuda.stream(self.stream)
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.cuda:
            self._ctx.__exit__(*a)
        return False


class DeviceDistributedPrefetcher(DeviceIterator):
    def __init__(self, devices, it: Iterator[ProtoDistributedBatch], pipeline_on=True, ops=None, group=None):
        super().__init__(devices)
        self.it = it
        self.pipeline_on = pipeline_on
        self.group = group
        cfg = self.it.session.config
        self.partition_book = cfg.partition_book
        self.cache = cfg.cache
        self.use_cache = bool(cfg.use_cache)
        # The GPU Session can run the whole exchange natively (RCCL, session.hip): its batches arrive
        # with x assembled and this iterator only double-buffers them like DevicePrefetcher.
        self.native = bool(getattr(self.it.session, "native_exchange", False))
        if self.native:
            self.rank, self.world_size = int(self.partition_book.rank), int(self.partition_book.world_size)
        else:
            self.rank = dist.get_rank(group)
            self.world_size = dist.get_world_size(group)
            assert self.world_size == int(self.partition_book.world_size), "partition book / process group mismatch"
            assert self.rank == int(self.partition_book.rank)
        self.other_ranks = [r for r in range(self.world_size) if r != self.rank]
        self.ops = ops if ops is not None else _HipFeatureOps()
        self.offsets = [int(v) for v in self.partition_book.partition_offsets.tolist()]
        self.rank_offset = self.offsets[self.rank]
        # all local rows, HBM resident (the session concatenated x_gpu and x_cpu)
        self.features = getattr(self.it.session, "_x", None)
        if self.features is None:
            self.features = cfg.x_gpu
        self.feature_dim = self.features.size(1)
        self.features_dtype = self.features.dtype
        if self.use_cache:
            self.cache_feats = self.cache.device_features() if hasattr(self.cache, "device_features") \
                else self.cache.cached_features
        else:
            self.cache_feats = None
        # the sampler's persistent delivery stream when there is one, default priority (see DevicePrefetcher)
        self.side = _SideStream(self.device, stream=getattr(self.it.session, "consumer_stream", None)
                                if _is_cuda(self.device) else None)
        # torch.distributed transport: ONE exchange (three collectives) per GROUP of batches -- the
        # batches of a sampler group become ready together, and a collective call costs ~80 us of host
        # time in c10d whatever its size
        self.G = max(1, int(getattr(self.it.session, "group_size", 1)))
        if self.side.cuda:
            self.counts_stream = torch.cuda.Stream(self.device)
            ring = 8               # > pipeline depth: a pinned buffer is reused only after its copy ran
            n_cnt = self.world_size * self.G
            self._sc_pinned = [torch.empty(n_cnt, dtype=torch.int64).pin_memory() for _ in range(ring)]
            self._rc_pinned = [torch.empty(n_cnt, dtype=torch.int64).pin_memory() for _ in range(ring)]
            self._ring_pos = 0
        self.q_counts = deque()    # groups whose counts exchange is in flight
        self.q_rows = deque()      # groups whose row exchange is in flight
        self.ready = deque()       # assembled batches of the current group
        self.next: Optional[list] = []
        self.next_event = None
        self.NUMBER_OF_SENT_BYTES = 0
        self.ALL_BATCHES = []                   # batches the caller wants recorded by collect_data
        self.ITERATION = 0
        self._exhausted = False
        # native exchange, default (group fetch, per-batch launch) mode: as in DevicePrefetcher the Session is told its
        # delivery stream once -- no stream context around every request, record_stream on the batch's arenas
        sess = self.it.session
        self._direct = bool(self.native and self.side.cuda and getattr(sess, "_member_mode", False) and
                            hasattr(sess, "set_export_stream"))
        if self._direct:
            sess.set_export_stream(self.side.stream)
        self._next_arenas = None
        if self.native:
            self._advance(produce_output=True)
        else:
            for _ in range(2 if pipeline_on else 0):
                self._advance(produce_output=False)
            self._fill_next()

    # ---- stages of the torch.distributed transport (one pass per group) ----
    def _stage_sample_and_counts(self):
        """pull the next group of sampled batches and start the exchange of their request counts (C1)"""
        protos = []
        runtime_stats_cuda.start_region("sampling2")
        while len(protos) < self.G and not self._exhausted:
            proto = next(self.it, None)
            if proto is None:
                self._exhausted = True
            else:
                protos.append(proto)
        runtime_stats_cuda.end_region("sampling2")
        if not protos:
            return
        P, r, G = self.world_size, self.rank, self.G
        cnt = [[int(t.numel()) for t in p.partition_nids] for p in protos]     # [batch][owner]
        mat = torch.zeros((P, G), dtype=torch.int64)                           # row m: what I ask of peer m, per batch
        for i, c in enumerate(cnt):
            for m in range(P):
                if m != r:
                    mat[m, i] = c[m]
        dev = protos[0].perm_partition_to_mfg.device
        if self.side.cuda:
            # The counts exchange runs on a stream of its own and lands in pinned memory behind an
            # event: the host later waits for THIS exchange only, not for the feature assembly that
            # is queued on the delivery stream.
            k = self._ring_pos
            self._ring_pos = (k + 1) % len(self._sc_pinned)
            sc_host, rc_host = self._sc_pinned[k], self._rc_pinned[k]
            sc_host.copy_(mat.view(-1))
            with torch.cuda.stream(self.counts_stream):
                sc = sc_host.to(dev, non_blocking=True)
                rc = torch.empty(P * G, dtype=torch.int64, device=dev)
                h = dist.all_to_all_single(rc, sc, group=self.group, async_op=True)      # C1 (G counts per peer)
                h.wait()
                rc_host.copy_(rc, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self.q_counts.append((protos, cnt, sc, rc, (ev, rc_host)))
        else:
            sc = mat.view(-1).clone()
            rc = torch.empty(P * G, dtype=torch.int64, device=dev)
            h = dist.all_to_all_single(rc, sc, group=self.group, async_op=True)          # C1
            self.q_counts.append((protos, cnt, sc, rc, h))

    def _stage_ids_serve_rows(self):
        """exchange the requested node ids (C2), gather the rows the peers asked for, exchange them (C3)"""
        if not self.q_counts:
            return
        protos, cnt, sc, rc, h = self.q_counts.popleft()
        P, r, G = self.world_size, self.rank, self.G
        if isinstance(h, tuple):
            ev, rc_host = h
            ev.synchronize()
            asked = rc_host.view(P, G).tolist()            # asked[m][i]: rows peer m wants from me for its batch i
        else:
            h.wait()
            asked = rc.view(P, G).tolist()
        dev = protos[0].perm_partition_to_mfg.device
        want = [sum(c[m] for c in cnt) if m != r else 0 for m in range(P)]          # ids out to / rows in from peer m
        serve = [int(sum(asked[m])) if m != r else 0 for m in range(P)]             # ids in from / rows out to peer m
        pieces = [p.partition_nids[m] for m in range(P) if m != r for p in protos]  # peer-major, then batch
        send_ids = torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.int64, device=dev)
        recv_ids = torch.empty(sum(serve), dtype=torch.int64, device=dev)
        h_ids = dist.all_to_all_single(recv_ids, send_ids, output_split_sizes=serve, input_split_sizes=want,
                                       group=self.group, async_op=True)                                   # C2
        h_ids.wait()
        send_rows = self.ops.gather_rows(self.features, recv_ids - self.rank_offset)
        recv_rows = torch.empty((sum(want), self.feature_dim), dtype=self.features_dtype, device=dev)
        h_rows = dist.all_to_all_single(recv_rows, send_rows, output_split_sizes=want, input_split_sizes=serve,
                                        group=self.group, async_op=True)                                  # C3
        self.NUMBER_OF_SENT_BYTES += sum(serve) * self.feature_dim * self.features.element_size() \
            + sum(want) * 8 + 8 * P * G
        for t in (send_ids, recv_ids, send_rows, recv_rows, sc, rc):
            if t.is_cuda and self.side.cuda:
                t.record_stream(self.side.stream)
        self.q_rows.append((protos, cnt, want, recv_rows, send_rows, h_rows))

    def _stage_assemble(self):
        """x of every batch of the oldest exchanged group, in MFG order -> self.ready"""
        if not self.q_rows:
            return
        protos, cnt, want, recv_rows, _send_rows, h_rows = self.q_rows.popleft()
        h_rows.wait()
        P, r = self.world_size, self.rank
        peer_start, acc = [], 0
        for m in range(P):                     # rows from peer m: one span, its batches in order
            peer_start.append(acc)
            acc += want[m]
        used = [0] * P
        for proto, c in zip(protos, cnt):
            seg = [0]
            for m in range(P):
                seg.append(seg[-1] + c[m])
            seg.append(seg[-1] + int(proto.cached_nids.numel()))
            recv_base = [peer_start[m] + used[m] for m in range(P)]
            for m in range(P):
                if m != r:
                    used[m] += c[m]
            n_id = getattr(proto, "n_id", None)
            if n_id is None:      # reference-shaped record (7 fields): rebuild the MFG order from the concat + perm
                ids = torch.cat(list(proto.partition_nids) +
                                ([self._cached_vertices(proto.cached_nids.device)[proto.cached_nids]]
                                 if self.use_cache else []))
                n_id = ids[proto.perm_partition_to_mfg]
            x = self.ops.assemble(n_id, proto.perm_partition_to_mfg, seg, P, r, self.rank_offset, self.features,
                                  recv_rows, self.cache_feats, proto.cached_nids, recv_base)
            y = proto.sliced_cpu_labels
            if y is not None and _is_cuda(self.device) and not y.is_cuda:
                y = y.to(self.device, non_blocking=True)
            self.ready.append(PreparedBatch(x, y, proto.adjs, proto.idx_range))

    def _cached_vertices(self, device):
        """global ids of the cache's rows on `device` (uploaded once)"""
        cv = getattr(self, "_cv_dev", None)
        if cv is None or cv.device != device:
            cv = self._cv_dev = self.cache.cached_vertices.to(device)
        return cv

    def _advance(self, produce_output: bool):
        if self.native:
            if not produce_output:
                return
            with (_NO_CTX if self._direct else self.side):
                runtime_stats_cuda.start_region("sampling2")
                proto = next(self.it, None)
                runtime_stats_cuda.end_region("sampling2")
                self.next_event = _ready_event(self.it) if proto is not None else None
                self._next_arenas = getattr(self.it.session, "last_arenas", None) if (self._direct and proto is not None) else None
                if proto is None:
                    self.next = None
                    sent, _recv = self.it.session.exchange_bytes()
                    self.NUMBER_OF_SENT_BYTES = sent
                else:
                    x = getattr(proto, "x", None)
                    if x is None:
                        # the reference's own ProtoDistributedBatch (fast_trainer/samplers.py:32-68) has no
                        # field for the assembled features: the Session kept them for this batch
                        kept = self.it.session.take_native_features(proto.idx_range)
                        if kept is None:
                            raise RuntimeError("native exchange: the batch record carries no x and the Session "
                                               f"holds none for seeds {proto.idx_range}")
                        x = kept[0]
                    self.next = [PreparedBatch(x, proto.sliced_cpu_labels, proto.adjs, proto.idx_range)]
            return
        # one pipeline step at GROUP granularity: assemble the oldest exchanged group, run the id/row
        # exchange of the next one, start the counts exchange of the one after
        with self.side:
            if produce_output:
                runtime_stats_cuda.start_region("stage_combine_features")
                self._stage_assemble()
                runtime_stats_cuda.end_region("stage_combine_features")
            self._stage_ids_serve_rows()
            self._stage_sample_and_counts()

    def _fill_next(self):
        """torch transport: the next batch to hand out, advancing the group pipeline when needed"""
        while not self.ready and (self.q_counts or self.q_rows or not self._exhausted):
            self._advance(produce_output=True)
        self.next = [self.ready.popleft()] if self.ready else None

    def __next__(self):
        ret = self.next
        self.next = []
        self.ITERATION += 1
        runtime_stats_cuda.start_region("data_transfer", runtime_stats_cuda.get_last_event())
        if self.side.cuda:
            if self.native and self.next_event is not None:      # this batch's group, not the one queued behind it
                torch.cuda.current_stream(self.device).wait_event(self.next_event)
            else:
                torch.cuda.current_stream(self.device).wait_stream(self.side.stream)
        runtime_stats_cuda.end_region("data_transfer")
        runtime_stats_cuda.start_region("sampling", runtime_stats_cuda.get_last_event())
        if not ret:
            if self.side.cuda:
                torch.cuda.synchronize()
            raise StopIteration
        if self.side.cuda:
            cur = torch.cuda.current_stream(self.device)
            if self._next_arenas:        # (of the batch handed out now; _advance below replaces them)
                for a in self._next_arenas:
                    a.record_stream(cur)
            else:
                for b in ret:
                    b.record_stream(cur)
        if self.native:
            self._advance(produce_output=True)
        else:
            check = getattr(self.ops, "check_async", None)
            if check is not None:
                check()
            self._fill_next()
        return ret

    def print_stats(self):
        aggregate_time_results.clear()          # reference :566-570: the totals are per epoch

    def collect_data(self, data_collector=None, ids=None, save_all_batch_data_to_disk=False):
        """End-of-epoch statistics hook of the reference's driver (transferers.py:843-887).  There the body is
        switched off (``save_all_batch_data_to_disk = False``) and the call returns None; the same here by default.
        With ``save_all_batch_data_to_disk=True`` the batches kept in ``ALL_BATCHES`` (the caller appends the ones it
        wants recorded) are converted with NumpyProtoDistributedBatch and written through the collector's
        ``get_epoch_data_filepath`` / ``np_savez_list``, one file per field; ``ids`` are the epoch's seed ids."""
        if save_all_batch_data_to_disk and data_collector is not None and self.ALL_BATCHES:
            columns = {k: [] for k in NumpyProtoDistributedBatch._fields}
            seeds = ids if ids is not None else self.it.session.config.idx
            for b in self.ALL_BATCHES:
                for k, v in NumpyProtoDistributedBatch.from_proto_batch(b, seeds)._asdict().items():
                    columns[k].append(v)
            for k, lst in columns.items():
                data_collector.np_savez_list(data_collector.get_epoch_data_filepath(k, use_rank=True), lst)
            self.ALL_BATCHES = []
        return None

    def quiesce(self):
        """See fast_sampler.Session.quiesce: drain what the native exchange has in flight before the
        caller issues collectives on its own process group."""
        q = getattr(self.it.session, "quiesce", None)
        if q is not None:
            q()
        if self.side.cuda:
            torch.cuda.synchronize()


class DevicePrefetcher(DeviceIterator):
    """Single-GPU double buffering (transferers.py:890-970): the next batch is sampled, sliced and
    exported on a side stream while the model consumes the current one."""

    def __init__(self, devices, it: Iterator[PreparedBatch], pipeline_on=True):
        super().__init__(devices)
        self.it = it
        # default priority on purpose: a high-priority side stream measured 1.9x SLOWER here (0.43 vs
        # 0.23 ms/batch) -- it serialises against the sampler's streams instead of overlapping
        sess_stream = getattr(getattr(it, "session", None), "consumer_stream", None) if len(devices) == 1 else None
        # the GPU sampler offers a persistent delivery stream of its own (distinct hardware queue
        # from the sampling streams); otherwise a plain side stream as in the reference
        self.streams = [sess_stream] if sess_stream is not None else [torch.cuda.Stream(device) for device in devices]
        # GPU Session in its default (group fetch, per-batch launch) mode: it is told the delivery stream once and a request
        # needs no stream context; the batch's tensors are views of <= 3 arenas, which is all record_stream has to see
        sess = getattr(it, "session", None)
        self._direct = bool(sess_stream is not None and getattr(sess, "_member_mode", False) and
                            hasattr(sess, "set_export_stream"))
        if self._direct:
            sess.set_export_stream(sess_stream)
        self.next = []
        self.next_events = []
        self.sampling_times = []
        self.preload(False)

    def preload(self, timing=True):
        self.next = []
        self.next_events = []
        if self._direct:
            t0 = time.perf_counter_ns()
            batch = next(self.it, None)
            if batch is not None:
                self.next.append(batch)
                self.next_events.append(None)
            self.sampling_times.append(time.perf_counter_ns() - t0)
            return
        for device, stream in zip(self.devices, self.streams):
            t0 = time.perf_counter_ns()
            with torch.cuda.stream(stream):
                batch = next(self.it, None)
                if batch is None:
                    break
                # batches of the GPU sampler are already in HBM: `.to` would only rebuild the records
                on_dev = batch.x is not None and batch.x.is_cuda and batch.x.device == torch.device(device)
                self.next.append(batch if on_dev else batch.to(device, non_blocking=True))
                # The GPU sampler delivers a whole group of batches with one launch and names the event after
                # which THIS batch is complete: waiting for that event instead of the whole side stream keeps
                # the model from waiting for the delivery of the group after it (already queued on the stream).
                self.next_events.append(_ready_event(self.it) if on_dev else None)
            self.sampling_times.append(time.perf_counter_ns() - t0)

    def __next__(self):
        runtime_stats_cuda.start_region("data_transfer", runtime_stats_cuda.get_last_event())
        cur_streams = [torch.cuda.current_stream(device) for device in self.devices]
        for k, (cur_stream, stream) in enumerate(zip(cur_streams, self.streams)):
            ev = self.next_events[k] if k < len(self.next_events) else None
            if ev is not None:
                cur_stream.wait_event(ev)
            else:
                cur_stream.wait_stream(stream)
        runtime_stats_cuda.end_region("data_transfer")
        runtime_stats_cuda.start_region("sampling", runtime_stats_cuda.get_last_event())
        ret = self.next
        if not ret:
            torch.cuda.synchronize()
            raise StopIteration
        arenas = getattr(self.it.session, "last_arenas", None) if self._direct else None
        if arenas:            # (of the batch handed out now: preload() below moves the Session on to the next one)
            for a in arenas:
                a.record_stream(cur_streams[0])
        else:
            for cur_stream, batch in zip(cur_streams, ret):
                batch.record_stream(cur_stream)
        self.preload()
        return ret


class DeviceTransferer(DeviceIterator):
    def __init__(self, devices, it: Iterator[PreparedBatch], pipeline_on=True):
        super().__init__(devices)
        self.it = it

    def __next__(self):
        ret = [batch.to(device, non_blocking=True) for device, batch in zip(self.devices, self.it)]
        if len(ret) == 0:
            raise StopIteration
        return ret


class DeviceSlicerTransferer(DeviceIterator):
    def __init__(self, devices, x: torch.Tensor, y: torch.Tensor, it: Iterator[ProtoBatch]):
        super().__init__(devices)
        self.x = x
        self.y = y
        self.it = it

    def __next__(self):
        ret = [PreparedBatch.from_proto_batch(self.x, self.y, pb).to(device, non_blocking=True)
               for device, pb in zip(self.devices, self.it)]
        if len(ret) == 0:
            raise StopIteration
        return ret
