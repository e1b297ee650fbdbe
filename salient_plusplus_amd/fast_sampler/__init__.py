"""Drop-in replacement for SALIENT++'s native ``fast_sampler`` module, backed by the MI355X data
path (libspp_hip.so, C ABI in include/spp.h).

It exports exactly the surface of the reference pybind module
(fast_sampler/fast_sampler.cpp:1280-1396): ``Config``, ``Session``, ``ProtoDistributedBatch``,
``RangePartitionBook``, ``Cache``, ``sample_adj``, ``multilayer_sample``, ``full_sample``,
``to_row_major``, ``serial_index`` -- same names, argument meaning and error behaviour
(``RuntimeError`` where the reference has ``TORCH_CHECK``; end of data is ``None``).

Differences that follow from the MI355X design (documented in DESIGN.md):
  * sampling, MFG construction, slicing and ownership bucketing run on the GPU; every tensor a
    batch carries lives in HBM (so a later ``.to(device)`` in the transferers is a no-op);
  * the whole feature matrix of the rank is HBM resident (288 GB): ``x_cpu`` is uploaded once,
    ``sliced_cpu_features`` is always empty and ``async_slice_tensors`` degenerates;
  * ``num_threads`` selects HIP streams instead of CPU worker threads, and
    ``max_items_in_queue`` bounds the batch slots in flight.
There is no CPU fallback: without the HIP extension and a GPU every entry point raises.
"""
import collections
import ctypes as C
import datetime
import math
import os
import threading
from typing import List, Optional, Sequence

import torch

from .. import _native as nat

__all__ = ["Config", "Session", "ProtoDistributedBatch", "RangePartitionBook", "Cache", "sample_adj",
           "multilayer_sample", "full_sample", "to_row_major", "serial_index", "NativeComm", "native_comm",
           "set_native_comm", "async_errors", "set_sampler_options", "sampler_options", "sampler_info", "TableRows", "RowRefs",
           "P2PPeers", "set_p2p_peers", "p2p_open_peers"]

# four slot-sets of 16 batches (~75 MB of workspace per slot at fanout [15,10,5], batch 1024: 4.8 GB).  With two sets, a
# set's next sampling chain could only start once its previous group was consumed and the consumer waited on chain
# latency at every group boundary (0.154 -> 0.144 ms per batch on S-papers); 16 batches per launch instead of 8
# amortise the latency-bound small hops (0.1416 -> 0.1344 in 20-step windows, 0.1311 -> 0.1302 in 192-step windows)
_MAX_SLOTS = int(os.environ.get("SPP_MAX_SLOTS", "64"))
# distributed Sessions pipeline three stages (sample -> exchange -> consume), one slot-set each plus
# one in hand: 4 sets of 16
_MAX_SLOTS_DIST = int(os.environ.get("SPP_MAX_SLOTS_DIST", "64"))
# group delivery: groups kept delivered ahead of the one being handed out (their slot-sets go back to the sampler
# that much earlier; each holds ~2.2 GB of outputs at papers scale)
_LOOKAHEAD_GROUPS = max(1, int(os.environ.get("SPP_LOOKAHEAD_GROUPS", "2")))


# --------------------------------------------------------------------------------------------
# helpers
# --------------------------------------------------------------------------------------------
def _lib():
    L = nat.load()
    nat.require_device()
    return L


def _device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device())


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None


def _stream_ptr(stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream)


class _ResidentCache:
    """HBM residency of the (large, epoch-invariant) host tensors a Config carries.

    The reference shares the CPU tensors of a Config between Sessions; here the graph and the
    features are uploaded once and stay in HBM for every later Session (keyed by storage identity)."""

    def __init__(self):
        self._d = {}
        self._lock = threading.Lock()

    def get(self, t: torch.Tensor, dtype=None) -> torch.Tensor:
        dev = _device()
        if t.is_cuda:
            r = t if t.device == dev else t.to(dev)
            r = r.contiguous()
            return r if dtype is None or r.dtype == dtype else r.to(dtype)
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, dtype, dev.index)
        with self._lock:
            hit = self._d.get(key)
            if hit is not None and hit[0] is t:
                return hit[1]
            src = t.contiguous()
            if dtype is not None and src.dtype != dtype:
                src = src.to(dtype)
            r = src.to(dev)
            self._d[key] = (t, r)          # keep `t` alive so the key cannot be recycled
            return r

    def get_rows(self, t: torch.Tensor) -> torch.Tensor:
        """HBM-resident copy of a 2-D feature table with rows padded to the HBM fetch granule
        (a strided [N, F] view of [N, stride] storage; see _row_stride_elems)."""
        dev = _device()
        se = _row_stride_elems(t.size(1), t.element_size()) if t.dim() == 2 else 0
        if t.dim() != 2 or se == t.size(1):
            return self.get(t)
        key = ("rows", t.data_ptr(), tuple(t.shape), tuple(t.stride()), t.dtype, dev.index)
        with self._lock:
            hit = self._d.get(key)
            if hit is not None and hit[0] is t:
                return hit[1]
            buf = torch.empty((t.size(0), se), dtype=t.dtype, device=dev)
            r = buf[:, :t.size(1)]
            r.copy_(t, non_blocking=False)
            self._d[key] = (t, r)
            return r

    def clear(self):
        with self._lock:
            self._d.clear()


def _row_stride_elems(cols: int, elem_bytes: int) -> int:
    """Elements between consecutive rows of a resident feature table.  HBM is fetched in 128-B
    granules: a dense 200-B row straddles 2.56 of them on average, a row that starts on a granule
    boundary exactly 2 -- 22 % less gather traffic for 28 % more table (288 GB of HBM per GPU).
    Rows shorter than a granule are padded to the next power of two so that none straddles one."""
    rb = cols * elem_bytes
    if rb < 16 or os.environ.get("SPP_ROW_PAD", "1") == "0":
        return cols
    if rb >= 128:
        stride = (rb + 127) // 128 * 128
        if stride * 3 > rb * 4:                  # more than 1/3 padding: settle for 64-B alignment, or none
            stride = (rb + 63) // 64 * 64
            if stride * 3 > rb * 4:
                return cols
    else:
        stride = 16
        while stride < rb:
            stride *= 2
    if stride % elem_bytes:
        return cols
    return stride // elem_bytes


def _stride_bytes(t: Optional[torch.Tensor]) -> int:
    return int(t.stride(0)) * t.element_size() if t is not None and t.dim() == 2 and t.size(0) > 1 else 0


_resident = _ResidentCache()
_groups_opened = 0          # sampling groups fetched by all Sessions of this process (measurement aid: groups_opened())


def groups_opened() -> int:
    """Sampling groups (up to 16 batches: one chain, one exchange) the Sessions of this process have fetched so far."""
    return _groups_opened


_graph_epoch = 0


def clear_resident_cache():
    """Drop every HBM copy made for host tensors (graph, features) and every pooled sampler."""
    global _graph_epoch
    _resident.clear()
    _SamplerPool.clear()
    for peers in _p2p_auto.values():   # mappings of the peers' partitions (P2P transport): the tables they name may go away
        peers.close()
    _p2p_auto.clear()
    _graph_epoch += 1          # graphs uploaded from now on never share derived tables with earlier ones


def _graph_generation(rowptr_d: torch.Tensor, col_d: torch.Tensor) -> int:
    """spp_sampler_cfg.graph_generation: changes when the resident cache was cleared (another graph may sit at the
    same address) and when torch saw an in-place write to the graph tensors (their version counters)."""
    return (_graph_epoch << 32) + ((int(rowptr_d._version) + int(col_d._version)) & 0xffffffff)


def _as_i64_list(sizes: Sequence[int]) -> List[int]:
    out = [int(s) for s in sizes]
    if not 1 <= len(out) <= nat.SPP_MAX_HOPS:
        raise RuntimeError(f"sizes must have between 1 and {nat.SPP_MAX_HOPS} entries, got {len(out)}")
    return out


# --------------------------------------------------------------------------------------------
# the batch's features as (resident table, node ids) -- opt-in, for a consumer that aggregates straight from the table
# --------------------------------------------------------------------------------------------
class TableRows:
    """``x = table[n_id]`` without the copy: what a Session with ``table_features`` puts in the place of the batch's
    feature matrix (fast_sampler.cpp:1004-1016 ``serial_index(x_cpu, n_id)``).  ``models.SAGE`` aggregates its first
    layer straight from ``table`` (spp_sage_operand_forward_table: same rows, same summation order, bit-identical
    operand), so the 242 MB of rows a papers-scale batch holds are neither written by the delivery nor read back.
    Everything else still sees a feature matrix through ``materialize()``; sizes / device / dtype read like the
    tensor's."""
    __slots__ = ("table", "n_id")

    def __init__(self, table: torch.Tensor, n_id: torch.Tensor):
        self.table, self.n_id = table, n_id

    # -- what iterators and records ask of a feature matrix --
    @property
    def is_cuda(self):
        return self.n_id.is_cuda

    @property
    def device(self):
        return self.n_id.device

    @property
    def dtype(self):
        return self.table.dtype

    @property
    def shape(self):
        return torch.Size((self.n_id.numel(), self.table.size(1)))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def numel(self):
        return self.n_id.numel() * self.table.size(1)

    def record_stream(self, stream):
        if self.n_id.is_cuda:
            self.n_id.record_stream(stream)              # (the table lives as long as the graph does)

    def to(self, device=None, non_blocking=False, **_kw):
        if device is None or torch.device(device) == self.n_id.device:
            return self
        return self.materialize().to(device=device, non_blocking=non_blocking)

    def materialize(self) -> torch.Tensor:
        """The feature matrix itself: the HIP row gather on the caller's current stream (a5, serial_index)."""
        t = self.table                                   # (a strided view of the padded resident table)
        out = torch.empty((self.n_id.numel(), t.size(1)), dtype=t.dtype, device=t.device)
        if out.numel():
            nat.check(_lib().spp_gather_rows_strided(_ptr(t), t.size(0), t.size(1) * t.element_size(), _stride_bytes(t),
                                                     _ptr(self.n_id), 8, self.n_id.numel(), self.n_id.numel(), _ptr(out),
                                                     _stream_ptr()))
        return out


class RowRefs:
    """``x = cat(features_gather + [cached])[perm]`` (transferers.py:472-486) without the copy: what a PARTITIONED Session
    with ``row_refs`` puts in the place of the batch's feature matrix.  ``addr[j]`` is the device address of the feature
    row of MFG node j -- in this rank's resident partition, in the VIP cache, in a peer's partition (P2P transport) or in
    ``x_remote`` (the rows received for this batch over RCCL: the only rows the delivery still copies).  ``models.SAGE``
    aggregates its first layer straight from these addresses (spp_sage_operand_forward_rows: same rows, same summation
    order as over the assembled matrix -- bit-identical operand); everything else sees a feature matrix through
    ``materialize()``.  ``keep`` holds the tensors the addresses point into."""
    __slots__ = ("addr", "n_id", "width", "_dtype", "x_remote", "keep")

    def __init__(self, addr: torch.Tensor, n_id, width: int, dtype, x_remote, keep):
        self.addr, self.n_id, self.width, self._dtype, self.x_remote, self.keep = addr, n_id, int(width), dtype, x_remote, keep

    @property
    def is_cuda(self):
        return self.addr.is_cuda

    @property
    def device(self):
        return self.addr.device

    @property
    def dtype(self):
        return self._dtype

    @property
    def shape(self):
        return torch.Size((self.addr.numel(), self.width))

    def size(self, dim=None):
        return self.shape if dim is None else self.shape[dim]

    def dim(self):
        return 2

    def numel(self):
        return self.addr.numel() * self.width

    def record_stream(self, stream):
        for t in (self.addr, self.n_id, self.x_remote):
            if t is not None and t.is_cuda:
                t.record_stream(stream)                  # (the resident tables live as long as the Sessions' graph does)

    def to(self, device=None, non_blocking=False, **_kw):
        if device is None or torch.device(device) == self.addr.device:
            return self
        return self.materialize().to(device=device, non_blocking=non_blocking)

    def materialize(self) -> torch.Tensor:
        """The feature matrix itself (on the caller's current stream)."""
        n = self.addr.numel()
        out = torch.empty((n, self.width), dtype=self._dtype, device=self.addr.device)
        if out.numel():
            nat.check(_lib().spp_gather_row_refs(_ptr(self.addr), n, self.width * out.element_size(), _ptr(out), _stream_ptr()))
        return out


# --------------------------------------------------------------------------------------------
# P2P transport: the peers' partitions mapped into this process (include/spp.h spp_exchange_cfg.peer_x_dev)
# --------------------------------------------------------------------------------------------
class P2PPeers:
    """Base address, in THIS process, of every rank's resident partition (``ptrs[m]``; rows ``stride`` bytes apart)."""

    def __init__(self, ptrs, stride, opened=(), keep=None):
        self.ptrs, self.stride, self._opened, self.keep = [int(v) for v in ptrs], int(stride), list(opened), keep

    def close(self):
        L = nat.load()
        for base in self._opened:
            L.spp_ipc_close(C.c_void_p(base))
        self._opened = []


_p2p_tls = threading.local()
_p2p_auto = {}


def set_p2p_peers(peers):
    """Pin the peer tables the P2P Sessions created by THIS thread use: a P2PPeers, a list of the ranks' resident
    partitions (in-process ranks: plain device tensors, rows padded alike), or None = automatic (p2p_open_peers)."""
    if peers is not None and not isinstance(peers, P2PPeers):
        tabs = list(peers)
        peers = P2PPeers([t.data_ptr() if t is not None and t.numel() else 0 for t in tabs],
                         _common_stride([_table_stride_bytes(t) for t in tabs if t is not None and t.numel()]), keep=tabs)
    _p2p_tls.peers = peers


def _table_stride_bytes(t: torch.Tensor) -> int:
    """Bytes between the rows of a resident partition.  A table of ONE row has no stride of its own: it follows the rule
    every resident table is laid out by (_row_stride_elems), which is what its peers' tables use."""
    return _stride_bytes(t) or _row_stride_elems(t.size(1), t.element_size()) * t.element_size()


def _common_stride(strides) -> int:
    if len(set(strides)) != 1:
        raise RuntimeError(f"P2P transport: the ranks' partitions must share one row stride, got {sorted(set(strides))}")
    return int(strides[0])


def p2p_open_peers(x_local: torch.Tensor, group=None) -> P2PPeers:
    """COLLECTIVE over `group` (any torch.distributed backend): every rank exports its resident partition `x_local`
    (hipIpcGetMemHandle of the allocation that holds it), the handles travel by all_gather_object, and every rank maps
    its peers' partitions (hipIpcOpenMemHandle).  The ranks must run on one node; peer access between their GPUs is
    enabled lazily by the runtime.  Close the result only after a barrier (a peer may still be reading)."""
    import torch.distributed as dist
    L = _lib()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    handle = (C.c_ubyte * nat.SPP_IPC_HANDLE_BYTES)()
    off = C.c_int64(0)
    mine = None
    if x_local is not None and x_local.numel():
        nat.check(L.spp_ipc_export(C.c_void_p(x_local.data_ptr()), handle, C.byref(off)))
        mine = (bytes(handle), int(off.value), os.getpid(), int(x_local.data_ptr()), _table_stride_bytes(x_local))
    got = [None] * world
    dist.all_gather_object(got, mine, group=group)
    ptrs, opened = [], []
    stride = _common_stride([rec[4] for rec in got if rec is not None])
    for m, rec in enumerate(got):
        if rec is None:
            ptrs.append(0)
            continue
        if m == rank or rec[2] == os.getpid():           # my own table / a rank of this very process
            ptrs.append(rec[3])
            continue
        buf = (C.c_ubyte * nat.SPP_IPC_HANDLE_BYTES).from_buffer_copy(rec[0])
        base = C.c_void_p()
        nat.check(L.spp_ipc_open(buf, _device().index, C.byref(base)))
        opened.append(int(base.value))
        ptrs.append(int(base.value) + rec[1])
    return P2PPeers(ptrs, stride, opened, keep=x_local)


# --------------------------------------------------------------------------------------------
# RangePartitionBook / Cache   (fast_sampler/range_partition_book.{hpp,cpp})
# --------------------------------------------------------------------------------------------
class RangePartitionBook:
    """``RangePartitionBook(rank, world_size, partition_offsets)`` (range_partition_book.hpp:31-57)."""

    def __init__(self, rank: int = 0, world_size: int = 1, partition_offsets: Optional[torch.Tensor] = None):
        self.rank = int(rank)
        self.world_size = int(world_size)
        self.partition_offsets = partition_offsets if partition_offsets is not None \
            else torch.zeros(0, dtype=torch.int64)

    def _offsets_host(self):
        offs = self.partition_offsets.detach().to("cpu", torch.int64).contiguous()
        if offs.numel() < 2 or offs.numel() > nat.SPP_MAX_PARTS + 1:
            raise RuntimeError(f"partition_offsets must hold 2..{nat.SPP_MAX_PARTS + 1} entries")
        return offs

    def nid2localnid(self, nids: torch.Tensor, partition_idx: int) -> torch.Tensor:
        # nids - partition_offsets[partition_idx]   (range_partition_book.cpp:95-96)
        return nids - self.partition_offsets[partition_idx].to(nids.device)

    def nid2partid(self, nids: torch.Tensor) -> torch.Tensor:
        # searchsorted(partition_offsets, nids, right=True) - 1   (range_partition_book.cpp:98-100)
        # one implementation: host tensors are looked up on the GPU too and come back as host tensors
        offs = self._offsets_host()
        L = _lib()
        src = nids.to(_device(), torch.int64).contiguous()
        out = torch.empty_like(src)
        nat.check(L.spp_nid2partid(C.c_void_p(offs.data_ptr()), offs.numel(), _ptr(src), src.numel(),
                                   _ptr(out), _stream_ptr()))
        return out if nids.is_cuda else out.to(nids.device)

    def nid_is_local(self, nids: torch.Tensor) -> torch.Tensor:
        lo = self.partition_offsets[self.rank].to(nids.device)
        hi = self.partition_offsets[self.rank + 1].to(nids.device)
        return (nids >= lo) * (nids < hi)       # range_partition_book.cpp:105-107

    def partid2nids(self, partition_idx: int) -> torch.Tensor:
        return torch.arange(int(self.partition_offsets[partition_idx]),
                            int(self.partition_offsets[partition_idx + 1]), dtype=torch.int64)


class Cache:
    """``Cache()`` / ``Cache(rank, world_size, cached_vertices, cached_features)``
    (range_partition_book.hpp:60-91).  The reference's two dense 2e8-entry host tables become one
    device-resident direct map ``int32[max_id+1]`` (-1 = not cached), built by spp_cache_build_map."""

    def __init__(self, rank: int = 0, world_size: int = 0, cached_vertices: Optional[torch.Tensor] = None,
                 cached_features: Optional[torch.Tensor] = None):
        self.rank = int(rank)
        self.world_size = int(world_size)
        # default ctor: empty tensors, not None (range_partition_book.cpp:116-119)
        self.cached_vertices = cached_vertices if cached_vertices is not None \
            else torch.empty(0, dtype=torch.int64)
        self.cached_features = cached_features if cached_features is not None \
            else torch.empty((0, 0), dtype=torch.float16)
        self._map = None
        self._vertices_dev = None
        self._features_dev = None

    # -- device side state, built lazily (needs a GPU) --
    def device_map(self) -> torch.Tensor:
        if self._map is None:
            L = _lib()
            cv = self.cached_vertices.to(_device(), torch.int64).contiguous()
            n = int(cv.max().item()) + 1 if cv.numel() > 0 else 1
            m = torch.empty(n, dtype=torch.int32, device=_device())
            nat.check(L.spp_cache_build_map(_ptr(cv), cv.numel(), _ptr(m), n, _stream_ptr()))
            self._vertices_dev = cv
            self._map = m
        return self._map

    def device_features(self) -> torch.Tensor:
        if self._features_dev is None:
            self._features_dev = _resident.get_rows(self.cached_features) if self.cached_features.dim() == 2 \
                else self.cached_features.to(_device()).contiguous()
        return self._features_dev

    def _lookup(self, nids: torch.Tensor, want_flag: bool):
        L = _lib()
        m = self.device_map()
        src = nids.to(_device(), torch.int64).contiguous()
        flag = torch.empty(src.numel(), dtype=torch.bool, device=src.device) if want_flag else None
        cid = None if want_flag else torch.empty(src.numel(), dtype=torch.int64, device=src.device)
        nat.check(L.spp_cache_lookup(_ptr(m), m.numel(), _ptr(src), src.numel(), _ptr(flag), _ptr(cid),
                                     _stream_ptr()))
        out = flag if want_flag else cid
        return out if nids.is_cuda else out.cpu()

    def nid_is_cached(self, nids: torch.Tensor) -> torch.Tensor:
        return self._lookup(nids, True)        # range_partition_book.cpp:161-183

    def nid2cachenid(self, nids: torch.Tensor) -> torch.Tensor:
        return self._lookup(nids, False)       # range_partition_book.cpp:185-195


# --------------------------------------------------------------------------------------------
# Config / ProtoDistributedBatch   (fast_sampler.cpp:515-531, :180-188)
# --------------------------------------------------------------------------------------------
class Config:
    """Default-constructible record with the read/write fields of fast_sampler.cpp:1290-1309."""

    def __init__(self):
        self.x_cpu = torch.empty(0)
        self.x_gpu = torch.empty(0)
        self.y = None
        self.rowptr = torch.empty(0, dtype=torch.int64)
        self.col = torch.empty(0, dtype=torch.int64)
        self.idx = torch.empty(0, dtype=torch.int64)
        self.batch_size = 0
        self.sizes = []
        self.skip_nonfull_batch = False
        self.pin_memory = False
        self.distributed = False
        self.partition_book = RangePartitionBook()
        self.cache = Cache()
        self.force_exact_num_batches = False
        self.exact_num_batches = 0
        self.count_remote_frequency = False
        self.use_cache = False


class ProtoDistributedBatch:
    """Batch record of the distributed worker branch (fast_sampler.cpp:180-188)."""

    def __init__(self):
        self.partition_nids = []
        self.sliced_cpu_features = None
        self.sliced_cpu_labels = None
        self.cached_nids = None
        self.perm_partition_to_mfg = None
        self.adjs = []
        self.idx_range = (0, 0)
        # extras of the GPU path (ignored by reference-style consumers)
        self.n_id = None
        self.x = None              # native exchange: the batch's features, already assembled in MFG order


# --------------------------------------------------------------------------------------------
# native RCCL communicator of the feature exchange (spp_comm, include/spp.h e1-e3)
# --------------------------------------------------------------------------------------------
class NativeComm:
    """Owns one spp_comm handle."""

    def __init__(self, handle, rank, world):
        self.handle, self.rank, self.world = handle, int(rank), int(world)

    def close(self):
        if self.handle is not None:
            nat.load().spp_comm_destroy(self.handle)
            self.handle = None

    @staticmethod
    def local(world: int) -> List["NativeComm"]:
        """`world` in-process ranks that copy device-to-device (single-GPU testing of the exchange
        logic; drive each rank from its own thread)."""
        L = _lib()
        arr = (C.c_void_p * world)()
        nat.check(L.spp_comm_create_local(world, _device().index, arr))
        return [NativeComm(C.c_void_p(arr[m]), m, world) for m in range(world)]


_trimmed_for_exchange = False
_comm_tls = threading.local()
_comm_auto = {}
_comm_lock = threading.Lock()


def set_native_comm(comm: Optional[NativeComm]):
    """Pin the communicator distributed Sessions created by THIS thread use (None = automatic)."""
    _comm_tls.comm = comm


def native_comm(group=None) -> Optional[NativeComm]:
    """The communicator of the native feature exchange, or None when the exchange has to go through
    torch.distributed (SPP_DIST_TRANSPORT=torch, no NCCL process group, ...).

    The first call is COLLECTIVE over `group`: rank 0's RCCL id is broadcast with torch.distributed
    and every rank joins with ncclCommInitRank."""
    pinned = getattr(_comm_tls, "comm", None)
    if pinned is not None:
        return pinned
    if os.environ.get("SPP_DIST_TRANSPORT", "rccl").lower() == "torch":
        return None
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend(group) != "nccl":
        return None
    key = id(group) if group is not None else 0
    with _comm_lock:
        if key in _comm_auto:
            return _comm_auto[key]
        L = _lib()
        dev = _device()
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        buf = torch.zeros(nat.SPP_COMM_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            nat.check(L.spp_comm_unique_id(C.c_void_p(buf.data_ptr())))
        buf = buf.to(dev)
        dist.broadcast(buf, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        token = buf.cpu().contiguous()
        h = C.c_void_p()
        nat.check(L.spp_comm_create(C.c_void_p(token.data_ptr()), rank, world, dev.index, C.byref(h)))
        comm = NativeComm(h, rank, world)
        _comm_auto[key] = comm
        return comm


# --------------------------------------------------------------------------------------------
# sampler pool: workspace outlives Sessions like the reference's global thread pool
# (fast_sampler.cpp:512-513)
# --------------------------------------------------------------------------------------------
_OPT_FIELDS = ("col32", "deg_tags", "row_stubs", "rng_arena", "rng_arena_mb", "fuse_scatter", "flag_tiled",
               "rows_coalesced", "dedup_preread", "fuse_max_edges", "initial_edge_cap")
_sampler_opts = {}


def set_sampler_options(**kw):
    """Pin chain variants (include/spp.h spp_sampler_opts) for the samplers created from now on -- Sessions and the free
    functions alike; pooled samplers are keyed by them.  Switches: 0 = automatic, > 0 on, < 0 off (True / False are
    accepted); no arguments = everything automatic again.  Every variant produces the same batches."""
    for k in kw:
        if k not in _OPT_FIELDS:
            raise RuntimeError(f"unknown sampler option {k!r} (known: {', '.join(_OPT_FIELDS)})")
    _sampler_opts.clear()
    for k, v in kw.items():
        if isinstance(v, bool):
            v = 1 if v else -1
        if int(v) != 0:
            _sampler_opts[k] = int(v)


class sampler_options:
    """``with sampler_options(row_stubs=False): ...`` -- set_sampler_options for the block."""

    def __init__(self, **kw):
        self._kw = kw

    def __enter__(self):
        self._saved = dict(_sampler_opts)
        set_sampler_options(**self._kw)
        return self

    def __exit__(self, *exc):
        _sampler_opts.clear()
        _sampler_opts.update(self._saved)
        return False


def sampler_info(handle) -> dict:
    """spp_sampler_get_info as a dict (per-hop lists in processing order, cut to the sampler's hops)."""
    info = nat.SamplerInfo()
    nat.check(_lib().spp_sampler_get_info(handle, C.byref(info)))
    H = int(info.num_hops)
    out = {}
    for name, _t in nat.SamplerInfo._fields_:
        v = getattr(info, name)
        out[name] = [int(v[h]) for h in range(H)] if hasattr(v, "__len__") else (float(v) if isinstance(v, float) else int(v))
    return out


class _SamplerPool:
    _pool = {}
    _lock = threading.Lock()

    @classmethod
    def acquire(cls, rowptr_d, col_d, sizes, max_batch, slots, device, replace=False, part=None):
        """`part`: None or (PartitionCfg, hashable key, tensors to keep alive) -- ownership bucketing
        fused into the sampling chain (distributed Sessions)."""
        gen = _graph_generation(rowptr_d, col_d)
        opts = tuple(sorted(_sampler_opts.items()))
        key = (rowptr_d.data_ptr(), col_d.data_ptr(), gen, tuple(sizes), device, bool(replace),
               part[1] if part is not None else None, opts)
        with cls._lock:
            lst = cls._pool.setdefault(key, [])
            for i, (h, mb, ns, keep, _k) in enumerate(lst):
                if mb >= max_batch and ns >= slots:
                    return lst.pop(i)
        L = _lib()
        cfg = nat.SamplerCfg()
        cfg.rowptr_dev, cfg.col_dev = rowptr_d.data_ptr(), col_d.data_ptr()
        cfg.num_nodes, cfg.nnz = rowptr_d.numel() - 1, col_d.numel()
        cfg.num_hops = len(sizes)
        for i, s in enumerate(sizes):
            cfg.sizes[i] = s
        cfg.max_batch, cfg.num_slots, cfg.device = max_batch, slots, device
        cfg.replace = int(bool(replace))
        cfg.graph_generation = gen
        for k, v in opts:
            setattr(cfg.opts, k, v)
        if part is not None:
            cfg.part = part[0]
        h = C.c_void_p()
        rc = L.spp_sampler_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            # the workspace is allocated with hipMalloc: give back what torch's caching allocator
            # hoards (e.g. temporaries of a dataset build) and try once more
            torch.cuda.empty_cache()
            rc = L.spp_sampler_create(C.byref(cfg), C.byref(h))
        nat.check(rc)
        return (h, max_batch, slots, (rowptr_d, col_d, part[2] if part is not None else None), key)

    @classmethod
    def release(cls, entry):
        with cls._lock:
            cls._pool.setdefault(entry[4], []).append(entry)

    @classmethod
    def clear(cls):
        with cls._lock:
            L = nat.load()
            for lst in cls._pool.values():
                for (h, _mb, _ns, _keep, _key) in lst:
                    L.spp_sampler_destroy(h)
            cls._pool.clear()


def _coarse(n: int) -> int:
    """n rounded up to a multiple of 2^(floor(log2 n) - 5): at most 3 % more, 32 sizes per octave"""
    if n <= 64:
        return max(n, 1)
    g = 1 << (n.bit_length() - 6)
    return (n + g - 1) // g * g


def _host_ranges(n, batch_size, skip_nonfull, force_exact, exact_k):
    """fast_sampler.cpp:587-627 (only used to size the pooled sampler before the native session exists)."""
    if force_exact:
        if exact_k <= 0:
            return 0, 1
        avg = n // exact_k - 1
        rem = n - avg * exact_k
        return exact_k, max(1, avg + (rem + exact_k - 1) // exact_k)
    nb, r = divmod(n, batch_size)
    if r and not skip_nonfull:
        nb += 1
    return nb, max(1, min(batch_size, n))


# --------------------------------------------------------------------------------------------
# Session   (fast_sampler.cpp:533-936 + worker :963-1274)
# --------------------------------------------------------------------------------------------
class Session:
    def __init__(self, num_threads: int, max_items_in_queue: int, config: Config):
        L = _lib()
        if max_items_in_queue <= 0:
            raise RuntimeError(f"max_items_in_queue ({max_items_in_queue}) must be positive")
        self.config = config
        self._L = L
        self._h = None
        self._final = {"total": 0, "consumed": 0, "group": 0, "blocked_us": 0, "blocked_n": 0, "xbytes": (0, 0)}
        self._dev = _device()
        self._sizes = _as_i64_list(config.sizes)
        self._rowptr = _resident.get(config.rowptr, torch.int64)
        self._col = _resident.get(config.col, torch.int64)
        self._idx = config.idx.to(self._dev, torch.int64).contiguous()
        self._distributed = bool(config.distributed)

        # features / labels resident in HBM
        if self._distributed:
            xg, xc = config.x_gpu, config.x_cpu
            parts = [t for t in (xg, xc) if t is not None and t.dim() == 2 and t.size(0) > 0]
            if len(parts) == 2:
                self._x = _resident_concat(xg, xc)
            elif len(parts) == 1:
                self._x = _resident.get_rows(parts[0])
            else:
                # a rank that owns no rows still takes part in the exchange: keep the width and dtype
                wide = [t for t in (xg, xc) if t is not None and t.dim() == 2 and t.size(1) > 0]
                self._x = torch.empty((0, wide[0].size(1)), dtype=wide[0].dtype, device=self._dev) if wide else None
        else:
            self._x = _resident.get_rows(config.x_cpu) \
                if config.x_cpu is not None and config.x_cpu.dim() == 2 and config.x_cpu.numel() > 0 else None
        if self._x is not None and not (self._x.dim() == 2 and self._x.stride(-1) == 1):
            raise RuntimeError("input must be 2D row-major tensor")
        y = config.y
        self._y = _resident.get(y) if y is not None and y.numel() > 0 else None
        if self._y is not None and self._y.dim() == 1:
            self._y = self._y.unsqueeze(-1)

        n = self._idx.numel()
        force_exact = bool(config.force_exact_num_batches)
        if force_exact and (config.exact_num_batches <= 0 or n // max(1, config.exact_num_batches) < 1):
            raise RuntimeError("force_exact_num_batches needs idx.numel() / exact_num_batches >= 1")
        nb, max_batch = _host_ranges(n, int(config.batch_size), bool(config.skip_nonfull_batch), force_exact,
                                     int(config.exact_num_batches))
        slots = max(1, min(int(max_items_in_queue), _MAX_SLOTS_DIST if self._distributed else _MAX_SLOTS, max(nb, 1)))
        self._part = self._partition_cfg() if self._distributed else None
        self._pool_entry = _SamplerPool.acquire(self._rowptr, self._col, self._sizes, max_batch, slots,
                                                self._dev.index, part=self._part)

        cfg = nat.SessionCfg()
        cfg.rowptr_dev, cfg.col_dev = self._rowptr.data_ptr(), self._col.data_ptr()
        cfg.num_nodes, cfg.nnz = self._rowptr.numel() - 1, self._col.numel()
        cfg.idx_dev, cfg.n_idx = (self._idx.data_ptr() if n > 0 else None), n
        cfg.batch_size = int(config.batch_size)
        cfg.num_hops = len(self._sizes)
        for i, s in enumerate(self._sizes):
            cfg.sizes[i] = s
        cfg.skip_nonfull_batch = int(bool(config.skip_nonfull_batch))
        cfg.force_exact_num_batches = int(force_exact)
        cfg.exact_num_batches = int(config.exact_num_batches)
        cfg.max_items_in_queue = slots
        cfg.group_size = int(os.environ.get("SPP_GROUP_SIZE", "0"))   # 0 = auto
        cfg.device = self._dev.index
        # idx / x / y / the cache map may still be outputs of work queued on the caller's stream (a
        # shuffle kernel writing this epoch's seeds, Cache.device_map()): the session's own streams
        # are ordered after what that stream holds now
        cfg.input_stream = torch.cuda.current_stream(self._dev).cuda_stream
        cfg.order_after_input_stream = 1
        cfg.sampler = self._pool_entry[0]
        if self._part is not None:
            cfg.part = C.pointer(self._part[0])
        self.native_exchange = False
        # Native exchange and the batch record.  A consumer built on the REFERENCE's façade wraps every
        # batch in its own 7-field NamedTuple (fast_trainer/samplers.py:32-88), which reads
        # partition_nids / cached_nids / perm_partition_to_mfg and has no room for `x`: by default the
        # record therefore carries the ownership buckets as well, and the assembled features stay
        # retrievable from the Session (take_native_features).  This repository's own façade sets
        # compact_native_records and gets only what its iterator reads (x, y, MFG).
        self.compact_native_records = False
        self._native_feats = {}
        self._cache_feats = None
        self.p2p = False
        self._peers = None
        if self._distributed:
            pb = config.partition_book
            # P2P transport (opt-in, SPP_DIST_TRANSPORT=p2p): remote rows are read in their owners' partitions, no exchange
            p2p = os.environ.get("SPP_DIST_TRANSPORT", "rccl").lower() == "p2p" and self._x is not None
            if p2p:
                peers = getattr(_p2p_tls, "peers", None)
                if peers is None:
                    key = (self._x.data_ptr(), int(pb.world_size), int(pb.rank))
                    peers = _p2p_auto.get(key)
                    if peers is None:
                        peers = _p2p_auto[key] = p2p_open_peers(self._x)      # collective, once per resident table
                if len(peers.ptrs) != int(pb.world_size):
                    raise RuntimeError(f"P2P transport: {len(peers.ptrs)} peer tables for {int(pb.world_size)} partitions")
                self._peers = peers
            comm = None if p2p else native_comm()
            if self._x is not None and (p2p or (comm is not None and
                                                (comm.world, comm.rank) == (int(pb.world_size), int(pb.rank)))):
                xc = nat.ExchangeCfg()
                xc.comm = None if p2p else comm.handle
                if p2p:
                    self._peer_arr = (C.c_void_p * len(peers.ptrs))(*[v or None for v in peers.ptrs])
                    xc.peer_x_dev = C.cast(self._peer_arr, C.POINTER(C.c_void_p))
                    xc.peer_x_stride_bytes = peers.stride
                    self.p2p = True
                xc.x_local_dev, xc.x_local_rows = self._x.data_ptr(), self._x.size(0)
                xc.row_bytes = self._x.size(1) * self._x.element_size()
                xc.x_local_stride_bytes = _stride_bytes(self._x)
                # who issues the exchanges: a session thread (most overlap) or the consumer inside
                # blocking_get_batch_distributed, at the same program point on every rank (safe next
                # to the caller's own collectives, e.g. DDP all-reduces) -- the default
                xc.issue_on_consumer = int(os.environ.get("SPP_EXCHANGE_ISSUE", "consumer").lower() != "thread")
                if bool(config.use_cache):
                    self._cache_feats = config.cache.device_features()
                    if self._cache_feats.numel():
                        if self._cache_feats.dtype != self._x.dtype or self._cache_feats.size(1) != self._x.size(1):
                            raise RuntimeError("cached_features must match the feature rows in dtype and width")
                        xc.cache_feats_dev, xc.cache_rows = self._cache_feats.data_ptr(), self._cache_feats.size(0)
                        xc.cache_stride_bytes = _stride_bytes(self._cache_feats)
                self._xc = xc
                cfg.exchange = C.pointer(xc)
                self.native_exchange = True
                global _trimmed_for_exchange
                if not _trimmed_for_exchange:
                    # the exchange buffers are hipMalloc'ed by the library as the first groups size them:
                    # hand back what torch's caching allocator hoards from set-up, once per process
                    torch.cuda.empty_cache()
                    _trimmed_for_exchange = True
        h = C.c_void_p()
        try:
            nat.check(L.spp_session_create(C.byref(cfg), C.byref(h)))
        except Exception:
            _SamplerPool.release(self._pool_entry)
            self._pool_entry = None
            raise
        self._h = h
        self._desc = nat.BatchDesc()
        # group-at-a-time delivery (include/spp.h spp_session_next_group / spp_session_export_group): the batches
        # of a sampling group are written by ONE launch into three arenas and handed out one by one
        # Opt-in (SPP_GROUP_DELIVERY=1): it more than halves the host time of a next() (43 vs 88 us) but measured
        # SLOWER than one launch per batch once that path's queue markers were trimmed (0.136 vs 0.133 ms per batch in
        # 192-step windows, 0.163 vs 0.146 in 20-step windows) -- for consumers that are host bound.
        self._group_mode = os.environ.get("SPP_GROUP_DELIVERY", "0") != "0"
        # The default: the group is FETCHED as a whole (spp_session_next_group: one blocking call, three allocations and
        # the views for all its batches) and delivered one launch per batch as the consumer asks (spp_session_export on
        # the group's members) -- the host side of group delivery with the launches of batch-at-a-time delivery.
        # SPP_GROUP_FETCH=0: spp_session_next / spp_session_export per batch, nine allocations each.
        self._member_mode = not self._group_mode and os.environ.get("SPP_GROUP_FETCH", "1") != "0"
        self._open = None                          # [next member, fetched group] while a group is partly handed out
        # opt-in (single-GPU sessions): the records carry TableRows(resident table, n_id) in the place of x and the
        # delivery launch skips the feature gather (labels + MFG + n_id are still delivered) -- for models.SAGE's
        # fused first layer.  Set before the first batch is asked for; SPP_TABLE_FEATURES=1 sets it for every Session.
        self.table_features = os.environ.get("SPP_TABLE_FEATURES", "0") != "0"
        # the same for PARTITIONED sessions with the native exchange / P2P transport: the records carry
        # RowRefs(addresses of the rows: local partition / cache / received rows / a peer's partition) in the place of x;
        # the delivery writes 8 bytes per row (+ a contiguous copy of the rows received over RCCL) instead of assembling x.
        # SPP_ROW_REFS=1 sets it for every such Session.
        self.row_refs = os.environ.get("SPP_ROW_REFS", "0") != "0" and self.native_exchange
        self.export_stream = None                  # set by a consumer that delivers on ONE fixed stream (DevicePrefetcher): the
        self._export_raw = None                    # per-batch path then skips the current-stream lookup / stream context
        self.last_arenas = None                    # the (<= 3) storages behind the views of the batch handed out last
        self._gdescs = (nat.BatchDesc * max(16, int(self._L.spp_session_group_size(self._h))))()   # one per batch of a group
        self._ready = collections.deque()          # (record, ready event, delivery stream) of delivered batches
        self._ended = False
        self.last_ready_event = None               # event after which the batch handed out last is complete
        self._consumer_stream = None
        self._e_id = torch.empty(0, dtype=torch.int64, device=self._dev)
        # (src pointer, rows, row bytes) of the resident feature / label matrices, built once
        self._x_args = (C.c_void_p(self._x.data_ptr()), self._x.size(0), self._x.size(1) * self._x.element_size(),
                        _stride_bytes(self._x)) if self._x is not None and self._x.numel() else (None, 0, 0, 0)
        self._y_args = (C.c_void_p(self._y.data_ptr()), self._y.size(0), self._y.size(1) * self._y.element_size()) \
            if self._y is not None and self._y.numel() else (None, 0, 0)
        self._slice_result = []
        # remote-frequency statistics (count_remote_frequency, fast_sampler.cpp:1093-1103 / :835-880)
        self._freq = None
        self.remote_frequency_tensor = torch.empty(0, dtype=torch.int64)
        self.remote_vertices_ordered_by_freq = torch.empty(0, dtype=torch.int64)
        self._freq_reduced = False

    def _partition_cfg(self):
        """(PartitionCfg, key, keep-alive): the ownership bucketing of the worker distributed branch
        (fast_sampler.cpp:1017-1272), run inside the native sampling chain."""
        cfg = self.config
        pb = cfg.partition_book
        if pb is None:
            raise RuntimeError("Config.distributed needs a partition_book")
        P, rank = int(pb.world_size), int(pb.rank)
        if P > nat.SPP_MAX_PARTS:
            raise RuntimeError(f"at most {nat.SPP_MAX_PARTS} partitions are supported, got {P}")
        offs = [int(v) for v in pb._offsets_host().tolist()]
        pc = nat.PartitionCfg()
        pc.num_parts, pc.rank = P, rank
        for i, v in enumerate(offs):
            pc.offsets[i] = v
        use_cache = bool(cfg.use_cache)
        cmap = cfg.cache.device_map() if use_cache else None
        pc.use_cache = int(use_cache)
        pc.cache_map_dev = cmap.data_ptr() if cmap is not None else None
        pc.cache_map_len = cmap.numel() if cmap is not None else 0
        key = (P, rank, tuple(offs), use_cache, cmap.data_ptr() if cmap is not None else 0,
               cmap.numel() if cmap is not None else 0)
        return (pc, key, cmap)

    # ---- lifetime ----
    def _finish(self):
        """End of data: nothing more will be produced, so the native session ends and the (pooled) sampler goes
        back to the pool NOW -- not when the last Python reference to this object dies.  A training loop
        typically builds the next epoch's iterator while the old one is still bound (`it = iter(sampler)`, a
        StopIteration being handled, ...); the next Session would then find the pool empty and build a second
        sampler (9 ms, 2.4 GB of slots and a second 5 GB stream arena at papers scale)."""
        self.close()

    def close(self):
        # batches this object fetched or delivered ahead and never handed out do not count as consumed
        opened = getattr(self, "_open", None)
        unseen = len(getattr(self, "_ready", None) or ()) + (opened[1][0] - opened[0] if opened is not None else 0)
        if getattr(self, "_ready", None):
            self._ready.clear()
        self._open = None
        if getattr(self, "_h", None) is not None:
            try:                # the counters outlive the native session (get_stats() after the last batch)
                self._final = {
                    "total": int(self._L.spp_session_num_total_batches(self._h)),
                    "consumed": int(self._L.spp_session_num_consumed_batches(self._h)) - unseen,
                    "group": int(self._L.spp_session_group_size(self._h)),
                    "blocked_us": int(self._L.spp_session_blocked_us(self._h)),
                    "blocked_n": int(self._L.spp_session_blocked_occasions(self._h)),
                    "xbytes": self.exchange_bytes(),
                }
            except Exception:
                pass
            self._L.spp_session_destroy(self._h)
            self._h = None
        if getattr(self, "_pool_entry", None) is not None:
            try:
                self._info = sampler_info(self._pool_entry[0])
            except Exception:
                pass
            _SamplerPool.release(self._pool_entry)
            self._pool_entry = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sampler_info(self) -> dict:
        """Which chain variants this Session's sampler runs and what its one-off tables cost (spp_sampler_get_info);
        after the Session has ended: as read when it gave its sampler back."""
        if getattr(self, "_pool_entry", None) is not None:
            self._info = sampler_info(self._pool_entry[0])
        return dict(getattr(self, "_info", None) or {})

    def set_export_stream(self, stream):
        """A consumer that delivers every batch on ONE stream says so once: the per-batch path then uses that stream's
        handle directly (no current-stream lookup, no stream context around every request) and allocates a group's
        arenas under it.  None: back to "the caller's current stream at each request"."""
        self.export_stream = stream
        self._export_raw = stream.cuda_stream if stream is not None else None

    @property
    def consumer_stream(self):
        """The sampler's persistent delivery stream as a torch stream: exporting batches on it keeps
        the per-batch delivery kernel on a hardware queue that no sampling stream shares."""
        if self._consumer_stream is None:
            ptr = self._L.spp_sampler_deliver_stream(self._pool_entry[0]) if self._pool_entry is not None else None
            self._consumer_stream = torch.cuda.ExternalStream(ptr, device=self._dev) if ptr else \
                torch.cuda.Stream(self._dev)
        return self._consumer_stream

    # ---- properties of fast_sampler.cpp:1325-1338 ----
    @property
    def num_total_batches(self) -> int:
        return int(self._L.spp_session_num_total_batches(self._h)) if self._h is not None else self._final["total"]

    @property
    def num_consumed_batches(self) -> int:
        n = int(self._L.spp_session_num_consumed_batches(self._h)) if self._h is not None else self._final["consumed"]
        n -= len(getattr(self, "_ready", ()))            # delivered to this object, not yet handed out
        opened = getattr(self, "_open", None)
        if opened is not None:                           # fetched as a group, members not yet handed out
            n -= opened[1][0] - opened[0]
        return n

    @property
    def group_size(self) -> int:
        """batches sampled (and, in distributed mode, exchanged) together"""
        return int(self._L.spp_session_group_size(self._h)) if self._h is not None else self._final["group"]

    @property
    def approx_num_complete_batches(self) -> int:
        return self.num_consumed_batches

    @property
    def total_blocked_dur(self) -> datetime.timedelta:
        us = int(self._L.spp_session_blocked_us(self._h)) if self._h is not None else self._final["blocked_us"]
        return datetime.timedelta(microseconds=us)

    @property
    def total_blocked_occasions(self) -> int:
        return int(self._L.spp_session_blocked_occasions(self._h)) if self._h is not None else self._final["blocked_n"]

    # ---- batch production ----
    def _next_desc(self, block=True):
        if self._h is None:     # ended (and released) earlier
            return False
        fn = self._L.spp_session_next if block else self._L.spp_session_try_next
        rc = fn(self._h, C.byref(self._desc))
        nat.check(rc)
        if rc == 0:             # end of data: release the sampler right away (see _finish)
            self._finish()
        return rc == 1          # 0: end of data, 2: not ready yet (non-blocking form only)

    def _alloc_mfg(self, counts, want_n_id=True, num_parts=0):
        """One int64 arena per batch for n_id and every hop's rowptr/col (a single allocator call
        instead of 2*hops+1), handed out as views; `e_id` is the shared empty tensor.  With
        num_parts = P > 0 the arena also holds the ownership buckets; then returns
        (out, n_id, adjs, (partition_nids, cached_nids, perm))."""
        H = counts.num_hops
        U = int(counts.num_nodes) if want_n_id else 0
        Ts = [int(counts.T[k]) for k in range(H)]
        Es = [int(counts.E[k]) for k in range(H)]
        P = num_parts
        pc = [int(counts.part_counts[m]) for m in range(P + 1)] if P else []
        owned = sum(pc[:P]) if P else 0
        n_part = (owned + pc[P] + int(counts.num_nodes)) if P else 0
        n_mfg = U + sum(Ts) + H + sum(Es)
        arena = torch.empty(n_mfg + n_part, dtype=torch.int64, device=self._dev)
        base = arena.data_ptr()
        out = nat.MfgOut()
        buckets = None
        if P:
            o = n_mfg
            bounds = [o]
            for m in range(P):
                bounds.append(bounds[-1] + pc[m])
            nids = [arena[bounds[m]:bounds[m + 1]] for m in range(P)]
            flat = arena[bounds[0]:bounds[P]]
            out.parts = (base + 8 * o) if owned else None
            o += owned
            cached = arena[o:o + pc[P]]
            out.cached = (base + 8 * o) if pc[P] else None
            o += pc[P]
            perm = arena[o:o + int(counts.num_nodes)]
            out.perm = (base + 8 * o) if int(counts.num_nodes) else None
            buckets = (nids, cached, perm, flat)
        n_id = None
        off = 0
        if want_n_id:
            n_id = arena[:U]
            out.n_id = base if U else None
            off = U
        adjs = []
        e_id = self._e_id                                              # sample_cpu.hpp:120: always empty
        for k in range(H):
            rp = arena[off:off + Ts[k] + 1]
            out.rowptr[k] = base + 8 * off
            off += Ts[k] + 1
            cl = arena[off:off + Es[k]]
            out.col[k] = (base + 8 * off) if Es[k] else None
            off += Es[k]
            adjs.append((rp, cl, e_id, (Ts[k], int(counts.S[k]))))
        if P:
            return out, n_id, adjs, buckets
        return out, n_id, adjs

    # ---- group-at-a-time delivery ----
    def _fetch_group(self, block: bool) -> bool:
        """Deliver the next sampling group: one allocation per output kind (int64 MFG arena, feature rows, labels),
        ONE launch for all its batches, the records appended to self._ready.  False: not ready yet (block=False)
        or no group left."""
        grp = self._open_group(block)
        if grp is None:
            return False
        n, outs, records, xa, ya, flags, _arenas = grp
        dev = self._dev
        distributed, native, count_remote, rank = flags
        stream = torch.cuda.current_stream(dev)
        nat.check(self._L.spp_session_export_group(self._h, n, outs, xa[0], xa[1], xa[2], xa[3], ya[0], ya[1], ya[2],
                                                   C.c_void_p(stream.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(stream)
        for r in records:
            self._ready.append((self._make_record(r, flags), ev, stream))
        return True

    def _make_record(self, r, flags):
        (x, y, adjs, rng, n_id, nids, flat, cached, perm, pc, U) = r
        distributed, native, count_remote, rank = flags
        if not distributed:
            if x is None:
                x = torch.empty((U, 0), device=self._dev)
            return (x, y, adjs, rng)
        return self._proto_record(x, y, adjs, rng, n_id, nids, flat, cached, perm, pc, native, count_remote, rank)

    def _table_mode(self) -> bool:
        if not self.table_features:
            return False
        if self._distributed or self._x is None:
            raise RuntimeError("table_features: only single-GPU sessions with a feature table deliver TableRows "
                               "(a partitioned session assembles x from three sources: set row_refs there)")
        return True

    def _refs_mode(self) -> bool:
        if not self.row_refs:
            return False
        if not (self._distributed and self.native_exchange and self._x is not None):
            raise RuntimeError("row_refs: only partitioned sessions with the native exchange (RCCL or P2P transport) deliver "
                               "RowRefs (single-GPU sessions: table_features)")
        if not (self._group_mode or self._member_mode):
            raise RuntimeError("row_refs needs group fetch (SPP_GROUP_FETCH=0 is not supported)")
        return True

    def _next_member(self, block: bool):
        """The default delivery: the sampling group is FETCHED as a whole (one blocking call, one allocation per output
        kind for all its batches, the views cut once) and its batches are then delivered one launch each, when asked
        for -- the host work of a batch is one export call and the record, the GPU sees the same per-batch launches as
        with batch-at-a-time calls."""
        if self._open is None:
            if self.export_stream is not None:     # the group's arenas belong to the stream its batches are delivered on
                with torch.cuda.stream(self.export_stream):
                    grp = self._open_group(block)
            else:
                grp = self._open_group(block)
            if grp is None:
                return None
            self._open = [0, grp]
        i, (n, outs, records, xa, ya, flags, arenas) = self._open
        self.last_arenas = arenas
        o = outs[i]
        # (the raw handle of the caller's current stream: torch.cuda.current_stream() builds a Stream object, 2-3 us)
        raw = self._export_raw if self._export_raw is not None else torch._C._cuda_getCurrentRawStream(self._dev.index)
        nat.check(self._L.spp_session_export(self._h, C.byref(o.mfg), xa[0], xa[1], xa[2], xa[3], o.x_out,
                                             ya[0], ya[1], ya[2], o.y_out, C.c_void_p(raw)))
        rec = self._make_record(records[i], flags)
        if i + 1 == n:
            self._open = None
        else:
            self._open[0] = i + 1
        return rec

    def _open_group(self, block: bool):
        """spp_session_next_group + the group's output arenas and views.  None: not ready yet (block=False) or no group
        left; else (n, outs, records, x source args, y source args, (distributed, native, count_remote, rank))."""
        if self._h is None or self._ended:
            return None
        n_c = C.c_int32(0)
        rc = self._L.spp_session_next_group(self._h, 1 if block else 0, self._gdescs, C.byref(n_c))
        nat.check(rc)
        if rc == 2:
            return None
        if rc == 0:                                   # end of data: the sampler goes back to the pool once the queue is handed out
            self._ended = True
            if not self._ready:
                self._finish()
            return None
        n = n_c.value
        global _groups_opened
        _groups_opened += 1
        dev = self._dev
        cfg = self.config
        distributed = self._distributed
        native = self.native_exchange
        P = int(cfg.partition_book.world_size) if distributed else 0
        rank = int(cfg.partition_book.rank) if distributed else 0
        use_cache = bool(cfg.use_cache) if distributed else False
        count_remote = distributed and bool(cfg.count_remote_frequency) and not use_cache
        want_parts = distributed and (not native or count_remote or not self.compact_native_records)
        table = self._table_mode()
        refs = self._refs_mode()
        want_n_id = distributed or table
        want_x = (self._x is not None) and (native or not distributed) and not table and not refs
        want_y = self._y is not None
        H = int(self._gdescs[0].counts.num_hops)
        # ---- sizes (host counts of every batch) and the arena layout
        seg = []                                      # lengths of all int64 segments, batch after batch
        info = []
        Us, bss, n_rem = [], [], []
        for i in range(n):
            d = self._gdescs[i]
            c = d.counts
            U = int(c.num_nodes)
            Ts = [int(c.T[k]) for k in range(H)]
            Ss = [int(c.S[k]) for k in range(H)]
            Es = [int(c.E[k]) for k in range(H)]
            pc = [int(c.part_counts[m]) for m in range(P + 1)] if want_parts else None
            first = len(seg)
            if want_n_id:
                seg.append(U)
            if refs:
                seg.append(U)                          # one address per row
            for k in range(H):
                seg.append(Ts[k] + 1)
                seg.append(Es[k])
            if want_parts:
                seg.extend(pc[:P])
                seg.append(pc[P])
                seg.append(U)
            info.append((first, U, Ts, Ss, Es, pc, int(d.start), int(d.stop)))
            Us.append(U)
            bss.append(int(d.stop) - int(d.start))
            if refs:                                   # rows received over RCCL for this batch (P2P: none are copied)
                n_rem.append(0 if self.p2p else U - int(c.part_counts[rank]) - int(c.part_counts[P]))
        # arena sizes are rounded up to a coarse grid (<= 3 %): group totals differ by fractions of a percent from
        # group to group, and the caching allocator would otherwise keep meeting sizes no cached block fits
        n_seg = sum(seg)
        arena = torch.empty(_coarse(n_seg), dtype=torch.int64, device=dev)
        if seg:
            seg.append(arena.numel() - n_seg)
        views = arena.split(seg) if seg else ()
        base = arena.data_ptr()
        x_views = y_views = None
        row_b = 0
        if want_x:
            F = self._x.size(1)
            row_b = F * self._x.element_size()
            # every batch starts on a 16-byte boundary of the arena (rows of 16k + 8 bytes: on an even row), so that the
            # delivery may store 16-byte pieces whatever the batches before it hold (gather_body.hip.h, kVecSpan)
            x_align = 16 // math.gcd(row_b, 16)
            x_split = []
            for U in Us:
                x_split += [U, (-U) % x_align]
            n_rows = sum(x_split)
            x_arena = torch.empty((_coarse(n_rows), F), dtype=self._x.dtype, device=dev)
            x_views = x_arena.split(x_split + [x_arena.size(0) - n_rows])[0::2]
            x_base = x_arena.data_ptr()
        xr_arena = xr_views = None
        if refs:
            F = self._x.size(1)
            row_b = F * self._x.element_size()
            xr_align = 16 // math.gcd(row_b, 16)        # every batch's received rows start 16-byte aligned
            xr_split = []
            for r_ in n_rem:
                xr_split += [r_, (-r_) % xr_align]
            if sum(n_rem):
                xr_arena = torch.empty((_coarse(sum(xr_split)), F), dtype=self._x.dtype, device=dev)
                xr_views = xr_arena.split(xr_split + [xr_arena.size(0) - sum(xr_split)])[0::2]
                xr_base = xr_arena.data_ptr()
        if want_y:
            y_arena = torch.empty((sum(bss), self._y.size(1)), dtype=self._y.dtype, device=dev)
            y_views = y_arena.split(bss)
            y_base, yrow_b = y_arena.data_ptr(), self._y.size(1) * self._y.element_size()
        outs = (nat.GroupOut * n)()
        e_id = self._e_id
        off = 0                                       # running element offset into the int64 arena
        xo = yo = xro = 0
        records = []
        for i in range(n):
            first, U, Ts, Ss, Es, pc, start, stop = info[i]
            o = outs[i]
            k = first
            n_id = None
            if want_n_id:
                n_id = views[k]
                o.mfg.n_id = (base + 8 * off) if U else None
                off += U
                k += 1
            addr = None
            if refs:
                addr = views[k]
                o.mfg.row_addr = (base + 8 * off) if U else None
                off += U
                k += 1
            adjs = []
            for h in range(H):
                o.mfg.rowptr[h] = base + 8 * off
                off += Ts[h] + 1
                o.mfg.col[h] = (base + 8 * off) if Es[h] else None
                off += Es[h]
                adjs.append((views[k], views[k + 1], e_id, (Ts[h], Ss[h])))
                k += 2
            nids = flat = cached = perm = None
            if want_parts:
                owned = sum(pc[:P])
                o.mfg.parts = (base + 8 * off) if owned else None
                nids = list(views[k:k + P])
                flat = arena[off:off + owned]
                off += owned
                k += P
                cached = views[k]
                o.mfg.cached = (base + 8 * off) if pc[P] else None
                off += pc[P]
                perm = views[k + 1]
                o.mfg.perm = (base + 8 * off) if U else None
                off += U
            x = y = None
            if table:
                x = TableRows(self._x, n_id)
            if refs:
                xr = None
                if n_rem[i]:
                    xr = xr_views[i]
                    o.mfg.x_remote = xr_base + xro * row_b
                    xro += n_rem[i] + (-n_rem[i]) % xr_align
                x = RowRefs(addr, n_id, self._x.size(1), self._x.dtype, xr, (self._x, self._cache_feats, self._peers))
            if want_x:
                x = x_views[i]
                o.x_out = (x_base + xo * row_b) if U else None
                xo += U + (-U) % x_align
            if want_y:
                y = y_views[i]
                o.y_out = (y_base + yo * yrow_b) if bss[i] else None
                yo += bss[i]
            records.append((x, y, adjs, (start, stop), n_id, nids, flat, cached, perm, pc, U))
        xa = self._x_args if (want_x and not native) else (None, 0, 0, 0)
        ya = self._y_args if want_y else (None, 0, 0)
        arenas = [t for t in (arena, x_arena if want_x else None, y_arena if want_y else None, xr_arena) if t is not None]
        return n, outs, records, xa, ya, (distributed, native, count_remote, rank), arenas

    def _proto_record(self, x, y, adjs, rng, n_id, nids, flat, cached, perm, pc, native, count_remote, rank):
        b = ProtoDistributedBatch()
        feat_dim = self._x.size(1) if self._x is not None else 0
        feat_dtype = self._x.dtype if self._x is not None else torch.float16
        if nids is None:                               # compact native record: only what this repository's iterator reads
            nids, flat = [], None
            cached = perm = torch.empty(0, dtype=torch.int64, device=self._dev)
        b.x = x if native else None
        b.partition_nids = nids
        b.partition_nids_flat = flat
        b.cached_nids = cached
        b.perm_partition_to_mfg = perm
        if pc is not None:
            b.partition_counts = pc
        b.sliced_cpu_features = torch.empty((0, feat_dim), dtype=feat_dtype)   # all local rows are in HBM
        b.sliced_cpu_labels = y if y is not None else torch.zeros(0)
        b.adjs = adjs
        b.idx_range = rng
        b.n_id = n_id
        if count_remote:
            self._count_remote(b.partition_nids, rank)
        return b

    def _pop_ready(self, block: bool):
        """Hand out the next delivered batch; keeps one group delivered AHEAD of the one being handed out, so that
        its (single) delivery launch overlaps the consumption of this one."""
        if not self._ready and not self._fetch_group(block):
            return None
        rec, ev, stream = self._ready.popleft()
        self.last_ready_event = ev
        cur = torch.cuda.current_stream(self._dev)
        if cur != stream:                              # delivered on another stream than the caller is on now
            cur.wait_event(ev)
        G = self.group_size
        if len(self._ready) < _LOOKAHEAD_GROUPS * G and not self._ended:
            # With the native exchange the look-ahead is taken at THIS program point on every rank, blocking: how
            # many groups a rank has exported decides which chains and exchanges its session threads may issue,
            # and a rank that ran ahead of its peers opportunistically would sit in a collective (or in quiesce())
            # that the others only join after a barrier it never reaches.  Consumer-issued exchanges are issued
            # here.  Single-GPU sessions look ahead opportunistically.
            self._fetch_group(self.native_exchange)
        if self._ended and not self._ready:
            self._finish()
        return rec

    def blocking_get_batch(self, _block=True):
        """-> None or (x, y-or-None, [(rowptr, col, e_id, (T, S)) ...], (start, stop))
        (worker non-distributed branch, fast_sampler.cpp:1004-1016)."""
        if self._group_mode:
            return self._pop_ready(_block)
        if self._member_mode:
            return self._next_member(_block)
        if not self._next_desc(_block):
            return None
        d = self._desc
        c = d.counts
        table = self._table_mode()
        out, n_id, adjs = self._alloc_mfg(c, want_n_id=table)          # n_id is not part of the tuple
        x = y = None
        if table:
            x = None
        elif self._x is not None:
            x = torch.empty((c.num_nodes, self._x.size(1)), dtype=self._x.dtype, device=self._dev)
        else:
            x = torch.empty((c.num_nodes, 0), device=self._dev)
        if self._y is not None:
            y = torch.empty((d.stop - d.start, self._y.size(1)), dtype=self._y.dtype, device=self._dev)
        self._export(out, x, y)
        return (TableRows(self._x, n_id) if table else x, y, adjs, (int(d.start), int(d.stop)))

    def try_get_batch(self):
        """Non-blocking (fast_sampler.cpp:658-670): None when no batch is ready yet or none is left."""
        return self.blocking_get_batch(_block=False)

    def _export(self, out, x, y):
        xa = self._x_args if x is not None and self._x is not None else (None, 0, 0, 0)
        ya = self._y_args if y is not None and self._y is not None else (None, 0, 0)
        nat.check(self._L.spp_session_export(
            self._h, C.byref(out),
            xa[0], xa[1], xa[2], xa[3], C.c_void_p(x.data_ptr()) if xa[0] is not None and x.numel() else None,
            ya[0], ya[1], ya[2], C.c_void_p(y.data_ptr()) if ya[0] is not None and y.numel() else None,
            C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)))

    def blocking_get_batch_distributed(self, _block=True):
        """-> None or ProtoDistributedBatch (worker distributed branch, fast_sampler.cpp:1017-1272)."""
        if self._group_mode or self._member_mode:
            b = self._pop_ready(_block) if self._group_mode else self._next_member(_block)
            if b is not None and self.native_exchange and not self.compact_native_records:
                # at most the two newest batches (double buffering); a consumer that never asks loses nothing
                while len(self._native_feats) >= 2:
                    self._native_feats.pop(next(iter(self._native_feats)))
                self._native_feats[b.idx_range] = (b.x, b.n_id)
            return b
        if not self._next_desc(_block):
            return None
        d = self._desc
        c = d.counts
        cfg = self.config
        pb = cfg.partition_book
        P, rank = int(pb.world_size), int(pb.rank)
        use_cache = bool(cfg.use_cache)
        if self.native_exchange:
            return self._native_distributed_batch(d, c, P, rank, use_cache)
        # the ownership buckets were built by the sampling chain; their sizes came with the counts
        out, n_id, adjs, (nids, cached, perm, flat) = self._alloc_mfg(c, num_parts=P)
        y = None
        if self._y is not None:
            y = torch.empty((d.stop - d.start, self._y.size(1)), dtype=self._y.dtype, device=self._dev)
        self._export(out, None, y)
        b = ProtoDistributedBatch()
        b.partition_nids = nids
        b.partition_nids_flat = flat
        b.cached_nids = cached
        b.perm_partition_to_mfg = perm
        b.partition_counts = [int(c.part_counts[m]) for m in range(P + 1)]
        feat_dim = self._x.size(1) if self._x is not None else 0
        feat_dtype = self._x.dtype if self._x is not None else torch.float16
        b.sliced_cpu_features = torch.empty((0, feat_dim), dtype=feat_dtype)   # all local rows are in HBM
        b.sliced_cpu_labels = y if y is not None else torch.zeros(0)
        b.adjs = adjs
        b.idx_range = (int(d.start), int(d.stop))
        b.n_id = n_id
        if cfg.count_remote_frequency and not use_cache:
            self._count_remote(b.partition_nids, rank)
        return b

    def try_get_batch_distributed(self):
        """Non-blocking (fast_sampler.cpp:672-711): None when the next batch (index order) is not ready."""
        return self.blocking_get_batch_distributed(_block=False)

    def _native_distributed_batch(self, d, c, P, rank, use_cache):
        """The exchange already ran natively (session.hip): one launch delivers the MFG, the labels
        and x assembled from {local partition, rows received over RCCL, VIP cache}."""
        cfg = self.config
        count_remote = bool(cfg.count_remote_frequency) and not use_cache
        want_parts = count_remote or not self.compact_native_records
        if want_parts:
            out, n_id, adjs, (nids, cached, perm, flat) = self._alloc_mfg(c, num_parts=P)
        else:
            out, n_id, adjs = self._alloc_mfg(c)
            nids, flat = [], None
            cached = perm = torch.empty(0, dtype=torch.int64, device=self._dev)
        x = torch.empty((c.num_nodes, self._x.size(1)), dtype=self._x.dtype, device=self._dev)
        y = None
        if self._y is not None:
            y = torch.empty((d.stop - d.start, self._y.size(1)), dtype=self._y.dtype, device=self._dev)
        ya = self._y_args if y is not None else (None, 0, 0)
        nat.check(self._L.spp_session_export(
            self._h, C.byref(out), None, 0, 0, 0, C.c_void_p(x.data_ptr()) if x.numel() else None,
            ya[0], ya[1], ya[2], C.c_void_p(y.data_ptr()) if ya[0] is not None and y.numel() else None,
            C.c_void_p(torch.cuda.current_stream(self._dev).cuda_stream)))
        b = ProtoDistributedBatch()
        b.x = x
        b.partition_nids = nids
        b.partition_nids_flat = flat
        b.cached_nids = cached
        b.perm_partition_to_mfg = perm
        b.partition_counts = [int(c.part_counts[m]) for m in range(P + 1)]
        b.sliced_cpu_features = torch.empty((0, self._x.size(1)), dtype=self._x.dtype)
        b.sliced_cpu_labels = y if y is not None else torch.zeros(0)
        b.adjs = adjs
        b.idx_range = (int(d.start), int(d.stop))
        b.n_id = n_id
        if count_remote:
            self._count_remote(b.partition_nids, rank)
        if not self.compact_native_records:
            # at most the two newest batches (double buffering); a consumer that never asks loses nothing
            while len(self._native_feats) >= 2:
                self._native_feats.pop(next(iter(self._native_feats)))
            self._native_feats[b.idx_range] = (x, n_id)
        return b

    def take_native_features(self, idx_range):
        """(x, n_id) of the batch covering seeds idx_range = (start, stop) or slice(start, stop), when the
        exchange ran natively and the consumer's record has no field for them (the reference's
        ProtoDistributedBatch, fast_trainer/samplers.py:32-68); None when unknown."""
        if isinstance(idx_range, slice):
            idx_range = (idx_range.start, idx_range.stop)
        return self._native_feats.pop((int(idx_range[0]), int(idx_range[1])), None)

    def quiesce(self):
        """Block until the session has nothing more to issue without further consumption and its GPU
        work is done (call on every rank before collectives of another communicator)."""
        if self._h is not None:
            nat.check(self._L.spp_session_quiesce(self._h))

    def exchange_bytes(self):
        """(sent, received) bytes of the native exchange so far."""
        if self._h is None:
            return self._final["xbytes"]
        a, b = C.c_int64(0), C.c_int64(0)
        nat.check(self._L.spp_session_exchange_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    # ---- host-memory slicing for peers (fast_sampler.cpp:716-775): nothing lives in host memory ----
    def async_slice_tensors(self, ids: List[torch.Tensor], my_rank: int):
        res = []
        for t in ids:
            n = t.numel()
            pos = torch.arange(n, dtype=torch.int64)
            neg = t.cpu() < 0 if n else torch.zeros(0, dtype=torch.bool)
            res.append([torch.empty(0, dtype=torch.int64), pos[~neg], pos[neg]])
        self._slice_result = res

    def wait_slice_tensors(self):
        return None

    def get_slice_tensors(self):
        return self._slice_result

    # ---- remote frequency counting (simulation cache strategy) ----
    def _count_remote(self, partition_nids, rank):
        # The ids were written by the delivery launch that was just enqueued: the counting runs on THAT stream.  (A consumer
        # that named its delivery stream once -- DevicePrefetcher / DeviceDistributedPrefetcher in direct mode -- calls
        # without a stream context; on its current stream the index_add_ would read ids no launch has written yet.)
        st = self.export_stream if self.export_stream is not None else torch.cuda.current_stream(self._dev)
        with torch.cuda.stream(st):
            n = self._rowptr.numel() - 1
            if self._freq is None:
                self._freq = torch.zeros(n, dtype=torch.int64, device=self._dev)
            for m, t in enumerate(partition_nids):
                if m != rank and t.numel():
                    self._freq.index_add_(0, t, torch.ones_like(t))
        self._freq_stream = st

    def reduce_multithreaded_frequency_counts(self):
        if self._freq_reduced:
            return
        if self._freq is not None:
            fst = getattr(self, "_freq_stream", None)
            if fst is not None:                        # the counts were accumulated on the delivery stream
                torch.cuda.current_stream(self._dev).wait_stream(fst)
            nz = self._freq.nonzero().view(-1)
            f = self._freq[nz]
            order = torch.argsort(f, descending=True, stable=True)
            self.remote_frequency_tensor = f[order].cpu()
            self.remote_vertices_ordered_by_freq = nz[order].cpu()
        self._freq_reduced = True

    def get_n_most_freq_remote_vertices(self, n: int) -> torch.Tensor:
        self.reduce_multithreaded_frequency_counts()
        return self.remote_vertices_ordered_by_freq[:n].clone()


def _resident_concat(x_gpu: torch.Tensor, x_cpu: torch.Tensor) -> torch.Tensor:
    """All local feature rows in one HBM table (rows padded as in _ResidentCache.get_rows):
    rows [0, |x_gpu|) then the former host rows."""
    key_t = x_cpu
    dev = _device()
    key = ("cat", x_gpu.data_ptr(), x_cpu.data_ptr(), tuple(x_gpu.shape), tuple(x_cpu.shape), dev.index)
    with _resident._lock:
        hit = _resident._d.get(key)
        if hit is not None and hit[0] is key_t:
            return hit[1]
        n0, n1, F = x_gpu.size(0), x_cpu.size(0), x_gpu.size(1)
        se = _row_stride_elems(F, x_gpu.element_size())
        buf = torch.empty((n0 + n1, se), dtype=x_gpu.dtype, device=dev)
        r = buf[:, :F]
        r[:n0].copy_(x_gpu)
        r[n0:].copy_(x_cpu)
        _resident._d[key] = (key_t, r)
        return r


# --------------------------------------------------------------------------------------------
# free functions   (fast_sampler.cpp:1339-1366)
# --------------------------------------------------------------------------------------------
class _ThreadGen(threading.local):
    """`thread_local std::mt19937 gen` of the calling thread (sample_cpu.hpp:11): default-seeded
    (5489) and never re-seeded by the free functions, so its position carries over between calls."""

    def __init__(self):
        self.seed = 5489
        self.pos = 0


_gen = _ThreadGen()


def _sample_once(rowptr, col, idx, sizes, replace=False):
    L = _lib()
    dev = _device()
    sizes = _as_i64_list(sizes)
    rowptr_d = _resident.get(rowptr, torch.int64)
    col_d = _resident.get(col, torch.int64)
    idx_d = idx.to(dev, torch.int64).contiguous()
    entry = _SamplerPool.acquire(rowptr_d, col_d, sizes, max(1, idx_d.numel()), 1, dev.index, replace)
    try:
        h = entry[0]
        st = _stream_ptr()
        nat.check(L.spp_sampler_sample(h, 0, _ptr(idx_d), idx_d.numel(), _gen.seed, _gen.pos, st))
        cnt = nat.MfgCounts()
        nat.check(L.spp_sampler_wait(h, 0, C.byref(cnt)))
        _gen.pos += int(cnt.draws)
        out = nat.MfgOut()
        n_id = torch.empty(cnt.num_nodes, dtype=torch.int64, device=dev)
        out.n_id = n_id.data_ptr() if n_id.numel() else None
        adjs = []
        e_id = torch.empty(0, dtype=torch.int64, device=dev)
        for k in range(cnt.num_hops):
            rp = torch.empty(cnt.T[k] + 1, dtype=torch.int64, device=dev)
            cl = torch.empty(cnt.E[k], dtype=torch.int64, device=dev)
            out.rowptr[k] = rp.data_ptr()
            out.col[k] = cl.data_ptr() if cl.numel() else None
            adjs.append((rp, cl, e_id, (int(cnt.T[k]), int(cnt.S[k]))))
        nat.check(L.spp_sampler_export(h, 0, C.byref(out), st))
        torch.cuda.current_stream().synchronize()
    finally:
        _SamplerPool.release(entry)
    return n_id, adjs


def sample_adj(rowptr, col, idx, num_neighbors: int, replace: bool, pin_memory: bool = False):
    """-> (rowptr, col, n_id, e_id); n_id is int32 like the reference's tensor overload
    (sample_cpu.hpp:154-165)."""
    n_id, adjs = _sample_once(rowptr, col, idx, [int(num_neighbors)], replace)
    rp, cl, e_id, _ = adjs[0]
    return rp, cl, n_id.to(torch.int32), e_id


def multilayer_sample(idx, sizes, rowptr, col, pin_memory: bool = False):
    """-> (n_id int64, [(rowptr, col, e_id, (T, S)) ...] outermost hop first) (fast_sampler.cpp:229-236)."""
    return _sample_once(rowptr, col, idx, sizes, False)


def full_sample(x, y, rowptr, col, idx, batch_size, sizes, skip_nonfull_batch=False, pin_memory=False):
    """Pre-sampler of fast_sampler.cpp:310-366: every batch of `idx`, as one list (the reference
    returns one list per OpenMP thread; callers chain them)."""
    cfg = Config()
    cfg.x_cpu, cfg.y, cfg.rowptr, cfg.col, cfg.idx = x, y, rowptr, col, idx
    cfg.batch_size, cfg.sizes, cfg.skip_nonfull_batch = int(batch_size), list(sizes), bool(skip_nonfull_batch)
    s = Session(1, _MAX_SLOTS, cfg)
    out = []
    try:
        while True:
            b = s.blocking_get_batch()
            if b is None:
                break
            out.append(b)
    finally:
        s.close()
    return [out]


def to_row_major(t: torch.Tensor) -> torch.Tensor:
    """fast_sampler.cpp:281-308."""
    if t.dim() != 2:
        raise RuntimeError("only support 2D tensors")
    tr, tc = t.size(0), t.size(1)
    if t.stride(0) == tc and t.stride(1) == 1:
        return t                               # already row major
    if not (t.stride(0) == 1 and t.stride(1) == tr):
        raise RuntimeError("input has unrecognizable stides")
    L = _lib()
    storage = t.t().to(_device())              # the column-major storage, viewed as contiguous [tc, tr]
    assert storage.is_contiguous()
    out = torch.empty((tr, tc), dtype=t.dtype, device=_device())
    nat.check(L.spp_to_row_major(_ptr(storage), tr, tc, t.element_size(), _ptr(out), _stream_ptr()))
    return out if t.is_cuda else out.cpu()


def async_errors(clear: bool = True, device=None) -> int:
    """Mask of the row-index errors kernels on `device` have met so far (include/spp.h SPP_AERR_*: 1 = a
    serial_index / gather index outside its table, 2 = served id, 4 = assembly source); the kernels clamp such an
    index to row 0 instead of faulting, so this is how a bad index surfaces.  A bit is visible once the offending
    kernel has RUN: poll after a synchronising point (`.cpu()`, `synchronize()`)."""
    dev = _device().index if device is None else torch.device(device).index
    return int(_lib().spp_async_errors(int(dev or 0), 1 if clear else 0))


def serial_index(inp: torch.Tensor, idx: torch.Tensor, n=None, pin_memory: bool = False) -> torch.Tensor:
    """out[i,:] = in[idx[i],:] (fast_sampler.cpp:238-279); `n` limits/sets the output rows.  Asynchronous: an
    index outside `inp` copies row 0 (the reference reads out of bounds) and raises SPP_AERR_GATHER_INDEX, which the
    caller reads with async_errors() after its next synchronising point."""
    if isinstance(n, bool):                    # serial_index(in, idx, pin_memory) overload
        n, pin_memory = None, n
    if not ((inp.dim() == 2 and inp.stride(-1) == 1) or inp.size(-1) == 1):
        raise RuntimeError("input must be 2D row-major tensor")
    L = _lib()
    src = _resident.get(inp)
    ind = idx.to(_device(), torch.int64).contiguous()
    n_out = ind.numel() if n is None else int(n)
    f = inp.size(-1)
    out = torch.empty((n_out, f), dtype=inp.dtype, device=_device())
    nat.check(L.spp_gather_rows(_ptr(src), src.size(0), f * src.element_size(), _ptr(ind), 8, ind.numel(), n_out,
                                _ptr(out), _stream_ptr()))
    return out
