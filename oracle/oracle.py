"""TEST INFRASTRUCTURE ONLY -- ctypes/numpy view of oracle/liborc.so.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
import this module; the product package never does.  See oracle/spp_oracle.h
for the reference lines each function restates and for the parity status
(pinned against the compiled, unmodified reference through tests/golden).
"""
import ctypes as C
import os
import subprocess
from typing import List, NamedTuple, Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")


def build(force: bool = False) -> str:
    """Compile liborc.so with gcc if it is missing or stale."""
    src = os.path.join(_HERE, "spp_oracle.c")
    hdr = os.path.join(_HERE, "spp_oracle.h")
    stale = (not os.path.exists(_LIB_PATH)
             or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(src), os.path.getmtime(hdr)))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        p = C.c_void_p
        i64 = C.c_int64
        L.orc_mt_fill.argtypes = [C.c_uint32, i64, i64, p]
        L.orc_batch_seed.argtypes = [C.c_int32]
        L.orc_batch_seed.restype = C.c_uint32
        L.orc_batch_ranges.argtypes = [i64, i64, C.c_int, C.c_int, i64, p]
        L.orc_batch_ranges.restype = i64
        L.orc_mt_seed.argtypes = [p, C.c_uint32]
        L.orc_multilayer_sample.argtypes = [p, p, p, i64, p, C.c_int, p]
        L.orc_multilayer_sample.restype = p
        L.orc_sample_adj.argtypes = [p, p, p, i64, C.c_int32, C.c_int, p]
        L.orc_sample_adj.restype = p
        L.orc_mfg_free.argtypes = [p]
        for name in ("orc_mfg_num_nodes", "orc_mfg_num_draws", "orc_mfg_total_edges"):
            getattr(L, name).argtypes = [p]
            getattr(L, name).restype = i64
        L.orc_mfg_num_hops.argtypes = [p]
        L.orc_mfg_num_hops.restype = C.c_int
        L.orc_mfg_n_id.argtypes = [p]
        L.orc_mfg_n_id.restype = p
        for name in ("orc_mfg_hop_T", "orc_mfg_hop_S", "orc_mfg_hop_E"):
            getattr(L, name).argtypes = [p, C.c_int]
            getattr(L, name).restype = i64
        for name in ("orc_mfg_hop_rowptr", "orc_mfg_hop_col"):
            getattr(L, name).argtypes = [p, C.c_int]
            getattr(L, name).restype = p
        L.orc_serial_index.argtypes = [p, i64, p, i64, i64, p]
        L.orc_to_row_major.argtypes = [p, i64, i64, i64, p]
        L.orc_nid2partid.argtypes = [p, C.c_int, p, i64, p]
        L.orc_nid2localnid.argtypes = [p, C.c_int, p, i64, p]
        L.orc_nid_is_local.argtypes = [p, C.c_int, p, i64, p]
        L.orc_cache_create.argtypes = [p, i64, i64]
        L.orc_cache_create.restype = p
        L.orc_cache_free.argtypes = [p]
        L.orc_cache_nid_is_cached.argtypes = [p, p, i64, p]
        L.orc_cache_nid2cachenid.argtypes = [p, p, i64, p]
        L.orc_partition_batch.argtypes = [p, i64, p, C.c_int, C.c_int, C.c_int, p, i64,
                                          p, p, p, p, p, p, p]
        L.orc_partition_batch.restype = C.c_int
        L.orc_epoch_run.argtypes = [p, p, p, i64, p, p, p, i64, p, C.c_int, C.c_int, p]
        L.orc_epoch_run.restype = C.c_int
        _lib = L
    return _lib


def _i64(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=np.int64)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class MT(C.Structure):
    """std::mt19937 state (orc_mt)."""
    _fields_ = [("mt", C.c_uint32 * 624), ("idx", C.c_int)]

    @classmethod
    def seeded(cls, seed: int) -> "MT":
        s = cls()
        lib().orc_mt_seed(C.byref(s), C.c_uint32(seed & 0xFFFFFFFF))
        return s


def batch_seed(stop: int) -> int:
    return int(lib().orc_batch_seed(C.c_int32(stop)))


def mt19937(seed: int, n: int, skip: int = 0) -> np.ndarray:
    out = np.empty(n, dtype=np.uint32)
    lib().orc_mt_fill(C.c_uint32(seed & 0xFFFFFFFF), skip, n, _ptr(out))
    return out


def batch_ranges(n: int, batch_size: int, skip_nonfull_batch: bool = False,
                 force_exact_num_batches: bool = False, exact_num_batches: int = 0) -> np.ndarray:
    nb = lib().orc_batch_ranges(n, batch_size, int(skip_nonfull_batch),
                                int(force_exact_num_batches), exact_num_batches, None)
    out = np.empty((nb, 2), dtype=np.int32)
    lib().orc_batch_ranges(n, batch_size, int(skip_nonfull_batch),
                           int(force_exact_num_batches), exact_num_batches, _ptr(out))
    return out


class Hop(NamedTuple):
    rowptr: np.ndarray   # int64[T+1]
    col: np.ndarray      # int64[E]
    size: tuple          # (T, S)


class MFG(NamedTuple):
    n_id: np.ndarray     # int64[U]
    hops: List[Hop]      # outermost hop first
    draws: int

    @property
    def num_edges(self) -> int:
        return int(sum(h.col.shape[0] for h in self.hops))


def _collect(h) -> MFG:
    L = lib()
    try:
        U = L.orc_mfg_num_nodes(h)
        n_id = np.ctypeslib.as_array(C.cast(L.orc_mfg_n_id(h), C.POINTER(C.c_int64)), shape=(max(U, 1),))[:U].copy()
        hops = []
        for k in range(L.orc_mfg_num_hops(h)):
            T, S, E = L.orc_mfg_hop_T(h, k), L.orc_mfg_hop_S(h, k), L.orc_mfg_hop_E(h, k)
            rp = np.ctypeslib.as_array(C.cast(L.orc_mfg_hop_rowptr(h, k), C.POINTER(C.c_int64)), shape=(T + 1,)).copy()
            cl = np.ctypeslib.as_array(C.cast(L.orc_mfg_hop_col(h, k), C.POINTER(C.c_int64)), shape=(max(E, 1),))[:E].copy()
            hops.append(Hop(rp, cl, (int(T), int(S))))
        return MFG(n_id, hops, int(L.orc_mfg_num_draws(h)))
    finally:
        L.orc_mfg_free(h)


def multilayer_sample(rowptr, col, seeds, sizes: Sequence[int], rng: MT) -> MFG:
    rowptr, col, seeds, sz = _i64(rowptr), _i64(col), _i64(seeds), _i64(list(sizes))
    h = lib().orc_multilayer_sample(_ptr(rowptr), _ptr(col), _ptr(seeds), seeds.shape[0],
                                    _ptr(sz), sz.shape[0], C.byref(rng))
    return _collect(h)


def sample_batch(rowptr, col, idx, start: int, stop: int, sizes: Sequence[int]) -> MFG:
    """What a Session worker produces for batch range (start, stop) (fast_sampler.cpp:990-1000)."""
    rng = MT.seeded(batch_seed(stop))
    return multilayer_sample(rowptr, col, _i64(idx)[start:stop], sizes, rng)


def sample_adj(rowptr, col, idx, num_neighbors: int, replace: bool, rng: MT) -> MFG:
    rowptr, col, idx = _i64(rowptr), _i64(col), _i64(idx)
    h = lib().orc_sample_adj(_ptr(rowptr), _ptr(col), _ptr(idx), idx.shape[0],
                             num_neighbors, int(replace), C.byref(rng))
    return _collect(h)


def serial_index(inp: np.ndarray, idx, n: Optional[int] = None) -> np.ndarray:
    inp = np.ascontiguousarray(inp)
    idx = _i64(idx)
    if n is None:
        n = idx.shape[0]
    f = inp.shape[-1] if inp.ndim == 2 else 1
    out = np.zeros((n, f), dtype=inp.dtype)
    lib().orc_serial_index(_ptr(inp), f * inp.dtype.itemsize, _ptr(idx), idx.shape[0], n, _ptr(out))
    return out


def to_row_major(col_major_storage: np.ndarray, tr: int, tc: int) -> np.ndarray:
    src = np.ascontiguousarray(col_major_storage)
    out = np.empty((tr, tc), dtype=src.dtype)
    lib().orc_to_row_major(_ptr(src), tr, tc, src.dtype.itemsize, _ptr(out))
    return out


def nid2partid(offsets, nids) -> np.ndarray:
    offsets, nids = _i64(offsets), _i64(nids)
    out = np.empty(nids.shape[0], dtype=np.int64)
    lib().orc_nid2partid(_ptr(offsets), offsets.shape[0], _ptr(nids), nids.shape[0], _ptr(out))
    return out


def nid2localnid(offsets, nids, partition_idx: int) -> np.ndarray:
    offsets, nids = _i64(offsets), _i64(nids)
    out = np.empty(nids.shape[0], dtype=np.int64)
    lib().orc_nid2localnid(_ptr(offsets), partition_idx, _ptr(nids), nids.shape[0], _ptr(out))
    return out


def nid_is_local(offsets, nids, rank: int) -> np.ndarray:
    offsets, nids = _i64(offsets), _i64(nids)
    out = np.empty(nids.shape[0], dtype=np.uint8)
    lib().orc_nid_is_local(_ptr(offsets), rank, _ptr(nids), nids.shape[0], _ptr(out))
    return out.astype(bool)


class Cache:
    """Dense direct-map cache tables (range_partition_book.cpp:116-195)."""

    def __init__(self, cached_vertices, table_len: int):
        self.cached_vertices = _i64(cached_vertices)
        self._h = lib().orc_cache_create(_ptr(self.cached_vertices), self.cached_vertices.shape[0], table_len)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_cache_free(self._h)
            self._h = None

    def nid_is_cached(self, nids) -> np.ndarray:
        nids = _i64(nids)
        out = np.empty(nids.shape[0], dtype=np.uint8)
        lib().orc_cache_nid_is_cached(self._h, _ptr(nids), nids.shape[0], _ptr(out))
        return out.astype(bool)

    def nid2cachenid(self, nids) -> np.ndarray:
        nids = _i64(nids)
        out = np.empty(nids.shape[0], dtype=np.int64)
        lib().orc_cache_nid2cachenid(self._h, _ptr(nids), nids.shape[0], _ptr(out))
        return out


class Partitioned(NamedTuple):
    partition_nids: List[np.ndarray]
    cached_nids: np.ndarray
    perm_partition_to_mfg: np.ndarray
    local_on_cpu: np.ndarray


def partition_batch(n_id, offsets, rank: int, cache: Optional[Cache] = None,
                    x_gpu_rows: int = 0) -> Partitioned:
    n_id, offsets = _i64(n_id), _i64(offsets)
    U, P = n_id.shape[0], offsets.shape[0] - 1
    parts = np.empty(max(U, 1), dtype=np.int64)
    counts = np.zeros(P, dtype=np.int64)
    cached = np.empty(max(U, 1), dtype=np.int64)
    perm = np.empty(max(U, 1), dtype=np.int64)
    cpu_local = np.empty(max(U, 1), dtype=np.int64)
    n_cached = C.c_int64(0)
    n_cpu = C.c_int64(0)
    rc = lib().orc_partition_batch(_ptr(n_id), U, _ptr(offsets), P, rank, int(cache is not None),
                                   cache._h if cache is not None else None, x_gpu_rows,
                                   _ptr(parts), _ptr(counts), _ptr(cached), C.byref(n_cached),
                                   _ptr(perm), _ptr(cpu_local), C.byref(n_cpu))
    assert rc == 0
    bounds = np.concatenate([[0], np.cumsum(counts)])
    plist = [parts[bounds[m]:bounds[m + 1]].copy() for m in range(P)]
    return Partitioned(plist, cached[:n_cached.value].copy(), perm[:U].copy(), cpu_local[:n_cpu.value].copy())


class EpochStats(C.Structure):
    _fields_ = [("batches", C.c_int64), ("sampled_edges", C.c_int64), ("mfg_nodes", C.c_int64),
                ("checksum", C.c_uint64), ("seconds", C.c_double)]


def epoch_run(rowptr, col, x: Optional[np.ndarray], y: Optional[np.ndarray], idx, ranges: np.ndarray,
              sizes: Sequence[int], num_threads: int) -> EpochStats:
    """CPU baseline ("port"): one pass of the non-distributed worker loop over `ranges`."""
    rowptr, col, idx, sz = _i64(rowptr), _i64(col), _i64(idx), _i64(list(sizes))
    ranges = np.ascontiguousarray(ranges, dtype=np.int32)
    xb = None if x is None else np.ascontiguousarray(x)
    yb = None if y is None else _i64(y)
    st = EpochStats()
    row_bytes = 0 if xb is None else xb.shape[-1] * xb.dtype.itemsize
    rc = lib().orc_epoch_run(_ptr(rowptr), _ptr(col), _ptr(xb), row_bytes, _ptr(yb), _ptr(idx),
                             _ptr(ranges), ranges.shape[0], _ptr(sz), sz.shape[0], num_threads, C.byref(st))
    assert rc == 0
    return st


# ---- f1: VIP analytic model (numpy restatement; PARITY UNPINNED) -----------------------------------
def vip_frequencies(rowptr, col, train_idx, fanouts: Sequence[int], batch_size: int) -> np.ndarray:
    """driver/drivers/ddp.py:135-239 get_frequency_tensors_fast, float64, the Taylor form the
    reference ships: p0 = batch_size/|train| on the training ids (:154-157); per fanout (in list
    order, :187): res = segment_csr(min(1, fanout/deg)[col] * p[col], rowptr, 'add') (:222-224),
    p_next = 1 - exp(-res) (:226); total = 1 - prod(1 - p_h) (:231-235).  The reference's chunking
    of the CSR (:161-185) only bounds device memory and does not change the values.

    Parity unpinned: the reference function needs torch_scatter / torch_sparse and a partitioned
    OGB dataset object, neither of which exists in this image, so it cannot be run here; this
    restatement is checked against a hand-computed case (tests/test_oracle_golden.py)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    n = rowptr.shape[0] - 1
    deg = (rowptr[1:] - rowptr[:-1]).astype(np.float64)
    p = np.zeros(n, dtype=np.float64)
    tr = np.asarray(train_idx, dtype=np.int64)
    p[tr] = (batch_size * 1.0) / tr.shape[0]
    seg = np.repeat(np.arange(n, dtype=np.int64), rowptr[1:] - rowptr[:-1])
    total = np.ones(n, dtype=np.float64)
    for f in fanouts:
        with np.errstate(divide="ignore"):
            w = np.minimum(np.ones_like(deg), f / deg)
        weighted = w[col] * p[col]
        res = np.zeros(n, dtype=np.float64)
        np.add.at(res, seg, weighted)
        p = 1 - np.exp(-res)
        total = total * (1 - p)
    return 1 - total
