"""-m gpu: parity of the raw C-ABI kernels (libspp_hip.so) against the oracle on seeded inputs."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def lib():
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    nat.require_device()
    return L


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check(L, rc):
    assert rc >= 0, L.spp_last_error().decode()


@pytest.mark.parametrize("seed,skip,n", [(5489, 0, 10000), (5, 0, 227), (64 * 17 + 5, 100, 5000),
                                          (1024 * 17 + 5, 0, 1), (0xFFFFFFFF, 623, 1300)])
def test_mt19937(lib, seed, skip, n):
    from oracle import oracle as orc
    out = torch.zeros(n, dtype=torch.int32, device="cuda")
    check(lib, lib.spp_mt19937_fill(seed, skip, n, P(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy().view(np.uint32), orc.mt19937(seed, n, skip))


@pytest.mark.parametrize("dtype,F", [(np.float16, 100), (np.float16, 128), (np.float16, 7), (np.float32, 5),
                                      (np.int64, 1), (np.uint8, 3), (np.float16, 768)])
@pytest.mark.parametrize("idx_bytes", [8, 4])
def test_gather_rows(lib, dtype, F, idx_bytes):
    from oracle import oracle as orc
    rng = np.random.default_rng(F)
    n_src, n_idx = 5000, 3001
    src = (rng.standard_normal((n_src, F)) * 100).astype(dtype)
    idx = rng.integers(0, n_src, size=n_idx)
    want = orc.serial_index(src, idx)
    d_src = dev(src.view(np.uint8).reshape(n_src, -1))
    d_idx = dev(idx.astype(np.int64 if idx_bytes == 8 else np.int32))
    row_bytes = F * src.dtype.itemsize
    out = torch.zeros((n_idx, row_bytes), dtype=torch.uint8, device="cuda")
    check(lib, lib.spp_gather_rows(P(d_src), n_src, row_bytes, P(d_idx), idx_bytes, n_idx, n_idx, P(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want.view(np.uint8).reshape(n_idx, -1))
    # `n` limit (labels): only the first n rows are written
    out2 = torch.full((10, row_bytes), 0xAB, dtype=torch.uint8, device="cuda")
    check(lib, lib.spp_gather_rows(P(d_src), n_src, row_bytes, P(d_idx), idx_bytes, n_idx, 7, P(out2), None))
    torch.cuda.synchronize()
    got = out2.cpu().numpy()
    np.testing.assert_array_equal(got[:7], want.view(np.uint8).reshape(n_idx, -1)[:7])
    assert (got[7:] == 0xAB).all()


@pytest.mark.parametrize("row_bytes,stride", [(200, 256), (200, 200), (14, 16), (344, 384), (6, 10), (1536, 1664)])
def test_gather_rows_strided_source(lib, row_bytes, stride):
    """Padded resident feature table: rows `stride` bytes apart, dense output; padding bytes are never
    copied (they hold a poison value)."""
    from oracle import oracle as orc
    rng = np.random.default_rng(row_bytes)
    n_src, n_idx = 4000, 2777
    table = np.full((n_src, stride), 0xEE, dtype=np.uint8)
    table[:, :row_bytes] = rng.integers(0, 256, size=(n_src, row_bytes), dtype=np.uint8)
    idx = rng.integers(0, n_src, size=n_idx).astype(np.int64)
    want = orc.serial_index(np.ascontiguousarray(table[:, :row_bytes]), idx)
    d_src, d_idx = dev(table), dev(idx)
    out = torch.zeros((n_idx, row_bytes), dtype=torch.uint8, device="cuda")
    check(lib, lib.spp_gather_rows_strided(P(d_src), n_src, row_bytes, stride, P(d_idx), 8, n_idx, n_idx, P(out), None))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    # a stride smaller than the row is refused
    assert lib.spp_gather_rows_strided(P(d_src), n_src, row_bytes, row_bytes - 1, P(d_idx), 8, n_idx, n_idx, P(out),
                                       None) != 0


@pytest.mark.parametrize("row_bytes,stride", [(200, 256), (200, 208), (8, 16), (24, 32), (408, 512), (504, 512), (88, 128)])
@pytest.mark.parametrize("n_idx", [1, 2, 63, 64, 65, 255, 1023, 4097, 70001])
def test_gather_rows_span_form(lib, row_bytes, stride, n_idx):
    """Rows of 16k + 8 bytes out of a 16-byte-strided table take the span form of the row gather (16-byte loads, pieces
    regrouped across lanes, aligned 16-byte stores: gather_body.hip.h kVecSpan) -- the same bytes as the 8-byte form, on whole
    and partial last iterations, and nothing outside the destination rows.  serial_index, fast_sampler.cpp:238-259."""
    from oracle import oracle as orc
    rng = np.random.default_rng(row_bytes * 131 + n_idx)
    n_src = 3000
    table = np.full((n_src, stride), 0xEE, dtype=np.uint8)
    table[:, :row_bytes] = rng.integers(0, 256, size=(n_src, row_bytes), dtype=np.uint8)
    idx = rng.integers(0, n_src, size=n_idx).astype(np.int32)
    idx[-1] = n_src - 1                       # the table's last row: its padding is the last thing the allocation holds
    want = orc.serial_index(np.ascontiguousarray(table[:, :row_bytes]), idx.astype(np.int64))
    d_src, d_idx = dev(table), dev(idx)
    assert d_src.data_ptr() % 16 == 0
    outs = []
    for span in (1, 0):
        prev = lib.spp_tune(b"gather_span", span)
        try:
            raw = torch.full((n_idx * row_bytes + 64,), 0xAB, dtype=torch.uint8, device="cuda")
            check(lib, lib.spp_gather_rows_strided(P(d_src), n_src, row_bytes, stride, P(d_idx), 4, n_idx, n_idx, P(raw), None))
            torch.cuda.synchronize()
        finally:
            lib.spp_tune(b"gather_span", prev)
        got = raw.cpu().numpy()
        np.testing.assert_array_equal(got[:n_idx * row_bytes].reshape(n_idx, row_bytes), want)
        assert (got[n_idx * row_bytes:] == 0xAB).all()
        outs.append(got)
    np.testing.assert_array_equal(outs[0], outs[1])


def test_to_row_major(lib):
    from oracle import oracle as orc
    rng = np.random.default_rng(0)
    for (tr, tc, dt) in [(3, 4, np.float32), (100, 37, np.float16), (65, 33, np.int64)]:
        storage = rng.integers(-1000, 1000, size=tr * tc).astype(dt)
        want = orc.to_row_major(storage, tr, tc)
        out = torch.zeros(tr * tc * storage.dtype.itemsize, dtype=torch.uint8, device="cuda")
        check(lib, lib.spp_to_row_major(P(dev(storage.view(np.uint8))), tr, tc, storage.dtype.itemsize, P(out), None))
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy().view(dt).reshape(tr, tc), want)


class GpuSampler:
    def __init__(self, lib, rowptr, col, sizes, max_batch, slots=2):
        from salient_plusplus_amd import _native as nat
        self.L, self.nat = lib, nat
        self.rowptr, self.col = dev(rowptr), dev(col)
        cfg = nat.SamplerCfg()
        cfg.rowptr_dev, cfg.col_dev = self.rowptr.data_ptr(), self.col.data_ptr()
        cfg.num_nodes, cfg.nnz = rowptr.shape[0] - 1, col.shape[0]
        cfg.num_hops = len(sizes)
        for i, s in enumerate(sizes):
            cfg.sizes[i] = s
        cfg.max_batch, cfg.num_slots, cfg.device = max_batch, slots, 0
        self.h = C.c_void_p()
        check(lib, lib.spp_sampler_create(C.byref(cfg), C.byref(self.h)))
        self.H = len(sizes)

    def close(self):
        self.L.spp_sampler_destroy(self.h)

    def sample(self, seeds, rng_seed, skip=0, slot=0):
        L, nat = self.L, self.nat
        d_seeds = dev(np.asarray(seeds, dtype=np.int64))
        check(L, L.spp_sampler_sample(self.h, slot, P(d_seeds), d_seeds.numel(), rng_seed, skip, None))
        cnt = nat.MfgCounts()
        check(L, L.spp_sampler_wait(self.h, slot, C.byref(cnt)))
        out = nat.MfgOut()
        n_id = torch.empty(cnt.num_nodes, dtype=torch.int64, device="cuda")
        out.n_id = n_id.data_ptr()
        hops = []
        for k in range(self.H):
            rp = torch.empty(cnt.T[k] + 1, dtype=torch.int64, device="cuda")
            cl = torch.empty(cnt.E[k], dtype=torch.int64, device="cuda")
            out.rowptr[k], out.col[k] = rp.data_ptr(), cl.data_ptr()
            hops.append((rp, cl, (cnt.T[k], cnt.S[k])))
        check(L, L.spp_sampler_export(self.h, slot, C.byref(out), None))
        torch.cuda.synchronize()
        return n_id.cpu().numpy(), [(r.cpu().numpy(), c.cpu().numpy(), s) for r, c, s in hops], cnt


def assert_mfg_equal(got, want):
    n_id, hops, cnt = got
    np.testing.assert_array_equal(n_id, want.n_id)
    assert len(hops) == len(want.hops)
    for (rp, cl, size), w in zip(hops, want.hops):
        assert tuple(size) == tuple(w.size)
        np.testing.assert_array_equal(rp, w.rowptr)
        np.testing.assert_array_equal(cl, w.col)
    assert cnt.draws == want.draws


@pytest.mark.parametrize("sizes", [[15, 10, 5], [20, 20, 20], [25, 15], [1], [0, 2], [5], [32, 3]])
def test_sampler_fast_path_golden_graph(lib, graph_a, sizes):
    from oracle import oracle as orc
    s = GpuSampler(lib, graph_a["rowptr"], graph_a["col"], sizes, 64)
    try:
        idx = graph_a["idx"]
        for (start, stop) in [(0, 64), (64, 128), (128, 192), (192, 200)]:
            want = orc.sample_batch(graph_a["rowptr"], graph_a["col"], idx, start, stop, sizes)
            got = s.sample(idx[start:stop], orc.batch_seed(stop), slot=(start // 64) % 2)
            assert_mfg_equal(got, want)
    finally:
        s.close()


def test_standalone_sampler_with_cached_partition(lib, graph_a):
    """spp_sampler_sample / wait / export with ownership bucketing and a cache, WITHOUT a Session: nobody built
    the cache membership bits, the bucketing kernels fall back to the map itself."""
    from oracle import oracle as orc
    from salient_plusplus_amd import _native as nat
    rowptr, col, idx = graph_a["rowptr"], graph_a["col"], graph_a["idx"]
    n = rowptr.shape[0] - 1
    Pn, rank, sizes = 4, 2, [15, 10, 5]
    offs = np.array([0, 700, 1500, 2100, n], dtype=np.int64)
    rng = np.random.default_rng(5)
    remote = np.setdiff1d(np.arange(n), np.arange(offs[rank], offs[rank + 1]))
    cv = np.sort(rng.choice(remote, size=250, replace=False)).astype(np.int64)
    cmap = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    d_cv = dev(cv)
    check(lib, lib.spp_cache_build_map(P(d_cv), d_cv.numel(), P(cmap), n, None))
    torch.cuda.synchronize()
    d_rowptr, d_col = dev(rowptr), dev(col)
    cfg = nat.SamplerCfg()
    cfg.rowptr_dev, cfg.col_dev = d_rowptr.data_ptr(), d_col.data_ptr()
    cfg.num_nodes, cfg.nnz, cfg.num_hops = n, col.shape[0], len(sizes)
    for i, f in enumerate(sizes):
        cfg.sizes[i] = f
    cfg.max_batch, cfg.num_slots, cfg.device = 64, 2, 0
    cfg.part.num_parts, cfg.part.rank = Pn, rank
    for k in range(Pn + 1):
        cfg.part.offsets[k] = int(offs[k])
    cfg.part.use_cache, cfg.part.cache_map_dev, cfg.part.cache_map_len = 1, cmap.data_ptr(), n
    h = C.c_void_p()
    check(lib, lib.spp_sampler_create(C.byref(cfg), C.byref(h)))
    try:
        ocache = orc.Cache(cv, n)
        for (start, stop) in [(0, 64), (128, 192)]:
            d_seeds = dev(np.asarray(idx[start:stop], dtype=np.int64))
            check(lib, lib.spp_sampler_sample(h, 0, P(d_seeds), d_seeds.numel(), orc.batch_seed(stop), 0, None))
            cnt = nat.MfgCounts()
            check(lib, lib.spp_sampler_wait(h, 0, C.byref(cnt)))
            m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
            want = orc.partition_batch(m.n_id, offs, rank, ocache, 0)
            pc = [int(cnt.part_counts[k]) for k in range(Pn + 1)]
            assert pc[:Pn] == [len(w) for w in want.partition_nids] and pc[Pn] == len(want.cached_nids)
            out = nat.MfgOut()
            U = int(cnt.num_nodes)
            n_id = torch.empty(U, dtype=torch.int64, device="cuda")
            parts = torch.empty(max(1, sum(pc[:Pn])), dtype=torch.int64, device="cuda")
            cached = torch.empty(max(1, pc[Pn]), dtype=torch.int64, device="cuda")
            perm = torch.empty(U, dtype=torch.int64, device="cuda")
            keep = []
            out.n_id, out.parts, out.cached, out.perm = n_id.data_ptr(), parts.data_ptr(), cached.data_ptr(), perm.data_ptr()
            for k in range(len(sizes)):
                rp = torch.empty(cnt.T[k] + 1, dtype=torch.int64, device="cuda")
                cl = torch.empty(max(1, cnt.E[k]), dtype=torch.int64, device="cuda")
                out.rowptr[k], out.col[k] = rp.data_ptr(), cl.data_ptr()
                keep += [rp, cl]
            check(lib, lib.spp_sampler_export(h, 0, C.byref(out), None))
            torch.cuda.synchronize()
            np.testing.assert_array_equal(n_id.cpu().numpy(), m.n_id)
            np.testing.assert_array_equal(parts.cpu().numpy()[:sum(pc[:Pn])], np.concatenate(want.partition_nids))
            np.testing.assert_array_equal(cached.cpu().numpy()[:pc[Pn]], want.cached_nids)
            np.testing.assert_array_equal(perm.cpu().numpy(), want.perm_partition_to_mfg)
    finally:
        lib.spp_sampler_destroy(h)


@pytest.mark.parametrize("sizes", [[-1], [3, -1], [-1, 2], [40, 2], [33]])
def test_sampler_generic_path(lib, graph_a, sizes):
    from oracle import oracle as orc
    s = GpuSampler(lib, graph_a["rowptr"], graph_a["col"], sizes, 64)
    try:
        idx = graph_a["idx"]
        for (start, stop) in [(0, 64), (192, 200)]:
            want = orc.sample_batch(graph_a["rowptr"], graph_a["col"], idx, start, stop, sizes)
            got = s.sample(idx[start:stop], orc.batch_seed(stop))
            assert_mfg_equal(got, want)
    finally:
        s.close()


def test_sampler_against_reference_golden(lib, graph_a, golden_dir):
    """Straight against the compiled reference's outputs (not via the oracle)."""
    g = np.load(os.path.join(golden_dir, "mfg_a_s15_10_5.npz"))
    s = GpuSampler(lib, graph_a["rowptr"], graph_a["col"], [15, 10, 5], 64)
    try:
        for b in range(int(g["num_batches"])):
            start, stop = (int(v) for v in g[f"b{b}_range"])
            n_id, hops, cnt = s.sample(graph_a["idx"][start:stop], (stop * 17 + 5) & 0xFFFFFFFF)
            np.testing.assert_array_equal(n_id, g[f"b{b}_n_id"])
            for h, (rp, cl, size) in enumerate(hops):
                np.testing.assert_array_equal(rp, g[f"b{b}_h{h}_rowptr"])
                np.testing.assert_array_equal(cl, g[f"b{b}_h{h}_col"])
                assert tuple(size) == tuple(int(v) for v in g[f"b{b}_h{h}_size"])
    finally:
        s.close()


def test_sampler_noncoalesced_and_rng_skip(lib, golden_dir):
    from oracle import oracle as orc
    g = np.load(os.path.join(golden_dir, "mfg_b_s4_3_2.npz"))
    s = GpuSampler(lib, g["rowptr"], g["col"], [4, 3, 2], 32)
    try:
        for b in range(int(g["num_batches"])):
            start, stop = (int(v) for v in g[f"b{b}_range"])
            n_id, hops, _ = s.sample(g["idx"][start:stop], orc.batch_seed(stop))
            np.testing.assert_array_equal(n_id, g[f"b{b}_n_id"])
            for h, (rp, cl, _sz) in enumerate(hops):
                np.testing.assert_array_equal(rp, g[f"b{b}_h{h}_rowptr"])
                np.testing.assert_array_equal(cl, g[f"b{b}_h{h}_col"])
        # a generator that has already produced `skip` outputs (free-function semantics)
        rng = orc.MT.seeded(5489)
        first = orc.multilayer_sample(g["rowptr"], g["col"], g["idx"][:20], [4, 3, 2], rng)
        second = orc.multilayer_sample(g["rowptr"], g["col"], g["idx"][20:50], [4, 3, 2], rng)
        got = s.sample(g["idx"][20:50], 5489, skip=first.draws)
        assert_mfg_equal(got, second)
    finally:
        s.close()


def test_sampler_larger_random_graph(lib):
    """Bigger than one workgroup per kernel: multi-block scans, hash collisions, hubs."""
    from oracle import oracle as orc
    rng = np.random.default_rng(7)
    n = 60000
    deg = np.minimum(rng.zipf(1.6, n), 4000).astype(np.int64)
    deg[rng.random(n) < 0.05] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, size=rowptr[-1]).astype(np.int64)
    idx = rng.permutation(n)[:3000].astype(np.int64)
    s = GpuSampler(lib, rowptr, col, [15, 10, 5], 1024)
    try:
        for (start, stop) in [(0, 1024), (1024, 2048), (2048, 3000)]:
            want = orc.sample_batch(rowptr, col, idx, start, stop, [15, 10, 5])
            got = s.sample(idx[start:stop], orc.batch_seed(stop), slot=(start // 1024) % 2)
            assert_mfg_equal(got, want)
            assert want.num_edges > 50000
    finally:
        s.close()
