"""-m gpu: the drop-in path (fast_sampler module + fast_trainer facade) against the compiled
reference's golden outputs and the oracle.  Reads like a test of the reference's own API."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
T = torch.from_numpy


@pytest.fixture(scope="module")
def fs():
    from salient_plusplus_amd import _native as nat
    nat.load()
    nat.require_device()
    from salient_plusplus_amd import fast_sampler
    return fast_sampler


def make_cfg(fs, g, sizes, batch_size, x, y, idx, **kw):
    from salient_plusplus_amd.fast_trainer.samplers import FastSamplerConfig
    d = dict(x_cpu=T(x), x_gpu=torch.empty(0), y=T(y).unsqueeze(-1) if y is not None else None,
             rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx), batch_size=batch_size, sizes=list(sizes),
             skip_nonfull_batch=False, pin_memory=False, distributed=False, partition_book=None,
             cache=fs.Cache(), force_exact_num_batches=False, exact_num_batches=0,
             count_remote_frequency=False, use_cache=False)
    d.update(kw)
    return FastSamplerConfig(**d)


@pytest.mark.parametrize("case", ["s15_10_5", "s20_20_20", "sall", "s25_15", "s3_all", "s1", "s0_2"])
def test_session_batches_match_reference(fs, graph_a, golden_dir, case):
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    g = np.load(os.path.join(golden_dir, f"mfg_a_{case}.npz"))
    sizes = [int(s) for s in g["sizes"]]
    n = graph_a["rowptr"].shape[0] - 1
    ids_as_x = np.arange(n, dtype=np.int64).reshape(n, 1)
    cfg = make_cfg(fs, graph_a, sizes, 64, ids_as_x, graph_a["y"], graph_a["idx"])
    sampler = FastSampler(2, 8, cfg)
    assert len(sampler) == int(g["num_batches"])
    nb = 0
    for b, batch in enumerate(iter(sampler)):
        assert batch.x.is_cuda
        assert (batch.idx_range.start, batch.idx_range.stop) == tuple(int(v) for v in g[f"b{b}_range"])
        np.testing.assert_array_equal(batch.x.cpu().numpy().reshape(-1), g[f"b{b}_n_id"])
        np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g[f"b{b}_y"])
        assert len(batch.adjs) == len(sizes)
        for h, adj in enumerate(batch.adjs):
            rp, cl, val = adj.adj_t.csr()
            np.testing.assert_array_equal(rp.cpu().numpy(), g[f"b{b}_h{h}_rowptr"])
            np.testing.assert_array_equal(cl.cpu().numpy(), g[f"b{b}_h{h}_col"])
            T_, S_ = (int(v) for v in g[f"b{b}_h{h}_size"])
            assert tuple(adj.size) == (S_, T_)                  # Adj.size = sparse_sizes[::-1]
            assert tuple(adj.adj_t.sparse_sizes()) == (T_, S_)
            assert adj.e_id.numel() == 0 and adj.e_id.dtype == torch.int64
        nb += 1
    assert nb == int(g["num_batches"])


def test_fp16_feature_slice_matches_reference(fs, graph_a, golden_dir):
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    g = np.load(os.path.join(golden_dir, "slice_a_s15_10_5.npz"))
    cfg = make_cfg(fs, graph_a, [15, 10, 5], 64, graph_a["x"], graph_a["y"], graph_a["idx"])
    dev = torch.device("cuda", torch.cuda.current_device())
    nb = 0
    for b, (batch,) in enumerate(DevicePrefetcher([dev], iter(FastSampler(3, 2, cfg)))):
        assert batch.x.dtype == torch.float16
        # tolerance stated by the north star: 1e-6; rows are copies, so the bits must match
        np.testing.assert_array_equal(batch.x.cpu().numpy().view(np.uint16), g[f"b{b}_x"].view(np.uint16))
        np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g[f"b{b}_y"].reshape(-1))
        nb += 1
    assert nb == 4


def test_batch_range_tables_match_reference(fs, graph_a, golden_dir):
    g = np.load(os.path.join(golden_dir, "batch_ranges.npz"))
    n = graph_a["rowptr"].shape[0] - 1
    for key in g.files:
        parts = key.split("_")
        kw = {}
        if parts[0] == "exact":
            nn, bs = int(parts[1]), 17
            kw = dict(force_exact_num_batches=True, exact_num_batches=int(parts[2]))
        else:
            nn, bs = int(parts[1]), int(parts[2])
            kw = dict(skip_nonfull_batch=bool(int(parts[3])))
        idx = np.arange(nn, dtype=np.int64) % n
        cfg = make_cfg(fs, graph_a, [1], bs, np.zeros((n, 1), dtype=np.int64), None, idx, **kw)
        s = fs.Session(2, 4, cfg.to_fast_sampler())
        assert s.num_total_batches == cfg.get_num_batches()
        got = []
        while True:
            b = s.blocking_get_batch()
            if b is None:
                break
            assert b[1] is None
            got.append(b[3])
        s.close()
        np.testing.assert_array_equal(np.array(got, dtype=np.int64).reshape(-1, 2), g[key].reshape(-1, 2))


def test_free_functions_follow_the_thread_generator(fs, graph_a, golden_dir):
    g = np.load(os.path.join(golden_dir, "free_functions.npz"))
    from salient_plusplus_amd.fast_sampler import _gen
    _gen.seed, _gen.pos = 5489, 0
    rp, col, idx = T(graph_a["rowptr"]), T(graph_a["col"]), T(g["idx"])
    a = fs.sample_adj(rp, col, idx, 5, False)
    assert a[2].dtype == torch.int32
    np.testing.assert_array_equal(a[0].cpu().numpy(), g["call0_rowptr"])
    np.testing.assert_array_equal(a[1].cpu().numpy(), g["call0_col"])
    np.testing.assert_array_equal(a[2].cpu().numpy().astype(np.int64), g["call0_n_id"])
    b = fs.sample_adj(rp, col, idx, 4, True)
    np.testing.assert_array_equal(b[0].cpu().numpy(), g["call1_rowptr"])
    np.testing.assert_array_equal(b[1].cpu().numpy(), g["call1_col"])
    np.testing.assert_array_equal(b[2].cpu().numpy().astype(np.int64), g["call1_n_id"])
    n_id, adjs = fs.multilayer_sample(idx, [3, 2], rp, col)
    assert n_id.dtype == torch.int64
    np.testing.assert_array_equal(n_id.cpu().numpy(), g["call2_n_id"])
    for h, (r_, c_, e_, sz) in enumerate(adjs):
        np.testing.assert_array_equal(r_.cpu().numpy(), g[f"call2_h{h}_rowptr"])
        np.testing.assert_array_equal(c_.cpu().numpy(), g[f"call2_h{h}_col"])
        assert tuple(sz) == tuple(int(v) for v in g[f"call2_h{h}_size"])
    c = fs.sample_adj(rp, col, idx, -1, False)
    np.testing.assert_array_equal(c[0].cpu().numpy(), g["call3_rowptr"])
    np.testing.assert_array_equal(c[1].cpu().numpy(), g["call3_col"])
    np.testing.assert_array_equal(c[2].cpu().numpy().astype(np.int64), g["call3_n_id"])


def test_serial_index_to_row_major_partition_book_cache(fs, graph_a, golden_dir):
    g = np.load(os.path.join(golden_dir, "serial_index.npz"))
    sel = T(g["sel"])
    x = T(graph_a["x"])
    np.testing.assert_array_equal(fs.serial_index(x, sel).cpu().numpy().view(np.uint16), g["half_all"].view(np.uint16))
    np.testing.assert_array_equal(fs.serial_index(x, sel, 7).cpu().numpy().view(np.uint16), g["half_n7"].view(np.uint16))
    np.testing.assert_array_equal(fs.serial_index(T(g["xf"]), sel).cpu().numpy(), g["float_all"])
    np.testing.assert_array_equal(fs.serial_index(T(graph_a["y"]).unsqueeze(-1), sel, 3).cpu().numpy(), g["long_n3"])
    with pytest.raises(RuntimeError, match="2D row-major"):
        fs.serial_index(x.t(), sel)
    cm = torch.arange(12, dtype=torch.float32).reshape(4, 3).t()
    np.testing.assert_array_equal(fs.to_row_major(cm).numpy(), g["trm_out"])
    rm = torch.arange(12, dtype=torch.float32).reshape(3, 4)
    assert fs.to_row_major(rm) is rm
    with pytest.raises(RuntimeError, match="2D"):
        fs.to_row_major(torch.zeros(3))

    p = np.load(os.path.join(golden_dir, "partition_book.npz"))
    pb = fs.RangePartitionBook(2, 4, T(p["offsets"]))
    nids = T(p["nids"])
    np.testing.assert_array_equal(pb.nid2partid(nids.cuda()).cpu().numpy(), p["partid"])
    np.testing.assert_array_equal(pb.nid2partid(nids).numpy(), p["partid"])
    np.testing.assert_array_equal(pb.nid2localnid(nids, 2).numpy(), p["localnid_p2"])
    np.testing.assert_array_equal(pb.partid2nids(1).numpy(), p["partid2nids_1"])
    cache = fs.Cache(2, 4, T(p["cached_vertices"]), torch.zeros((6, 4), dtype=torch.float16))
    np.testing.assert_array_equal(cache.nid_is_cached(T(p["probe"])).numpy(), p["is_cached"])
    empty = fs.Cache()
    assert empty.cached_vertices.numel() == 0 and empty.cached_features.dtype == torch.float16


@pytest.mark.parametrize("sizes", [[15, 10, 5], [3, -1], [40, 2]])      # fast path, all-neighbour hop, fanout > 32
@pytest.mark.parametrize("P,rank,use_cache", [(2, 0, False), (2, 1, True), (4, 0, True), (4, 1, False), (4, 3, True)])
def test_distributed_proto_batch_vs_oracle(fs, graph_a, P, rank, use_cache, sizes):
    """Worker distributed branch (fast_sampler.cpp:1017-1262) on the GPU vs the oracle's restatement."""
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    layouts = {2: np.array([0, 1400, 3000], dtype=np.int64), 4: np.array([0, 700, 1500, 2100, 3000], dtype=np.int64)}
    offs = layouts[P]
    n = graph_a["rowptr"].shape[0] - 1
    x = graph_a["x"]
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    rng = np.random.default_rng(P * 10 + rank)
    if use_cache:
        remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
        cv = rng.choice(remote, size=300, replace=False).astype(np.int64)
        cache = fs.Cache(rank, P, T(cv), T(x[cv].copy()))
        ocache = orc.Cache(cv, n)
    else:
        cv, cache, ocache = None, fs.Cache(), None
    G = 100
    cfg = make_cfg(fs, graph_a, sizes, 64, x[lo:hi][G:].copy(), graph_a["y"], graph_a["idx"],
                   x_gpu=T(x[lo:hi][:G].copy()).cuda(), distributed=True,
                   partition_book=fs.RangePartitionBook(rank, P, T(offs)), cache=cache,
                   force_exact_num_batches=True, exact_num_batches=3, use_cache=use_cache)
    ranges = orc.batch_ranges(200, 64, False, True, 3)
    nb = 0
    for b, proto in enumerate(iter(FastSampler(2, 4, cfg))):
        start, stop = int(ranges[b][0]), int(ranges[b][1])
        assert (proto.idx_range.start, proto.idx_range.stop) == (start, stop)
        m = orc.sample_batch(graph_a["rowptr"], graph_a["col"], graph_a["idx"], start, stop, sizes)
        want = orc.partition_batch(m.n_id, offs, rank, ocache, 0)
        for k in range(P):
            np.testing.assert_array_equal(proto.partition_nids[k].cpu().numpy(), want.partition_nids[k])
        np.testing.assert_array_equal(proto.cached_nids.cpu().numpy(), want.cached_nids)
        np.testing.assert_array_equal(proto.perm_partition_to_mfg.cpu().numpy(), want.perm_partition_to_mfg)
        np.testing.assert_array_equal(proto.n_id.cpu().numpy(), m.n_id)
        np.testing.assert_array_equal(proto.sliced_cpu_labels.cpu().numpy().reshape(-1), graph_a["y"][m.n_id[:stop - start]])
        assert proto.sliced_cpu_features.shape[0] == 0          # every local row is HBM resident
        for h, adj in enumerate(proto.adjs):
            rp, cl, _ = adj.adj_t.csr()
            np.testing.assert_array_equal(rp.cpu().numpy(), m.hops[h].rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), m.hops[h].col)
        nb += 1
    assert nb == 3


def test_sampler_goes_back_to_the_pool_at_end_of_data(fs, graph_a):
    """An exhausted iterator that is still referenced (the usual `it = iter(sampler)` of the next epoch, a
    StopIteration being handled) must not keep the pooled sampler: the next epoch's Session reuses it instead
    of building a second workspace."""
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    cfg = make_cfg(fs, graph_a, [15, 10, 5], 64, graph_a["x"], graph_a["y"], graph_a["idx"])
    smp = FastSampler(2, 4, cfg)
    it1 = iter(smp)
    h1 = it1.session._pool_entry[0].value
    n1 = sum(1 for _ in it1)
    assert n1 > 0 and it1.session._pool_entry is None            # released at end of data, object still alive
    stats = it1.get_stats()
    assert stats.total_blocked_occasions >= 0                     # counters survive the native session
    it2 = iter(smp)                                               # it1 still bound
    assert it2.session._pool_entry[0].value == h1
    assert sum(1 for _ in it2) == n1
    assert it1.session.num_consumed_batches == n1 and next(it1, None) is None


def test_cache_map_rebuilt_in_place_between_sessions(fs, graph_a):
    """The bucketing kernels test cache membership through a bitmap derived from the cache map; the pooled
    sampler keeps it across Sessions, so it is rebuilt at every Session start.  Two Sessions over the SAME map
    buffer, its contents rewritten in between (a cache rebuilt in place): the second must bucket by the new
    contents."""
    from oracle import oracle as orc
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    import ctypes as C
    P, rank, sizes = 4, 1, [15, 10, 5]
    offs = np.array([0, 700, 1500, 2100, 3000], dtype=np.int64)
    n = graph_a["rowptr"].shape[0] - 1
    x = graph_a["x"]
    lo, hi = int(offs[rank]), int(offs[rank + 1])
    remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
    rng = np.random.default_rng(77)
    cv_a = np.sort(rng.choice(remote, size=300, replace=False)).astype(np.int64)
    cv_b = np.sort(rng.choice(remote, size=300, replace=False)).astype(np.int64)
    cv_b[-1] = cv_a.max()                                   # same map length (max id + 1)
    cv_b = np.unique(cv_b)
    cache = fs.Cache(rank, P, T(cv_a), T(x[cv_a].copy()))
    shared_map = cache.device_map()                         # the buffer both Sessions will see

    def run(cv):
        ocache = orc.Cache(cv, n)
        cfg = make_cfg(fs, graph_a, sizes, 64, x[lo:hi][100:].copy(), graph_a["y"], graph_a["idx"],
                       x_gpu=T(x[lo:hi][:100].copy()).cuda(), distributed=True,
                       partition_book=fs.RangePartitionBook(rank, P, T(offs)), cache=cache,
                       force_exact_num_batches=True, exact_num_batches=3, use_cache=True)
        ranges = orc.batch_ranges(200, 64, False, True, 3)
        for b, proto in enumerate(iter(FastSampler(2, 4, cfg))):
            start, stop = int(ranges[b][0]), int(ranges[b][1])
            m = orc.sample_batch(graph_a["rowptr"], graph_a["col"], graph_a["idx"], start, stop, sizes)
            want = orc.partition_batch(m.n_id, offs, rank, ocache, 0)
            for k in range(P):
                np.testing.assert_array_equal(proto.partition_nids[k].cpu().numpy(), want.partition_nids[k])
            np.testing.assert_array_equal(proto.cached_nids.cpu().numpy(), want.cached_nids)
            np.testing.assert_array_equal(proto.perm_partition_to_mfg.cpu().numpy(), want.perm_partition_to_mfg)

    run(cv_a)
    # rewrite the SAME device buffer with cache B's map
    L = nat.load()
    cvb_dev = T(cv_b).cuda()
    nat.check(L.spp_cache_build_map(C.c_void_p(cvb_dev.data_ptr()), cvb_dev.numel(), C.c_void_p(shared_map.data_ptr()),
                                    shared_map.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    cache.cached_vertices = T(cv_b)
    cache.cached_features = T(x[cv_b].copy())
    cache._vertices_dev = cvb_dev
    cache._features_dev = None
    run(cv_b)


def test_distributed_prefetcher_world_size_1_rccl(fs, graph_a):
    """The RCCL exchange path end to end with one rank (all_to_all_single with itself): the assembled
    x must equal x[n_id] (transferers.py:479-484 identity)."""
    import torch.distributed as dist
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        n = graph_a["rowptr"].shape[0] - 1
        x = graph_a["x"]
        offs = np.array([0, n], dtype=np.int64)
        cfg = make_cfg(fs, graph_a, [15, 10, 5], 64, np.zeros((0, x.shape[1]), dtype=np.float16), graph_a["y"],
                       graph_a["idx"], x_gpu=T(x).cuda(), distributed=True,
                       partition_book=fs.RangePartitionBook(0, 1, T(offs)), cache=fs.Cache(),
                       force_exact_num_batches=True, exact_num_batches=3)
        dev = torch.device("cuda", torch.cuda.current_device())
        ranges = orc.batch_ranges(200, 64, False, True, 3)
        for pipeline_on in (True, False):
            nb = 0
            for b, (batch,) in enumerate(DeviceDistributedPrefetcher([dev], iter(FastSampler(2, 4, cfg)), pipeline_on)):
                start, stop = int(ranges[b][0]), int(ranges[b][1])
                m = orc.sample_batch(graph_a["rowptr"], graph_a["col"], graph_a["idx"], start, stop, [15, 10, 5])
                np.testing.assert_array_equal(batch.x.cpu().numpy().view(np.uint16), x[m.n_id].view(np.uint16))
                np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), graph_a["y"][m.n_id[:stop - start]])
                nb += 1
            assert nb == 3
    finally:
        dist.destroy_process_group()


def test_distributed_proto_batch_vs_reference_golden(fs, graph_a, golden_dir):
    """Against the compiled reference's own distributed batches (tests/golden/distributed.npz,
    generated on the GPU box): ownership buckets, cache indices, permutation, labels, node order."""
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    g = np.load(os.path.join(golden_dir, "distributed.npz"))
    x = graph_a["x"]
    n_checked = 0
    for ci in range(int(g["num_cfgs"])):
        tag = f"c{ci}"
        P, rank, G, use_cache = (int(v) for v in g[f"{tag}_meta"])
        if G != 100:            # G only changes which local rows the reference slices on the host
            continue
        offs = g[f"{tag}_offsets"]
        lo, hi = int(offs[rank]), int(offs[rank + 1])
        cv = g[f"{tag}_cached_vertices"]
        cache = fs.Cache(rank, P, T(cv), T(x[cv].copy())) if use_cache else fs.Cache()
        cfg = make_cfg(fs, graph_a, [15, 10, 5], 64, x[lo:hi][G:].copy(), graph_a["y"], graph_a["idx"],
                       x_gpu=T(x[lo:hi][:G].copy()).cuda(), distributed=True,
                       partition_book=fs.RangePartitionBook(rank, P, T(offs)), cache=cache,
                       force_exact_num_batches=True, exact_num_batches=3, use_cache=bool(use_cache))
        for b, proto in enumerate(iter(FastSampler(2, 4, cfg))):
            assert (proto.idx_range.start, proto.idx_range.stop) == tuple(int(v) for v in g[f"{tag}_b{b}_range"])
            for m in range(P):
                np.testing.assert_array_equal(proto.partition_nids[m].cpu().numpy(), g[f"{tag}_b{b}_part{m}"])
            np.testing.assert_array_equal(proto.cached_nids.cpu().numpy(), g[f"{tag}_b{b}_cached_nids"])
            np.testing.assert_array_equal(proto.perm_partition_to_mfg.cpu().numpy(), g[f"{tag}_b{b}_perm"])
            np.testing.assert_array_equal(proto.sliced_cpu_labels.cpu().numpy(), g[f"{tag}_b{b}_labels"])
            np.testing.assert_array_equal(proto.n_id.cpu().numpy(), g[f"{tag}_b{b}_n_id"])
            n_checked += 1
    assert n_checked == 8 * 3
