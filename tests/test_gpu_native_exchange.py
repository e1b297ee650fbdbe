"""GPU: the native feature exchange of the Session (include/spp.h e1-e3, session.hip) -- counts
all-gather, int32 id exchange, row serving, row exchange and the fused assembly -- with TWO ranks
on one GPU.  RCCL refuses two ranks on one device, so the ranks live in one process (one thread
each) on the in-process transport (spp_comm_create_local); everything above the transport's
send/recv primitives is the product path.  A world-size-1 RCCL communicator covers the RCCL
binding itself (dlopen, ncclCommInitRank, all-gather)."""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

SIZES = [15, 10, 5]


def _graph():
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    return {k: g[k] for k in g.files}


def _rank_cfg(g, rank, P, offsets, use_cache, nb, bs, fs):
    from salient_plusplus_amd.fast_trainer.samplers import FastSamplerConfig
    T = torch.from_numpy
    n = g["rowptr"].shape[0] - 1
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    x = g["x"]
    rng = np.random.default_rng(100 + rank)
    remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
    cache = fs.Cache()
    if use_cache:
        cv = np.sort(rng.choice(remote, size=250, replace=False)).astype(np.int64)
        cache = fs.Cache(rank, P, T(cv), T(x[cv].copy()))
    idx = g["idx"][(len(g["idx"]) * rank) // P:(len(g["idx"]) * (rank + 1)) // P]
    cut = (hi - lo) // 3
    cfg = FastSamplerConfig(
        x_cpu=T(x[lo:hi][cut:].copy()), x_gpu=T(x[lo:hi][:cut].copy()).cuda(), y=T(g["y"]).unsqueeze(-1),
        rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx), batch_size=bs, sizes=SIZES,
        skip_nonfull_batch=False, pin_memory=False, distributed=True,
        partition_book=fs.RangePartitionBook(rank, P, T(np.asarray(offsets, dtype=np.int64))), cache=cache,
        force_exact_num_batches=True, exact_num_batches=nb, count_remote_frequency=False, use_cache=use_cache)
    return cfg, idx


def _check_batch(batch, k, ranges, g, idx, x, orc):
    start, stop = int(ranges[k][0]), int(ranges[k][1])
    m = orc.sample_batch(g["rowptr"], g["col"], idx, start, stop, SIZES)
    assert batch.x.is_cuda
    np.testing.assert_array_equal(batch.x.cpu().numpy().view(np.uint16), x[m.n_id].view(np.uint16))
    np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g["y"][m.n_id[:stop - start]])
    for adj, hop in zip(batch.adjs, m.hops):
        rp, cl, _ = adj.adj_t.csr()
        np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
        np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)


def _run_rank(rank, P, comms, g, offsets, use_cache, nb, bs, slots, errors, stats):
    it = None
    try:
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        torch.cuda.set_device(0)
        fs.set_native_comm(comms[rank])
        cfg, idx = _rank_cfg(g, rank, P, offsets, use_cache, nb, bs, fs)
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        dev = torch.device("cuda", 0)
        x = g["x"]
        for epoch in range(2):          # the second epoch reuses the pooled sampler and grown buffers
            it = iter(FastSampler(2, slots, cfg))
            assert it.session.native_exchange
            pre = DeviceDistributedPrefetcher([dev], it, True)
            got = 0
            held = []
            for (batch,) in pre:
                held.append(batch)
                got += 1
                if got == 2:
                    pre.quiesce()       # every rank at the same batch: all in-flight exchanges complete
                if epoch == 0:          # epoch 1 compares after the epoch: no host sync between batches
                    _check_batch(held.pop(), got - 1, ranges, g, idx, x, orc)
            for k, batch in enumerate(held):
                _check_batch(batch, k, ranges, g, idx, x, orc)
            assert got == nb
            stats[rank] = pre.NUMBER_OF_SENT_BYTES
            it.session.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        comms[rank].close()     # wakes the peers out of the rendezvous
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.set_native_comm(None)


@pytest.mark.parametrize("P,use_cache,nb,bs,slots", [
    (2, False, 3, 32, 6),       # one group holds all batches
    (2, True, 7, 16, 4),        # several groups, 2-slot sets, ragged last group, VIP cache
    (3, True, 5, 24, 16),       # three ranks
    (2, True, 37, 4, 32),       # four slot-sets of 8 in flight, ragged tail
    (8, True, 9, 8, 32),        # the scaling bench's rank count
    (-2, False, 4, 16, 8),      # degenerate book: rank 0 owns every vertex, rank 1 none (all its rows are remote)
])
@pytest.mark.parametrize("issue", ["thread", "consumer"])
def test_native_exchange_in_process_ranks(P, use_cache, nb, bs, slots, issue, monkeypatch):
    from salient_plusplus_amd import fast_sampler as fs
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", issue)       # who issues the exchanges: session thread / consumer
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    if P == -2:
        P, offsets = 2, [0, n, n]
    else:
        offsets = {2: [0, 1400, n], 3: [0, 900, 2100, n]}.get(P) or [int(v) for v in np.linspace(0, n, P + 1)]
    comms = fs.NativeComm.local(P)
    errors, stats = [], {}
    ts = [threading.Thread(target=_run_rank, args=(r, P, comms, g, offsets, use_cache, nb, bs, slots, errors, stats))
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
    assert all(stats[r] > 0 for r in range(P))     # counts / ids / rows really travelled


@pytest.mark.parametrize("lag", ["0", "1", "2"])
def test_three_slot_sets_consumer_issued_exchange_under_every_refill_lag(lag, monkeypatch):
    """The launcher's window is `sets - refill_lag` groups beyond the finished ones (session.hip launch_window; spp.h
    spp_session_next) and a consumer-issued exchange needs the NEXT group's chain launched (issue_exchanges_up_to(g + 1)
    with three or more sets): with exactly three sets the two meet.  Every lag must deliver every batch bit for bit and
    terminate (a window that is one group short would wait for a chain nobody launches)."""
    from salient_plusplus_amd import fast_sampler as fs
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", "consumer")
    monkeypatch.setenv("SPP_REFILL_LAG", lag)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    P, offsets = 2, [0, 1400, n]
    comms = fs.NativeComm.local(P)
    errors, stats = [], {}
    # 24 slots = three slot-sets of 8; 41 batches of 4 seeds = six groups, the last one ragged
    ts = [threading.Thread(target=_run_rank, args=(r, P, comms, g, offsets, True, 41, 4, 24, errors, stats)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"


def _run_rank_counting(rank, P, comms, g, offsets, nb, bs, errors, out):
    it = None
    try:
        import dataclasses
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        torch.cuda.set_device(0)
        fs.set_native_comm(comms[rank])
        cfg, idx = _rank_cfg(g, rank, P, offsets, False, nb, bs, fs)
        cfg = dataclasses.replace(cfg, count_remote_frequency=True)
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        dev = torch.device("cuda", 0)
        it = iter(FastSampler(2, 8, cfg))
        assert it.session.native_exchange
        pre = DeviceDistributedPrefetcher([dev], it, True)
        # a backlog on the delivery stream: the counting must follow the delivery launches on it, not race them on the
        # consumer's stream
        with torch.cuda.stream(pre.side.stream):
            torch.cuda._sleep(200_000_000)
        got = sum(1 for _ in pre)
        assert got == nb
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        want = {}
        for k in range(nb):
            m = orc.sample_batch(g["rowptr"], g["col"], idx, int(ranges[k][0]), int(ranges[k][1]), SIZES)
            for v in m.n_id[(m.n_id < lo) | (m.n_id >= hi)]:
                want[int(v)] = want.get(int(v), 0) + 1
        st = it.get_distributed_stats()
        f, v = st.remote_frequency_tensor.numpy(), st.remote_vertices_ordered_by_freq.numpy()
        assert f.shape[0] == len(want) and (np.diff(f) <= 0).all()
        assert all(want[int(vv)] == int(ff) for vv, ff in zip(v, f))
        out[rank] = len(want)
        it.session.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        comms[rank].close()
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.set_native_comm(None)


def test_remote_frequency_counting_through_the_native_prefetcher():
    """count_remote_frequency (fast_sampler.cpp:1093-1103, :835-880) with the native exchange behind
    DeviceDistributedPrefetcher: the prefetcher requests batches without a stream context (the Session was told the
    delivery stream once), so the counting has to order itself behind the delivery launch that writes the ids."""
    from salient_plusplus_amd import fast_sampler as fs
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    P, offsets = 2, [0, 1400, n]
    comms = fs.NativeComm.local(P)
    errors, out = [], {}
    ts = [threading.Thread(target=_run_rank_counting, args=(r, P, comms, g, offsets, 5, 24, errors, out)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"
    assert all(out[r] > 0 for r in range(P))


@pytest.mark.parametrize("P,use_cache,nb,bs,slots", [(2, True, 7, 16, 4), (3, True, 5, 24, 16), (8, True, 9, 8, 32),
                                                     (2, True, 37, 4, 32)])
@pytest.mark.parametrize("issue", ["thread", "consumer"])
def test_native_exchange_with_group_delivery(P, use_cache, nb, bs, slots, issue, monkeypatch):
    """the same exchange with one delivery launch per GROUP (SPP_GROUP_DELIVERY=1): the look-ahead is taken at a fixed,
    blocking program point on every rank, the assembly of all of a group's batches runs in one launch"""
    monkeypatch.setenv("SPP_GROUP_DELIVERY", "1")
    test_native_exchange_in_process_ranks(P, use_cache, nb, bs, slots, issue, monkeypatch)


def test_rccl_world1_comm_and_session():
    """ncclCommInitRank through the late-bound RCCL, then a distributed Session on it (no peers:
    every row is local, but the counts all-gather and the assembly launch run)."""
    import ctypes as C
    from oracle import oracle as orc
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    L = nat.load()
    token = (C.c_uint8 * nat.SPP_COMM_ID_BYTES)()
    nat.check(L.spp_comm_unique_id(token))
    h = C.c_void_p()
    nat.check(L.spp_comm_create(token, 0, 1, 0, C.byref(h)))
    assert L.spp_comm_rank(h) == 0 and L.spp_comm_world(h) == 1
    comm = fs.NativeComm(h, 0, 1)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    try:
        fs.set_native_comm(comm)
        cfg, idx = _rank_cfg(g, 0, 1, [0, n], False, 4, 32, fs)
        ranges = orc.batch_ranges(len(idx), 32, False, True, 4)
        it = iter(FastSampler(2, 4, cfg))
        assert it.session.native_exchange
        for k, proto in enumerate(it):
            m = orc.sample_batch(g["rowptr"], g["col"], idx, int(ranges[k][0]), int(ranges[k][1]), SIZES)
            np.testing.assert_array_equal(proto.x.cpu().numpy().view(np.uint16), g["x"][m.n_id].view(np.uint16))
            np.testing.assert_array_equal(proto.n_id.cpu().numpy(), m.n_id)
        it.session.close()
    finally:
        fs.set_native_comm(None)
        comm.close()


def test_exchange_rejects_mismatched_communicator():
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    comms = fs.NativeComm.local(2)
    try:
        fs.set_native_comm(comms[0])
        # partition book says rank 1 of 2, the communicator is rank 0: the Session must not pick it up
        cfg, _ = _rank_cfg(g, 1, 2, [0, 1400, n], False, 2, 32, fs)
        it = iter(FastSampler(2, 4, cfg))
        assert not it.session.native_exchange
        it.session.close()
    finally:
        fs.set_native_comm(None)
        for c in comms:
            c.close()


def _products_rank(rank, P, comms, wl, nb, errors, out):
    it = None
    try:
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
        torch.cuda.set_device(0)
        fs.set_native_comm(comms[rank])
        N, F = wl.num_nodes, wl.x.size(1)
        offsets = torch.linspace(0, N, P + 1).long()
        offsets[-1] = N
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        deg = (wl.rowptr[1:] - wl.rowptr[:-1]).clone()
        deg[lo:hi] = -1
        cv = torch.topk(deg, 20000).indices.sort().values
        cache = fs.Cache(rank, P, cv, wl.x[cv].contiguous())
        bs = wl.batch_size
        idx = wl.train_idx[rank * nb * bs:(rank + 1) * nb * bs]
        cfg = FastSamplerConfig(
            x_cpu=torch.empty((0, F), dtype=wl.x.dtype), x_gpu=wl.x[lo:hi].contiguous(), y=wl.y.unsqueeze(-1),
            rowptr=wl.rowptr, col=wl.col, idx=idx, batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False,
            pin_memory=False, distributed=True, partition_book=fs.RangePartitionBook(rank, P, offsets), cache=cache,
            force_exact_num_batches=True, exact_num_batches=nb, count_remote_frequency=False, use_cache=True)
        it = iter(FastSampler(2, 16, cfg))
        assert it.session.native_exchange
        k = 0
        for proto in it:
            assert torch.equal(proto.x, wl.x[proto.n_id]), f"rank {rank} batch {k}: assembled rows differ"
            assert torch.equal(proto.sliced_cpu_labels.view(-1), wl.y[proto.n_id[:proto.sliced_cpu_labels.numel()]])
            k += 1
        assert k == nb
        out[rank] = it.session.exchange_bytes()
        it.session.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        comms[rank].close()
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.set_native_comm(None)


def test_native_exchange_products_scale_two_ranks():
    """Headline-size batches (~770k MFG nodes, ~330k remote rows each) through the native exchange:
    20 batches per rank = 2.5 groups of 8, buffers grown on the fly, 20k-row VIP cache."""
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.synthetic import make_workload
    wl = make_workload("S-products", device=torch.device("cuda", 0))
    P, nb = 2, 20
    comms = fs.NativeComm.local(P)
    errors, out = [], {}
    ts = [threading.Thread(target=_products_rank, args=(r, P, comms, wl, nb, errors, out)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"
    # what rank 0 sent in rows is what rank 1 received, and vice versa (ids + rows + counts)
    assert out[0][0] > 1e8 and out[1][0] > 1e8
    fs.clear_resident_cache()


def test_partitioned_dataset_from_disk_through_exchange(tmp_path):
    """f1 -> f2 -> exchange: VIP probabilities order the vertices inside each partition, the dataset
    is written in the reference's partitioned layout, every rank loads its own x<r>.pt and the
    native exchange reassembles exactly the rows of the reordered full table."""
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.dataset import DisjointPartFeatReorderedDataset as D, FastDataset
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.vip_cache import vip_frequencies
    g = _graph()
    T = torch.from_numpy
    n = g["rowptr"].shape[0] - 1
    P, bs, nb = 2, 16, 5
    rng = np.random.default_rng(5)
    split = {"train": T(g["idx"].copy()), "valid": T(np.arange(10)), "test": T(np.arange(10, 20))}
    ds = FastDataset.from_tensors("toy", T(g["x"].astype(np.float32)), T(g["y"]), T(g["rowptr"]), T(g["col"]), split, 8)
    labels = T(rng.integers(0, P, size=n))
    prob = vip_frequencies(ds.rowptr, ds.col, split["train"], SIZES, bs).cpu()
    root = D.reorder_and_save(ds, labels, prob, tmp_path).parent
    parts = [D.from_path(root, "toy", r) for r in range(P)]
    x_full = torch.cat([p.x for p in parts]).numpy()
    rowptr, col = parts[0].rowptr.numpy(), parts[0].col.numpy()
    comms = fs.NativeComm.local(P)
    errors = []

    def run(r):
        it = None
        try:
            torch.cuda.set_device(0)
            fs.set_native_comm(comms[r])
            d = parts[r]
            idx = d.split_idx_parts[r]["train"][:nb * bs]
            cfg = FastSamplerConfig(
                x_cpu=torch.empty((0, d.num_features), dtype=d.x.dtype), x_gpu=d.x.cuda(), y=d.y.unsqueeze(-1),
                rowptr=d.rowptr, col=d.col, idx=idx, batch_size=bs, sizes=SIZES, skip_nonfull_batch=False,
                pin_memory=False, distributed=True, partition_book=d.get_RangePartitionBook(), cache=fs.Cache(),
                force_exact_num_batches=True, exact_num_batches=nb, count_remote_frequency=False, use_cache=False)
            ranges = orc.batch_ranges(idx.numel(), bs, False, True, nb)
            it = iter(FastSampler(2, 8, cfg))
            assert it.session.native_exchange
            for k, proto in enumerate(it):
                m = orc.sample_batch(rowptr, col, idx.numpy(), int(ranges[k][0]), int(ranges[k][1]), SIZES)
                np.testing.assert_array_equal(proto.n_id.cpu().numpy(), m.n_id)
                np.testing.assert_array_equal(proto.x.cpu().numpy().view(np.uint16), x_full[m.n_id].view(np.uint16))
            it.session.close()
        except BaseException as e:  # noqa: BLE001
            import traceback
            errors.append(f"rank {r}: {e}\n{traceback.format_exc()}")
            if it is not None:
                it.session.close()
            comms[r].close()
        finally:
            fs.set_native_comm(None)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not any(t.is_alive() for t in ts)


def test_mismatched_batch_counts_are_refused_not_hung():
    """Ranks that would run different numbers of batches are told so at Session creation (the exchange
    is one collective sequence per group; the mismatch would otherwise be a hang)."""
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    comms = fs.NativeComm.local(2)
    out = {}

    def run(r):
        try:
            torch.cuda.set_device(0)
            fs.set_native_comm(comms[r])
            cfg, _ = _rank_cfg(g, r, 2, [0, 1400, n], False, 3 + r, 16, fs)      # 3 batches on rank 0, 4 on rank 1
            iter(FastSampler(2, 4, cfg))
            out[r] = "created"
        except nat.SppError as e:
            out[r] = str(e)
        finally:
            fs.set_native_comm(None)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
    for c in comms:
        c.close()
    assert not any(t.is_alive() for t in ts)
    assert all("same number of batches" in out[r] for r in range(2)), out


def test_native_exchange_on_the_generic_sampling_path(monkeypatch):
    """An all-neighbour hop forces one-batch groups (the generic path sizes launches on the host), so
    the exchange runs once per batch: same result."""
    import sys as _sys
    from salient_plusplus_amd import fast_sampler as fs
    monkeypatch.setattr(_sys.modules[__name__], "SIZES", [3, -1])
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    comms = fs.NativeComm.local(2)
    errors, stats = [], {}
    ts = [threading.Thread(target=_run_rank, args=(r, 2, comms, g, [0, 1400, n], True, 5, 8, 4, errors, stats))
          for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not any(t.is_alive() for t in ts)


def test_native_exchange_eight_ranks_headline_batches():
    """The 8-GPU composition on one GPU: eight in-process ranks at S-products scale with the planted
    8-block locality, batch 1024, federated seeds, analytic VIP cache of 10 % of N/P -- a third of every batch's ~750 k rows
    arrives from seven peers.  Three sampling groups per rank and epoch (the 23 batches its own training vertices give), two epochs with different seed
    orders (exchange buffers grown on the fly, then reused by the pooled sampler); every rank's x bit for bit
    x_full[n_id] (tools/exchange_p8.py does the work and is also what the profile of this load runs)."""
    import json
    import subprocess
    env = dict(os.environ, SPP_ALLOW_LOCAL_COMM="1", VERIFY="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exchange_p8.py"), "8", "24", "2"], env=env,
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("EXCHANGE_P8 ")]
    assert len(line) == 1, r.stdout[-2000:]
    d = json.loads(line[0][len("EXCHANGE_P8 "):])
    ranks = [ln for ln in r.stdout.splitlines() if ln.startswith("rank ")]
    assert len(ranks) == 8 and all("bit exact True" in ln for ln in ranks), "\n".join(ranks)
    assert d["P"] == 8 and d["batches_all_ranks"] >= 8 * 16 * 2 and d["batches_all_ranks"] % 16 == 0
    assert d["rows_served"] == d["rows_fetched"] and d["rows_local"] + d["rows_cache"] + d["rows_fetched"] == d["rows_delivered"]
    assert 0.15 < d["frac_fetched"] < 0.6 and d["frac_cache"] > 0.0        # a realistic remote fraction, cache in use
