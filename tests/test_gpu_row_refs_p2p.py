"""-m gpu: the two opt-in forms of the partitioned path's feature delivery.

ROW REFERENCES (Session.row_refs, spp_mfg_out.row_addr): the delivery writes where every row of the batch lives -- this
rank's partition, the VIP cache, the rows received for the batch (copied contiguously into x_remote), or a peer's
partition -- instead of assembling ``x = cat(features_gather + [cached])[perm]`` (transferers.py:472-486), and
``models.SAGE`` aggregates its first layer from the addresses (driver/models.py:41-50).  Checked: RowRefs.materialize()
is bit-equal to the assembled matrix (= x_full[n_id] with the oracle's n_id), the MFG is the oracle's, and the model's
output over RowRefs is bit-equal to its output over the materialised matrix.

P2P TRANSPORT (SPP_DIST_TRANSPORT=p2p, spp_exchange_cfg.peer_x_dev): no id exchange, no serve gather, no send / receive
buffers -- the delivery reads remote rows in their owners' partitions.  In-process ranks (plain device pointers) for
P in {2, 3, 8} with and without the cache, and TWO PROCESSES on the one GPU of the box whose partitions reach each
other through hipIpcGetMemHandle / hipIpcOpenMemHandle (spp_ipc_export / spp_ipc_open) -- the mapping path an 8-GPU node
uses, minus the xGMI hop."""
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
T = torch.from_numpy


def _graph(F, seed=5):
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    g = {k: g[k] for k in g.files}
    if F != g["x"].shape[1]:
        n = g["rowptr"].shape[0] - 1
        g["x"] = np.random.default_rng(seed).standard_normal((n, F)).astype(np.float16)
    return g


def _rank_cfg(g, rank, P, offsets, use_cache, nb, bs, fs, split):
    from salient_plusplus_amd.fast_trainer.samplers import FastSamplerConfig
    n = g["rowptr"].shape[0] - 1
    lo, hi = int(offsets[rank]), int(offsets[rank + 1])
    x = g["x"]
    rng = np.random.default_rng(100 + rank)
    remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
    cache = fs.Cache()
    if use_cache:
        cv = np.sort(rng.choice(remote, size=min(250, len(remote)), replace=False)).astype(np.int64)
        cache = fs.Cache(rank, P, T(cv), T(x[cv].copy()))
    idx = g["idx"][(len(g["idx"]) * rank) // P:(len(g["idx"]) * (rank + 1)) // P]
    cut = (hi - lo) // 3 if split else hi - lo
    x_gpu = T(x[lo:hi][:cut].copy()).cuda()
    cfg = FastSamplerConfig(
        x_cpu=T(x[lo:hi][cut:].copy()), x_gpu=x_gpu, y=T(g["y"]).unsqueeze(-1),
        rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx), batch_size=bs, sizes=SIZES,
        skip_nonfull_batch=False, pin_memory=False, distributed=True,
        partition_book=fs.RangePartitionBook(rank, P, T(np.asarray(offsets, dtype=np.int64))), cache=cache,
        force_exact_num_batches=True, exact_num_batches=nb, count_remote_frequency=False, use_cache=use_cache)
    return cfg, idx


def _check(batch, k, ranges, g, idx, orc, fs, refs, model):
    start, stop = int(ranges[k][0]), int(ranges[k][1])
    m = orc.sample_batch(g["rowptr"], g["col"], idx, start, stop, SIZES)
    x = batch.x
    if refs:
        assert isinstance(x, fs.RowRefs) and x.is_cuda and tuple(x.shape) == (len(m.n_id), g["x"].shape[1])
        np.testing.assert_array_equal(x.n_id.cpu().numpy(), m.n_id)
        xm = x.materialize()
    else:
        assert isinstance(x, torch.Tensor)
        xm = x
    np.testing.assert_array_equal(xm.cpu().numpy().view(np.uint16), g["x"][m.n_id].view(np.uint16))
    np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g["y"][m.n_id[:stop - start]])
    for adj, hop in zip(batch.adjs, m.hops):
        rp, cl, _ = adj.adj_t.csr()
        np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
        np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
    if refs and model is not None:
        # (1) this repository's kernels, bit for bit: the first layer's operand [mean_j x_j | x_target] from the addresses
        # equals the one from the materialised matrix (same rows, same summation order)
        import ctypes as C
        from salient_plusplus_amd import _native as nat
        L = nat.load()
        rp, cl, _ = batch.adjs[0].adj_t.csr()
        Tn, F = rp.numel() - 1, xm.size(1)
        A = [torch.full((Tn, 2 * F), float("nan"), device=xm.device) for _ in range(2)]
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        p = lambda t: C.c_void_p(t.data_ptr())                                                  # noqa: E731
        nat.check(L.spp_sage_operand_forward_rows(p(rp), p(cl), Tn, p(x.addr), int(xm.dtype == torch.float16), F, p(A[0]), 2 * F, st))
        nat.check(L.spp_sage_operand_forward(p(rp), p(cl), Tn, p(xm), int(xm.dtype == torch.float16), F, F, p(A[1]), 2 * F, st))
        assert torch.equal(A[0], A[1]) and not bool(A[0].isnan().any())
        # (2) the model over RowRefs against the model over the materialised matrix: the same operand through the same
        # library GEMMs (compared with a tolerance: the library may pick another algorithm on a first call)
        with torch.no_grad():
            a = model(x, batch.adjs)
            b = model(xm, batch.adjs)
        assert a.shape == (stop - start, 5) and torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def _run_rank(rank, P, transport, comms, tables, g, offsets, use_cache, nb, bs, slots, refs, errors):
    it = None
    try:
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        from salient_plusplus_amd.models import SAGE
        torch.cuda.set_device(0)
        if transport == "p2p":
            fs.set_p2p_peers(tables)
        else:
            fs.set_native_comm(comms[rank])
        cfg, idx = _rank_cfg(g, rank, P, offsets, use_cache, nb, bs, fs, split=transport != "p2p")
        if transport == "p2p":
            cfg.x_gpu = tables["x_gpu"][rank]             # the tensor whose resident copy the peers were given
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        dev = torch.device("cuda", 0)
        F = g["x"].shape[1]
        model = None
        if refs and F % 4 == 0:
            torch.manual_seed(7)
            model = SAGE(F, 8, 5, len(SIZES)).to(dev).eval()
        for epoch in range(2):
            it = iter(FastSampler(2, slots, cfg, row_refs=refs))
            assert it.session.native_exchange and it.session.p2p == (transport == "p2p")
            pre = DeviceDistributedPrefetcher([dev], it, True)
            got, held = 0, []
            for (batch,) in pre:
                held.append(batch)
                got += 1
                if epoch == 0:
                    _check(held.pop(), got - 1, ranges, g, idx, orc, fs, refs, model)
            for k, batch in enumerate(held):              # epoch 1: every batch held until the epoch is over
                _check(batch, k, ranges, g, idx, orc, fs, refs, model)
            assert got == nb
            it.session.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        if comms:
            comms[rank].close()
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.set_native_comm(None)
        fs.set_p2p_peers(None)


def _run_ranks(P, transport, use_cache, nb, bs, slots, refs, F, monkeypatch, issue="consumer"):
    from salient_plusplus_amd import fast_sampler as fs
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", issue)
    monkeypatch.setenv("SPP_DIST_TRANSPORT", "p2p" if transport == "p2p" else "rccl")
    g = _graph(F)
    n = g["rowptr"].shape[0] - 1
    if P == -2:
        P, offsets = 2, [0, n, n]
    else:
        offsets = {2: [0, 1400, n], 3: [0, 900, 2100, n]}.get(P) or [int(v) for v in np.linspace(0, n, P + 1)]
    comms, tables = None, None
    if transport == "p2p":
        xg = [T(g["x"][int(offsets[r]):int(offsets[r + 1])].copy()).cuda() for r in range(P)]
        res = [fs._resident.get_rows(t) if t.numel() else None for t in xg]     # what the ranks' Sessions keep resident
        tables = _Peers(res, xg)
    else:
        comms = fs.NativeComm.local(P)
    errors = []
    ts = [threading.Thread(target=_run_rank, args=(r, P, transport, comms, tables, g, offsets, use_cache, nb, bs, slots, refs, errors))
          for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(240)
    hung = [t for t in ts if t.is_alive()]
    for c in comms or []:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"
    fs.clear_resident_cache()


class _Peers(list):
    """the ranks' resident partitions (what set_p2p_peers takes) + the tensors they are the resident copies of"""

    def __init__(self, resident, x_gpu):
        super().__init__(resident)
        self._x_gpu = x_gpu

    def __getitem__(self, k):
        return {"x_gpu": self._x_gpu}[k] if isinstance(k, str) else super().__getitem__(k)


@pytest.mark.parametrize("P,use_cache,nb,bs,slots,F", [
    (2, False, 3, 32, 6, 16),       # one group holds all batches
    (2, True, 7, 16, 4, 100),       # several groups, ragged last group, cache, 200-byte rows out of 256-byte strides
    (3, True, 5, 24, 16, 16),
    (8, True, 9, 8, 32, 128),       # the scaling bench's rank count and row width
    (8, False, 9, 8, 32, 7),        # 14-byte rows: byte-wise copies, materialize() only
    (-2, False, 4, 16, 8, 16),      # rank 1 owns nothing: every one of its rows is remote
])
@pytest.mark.parametrize("issue", ["thread", "consumer"])
def test_row_refs_over_the_exchange(P, use_cache, nb, bs, slots, F, issue, monkeypatch):
    _run_ranks(P, "local", use_cache, nb, bs, slots, True, F, monkeypatch, issue)


@pytest.mark.parametrize("refs", [False, True])
@pytest.mark.parametrize("P,use_cache,nb,bs,slots,F", [
    (2, False, 3, 32, 6, 16),
    (2, True, 7, 16, 4, 100),
    (3, True, 37, 4, 32, 7),
    (8, True, 9, 8, 32, 128),
    (8, False, 9, 8, 32, 100),
    (-2, False, 4, 16, 8, 16),
])
def test_p2p_transport_in_process_ranks(P, use_cache, nb, bs, slots, F, refs, monkeypatch):
    _run_ranks(P, "p2p", use_cache, nb, bs, slots, refs, F, monkeypatch)


def test_p2p_partition_of_one_row_follows_the_resident_stride(monkeypatch):
    """A rank that owns ONE vertex: its table has no stride of its own, and 400-byte rows are kept 512 bytes apart in every
    resident table -- the peers' stride must be that rule's, not the row length (found by the random suite in round 6)."""
    from salient_plusplus_amd import fast_sampler as fs
    monkeypatch.setenv("SPP_DIST_TRANSPORT", "p2p")
    g = _graph(200)
    n = g["rowptr"].shape[0] - 1
    offsets = [0, 1, n]
    xg = [T(g["x"][offsets[r]:offsets[r + 1]].copy()).cuda() for r in range(2)]
    tables = _Peers([fs._resident.get_rows(t) for t in xg], xg)
    errors = []
    ts = [threading.Thread(target=_run_rank, args=(r, 2, "p2p", None, tables, g, offsets, False, 3, 16, 8, r == 1, errors)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(240)
    assert not errors, "\n".join(errors)
    fs.clear_resident_cache()


def test_group_delivery_with_row_refs(monkeypatch):
    """one delivery launch per sampling group (SPP_GROUP_DELIVERY=1): the same references, the same copies"""
    monkeypatch.setenv("SPP_GROUP_DELIVERY", "1")
    _run_ranks(2, "local", True, 21, 8, 16, True, 16, monkeypatch)
    _run_ranks(3, "p2p", True, 21, 8, 16, True, 16, monkeypatch)


# ---- two PROCESSES on one GPU: the partitions reach each other through HIP IPC ------------------------------------
def _ipc_worker(rank, port, use_cache, refs, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ["SPP_DIST_TRANSPORT"] = "p2p"
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        g = _graph(100)
        n = g["rowptr"].shape[0] - 1
        offsets = [0, 1400, n]
        nb, bs = 5, 16
        cfg, idx = _rank_cfg(g, rank, 2, offsets, use_cache, nb, bs, fs, split=False)
        ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
        dev = torch.device("cuda", 0)
        it = iter(FastSampler(2, 8, cfg, row_refs=refs))          # collective: the peers' tables are mapped here
        assert it.session.native_exchange and it.session.p2p
        peers = it.session._peers
        assert len(peers._opened) == 1                            # the other process's partition, through an IPC handle
        got = 0
        for (batch,) in DeviceDistributedPrefetcher([dev], it, True):
            _check(batch, got, ranges, g, idx, orc, fs, refs, None)
            got += 1
        assert got == nb
        torch.cuda.synchronize()
        dist.barrier()                                            # nobody unmaps while a peer may still read
        peers.close()
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(f"rank {rank}: {e}\n{traceback.format_exc()}")
        raise


@pytest.mark.parametrize("use_cache,refs", [(False, False), (True, True)])
def test_p2p_transport_two_processes_through_hip_ipc(use_cache, refs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29760 + 5 * int(use_cache)
    procs = [ctx.Process(target=_ipc_worker, args=(r, port, use_cache, refs, q)) for r in range(2)]
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
