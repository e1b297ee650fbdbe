"""-m gpu: failure and ordering behaviour of the Session runtime (round-1 advisor findings).

* device-resident inputs still being produced on the caller's stream when the Session is created;
* a rank that never arrives: creation / next() raise after SPP_EXCHANGE_TIMEOUT_S and close() returns;
* one slot-set with consumer-issued exchanges (the look-ahead used to deadlock);
* row indices outside their table are reported (spp_async_errors), not silently clamped."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

SIZES = [15, 10, 5]


def test_session_is_ordered_after_the_producer_of_device_inputs(graph_a):
    """The seed ids are written by a copy queued BEHIND a long kernel on the caller's stream; the
    sampler's own (non-blocking) streams must not read them earlier."""
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    g = graph_a
    T = torch.from_numpy
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    idx_host = g["idx"][:160].astype(np.int64)
    real = T(idx_host).to(dev)
    x_d, y_d = T(g["x"]).to(dev), T(g["y"]).unsqueeze(-1).to(dev)
    for epoch in range(2):                      # the second Session reuses the pooled sampler (no allocation syncs at all)
        stale = torch.zeros_like(real)          # what an unordered sampler would see
        torch.cuda.synchronize()
        torch.cuda._sleep(int(6e8))             # ~0.3 s of queue backlog on the caller's stream
        stale.copy_(real)                       # "Shuffler.get_idx()": queued behind the backlog
        cfg = FastSamplerConfig(
            x_cpu=x_d, x_gpu=torch.empty(0), y=y_d, rowptr=T(g["rowptr"]), col=T(g["col"]), idx=stale,
            batch_size=32, sizes=SIZES, skip_nonfull_batch=False, pin_memory=False, distributed=False,
            partition_book=None, cache=fs.Cache(), force_exact_num_batches=False, exact_num_batches=0,
            count_remote_frequency=False, use_cache=False)
        nb = 0
        for (batch,) in DevicePrefetcher([dev], iter(FastSampler(2, 4, cfg))):
            start, stop = batch.idx_range.start, batch.idx_range.stop
            want = orc.sample_batch(g["rowptr"], g["col"], idx_host, start, stop, SIZES)
            np.testing.assert_array_equal(batch.x.cpu().numpy().view(np.uint16), g["x"][want.n_id].view(np.uint16))
            for adj, hop in zip(batch.adjs, want.hops):
                rp, cl, _ = adj.adj_t.csr()
                np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
            nb += 1
        assert nb == 5


def test_missing_rank_raises_and_close_returns(monkeypatch):
    """Two in-process ranks, only rank 0 ever creates a Session: after SPP_EXCHANGE_TIMEOUT_S the
    creation raises with a diagnostic (instead of hanging) and nothing blocks afterwards."""
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from test_gpu_native_exchange import _graph, _rank_cfg
    monkeypatch.setenv("SPP_EXCHANGE_TIMEOUT_S", "2")
    torch.cuda.set_device(0)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    comms = fs.NativeComm.local(2)
    out = {}

    def rank0():
        try:
            fs.set_native_comm(comms[0])
            cfg, _idx = _rank_cfg(g, 0, 2, [0, 1400, n], False, 4, 16, fs)
            t0 = time.time()
            try:
                it = iter(FastSampler(2, 8, cfg))
                next(it)
                out["err"] = None
                it.session.close()
            except RuntimeError as e:
                out["err"] = str(e)
            out["dt"] = time.time() - t0
        finally:
            fs.set_native_comm(None)
    t = threading.Thread(target=rank0)
    t.start()
    t.join(60)
    alive = t.is_alive()
    for c in comms:
        c.close()
    assert not alive, "the rank without a peer hung instead of raising"
    assert out.get("err"), "no error was raised for the missing rank"
    assert "waited" in out["err"] and "peers" in out["err"], out["err"]
    assert out["dt"] < 30
    # the process (and the pooled sampler) are still usable afterwards
    torch.cuda.synchronize()


def test_single_slot_set_with_consumer_issued_exchanges(monkeypatch):
    """SPP_GROUP_SIZE=8 with 8 slots gives ONE slot-set: the consumer must not try to issue the next
    group's exchange from the middle of the current one (nothing could ever launch that group)."""
    from salient_plusplus_amd import fast_sampler as fs
    from test_gpu_native_exchange import _graph, _run_rank
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", "consumer")
    monkeypatch.setenv("SPP_GROUP_SIZE", "8")
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    P, nb, bs, slots = 2, 19, 8, 8
    comms = fs.NativeComm.local(P)
    errors, stats = [], {}
    ts = [threading.Thread(target=_run_rank, args=(r, P, comms, g, [0, 1400, n], True, nb, bs, slots, errors, stats))
          for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not hung, "deadlock with a single slot-set"
    assert not errors, "\n".join(errors)


def test_gather_index_out_of_range_is_reported():
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    src = torch.arange(100 * 16, dtype=torch.float16, device=dev).view(100, 16)
    idx = torch.tensor([3, 99, 100, -1, 7], dtype=torch.int64, device=dev)      # 100 and -1 are outside
    out = torch.empty((5, 16), dtype=torch.float16, device=dev)
    assert L.spp_async_errors(0, 1) >= 0                                       # clear
    P = lambda t: C.c_void_p(t.data_ptr())                                      # noqa: E731
    nat.check(L.spp_gather_rows(P(src), 100, 32, P(idx), 8, 5, 5, P(out), None))
    torch.cuda.synchronize()
    bits = L.spp_async_errors(0, 1)
    assert bits & 1, f"out-of-range gather index not reported (mask {bits})"
    np.testing.assert_array_equal(out[[0, 1, 4]].cpu().numpy(), src[[3, 99, 7]].cpu().numpy())   # valid rows unaffected
    # and an in-range gather raises nothing
    nat.check(L.spp_gather_rows(P(src), 100, 32, P(idx[:2]), 8, 2, 2, P(out), None))
    torch.cuda.synchronize()
    assert L.spp_async_errors(0, 1) == 0


def test_peer_requesting_rows_it_should_not_is_reported(monkeypatch):
    """The ranks disagree on the partition book (rank 0 believes rank 1 owns [1000, n), rank 1 holds
    rows from 1400 on): rank 1 is asked for rows it does not own.  Served rows used to be clamped
    silently; now the exchange fails with a message on the serving rank."""
    from salient_plusplus_amd import _native as nat
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from test_gpu_native_exchange import _graph, _rank_cfg
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", "thread")
    monkeypatch.setenv("SPP_EXCHANGE_TIMEOUT_S", "20")
    L = nat.load()
    torch.cuda.set_device(0)
    L.spp_async_errors(0, 1)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    books = {0: [0, 1000, n], 1: [0, 1400, n]}
    comms = fs.NativeComm.local(2)
    raised, errors = {}, []

    def run(rank):
        it = None
        try:
            fs.set_native_comm(comms[rank])
            cfg, _idx = _rank_cfg(g, rank, 2, books[rank], False, 6, 16, fs)
            it = iter(FastSampler(2, 8, cfg))
            try:
                for _ in it:
                    torch.cuda.synchronize()
            except RuntimeError as e:
                raised[rank] = str(e)
        except BaseException as e:  # noqa: BLE001
            errors.append(f"rank {rank}: {e!r}")
        finally:
            if it is not None:
                it.session.close()
            comms[rank].close()
            fs.set_native_comm(None)
    ts = [threading.Thread(target=run, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert not [t for t in ts if t.is_alive()], "hung"
    assert not errors, errors
    assert any("does not own" in m or "outside its table" in m for m in raised.values()), raised
    L.spp_async_errors(0, 1)
