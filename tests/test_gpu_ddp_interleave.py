"""-m gpu: the training-loop shape of the distributed path -- collectives of the CALLER's process group
(DistributedDataParallel gradient all-reduces, driver/drivers/ddp.py:349-350 + fast_trainer/train.py:15-71)
interleaved with the feature exchanges of the native communicator, issued by the consumer
(SPP_EXCHANGE_ISSUE=consumer, include/spp.h spp_exchange_cfg.issue_on_consumer).

One GPU is all a test box has, so the two halves are covered separately:
  * a world-size-1 NCCL process group + the world-size-1 RCCL communicator derived from it: models.SAGE under
    DDP trains through DeviceDistributedPrefetcher, every x is compared with x_full[n_id];
  * two in-process ranks (threads) on the in-process transport, each issuing an all_reduce on the torch group
    between its next() calls -- the exchange of group g+1 is issued from mid-group g on both ranks while the
    caller's collectives run on another stream.
"""
import os
import socket
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

from test_gpu_native_exchange import SIZES, _check_batch, _graph, _rank_cfg  # noqa: E402


@pytest.fixture(scope="module")
def nccl_world1():
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        from salient_plusplus_amd import fast_sampler as fs
        fs.clear_resident_cache()
        dist.destroy_process_group()


@pytest.mark.parametrize("group_delivery", ["0", "1"])
def test_ddp_training_steps_between_consumer_issued_exchanges(nccl_world1, group_delivery, monkeypatch):
    """SAGE under DistributedDataParallel, fed by DeviceDistributedPrefetcher over the native exchange on the
    RCCL communicator derived from the process group: every backward issues gradient all-reduces between two
    exchanges, x stays bit-exact, the loss goes down."""
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
    from salient_plusplus_amd.models import SAGE
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", "consumer")
    monkeypatch.setenv("SPP_GROUP_DELIVERY", group_delivery)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    dev = torch.device("cuda", 0)
    nb, bs = 12, 16
    cfg, idx = _rank_cfg(g, 0, 1, [0, n], False, nb, bs, fs)
    ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
    n_classes = int(g["y"].max()) + 1
    torch.manual_seed(0)
    model = SAGE(g["x"].shape[1], 32, n_classes, 3).to(dev)
    ddp = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=True)
    opt = torch.optim.Adam(ddp.parameters(), lr=5e-3)
    losses = []
    for epoch in range(3):
        it = iter(FastSampler(2, 8, cfg))
        assert it.session.native_exchange
        pre = DeviceDistributedPrefetcher([dev], it, True)
        for k, (batch,) in enumerate(pre):
            if epoch == 0:
                _check_batch(batch, k, ranges, g, idx, g["x"], orc)
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.nll_loss(ddp(batch.x, batch.adjs), batch.y.reshape(-1))
            loss.backward()                    # gradient all-reduce on the torch group
            opt.step()
            losses.append(float(loss.detach()))
        pre.quiesce()
        it.session.close()
    assert np.isfinite(losses).all()
    assert np.mean(losses[-nb:]) < np.mean(losses[:nb])


@pytest.mark.parametrize("slots,group_delivery", [(4, "0"), (16, "0"), (16, "1")])
def test_caller_collectives_between_next_calls_two_in_process_ranks(nccl_world1, slots, group_delivery, monkeypatch):
    """Two ranks (threads, in-process transport), consumer-issued exchanges; between two next() calls every rank
    runs an all_reduce on the caller's NCCL group (as a DDP backward would) on its own stream."""
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler
    from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
    dist = nccl_world1
    monkeypatch.setenv("SPP_EXCHANGE_ISSUE", "consumer")
    monkeypatch.setenv("SPP_GROUP_DELIVERY", group_delivery)
    g = _graph()
    n = g["rowptr"].shape[0] - 1
    P, nb, bs = 2, 11, 8
    offsets = [0, 1400, n]
    comms = fs.NativeComm.local(P)
    errors = []
    pg_lock = threading.Lock()          # one process group object shared by the rank threads

    def run(rank):
        it = None
        try:
            torch.cuda.set_device(0)
            fs.set_native_comm(comms[rank])
            dev = torch.device("cuda", 0)
            cfg, idx = _rank_cfg(g, rank, P, offsets, True, nb, bs, fs)
            ranges = orc.batch_ranges(len(idx), bs, False, True, nb)
            it = iter(FastSampler(2, slots, cfg))
            assert it.session.native_exchange
            pre = DeviceDistributedPrefetcher([dev], it, True)
            grad = torch.ones(1 << 16, device=dev)
            side = torch.cuda.Stream(dev)
            held = []
            for k, (batch,) in enumerate(pre):
                held.append(batch)
                with torch.cuda.stream(side), pg_lock:
                    dist.all_reduce(grad)
            torch.cuda.synchronize()
            assert len(held) == nb
            for k, batch in enumerate(held):
                _check_batch(batch, k, ranges, g, idx, g["x"], orc)
            it.session.close()
        except BaseException as e:  # noqa: BLE001
            import traceback
            errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
            if it is not None:
                it.session.close()
            comms[rank].close()
        finally:
            fs.set_native_comm(None)

    ts = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(180)
    hung = [t for t in ts if t.is_alive()]
    for c in comms:
        c.close()
    assert not errors, "\n".join(errors)
    assert not hung, "rank thread hung"
