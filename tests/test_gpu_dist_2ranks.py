"""-m gpu: TWO ranks sharing the one GPU of the box run the complete distributed data path
(real Session + HIP kernels + DeviceDistributedPrefetcher, features range-partitioned 2-way, VIP
cache on/off).  RCCL cannot place two ranks on one device, so in this test only the transport of
`all_to_all_single` is swapped for gloo with host staging; everything else is the product path.
Each rank checks that every assembled batch equals x_full[n_id] with the oracle's n_id."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 2
SIZES = [15, 10, 5]


def _staged_all_to_all_single(real):
    class _Done:
        def wait(self):
            return True

    def fn(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
        o = torch.empty(output.shape, dtype=output.dtype)
        real(o, input.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
        output.copy_(o)
        return _Done() if async_op else None
    return fn


def _worker(rank, port, use_cache, pipeline_on, q):
    try:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=P)
        dist.all_to_all_single = _staged_all_to_all_single(dist.all_to_all_single)
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
        g = {k: g[k] for k in g.files}
        T = torch.from_numpy
        n = g["rowptr"].shape[0] - 1
        offsets = np.array([0, 1400, n], dtype=np.int64)
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        x = g["x"]
        rng = np.random.default_rng(100 + rank)
        remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
        cv = np.sort(rng.choice(remote, size=250, replace=False)).astype(np.int64)
        cache = fs.Cache(rank, P, T(cv), T(x[cv].copy())) if use_cache else fs.Cache()
        idx = g["idx"][(len(g["idx"]) * rank) // P:(len(g["idx"]) * (rank + 1)) // P]
        nb = 3
        cfg = FastSamplerConfig(
            x_cpu=T(x[lo:hi][200:].copy()), x_gpu=T(x[lo:hi][:200].copy()).cuda(), y=T(g["y"]).unsqueeze(-1),
            rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx), batch_size=32, sizes=SIZES,
            skip_nonfull_batch=False, pin_memory=False, distributed=True,
            partition_book=fs.RangePartitionBook(rank, P, T(offsets)), cache=cache, force_exact_num_batches=True,
            exact_num_batches=nb, count_remote_frequency=False, use_cache=use_cache)
        ranges = orc.batch_ranges(len(idx), 32, False, True, nb)
        dev = torch.device("cuda", 0)
        got = 0
        for (batch,) in DeviceDistributedPrefetcher([dev], iter(FastSampler(2, 6, cfg)), pipeline_on):
            start, stop = int(ranges[got][0]), int(ranges[got][1])
            m = orc.sample_batch(g["rowptr"], g["col"], idx, start, stop, SIZES)
            assert batch.x.is_cuda
            np.testing.assert_array_equal(batch.x.cpu().numpy().view(np.uint16), x[m.n_id].view(np.uint16))
            np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g["y"][m.n_id[:stop - start]])
            for adj, hop in zip(batch.adjs, m.hops):
                rp, cl, _ = adj.adj_t.csr()
                np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
            got += 1
        assert got == nb
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(f"rank {rank}: {e}\n{traceback.format_exc()}")
        raise


@pytest.mark.parametrize("use_cache,pipeline_on", [(False, True), (True, True), (True, False)])
def test_two_ranks_one_gpu_full_path(use_cache, pipeline_on):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29700 + 7 * int(use_cache) + 3 * int(pipeline_on)
    procs = [ctx.Process(target=_worker, args=(r, port, use_cache, pipeline_on, q)) for r in range(P)]
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
