"""GPU: f1 VIP analytic model and cache construction (ddp.py:135-239, :417-570) against the numpy
oracle (parity unpinned for this row: the reference function needs torch_scatter)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _graph():
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    return {k: g[k] for k in g.files}


@pytest.mark.parametrize("fanouts,bs", [([15, 10, 5], 32), ([20, 20, 20], 1024), ([2], 7), ([1, 1, 1, 1], 3)])
def test_vip_frequencies_vs_oracle(fanouts, bs):
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_trainer.vip_cache import vip_frequencies
    g = _graph()
    T = torch.from_numpy
    train = g["idx"][:300]
    want = orc.vip_frequencies(g["rowptr"], g["col"], train, fanouts, bs)
    got = vip_frequencies(T(g["rowptr"]), T(g["col"]), T(train), fanouts, bs).cpu().numpy()
    # float64; only the summation order inside a row differs from the oracle's
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
    assert (got[train] >= 0).all() and got.max() <= 1.0


def test_vip_frequencies_products_scale_properties():
    """Full-size graph: probabilities in [0,1], monotone in the fanout, zero exactly where no
    training vertex is within reach (checked on isolated vertices), and equal to the oracle on a
    random sample of rows recomputed from the same inputs."""
    from salient_plusplus_amd.fast_trainer.vip_cache import vip_frequencies
    from salient_plusplus_amd.synthetic import make_workload
    wl = make_workload("S-products", device=torch.device("cuda", 0))
    tr = wl.train_idx[:50000]
    f1 = vip_frequencies(wl.rowptr, wl.col, tr, [15, 10, 5], 1024)
    f2 = vip_frequencies(wl.rowptr, wl.col, tr, [30, 20, 10], 1024)
    assert float(f1.min()) >= 0.0 and float(f1.max()) <= 1.0
    assert bool((f2 >= f1 - 1e-15).all())
    deg = wl.rowptr[1:] - wl.rowptr[:-1]
    assert bool((f1[deg == 0] == 0).all())
    # one-hop closed form on a sample of rows: p1[v] = 1 - exp(-sum_u min(1, f/deg u) * p0[u])
    one = vip_frequencies(wl.rowptr, wl.col, tr, [15], 1024)
    p0 = torch.zeros(wl.num_nodes, dtype=torch.float64, device=one.device)
    p0[tr] = 1024.0 / tr.numel()
    q = torch.minimum(torch.ones_like(p0), 15.0 / deg.to(torch.float64)) * p0
    rows = torch.randint(0, wl.num_nodes, (2000,), device=one.device)
    for v in rows[:200].tolist():
        s = q[wl.col[int(wl.rowptr[v]):int(wl.rowptr[v + 1])]].sum()
        assert abs(float(one[v]) - float(1 - torch.exp(-s))) <= 1e-12
    from salient_plusplus_amd import fast_sampler as fs
    fs.clear_resident_cache()


def _worker(rank, port, strategy, q):
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=2)
        real = dist.all_to_all_single

        def staged(output, input, output_split_sizes=None, input_split_sizes=None, group=None, async_op=False):
            o = torch.empty(output.shape, dtype=output.dtype)
            real(o, input.cpu(), output_split_sizes=output_split_sizes, input_split_sizes=input_split_sizes, group=group)
            output.copy_(o)
        dist.all_to_all_single = staged
        from oracle import oracle as orc
        from salient_plusplus_amd import fast_sampler as fs
        from salient_plusplus_amd.fast_trainer.vip_cache import create_vip_cache
        g = _graph()
        T = torch.from_numpy
        n = g["rowptr"].shape[0] - 1
        offsets = np.array([0, 1400, n], dtype=np.int64)
        lo, hi = int(offsets[rank]), int(offsets[rank + 1])
        pb = fs.RangePartitionBook(rank, 2, T(offsets))
        train = g["idx"][(len(g["idx"]) * rank) // 2:(len(g["idx"]) * (rank + 1)) // 2]
        fan, bs, pct = [15, 10, 5], 32, 20.0
        cache = create_vip_cache(pb, n, T(g["x"][lo:hi].copy()).cuda(), pct, strategy, rowptr=T(g["rowptr"]),
                                 col=T(g["col"]), train_idx=T(train), fanouts=fan, batch_size=bs)
        cv = cache.cached_vertices.cpu().numpy().reshape(-1)
        cf = cache.cached_features.cpu().numpy()
        k = int(n / 2 * pct / 100)
        ext = np.concatenate([np.arange(0, lo), np.arange(hi, n)])
        if strategy == "vip":
            freq = orc.vip_frequencies(g["rowptr"], g["col"], train, fan, bs)[ext]
            k = min(k, int(np.count_nonzero(freq)))
            thr = np.sort(freq)[::-1][k - 1]
            got_f = orc.vip_frequencies(g["rowptr"], g["col"], train, fan, bs)[cv]
            assert cv.shape[0] == k and (got_f >= thr * (1 - 1e-12)).all()      # the top-k set (ties aside)
        else:
            deg = (g["rowptr"][1:] - g["rowptr"][:-1])[ext]
            want = ext[np.argsort(deg, kind="stable")[:k]]
            np.testing.assert_array_equal(np.sort(cv), np.sort(want))
        assert ((cv < lo) | (cv >= hi)).all() and np.unique(cv).shape[0] == cv.shape[0]
        np.testing.assert_array_equal(cf.view(np.uint16), g["x"][cv].view(np.uint16))   # rows fetched from the owner
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(f"rank {rank}: {e}\n{traceback.format_exc()}")
        raise


@pytest.mark.parametrize("strategy", ["vip", "degree"])
def test_create_vip_cache_two_ranks(strategy):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29810 + (1 if strategy == "vip" else 0)
    procs = [ctx.Process(target=_worker, args=(r, port, strategy, q)) for r in range(2)]
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
