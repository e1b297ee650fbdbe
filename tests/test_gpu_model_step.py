"""GPU: f3 mean aggregation kernels and the SAGE fallback against a plain PyTorch fp32 reference of
the same op (index_add formulation).  Tolerances: forward 1e-5 relative (fp32 sums in a different
order), backward 1e-4 (atomic accumulation)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu


def _ref_mean(x, rowptr, col, T):
    cnt = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(T, device=x.device), cnt)
    out = torch.zeros((T, x.size(1)), dtype=torch.float32, device=x.device).index_add_(0, row, x.float()[col])
    return out / cnt.clamp(min=1).unsqueeze(-1).float()


def _random_hop(T, S, maxdeg, seed):
    g = torch.Generator().manual_seed(seed)
    deg = torch.randint(0, maxdeg + 1, (T,), generator=g)
    deg[::7] = 0                                             # empty rows
    rowptr = torch.zeros(T + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(deg, 0)
    col = torch.randint(0, S, (int(rowptr[-1]),), generator=g)
    return rowptr.cuda(), col.cuda()


@pytest.mark.parametrize("F,dtype", [(100, torch.float16), (256, torch.float32), (47, torch.float32),
                                     (3, torch.float16), (1024, torch.float32), (128, torch.float16)])
def test_mean_aggregate_forward_backward(F, dtype):
    from salient_plusplus_amd.models import mean_aggregate
    T, S = 3000, 9000
    rowptr, col = _random_hop(T, S, 20, F)
    x = torch.randn((S, F), generator=torch.Generator().manual_seed(1)).to(dtype).cuda()
    want = _ref_mean(x, rowptr, col, T)
    got = mean_aggregate(x, rowptr, col, T)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    if dtype == torch.float32:
        xg = x.clone().requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        w = torch.randn((T, F), device="cuda")
        (mean_aggregate(xg, rowptr, col, T) * w).sum().backward()
        (_ref_mean(xr, rowptr, col, T) * w).sum().backward()
        torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-4, atol=1e-5)


def test_mean_aggregate_strided_rows_and_empty():
    from salient_plusplus_amd.models import mean_aggregate
    T, S, F = 500, 1200, 100
    rowptr, col = _random_hop(T, S, 9, 5)
    buf = torch.randn((S, 128), device="cuda").half()
    x = buf[:, :F]                                           # padded rows, as the resident table
    torch.testing.assert_close(mean_aggregate(x, rowptr, col, T), _ref_mean(x, rowptr, col, T), rtol=1e-5, atol=1e-5)
    z = mean_aggregate(x, torch.zeros(1, dtype=torch.int64, device="cuda"), col[:0], 0)
    assert z.shape == (0, F)


def test_sage_matches_plain_torch_on_a_sampled_batch():
    """The whole model on a real MFG from the GPU sampler, against the same weights evaluated with
    plain torch ops; then one optimiser step on both and the weights still agree."""
    import bench
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    from salient_plusplus_amd.models import SAGE
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    T_ = torch.from_numpy
    cfg = FastSamplerConfig(
        x_cpu=T_(g["x"]), x_gpu=torch.empty(0), y=T_(g["y"]).unsqueeze(-1), rowptr=T_(g["rowptr"]), col=T_(g["col"]),
        idx=T_(g["idx"]), batch_size=64, sizes=[15, 10, 5], skip_nonfull_batch=False, pin_memory=False,
        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=False, exact_num_batches=0,
        count_remote_frequency=False, use_cache=False)
    dev = torch.device("cuda", 0)
    (batch,) = next(iter(DevicePrefetcher([dev], iter(FastSampler(2, 4, cfg)))))
    torch.cuda.synchronize()
    Fin, C = batch.x.size(1), int(g["y"].max()) + 1
    torch.manual_seed(0)
    hip = SAGE(Fin, 64, C, 3).to(dev).eval()
    ref = bench.TorchSAGE(Fin, 64, C, 3).to(dev).eval()
    for i in range(3):
        ref.lin_l[i].weight.data.copy_(hip.convs[i].lin_l.weight.data)
        ref.lin_r[i].weight.data.copy_(hip.convs[i].lin_r.weight.data)
    out_h, out_r = hip(batch.x, batch.adjs), ref(batch.x, batch.adjs)
    torch.testing.assert_close(out_h, out_r, rtol=1e-4, atol=1e-5)
    y = batch.y.reshape(-1)
    for m in (hip, ref):
        m.zero_grad()
    torch.nn.functional.nll_loss(out_h, y).backward()
    torch.nn.functional.nll_loss(out_r, y).backward()
    for i in range(3):
        torch.testing.assert_close(hip.convs[i].lin_l.weight.grad, ref.lin_l[i].weight.grad, rtol=1e-3, atol=1e-6)
        torch.testing.assert_close(hip.convs[i].lin_r.weight.grad, ref.lin_r[i].weight.grad, rtol=1e-3, atol=1e-6)


@pytest.mark.parametrize("mode", ["member", "group", "batch"])
@pytest.mark.parametrize("F", [128, 100])
def test_fused_first_layer_from_the_resident_table(mode, F, monkeypatch):
    """Row g1 (transferers.py:890-970 + driver/models.py:41-50): with ``table_features`` the Session delivers MFG + labels +
    n_id and NO feature rows; PreparedBatch.x is TableRows(resident table, n_id) and models.SAGE aggregates its first layer
    straight from the table.  Same seeds, same batches as the default Session: n_id names exactly the rows the default x
    holds, the fused model's output is BIT-equal to the materialised path's (same summation order), the weight gradients
    agree, both agree with plain torch, and anything that is not models.SAGE still gets a feature matrix (materialize())."""
    import bench
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    from salient_plusplus_amd.models import GAT, SAGE
    monkeypatch.setenv("SPP_GROUP_DELIVERY", "1" if mode == "group" else "0")
    monkeypatch.setenv("SPP_GROUP_FETCH", "0" if mode == "batch" else "1")
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    T_ = torch.from_numpy
    rng = np.random.default_rng(F)
    n = g["rowptr"].shape[0] - 1
    x_host = T_((rng.standard_normal((n, F)) * 4).astype(np.float16))
    cfg = FastSamplerConfig(
        x_cpu=x_host, x_gpu=torch.empty(0), y=T_(g["y"]).unsqueeze(-1), rowptr=T_(g["rowptr"]), col=T_(g["col"]),
        idx=T_(g["idx"]), batch_size=64, sizes=[15, 10, 5], skip_nonfull_batch=False, pin_memory=False,
        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=False, exact_num_batches=0,
        count_remote_frequency=False, use_cache=False)
    dev = torch.device("cuda", 0)
    plain = [b for (b,) in DevicePrefetcher([dev], iter(FastSampler(2, 4, cfg)))]
    fused = [b for (b,) in DevicePrefetcher([dev], iter(FastSampler(2, 4, cfg, table_features=True)))]
    torch.cuda.synchronize()
    assert len(plain) == len(fused) > 1
    C = int(g["y"].max()) + 1
    torch.manual_seed(0)
    hip = SAGE(F, 64, C, 3).to(dev).train()
    ref = bench.TorchSAGE(F, 64, C, 3).to(dev).eval()
    for i in range(3):
        ref.lin_l[i].weight.data.copy_(hip.convs[i].lin_l.weight.data)
        ref.lin_r[i].weight.data.copy_(hip.convs[i].lin_r.weight.data)
    for bp, bf in zip(plain, fused):
        assert isinstance(bf.x, fs.TableRows) and bf.x.shape == bp.x.shape and bf.x.dtype == bp.x.dtype
        assert bf.idx_range == bp.idx_range and torch.equal(bf.y, bp.y)
        assert torch.equal(bf.x.materialize(), bp.x)                  # the same rows, by the HIP row gather
        assert torch.equal(x_host.to(dev)[bf.x.n_id], bp.x)           # ... and they are x[n_id]
        for hp, hf in zip(bp.adjs, bf.adjs):
            for a, b in zip(hp.adj_t.csr()[:2], hf.adj_t.csr()[:2]):
                assert torch.equal(a, b)
        outs, grads = [], []
        for b in (bp, bf):
            hip.zero_grad()
            torch.manual_seed(1234)                                   # the dropout seeds come from torch's CPU generator
            out = hip(b.x, b.adjs)
            torch.nn.functional.nll_loss(out, b.y.reshape(-1)).backward()
            outs.append(out.detach().clone())
            grads.append([p.grad.detach().clone() for p in hip.parameters()])
        assert torch.equal(outs[0], outs[1])                          # forward: the same sums in the same order
        for ga, gb in zip(*grads):                                    # backward: the small hops' input gradients are fp32
            torch.testing.assert_close(ga, gb, rtol=1e-4, atol=1e-6)  # atomics (order not fixed) in BOTH paths
        hip.eval()
        with torch.no_grad():
            torch.testing.assert_close(hip(bf.x, bf.adjs), ref(bp.x, bp.adjs), rtol=1e-4, atol=1e-5)
        hip.train()
    # a consumer without the fused layer: GAT materialises the rows itself
    gat = GAT(F, 32, C, 3).to(dev).eval()
    with torch.no_grad():
        assert torch.equal(gat(fused[0].x, fused[0].adjs), gat(plain[0].x, plain[0].adjs))


def test_table_features_refused_for_partitioned_sessions():
    from salient_plusplus_amd import fast_sampler as fs
    g = np.load(os.path.join(ROOT, "tests", "golden", "graph_a.npz"))
    T_ = torch.from_numpy
    n = g["rowptr"].shape[0] - 1
    c = fs.Config()
    c.x_cpu, c.x_gpu, c.y = torch.empty((0, g["x"].shape[1]), dtype=torch.float16), T_(g["x"]), T_(g["y"]).unsqueeze(-1)
    c.rowptr, c.col, c.idx = T_(g["rowptr"]), T_(g["col"]), T_(g["idx"])
    c.batch_size, c.sizes = 64, [5, 5]
    c.skip_nonfull_batch = c.pin_memory = c.force_exact_num_batches = False
    c.exact_num_batches = 0
    c.distributed = True
    c.partition_book = fs.RangePartitionBook(0, 1, torch.tensor([0, n]))
    c.cache = fs.Cache()
    c.count_remote_frequency = c.use_cache = False
    s = fs.Session(2, 4, c)
    s.table_features = True
    with pytest.raises(RuntimeError, match="table_features"):
        s.blocking_get_batch_distributed()
    s.close()


def test_end_to_end_training_learns_through_the_data_path():
    """The reference's only integration check is 'training reaches accuracy' (SURVEY §4).  A small
    graph whose labels depend on a node's own features AND on the mean of its neighbours' is
    trained through FastSampler -> DevicePrefetcher -> models.SAGE; held-out accuracy must end far
    above chance, which fails if the MFG orientation, the feature order or the labels were wrong."""
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.shufflers import Shuffler
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    from salient_plusplus_amd.models import SAGE
    from salient_plusplus_amd.synthetic import make_graph
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    n, Fin, C = 6000, 16, 4
    rowptr, col = make_graph(n, 30000, 5, dev)
    x = torch.randn((n, Fin), device=dev)
    deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1)
    row = torch.repeat_interleave(torch.arange(n, device=dev), rowptr[1:] - rowptr[:-1])
    nb_mean = torch.zeros_like(x).index_add_(0, row, x[col]) / deg.unsqueeze(-1)
    w_self, w_nb = torch.randn((Fin, C), device=dev), torch.randn((Fin, C), device=dev)
    y = (x @ w_self + 3.0 * (nb_mean @ w_nb)).argmax(-1)
    perm = torch.randperm(n, device=dev)
    train, test = perm[:4500], perm[4500:]

    def loader(idx, bs):
        cfg = FastSamplerConfig(
            x_cpu=x.half(), x_gpu=torch.empty(0), y=y.unsqueeze(-1), rowptr=rowptr, col=col, idx=idx, batch_size=bs,
            sizes=[10, 10], skip_nonfull_batch=False, pin_memory=False, distributed=False, partition_book=None,
            cache=fs.Cache(), force_exact_num_batches=True, exact_num_batches=max(1, idx.numel() // bs),
            count_remote_frequency=False, use_cache=False)
        return FastSampler(2, 8, cfg)

    model = SAGE(Fin, 64, C, 2).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    shuffler = Shuffler(train)
    sampler = loader(train, 256)
    first = last = None
    for epoch in range(6):
        shuffler.set_epoch(epoch)
        sampler.idx = shuffler.get_idx()
        model.train()
        for (b,) in DevicePrefetcher([dev], iter(sampler)):
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.nll_loss(model(b.x, b.adjs), b.y.reshape(-1))
            loss.backward()
            opt.step()
            first = float(loss.detach()) if first is None else first
            last = float(loss.detach())
    assert last < 0.6 * first, (first, last)
    model.eval()
    hit = tot = 0
    with torch.no_grad():
        for (b,) in DevicePrefetcher([dev], iter(loader(test, 250))):
            pred = model(b.x, b.adjs).argmax(-1)
            hit += int((pred == b.y.reshape(-1)).sum())
            tot += pred.numel()
    assert tot == test.numel() and hit / tot > 0.6, hit / tot        # chance is 0.25


def _ref_gat(h, a_src, a_dst, rowptr, col, T, slope=0.2):
    """plain torch: drop diagonal entries, add one self loop per target, edge softmax, weighted sum"""
    cnt = rowptr[1:] - rowptr[:-1]
    row = torch.repeat_interleave(torch.arange(T, device=h.device), cnt)
    keep = col != row
    row = torch.cat([row[keep], torch.arange(T, device=h.device)])
    src = torch.cat([col[keep], torch.arange(T, device=h.device)])
    e = torch.nn.functional.leaky_relu(a_src[src] + a_dst[row], slope)
    m = torch.full((T,), -float("inf"), device=h.device).scatter_reduce(0, row, e, "amax")
    w = torch.exp(e - m[row])
    s = torch.zeros(T, device=h.device).index_add_(0, row, w)
    return torch.zeros((T, h.size(1)), device=h.device).index_add_(0, row, (w / s[row]).unsqueeze(-1) * h[src])


@pytest.mark.parametrize("F", [256, 47, 8, 100])
def test_gat_aggregate_forward_backward(F):
    from salient_plusplus_amd.models import _GatAggregate
    T, S = 2000, 7000
    rowptr, col = _random_hop(T, S, 15, F)
    col[::11] = torch.repeat_interleave(torch.arange(T, device="cuda"), rowptr[1:] - rowptr[:-1])[::11]   # some diagonal entries
    g = torch.Generator().manual_seed(2)
    h0 = torch.randn((S, F), generator=g).cuda()
    as0 = torch.randn(S, generator=g).cuda()
    ad0 = torch.randn(T, generator=g).cuda()
    ins_a = [t.clone().requires_grad_(True) for t in (h0, as0, ad0)]
    ins_b = [t.clone().requires_grad_(True) for t in (h0, as0, ad0)]
    out_a = _GatAggregate.apply(ins_a[0], ins_a[1], ins_a[2], rowptr, col, 0.2)
    out_b = _ref_gat(ins_b[0], ins_b[1], ins_b[2], rowptr, col, T)
    torch.testing.assert_close(out_a, out_b, rtol=1e-4, atol=1e-5)
    w = torch.randn((T, F), device="cuda")
    (out_a * w).sum().backward()
    (out_b * w).sum().backward()
    for a, b in zip(ins_a, ins_b):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-3, atol=1e-4)


def test_gat_model_runs_and_learns_shapes():
    from salient_plusplus_amd.fast_trainer.monkeypatch import SparseTensor
    from salient_plusplus_amd.models import GAT
    torch.manual_seed(0)
    T1, T0, S = 50, 300, 1500
    rp1, c1 = _random_hop(T0, S, 6, 1)
    rp0, c0 = _random_hop(T1, T0, 6, 2)
    adjs = [(SparseTensor(rowptr=rp1, col=c1, sparse_sizes=(T0, S)), None, (S, T0)),
            (SparseTensor(rowptr=rp0, col=c0, sparse_sizes=(T1, T0)), None, (T0, T1))]
    x = torch.randn((S, 20), device="cuda").half()
    y = torch.randint(0, 5, (T1,), device="cuda")
    model = GAT(20, 32, 5, 2).cuda()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        out = model(x, adjs)
        assert out.shape == (T1, 5)
        loss = torch.nn.functional.nll_loss(out, y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0]


@pytest.mark.parametrize("T,S,F,maxdeg,dtype", [
    (300, 900, 64, 12, torch.float32),        # small hop: backward by fp32 atomics
    (6000, 40000, 256, 40, torch.float32),    # E * F >= 2^22: backward by gather over the transposed hop
    (5000, 30000, 128, 30, torch.float16),    # the first layer's shape: fp16 rows in, no input gradient
])
def test_sage_operand_forward_backward(T, S, F, maxdeg, dtype):
    """[mean_j x_j | x[:T]] in one kernel and its backward (both formulations) against plain torch."""
    from salient_plusplus_amd.models import _MeanAggregate
    rowptr, col = _random_hop(T, S, maxdeg, T + F)
    x = torch.randn((S, F), generator=torch.Generator().manual_seed(3)).to(dtype).cuda()
    got = _MeanAggregate.apply(x, rowptr, col, T, True)
    want = torch.cat([_ref_mean(x, rowptr, col, T), x[:T].float()], dim=1)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
    if dtype == torch.float32:
        xg = x.clone().requires_grad_(True)
        xr = x.clone().requires_grad_(True)
        w = torch.randn((T, 2 * F), device="cuda")
        (_MeanAggregate.apply(xg, rowptr, col, T, True) * w).sum().backward()
        (torch.cat([_ref_mean(xr, rowptr, col, T), xr[:T]], dim=1) * w).sum().backward()
        torch.testing.assert_close(xg.grad, xr.grad, rtol=1e-4, atol=1e-5)


def test_relu_dropout_fused():
    """relu + dropout in one pass: eval mode = relu exactly; training: every output is 0 or 2*relu(x),
    about half of the positive inputs survive, the backward is grad * 2 exactly where the output is
    positive, and torch.manual_seed makes the mask repeatable."""
    from salient_plusplus_amd.models import relu_dropout
    x = torch.randn((4099, 257), device="cuda")          # odd sizes: the scalar tail is exercised
    torch.testing.assert_close(relu_dropout(x, 0.5, False), torch.relu(x), rtol=0, atol=0)
    torch.manual_seed(7)
    xg = x.clone().requires_grad_(True)
    y = relu_dropout(xg, 0.5, True)
    pos = x > 0
    assert bool(((y == 0) | (y == 2 * torch.relu(x))).all())
    assert not bool((y[~pos] != 0).any())
    kept = float((y[pos] > 0).float().mean())
    assert 0.49 < kept < 0.51, kept
    g = torch.randn_like(x)
    y.backward(g)
    torch.testing.assert_close(xg.grad, torch.where(y > 0, 2 * g, torch.zeros_like(g)), rtol=0, atol=0)
    torch.manual_seed(7)
    assert torch.equal(relu_dropout(x, 0.5, True), y.detach())
    torch.manual_seed(8)
    assert not torch.equal(relu_dropout(x, 0.5, True), y.detach())
    kept25 = float((relu_dropout(x, 0.75, True)[pos] > 0).float().mean())
    assert 0.24 < kept25 < 0.26, kept25


@pytest.mark.parametrize("training", [0, 1])
def test_activation_on_load_equals_the_separate_pass(training):
    """spp_sage_operand_forward_act (ReLU + dropout applied to the rows as they are loaded) and
    spp_relu_dropout_backward_pre (mask recomputed from the pre-activation) give, bit for bit, what the separate
    pass + the plain operand kernel + the y-based backward give for the same (p, training, seed)."""
    import ctypes as C
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    T, S, F, p_drop, seed = 1700, 6100, 256, 0.5, 0x1234567890ABCDEF >> 1
    rowptr, col = _random_hop(T, S, 11, 99)
    P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    z = torch.randn((S, F), device="cuda")
    y = torch.empty_like(z)
    nat.check(L.spp_relu_dropout_forward(P(z), z.numel(), p_drop, training, seed, P(y), st))
    a_ref = torch.empty((T, 2 * F), device="cuda")
    nat.check(L.spp_sage_operand_forward(P(rowptr), P(col), T, P(y), 0, F, F, P(a_ref), 2 * F, st))
    a_act = torch.empty((T, 2 * F), device="cuda")
    nat.check(L.spp_sage_operand_forward_act(P(rowptr), P(col), T, P(z), F, P(a_act), 2 * F, p_drop, training, seed, st))
    torch.cuda.synchronize()
    assert torch.equal(a_act, a_ref)
    g = torch.randn((S, F), device="cuda")
    gx_ref, gx_pre = torch.empty_like(g), torch.empty_like(g)
    nat.check(L.spp_relu_dropout_backward(P(g), P(y), g.numel(), 2.0 if training else 1.0, P(gx_ref), st))
    nat.check(L.spp_relu_dropout_backward_pre(P(g), P(z), g.numel(), p_drop, training, seed, P(gx_pre), st))
    torch.cuda.synchronize()
    assert torch.equal(gx_pre, gx_ref)
    # ... and the gather backward with that step in its epilogue equals gather + separate step
    E = int(col.numel())
    gA = torch.randn((T, 2 * F), device="cuda")
    nbytes = int(L.spp_sage_operand_backward_workspace_bytes(T, S, E))
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    g_plain, g_act = torch.empty((S, F), device="cuda"), torch.empty((S, F), device="cuda")
    nat.check(L.spp_sage_operand_backward_gather(P(rowptr), P(col), T, S, E, P(gA), 2 * F, F, P(g_plain), P(ws), nbytes, st))
    nat.check(L.spp_relu_dropout_backward_pre(P(g_plain), P(z), g_plain.numel(), p_drop, training, seed, P(g_plain), st))
    nat.check(L.spp_sage_operand_backward_gather_act(P(rowptr), P(col), T, S, E, P(gA), 2 * F, F, P(g_act), P(ws), nbytes,
                                                     P(z), p_drop, training, seed, st))
    torch.cuda.synchronize()
    # (the transposed hop is filled with atomics: the order of a source's targets, hence the last bits of the sums,
    # differ from call to call -- the mask must agree exactly, the values to rounding)
    assert torch.equal(g_act == 0, g_plain == 0)
    torch.testing.assert_close(g_act, g_plain, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("K,N,dtype", [(128, 256, torch.float16), (256, 64, torch.float32), (100, 47, torch.float32)])
def test_gat_layer_aggregate_then_project_matches_project_then_aggregate(K, N, dtype):
    """_GatLayer (logits from W^T att, aggregation of the raw rows, projection of the targets only) against
    GATConv's own order of operations written with plain torch ops: forward and every gradient."""
    from salient_plusplus_amd.models import _GatLayer
    torch.manual_seed(1000 + K)        # the cotangent below comes from the global generator: independent of test order
    T, S = 1500, 6000
    rowptr, col = _random_hop(T, S, 12, K + N)
    col[::13] = torch.repeat_interleave(torch.arange(T, device="cuda"), rowptr[1:] - rowptr[:-1])[::13]   # diagonal entries
    g = torch.Generator().manual_seed(4)
    x0 = (0.5 * torch.randn((S, K), generator=g)).to(dtype).cuda()
    W0 = (torch.randn((N, K), generator=g) / K ** 0.5).cuda()
    as0, ad0 = torch.randn(N, generator=g).cuda(), torch.randn(N, generator=g).cuda()
    need_gx = dtype == torch.float32
    xa = x0.clone().requires_grad_(need_gx)
    xb = x0.clone().float().requires_grad_(need_gx)
    pa = [t.clone().requires_grad_(True) for t in (W0, as0, ad0)]
    pb = [t.clone().requires_grad_(True) for t in (W0, as0, ad0)]
    out_a = _GatLayer.apply(xa, pa[0], pa[1], pa[2], rowptr, col, T, 0.2)
    h = xb @ pb[0].t()                                        # PyG's order: project every source row
    out_b = _ref_gat(h, h @ pb[1], h[:T] @ pb[2], rowptr, col, T)
    torch.testing.assert_close(out_a, out_b, rtol=2e-4, atol=2e-5)
    w = torch.randn((T, N), device="cuda")
    (out_a * w).sum().backward()
    (out_b * w).sum().backward()
    # (1500-term sums in another association, fp32 atomics: a few 1e-4 absolute on entries of magnitude ~1)
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.grad, b.grad, rtol=2e-3, atol=5e-4)
    if need_gx:
        torch.testing.assert_close(xa.grad, xb.grad, rtol=2e-3, atol=5e-4)
