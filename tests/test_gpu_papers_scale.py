"""-m gpu: maximum sizes.  A papers100M-scale topology (111 M nodes, > 2^31 edge slots, int64
offsets beyond the 32-bit range, 18 GB `col`) resident in HBM: two batches are compared bit-for-bit
with the oracle; feature rows come from a 111 M x 16 fp16 matrix."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def test_papers100m_scale_topology_bit_exact():
    from salient_plusplus_amd import _native as nat
    nat.load()
    nat.require_device()
    free, total = torch.cuda.mem_get_info()
    if free < 60 * (1 << 30):
        pytest.skip("needs ~45 GB of free HBM")
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(2024)
    N = 111_059_956
    deg = torch.randint(0, 41, (N,), generator=g, device=dev, dtype=torch.int64)      # mean 20 -> 2.2e9 edges
    rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(deg, 0, out=rowptr[1:])
    del deg
    nnz = int(rowptr[-1])
    assert nnz > 2**31
    col = torch.empty(nnz, dtype=torch.int64, device=dev)
    step = 1 << 28
    for lo in range(0, nnz, step):                                                   # chunked: bounded temporaries
        hi = min(nnz, lo + step)
        col[lo:hi] = torch.randint(0, N, (hi - lo,), generator=g, device=dev, dtype=torch.int64)
    F = 16
    x = torch.randn((N, F), generator=g, device=dev, dtype=torch.float16)
    y = torch.randint(0, 172, (N,), generator=g, device=dev, dtype=torch.int64)
    # seeds from the END of the id range, where row offsets exceed 2^31
    idx = (N - 1 - torch.randperm(1_000_000, generator=g, device=dev)[:2048]).contiguous()
    assert int(rowptr[idx.min()]) > 2**31
    cfg = FastSamplerConfig(
        x_cpu=x, x_gpu=torch.empty(0), y=y.unsqueeze(-1), rowptr=rowptr, col=col, idx=idx, batch_size=1024,
        sizes=[15, 10, 5], skip_nonfull_batch=False, pin_memory=False, distributed=False, partition_book=None,
        cache=fs.Cache(), force_exact_num_batches=True, exact_num_batches=2, count_remote_frequency=False,
        use_cache=False)
    batches = list(iter(FastSampler(2, 4, cfg)))
    assert len(batches) == 2
    # the oracle needs the topology on the host
    rowptr_h, col_h, idx_h = rowptr.cpu().numpy(), col.cpu().numpy(), idx.cpu().numpy()
    for b, batch in enumerate(batches):
        start, stop = batch.idx_range.start, batch.idx_range.stop
        m = orc.sample_batch(rowptr_h, col_h, idx_h, start, stop, [15, 10, 5])
        assert m.num_edges > 500_000
        for adj, hop in zip(batch.adjs, m.hops):
            rp, cl, _ = adj.adj_t.csr()
            np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
        n_id = torch.from_numpy(m.n_id).to(dev)
        assert torch.equal(batch.x, x[n_id])
        assert torch.equal(batch.y, y[n_id[:stop - start]])
    fs.clear_resident_cache()
