"""-m gpu: randomly drawn graphs, fan-outs, batch sizes and slot counts through the Session, bit for bit against the oracle.

hypothesis draws the cases (derandomised: the same ones every run); what varies is what the hand-written cases of
test_gpu_edge_cases.py / test_gpu_pipeline.py fix: the degree distribution (isolated vertices, rows at, just below and far
above every fan-out, a few hubs that collect a large share of a hop's edges -> bucket regions overflow), duplicated seeds,
ragged last batches, fast (<= 32) / generic (-1, > 32) hops mixed, one to several sampling groups in flight.
SPP_FUZZ_EXAMPLES=<n> runs more cases, SPP_FUZZ_RANDOM=1 draws fresh ones (tools/fuzz_long.sh)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402

T = torch.from_numpy
FANOUTS = [[15, 10, 5], [5, 5], [3], [32], [1, 1, 1, 1], [20, 20, 20], [25, 15], [0, 4], [33], [-1], [4, -1], [2, 40]]
EXAMPLES = int(os.environ.get("SPP_FUZZ_EXAMPLES", "64"))
DERANDOMIZE = os.environ.get("SPP_FUZZ_RANDOM", "0") != "1"


@pytest.fixture(scope="module")
def fs():
    from salient_plusplus_amd import _native as nat
    nat.load()
    nat.require_device()
    from salient_plusplus_amd import fast_sampler
    return fast_sampler


def _graph(rng, n, mean_deg, zero_frac, n_hubs, hub_share):
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    deg[rng.random(n) < zero_frac] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    if n_hubs and col.size:
        hubs = rng.choice(n, size=n_hubs, replace=False)
        hit = rng.random(col.size) < hub_share            # this share of ALL edges points at one of a few hubs
        col[hit] = hubs[rng.integers(0, n_hubs, int(hit.sum()))]
    return rowptr, col


@settings(max_examples=EXAMPLES, deadline=None, derandomize=DERANDOMIZE, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(40, 6000), mean_deg=st.floats(0.5, 40.0), zero_frac=st.floats(0.0, 0.5),
       n_hubs=st.integers(0, 3), hub_share=st.floats(0.0, 0.6), sizes=st.sampled_from(FANOUTS), bs=st.sampled_from([1, 7, 64, 256, 1024]),
       n_batches=st.integers(1, 9), slots=st.sampled_from([1, 2, 5, 16, 64]), dup=st.booleans(),
       variant=st.sampled_from([None, None, None, dict(row_stubs=False), dict(deg_tags=False), dict(col32=False), dict(rng_arena=False),
                                dict(flag_tiled=False, rows_coalesced=False), dict(fuse_scatter=-1), dict(fuse_scatter=2),
                                dict(col32=False, row_stubs=False, rng_arena=False), dict(initial_edge_cap=256)]))
def test_random_graph_against_the_oracle(fs, seed, n, mean_deg, zero_frac, n_hubs, hub_share, sizes, bs, n_batches, slots, dup, variant):
    """(round 6: `variant` pins one of the chain's forms -- include/spp.h spp_sampler_opts -- for the case's samplers)"""
    from oracle import oracle as orc
    if variant:
        with fs.sampler_options(**variant):
            return test_random_graph_against_the_oracle.hypothesis.inner_test(fs, seed, n, mean_deg, zero_frac, n_hubs, hub_share, sizes, bs,
                                                                              n_batches, slots, dup, None)
    if os.environ.get("SPP_FUZZ_LOG"):          # the case about to run: the last line names the one that hung or crashed
        with open(os.environ["SPP_FUZZ_LOG"], "a") as f:
            f.write(repr(dict(seed=seed, n=n, mean_deg=mean_deg, zero_frac=zero_frac, n_hubs=n_hubs, hub_share=hub_share, sizes=sizes,
                              bs=bs, n_batches=n_batches, slots=slots, dup=dup, variant=dict(fs._sampler_opts))) + "\n")
    rng = np.random.default_rng(seed)
    rowptr, col = _graph(rng, n, mean_deg, zero_frac, n_hubs, hub_share)
    n_idx = max(1, min(bs * n_batches - int(rng.integers(0, bs)), 4 * n))
    idx = rng.integers(0, n, n_idx) if dup else rng.permutation(np.resize(np.arange(n), n_idx))
    idx = idx.astype(np.int64)
    cfg = fs.Config()
    cfg.x_cpu = T(np.arange(n, dtype=np.int64).reshape(n, 1))
    cfg.y = T((np.arange(n, dtype=np.int64) * 3 + 1).reshape(n, 1))
    cfg.rowptr, cfg.col, cfg.idx = T(rowptr), T(col), T(idx)
    cfg.batch_size, cfg.sizes = bs, list(sizes)
    s = fs.Session(2, slots, cfg)
    got = []
    while True:
        b = s.blocking_get_batch()
        if b is None:
            break
        got.append(b)
    s.close()
    ranges = orc.batch_ranges(len(idx), bs)
    assert len(got) == len(ranges)
    for (x, y, adjs, (start, stop)), (r0, r1) in zip(got, ranges):
        assert (start, stop) == (int(r0), int(r1))
        m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
        np.testing.assert_array_equal(x.cpu().numpy().reshape(-1), m.n_id)
        np.testing.assert_array_equal(y.cpu().numpy().reshape(-1), m.n_id[:stop - start] * 3 + 1)
        for (rp, cl, _e, size), hop in zip(adjs, m.hops):
            np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
            assert tuple(size) == tuple(hop.size)


@pytest.mark.parametrize("k", range(len(FANOUTS)))
def test_larger_random_graph_with_hubs_against_the_oracle(fs, k):
    """one larger case per fan-out list (20 k vertices, three hubs collecting 40 % of all edges, 5 batches of 512 seeds with
    duplicates, 16 slots): the sizes hypothesis rarely draws"""
    test_random_graph_against_the_oracle.hypothesis.inner_test(
        fs, seed=1000 + k, n=20_000, mean_deg=12.0 + k, zero_frac=0.05, n_hubs=3, hub_share=0.4, sizes=FANOUTS[k], bs=512,
        n_batches=5, slots=16, dup=bool(k & 1), variant=None)


@settings(max_examples=EXAMPLES, deadline=None, derandomize=DERANDOMIZE, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(30, 4000), mean_deg=st.floats(0.5, 45.0), zero_frac=st.floats(0.0, 0.5),
       gen_seed=st.integers(0, 2**32 - 1),
       calls=st.lists(st.one_of(st.tuples(st.just("adj"), st.sampled_from([-1, 0, 1, 4, 15, 32, 33, 50]), st.booleans(), st.integers(1, 700)),
                                st.tuples(st.just("multi"), st.sampled_from(FANOUTS), st.just(False), st.integers(1, 300))), min_size=1, max_size=5))
def test_free_functions_random_call_sequences(fs, seed, n, mean_deg, zero_frac, gen_seed, calls):
    """sample_adj / multilayer_sample (fast_sampler.cpp:1339-1350) draw from ONE per-thread generator: a random sequence of calls
    (fan-outs incl. all-neighbour and > 32, with and without replacement, duplicated seeds) against the oracle fed by one
    std::mt19937 through the same sequence."""
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_sampler import _gen
    rng = np.random.default_rng(seed)
    rowptr, col = _graph(rng, n, mean_deg, zero_frac, 0, 0.0)
    _gen.seed, _gen.pos = gen_seed, 0
    mt = orc.MT.seeded(gen_seed)
    rp, cl = T(rowptr), T(col)
    for kind, arg, replace, n_seeds in calls:
        idx = rng.integers(0, n, n_seeds).astype(np.int64)
        if kind == "adj":
            got = fs.sample_adj(rp, cl, T(idx), arg, replace)
            want = orc.sample_adj(rowptr, col, idx, arg, replace, mt)
            np.testing.assert_array_equal(got[0].cpu().numpy(), want.hops[0].rowptr)
            np.testing.assert_array_equal(got[1].cpu().numpy(), want.hops[0].col)
            np.testing.assert_array_equal(got[2].cpu().numpy().astype(np.int64), want.n_id)
        else:
            n_id, adjs = fs.multilayer_sample(T(idx), list(arg), rp, cl)
            want = orc.multilayer_sample(rowptr, col, idx, arg, mt)
            np.testing.assert_array_equal(n_id.cpu().numpy(), want.n_id)
            assert len(adjs) == len(want.hops)
            for (r_, c_, _e, sz), hop in zip(adjs, want.hops):
                np.testing.assert_array_equal(r_.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(c_.cpu().numpy(), hop.col)
                assert tuple(sz) == tuple(hop.size)


@settings(max_examples=EXAMPLES, deadline=None, derandomize=DERANDOMIZE, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(1, 100_000), P=st.integers(1, 9), empty=st.integers(0, 2), n_probe=st.integers(0, 5000),
       cache_frac=st.sampled_from([0.0, 0.001, 0.1, 1.0]), on_gpu=st.booleans())
def test_partition_book_and_cache_lookups_random(fs, seed, n, P, empty, n_probe, cache_frac, on_gpu):
    """RangePartitionBook (range_partition_book.cpp:85-112) and Cache (:116-195) lookups over random range tables (incl. empty
    partitions), probe lists (incl. empty ones, host and device tensors) and cached sets from none to every remote vertex."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.sort(rng.integers(0, n + 1, P - 1)), [n]]).astype(np.int64)
    for _ in range(min(empty, P - 1)):
        k = int(rng.integers(1, P))
        off[k] = off[k - 1]
    off = np.maximum.accumulate(off)
    rank = int(rng.integers(0, P))
    probe = rng.integers(0, n, n_probe).astype(np.int64)
    pb = fs.RangePartitionBook(rank, P, T(off))
    tp = T(probe).cuda() if on_gpu else T(probe)
    np.testing.assert_array_equal(pb.nid2partid(tp).cpu().numpy(), orc.nid2partid(off, probe))
    part = int(rng.integers(0, P))
    inside = probe[(probe >= off[part]) & (probe < off[part + 1])]
    np.testing.assert_array_equal(pb.nid2localnid(T(inside), part).cpu().numpy(), orc.nid2localnid(off, inside, part))
    np.testing.assert_array_equal(pb.partid2nids(part).cpu().numpy(), np.arange(off[part], off[part + 1]))
    remote = np.setdiff1d(np.arange(n), np.arange(off[rank], off[rank + 1]))
    cv = np.sort(rng.choice(remote, size=int(round(cache_frac * remote.size)), replace=False)).astype(np.int64)
    cache = fs.Cache(rank, P, T(cv), torch.zeros((cv.size, 2), dtype=torch.float16))
    ocache = orc.Cache(cv, n)
    np.testing.assert_array_equal(cache.nid_is_cached(tp).cpu().numpy().astype(bool), ocache.nid_is_cached(probe))
    hit = probe[ocache.nid_is_cached(probe)]
    np.testing.assert_array_equal(cache.nid2cachenid(T(hit)).cpu().numpy(), ocache.nid2cachenid(hit))


@settings(max_examples=EXAMPLES, deadline=None, derandomize=DERANDOMIZE, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(40, 5000), mean_deg=st.floats(0.5, 35.0), sizes=st.sampled_from(FANOUTS),
       bs=st.sampled_from([1, 3, 50, 256, 1024]), n_idx=st.integers(1, 4000), skip=st.booleans(), force=st.booleans(), exact=st.integers(1, 12),
       slots=st.sampled_from([1, 3, 16, 64]), xdt=st.sampled_from(["float16", "float32", "int64", "uint8"]), F=st.sampled_from([1, 3, 16, 100, 128]),
       x_on_gpu=st.booleans(), with_y=st.booleans(), epochs=st.integers(1, 2))
def test_random_facade_configurations(fs, seed, n, mean_deg, sizes, bs, n_idx, skip, force, exact, slots, xdt, F, x_on_gpu, with_y, epochs):
    """The façade (FastSamplerConfig -> FastSampler -> DevicePrefetcher; samplers.py:213-399, transferers.py:890-970) over random
    batch-range options (skip_nonfull_batch, force_exact_num_batches: fast_sampler.cpp:587-627), feature dtypes and widths, the
    feature matrix handed over as a host or a device tensor, with and without labels, and a second epoch with new seeds on the
    same sampler: x = features[n_id], y = labels[n_id[:batch]], the MFG bit for bit."""
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
    from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
    rng = np.random.default_rng(seed)
    rowptr, col = _graph(rng, n, mean_deg, 0.1, 0, 0.0)
    x = rng.integers(0, 250, (n, F)).astype(xdt)
    y = rng.integers(0, 40, n).astype(np.int64)
    if force:
        n_idx = max(n_idx, exact)              # at least one seed per batch
    dev = torch.device("cuda", 0)
    cfg = FastSamplerConfig(
        x_cpu=T(x).cuda() if x_on_gpu else T(x), x_gpu=torch.empty(0), y=T(y).unsqueeze(-1) if with_y else None, rowptr=T(rowptr), col=T(col),
        idx=T(rng.integers(0, n, n_idx).astype(np.int64)), batch_size=bs, sizes=list(sizes), skip_nonfull_batch=skip, pin_memory=False,
        distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=force, exact_num_batches=exact,
        count_remote_frequency=False, use_cache=False)
    sampler = FastSampler(2, slots, cfg)
    for epoch in range(epochs):
        if epoch:
            sampler.idx = T(rng.integers(0, n, n_idx).astype(np.int64))
        idx = sampler.idx.numpy()
        ranges = orc.batch_ranges(len(idx), bs, skip, force, exact)
        assert len(sampler) == len(ranges)
        got = [b for (b,) in DevicePrefetcher([dev], iter(sampler))]
        assert len(got) == len(ranges)
        for b, (r0, r1) in zip(got, ranges):
            assert (b.idx_range.start, b.idx_range.stop) == (int(r0), int(r1))
            m = orc.sample_batch(rowptr, col, idx, int(r0), int(r1), sizes)
            assert b.x.dtype == T(x[:0]).dtype
            np.testing.assert_array_equal(b.x.cpu().numpy(), x[m.n_id])
            if with_y:
                np.testing.assert_array_equal(b.y.cpu().numpy().reshape(-1), y[m.n_id[:int(r1) - int(r0)]])
            else:
                assert b.y is None
            for adj, hop in zip(b.adjs, m.hops):
                rp, cl, _ = adj.adj_t.csr()
                np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
                assert tuple(adj.size) == (hop.size[1], hop.size[0])
