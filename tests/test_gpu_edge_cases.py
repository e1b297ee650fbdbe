"""-m gpu: edge cases of the sampler against the oracle (empty / ragged inputs, isolated nodes,
duplicated seeds, fanout boundaries, hubs, ids at the end of the range)."""
import ctypes as C

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


def make_graph(n, seed, hub=None, zero_frac=0.1, max_deg=60):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, max_deg, n).astype(np.int64)
    deg[rng.random(n) < zero_frac] = 0
    if hub is not None:
        deg[hub] = min(n - 1, 50_000)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    return rowptr, col


def run_session(fs, rowptr, col, idx, sizes, bs, **kw):
    cfg = fs.Config()
    n = rowptr.shape[0] - 1
    cfg.x_cpu = T(np.arange(n, dtype=np.int64).reshape(n, 1))
    cfg.y = None
    cfg.rowptr, cfg.col, cfg.idx = T(rowptr), T(col), T(np.asarray(idx, dtype=np.int64))
    cfg.batch_size, cfg.sizes = bs, list(sizes)
    for k, v in kw.items():
        setattr(cfg, k, v)
    s = fs.Session(2, 8, cfg)
    out = []
    while True:
        b = s.blocking_get_batch()
        if b is None:
            break
        out.append(b)
    s.close()
    return out


def check_against_oracle(batches, rowptr, col, idx, sizes):
    from oracle import oracle as orc
    idx = np.asarray(idx, dtype=np.int64)
    for (x, y, adjs, (start, stop)) in batches:
        m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
        np.testing.assert_array_equal(x.cpu().numpy().reshape(-1), m.n_id)
        assert y is None
        for (rp, cl, e_id, size), hop in zip(adjs, m.hops):
            np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
            assert tuple(size) == tuple(hop.size)


def test_empty_index_gives_no_batches(fs):
    rowptr, col = make_graph(100, 0)
    assert run_session(fs, rowptr, col, [], [5, 5], 16) == []


def test_single_seed_and_isolated_seeds(fs):
    rowptr, col = make_graph(500, 1, zero_frac=0.5)
    deg = np.diff(rowptr)
    iso = np.flatnonzero(deg == 0)[:7]
    for idx in ([3], list(iso), list(iso) + [11]):
        b = run_session(fs, rowptr, col, idx, [15, 10, 5], 64)
        assert len(b) == 1
        check_against_oracle(b, rowptr, col, idx, [15, 10, 5])
    # a batch of isolated seeds has an empty MFG but still U == #seeds rows
    b = run_session(fs, rowptr, col, list(iso), [4, 4], 64)
    assert b[0][0].shape[0] == len(iso) and all(a[1].numel() == 0 for a in b[0][2])


def test_all_seeds_identical_and_heavy_duplication(fs):
    rowptr, col = make_graph(300, 2, zero_frac=0.0)
    for idx in ([42] * 40, [1, 2, 1, 2, 1, 2, 3, 3, 3], list(range(20)) * 3):
        b = run_session(fs, rowptr, col, idx, [6, 3], 64)
        check_against_oracle(b, rowptr, col, idx, [6, 3])


@pytest.mark.parametrize("sizes", [[32], [33], [32, 32], [31, 2], [0], [0, 0, 3]])
def test_fanout_boundaries(fs, sizes):
    rowptr, col = make_graph(2000, 3, zero_frac=0.05, max_deg=120)
    idx = np.random.default_rng(3).permutation(2000)[:96]
    b = run_session(fs, rowptr, col, idx, sizes, 32)
    assert len(b) == 3
    check_against_oracle(b, rowptr, col, idx, sizes)


def test_degree_exactly_at_the_fanout(fs):
    """deg == f takes all neighbours without consuming draws, deg == f+1 runs Floyd."""
    n = 64
    deg = np.array([5, 6, 4, 0, 5, 6] * 10 + [5, 6, 5, 6], dtype=np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = np.random.default_rng(4).integers(0, n, rowptr[-1]).astype(np.int64)
    idx = np.arange(n)
    b = run_session(fs, rowptr, col, idx, [5, 5], 16)
    check_against_oracle(b, rowptr, col, idx, [5, 5])


def test_hub_node_and_last_node_ids(fs):
    n = 60_000
    rowptr, col = make_graph(n, 5, hub=n - 1)
    col[: 2000] = n - 1                         # many edges point at the hub / the last id
    idx = np.concatenate([[n - 1, n - 2, 0], np.random.default_rng(5).permutation(n)[:125]])
    for sizes in ([15, 10, 5], [-1], [25, 15]):
        b = run_session(fs, rowptr, col, idx, sizes, 64)
        check_against_oracle(b, rowptr, col, idx, sizes)


def test_ragged_last_batch_and_skip_nonfull(fs):
    rowptr, col = make_graph(1000, 6)
    idx = np.random.default_rng(6).permutation(1000)[:150]
    full = run_session(fs, rowptr, col, idx, [5, 5], 64)
    assert [b[3] for b in full] == [(0, 64), (64, 128), (128, 150)]
    check_against_oracle(full, rowptr, col, idx, [5, 5])
    skipped = run_session(fs, rowptr, col, idx, [5, 5], 64, skip_nonfull_batch=True)
    assert [b[3] for b in skipped] == [(0, 64), (64, 128)]


def test_many_small_batches_span_several_groups(fs):
    """More batches than slots: every slot-set is recycled several times (ping-pong RNG buffers)."""
    rowptr, col = make_graph(5000, 7)
    idx = np.random.default_rng(7).permutation(5000)[:2000]
    b = run_session(fs, rowptr, col, idx, [15, 10, 5], 16)
    assert len(b) == 125
    check_against_oracle(b, rowptr, col, idx, [15, 10, 5])


def test_capacity_error_is_reported_not_silent(fs):
    """A batch larger than the sampler was created for must fail loudly (SPP_ERR_INVALID)."""
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    rowptr, col = make_graph(100, 8)
    rp, cl = T(rowptr).cuda(), T(col).cuda()
    cfg = nat.SamplerCfg()
    cfg.rowptr_dev, cfg.col_dev, cfg.num_nodes, cfg.nnz = rp.data_ptr(), cl.data_ptr(), 100, cl.numel()
    cfg.num_hops, cfg.max_batch, cfg.num_slots, cfg.device = 1, 4, 1, 0
    cfg.sizes[0] = 3
    h = C.c_void_p()
    assert L.spp_sampler_create(C.byref(cfg), C.byref(h)) == 0
    seeds = torch.arange(8, dtype=torch.int64, device="cuda")
    assert L.spp_sampler_sample(h, 0, C.c_void_p(seeds.data_ptr()), 8, 1, 0, None) < 0
    assert b"max_batch" in L.spp_last_error()
    assert L.spp_sampler_sample(h, 5, C.c_void_p(seeds.data_ptr()), 2, 1, 0, None) < 0
    L.spp_sampler_destroy(h)


@pytest.mark.parametrize("sizes,bs", [([15, 10, 5], 512), ([4, 4], 1024), ([20, 20, 20], 256)])
def test_one_node_reached_by_thousands_of_edges_of_a_hop(fs, sizes, bs):
    """Fixed-capacity bucket regions (round 4): every vertex lists vertex 7 among its few neighbours, so one dedup
    bucket receives thousands of pairs per hop -- far past its region -- and the overflow list carries them; a second
    family of hubs (ids = 0 mod 97) spreads more overflow over several buckets.  Bit-exact against the oracle."""
    n = 40_000
    rng = np.random.default_rng(11)
    deg = rng.integers(3, 9, n).astype(np.int64)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    col[rowptr[:-1]] = 7                                        # first neighbour of everyone
    second = rowptr[:-1] + 1
    col[second] = (rng.integers(0, n // 97, n) * 97)            # second neighbour: one of ~400 hubs
    idx = rng.permutation(n)[:2 * bs]
    b = run_session(fs, rowptr, col, idx, sizes, bs)
    assert len(b) == 2
    check_against_oracle(b, rowptr, col, idx, sizes)
