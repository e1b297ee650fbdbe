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
       n_batches=st.integers(1, 9), slots=st.sampled_from([1, 2, 5, 16, 64]), dup=st.booleans())
def test_random_graph_against_the_oracle(fs, seed, n, mean_deg, zero_frac, n_hubs, hub_share, sizes, bs, n_batches, slots, dup):
    from oracle import oracle as orc
    if os.environ.get("SPP_FUZZ_LOG"):          # the case about to run: the last line names the one that hung or crashed
        with open(os.environ["SPP_FUZZ_LOG"], "a") as f:
            f.write(repr(dict(seed=seed, n=n, mean_deg=mean_deg, zero_frac=zero_frac, n_hubs=n_hubs, hub_share=hub_share, sizes=sizes,
                              bs=bs, n_batches=n_batches, slots=slots, dup=dup)) + "\n")
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
        n_batches=5, slots=16, dup=bool(k & 1))
