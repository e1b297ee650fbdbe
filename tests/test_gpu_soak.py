"""-m gpu: seeded randomised soak of the Session runtime against the oracle -- slot counts, group
sizes, batch sizes, fanouts (fast and generic paths), exact-count splits, sessions abandoned in
mid-epoch and pooled samplers reused right after.  Every delivered batch is compared bit for bit;
the point is to catch ordering bugs between the launcher thread, the slot-set streams, the RNG
ping-pong buffers and slot reuse, which single fixed configurations do not provoke."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
T = torch.from_numpy


def _graph(n, seed):
    rng = np.random.default_rng(seed)
    deg = rng.integers(0, 40, n).astype(np.int64)
    deg[rng.random(n) < 0.05] = 0
    deg[rng.integers(0, n, 3)] = 900                    # a few hubs
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    return rowptr, rng.integers(0, n, rowptr[-1]).astype(np.int64)


def _compare(got, rowptr, col, idx, sizes, feats):
    from oracle import oracle as orc
    x, _y, adjs, (start, stop) = got
    m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
    np.testing.assert_array_equal(x.cpu().numpy(), feats[m.n_id])
    for (rp, cl, _e, size), hop in zip(adjs, m.hops):
        np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
        np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
        assert tuple(size) == tuple(hop.size)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_randomised_sessions_match_oracle(seed, monkeypatch):
    from oracle import oracle as orc
    from salient_plusplus_amd import fast_sampler as fs
    rng = np.random.default_rng(1000 + seed)
    n = 4000
    rowptr, col = _graph(n, seed)
    feats = np.arange(n * 3, dtype=np.int32).reshape(n, 3)
    checked = 0
    for trial in range(14):
        sizes = [[15, 10, 5], [5], [25, 15], [3, -1], [2, 2, 2, 2], [40, 3]][int(rng.integers(0, 6))]
        bs = int(rng.choice([1, 7, 32, 100, 257]))
        n_idx = int(rng.integers(1, 1200))
        idx = rng.integers(0, n, n_idx).astype(np.int64)          # duplicates allowed
        slots = int(rng.choice([1, 2, 3, 5, 8, 16]))
        group = int(rng.choice([0, 1, 2, 4, 8]))
        exact = bool(rng.random() < 0.4) and n_idx >= 8
        k = int(rng.integers(1, max(2, n_idx // 4))) if exact else 0
        skip = (not exact) and bool(rng.random() < 0.3)
        monkeypatch.setenv("SPP_GROUP_SIZE", str(group))
        cfg = fs.Config()
        cfg.x_cpu, cfg.y = T(feats), None
        cfg.rowptr, cfg.col, cfg.idx = T(rowptr), T(col), T(idx)
        cfg.batch_size, cfg.sizes = bs, list(sizes)
        cfg.skip_nonfull_batch, cfg.force_exact_num_batches, cfg.exact_num_batches = skip, exact, k
        ranges = orc.batch_ranges(n_idx, bs, skip, exact, k)
        s = fs.Session(2, slots, cfg)
        assert s.num_total_batches == ranges.shape[0]
        stop_after = int(rng.integers(0, ranges.shape[0] + 1)) if rng.random() < 0.35 else ranges.shape[0]
        defer = bool(rng.random() < 0.5)     # compare after the epoch: the consumer never syncs in between,
        held = []                            # so slot reuse has to be ordered by the runtime's own events
        for b in range(stop_after):
            got = s.blocking_get_batch()
            assert got is not None
            assert got[3] == (int(ranges[b][0]), int(ranges[b][1]))
            held.append(got)
            if not defer:
                _compare(held.pop(), rowptr, col, idx, sizes, feats)
                checked += 1
        for got in held:
            _compare(got, rowptr, col, idx, sizes, feats)
            checked += 1
        if stop_after == ranges.shape[0]:
            assert s.blocking_get_batch() is None
        s.close()                                        # possibly in mid-epoch: the sampler goes back to the pool
    assert checked > 50
    fs.clear_resident_cache()
