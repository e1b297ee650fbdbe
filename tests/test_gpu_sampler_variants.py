"""-m gpu: every form of the sampling chain the library can select BY ITSELF -- by the size of the graph, the free HBM
or the fan-outs -- computes the same batches (sample_adj, sample_cpu.hpp:25-143; multilayer_sample,
fast_sampler.cpp:191-227; per-batch seeding :994).  The automatic rules (include/spp.h spp_sampler_opts) are pinned
one way and the other through the sampler configuration, inside ONE process:

  * no row stubs (S-mag on one GPU: 128 B x N exceeds an eighth of the free HBM) -- degrees from rowptr,
    cooperative reads of the int32 array (k_hop_pick<int32, no stub>);
  * no degree tags (a fan-out >= the tag cap) -- the degree pass reads the stub headers (k_hop_pick<int32, stub>);
  * the int64 neighbour array, with and without stubs (k_hop_pick<int64, ...>);
  * per-group mt19937 generation instead of the epoch arena (k_rng_fill: the arena over a quarter of the free HBM);
  * the position-ordered flag pass (k_hop_flag), the lane-per-row rows kernel (k_hop_rows);
  * the bucket scatter never / everywhere folded into the pick; the dedup's table pre-read.

Each variant runs the reference's own fixtures on graph_a (every fan-out list tests/golden holds) AND randomly drawn
graphs with hubs at the headline batch size (1024 seeds per batch, several groups in flight) against the oracle, and
spp_sampler_get_info must report the variant that was asked for -- so a test of "the default" cannot silently be a
test of something else."""
import os

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


# name -> (options, what spp_sampler_get_info must say for a fast-path sampler)
VARIANTS = {
    "default": (dict(), dict(col32=1, row_stubs=1, deg_tags=1, rng_arena=1)),
    "no_row_stubs": (dict(row_stubs=False), dict(row_stubs=0, deg_tags=0, col32=1)),
    "no_degree_tags": (dict(deg_tags=False), dict(deg_tags=0, row_stubs=1, col32=1, idbits=32)),
    "int64_col": (dict(col32=False), dict(col32=0, row_stubs=1, deg_tags=0)),
    "int64_col_no_stubs": (dict(col32=False, row_stubs=False), dict(col32=0, row_stubs=0, deg_tags=0)),
    "rng_per_group": (dict(rng_arena=False), dict(rng_arena=0)),
    "rng_arena_over_budget": (dict(rng_arena_mb=1), dict()),          # 1 MiB: the headline-size epochs do not fit
    "flag_not_tiled": (dict(flag_tiled=False), dict(flag_tiled=0)),
    "rows_not_coalesced": (dict(rows_coalesced=False), dict(rows_coalesced=0)),
    "fuse_never": (dict(fuse_scatter=-1), dict(fused_pick=0)),
    "fuse_everywhere": (dict(fuse_scatter=2), dict()),
    "dedup_preread": (dict(dedup_preread=True), dict()),
    "everything_off": (dict(col32=False, row_stubs=False, rng_arena=False, flag_tiled=False, rows_coalesced=False,
                            fuse_scatter=-1), dict(col32=0, row_stubs=0, rng_arena=0, flag_tiled=0, rows_coalesced=0,
                                                    fused_pick=0)),
    "no_stubs_per_group_rng": (dict(row_stubs=False, rng_arena=False), dict(row_stubs=0, rng_arena=0)),   # S-mag when HBM is short
}
GOLDEN_CASES = ["s15_10_5", "s20_20_20", "sall", "s25_15", "s3_all", "s1", "s0_2"]


def _check_info(info, want, sizes):
    fast = all(0 <= f <= 32 for f in sizes)
    for k, v in want.items():
        got = info[k]
        if isinstance(got, list):
            assert all(g == v for g in got), (k, got, v)
        elif k == "rng_arena":
            assert got == v or all(f < 0 for f in sizes), (k, got, v)   # (all-neighbour hops draw nothing: no streams at all)
        elif fast or k == "col32":                                      # (stubs / tags exist on the fast path only)
            assert got == v, (k, got, v)


def _drain(s):
    out = []
    while True:
        b = s.blocking_get_batch()
        if b is None:
            return out
        out.append(b)


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_on_the_reference_fixtures(fs, graph_a, golden_dir, name):
    opts, want = VARIANTS[name]
    n = graph_a["rowptr"].shape[0] - 1
    with fs.sampler_options(**opts):
        for case in GOLDEN_CASES:
            g = np.load(os.path.join(golden_dir, f"mfg_a_{case}.npz"))
            sizes = [int(v) for v in g["sizes"]]
            cfg = fs.Config()
            cfg.x_cpu = T(np.arange(n, dtype=np.int64).reshape(n, 1))
            cfg.y = T(graph_a["y"]).unsqueeze(-1)
            cfg.rowptr, cfg.col, cfg.idx = T(graph_a["rowptr"]), T(graph_a["col"]), T(graph_a["idx"])
            cfg.batch_size, cfg.sizes = 64, sizes
            s = fs.Session(2, 8, cfg)
            got = _drain(s)
            info = s.sampler_info()
            s.close()
            _check_info(info, want, sizes)
            assert len(got) == int(g["num_batches"])
            for b, (x, y, adjs, rng) in enumerate(got):
                assert tuple(rng) == tuple(int(v) for v in g[f"b{b}_range"])
                np.testing.assert_array_equal(x.cpu().numpy().reshape(-1), g[f"b{b}_n_id"])
                np.testing.assert_array_equal(y.cpu().numpy().reshape(-1), g[f"b{b}_y"])
                for h, (rp, cl, e_id, size) in enumerate(adjs):
                    np.testing.assert_array_equal(rp.cpu().numpy(), g[f"b{b}_h{h}_rowptr"])
                    np.testing.assert_array_equal(cl.cpu().numpy(), g[f"b{b}_h{h}_col"])
                    assert tuple(size) == tuple(int(v) for v in g[f"b{b}_h{h}_size"]) and e_id.numel() == 0


def _hub_graph(seed, n, mean_deg, n_hubs, hub_share):
    rng = np.random.default_rng(seed)
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    deg[rng.random(n) < 0.05] = 0
    big = rng.random(n) < 0.03                                          # rows of 60-300 neighbours: picks beyond the stub's 29
    deg[big] = rng.integers(60, 300, int(big.sum()))
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    if n_hubs:
        hubs = rng.choice(n, size=n_hubs, replace=False)
        hit = rng.random(col.size) < hub_share
        col[hit] = hubs[rng.integers(0, n_hubs, int(hit.sum()))]
    return rng, rowptr, col


# (fan-outs, vertices, mean degree): the headline list on a graph large enough that the last hop runs the tile kernels,
# S-mag's two hops (fan-out 25: a fused pick at its LDS ceiling), and the batchwise-inference list whose middle hop
# exceeds fuse_max_edges
HEADLINE = [([15, 10, 5], 60_000, 18.0), ([25, 15], 40_000, 30.0), ([20, 20, 20], 14_000, 24.0)]


@pytest.mark.parametrize("name", list(VARIANTS))
def test_variant_at_the_headline_batch_size(fs, name):
    from oracle import oracle as orc
    opts, want = VARIANTS[name]
    with fs.sampler_options(**opts):
        for k, (sizes, n, mean_deg) in enumerate(HEADLINE):
            rng, rowptr, col = _hub_graph(4242 + k, n, mean_deg, 2, 0.15)
            bs, nb = 1024, 19                                           # 19 batches: a full group of 16 and a ragged one
            idx = rng.integers(0, n, bs * nb - 300).astype(np.int64)     # duplicated seeds, ragged last batch
            cfg = fs.Config()
            cfg.x_cpu = T(np.arange(n, dtype=np.int64).reshape(n, 1))
            cfg.y = T((np.arange(n, dtype=np.int64) * 3 + 1).reshape(n, 1))
            cfg.rowptr, cfg.col, cfg.idx = T(rowptr), T(col), T(idx)
            cfg.batch_size, cfg.sizes = bs, list(sizes)
            for epoch in range(2):                                      # the second Session borrows the pooled sampler (and its arena)
                s = fs.Session(2, 64, cfg)
                got = _drain(s)
                info = s.sampler_info()
                s.close()
                _check_info(info, want, sizes)
                if name == "rng_arena_over_budget":
                    assert info["rng_arena"] == 0 and info["rng_arena_bytes"] == 0
                if name == "fuse_everywhere":
                    assert all(info["fused_pick"]) or sizes != [15, 10, 5], info
                if name == "default" and sizes == [15, 10, 5]:
                    assert info["fused_pick"] == [1, 1, 0] and info["flag_tiled"] == [0, 0, 1] and info["rows_coalesced"] == [1, 1, 1]
                ranges = orc.batch_ranges(len(idx), bs)
                assert len(got) == len(ranges) == nb
                for (x, y, adjs, (start, stop)), (r0, r1) in zip(got, ranges):
                    assert (start, stop) == (int(r0), int(r1))
                    if epoch == 1 and start % (3 * bs):                 # every batch once, a third of them again
                        continue
                    m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
                    np.testing.assert_array_equal(x.cpu().numpy().reshape(-1), m.n_id)
                    np.testing.assert_array_equal(y.cpu().numpy().reshape(-1), m.n_id[:stop - start] * 3 + 1)
                    for (rp, cl, _e, size), hop in zip(adjs, m.hops):
                        np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                        np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
                        assert tuple(size) == tuple(hop.size)
    fs.clear_resident_cache()                                           # the variants' samplers and tables do not pile up


@pytest.mark.parametrize("sizes", [[-1, -1], [3, -1], [-1, 4], [40, -1]])
def test_all_neighbour_hop_that_grows_its_scratch_while_a_bucket_overflows(fs, sizes):
    """An all-neighbour hop sizes its per-edge scratch after a host read of the hop's edge count.  When it has to GROW the
    scratch, the overflow list behind the fixed-capacity bucket regions must be placed by the NEW capacity (the regions are
    sized from the hop's edges): one hub collects 40 % of all edges, so its bucket overflows in every hop, and the starting
    capacity is far below the second hop's edge count."""
    from oracle import oracle as orc
    rng, rowptr, col = _hub_graph(77, 20_000, 30.0, 1, 0.4)
    n = rowptr.shape[0] - 1
    bs = 2048
    idx = rng.integers(0, n, 2 * bs + 100).astype(np.int64)
    with fs.sampler_options(initial_edge_cap=1024):
        cfg = fs.Config()
        cfg.x_cpu = T(np.arange(n, dtype=np.int64).reshape(n, 1))
        cfg.y = None
        cfg.rowptr, cfg.col, cfg.idx = T(rowptr), T(col), T(idx)
        cfg.batch_size, cfg.sizes = bs, list(sizes)
        s = fs.Session(1, 4, cfg)
        got = _drain(s)
        s.close()
    ranges = orc.batch_ranges(len(idx), bs)
    assert len(got) == len(ranges)
    grew = False
    for (x, _y, adjs, (start, stop)) in got:
        m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
        grew |= max(h.col.shape[0] for h in m.hops) > 300_000
        np.testing.assert_array_equal(x.cpu().numpy().reshape(-1), m.n_id)
        for (rp, cl, _e, size), hop in zip(adjs, m.hops):
            np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
    assert grew or sizes != [-1, -1]
    fs.clear_resident_cache()
