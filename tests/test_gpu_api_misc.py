"""-m gpu: the rest of the fast_sampler.Session surface (fast_sampler.cpp:1310-1338): statistics
properties, remote-frequency counting for the `simulation` cache strategy, the host-slice API,
full_sample / FastPreSampler, error behaviour."""
import datetime
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


def base_cfg(fs, g, **kw):
    cfg = fs.Config()
    cfg.x_cpu, cfg.y = T(g["x"]), T(g["y"]).unsqueeze(-1)
    cfg.rowptr, cfg.col, cfg.idx = T(g["rowptr"]), T(g["col"]), T(g["idx"])
    cfg.batch_size, cfg.sizes = 64, [15, 10, 5]
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


def test_session_properties_and_errors(fs, graph_a):
    with pytest.raises(RuntimeError, match="max_items_in_queue"):
        fs.Session(2, 0, base_cfg(fs, graph_a))
    s = fs.Session(2, 4, base_cfg(fs, graph_a))
    assert s.num_total_batches == 4 and s.num_consumed_batches == 0
    assert isinstance(s.total_blocked_dur, datetime.timedelta) and s.total_blocked_occasions >= 0
    assert s.config.batch_size == 64
    n = polls = 0
    while s.num_consumed_batches < s.num_total_batches:    # the reference's polling pattern (:783-799)
        polls += 1
        if s.try_get_batch() is None:                      # not ready yet: never blocks
            continue
        n += 1
        assert s.num_consumed_batches == n == s.approx_num_complete_batches
    assert n == 4 and polls >= 4
    assert s.try_get_batch() is None and s.blocking_get_batch() is None       # end of data stays None
    s.close()
    bad = base_cfg(fs, graph_a, force_exact_num_batches=True, exact_num_batches=1000)
    with pytest.raises(RuntimeError):
        fs.Session(1, 2, bad)                                # n/k == 0: the reference asserts


def test_remote_frequency_counting(fs, graph_a):
    """count_remote_frequency (fast_sampler.cpp:1093-1103) + reduce / top-n (:835-880)."""
    from oracle import oracle as orc
    offs = np.array([0, 1400, 3000], dtype=np.int64)
    x = graph_a["x"]
    cfg = base_cfg(fs, graph_a, x_cpu=T(x[:0].copy()), x_gpu=T(x[:1400].copy()).cuda(), distributed=True,
                   partition_book=fs.RangePartitionBook(0, 2, T(offs)), force_exact_num_batches=True,
                   exact_num_batches=3, count_remote_frequency=True, use_cache=False)
    s = fs.Session(2, 4, cfg)
    freq = {}
    ranges = orc.batch_ranges(200, 64, False, True, 3)
    b = 0
    while True:
        proto = s.blocking_get_batch_distributed()
        if proto is None:
            break
        m = orc.sample_batch(graph_a["rowptr"], graph_a["col"], graph_a["idx"], int(ranges[b][0]), int(ranges[b][1]),
                             [15, 10, 5])
        for v in m.n_id[m.n_id >= 1400]:
            freq[int(v)] = freq.get(int(v), 0) + 1
        b += 1
    s.reduce_multithreaded_frequency_counts()
    f = s.remote_frequency_tensor.numpy()
    v = s.remote_vertices_ordered_by_freq.numpy()
    assert f.shape[0] == len(freq) and (np.diff(f) <= 0).all()          # sorted descending
    assert all(freq[int(vv)] == int(ff) for vv, ff in zip(v, f))
    top = s.get_n_most_freq_remote_vertices(10).numpy()
    kth = sorted(freq.values(), reverse=True)[9]
    assert all(freq[int(t)] >= kth for t in top)
    s.close()


def test_host_slice_api_degenerates(fs, graph_a):
    """async_slice_tensors / wait / get (fast_sampler.cpp:716-775): no feature row lives in host
    memory, so the host gather returns empty features and positions only."""
    s = fs.Session(1, 2, base_cfg(fs, graph_a))
    ids = [torch.tensor([3, -1, 7, -5]), torch.tensor([], dtype=torch.int64)]
    s.async_slice_tensors(ids, 0)
    s.wait_slice_tensors()
    out = s.get_slice_tensors()
    assert len(out) == 2 and out[0][0].numel() == 0
    assert out[0][1].tolist() == [0, 2] and out[0][2].tolist() == [1, 3]
    assert out[1][1].numel() == 0 and out[1][2].numel() == 0
    s.close()


def test_full_sample_and_presampler(fs, graph_a, golden_dir):
    from salient_plusplus_amd.fast_trainer.samplers import FastPreSampler, FastSamplerConfig
    g = np.load(os.path.join(golden_dir, "mfg_a_s15_10_5.npz"))
    n = graph_a["rowptr"].shape[0] - 1
    ids_as_x = np.arange(n, dtype=np.int64).reshape(n, 1)
    cfg = FastSamplerConfig(
        x_cpu=T(ids_as_x), x_gpu=torch.empty(0), y=T(graph_a["y"]).unsqueeze(-1), rowptr=T(graph_a["rowptr"]),
        col=T(graph_a["col"]), idx=T(graph_a["idx"]), batch_size=64, sizes=[15, 10, 5], skip_nonfull_batch=False,
        pin_memory=False, distributed=False, partition_book=None, cache=fs.Cache(), force_exact_num_batches=False,
        exact_num_batches=0, count_remote_frequency=False, use_cache=False)
    pre = FastPreSampler(cfg)
    assert len(pre) == 4
    batches = list(iter(pre))
    assert len(batches) == 4
    for b, batch in enumerate(batches):
        np.testing.assert_array_equal(batch.x.cpu().numpy().reshape(-1), g[f"b{b}_n_id"])
        np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), g[f"b{b}_y"])


def test_sampler_workspace_is_reused_across_sessions(fs, graph_a):
    """The pooled sampler (the counterpart of the reference's process-global worker pool,
    fast_sampler.cpp:512-513) outlives Sessions: a second epoch must not allocate a new one."""
    from salient_plusplus_amd.fast_sampler import _SamplerPool
    cfg = base_cfg(fs, graph_a)
    s1 = fs.Session(2, 4, cfg)
    h1 = s1._pool_entry[0].value
    while s1.blocking_get_batch() is not None:
        pass
    s1.close()
    s2 = fs.Session(2, 4, cfg)
    assert s2._pool_entry[0].value == h1
    first = s2.blocking_get_batch()
    assert first[3] == (0, 64)
    s2.close()
    assert sum(len(v) for v in _SamplerPool._pool.values()) >= 1
