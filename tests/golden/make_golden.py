#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the UNMODIFIED reference.

The reference native module is built by ``oracle/build_ref.sh`` (direct clang++
invocation on /root/reference/fast_sampler/*.cpp, outputs only under
oracle/_ref/).  This script imports that binary module, feeds it small seeded
synthetic inputs and stores inputs + outputs as .npz fixtures.  A fixture is
data only; no reference source text is stored.

  part "cpu"  : everything the reference can do without a GPU runtime
                (Session non-distributed path, free functions,
                RangePartitionBook, Cache.nid_is_cached, serial_index,
                to_row_major).  Run in the build container:
                    python tests/golden/make_golden.py --part cpu
  part "dist" : the distributed worker branch and Cache.nid2cachenid, which
                allocate pinned host memory (fast_sampler.cpp:1026,...;
                range_partition_book.cpp:188) and therefore need a GPU runtime.
                Run on the GPU box (the prebuilt oracle/_ref/*.so travels there):
                    gpurun -- python tests/golden/make_golden.py --part dist --out gpurun_out/golden
                then copy gpurun_out/golden/*.npz into tests/golden/.
"""
import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_reference():
    import torch  # noqa: F401  (must be imported before the extension)
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    import fast_sampler as ref
    assert ref.__file__.startswith(os.path.join(ROOT, "oracle", "_ref")), ref.__file__
    return ref


def make_graph(seed: int, n: int, coalesced: bool = True):
    """Mixed-degree CSR: some isolated nodes, some deg<=5, a bulk of 6..40, some hubs."""
    rng = np.random.default_rng(seed)
    kind = rng.random(n)
    deg = np.where(kind < 0.10, 0,
          np.where(kind < 0.40, rng.integers(1, 6, n),
          np.where(kind < 0.85, rng.integers(6, 41, n), rng.integers(41, 301, n)))).astype(np.int64)
    deg = np.minimum(deg, n - 1)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    cols = []
    for v in range(n):
        d = int(deg[v])
        if coalesced:
            c = rng.choice(n, size=d, replace=False)
            c.sort()
        else:
            c = rng.integers(0, n, size=d)      # duplicates and self loops allowed, unsorted
        cols.append(c.astype(np.int64))
        rowptr[v + 1] = rowptr[v] + d
    col = np.concatenate(cols) if cols else np.zeros(0, dtype=np.int64)
    return rowptr, col


def new_config(ref, torch, **kw):
    cfg = ref.Config()
    defaults = dict(x_gpu=torch.empty(0), skip_nonfull_batch=False, pin_memory=False,
                    distributed=False, force_exact_num_batches=False, exact_num_batches=0,
                    count_remote_frequency=False, use_cache=False)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(cfg, k, v)
    return cfg


def drain(session):
    out = []
    while True:
        b = session.blocking_get_batch()
        if b is None:
            break
        out.append(b)
    out.sort(key=lambda b: b[3][0])
    return out


def part_cpu(ref, out_dir):
    import torch
    T = torch.from_numpy

    # ---------------- known-answer MT19937 vectors ----------------
    # The reference does not export its generator; the stream is specified by the C++
    # standard (std::mt19937 == MT19937 with init_genrand seeding).  Independent
    # source: numpy's MT19937 bit generator with legacy (init_genrand) seeding.
    seeds = [5489, 5, 64 * 17 + 5, 1024 * 17 + 5, 0xFFFFFFFF]
    mt = {}
    for s in seeds:
        bg = np.random.MT19937()
        bg._legacy_seeding(s)
        mt[f"seed_{s}"] = bg.random_raw(2000).astype(np.uint32)
    bg = np.random.MT19937()
    bg._legacy_seeding(5489)
    raw = bg.random_raw(10000)
    assert int(raw[9999]) == 4123659995      # [rand.predef] 10000th output of default mt19937
    np.savez_compressed(os.path.join(out_dir, "mt19937.npz"), seeds=np.array(seeds, dtype=np.int64), **mt)

    # ---------------- MFG fixtures through the Session worker path ----------------
    n = 3000
    rowptr, col = make_graph(1234, n, coalesced=True)
    rng = np.random.default_rng(99)
    idx = rng.permutation(n)[:200].astype(np.int64)
    idx[17] = idx[3]            # duplicated seeds: map keeps the LAST position (sample_cpu.hpp:13-19)
    idx[150] = idx[149]
    y = rng.integers(0, 47, size=n).astype(np.int64)
    xh = rng.standard_normal((n, 7)).astype(np.float16)
    ids_as_x = np.arange(n, dtype=np.int64).reshape(n, 1)
    np.savez_compressed(os.path.join(out_dir, "graph_a.npz"), rowptr=rowptr, col=col, idx=idx, y=y, x=xh)

    def run_session(sizes, batch_size, x, yy, the_idx, threads=2, **kw):
        cfg = new_config(ref, torch, x_cpu=T(x), y=T(yy).unsqueeze(-1) if yy is not None else None,
                         rowptr=T(rowptr), col=T(col), idx=T(the_idx), batch_size=batch_size,
                         sizes=list(sizes), **kw)
        s = ref.Session(threads, 8, cfg)
        nb = s.num_total_batches
        got = drain(s)
        assert len(got) == nb
        return got

    cases = {"s15_10_5": [15, 10, 5], "s20_20_20": [20, 20, 20], "sall": [-1], "s25_15": [25, 15],
             "s3_all": [3, -1], "s1": [1], "s0_2": [0, 2]}
    for name, sizes in cases.items():
        got = run_session(sizes, 64, ids_as_x, y, idx)
        d = {"sizes": np.array(sizes, dtype=np.int64), "num_batches": np.array(len(got))}
        for bi, (x_s, y_s, adjs, rng_) in enumerate(got):
            d[f"b{bi}_range"] = np.array(rng_, dtype=np.int64)
            d[f"b{bi}_n_id"] = x_s.numpy().reshape(-1)
            d[f"b{bi}_y"] = y_s.numpy().reshape(-1)
            for hi, (rp, cl, e_id, sz) in enumerate(adjs):
                assert e_id.numel() == 0 and e_id.dtype == torch.int64
                d[f"b{bi}_h{hi}_rowptr"] = rp.numpy()
                d[f"b{bi}_h{hi}_col"] = cl.numpy()
                d[f"b{bi}_h{hi}_size"] = np.array(sz, dtype=np.int64)
        np.savez_compressed(os.path.join(out_dir, f"mfg_a_{name}.npz"), **d)

    # fp16 feature slice + labels for one configuration (serial_index, fast_sampler.cpp:1006-1010)
    got = run_session([15, 10, 5], 64, xh, y, idx)
    d = {}
    for bi, (x_s, y_s, adjs, rng_) in enumerate(got):
        d[f"b{bi}_x"] = x_s.numpy()
        d[f"b{bi}_y"] = y_s.numpy()
        d[f"b{bi}_range"] = np.array(rng_, dtype=np.int64)
    np.savez_compressed(os.path.join(out_dir, "slice_a_s15_10_5.npz"), **d)

    # thread-count independence (SURVEY 8(b) threading): 1 vs 4 workers give identical batches
    g1 = run_session([15, 10, 5], 64, ids_as_x, y, idx, threads=1)
    g4 = run_session([15, 10, 5], 64, ids_as_x, y, idx, threads=4)
    for a, b in zip(g1, g4):
        assert torch.equal(a[0], b[0]) and a[3] == b[3]

    # non-coalesced graph with duplicate columns and self loops
    rowptr_b, col_b = make_graph(77, 500, coalesced=False)
    idx_b = np.random.default_rng(5).permutation(500)[:96].astype(np.int64)
    save_rowptr, save_col = rowptr, col
    rowptr, col = rowptr_b, col_b
    got = run_session([4, 3, 2], 32, np.arange(500, dtype=np.int64).reshape(-1, 1), None, idx_b)
    d = {"rowptr": rowptr_b, "col": col_b, "idx": idx_b, "sizes": np.array([4, 3, 2], dtype=np.int64),
         "num_batches": np.array(len(got))}
    for bi, (x_s, y_s, adjs, rng_) in enumerate(got):
        assert y_s is None
        d[f"b{bi}_range"] = np.array(rng_, dtype=np.int64)
        d[f"b{bi}_n_id"] = x_s.numpy().reshape(-1)
        for hi, (rp, cl, e_id, sz) in enumerate(adjs):
            d[f"b{bi}_h{hi}_rowptr"] = rp.numpy()
            d[f"b{bi}_h{hi}_col"] = cl.numpy()
            d[f"b{bi}_h{hi}_size"] = np.array(sz, dtype=np.int64)
    np.savez_compressed(os.path.join(out_dir, "mfg_b_s4_3_2.npz"), **d)
    rowptr, col = save_rowptr, save_col

    # ---------------- batch range tables (fast_sampler.cpp:587-627) ----------------
    d = {}
    tiny_x = np.zeros((n, 1), dtype=np.int64)
    for (nn, k) in [(300, 4), (1000, 7), (5, 5), (200, 3), (64, 1)]:
        the_idx = np.arange(nn, dtype=np.int64) % n
        got = run_session([1], 17, tiny_x, None, the_idx, force_exact_num_batches=True, exact_num_batches=k)
        d[f"exact_{nn}_{k}"] = np.array([g[3] for g in got], dtype=np.int64)
    for (nn, bs, skip) in [(200, 64, False), (200, 64, True), (128, 64, True), (10, 64, False)]:
        the_idx = np.arange(nn, dtype=np.int64) % n
        got = run_session([1], bs, tiny_x, None, the_idx, skip_nonfull_batch=skip)
        d[f"plain_{nn}_{bs}_{int(skip)}"] = np.array([g[3] for g in got], dtype=np.int64).reshape(-1, 2)
    np.savez_compressed(os.path.join(out_dir, "batch_ranges.npz"), **d)

    # ---------------- free functions on the calling thread ----------------
    # `gen` is thread_local and default-constructed (seed 5489) on the Python thread; its state
    # carries over from call to call, so the call ORDER below is part of the fixture.
    idx10 = idx[:10].copy()
    d = {"idx": idx10}
    rp1, c1, n1, e1 = ref.sample_adj(T(rowptr), T(col), T(idx10), 5, False)
    assert n1.dtype == torch.int32          # quirk: free sample_adj returns int32 n_id
    d.update(call0_rowptr=rp1.numpy(), call0_col=c1.numpy(), call0_n_id=n1.numpy().astype(np.int64))
    rp2, c2, n2, e2 = ref.sample_adj(T(rowptr), T(col), T(idx10), 4, True)
    d.update(call1_rowptr=rp2.numpy(), call1_col=c2.numpy(), call1_n_id=n2.numpy().astype(np.int64))
    n3, adjs3 = ref.multilayer_sample(T(idx10), [3, 2], T(rowptr), T(col))
    assert n3.dtype == torch.int64
    d.update(call2_n_id=n3.numpy())
    for hi, (rp, cl, e_id, sz) in enumerate(adjs3):
        d[f"call2_h{hi}_rowptr"] = rp.numpy()
        d[f"call2_h{hi}_col"] = cl.numpy()
        d[f"call2_h{hi}_size"] = np.array(sz, dtype=np.int64)
    rp4, c4, n4, e4 = ref.sample_adj(T(rowptr), T(col), T(idx10), -1, False)
    d.update(call3_rowptr=rp4.numpy(), call3_col=c4.numpy(), call3_n_id=n4.numpy().astype(np.int64))
    np.savez_compressed(os.path.join(out_dir, "free_functions.npz"), **d)

    # ---------------- serial_index / to_row_major ----------------
    sel = rng.integers(0, n, size=50).astype(np.int64)
    d = {"sel": sel}
    d["half_all"] = ref.serial_index(T(xh), T(sel)).numpy()
    d["half_n7"] = ref.serial_index(T(xh), T(sel), 7).numpy()[:7]     # rows >= n_idx are uninitialised
    xf = rng.standard_normal((n, 5)).astype(np.float32)
    d["xf"] = xf
    d["float_all"] = ref.serial_index(T(xf), T(sel)).numpy()
    d["long_n3"] = ref.serial_index(T(y).unsqueeze(-1), T(sel), 3).numpy()
    cm = torch.arange(12, dtype=torch.float32).reshape(4, 3).t()      # 3x4 column-major view
    d["trm_in_storage"] = np.arange(12, dtype=np.float32)
    d["trm_out"] = ref.to_row_major(cm).numpy()
    np.savez_compressed(os.path.join(out_dir, "serial_index.npz"), **d)

    # ---------------- RangePartitionBook / Cache.nid_is_cached ----------------
    offsets = np.array([0, 700, 1500, 2100, 3000], dtype=np.int64)
    edge_ids = np.array([0, 1, 699, 700, 701, 1499, 1500, 2099, 2100, 2999], dtype=np.int64)
    pb = ref.RangePartitionBook(2, 4, T(offsets))
    d = {"offsets": offsets, "nids": edge_ids,
         "partid": pb.nid2partid(T(edge_ids)).numpy(),
         "localnid_p2": pb.nid2localnid(T(edge_ids), 2).numpy(),
         "partid2nids_1": pb.partid2nids(1).numpy()}
    cached_vertices = np.array([5, 2999, 1500, 42, 5, 800], dtype=np.int64)   # 5 twice: last index wins
    cache = ref.Cache(2, 4, T(cached_vertices), torch.zeros((6, 4), dtype=torch.float16))
    probe = np.array([5, 6, 2999, 1500, 0, 42, 800, 801], dtype=np.int64)
    d.update(cached_vertices=cached_vertices, probe=probe, is_cached=cache.nid_is_cached(T(probe)).numpy())
    np.savez_compressed(os.path.join(out_dir, "partition_book.npz"), **d)
    print("[make_golden] cpu part written to", out_dir)


def part_dist(ref, out_dir):
    """Distributed worker branch (fast_sampler.cpp:1017-1262).  Needs a GPU runtime (pinned memory)."""
    import torch
    T = torch.from_numpy
    g = np.load(os.path.join(HERE, "graph_a.npz"))
    rowptr, col, idx, y, xh = g["rowptr"], g["col"], g["idx"], g["y"], g["x"]
    n = rowptr.shape[0] - 1
    d = {}

    # Cache.nid2cachenid
    cached_vertices = np.array([5, 2999, 1500, 42, 5, 800], dtype=np.int64)
    cache0 = ref.Cache(2, 4, T(cached_vertices), torch.zeros((6, 4), dtype=torch.float16))
    probe = np.array([5, 2999, 1500, 42, 800], dtype=np.int64)
    d["c2c_cached_vertices"] = cached_vertices
    d["c2c_probe"] = probe
    d["c2c_out"] = cache0.nid2cachenid(T(probe)).numpy()

    layouts = {2: np.array([0, 1400, 3000], dtype=np.int64),
               4: np.array([0, 700, 1500, 2100, 3000], dtype=np.int64)}
    crng = np.random.default_rng(4242)
    cfgs = []
    for P, offs in layouts.items():
        for rank in (0, 1):
            own = int(offs[rank + 1] - offs[rank])
            for G in (0, 100, own):
                for use_cache in (False, True):
                    cfgs.append((P, offs, rank, G, use_cache))
    caches = {}
    for ci, (P, offs, rank, G, use_cache) in enumerate(cfgs):
        own_lo, own_hi = int(offs[rank]), int(offs[rank + 1])
        # split the way the driver does (base.py:107-116): torch slices keep row-major strides
        # even when one side is empty (a numpy copy of an empty slice trips the reference's
        # "input must be 2D row-major tensor" check, fast_sampler.cpp:243-245)
        x_local = T(xh[own_lo:own_hi].copy())
        x_gpu = x_local[:G]
        x_cpu = x_local[G:]
        if use_cache:
            key = (P, rank)
            if key not in caches:     # each Cache leaks ~1 GB of lookup tables: build few
                remote = np.setdiff1d(np.arange(n), np.arange(own_lo, own_hi))
                cv = crng.choice(remote, size=300, replace=False).astype(np.int64)
                caches[key] = (cv, ref.Cache(rank, P, T(cv), T(xh[cv].copy())))
            cv, cache = caches[key]
        else:
            cv, cache = np.zeros(0, dtype=np.int64), ref.Cache()
        cfg = ref.Config()
        cfg.x_cpu = x_cpu
        cfg.x_gpu = x_gpu
        cfg.y = T(y).unsqueeze(-1)
        cfg.rowptr = T(rowptr)
        cfg.col = T(col)
        cfg.idx = T(idx)
        cfg.batch_size = 64
        cfg.sizes = [15, 10, 5]
        cfg.skip_nonfull_batch = False
        cfg.pin_memory = False
        cfg.distributed = True
        cfg.partition_book = ref.RangePartitionBook(rank, P, T(offs))
        cfg.cache = cache
        cfg.force_exact_num_batches = True
        cfg.exact_num_batches = 3
        cfg.count_remote_frequency = False
        cfg.use_cache = use_cache
        s = ref.Session(2, 8, cfg)
        tag = f"c{ci}"
        d[f"{tag}_meta"] = np.array([P, rank, G, int(use_cache)], dtype=np.int64)
        d[f"{tag}_offsets"] = offs
        d[f"{tag}_cached_vertices"] = cv
        bi = 0
        while True:
            b = s.blocking_get_batch_distributed()
            if b is None:
                break
            d[f"{tag}_b{bi}_range"] = np.array(b.idx_range, dtype=np.int64)
            for m, t in enumerate(b.partition_nids):
                d[f"{tag}_b{bi}_part{m}"] = t.numpy().copy()
            d[f"{tag}_b{bi}_cached_nids"] = b.cached_nids.numpy().copy()
            d[f"{tag}_b{bi}_perm"] = b.perm_partition_to_mfg.numpy().copy()
            d[f"{tag}_b{bi}_cpu_feats"] = b.sliced_cpu_features.numpy().copy()
            d[f"{tag}_b{bi}_labels"] = b.sliced_cpu_labels.numpy().copy()
            # the MFG itself (b.adjs) is the same sampler output already pinned by mfg_a_*.npz and
            # is not stored again; the concat + perm identity below ties the two together
            ids = torch.cat(list(b.partition_nids) + ([T(cv)[b.cached_nids]] if use_cache else []))
            d[f"{tag}_b{bi}_n_id"] = ids[b.perm_partition_to_mfg].numpy().copy()
            bi += 1
        d[f"{tag}_num_batches"] = np.array(bi)
        del s
    d["num_cfgs"] = np.array(len(cfgs))
    np.savez_compressed(os.path.join(out_dir, "distributed.npz"), **d)
    print("[make_golden] dist part written to", out_dir)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", choices=["cpu", "dist", "all"], default="cpu")
    ap.add_argument("--out", default=HERE)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ref = load_reference()
    if a.part in ("cpu", "all"):
        part_cpu(ref, a.out)
    if a.part in ("dist", "all"):
        part_dist(ref, a.out)


if __name__ == "__main__":
    main()
