"""Isolated timings of the C-ABI kernels on one MI355X (development aid, not part of the product)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import _native as nat  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

L = nat.load()
nat.require_device()
dev = torch.device("cuda", 0)


def P(t):
    return C.c_void_p(t.data_ptr())


def timeit(fn, n=300, warm=50):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    what = sys.argv[1:] or ["mt", "gather", "sample"]
    if "mt" in what:
        for n in (15360, 163840, 1000000):
            out = torch.empty(n, dtype=torch.int32, device=dev)
            ms = timeit(lambda: L.spp_mt19937_fill(12345, 0, n, P(out), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            print(f"mt19937_fill n={n}: {ms*1e3:.1f} us  ({n/ms/1e6:.3f} G draws/s)", flush=True)
    wl = None
    if "gather" in what or "sample" in what:
        wl = make_workload(os.environ.get("WL", "S-products"), device=dev)
        torch.cuda.synchronize()
    if "gather" in what:
        N, F = wl.x.shape
        for U in (100_000, 770_000):
            idx = torch.randint(0, N, (U,), device=dev, dtype=torch.int64)
            idx32 = idx.to(torch.int32)
            out = torch.empty((U, F), dtype=wl.x.dtype, device=dev)
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            for name, ind, eb in (("i64", idx, 8), ("i32", idx32, 4)):
                ms = timeit(lambda: L.spp_gather_rows(P(wl.x), N, F * 2, P(ind), eb, U, U, P(out), st))
                print(f"gather U={U} F={F} idx={name}: {ms*1e3:.1f} us  alg {(U*(4*F+8))/ms/1e6:.1f} GB/s", flush=True)
            # the resident layout: rows padded to the 128-B fetch granule
            pad = torch.empty((N, 128), dtype=wl.x.dtype, device=dev)
            pad[:, :F].copy_(wl.x)
            ms = timeit(lambda: L.spp_gather_rows_strided(P(pad), N, F * 2, 256, P(idx32), 4, U, U, P(out), st))
            print(f"gather U={U} F={F} idx=i32 stride=256: {ms*1e3:.1f} us  alg {(U*(4*F+8))/ms/1e6:.1f} GB/s",
                  flush=True)
            del pad
            ms = timeit(lambda: torch.index_select(wl.x, 0, idx))
            print(f"torch.index_select U={U}: {ms*1e3:.1f} us  alg {(U*(4*F+8))/ms/1e6:.1f} GB/s", flush=True)
    if "sample" in what:
        cfg = nat.SamplerCfg()
        cfg.rowptr_dev, cfg.col_dev = wl.rowptr.data_ptr(), wl.col.data_ptr()
        cfg.num_nodes, cfg.nnz = wl.num_nodes, wl.col.numel()
        cfg.num_hops = len(wl.fanouts)
        for i, s in enumerate(wl.fanouts):
            cfg.sizes[i] = s
        cfg.max_batch, cfg.num_slots, cfg.device = wl.batch_size, 2, 0
        h = C.c_void_p()
        nat.check(L.spp_sampler_create(C.byref(cfg), C.byref(h)))
        seeds = wl.train_idx[:wl.batch_size].contiguous()
        cnt = nat.MfgCounts()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nat.check(L.spp_sampler_sample(h, 0, P(seeds), seeds.numel(), 17413, 0, st))
            t1 = time.perf_counter()
            nat.check(L.spp_sampler_wait(h, 0, C.byref(cnt)))
            t2 = time.perf_counter()
            print(f"sample: enqueue {1e6*(t1-t0):.0f} us, total latency {1e6*(t2-t0):.0f} us, U={cnt.num_nodes} "
                  f"E={[cnt.E[k] for k in range(cnt.num_hops)]} draws={cnt.draws}", flush=True)
        L.spp_sampler_destroy(h)


def chain_only():
    """Sampling throughput with the consumer side switched off (batches are dropped, never exported)."""
    wl = make_workload(os.environ.get("WL", "S-products"), device=dev)
    torch.cuda.synchronize()
    cfgs = ((24, 8), (16, 8), (32, 16), (12, 4))
    if os.environ.get("CHAIN_CFG"):          # e.g. CHAIN_CFG=16,8
        cfgs = (tuple(int(v) for v in os.environ["CHAIN_CFG"].split(",")),)
    for slots, group in cfgs:
        cfg = nat.SessionCfg()
        cfg.rowptr_dev, cfg.col_dev = wl.rowptr.data_ptr(), wl.col.data_ptr()
        cfg.num_nodes, cfg.nnz = wl.num_nodes, wl.col.numel()
        cfg.idx_dev, cfg.n_idx = wl.train_idx.data_ptr(), wl.train_idx.numel()
        cfg.batch_size, cfg.num_hops = wl.batch_size, len(wl.fanouts)
        for i, f in enumerate(wl.fanouts):
            cfg.sizes[i] = f
        cfg.force_exact_num_batches, cfg.exact_num_batches = 1, wl.train_idx.numel() // wl.batch_size
        cfg.max_items_in_queue, cfg.group_size, cfg.device = slots, group, 0
        part = None
        if os.environ.get("CHAIN_PARTS"):        # e.g. CHAIN_PARTS=8: ownership bucketing of rank 0 of 8 (+ a 10 % cache map)
            P_ = int(os.environ["CHAIN_PARTS"])
            part = nat.PartitionCfg()
            part.num_parts, part.rank = P_, 0
            for k in range(P_ + 1):
                part.offsets[k] = wl.num_nodes * k // P_
            if os.environ.get("CHAIN_CACHE", "1") != "0":
                cmap = torch.full((wl.num_nodes,), -1, dtype=torch.int32, device=dev)
                hit = torch.randperm(wl.num_nodes, device=dev)[: wl.num_nodes // (10 * P_)]
                cmap[hit] = torch.arange(hit.numel(), dtype=torch.int32, device=dev)
                part.use_cache, part.cache_map_dev, part.cache_map_len = 1, cmap.data_ptr(), wl.num_nodes
            cfg.part = C.pointer(part)
        h = C.c_void_p()
        nat.check(L.spp_session_create(C.byref(cfg), C.byref(h)))
        d = nat.BatchDesc()
        n = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while L.spp_session_next(h, C.byref(d)) == 1:
            n += 1
        dt = time.perf_counter() - t0
        print(f"chain only: slots={slots} group={group}: {n} batches, {dt/n*1e6:.1f} us/batch", flush=True)
        L.spp_session_destroy(h)


if __name__ == "__main__":
    if "chain" in sys.argv[1:]:
        chain_only()
        sys.exit(0)
    main()
