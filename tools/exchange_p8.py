"""The 8-rank composition of the feature exchange rehearsed on ONE GPU: P in-process ranks (threads, the in-process
transport: everything above send / recv / all-gather is the product path) at S-products scale with the planted
8-block locality, batch 1024, federated seeds, analytic VIP cache of 10 % of N/P.  Prints, per rank, the
local / cache-hit / fetched composition of its batches and the rows it served, and the per-batch time (ranks share
one GPU: NOT a performance number).  Run under rocprofv3 (--kernel-trace --stats, or one --pmc counter) for the
per-kernel figures of k_serve_rows / k_pack_remote_ids / k_gpart_* / k_deliver at that load (tools/exchange_p8.sh).
TRANSPORT=p2p: the opt-in P2P transport (remote rows read in the owner's partition: no exchange kernels at all);
ROW_REFS=1: the records carry RowRefs (one address per row + a contiguous copy of the received rows) instead of the
assembled matrix (verified through materialize()).
usage: [WL=S-papers CACHE_FRAC=0.1 CACHE_STRATEGY=vip TRANSPORT=local|p2p ROW_REFS=0|1] exchange_p8.py [P=8] [batches per rank=24] [epochs=2]"""
import json
import os
os.environ.setdefault("SPP_ALLOW_LOCAL_COMM", "1")   # rehearsal transport: opt-in
os.environ.setdefault("OMP_NUM_THREADS", "1")
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from salient_plusplus_amd import fast_sampler as fs  # noqa: E402
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig  # noqa: E402
from salient_plusplus_amd.fast_trainer.vip_cache import rank_remote_vertices  # noqa: E402
from salient_plusplus_amd.synthetic import make_workload  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 24
EPOCHS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
CACHE_FRAC = float(os.environ.get("CACHE_FRAC", "0.10"))          # of N / P rows, as bench.py --cache-frac
CACHE_STRATEGY = os.environ.get("CACHE_STRATEGY", "vip")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
wl = make_workload(os.environ.get("WL", "S-products-local"), device=dev)
N, F = wl.num_nodes, wl.x.size(1)
offsets = torch.linspace(0, N, P + 1).long()
offsets[-1] = N
off_dev = offsets.to(dev)
TRANSPORT = os.environ.get("TRANSPORT", "local")
ROW_REFS = os.environ.get("ROW_REFS", "0") != "0"
x_parts = [wl.x[int(offsets[r]):int(offsets[r + 1])].contiguous() for r in range(P)]
if TRANSPORT == "p2p":
    os.environ["SPP_DIST_TRANSPORT"] = "p2p"
    comms = []
    peer_tables = [fs._resident.get_rows(t) for t in x_parts]       # what the ranks' Sessions keep resident
else:
    comms = fs.NativeComm.local(P)
res, errors = {}, []
# every rank runs the same number of batches (force_exact_num_batches): the smallest pool of own training vertices decides
own = torch.bincount(torch.searchsorted(off_dev, wl.train_idx, right=True) - 1, minlength=P)
NB = min(NB, int(own.min()) // wl.batch_size)


def rank_main(r):
    try:
        torch.cuda.set_device(0)
        if TRANSPORT == "p2p":
            fs.set_p2p_peers(peer_tables)
        else:
            fs.set_native_comm(comms[r])
        lo, hi = int(offsets[r]), int(offsets[r + 1])
        pb = fs.RangePartitionBook(r, P, offsets)
        bs = wl.batch_size
        mine = wl.train_idx[(wl.train_idx >= lo) & (wl.train_idx < hi)].contiguous()      # federated seeds
        nb = NB
        cv = rank_remote_vertices(CACHE_STRATEGY, pb, N, int(CACHE_FRAC * N / P), rowptr=wl.rowptr, col=wl.col, train_idx=mine,
                                  fanouts=wl.fanouts, batch_size=bs).sort().values
        cache = fs.Cache(r, P, cv, wl.x[cv].contiguous())
        in_cache = torch.zeros(N, dtype=torch.bool, device=dev)
        in_cache[cv] = True
        comp = torch.zeros(3, dtype=torch.int64, device=dev)          # local, cache hits, fetched
        by_owner = torch.zeros(P, dtype=torch.int64, device=dev)      # rows fetched from each owner
        for epoch in range(EPOCHS):
            g = torch.Generator()
            g.manual_seed(1000 * epoch + r)
            idx = mine[torch.randperm(mine.numel(), generator=g).to(dev)][:nb * bs].contiguous()
            cfg = FastSamplerConfig(
                x_cpu=torch.empty((0, F), dtype=wl.x.dtype), x_gpu=x_parts[r], y=wl.y.unsqueeze(-1),
                rowptr=wl.rowptr, col=wl.col, idx=idx, batch_size=bs, sizes=wl.fanouts, skip_nonfull_batch=False,
                pin_memory=False, distributed=True, partition_book=pb, cache=cache, force_exact_num_batches=True,
                exact_num_batches=nb, count_remote_frequency=False, use_cache=True)
            it = iter(FastSampler(2, int(os.environ.get("SPP_MAX_SLOTS_DIST", "64")), cfg, row_refs=ROW_REFS))
            assert it.session.native_exchange and it.session.p2p == (TRANSPORT == "p2p")
            t0 = time.perf_counter()
            n = 0
            ok = True
            for b in it:                                              # the native records carry x (assembled) and n_id
                n += 1
                if os.environ.get("VERIFY", "1") != "0":
                    ok = ok and bool(torch.equal(b.x.materialize() if ROW_REFS else b.x, wl.x[b.n_id]))
                loc = (b.n_id >= lo) & (b.n_id < hi)
                hit = ~loc & in_cache[b.n_id]
                rem = ~loc & ~hit
                comp += torch.stack([loc.sum(), hit.sum(), rem.sum()])
                by_owner += torch.bincount(torch.searchsorted(off_dev, b.n_id[rem], right=True) - 1, minlength=P)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            res[(r, epoch)] = dict(batches=n, us_per_batch=dt / max(n, 1) * 1e6, bit_exact=ok,
                                   exchange_bytes=it.session.exchange_bytes())
            it.session.close()
        res[(r, "comp")] = comp.cpu().tolist()
        res[(r, "by_owner")] = by_owner.cpu().tolist()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {r}: {e}\n{traceback.format_exc()}")
        if comms:
            comms[r].close()
    finally:
        fs.set_native_comm(None)
        fs.set_p2p_peers(None)


ts = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
for t in ts:
    t.start()
for t in ts:
    t.join()
for c in comms:
    c.close()
if errors:
    print("\n".join(errors))
    sys.exit(1)
tot = [0, 0, 0]
served = [0] * P
nbatches = 0
for r in range(P):
    c = res[(r, "comp")]
    tot = [a + b for a, b in zip(tot, c)]
    for m, v in enumerate(res[(r, "by_owner")]):
        served[m] += v
    nb_r = sum(res[(r, e)]["batches"] for e in range(EPOCHS))
    nbatches += nb_r
    s = sum(c)
    print(f"rank {r}: {nb_r} batches, {s / nb_r / 1e3:.0f} k nodes per batch: local {100 * c[0] / s:.1f} %, cache hits {100 * c[1] / s:.1f} %, "
          f"fetched {100 * c[2] / s:.1f} % ({c[2] / nb_r / 1e3:.0f} k rows = {c[2] / nb_r * F * 2 / 1e6:.1f} MB per batch); "
          f"bit exact {all(res[(r, e)]['bit_exact'] for e in range(EPOCHS))}; " +
          ", ".join(f"epoch {e}: {res[(r, e)]['us_per_batch']:.0f} us/batch" for e in range(EPOCHS)))
s = sum(tot)
summary = {"P": P, "transport": TRANSPORT, "row_refs": ROW_REFS, "cache_frac": CACHE_FRAC, "cache_strategy": CACHE_STRATEGY, "cache_rows_per_rank": int(CACHE_FRAC * N / P),
           "exchange_bytes_per_batch_and_rank": [sum(res[(r, e)]["exchange_bytes"][k] for r in range(P) for e in range(EPOCHS)) / max(1, nbatches)
                                                  for k in (0, 1)],
           "workload": wl.name, "F": F, "row_bytes": 2 * F, "batches_all_ranks": nbatches, "rows_delivered": s,
           "rows_local": tot[0], "rows_cache": tot[1], "rows_fetched": tot[2], "rows_served": sum(served),
           "frac_local": tot[0] / s, "frac_cache": tot[1] / s, "frac_fetched": tot[2] / s}
print("EXCHANGE_P8 " + json.dumps(summary), flush=True)
