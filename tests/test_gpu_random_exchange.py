"""-m gpu: randomly drawn PARTITIONED cases through the native exchange (Session + DeviceDistributedPrefetcher on the in-process
transport, one thread per rank), every rank's batches bit for bit against the oracle.

What the hand-written cases of test_gpu_native_exchange.py fix and hypothesis varies here: the number of ranks, uneven and
EMPTY partitions, the share of a partition's rows that arrive in x_gpu / x_cpu, the VIP cache (none, empty, a few rows, every
remote vertex), the feature width (2-byte to 400-byte rows, so every vector width of the row movers), fan-outs (fast and
generic hops), batch sizes, batches per epoch, slots in flight, who issues the exchange, per-batch / per-group delivery and --
round 6 -- the transport (the exchange, or the P2P transport that reads remote rows in their owners' partitions) and whether the
records carry the assembled matrix or row references (checked through RowRefs.materialize()).
SPP_FUZZ_EXAMPLES=<n> runs more cases, SPP_FUZZ_RANDOM=1 draws fresh ones (tools/fuzz_long.sh)."""
import os
import sys
import threading

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402

T = torch.from_numpy
FANOUTS = [[15, 10, 5], [5, 5], [3], [32], [2, 2, 2, 2], [25, 15], [0, 4], [4, -1], [33]]
EXAMPLES = int(os.environ.get("SPP_FUZZ_EXAMPLES", "24"))
DERANDOMIZE = os.environ.get("SPP_FUZZ_RANDOM", "0") != "1"


def _graph(rng, n, mean_deg, zero_frac):
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    deg[rng.random(n) < zero_frac] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    return rowptr, rng.integers(0, n, rowptr[-1]).astype(np.int64)


def _offsets(rng, n, P, empty_parts):
    cuts = np.sort(rng.integers(0, n + 1, P - 1))
    off = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    for _ in range(empty_parts):                       # make a partition empty: its rank owns no vertex at all
        k = int(rng.integers(1, P))
        off[k] = off[k - 1]
    return np.maximum.accumulate(off)


def _rank(rank, P, comms, case, errors, sent):
    it = None
    from salient_plusplus_amd import fast_sampler as fs
    try:
        from oracle import oracle as orc
        from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
        from salient_plusplus_amd.fast_trainer.transferers import DeviceDistributedPrefetcher
        torch.cuda.set_device(0)
        p2p = case["transport"] == "p2p"
        if p2p:
            fs.set_p2p_peers(case["tables"])
        else:
            fs.set_native_comm(comms[rank])
        rowptr, col, x, y, off, sizes = case["rowptr"], case["col"], case["x"], case["y"], case["off"], case["sizes"]
        n = rowptr.shape[0] - 1
        lo, hi = int(off[rank]), int(off[rank + 1])
        rng = np.random.default_rng(case["seed"] * 31 + rank)
        cache = fs.Cache()
        if case["cache"] is not None:
            remote = np.setdiff1d(np.arange(n), np.arange(lo, hi))
            k = min(remote.size, int(round(case["cache"] * remote.size)))
            cv = np.sort(rng.choice(remote, size=k, replace=False)).astype(np.int64)
            cache = fs.Cache(rank, P, T(cv), T(x[cv].copy()))
        idx = case["idx"][rank]
        cut = int(round(case["gpu_share"] * (hi - lo)))
        # (P2P: the peers were given the resident copy of the rank's whole partition, handed in as x_gpu)
        x_gpu = case["x_gpu"][rank] if p2p else T(x[lo:hi][:cut].copy()).cuda()
        cfg = FastSamplerConfig(
            x_cpu=T(x[lo:hi][hi - lo if p2p else cut:].copy()), x_gpu=x_gpu, y=T(y).unsqueeze(-1), rowptr=T(rowptr), col=T(col),
            idx=T(idx), batch_size=case["bs"], sizes=sizes, skip_nonfull_batch=False, pin_memory=False, distributed=True,
            partition_book=fs.RangePartitionBook(rank, P, T(off)), cache=cache, force_exact_num_batches=True,
            exact_num_batches=case["nb"], count_remote_frequency=False, use_cache=case["cache"] is not None)
        ranges = orc.batch_ranges(len(idx), case["bs"], False, True, case["nb"])
        it = iter(FastSampler(2, case["slots"], cfg, row_refs=case["refs"]))
        assert it.session.native_exchange and it.session.p2p == p2p
        pre = DeviceDistributedPrefetcher([torch.device("cuda", 0)], it, True)
        held = [batch for (batch,) in pre]              # compared after the epoch: no host sync between batches
        assert len(held) == case["nb"]
        for k, batch in enumerate(held):
            start, stop = int(ranges[k][0]), int(ranges[k][1])
            m = orc.sample_batch(rowptr, col, idx, start, stop, sizes)
            assert (batch.idx_range.start, batch.idx_range.stop) == (start, stop)
            bx = batch.x
            if case["refs"]:
                assert isinstance(bx, fs.RowRefs) and tuple(bx.shape) == (len(m.n_id), x.shape[1])
                bx = bx.materialize()
            np.testing.assert_array_equal(bx.cpu().numpy().view(np.uint16), x[m.n_id].view(np.uint16))
            np.testing.assert_array_equal(batch.y.cpu().numpy().reshape(-1), y[m.n_id[:stop - start]])
            assert len(batch.adjs) == len(m.hops)
            for adj, hop in zip(batch.adjs, m.hops):
                rp, cl, _ = adj.adj_t.csr()
                np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
        sent[rank] = pre.NUMBER_OF_SENT_BYTES
        it.session.close()
    except BaseException as e:  # noqa: BLE001
        import traceback
        errors.append(f"rank {rank}: {e}\n{traceback.format_exc()}")
        if it is not None:
            it.session.close()
        if comms:
            comms[rank].close()     # wakes the peers out of the rendezvous
    finally:
        fs.set_native_comm(None)
        fs.set_p2p_peers(None)


@settings(max_examples=EXAMPLES, deadline=None, derandomize=DERANDOMIZE, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(60, 5000), mean_deg=st.floats(0.5, 30.0), zero_frac=st.floats(0.0, 0.4),
       P=st.sampled_from([2, 2, 3, 4, 5, 8]), empty_parts=st.integers(0, 1), gpu_share=st.sampled_from([0.0, 0.3, 1.0]),
       cache=st.sampled_from([None, 0.0, 0.05, 0.3, 1.0]), F=st.sampled_from([1, 2, 3, 4, 8, 20, 64, 100, 128, 200]),
       sizes=st.sampled_from(FANOUTS), bs=st.sampled_from([1, 5, 32, 128, 512]), nb=st.integers(1, 20),
       slots=st.sampled_from([1, 2, 6, 16, 64]), issue=st.sampled_from(["thread", "consumer"]), group_delivery=st.booleans(),
       transport=st.sampled_from(["local", "local", "p2p"]), refs=st.booleans())
def test_random_partitioned_case_against_the_oracle(seed, n, mean_deg, zero_frac, P, empty_parts, gpu_share, cache, F, sizes, bs, nb,
                                                    slots, issue, group_delivery, transport, refs):
    from salient_plusplus_amd import _native as nat
    nat.load()
    nat.require_device()
    from salient_plusplus_amd import fast_sampler as fs
    if os.environ.get("SPP_FUZZ_LOG"):          # the case about to run: the last line names the one that hung or crashed
        with open(os.environ["SPP_FUZZ_LOG"], "a") as f:
            f.write(repr(dict(seed=seed, n=n, mean_deg=mean_deg, zero_frac=zero_frac, P=P, empty_parts=empty_parts, gpu_share=gpu_share,
                              cache=cache, F=F, sizes=sizes, bs=bs, nb=nb, slots=slots, issue=issue, group_delivery=group_delivery,
                              transport=transport, refs=refs)) + "\n")
    rng = np.random.default_rng(seed)
    rowptr, col = _graph(rng, n, mean_deg, zero_frac)
    case = dict(seed=seed, rowptr=rowptr, col=col, sizes=list(sizes), bs=bs, nb=nb, slots=slots, cache=cache, gpu_share=gpu_share,
                x=rng.integers(0, 65536, (n, F), dtype=np.uint16).view(np.float16), y=rng.integers(0, 47, n).astype(np.int64),
                off=_offsets(rng, n, P, empty_parts),
                # every rank trains on its own seeds (any vertex, duplicates allowed), at least one per batch
                idx=[rng.integers(0, n, max(nb, bs * nb - int(rng.integers(0, bs)))).astype(np.int64) for _ in range(P)])
    case["transport"], case["refs"] = transport, refs
    old = {k: os.environ.get(k) for k in ("SPP_EXCHANGE_ISSUE", "SPP_GROUP_DELIVERY", "SPP_DIST_TRANSPORT")}
    os.environ["SPP_EXCHANGE_ISSUE"] = issue
    os.environ["SPP_GROUP_DELIVERY"] = "1" if group_delivery else "0"
    os.environ["SPP_DIST_TRANSPORT"] = "p2p" if transport == "p2p" else "rccl"
    try:
        comms = []
        if transport == "p2p":
            off = case["off"]
            case["x_gpu"] = [T(case["x"][int(off[r]):int(off[r + 1])].copy()).cuda() for r in range(P)]
            case["tables"] = [fs._resident.get_rows(t) if t.numel() else None for t in case["x_gpu"]]
            if all(t is None for t in case["tables"]):
                return
        else:
            comms = fs.NativeComm.local(P)
        errors, sent = [], {}
        ts = [threading.Thread(target=_rank, args=(r, P, comms, case, errors, sent)) for r in range(P)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(240)
        hung = [t for t in ts if t.is_alive()]
        for c in comms:
            c.close()
        assert not errors, "\n".join(errors)
        assert not hung, "rank thread hung"
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
