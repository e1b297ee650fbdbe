"""-m gpu: the configuration-dependent code paths of the sampling chain give the same bits as the default
one.  Each case runs in a fresh interpreter (the knobs are read once per process) and compares every batch
of a short epoch with the oracle:
  * SPP_RNG_ARENA_MB=0   -- no epoch arena: mt19937 streams generated per group into the slots' ping-pong buffers;
  * SPP_XCD_AFFINITY=0   -- batch-major instead of batch-interleaved workgroup ids;
  * SPP_COL32=0          -- the int64 neighbour array (picks at positions >= 29 read it; the row stubs still serve the rest);
  * SPP_ROW_STUBS=0      -- no row stubs: degrees from rowptr, cooperative reads of the int32 neighbour array;
  * both off             -- lane-private reads of the int64 array;
  * SPP_GROUP_SIZE=3     -- ragged groups, slot-sets of 3;
  * SPP_DEDUP_BUCKET=64  -- many small dedup buckets (coarse/fine bucket runs, 2^11-slot LDS tables);
  * SPP_DEG_TAGS=0       -- plain neighbour ids: the degree pass of hops >= 1 reads the stub headers instead of the tags the nodes bring along;
  * SPP_GROUP_DELIVERY=1 -- one delivery launch per sampling group instead of one per batch;
  * SPP_WHATIF_DUP=...   -- the measurement aid that launches the idempotent kernels twice changes nothing;
  * SPP_STREAM_PRIORITY=low, SPP_GROUP_SIZE=16 -- the data path's streams below the consumer's, the largest group;
  * SPP_DEDUP_PREREAD=1  -- the dedup table is read before every compare-and-swap (the form until round 3);
  * SPP_GROUP_FETCH=0    -- batch-at-a-time calls (spp_session_next / spp_session_export) instead of fetching the group
                            as a whole and exporting its members one by one."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

_CHILD = r"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, %r)
from oracle import oracle as orc
from salient_plusplus_amd import fast_sampler as fs
from salient_plusplus_amd.fast_trainer.samplers import FastSampler, FastSamplerConfig
from salient_plusplus_amd.fast_trainer.transferers import DevicePrefetcher
g = np.load(os.path.join(%r, "tests", "golden", "graph_a.npz"))
g = {k: g[k] for k in g.files}
T = torch.from_numpy
sizes = [15, 10, 5]
idx = g["idx"]
dev = torch.device("cuda", 0)
for epoch in range(2):                       # the second epoch reuses the pooled sampler (and the arena, if any)
    cfg = FastSamplerConfig(
        x_cpu=T(g["x"]), x_gpu=torch.empty(0), y=T(g["y"]).unsqueeze(-1), rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx),
        batch_size=24, sizes=sizes, skip_nonfull_batch=False, pin_memory=False, distributed=False, partition_book=None,
        cache=fs.Cache(), force_exact_num_batches=False, exact_num_batches=0, count_remote_frequency=False, use_cache=False)
    ranges = orc.batch_ranges(len(idx), 24, False, False, 0)
    n = 0
    for (b,) in DevicePrefetcher([dev], iter(FastSampler(2, 12, cfg))):
        start, stop = int(ranges[n][0]), int(ranges[n][1])
        m = orc.sample_batch(g["rowptr"], g["col"], idx, start, stop, sizes)
        np.testing.assert_array_equal(b.x.cpu().numpy().view(np.uint16), g["x"][m.n_id].view(np.uint16))
        for adj, hop in zip(b.adjs, m.hops):
            rp, cl, _ = adj.adj_t.csr()
            np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
            np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
        n += 1
    assert n == len(ranges), (n, len(ranges))
print("CHILD_OK", n)
""" % (ROOT, ROOT)


@pytest.mark.parametrize("env", [
    {"SPP_RNG_ARENA_MB": "0"},
    {"SPP_XCD_AFFINITY": "0"},
    {"SPP_COL32": "0"},
    {"SPP_ROW_STUBS": "0"},
    {"SPP_ROW_STUBS": "0", "SPP_COL32": "0"},
    {"SPP_GROUP_SIZE": "3"},
    {"SPP_DEDUP_BUCKET": "64"},
    {"SPP_DEG_TAGS": "0"},
    {"SPP_DEG_TAGS": "1", "SPP_GROUP_SIZE": "2", "SPP_DEDUP_BUCKET": "64"},
    {"SPP_GROUP_DELIVERY": "1"},
    {"SPP_GROUP_DELIVERY": "1", "SPP_GROUP_SIZE": "3", "SPP_LOOKAHEAD_GROUPS": "1"},
    {"SPP_WHATIF_DUP": "count,pick,tiles,flag,rows"},
    {"SPP_RNG_ARENA_MB": "0", "SPP_GROUP_SIZE": "5", "SPP_XCD_AFFINITY": "0"},
    {"SPP_STREAM_PRIORITY": "low", "SPP_GROUP_SIZE": "16"},
    {"SPP_DEDUP_PREREAD": "1"},
    {"SPP_GROUP_FETCH": "0"},
    {"SPP_GROUP_FETCH": "0", "SPP_GROUP_SIZE": "3"},
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_alternative_chain_paths_are_bit_exact(env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _CHILD], env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
