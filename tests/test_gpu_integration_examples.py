"""-m gpu: the two ctypes listings of INTEGRATION.md section 3 are executed VERBATIM (extracted from the
markdown by their `<!-- example: ... -->` markers) and every batch they produce is compared with the
oracle -- a maintainer following the document gets running code, and the document cannot drift from
include/spp.h again (round 1 shipped an 11-argument spp_session_export call against the 12-argument ABI)."""
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES = [15, 10, 5]


def _listing(name):
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- example: %s -->\s*```python\n(.*?)```" % re.escape(name), md, re.S)
    assert m, f"INTEGRATION.md has no listing marked 'example: {name}'"
    return m.group(1)


def test_integration_md_calls_match_the_header():
    """CPU part: every spp_session_export call in the document passes as many arguments as spp.h declares"""
    hdr = open(os.path.join(ROOT, "include", "spp.h")).read()
    decl = re.search(r"spp_status\s+spp_session_export\s*\((.*?)\);", hdr, re.S).group(1)
    n_decl = len([a for a in decl.split(",") if a.strip()])
    for name in ("nondist_epoch", "dist_epoch"):
        src = _listing(name)
        compile(src, f"INTEGRATION.md:{name}", "exec")
        src = re.sub(r"#[^\n]*", "", src)          # comments carry commas of their own
        for m in re.finditer(r"L\.spp_session_export\(", src):
            depth, i, args, cur = 1, m.end(), [], ""
            while depth:
                ch = src[i]
                if ch in "([":
                    depth += 1
                elif ch in ")]":
                    depth -= 1
                if depth == 1 and ch == ",":
                    args.append(cur)
                    cur = ""
                elif depth:
                    cur += ch
                i += 1
            args.append(cur)
            assert len([a for a in args if a.strip()]) == n_decl, (name, args)


@pytest.mark.gpu
def test_integration_md_listings_run_and_match_the_oracle(graph_a):
    from oracle import oracle as orc
    from salient_plusplus_amd import _native as nat
    g = graph_a
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)     # noqa: E731
    F = g["x"].shape[1]
    batch_size = 64
    idx_host = g["idx"][:192].astype(np.int64)
    ns = dict(rowptr=T(g["rowptr"]), col=T(g["col"]), idx=T(idx_host), x_all=T(g["x"]),
              y_all=T(g["y"]).unsqueeze(-1).contiguous(), F=F, batch_size=batch_size)
    exec(compile(_listing("nondist_epoch"), "INTEGRATION.md:nondist_epoch", "exec"), ns)
    ranges = orc.batch_ranges(len(idx_host), batch_size, False, True, len(idx_host) // batch_size)

    def check(batches, with_n_id):
        assert len(batches) == len(ranges)
        for k, b in enumerate(batches):
            x, y, hops, (start, stop) = b[:4]
            assert (start, stop) == (int(ranges[k][0]), int(ranges[k][1]))
            m = orc.sample_batch(g["rowptr"], g["col"], idx_host, start, stop, SIZES)
            np.testing.assert_array_equal(x.cpu().numpy().view(np.uint16), g["x"][m.n_id].view(np.uint16))
            np.testing.assert_array_equal(y.cpu().numpy().reshape(-1), g["y"][m.n_id[:stop - start]])
            for (rp, cl, size), hop in zip(hops, m.hops):
                np.testing.assert_array_equal(rp.cpu().numpy(), hop.rowptr)
                np.testing.assert_array_equal(cl.cpu().numpy(), hop.col)
                assert tuple(size) == tuple(hop.size)
            if with_n_id:
                np.testing.assert_array_equal(b[4].cpu().numpy(), m.n_id)
    check(ns["batches"], False)

    # the distributed listing continues in the same namespace ("rest of cfg as above"); one rank = a
    # world-size-1 RCCL communicator, so every row is local and the cache is never hit
    L = nat.load()
    n = g["rowptr"].shape[0] - 1
    cv = torch.arange(5, 25, dtype=torch.int64, device=dev)
    cache_map = torch.empty(int(cv.max()) + 1, dtype=torch.int32, device=dev)
    import ctypes as C
    nat.check(L.spp_cache_build_map(C.c_void_p(cv.data_ptr()), cv.numel(), C.c_void_p(cache_map.data_ptr()),
                                    cache_map.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    ns.update(rank=0, world=1, device=0, partition_offsets=[0, n], cache_map=cache_map,
              cache_feats=ns["x_all"][cv].contiguous(), x_local=ns["x_all"])
    ns.pop("batches")
    exec(compile(_listing("dist_epoch"), "INTEGRATION.md:dist_epoch", "exec"), ns)
    check(ns["batches"], True)
