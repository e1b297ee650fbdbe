"""-m gpu: randomly drawn cases (hypothesis) for the raw C-ABI row gather -- random widths, strides and misaligned
addresses -- against the oracle.  In a module of its own: a box without hypothesis skips THIS, not the deterministic
kernel tests of test_gpu_kernels.py."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def lib():
    from salient_plusplus_amd import _native as nat
    L = nat.load()
    nat.require_device()
    return L


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def check(L, rc):
    assert rc >= 0, L.spp_last_error().decode()


hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402


@settings(max_examples=int(os.environ.get("SPP_FUZZ_EXAMPLES", "120")), deadline=None,
          derandomize=os.environ.get("SPP_FUZZ_RANDOM", "0") != "1", suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), row_bytes=st.one_of(st.integers(1, 64), st.integers(1, 2100), st.sampled_from([128, 200, 256, 512, 1536])),
       pad=st.sampled_from([0, 0, 1, 2, 6, 8, 56, 128]), n_src=st.integers(1, 5000), n_idx=st.integers(0, 9000), idx_bytes=st.sampled_from([4, 8]),
       src_off=st.sampled_from([0, 0, 1, 2, 4, 8, 16]), dst_off=st.sampled_from([0, 0, 1, 2, 4, 8, 16]), limit=st.floats(0.0, 1.0))
def test_gather_rows_random_shapes(lib, seed, row_bytes, pad, n_src, n_idx, idx_bytes, src_off, dst_off, limit):
    """serial_index (fast_sampler.cpp:238-279) over randomly drawn row widths, paddings, table and index sizes, index widths
    and source / destination addresses that are not multiples of the row movers' vector widths: the bytes of the chosen rows,
    nothing before, between or after them."""
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    stride = row_bytes + pad
    table = np.full((n_src, stride), 0xEE, dtype=np.uint8)
    table[:, :row_bytes] = rng.integers(0, 256, size=(n_src, row_bytes), dtype=np.uint8)
    idx = rng.integers(0, n_src, size=n_idx).astype(np.int64)
    n = int(round(limit * n_idx))
    want = orc.serial_index(np.ascontiguousarray(table[:, :row_bytes]), idx[:n]) if n else np.zeros((0, row_bytes), np.uint8)
    raw_src = torch.full((n_src * stride + 64,), 0x77, dtype=torch.uint8, device="cuda")
    raw_src[src_off:src_off + n_src * stride] = torch.from_numpy(table.reshape(-1)).cuda()
    d_idx = dev(idx.astype(np.int64 if idx_bytes == 8 else np.int32)) if n_idx else torch.zeros(1, dtype=torch.int64, device="cuda")
    raw_dst = torch.full((n_idx * row_bytes + 96,), 0xAB, dtype=torch.uint8, device="cuda")
    check(lib, lib.spp_gather_rows_strided(C.c_void_p(raw_src.data_ptr() + src_off), n_src, row_bytes, stride, P(d_idx), idx_bytes, n_idx, n,
                                           C.c_void_p(raw_dst.data_ptr() + dst_off), None))
    torch.cuda.synchronize()
    got = raw_dst.cpu().numpy()
    np.testing.assert_array_equal(got[dst_off:dst_off + n * row_bytes].reshape(n, row_bytes), want)
    assert (got[:dst_off] == 0xAB).all() and (got[dst_off + n * row_bytes:] == 0xAB).all()
