"""-m gpu: randomly drawn graphs (hypothesis) for the VIP access probabilities (ddp.py:135-239) against the float64
restatement.  In a module of its own: a box without hypothesis skips THIS, not the deterministic VIP cache tests."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


hyp = pytest.importorskip("hypothesis")
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402


@settings(max_examples=int(os.environ.get("SPP_FUZZ_EXAMPLES", "40")), deadline=None,
          derandomize=os.environ.get("SPP_FUZZ_RANDOM", "0") != "1", suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 2**31 - 1), n=st.integers(5, 4000), mean_deg=st.floats(0.3, 30.0), zero_frac=st.floats(0.0, 0.5),
       n_train=st.integers(1, 1500), fanouts=st.sampled_from([[15, 10, 5], [20, 20, 20], [25, 15], [1], [2, 2, 2, 2], [40, 3]]),
       bs=st.sampled_from([1, 7, 64, 1024]))
def test_vip_frequencies_random_graphs(seed, n, mean_deg, zero_frac, n_train, fanouts, bs):
    """The analytic access probabilities (ddp.py:135-239) on random graphs with isolated vertices, training sets from one
    vertex to all of them (a batch larger than the training set: every training vertex has probability 1) against the
    float64 restatement: only the summation order inside a row differs."""
    from oracle import oracle as orc
    from salient_plusplus_amd.fast_trainer.vip_cache import vip_frequencies
    rng = np.random.default_rng(seed)
    deg = rng.poisson(mean_deg, n).astype(np.int64)
    deg[rng.random(n) < zero_frac] = 0
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(deg)
    col = rng.integers(0, n, rowptr[-1]).astype(np.int64)
    train = rng.choice(n, size=min(n, n_train), replace=False).astype(np.int64)
    T = torch.from_numpy
    want = orc.vip_frequencies(rowptr, col, train, fanouts, bs)
    got = vip_frequencies(T(rowptr), T(col), T(train), fanouts, bs).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-300)
    assert got.min() >= 0.0 and got.max() <= 1.0
