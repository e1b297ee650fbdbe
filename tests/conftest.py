import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the in-process transport (spp_comm_create_local: ranks as threads on one GPU) is a test aid the library refuses by default
os.environ.setdefault("SPP_ALLOW_LOCAL_COMM", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def graph_a():
    import numpy as np
    g = np.load(os.path.join(GOLDEN, "graph_a.npz"))
    return {k: g[k] for k in g.files}
