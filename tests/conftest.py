import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
PKG = os.path.join(REPO, "dex-nerf_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def rel_err(a, b):
    """max|a-b| / max|b| : the per-tensor tolerance definition of SURVEY.md section 8c."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), m), "finite/NaN pattern differs"
    if not m.any():
        return 0.0
    return float(np.abs(a[m] - b[m]).max() / max(np.abs(b[m]).max(), 1e-30))
