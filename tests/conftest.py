import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ADV_CS = (2, 4, 8, 16, 32, 64, 128)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes front-end of libuspmv.so); builds the library if needed."""
    import __graft_entry__ as ge
    p = ge.load_package()
    if not os.path.exists(p.library_path()):
        p.build_library()
    p.lib()
    return p


@pytest.fixture(scope="session")
def orc():
    """The parity oracle (CPU restatement of the reference) -- checker only."""
    from oracle import oracle
    oracle.lib()
    return oracle


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def mtx_path(name):
    return os.path.join(GOLDEN, "matrices", name + ".mtx")


def make_x(n, dtype=np.float64):
    return (1.0 + 1e-3 * (np.arange(n) % 1000).astype(np.float64)).astype(dtype)


def block_x(xp, n_used, b, ld, rowwise):
    """Block vector used by the golden generator: column v = xp * (1 + v/8)."""
    X = np.zeros(b * ld, xp.dtype)
    for v in range(b):
        col = (xp[:n_used] * xp.dtype.type(1.0 + v / 8.0)).astype(xp.dtype)
        if rowwise:
            X[np.arange(n_used) * b + v] = col
        else:
            X[v * ld: v * ld + n_used] = col
    return X
