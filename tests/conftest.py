import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def load_golden(name):
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip("golden %s not generated" % name)
    z = np.load(path, allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Compiled on first use with gcc."""
    from oracle import mpi_oracle
    mpi_oracle.lib()
    return mpi_oracle


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        return int((a.view("u%d" % a.dtype.itemsize) != b.view("u%d" % a.dtype.itemsize)).sum())
    return int((a != b).sum())


def max_abs(a, b):
    return float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)))) if np.size(a) else 0.0
