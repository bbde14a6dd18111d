import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))

    return load


def rel_err(a, b):
    """Element-wise relative error with NaN == NaN and exact matches counted as 0; a NaN on one side
    only is +inf."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    with np.errstate(all="ignore"):
        d = np.abs(a - b) / np.maximum(np.abs(b), 1e-300)
    d[(a == b) | (np.isnan(a) & np.isnan(b))] = 0
    d[np.isnan(a) ^ np.isnan(b)] = np.inf
    return d


@pytest.fixture(scope="session")
def relerr():
    return rel_err
