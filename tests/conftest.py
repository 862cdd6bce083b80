import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle():
    o, _ = graft.load_oracle()
    return o


@pytest.fixture(scope="session")
def coracle():
    _, c = graft.load_oracle()
    return c.COracle()


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def scaled_err(a, b):
    """max |a-b| / max(1, |b|) with NaN sentinels required to coincide."""
    a, b = np.asarray(a), np.asarray(b)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN sentinels differ"
    a, b = np.nan_to_num(a), np.nan_to_num(b)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0
