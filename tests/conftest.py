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
def gS1():
    return np.load(os.path.join(GOLDEN, "golden_S1.npz"))


@pytest.fixture(scope="session")
def gS2():
    return np.load(os.path.join(GOLDEN, "golden_S2.npz"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    return o


def relmax(a, b):
    """max |a-b| / max |b| (per array): the 'relative' of north_star's 1e-5 tolerance."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def relmax_rows(a, b):
    """per-row relmax for [nvox, n] arrays."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.max(np.abs(b), axis=1)
    den = np.where(den > 0, den, 1.0)
    return np.max(np.abs(a - b), axis=1) / den
