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
def op_gold():
    return np.load(os.path.join(GOLDEN, "op_goldens.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def net_gold():
    return np.load(os.path.join(GOLDEN, "net_goldens.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def wsm_gold():
    return np.load(os.path.join(GOLDEN, "wsm_goldens.npz"), allow_pickle=False)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))
