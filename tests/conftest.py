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


# kernel-variant census of the operator-level parity modules (rdm_census_*): each module resets the census when it starts and leaves its
# snapshot here when it ends; tests/test_gpu_zz_coverage.py (runs last) holds the headline step's variants against the union
CENSUS = {}
RAN = {}


@pytest.fixture(scope="module")
def op_census(request):
    from md_rdm_amd import _lib
    L = _lib.lib()
    L.rdm_census_reset()
    L.rdm_census_enable(1)
    name = request.module.__name__
    RAN.setdefault(name, set())
    yield RAN[name]
    CENSUS[name] = _lib.census()
    L.rdm_census_enable(0)
