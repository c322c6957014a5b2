import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
EPISODES = ["const_2_5", "random_a", "random_b", "zeros", "max", "det_influent"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def tables():
    t = golden("influent_tables")
    return np.ascontiguousarray(t["means"]), np.ascontiguousarray(t["stds"])


def gate(x, ref):
    """The parity gate of BASELINE.md section 3: |x - ref| / (1e-5*|ref| + 1e-5*scale_i); pass <= 1."""
    from oracle import sbr_params as P
    x, ref = np.asarray(x), np.asarray(ref)
    return np.abs(x - ref) / (P.RTOL_GATE * np.abs(ref) + P.RTOL_GATE * P.STATE_SCALE)
