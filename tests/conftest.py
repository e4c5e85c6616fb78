import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def rel_err(y, ref):
    """Parity metric of SURVEY.md 8d: max|y-ref| / max|ref| per row."""
    y = np.asarray(y)
    ref = np.asarray(ref)
    num = np.abs(y - ref).max(axis=-1)
    den = np.abs(ref).max(axis=-1)
    return num / den
