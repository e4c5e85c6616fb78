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


@pytest.fixture
def option():
    """Sets a named switch of the library (ghostcwt_debug.h: gcwt_debug_set_option) for the test and
    restores the default afterwards: ``option("synth16", 1)``; ``option("synth16", None)`` clears."""
    from ghost_amd.engine import set_option
    touched = set()

    def _set(name, value):
        set_option(name, value)
        touched.add(name)
    yield _set
    for name in touched:
        set_option(name, None)
