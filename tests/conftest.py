import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
DATA = os.path.join(ROOT, "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_TAGS = ["rate06_nograin", "rate06_grain", "rate06_default", "rate12_grain"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def racgpu():
    """The product package (directory name has a hyphen, hence importlib)."""
    return importlib.import_module("rac-2d_amd")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement of the reference path -- the checker, never the thing under test."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_ctypes
    oracle_ctypes.lib()
    return oracle_ctypes


def load_golden(tag):
    return np.load(os.path.join(GOLDEN, tag + ".npz"))


def relerr(a, b, floor=0.0):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor))) if a.size else 0.0


def major_relerr(y, yref, thr=1e-6):
    m = yref >= thr
    return float(np.max(np.abs(y[m] - yref[m]) / yref[m]))
