import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
ORACLE_DIR = os.path.join(ROOT, "oracle")
if ORACLE_DIR not in sys.path:
    sys.path.insert(0, ORACLE_DIR)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc  # oracle/oracle.py : TEST INFRASTRUCTURE (the checker)
    orc.lib()
    return orc


def ulp_close(a, b, max_ulp=1, atol=1e-4):
    """box-coordinate tolerance: |a-b| <= 1e-4 OR within max_ulp float32 ulps (coordinates above 1024 have ulp 1.2e-4)."""
    a = np.asarray(a, np.float32)
    b = np.asarray(b, np.float32)
    if a.shape != b.shape:
        return False
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32)).astype(np.float64)
    return bool(np.all((d <= atol) | (d <= max_ulp * ulp)))
