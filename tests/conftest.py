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
    config.addinivalue_line("markers", "offscope: GPU tests of code OUTSIDE the hot path of SURVEY.md section 8 (the frozen backbone epilogue); "
                                       "additive to `gpu`: -m \"gpu and not offscope\" counts the hot-path tests only")


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


def threshold_boundary_dets():
    """dets [N,5] float32 for the NMS threshold tests: nested integer boxes whose IoU is exactly 1/2, 1/3, 1/4, 3/4, 2/3, 9/13,
    7/14; near misses (one side longer by 2^-1 .. 2^-13); zero-area, inverted and duplicated boxes.  Used by the CPU pinning
    of the oracle against the reference's Cython and by the GPU parity test."""
    import numpy as np
    from detectorch_amd import synth
    rs = synth.rng(7, 4321)
    rows = []
    for (w1, h1, w2, h2) in [(9, 9, 9, 19), (9, 9, 9, 29), (9, 9, 19, 19), (9, 29, 9, 39), (9, 19, 9, 29), (11, 11, 11, 17), (6, 6, 13, 6)]:
        for k in range(12):
            ox, oy = float(rs.randint(0, 900)), float(rs.randint(0, 500))
            rows.append([ox, oy, ox + w1, oy + h1])
            rows.append([ox, oy, ox + w2, oy + h2])
    for k in range(60):
        ox, oy = np.float32(rs.uniform(0, 900)), np.float32(rs.uniform(0, 500))
        e = np.float32(2.0 ** -rs.randint(1, 14))
        rows.append([ox, oy, ox + 9, oy + 9])
        rows.append([ox, oy, ox + 9, oy + 19 + e])
    rows += [[50, 50, 49, 80], [50, 50, 49, 80], [70, 70, 60, 60], [70, 70, 60, 60], [10, 10, 10, 10], [10, 10, 10, 10]]
    rows += [[200, 200, 260, 240]] * 5
    b = np.array(rows, np.float32)
    s = synth.dedupe_scores(rs.uniform(0, 1, b.shape[0]).astype(np.float32))
    return np.ascontiguousarray(np.hstack([b, s[:, None]]), np.float32)


BOUNDARY_THRESHOLDS = (0.5, 1.0 / 3.0, 0.25, 0.75, 2.0 / 3.0, 0.7, 1.0, 0.0, -0.5, 1e-30)
