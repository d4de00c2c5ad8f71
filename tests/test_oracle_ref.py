"""Pin oracle/oracle.c against the reference's OWN compiled code in oracle/_ref (built by `make -C oracle ref` from
/root/reference; the built .so files travel to the GPU box, the sources do not).  CPU only, bit-exact.

  A1: lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp (unmodified)   A5/A6: lib/utils_cython/cython_nms.pyx
"""
import os

import numpy as np
import pytest

from detectorch_amd import synth

REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "libref_roialign.so")),
                                reason="oracle/_ref not built (needs /root/reference)")


@pytest.fixture(scope="module")
def ref():
    import ref_harness as rh
    return rh


@pytest.mark.parametrize("ph,pw,sr,scale,C,H,W", [(7, 7, 2, 0.25, 16, 50, 84), (14, 14, 0, 1 / 16., 8, 50, 84),
                                                  (7, 7, 0, 1 / 16., 8, 50, 84), (14, 14, 2, 1 / 32., 16, 25, 42)])
def test_roi_align_vs_reference_cpu_loop(oracle, ref, ph, pw, sr, scale, C, H, W):
    rs = synth.rng(11, ph + sr)
    feat = rs.standard_normal((2, C, H, W)).astype(np.float32)
    rois = synth.make_rois(rs, 64)
    rois5 = np.hstack([rs.randint(0, 2, (64, 1)).astype(np.float32), rois])
    assert np.array_equal(oracle.roi_align_forward(feat, rois5, ph, pw, scale, sr),
                          ref.ref_roi_align(feat, rois5, ph, pw, scale, sr))


@pytest.mark.parametrize("n,thr", [(1, 0.5), (2, 0.5), (1000, 0.7), (3000, 0.7), (1000, 0.3)])
def test_nms_vs_reference_cython(oracle, ref, n, thr):
    cn, _ = ref.load_ref_cython()
    rs = synth.rng(12, n)
    dets = np.hstack([synth.make_rois(rs, n), synth.dedupe_scores(rs.uniform(0, 1, n).astype(np.float32))[:, None]])
    dets = np.ascontiguousarray(dets, np.float32)
    assert np.array_equal(oracle.nms(dets, thr), cn.nms(dets, np.float32(thr)))


def test_nms_threshold_boundary_set_vs_reference_cython(oracle, ref):
    """IoU exactly on / a hair off the threshold, non-positive thresholds, degenerate boxes: oracle == the reference's Cython."""
    from conftest import BOUNDARY_THRESHOLDS, threshold_boundary_dets
    cn, _ = ref.load_ref_cython()
    d = threshold_boundary_dets()
    for thr in BOUNDARY_THRESHOLDS:
        assert np.array_equal(oracle.nms(d, thr), cn.nms(d, np.float32(thr))), thr


@pytest.mark.parametrize("method", ["hard", "linear", "gaussian"])
def test_soft_nms_vs_reference_cython(oracle, ref, method):
    cn, _ = ref.load_ref_cython()
    rs = synth.rng(13, 0)
    n = 700
    dets = np.hstack([synth.make_rois(rs, n, min_side=30, max_side=400),
                      synth.dedupe_scores(rs.uniform(0, 1, n).astype(np.float32))[:, None]])
    dets = np.ascontiguousarray(dets, np.float32)
    m = {"hard": 0, "linear": 1, "gaussian": 2}[method]
    rd, rk = cn.soft_nms(dets, np.float32(0.5), np.float32(0.3), np.float32(0.001), np.uint8(m))
    d, k = oracle.soft_nms(dets, 0.5, 0.3, 0.001, method)
    assert np.array_equal(k, np.asarray(rk, np.int64))
    assert np.array_equal(d, rd)


def test_bbox_overlaps_and_box_voting_vs_reference(oracle, ref):
    """orc_bbox_overlaps vs the reference's own Cython build; orc_box_voting vs lib/utils/boxes.py:280 imported in place
    (hundreds of voters per top det: numpy's pairwise float32 sum beyond 128 elements is on the path)."""
    _, cb = ref.load_ref_cython()
    ns = ref.load_reference()
    rs = synth.rng(13, 0)
    for t in range(6):
        b = synth.make_rois(rs, 150 + 37 * t); q = synth.make_rois(rs, 90 + 11 * t)
        assert np.array_equal(oracle.bbox_overlaps(b, q), cb.bbox_overlaps(np.ascontiguousarray(b), np.ascontiguousarray(q)))
    base = np.array([[50, 60, 200, 220], [300, 100, 420, 300], [10, 10, 600, 400]], np.float32)
    for t in range(4):
        n = 400 + 300 * t
        a = base[rs.randint(0, 3, n)] + rs.standard_normal((n, 4)).astype(np.float32) * 3
        all_d = np.ascontiguousarray(np.hstack([a, rs.uniform(0, 1, (n, 1))]), np.float32)
        top = np.ascontiguousarray(all_d[rs.choice(n, 10, replace=False)])
        assert np.array_equal(oracle.box_voting(top, all_d, 0.5), ns.boxes.box_voting(top, all_d, 0.5))
