"""A1 parity of the BAND-SWEEP RoIAlign kernel (csrc/roi_align_band.hip, entry dtc_roi_align_forward_banded) against the oracle
(the reference's roi_align_forward_loop restated) and against the packed entry's kernels: bit-exact.  -m gpu."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from detectorch_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hip():
    from detectorch_amd import hip as h
    h.lib()
    return h


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def visiting_order(rois5, lv, band_log2=5):
    """(image, level, band of 2^band_log2 feature rows of the box centre, x centre): dtc_fpn_collect_distribute's order."""
    fs = lv.astype(np.int64) + 2
    yc = np.clip((rois5[:, 2] + rois5[:, 4]) * np.float32(0.5), 0, 65535).astype(np.int64)
    xc = np.clip((rois5[:, 1] + rois5[:, 3]) * np.float32(0.5), 0, 65535).astype(np.int64)
    band = np.minimum((yc >> fs) >> band_log2, 63)
    return np.lexsort((np.arange(len(lv)), xc >> fs, band, lv, rois5[:, 0])).astype(np.int64)


def make_desc(rois5, lv, order, pad_rows=()):
    """packed descriptors in the given order (+ padding rows: level -1, their output rows are zero-filled)"""
    n = len(order)
    desc = np.zeros((n + len(pad_rows), 8), np.float32)
    desc[:n, :5] = rois5[order]
    desc[:n, 5] = lv[order]
    desc[:n, 6] = order
    for i, r in enumerate(pad_rows):
        desc[n + i] = [0, 1, 2, 30, 40, -1, r, 0]
    return desc


def oracle_ref(oracle, feats, rois5, lv, ph):
    ref = np.zeros((rois5.shape[0], feats[0].shape[1], ph, ph), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(feats[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], 2)
    return ref


def fpn_case(oracle, rs, R, C, batch, max_side, ph=7):
    shapes = synth.fpn_level_shapes()[:4]
    feats = [synth.make_features(rs, (batch, C, h, w)) - 0.25 for (h, w) in shapes]      # negative values too
    rois = synth.make_rois(rs, R, max_side=max_side)
    lv = (oracle.map_rois_to_fpn_levels(rois, 2, 5) - 2).astype(np.int32)
    bidx = rs.randint(0, batch, (R, 1)).astype(np.float32)
    rois5 = np.hstack([bidx, rois]).astype(np.float32)
    return feats, rois5, lv


@pytest.mark.parametrize("order_kind", ["visiting", "shuffled", "identity"])
def test_band_sweep_vs_oracle_any_order(hip, oracle, order_kind):
    """Dense small boxes (the box head's distribution: most RoIs on P2, long bands) + large ones on P3-P5, two images, in the
    visiting order the sweep is built for, in a random order (every RoI its own band item) and in the given order: bit-exact
    against the oracle each time, padding rows zero-filled, untouched rows untouched."""
    rs = synth.rng(31, 1)
    f1, r1, l1 = fpn_case(oracle, rs, 700, 16, 2, 64.0)
    _, r2, l2 = fpn_case(oracle, rs, 100, 16, 2, 600.0)
    feats, rois5, lv = f1, np.vstack([r1, r2]), np.concatenate([l1, l2])
    R = rois5.shape[0]
    ref = oracle_ref(oracle, feats, rois5, lv, 7)
    order = {"visiting": visiting_order(rois5, lv), "shuffled": np.random.RandomState(5).permutation(R),
             "identity": np.arange(R)}[order_kind]
    desc = make_desc(rois5, lv, order, pad_rows=(R, R + 1, R + 2))
    out = torch.full((R + 4, 16, 7, 7), 7.0, device="cuda")
    hip.roi_align_forward_banded([cu(f) for f in feats], synth.FPN_ROI_SCALES, cu(desc), 7, 7, 2, out=out)
    res = out.cpu().numpy()
    assert np.array_equal(res[:R], ref)
    assert not res[R:R + 3].any() and (res[R + 3] == 7.0).all()


def test_band_sweep_edge_cases(hip, oracle):
    """Windows the ring cannot hold (wider than 64 columns; a band whose windows cover more rows than the LDS image), RoIs over
    every border, degenerate / identical / far-outside boxes, a level whose width is not a multiple of 4 (P5: 42 columns ->
    scalar staging), all mixed into dense bands: bit-exact against the oracle and against the packed entry's kernels."""
    rs = synth.rng(32, 2)
    feats, rois5, lv = fpn_case(oracle, rs, 300, 8, 2, 48.0)
    extra = np.array([[0, 0, 1343, 799], [0, 100, 1343, 140], [100, 0, 140, 799], [-50, -50, 30, 30], [1300, 760, 1500, 900],
                      [100, 100, 90, 90], [5000, 5000, 6000, 6000], [0, 0, 0, 0], [1343, 799, 1343, 799],
                      [200, 96, 260, 420], [204, 100, 250, 400], [600, 300, 640, 330], [600, 300, 640, 330]], np.float32)
    e5 = np.hstack([(np.arange(len(extra)) % 2)[:, None].astype(np.float32), extra])
    elv = np.array([0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.int32)
    r_up = synth.make_rois(rs, 60, max_side=700.0)
    up5 = np.hstack([rs.randint(0, 2, (60, 1)).astype(np.float32), r_up])
    uplv = (np.arange(60) % 3 + 1).astype(np.int32)            # P3 / P4 / P5 whatever the size: big windows on coarse maps
    rois5 = np.vstack([rois5, e5, up5]).astype(np.float32)
    lv = np.concatenate([lv, elv, uplv])
    R = rois5.shape[0]
    ref = oracle_ref(oracle, feats, rois5, lv, 7)
    tf = [cu(f) for f in feats]
    for order in (visiting_order(rois5, lv), np.random.RandomState(9).permutation(R)):
        desc = cu(make_desc(rois5, lv, order))
        out = hip.roi_align_forward_banded(tf, synth.FPN_ROI_SCALES, desc, 7, 7, 2)
        assert np.array_equal(out.cpu().numpy(), ref)
        lvs, ch, dt = hip.make_levels(tf, synth.FPN_ROI_SCALES)
        outp = torch.empty_like(out)
        assert hip.lib().dtc_roi_align_forward_packed(lvs, 4, ch, 0, desc.data_ptr(), R, 7, 7, 2, outp.data_ptr(), 0,
                                                      hip.stream_ptr()) == 0
        assert torch.equal(out, outp)


def test_band_sweep_bench_shape(hip, oracle):
    """C = 256, two images x 1000 dense RoIs in visiting order (the box-head launch of bench.py per image), fp32 and the
    16-bit output types; the oracle checks 300 of the rows, the cluster-stationary kernel all of them."""
    rs = synth.rng(33, 3)
    feats, rois5, lv = fpn_case(oracle, rs, 2000, 256, 2, 80.0)
    R = rois5.shape[0]
    order = visiting_order(rois5, lv)
    desc = cu(make_desc(rois5, lv, order))
    tf = [cu(f) for f in feats]
    out = hip.roi_align_forward_banded(tf, synth.FPN_ROI_SCALES, desc, 7, 7, 2)
    lvs, ch, dt = hip.make_levels(tf, synth.FPN_ROI_SCALES)
    outp = torch.empty_like(out)
    assert hip.lib().dtc_roi_align_forward_packed(lvs, 4, ch, 0, desc.data_ptr(), R, 7, 7, 2, outp.data_ptr(), 0,
                                                  hip.stream_ptr()) == 0
    assert torch.equal(out, outp)
    pick = np.random.RandomState(1).choice(R, 300, replace=False)
    ref = oracle_ref(oracle, feats, rois5[pick], lv[pick], 7)
    assert np.array_equal(out.cpu().numpy()[pick], ref)
    for odt in (torch.float16, torch.bfloat16):
        o16 = hip.roi_align_forward_banded(tf, synth.FPN_ROI_SCALES, desc, 7, 7, 2, out_dtype=odt)
        assert torch.equal(o16, out.to(odt))                   # same round-to-nearest-even of the same float32 results


def test_band_entry_falls_through_for_other_configurations(hip, oracle):
    """14x14 bins, sampling ratio 0, fp16 maps: the banded entry hands them to the packed entry's kernels (same results)."""
    rs = synth.rng(34, 4)
    feats, rois5, lv = fpn_case(oracle, rs, 120, 8, 1, 300.0)
    order = visiting_order(rois5, lv)
    desc = cu(make_desc(rois5, lv, order))
    tf = [cu(f) for f in feats]
    ref14 = oracle_ref(oracle, feats, rois5, lv, 14)
    assert np.array_equal(hip.roi_align_forward_banded(tf, synth.FPN_ROI_SCALES, desc, 14, 14, 2).cpu().numpy(), ref14)
    h16 = [t.half() for t in tf]
    up = [t.float().cpu().numpy() for t in h16]
    ref16 = oracle_ref(oracle, up, rois5, lv, 7)
    assert np.array_equal(hip.roi_align_forward_banded(h16, synth.FPN_ROI_SCALES, desc, 7, 7, 2).cpu().numpy(), ref16)


_CHILD = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
import test_hip_roi_align_band as T
import oracle as orc
from detectorch_amd import hip, synth
rs = synth.rng(35, 5)
feats, rois5, lv = T.fpn_case(orc, rs, 900, 16, 2, 100.0)
R = rois5.shape[0]
ref = T.oracle_ref(orc, feats, rois5, lv, 7)
tf = [T.cu(f) for f in feats]
for order in (T.visiting_order(rois5, lv), np.random.RandomState(2).permutation(R)):
    out = hip.roi_align_forward_banded(tf, synth.FPN_ROI_SCALES, T.cu(T.make_desc(rois5, lv, order)), 7, 7, 2)
    assert np.array_equal(out.cpu().numpy(), ref)
print("ok")
"""


@pytest.mark.parametrize("env", ["DTC_RA_BAND_ROWS=16", "DTC_RA_BAND_ROWS=9 DTC_RA_BAND_K=3", "DTC_RA_BAND_K=1",
                                 "DTC_RA_BAND_GRID=3", "DTC_RA_BAND_GRID=700 DTC_RA_BAND_MAXUNITS=1", "DTC_RA_BAND=0"])
def test_band_knobs_bit_exact_in_child_process(hip, oracle, env):
    """The sweep's shape knobs (rows the LDS image holds -> how many RoIs take the gather path; RoIs per batch; persistent
    workgroups vs one unit per workgroup, i.e. the stealing / mop-up logic of the work queues) change nothing in the result.
    The knobs are resolved once per process, so each setting runs in a child process."""
    e = dict(os.environ)
    for kv in env.split():
        k, v = kv.split("=")
        e[k] = v
    code = _CHILD % (ROOT, os.path.join(ROOT, "tests"))
    code = code.replace("import oracle as orc", "sys.path.insert(0, %r); import oracle as orc" % os.path.join(ROOT, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, env + "\n" + r.stdout[-1500:] + r.stderr[-3000:]
