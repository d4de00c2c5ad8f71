"""CPU-only checks: the C-ABI library builds/loads and exports every symbol include/detectorch_hip.h declares (no compute
calls without a GPU), plus the host-side logic of the reference-shaped modules."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "detectorch_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(dtc_[a-z0-9_]+|launch_roi_align_forward_hip)\s*\(", src)
    return sorted(set(names))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from detectorch_amd import build
    lib_path = build.build()
    lib = ctypes.CDLL(lib_path)
    syms = declared_symbols()
    assert len(syms) >= 18
    for s in syms:
        assert hasattr(lib, s), "include/detectorch_hip.h declares %s but the library does not export it" % s
    lib.dtc_target_arch.restype = ctypes.c_char_p
    assert lib.dtc_target_arch() == b"gfx950"
    # pure host-side size queries are safe without a GPU
    lib.dtc_nms_workspace_bytes.restype = ctypes.c_size_t
    lib.dtc_nms_workspace_bytes.argtypes = [ctypes.c_int]
    assert lib.dtc_nms_workspace_bytes(6000) > 6000 * 94 * 8


def test_python_binding_covers_the_header():
    from detectorch_amd import hip
    L = hip.lib()
    for s in declared_symbols():
        assert getattr(L, s) is not None


def test_no_cpu_fallback_in_the_product():
    # the product must fail loudly on CPU tensors instead of silently computing somewhere else
    from detectorch_amd.model.roi_align import RoIAlign
    with pytest.raises(RuntimeError):
        RoIAlign(7, 7, 0.25, 2)(torch.zeros(1, 4, 8, 8), torch.zeros(2, 5))
    # ... and nothing under detectorch_amd/ may import the oracle
    for dp, _, files in os.walk(os.path.join(ROOT, "detectorch_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "liboracle" not in txt and "ref_harness" not in txt, f


def test_generate_anchors_known_answer_and_golden():
    from conftest import golden
    from detectorch_amd.utils.generate_anchors import generate_anchors
    table = np.array([[-83, -39, 100, 56], [-175, -87, 192, 104], [-359, -183, 376, 200], [-55, -55, 72, 72],
                      [-119, -119, 136, 136], [-247, -247, 264, 264], [-35, -79, 52, 96], [-79, -167, 96, 184],
                      [-167, -343, 184, 360]], np.float64)          # lib/utils/generate_anchors.py:26-51 (1-based)
    assert np.array_equal(generate_anchors(16, (128, 256, 512), (0.5, 1, 2)), table - 1.0)
    g = golden("anchors")
    k = 0
    while "in%d" % k in g:
        spec = list(g["in%d" % k])
        sep = spec.index(-1.0)
        assert np.array_equal(generate_anchors(spec[0], spec[1:sep], spec[sep + 1:]), g["out%d" % k])
        k += 1


def test_preprocess_rois_and_module_ctor():
    from detectorch_amd.model.generate_proposals import GenerateProposals
    from detectorch_amd.model.roi_align import preprocess_rois
    r = preprocess_rois([torch.ones(2, 4), torch.ones(3, 4) * 2])
    assert tuple(r.shape) == (5, 5) and float(r[:, 0].abs().sum()) == 0
    assert tuple(preprocess_rois(torch.zeros(1, 6, 5)).shape) == (6, 5)
    gp = GenerateProposals()
    assert gp._num_anchors == 15 and gp.rpn_pre_nms_top_n == 6000 and gp.rpn_post_nms_top_n == 1000 and gp.rpn_nms_thresh == 0.7
    assert GenerateProposals(train=True).rpn_pre_nms_top_n == 12000


def test_rle_encode_matches_coco_format():
    from detectorch_amd.utils.result_utils import rle_encode
    m = np.zeros((4, 5), np.uint8)
    m[1:3, 1:4] = 1
    rle = rle_encode(m)
    assert rle["size"] == [4, 5]
    # column-major runs: 5 zeros, (2 ones, 2 zeros) x3 -> counts [5,2,2,2,2,2,5]; COCO string of that sequence
    assert rle["counts"] == "5220003"   # deltas vs counts[i-2] from the 4th run on
    assert rle_encode(np.ones((2, 2), np.uint8))["counts"] == "04"


def test_synthetic_shapes():
    from detectorch_amd import synth
    assert synth.fpn_level_shapes() == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert synth.c4_shape() == (50, 84)
    s = synth.dedupe_scores(np.array([0.5, 0.5, 0.25, 0.5], np.float32))
    assert len(np.unique(s)) == 4


def test_caffe2_blob_names():
    from detectorch_amd.model.detector import _caffe2_name
    assert _caffe2_name("conv1.weight") == "conv1_w"
    assert _caffe2_name("bn1.bias") == "res_conv1_bn_b"
    assert _caffe2_name("layer1.0.conv1.weight") == "res2_0_branch2a_w"
    assert _caffe2_name("layer3.5.bn3.weight") == "res4_5_branch2c_bn_s"
    assert _caffe2_name("layer2.0.downsample.0.weight") == "res3_0_branch1_w"
    assert _caffe2_name("layer2.0.downsample.1.bias") == "res3_0_branch1_bn_b"


def test_argument_validation_returns_error_codes_without_touching_the_gpu():
    """Every entry validates its arguments before the first HIP call: bad calls return DTC_E* (negative), never crash."""
    from detectorch_amd import hip
    L = hip.lib()
    EINVAL, EUNSUP = -1, -4
    lv = (hip.FeatLevel * 1)()
    assert L.dtc_roi_align_forward(None, 1, 256, 0, None, 5, None, 10, 7, 7, 2, None, 0, None) == EINVAL
    assert L.dtc_roi_align_forward(lv, 1, 256, 0, None, 3, None, 10, 7, 7, 2, None, 0, None) == EINVAL      # roi_cols
    assert L.dtc_roi_align_forward(lv, 9, 256, 0, None, 5, None, 0, 7, 7, 2, None, 0, None) == EINVAL       # n_levels
    assert L.launch_roi_align_forward_hip(0, None, None, 1.0, 0, 1, 1, 7, 7, 2, None, None) == 0            # reference: 0 = error
    assert L.dtc_nms(None, -1, 0.5, None, 0, None, None, None) == EINVAL
    assert L.dtc_nms(None, 20000, 0.5, None, 0, None, ctypes.c_void_p(8), None) == EUNSUP                 # n > 16384
    assert L.dtc_soft_nms(None, 10, 0.5, 0.3, 0.001, 7, None, None, ctypes.c_void_p(8), None) == EINVAL     # method
    assert L.dtc_postprocess_detections(None, None, None, None, None, None, 1, 0, 81, 10., 10., 5., 5., .05, .5, 100,
                                        None, 0, None, None, None, None, 128, None) == EINVAL
    assert L.dtc_mask_paste(None, None, 81, 100, None, None, None, 1, 10, 0.5, 1, None, 0, None, None, None, None, None) == EINVAL
    assert L.dtc_fpn_collect_distribute(None, None, None, 1, 9, 10, 10, 2, 5, None, None, None, None, None, None, None,
                                        None, None, 0, None) == EINVAL
    lvl = (hip.RpnLevel * 1)()
    assert L.dtc_rpn_topk_decode_workspace_bytes(lvl, 1, 1, 0) == 0                                        # invalid level -> 0
    assert L.dtc_rpn_topk_decode(None, 1, 1, 800., 1333., 0., None, 0, None, None, None, 0, None) == EINVAL


def test_header_is_plain_c_and_cxx():
    """The boundary is a C ABI: include/detectorch_hip.h must compile as C99 (pedantic) and as C++11, with no torch/HIP types."""
    import shutil
    import subprocess
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "detectorch_hip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", hdr])
    subprocess.check_call(["g++", "-x", "c++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", hdr])
    code = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)          # declarations only (comments cite torch call sites)
    assert "torch" not in code.lower().replace("detectorch", "") and "#include <hip" not in code and "at::" not in code


def test_prep_plan_host_logic_matches_oracle(oracle):
    """dtc_prep_plan is pure host code (scale selection of blob.py:75-82, cv2's dsize rounding, im_list_to_blob's padded
    shape): checked here without a GPU against the oracle restatement, incl. the long-side cap and half-way roundings."""
    from detectorch_amd import hip
    L = hip.lib()
    sizes = [(500, 833), (480, 640), (640, 480), (400, 1000), (1000, 400), (333, 500), (801, 1333), (1, 1), (7, 2000), (1200, 1201)]
    B = len(sizes)
    hs = (ctypes.c_int32 * B)(*[h for h, _ in sizes]); ws = (ctypes.c_int32 * B)(*[w for _, w in sizes])
    for (target, mx, stride) in [(800, 1333, 32), (800, 1333, 1), (600, 1000, 32), (64, 100, 32)]:
        scales = (ctypes.c_double * B)(); out_hw = (ctypes.c_int32 * (2 * B))(); blob_hw = (ctypes.c_int32 * 2)()
        assert L.dtc_prep_plan(hs, ws, B, target, mx, stride, scales, out_hw, blob_hw) == 0
        mh = mw = 0
        for k, (h, w) in enumerate(sizes):
            s = oracle.lib().orc_prep_scale(h, w, target, mx)
            assert scales[k] == s
            oh, ow = max(int(np.round(h * s)), 1), max(int(np.round(w * s)), 1)
            assert (out_hw[2 * k], out_hw[2 * k + 1]) == (oh, ow)
            mh, mw = max(mh, oh), max(mw, ow)
        if stride > 1:
            mh, mw = -(-mh // stride) * stride, -(-mw // stride) * stride
        assert (blob_hw[0], blob_hw[1]) == (mh, mw)
    scales = (ctypes.c_double * 1)(); out_hw = (ctypes.c_int32 * 2)(); blob_hw = (ctypes.c_int32 * 2)()
    assert L.dtc_prep_plan((ctypes.c_int32 * 1)(500), (ctypes.c_int32 * 1)(833), 1, 800, 1333, 32, scales, out_hw, blob_hw) == 0
    assert scales[0] == 1.6 and (out_hw[0], out_hw[1]) == (800, 1333) and (blob_hw[0], blob_hw[1]) == (800, 1344)   # BASELINE input
    assert L.dtc_prep_plan(None, None, 1, 800, 1333, 32, scales, out_hw, blob_hw) == -1
    assert L.dtc_bbox_overlaps(None, -1, 4, None, 1, 4, None, None) == -1 and L.dtc_box_voting(None, 1, None, 9000, 0.5, None, None, None) == -1
    assert L.dtc_mask_rle(None, 0, None, None, None, None, 1, 1, None, 1, None, None, 1, None, None) == -1
