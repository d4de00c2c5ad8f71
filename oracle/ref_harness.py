"""Import the reference's own Python for the hot path, IN PLACE, from /root/reference (build container only).

TEST INFRASTRUCTURE ONLY: used by tests/golden/make_golden.py to generate the committed golden vectors and by the
(non-GPU) pinning tests when /root/reference is mounted.  Nothing in the product imports this file, and nothing on the
GPU box can (the reference tree does not travel).

What has to be shimmed for the 2018-era reference to import under Python 3.10 / numpy 2.2 / torch 2.10
(SURVEY.md section 8c):
  * np.float / np.int aliases            (lib/utils/generate_anchors.py:63-64,72)
  * utils_cython.cython_nms / cython_bbox (lib/utils/boxes.py:53-69 would otherwise try to compile inside the
    read-only tree) -> the copies built by oracle/Makefile into oracle/_ref/utils_cython
  * cv2, pycocotools.mask                 (imported at lib/utils/result_utils.py:21-22, unused by the functions we call)
  * sys.dont_write_bytecode               (no __pycache__ inside /root/reference)
"""
import ctypes
import importlib
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("DETECTORCH_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
REF_BUILD = os.path.join(HERE, "_ref")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "lib", "model"))


def load_ref_cython():
    """The reference's cython_nms / cython_bbox, built by `make -C oracle ref` (works wherever oracle/_ref exists)."""
    path = os.path.join(REF_BUILD, "utils_cython")
    if not os.path.isdir(path):
        raise RuntimeError("oracle/_ref/utils_cython missing: run `make -C oracle ref` in the build container")
    if REF_BUILD not in sys.path:
        sys.path.insert(0, REF_BUILD)
    nms = importlib.import_module("utils_cython.cython_nms")
    bbox = importlib.import_module("utils_cython.cython_bbox")
    return nms, bbox


def load_ref_roialign():
    """ctypes handle on the reference's roi_align_forward_loop (lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp:118)."""
    lib = ctypes.CDLL(os.path.join(REF_BUILD, "libref_roialign.so"))
    f = lib.roi_align_forward_loop
    f.restype = None
    f.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return f


def ref_roi_align(features, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio):
    """numpy wrapper over the reference CPU loop. features [B,C,H,W] f32, rois [R,4|5] f32 -> [R,C,PH,PW] f32."""
    f = load_ref_roialign()
    features = np.ascontiguousarray(features, dtype=np.float32)
    rois = np.ascontiguousarray(rois, dtype=np.float32)
    _, C, H, W = features.shape
    R, cols = rois.shape
    out = np.zeros((R, C, pooled_h, pooled_w), dtype=np.float32)
    f(out.size, features.ctypes.data, rois.ctypes.data, spatial_scale, C, H, W, pooled_h, pooled_w, sampling_ratio,
      cols, out.ctypes.data)
    return out


_loaded = None


def load_reference():
    """Return a namespace with the reference modules (generate_proposals, collect..., boxes, result_utils, ...)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference tree not mounted at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    load_ref_cython()  # registers utils_cython.* in sys.modules before lib/utils/boxes.py asks for it
    for name in ("cv2", "pycocotools", "pycocotools.mask"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["pycocotools"].mask = sys.modules["pycocotools.mask"]
    lib = os.path.join(REF_ROOT, "lib")
    if lib not in sys.path:
        sys.path.insert(0, lib)
    ns = types.SimpleNamespace()
    ns.boxes = importlib.import_module("utils.boxes")
    ns.generate_anchors = importlib.import_module("utils.generate_anchors")
    ns.multilevel_rois = importlib.import_module("utils.multilevel_rois")
    ns.result_utils = importlib.import_module("utils.result_utils")
    ns.generate_proposals = importlib.import_module("model.generate_proposals")
    ns.collect = importlib.import_module("model.collect_and_distribute_fpn_rpn_proposals")
    ns.cython_nms, ns.cython_bbox = load_ref_cython()
    _loaded = ns
    return ns
