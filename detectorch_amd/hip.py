"""ctypes binding of libdetectorch_hip.so (C ABI: include/detectorch_hip.h) + the torch plumbing around it.

PyTorch is used for device memory and the current HIP stream only; every computation below happens in the hand-written
HIP kernels of detectorch_amd/csrc.  There is NO fallback: if the library is missing this module raises on first use.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdetectorch_hip.so")

DTC_OK = 0
DTC_F32, DTC_F16 = 0, 1
DTC_MAX_LEVELS = 8
_ERR = {-1: "DTC_EINVAL", -2: "DTC_ELAUNCH", -3: "DTC_EWORKSPACE", -4: "DTC_EUNSUPPORTED"}


class FeatLevel(C.Structure):
    """struct dtc_feat_level (include/detectorch_hip.h)"""
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("spatial_scale", C.c_float),
                ("_pad", C.c_int32), ("stride_n", C.c_int64), ("stride_c", C.c_int64), ("stride_h", C.c_int64),
                ("stride_w", C.c_int64)]


_lib = None


def lib():
    """Load the native library or fail loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "detectorch_amd: %s is missing. Build it with `python -m detectorch_amd.build` (hipcc, gfx950). "
            "There is no CPU/PyTorch fallback for the region-proposal hot path." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    p, i, f = C.c_void_p, C.c_int, C.c_float
    L.dtc_version.restype = C.c_char_p
    L.dtc_target_arch.restype = C.c_char_p
    L.launch_roi_align_forward_hip.argtypes = [i, p, p, f, i, i, i, i, i, i, p, p]
    L.launch_roi_align_forward_hip.restype = i
    L.dtc_roi_align_forward.argtypes = [C.POINTER(FeatLevel), i, i, i, p, i, p, i, i, i, i, p, i, p]
    L.dtc_roi_align_forward.restype = i
    sz = C.c_size_t
    L.dtc_nms_workspace_bytes.argtypes = [i]
    L.dtc_nms_workspace_bytes.restype = sz
    L.dtc_nms.argtypes = [p, i, f, p, sz, p, p, p]
    L.dtc_nms.restype = i
    L.dtc_nms_sorted_workspace_bytes.argtypes = [i, i]
    L.dtc_nms_sorted_workspace_bytes.restype = sz
    L.dtc_nms_sorted.argtypes = [p, p, i, i, f, i, p, sz, p, i, p, p]
    L.dtc_nms_sorted.restype = i
    L.dtc_segment_sort_desc.argtypes = [p, i, p, i, p, i, i, p, p, p, p]
    L.dtc_segment_sort_desc.restype = i
    _lib = L
    return L


def check(rc, what):
    if rc != DTC_OK:
        raise RuntimeError("detectorch_hip: %s failed with %s" % (what, _ERR.get(rc, rc)))


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _dtype_code(t):
    if t == torch.float32:
        return DTC_F32
    if t == torch.float16:
        return DTC_F16
    raise TypeError("detectorch_hip supports float32 / float16 features, got %s" % t)


def _require_cuda(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("detectorch_amd runs the region-proposal hot path on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise TypeError("all tensors must be on the same device")
    return dev


def make_levels(features, spatial_scales):
    """list of [B,C,H,W] tensors (any strides: NCHW-contiguous or channels_last) -> (FeatLevel array, C, dtype)."""
    if len(features) > DTC_MAX_LEVELS:
        raise ValueError("at most %d levels" % DTC_MAX_LEVELS)
    arr = (FeatLevel * len(features))()
    ch, dt = features[0].shape[1], features[0].dtype
    for k, (t, s) in enumerate(zip(features, spatial_scales)):
        if t.dim() != 4 or t.shape[1] != ch or t.dtype != dt:
            raise ValueError("feature levels must be [B,C,H,W] with equal C and dtype")
        sn, sc, sh, sw = t.stride()
        arr[k] = FeatLevel(t.data_ptr(), t.shape[2], t.shape[3], float(s), 0, sn, sc, sh, sw)
    return arr, ch, dt


def roi_align_forward(features, spatial_scales, rois, pooled_h, pooled_w, sampling_ratio, roi_levels=None,
                      out_dtype=None, out=None):
    """Multi-level RoIAlign forward (dtc_roi_align_forward).

    features: tensor or list of tensors [B,C,H_l,W_l]; rois [R,4|5] float32; roi_levels int32 [R] or None.
    Returns [R,C,PH,PW] in roi order.
    """
    if torch.is_tensor(features):
        features, spatial_scales = [features], [spatial_scales]
    dev = _require_cuda(rois, roi_levels, *features)
    if rois.dtype != torch.float32:
        raise TypeError("rois must be float32")
    rois = rois.contiguous()
    R, cols = (rois.shape[0], rois.shape[1]) if rois.dim() == 2 else (0, 5)
    lv, ch, dt = make_levels(features, spatial_scales)
    odt = out_dtype or torch.float32
    if out is None:
        out = torch.empty((R, ch, pooled_h, pooled_w), dtype=odt, device=dev)
    if roi_levels is not None:
        roi_levels = roi_levels.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc = lib().dtc_roi_align_forward(lv, len(features), ch, _dtype_code(dt), rois.data_ptr(), cols if R else 5,
                                         roi_levels.data_ptr() if roi_levels is not None else None, R, int(pooled_h),
                                         int(pooled_w), int(sampling_ratio), out.data_ptr(), _dtype_code(odt),
                                         stream_ptr(dev))
    check(rc, "dtc_roi_align_forward")
    return out


def _ptr(t):
    return None if t is None else t.data_ptr()


def workspace(nbytes, device):
    """uint8 scratch tensor (torch owns the memory; the kernels only see the pointer)."""
    return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)


def nms(dets, thresh):
    """dtc_nms: dets [N,5] float32 CUDA tensor -> int64 CUDA tensor of ascending original indices (one D2H for the count)."""
    dev = _require_cuda(dets)
    n = dets.shape[0]
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    dets = dets.contiguous()
    if dets.dtype != torch.float32 or dets.dim() != 2 or dets.shape[1] != 5:
        raise TypeError("dets must be float32 [N,5]")
    L = lib()
    ws = workspace(L.dtc_nms_workspace_bytes(n), dev)
    keep = torch.empty((n,), dtype=torch.int64, device=dev)
    cnt = torch.empty((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.dtc_nms(dets.data_ptr(), n, float(thresh), ws.data_ptr(), ws.numel(), keep.data_ptr(), cnt.data_ptr(),
                       stream_ptr(dev))
    check(rc, "dtc_nms")
    return keep[:int(cnt.item())]


def nms_sorted(boxes, counts, thresh, max_keep=0, keep_stride=None):
    """dtc_nms_sorted: boxes [S,N,4] score-sorted, counts int32 [S] or None -> (keep int32 [S,keep_stride], keep_count [S])."""
    dev = _require_cuda(boxes, counts)
    boxes = boxes.contiguous()
    S, N = boxes.shape[0], boxes.shape[1]
    ks = int(keep_stride or (max_keep if max_keep > 0 else N))
    L = lib()
    ws = workspace(L.dtc_nms_sorted_workspace_bytes(S, N), dev)
    keep = torch.empty((S, max(ks, 1)), dtype=torch.int32, device=dev)
    cnt = torch.empty((max(S, 1),), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.dtc_nms_sorted(boxes.data_ptr(), _ptr(counts), S, N, float(thresh), int(max_keep), ws.data_ptr(),
                              ws.numel(), keep.data_ptr(), ks, cnt.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_nms_sorted")
    return keep, cnt[:S]
