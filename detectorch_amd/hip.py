"""ctypes binding of libdetectorch_hip.so (C ABI: include/detectorch_hip.h) + the torch plumbing around it.

PyTorch is used for device memory and the current HIP stream only; every computation below happens in the hand-written
HIP kernels of detectorch_amd/csrc.  There is NO fallback: if the library is missing this module raises on first use.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DETECTORCH_HIP_LIB") or os.path.join(_HERE, "lib", "libdetectorch_hip.so")

DTC_OK = 0
DTC_F32, DTC_F16, DTC_U8, DTC_BF16 = 0, 1, 2, 3
DTC_MAX_LEVELS = 8
_ERR = {-1: "DTC_EINVAL", -2: "DTC_ELAUNCH", -3: "DTC_EWORKSPACE", -4: "DTC_EUNSUPPORTED"}


class Image(C.Structure):
    """struct dtc_image (include/detectorch_hip.h)"""
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("dtype", C.c_int32),
                ("row_stride", C.c_int32)]


class RpnLevel(C.Structure):
    """struct dtc_rpn_level (include/detectorch_hip.h)"""
    _fields_ = [("cls_prob", C.c_void_p), ("bbox_pred", C.c_void_p), ("num_anchors", C.c_int32), ("height", C.c_int32),
                ("width", C.c_int32), ("pre_nms_top_n", C.c_int32), ("feat_stride", C.c_float), ("score_is_logit", C.c_int32),
                ("anchors", C.c_float * 64)]


class FeatLevel(C.Structure):
    """struct dtc_feat_level (include/detectorch_hip.h)"""
    _fields_ = [("data", C.c_void_p), ("height", C.c_int32), ("width", C.c_int32), ("spatial_scale", C.c_float),
                ("_pad", C.c_int32), ("stride_n", C.c_int64), ("stride_c", C.c_int64), ("stride_h", C.c_int64),
                ("stride_w", C.c_int64)]


class FpnMapOut(C.Structure):
    """struct dtc_fpn_map_out (include/detectorch_hip.h)"""
    _fields_ = [("rois5", C.c_void_p), ("roi_levels", C.c_void_p), ("n_out", C.c_void_p), ("rois_by_level", C.c_void_p),
                ("level_counts", C.c_void_p), ("idx_restore", C.c_void_p), ("roi_order", C.c_void_p), ("roi_desc", C.c_void_p),
                ("k_min", C.c_int32), ("k_max", C.c_int32)]


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
    L.dtc_roi_align_set_exact.argtypes = [i]
    L.dtc_roi_align_set_exact.restype = None
    L.dtc_roi_align_get_exact.argtypes = []
    L.dtc_roi_align_get_exact.restype = i
    L.dtc_roi_align_forward_ordered.argtypes = [C.POINTER(FeatLevel), i, i, i, p, i, p, p, i, i, i, i, p, i, p]
    L.dtc_roi_align_forward_ordered.restype = i
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
    L.dtc_rpn_topk_decode_workspace_bytes.argtypes = [C.POINTER(RpnLevel), i, i, i]
    L.dtc_rpn_topk_decode_workspace_bytes.restype = sz
    L.dtc_rpn_topk_decode.argtypes = [C.POINTER(RpnLevel), i, i, f, f, f, p, sz, p, p, p, i, p]
    L.dtc_rpn_topk_decode.restype = i
    L.dtc_gather_kept.argtypes = [p, p, i, i, p, p, i, p, p, p]
    L.dtc_gather_kept.restype = i
    ll = C.c_longlong
    L.dtc_fpn_collect_distribute.argtypes = [p, p, p, i, i, i, i, i, i, p, p, p, p, p, p, p, p, p, i, p]
    L.dtc_fpn_collect_distribute_kept.argtypes = [p, p, i, p, p, i, i, i, i, i, i, p, p, p, p, p, p, p, p, p, p]
    L.dtc_fpn_collect_distribute_kept.restype = i
    L.dtc_roi_align_forward_packed.argtypes = [C.POINTER(FeatLevel), i, i, i, p, i, i, i, i, p, i, p]
    L.dtc_roi_align_forward_packed.restype = i
    L.dtc_roi_align_workspace_bytes.argtypes = [i]
    L.dtc_roi_align_workspace_bytes.restype = C.c_size_t
    L.dtc_roi_align_forward_packed_ws.argtypes = [C.POINTER(FeatLevel), i, i, i, p, i, i, i, i, p, i, p, C.c_size_t, p]
    L.dtc_roi_align_forward_packed_ws.restype = i
    L.dtc_fpn_collect_distribute.restype = i
    L.dtc_postprocess_detections_workspace_bytes.argtypes = [i, i, i]
    L.dtc_postprocess_detections_workspace_bytes.restype = sz
    L.dtc_postprocess_detections.argtypes = [p, p, p, p, p, p, i, i, i, f, f, f, f, f, f, i, p, sz, p, p, p, p, i, p]
    L.dtc_postprocess_detections.restype = i
    L.dtc_postprocess_detections_logits.argtypes = L.dtc_postprocess_detections.argtypes
    L.dtc_postprocess_detections_logits.restype = i
    L.dtc_postprocess_detections_fpn.argtypes = [p, p, p, i, p, p, p, i, i, i, f, f, f, f, f, f, i, p, sz, p, p, p, p, i,
                                                 C.POINTER(FpnMapOut), p]
    L.dtc_postprocess_detections_fpn.restype = i
    L.dtc_box_results_nms_limit.argtypes = [p, p, p, i, i, i, f, f, i, p, sz, p, p, p, i, p]
    L.dtc_box_results_nms_limit.restype = i
    L.dtc_bias_act.argtypes = [p, p, p, i, i, i, i, i, i, i, i, p]
    L.dtc_bias_act.restype = i
    L.dtc_mask_paste.argtypes = [p, p, i, i, p, p, p, i, i, f, i, p, ll, p, p, p, p, p]
    L.dtc_mask_paste.restype = i
    L.dtc_mask_rle.argtypes = [p, ll, p, p, p, p, i, i, p, i, p, p, i, p, p]
    L.dtc_mask_rle.restype = i
    L.dtc_prep_plan.argtypes = [p, p, i, i, i, i, p, p, p]
    L.dtc_prep_plan.restype = i
    L.dtc_prep_images.argtypes = [C.POINTER(Image), i, p, p, p, p, i, i, p]
    L.dtc_prep_images.restype = i
    L.dtc_bbox_overlaps.argtypes = [p, i, i, p, i, i, p, p]
    L.dtc_bbox_overlaps.restype = i
    L.dtc_box_voting.argtypes = [p, i, p, i, f, p, p, p]
    L.dtc_box_voting.restype = i
    L.dtc_soft_nms.argtypes = [p, i, f, f, f, i, p, p, p, p]
    L.dtc_soft_nms.restype = i
    L.dtc_bbox_transform.argtypes = [p, p, i, i, f, f, f, f, i, f, f, p, p]
    L.dtc_bbox_transform.restype = i
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
    if t == torch.bfloat16:
        return DTC_BF16
    raise TypeError("detectorch_hip supports float32 / float16 / bfloat16 features, got %s" % t)


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


def bias_act_(x, bias=None, residual=None, relu=True, residual_up2=False):
    """dtc_bias_act: in place  x = act(x + bias[c] + residual)  on a dense [N,C,H,W] tensor (NCHW-contiguous or channels_last;
    float32 / float16 / bfloat16).  bias float32 [C]; residual of x's dtype and layout, [N,C,H,W] or -- residual_up2 --
    [N,C,H/2,W/2] read with nearest x2 upsampling (the FPN top-down sum).  Returns x."""
    dev = _require_cuda(x, bias, residual)
    if x.dim() != 4:
        raise ValueError("x must be [N,C,H,W]")
    n, c, h, w = x.shape
    if x.is_contiguous():
        cl = 0
    elif x.is_contiguous(memory_format=torch.channels_last):
        cl = 1
    else:
        raise ValueError("x must be dense (NCHW-contiguous or channels_last)")
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != c or not bias.is_contiguous():
            raise ValueError("bias must be a contiguous float32 [C] tensor")
    if residual is not None:
        want = (n, c, h // 2, w // 2) if residual_up2 else (n, c, h, w)
        if tuple(residual.shape) != want or residual.dtype != x.dtype:
            raise ValueError("residual must be %s of dtype %s" % (want, x.dtype))
        ok = residual.is_contiguous(memory_format=torch.channels_last) if cl else residual.is_contiguous()
        if not ok:
            raise ValueError("residual must have x's memory layout")
    elif residual_up2:
        raise ValueError("residual_up2 needs a residual")
    with torch.cuda.device(dev):      # like every other launcher: the kernel goes to x's device whatever the current one is
        check(lib().dtc_bias_act(x.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 residual.data_ptr() if residual is not None else None, n, c, h, w, _dtype_code(x.dtype), cl,
                                 1 if relu else 0, 1 if residual_up2 else 0, stream_ptr(dev)), "dtc_bias_act")
    return x


def roi_align_set_exact(exact=True):
    """Process-wide switch (dtc_roi_align_set_exact): True (default) = bit-identical to the reference; False = the C4 (adaptive
    sampling, single level) kernel may merge taps -- the same sums in exact arithmetic, <= 1e-5 from the reference in float32.
    Read by the host at LAUNCH time: a hipGraph keeps the mode it was captured with; shared by all threads / streams of the process
    (include/detectorch_hip.h)."""
    lib().dtc_roi_align_set_exact(1 if exact else 0)


def roi_align_forward(features, spatial_scales, rois, pooled_h, pooled_w, sampling_ratio, roi_levels=None,
                      out_dtype=None, out=None, roi_order=None):
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
    if (len(features) == 1 and features[0].shape[0] == 1 and rois.dim() == 2 and rois.shape[1] == 5 and rois.shape[0] > 0
            and roi_levels is None):
        # one image: the batch column can only hold 0 (lib/cppcuda/roi_align_cpu.cpp:143-147 indexes the batch with it), so the
        # RoIs go down as 4 columns -- which tells the library that they all belong to one map (map-stationary kernel for C4)
        rois = rois[:, 1:].contiguous()
    R, cols = (rois.shape[0], rois.shape[1]) if rois.dim() == 2 else (0, 5)
    lv, ch, dt = make_levels(features, spatial_scales)
    odt = out_dtype or (out.dtype if out is not None else torch.float32)
    if out is not None and (out.dtype != odt or not out.is_contiguous()):
        raise TypeError("out must be contiguous and of dtype out_dtype")
    if out is None:
        out = torch.empty((R, ch, pooled_h, pooled_w), dtype=odt, device=dev)
    if roi_levels is not None:
        roi_levels = roi_levels.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc = lib().dtc_roi_align_forward_ordered(
            lv, len(features), ch, _dtype_code(dt), rois.data_ptr(), cols if R else 5,
            roi_levels.data_ptr() if roi_levels is not None else None,
            roi_order.to(torch.int32).contiguous().data_ptr() if roi_order is not None else None, R, int(pooled_h),
            int(pooled_w), int(sampling_ratio), out.data_ptr(), _dtype_code(odt), stream_ptr(dev))
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


def make_rpn_levels(cls_probs, bbox_preds, anchors, feat_strides, pre_nms_top_n, scores_are_logits=False):
    """lists (one entry per level) of [B,A,H,W] / [B,4A,H,W] float32 CUDA tensors + base anchors [A,4] -> RpnLevel array.
    scores_are_logits: cls_probs holds the pre-sigmoid RPN logits (dtc_rpn_level.score_is_logit)."""
    n = len(cls_probs)
    arr = (RpnLevel * n)()
    keep_alive = []
    for k in range(n):
        c, d = cls_probs[k].contiguous(), bbox_preds[k].contiguous()
        if c.dtype != torch.float32 or d.dtype != torch.float32:
            raise TypeError("RPN outputs must be float32")
        B, A, H, W = c.shape
        if d.shape != (B, 4 * A, H, W):
            raise ValueError("rpn_bbox_pred must be [B,4A,H,W]")
        a = [float(v) for v in anchors[k].reshape(-1)]
        if len(a) != 4 * A or A > 16:
            raise ValueError("need A<=16 base anchors of 4 coordinates")
        lv = RpnLevel(c.data_ptr(), d.data_ptr(), A, H, W, int(pre_nms_top_n[k]), float(feat_strides[k]),
                      1 if scores_are_logits else 0)
        for q, v in enumerate(a):
            lv.anchors[q] = v
        arr[k] = lv
        keep_alive += [c, d]
    return arr, keep_alive


def generate_proposals(cls_probs, bbox_preds, anchors, feat_strides, im_h, im_w, pre_nms_top_n, post_nms_top_n,
                       nms_thresh, min_size_scaled=0.0, scores_are_logits=False):
    """Batched multi-level GenerateProposals (generate_proposals.py:31-122) with zero host round trips.
    scores_are_logits=True folds the RPN head's sigmoid (detector.py:125) into the top-k: pass the raw logits, get the
    proposals and PROBABILITY scores the reference would produce from sigmoid(logits), without materialising them.

    Returns (boxes [B,L,P,4], scores [B,L,P], counts int32 [B,L]) with P = post_nms_top_n (rows >= count undefined),
    plus the pre-NMS (sorted) boxes/scores/counts for callers that want them.
    """
    dev = _require_cuda(*cls_probs, *bbox_preds)
    L_ = lib()
    nl = len(cls_probs)
    B = cls_probs[0].shape[0]
    lv, alive = make_rpn_levels(cls_probs, bbox_preds, anchors, feat_strides, pre_nms_top_n, scores_are_logits)
    kmax = 0
    for k in range(nl):
        N = cls_probs[k].shape[1] * cls_probs[k].shape[2] * cls_probs[k].shape[3]
        K = N if (pre_nms_top_n[k] <= 0 or pre_nms_top_n[k] >= N) else int(pre_nms_top_n[k])
        kmax = max(kmax, K)
    S = B * nl
    ws = workspace(L_.dtc_rpn_topk_decode_workspace_bytes(lv, nl, B, kmax), dev)
    pre_boxes = torch.empty((S, kmax, 4), dtype=torch.float32, device=dev)
    pre_scores = torch.empty((S, kmax), dtype=torch.float32, device=dev)
    pre_counts = torch.empty((S,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L_.dtc_rpn_topk_decode(lv, nl, B, float(im_h), float(im_w), float(min_size_scaled), ws.data_ptr(),
                                    ws.numel(), pre_boxes.data_ptr(), pre_scores.data_ptr(), pre_counts.data_ptr(),
                                    kmax, stream_ptr(dev))
    check(rc, "dtc_rpn_topk_decode")
    if nms_thresh <= 0:                                   # generate_proposals.py:114
        return (pre_boxes.view(B, nl, kmax, 4), pre_scores.view(B, nl, kmax), pre_counts.view(B, nl),
                pre_boxes, pre_scores, pre_counts)
    P = int(post_nms_top_n) if post_nms_top_n > 0 else kmax
    P = min(P, kmax)
    keep, kcnt = nms_sorted(pre_boxes, pre_counts, nms_thresh, max_keep=P, keep_stride=P)
    out_boxes = torch.empty((S, P, 4), dtype=torch.float32, device=dev)
    out_scores = torch.empty((S, P), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = L_.dtc_gather_kept(pre_boxes.data_ptr(), pre_scores.data_ptr(), S, kmax, keep.data_ptr(), kcnt.data_ptr(),
                                P, out_boxes.data_ptr(), out_scores.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_gather_kept")
    del alive
    return (out_boxes.view(B, nl, P, 4), out_scores.view(B, nl, P), kcnt.view(B, nl), pre_boxes, pre_scores, pre_counts)


def fpn_collect_distribute(boxes, scores, counts, post_nms_top_n, k_min=2, k_max=5, inputs_sorted=False):
    """dtc_fpn_collect_distribute.  boxes [B,L,P,4], scores [B,L,P] or None, counts int32 [B,L].
    -> dict(rois5 [B,T,5], roi_scores, roi_levels [B,T], n_out [B], rois_by_level [B,T,4], level_counts [B,nl],
            idx_restore [B,T])"""
    dev = _require_cuda(boxes, scores, counts)
    boxes = boxes.contiguous()
    B, Lin, P = boxes.shape[0], boxes.shape[1], boxes.shape[2]
    T = int(post_nms_top_n)
    nl = k_max - k_min + 1
    f32, i32 = torch.float32, torch.int32
    out = dict(rois5=torch.empty((B, T, 5), dtype=f32, device=dev),
               roi_scores=torch.empty((B, T), dtype=f32, device=dev) if scores is not None else None,
               roi_levels=torch.empty((B, T), dtype=i32, device=dev), n_out=torch.empty((B,), dtype=i32, device=dev),
               rois_by_level=torch.empty((B, T, 4), dtype=f32, device=dev),
               level_counts=torch.empty((B, nl), dtype=i32, device=dev),
               idx_restore=torch.empty((B, T), dtype=i32, device=dev),
               roi_order=torch.empty((B, T), dtype=i32, device=dev),
               roi_desc=torch.empty((B, T, 8), dtype=f32, device=dev))
    if scores is not None:
        scores = scores.contiguous()
    counts = counts.to(i32).contiguous()
    with torch.cuda.device(dev):
        rc = lib().dtc_fpn_collect_distribute(boxes.data_ptr(), _ptr(scores), counts.data_ptr(), B, Lin, P, T, k_min,
                                              k_max, out["rois5"].data_ptr(), _ptr(out["roi_scores"]),
                                              out["roi_levels"].data_ptr(), out["n_out"].data_ptr(),
                                              out["rois_by_level"].data_ptr(), out["level_counts"].data_ptr(),
                                              out["idx_restore"].data_ptr(), out["roi_order"].data_ptr(),
                                              out["roi_desc"].data_ptr(), 1 if inputs_sorted else 0, stream_ptr(dev))
    check(rc, "dtc_fpn_collect_distribute")
    return out


def postprocess_detections(rois5, n_rois, cls_score, bbox_pred, scaling_factor, im_size, weights=(10., 10., 5., 5.),
                           score_thresh=0.05, nms_thresh=0.5, max_det=100, max_out=None, ws=None, scores_are_logits=False):
    """dtc_postprocess_detections.  rois5 [B,R,5], cls_score [B,R,C], bbox_pred [B,R,4C], scaling_factor [B], im_size [B,2].
    scores_are_logits=True: cls_score holds the raw cls_score-layer output and the softmax of detector.py:281 is folded into
    the kernel (dtc_postprocess_detections_logits); the probability map is never materialised.
    -> (dets [B,max_out,6], det_roi [B,max_out], det_rois_scaled [B,max_out,4], det_count [B])"""
    dev = _require_cuda(rois5, n_rois, cls_score, bbox_pred, scaling_factor, im_size)
    B, R, ncls = cls_score.shape
    if max_out is None:
        max_out = 128 if max_det > 0 else R * (ncls - 1)
    L_ = lib()
    need = L_.dtc_postprocess_detections_workspace_bytes(B, R, ncls)
    if ws is None or ws.numel() < need:
        ws = workspace(need, dev)
    f32, i32 = torch.float32, torch.int32
    dets = torch.zeros((B, max_out, 6), dtype=f32, device=dev)
    det_roi = torch.zeros((B, max_out), dtype=i32, device=dev)
    det_scaled = torch.zeros((B, max_out, 4), dtype=f32, device=dev)
    det_count = torch.empty((B,), dtype=i32, device=dev)
    rois5, cls_score, bbox_pred = rois5.contiguous(), cls_score.contiguous(), bbox_pred.contiguous()
    scaling_factor, im_size = scaling_factor.to(f32).contiguous(), im_size.to(f32).contiguous()
    with torch.cuda.device(dev):
        fn = L_.dtc_postprocess_detections_logits if scores_are_logits else L_.dtc_postprocess_detections
        rc = fn(rois5.data_ptr(), _ptr(n_rois), cls_score.data_ptr(), bbox_pred.data_ptr(),
                                           scaling_factor.data_ptr(), im_size.data_ptr(), B, R, ncls,
                                           *[float(w) for w in weights], float(score_thresh), float(nms_thresh),
                                           int(max_det), ws.data_ptr(), ws.numel(), dets.data_ptr(), det_roi.data_ptr(),
                                           det_scaled.data_ptr(), det_count.data_ptr(), int(max_out), stream_ptr(dev))
    check(rc, "dtc_postprocess_detections")
    return dets, det_roi, det_scaled, det_count


def box_results_nms_limit(scores, boxes, n_rois=None, score_thresh=0.05, nms_thresh=0.5, max_det=100, max_out=None, ws=None):
    """dtc_box_results_nms_limit: scores [B,R,C], decoded boxes [B,R,4C] -> (dets [B,max_out,6], det_roi [B,max_out], det_count [B])"""
    dev = _require_cuda(scores, boxes, n_rois)
    B, R, ncls = scores.shape
    if max_out is None:
        max_out = 128 if max_det > 0 else R * (ncls - 1)
    L_ = lib()
    need = L_.dtc_postprocess_detections_workspace_bytes(B, R, ncls)
    if ws is None or ws.numel() < need:
        ws = workspace(need, dev)
    dets = torch.zeros((B, max_out, 6), dtype=torch.float32, device=dev)
    det_roi = torch.zeros((B, max_out), dtype=torch.int32, device=dev)
    det_count = torch.empty((B,), dtype=torch.int32, device=dev)
    scores, boxes = scores.contiguous(), boxes.contiguous()
    with torch.cuda.device(dev):
        rc = L_.dtc_box_results_nms_limit(scores.data_ptr(), boxes.data_ptr(), _ptr(n_rois), B, R, ncls, float(score_thresh),
                                          float(nms_thresh), int(max_det), ws.data_ptr(), ws.numel(), dets.data_ptr(),
                                          det_roi.data_ptr(), det_count.data_ptr(), int(max_out), stream_ptr(dev))
    check(rc, "dtc_box_results_nms_limit")
    return dets, det_roi, det_count


def mask_paste(masks, dets, det_count, im_size, M, per_image_capacity, mask_index=None, thresh=0.5, cls_specific=True):
    """dtc_mask_paste -> dict(crops uint8 [B,cap], boxes int32 [B,D,4], rects int32 [B,D,4], offsets int64 [B,D], bytes int64 [B])"""
    dev = _require_cuda(masks, dets, det_count, im_size, mask_index)
    masks, dets = masks.contiguous(), dets.contiguous()
    B, D = dets.shape[0], dets.shape[1]
    cap = int(per_image_capacity)
    out = dict(crops=torch.empty((B, max(cap, 1)), dtype=torch.uint8, device=dev),
               boxes=torch.zeros((B, D, 4), dtype=torch.int32, device=dev),
               rects=torch.zeros((B, D, 4), dtype=torch.int32, device=dev),
               offsets=torch.zeros((B, D), dtype=torch.int64, device=dev),
               bytes=torch.zeros((B,), dtype=torch.int64, device=dev))
    with torch.cuda.device(dev):
        rc = lib().dtc_mask_paste(masks.data_ptr(), _ptr(mask_index), masks.shape[1], int(M), dets.data_ptr(),
                                  det_count.data_ptr(), im_size.to(torch.float32).contiguous().data_ptr(), B, D,
                                  float(thresh), 1 if cls_specific else 0, out["crops"].data_ptr(), cap,
                                  out["boxes"].data_ptr(), out["rects"].data_ptr(), out["offsets"].data_ptr(),
                                  out["bytes"].data_ptr(), stream_ptr(dev))
    check(rc, "dtc_mask_paste")
    return out


def mask_rle(paste, det_count, im_size, runs_stride=4096, str_stride=8192):
    """dtc_mask_rle on the dict returned by mask_paste -> dict(counts uint32 [B,D,runs_stride], n_runs int32 [B,D],
    str uint8 [B,D,str_stride], str_len int32 [B,D]).  Negative n_runs / str_len: that detection did not fit."""
    crops, rects, offs = paste["crops"], paste["rects"], paste["offsets"]
    dev = _require_cuda(crops, rects, offs, det_count, im_size)
    B, D = rects.shape[0], rects.shape[1]
    out = dict(counts=torch.empty((B, D, int(runs_stride)), dtype=torch.int32, device=dev),
               n_runs=torch.zeros((B, D), dtype=torch.int32, device=dev),
               str=torch.empty((B, D, int(str_stride)), dtype=torch.uint8, device=dev),
               str_len=torch.zeros((B, D), dtype=torch.int32, device=dev))
    with torch.cuda.device(dev):
        rc = lib().dtc_mask_rle(crops.data_ptr(), int(crops.shape[1]), rects.data_ptr(), offs.data_ptr(),
                                det_count.data_ptr(), im_size.to(torch.float32).contiguous().data_ptr(), B, D,
                                out["counts"].data_ptr(), int(runs_stride), out["n_runs"].data_ptr(),
                                out["str"].data_ptr(), int(str_stride), out["str_len"].data_ptr(), stream_ptr(dev))
    check(rc, "dtc_mask_rle")
    return out


def bbox_overlaps(boxes, query_boxes):
    """dtc_bbox_overlaps: [N,>=4], [K,>=4] float32 CUDA -> [N,K] float32 (cython_bbox.pyx:32-72)."""
    dev = _require_cuda(boxes, query_boxes)
    boxes, query_boxes = boxes.contiguous(), query_boxes.contiguous()
    n, k = boxes.shape[0], query_boxes.shape[0]
    out = torch.zeros((n, k), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().dtc_bbox_overlaps(boxes.data_ptr(), n, boxes.shape[1] if n else 4, query_boxes.data_ptr(), k,
                                     query_boxes.shape[1] if k else 4, out.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_bbox_overlaps")
    return out


def box_voting(top_dets, all_dets, thresh):
    """dtc_box_voting ('ID' scoring): [T,5], [A,5] float32 CUDA -> ([T,5], n_voters int32 [T])."""
    dev = _require_cuda(top_dets, all_dets)
    top_dets, all_dets = top_dets.contiguous(), all_dets.contiguous()
    t, a = top_dets.shape[0], all_dets.shape[0]
    out = torch.empty((t, 5), dtype=torch.float32, device=dev)
    nv = torch.zeros((max(t, 1),), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().dtc_box_voting(top_dets.data_ptr(), t, all_dets.data_ptr(), a, float(thresh), out.data_ptr(),
                                  nv.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_box_voting")
    return out, nv[:t]


def prep_images(images, pixel_means=(122.7717, 115.9465, 102.9801), target_size=800, max_size=1333, pad_stride=32):
    """dtc_prep_plan + dtc_prep_images: list of HWC BGR CUDA tensors (uint8 or float32, [h,w,3]) ->
    (blob float32 [B,3,Hb,Wb] on the device, im_scales list of float, resized sizes list of (h, w))."""
    dev = _require_cuda(*images)
    B = len(images)
    ims = [im if im.stride(2) == 1 and im.stride(1) == 3 else im.contiguous() for im in images]
    hs = (C.c_int32 * B)(*[int(im.shape[0]) for im in ims])
    ws_ = (C.c_int32 * B)(*[int(im.shape[1]) for im in ims])
    scales = (C.c_double * B)()
    out_hw = (C.c_int32 * (2 * B))()
    blob_hw = (C.c_int32 * 2)()
    check(lib().dtc_prep_plan(hs, ws_, B, int(target_size), int(max_size), int(pad_stride), scales, out_hw, blob_hw),
          "dtc_prep_plan")
    arr = (Image * B)()
    for k, im in enumerate(ims):
        if im.dtype == torch.uint8:
            dt = 2
        elif im.dtype == torch.float32:
            dt = 0
        else:
            raise TypeError("images must be uint8 or float32")
        if im.dim() != 3 or im.shape[2] != 3:
            raise ValueError("images must be [h,w,3] (BGR)")
        arr[k] = Image(im.data_ptr(), int(im.shape[0]), int(im.shape[1]), dt, int(im.stride(0)))
    means = (C.c_double * 3)(*[float(m) for m in pixel_means])
    blob = torch.empty((B, 3, blob_hw[0], blob_hw[1]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().dtc_prep_images(arr, B, means, scales, out_hw, blob.data_ptr(), blob_hw[0], blob_hw[1], stream_ptr(dev))
    check(rc, "dtc_prep_images")
    del ims
    return blob, [float(s) for s in scales], [(out_hw[2 * k], out_hw[2 * k + 1]) for k in range(B)]


def soft_nms(dets, sigma, overlap_thresh, score_thresh, method):
    """dtc_soft_nms: dets [N,5] float32 CUDA -> (dets' [N',5], inds int64 [N'])."""
    dev = _require_cuda(dets)
    dets = dets.contiguous()
    n = dets.shape[0]
    out = torch.empty((max(n, 1), 5), dtype=torch.float32, device=dev)
    inds = torch.empty((max(n, 1),), dtype=torch.int64, device=dev)
    cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib().dtc_soft_nms(dets.data_ptr(), n, float(sigma), float(overlap_thresh), float(score_thresh), int(method),
                                out.data_ptr(), inds.data_ptr(), cnt.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_soft_nms")
    k = int(cnt.item())
    return out[:k], inds[:k]


def bbox_transform(boxes, deltas, weights, clip_to=None):
    """dtc_bbox_transform: boxes [N,4], deltas [N,4K] CUDA float32 -> [N,4K]; clip_to=(im_h, im_w) also clips."""
    dev = _require_cuda(boxes, deltas)
    boxes, deltas = boxes.contiguous(), deltas.contiguous()
    n, k = deltas.shape[0], deltas.shape[1] // 4
    out = torch.empty_like(deltas)
    with torch.cuda.device(dev):
        rc = lib().dtc_bbox_transform(boxes.data_ptr(), deltas.data_ptr(), n, k, *[float(w) for w in weights],
                                      1 if clip_to is not None else 0, float(clip_to[0]) if clip_to is not None else 0.0,
                                      float(clip_to[1]) if clip_to is not None else 0.0, out.data_ptr(), stream_ptr(dev))
    check(rc, "dtc_bbox_transform")
    return out
