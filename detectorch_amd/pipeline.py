"""Device-resident region-proposal hot path for a BATCH of images (Mask R-CNN FPN flavour, BASELINE cfg3/cfg4).

This is the MI355X-first composition of the kernels behind the reference-shaped modules: everything the reference does
between the RPN-head outputs and the box/mask-head GEMMs, and after them, for B images at once, with zero host round
trips and (optionally) replayed from one hipGraph:

    RPN outputs (5 levels) --[rpn_topk_decode, nms_sorted, gather]--> per-level proposals      generate_proposals.py:31-122
      --[fpn_collect_distribute]--> rois5 [B,1000,5] + level ids                               collect_and_distribute...py:84-128
      --[roi_align 7x7, all levels, one launch]--> box-head features [B*1000,256,7,7]          detector.py:263-270
      (box head fc6/fc7/cls/bbox: hipBLASLt GEMMs, not part of the hot path -- supplied by the caller / synthetic)
      --[postprocess_detections]--> dets [B,128,6]                                             result_utils.py:76-168
      --[fpn_collect_distribute (no sort)]--> mask-branch level ids                            multilevel_rois.py:19-39
      --[roi_align 14x14]--> mask-head features [B*128,256,14,14]                              detector.py:99-106
      (mask head convs: MIOpen, not part of the hot path -- supplied by the caller / synthetic)
      --[mask_paste]--> binarised crops                                                        result_utils.py:170-214
"""
import numpy as np
import torch

from . import hip, synth
from .utils.generate_anchors import generate_anchors


class FpnRegionPath:
    def __init__(self, batch, device, channels=256, n_cls=81, pre_nms_top_n=1000, post_nms_top_n=1000,
                 collect_top_n=1000, rpn_nms_thresh=0.7, max_det=100, max_out=128, mask_res=28,
                 box_pooled=7, mask_pooled=14, sampling_ratio=2, pad_h=synth.FPN_PAD_H, pad_w=synth.FPN_PAD_W,
                 feat_dtype=torch.float32, crop_capacity=8 << 20, cls_logits=False, with_rle=False,
                 rle_runs_stride=4096, rle_str_stride=8192):
        self.B, self.dev = batch, device
        self.with_rle, self.rle_runs_stride, self.rle_str_stride = with_rle, int(rle_runs_stride), int(rle_str_stride)
        self.cls_logits = cls_logits       # bind() receives the cls_score layer's raw output; softmax folded into the kernel
        self.C, self.n_cls = channels, n_cls
        self.pre, self.post, self.top_n = pre_nms_top_n, post_nms_top_n, collect_top_n
        self.rpn_thresh, self.max_det, self.max_out, self.M = rpn_nms_thresh, max_det, max_out, mask_res
        self.box_p, self.mask_p, self.sr = box_pooled, mask_pooled, sampling_ratio
        self.pad_h, self.pad_w = pad_h, pad_w
        self.shapes = synth.fpn_level_shapes(pad_h, pad_w)
        self.strides = [float(s) for s in synth.FPN_STRIDES]
        self.anchors = [generate_anchors(stride=self.strides[l], sizes=(32.0 * 2 ** l,), aspect_ratios=(0.5, 1, 2))
                        for l in range(5)]                         # detector.py:203-205
        self.roi_scales = list(synth.FPN_ROI_SCALES)
        self.feat_dtype = feat_dtype
        self.crop_capacity = crop_capacity
        self.graph = None
        self._alloc()

    # ---- buffers (allocated once; the step itself never allocates) ---------------------------------------------------
    def _alloc(self):
        B, dev, f32, i32 = self.B, self.dev, torch.float32, torch.int32
        L = hip.lib()
        S = B * 5
        self.kmax = self.pre
        e = lambda *shape, dtype=f32: torch.empty(shape, dtype=dtype, device=dev)
        self.pre_boxes, self.pre_scores, self.pre_counts = e(S, self.kmax, 4), e(S, self.kmax), e(S, dtype=i32)
        self.P = min(self.post, self.kmax)
        self.keep, self.keep_cnt = e(S, self.P, dtype=i32), e(S, dtype=i32)
        self._prop_boxes, self._prop_scores = torch.zeros((S, self.P, 4), device=dev), torch.zeros((S, self.P), device=dev)
        # collect reads proposals[keep] in place (dtc_fpn_collect_distribute_kept) where its merge kernel holds the shape; otherwise the
        # gather launch + the plain entry point
        self.fused_gather = self.top_n <= 2048 and self.P <= 1024 and 5 * self.P <= 8192
        self.nms_ws = hip.workspace(L.dtc_nms_sorted_workspace_bytes(S, self.kmax), dev)
        T = self.top_n
        self.rois5, self.roi_scores = e(B, T, 5), e(B, T)
        self.roi_levels, self.n_rois = e(B, T, dtype=i32), e(B, dtype=i32)
        self.rois_by_level, self.level_counts, self.idx_restore = e(B, T, 4), e(B, 4, dtype=i32), e(B, T, dtype=i32)
        self.roi_order, self.roi_desc = e(B, T, dtype=i32), e(B, T, 8)
        self.box_feats = e(B * T, self.C, self.box_p, self.box_p, dtype=self.feat_dtype)
        D = self.max_out
        self.dets, self.det_roi = torch.zeros((B, D, 6), device=dev), torch.zeros((B, D), dtype=i32, device=dev)
        self.det_scaled, self.det_count = torch.zeros((B, D, 4), device=dev), e(B, dtype=i32)
        self.det_ws = hip.workspace(L.dtc_postprocess_detections_workspace_bytes(B, T, self.n_cls), dev)
        self.det_count_c = e(B, 1, dtype=i32)
        self.m_rois5, self.m_levels, self.m_n = e(B, D, 5), e(B, D, dtype=i32), e(B, dtype=i32)
        self.m_by_level, self.m_level_counts, self.m_restore = e(B, D, 4), e(B, 4, dtype=i32), e(B, D, dtype=i32)
        self.m_order, self.m_desc = e(B, D, dtype=i32), e(B, D, 8)
        # the mask branch's level mapping comes out of the detection launch itself (dtc_postprocess_detections_fpn, <= 512 rows)
        self.fused_mask_map = D <= 512
        self.m_map = hip.FpnMapOut(self.m_rois5.data_ptr(), self.m_levels.data_ptr(), self.m_n.data_ptr(), self.m_by_level.data_ptr(),
                                   self.m_level_counts.data_ptr(), self.m_restore.data_ptr(), self.m_order.data_ptr(),
                                   self.m_desc.data_ptr(), 2, 5)
        self.mask_feats = e(B * D, self.C, self.mask_p, self.mask_p, dtype=self.feat_dtype)
        self.crops = torch.empty((B, self.crop_capacity), dtype=torch.uint8, device=dev)
        self.mask_boxes, self.mask_rects = torch.zeros((B, D, 4), dtype=i32, device=dev), torch.zeros((B, D, 4), dtype=i32, device=dev)
        self.mask_offsets, self.mask_bytes = torch.zeros((B, D), dtype=torch.int64, device=dev), torch.zeros((B,), dtype=torch.int64, device=dev)
        if self.with_rle:    # COCO RLE of every pasted mask, on the device (dtc_mask_rle): ~100 bytes per mask leave the GPU
            self.rle_counts = e(B, D, self.rle_runs_stride, dtype=i32)
            self.rle_n_runs, self.rle_str_len = torch.zeros((B, D), dtype=i32, device=dev), torch.zeros((B, D), dtype=i32, device=dev)
            self.rle_str = torch.zeros((B, D, self.rle_str_stride), dtype=torch.uint8, device=dev)

    def bind(self, rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, scaling_factor, im_size):
        """Attach the (device) inputs of one batch.  Pointers are baked into the launch descriptors (and the graph), so
        new data is COPIED into these tensors between steps, like any static-shape serving loop."""
        self.bind_rpn(rpn_cls, rpn_bbox, feats)
        self.bind_heads(cls_score, bbox_pred, scaling_factor, im_size)
        self.bind_masks(masks)

    # The three stages can also be bound / launched one by one by a model that runs its head GEMMs / convs in between
    # (detectorch_amd.model.detector.forward_batched):  launch_proposals -> box head -> launch_detections -> mask head ->
    # launch_masks.
    def bind_rpn(self, rpn_cls, rpn_bbox, feats, scores_are_logits=False):
        self.rpn_cls, self.rpn_bbox, self.feats = rpn_cls, rpn_bbox, feats
        self.rpn_lv, self._alive = hip.make_rpn_levels(rpn_cls, rpn_bbox, self.anchors, self.strides, [self.pre] * 5,
                                                       scores_are_logits=scores_are_logits)
        need = hip.lib().dtc_rpn_topk_decode_workspace_bytes(self.rpn_lv, 5, self.B, self.kmax)
        if getattr(self, "rpn_ws", None) is None or self.rpn_ws.numel() < need:
            self.rpn_ws = hip.workspace(need, self.dev)
        self.feat_lv, _, _ = hip.make_levels(feats, self.roi_scales)
        self.feat_code = hip._dtype_code(feats[0].dtype)
        self.out_code = hip._dtype_code(self.feat_dtype)
        self.graph = None

    def bind_heads(self, cls_score, bbox_pred, scaling_factor, im_size):
        self.cls_score, self.bbox_pred = cls_score, bbox_pred
        self.sf, self.im_size = scaling_factor, im_size
        self.graph = None

    def bind_masks(self, masks):
        self.masks = masks
        self.graph = None

    # ---- one pass of the hot path over the bound batch ---------------------------------------------------------------
    def launch_proposals(self, st=None):
        """RPN outputs -> rois5 / level ids / visiting order -> box-head features (self.box_feats [B*T, C, 7, 7])."""
        L, B, ck = hip.lib(), self.B, hip.check
        st = st or hip.stream_ptr(self.dev)
        S, T = B * 5, self.top_n
        ck(L.dtc_rpn_topk_decode(self.rpn_lv, 5, B, float(self.pad_h), float(self.pad_w), 0.0, self.rpn_ws.data_ptr(),
                                 self.rpn_ws.numel(), self.pre_boxes.data_ptr(), self.pre_scores.data_ptr(),
                                 self.pre_counts.data_ptr(), self.kmax, st), "rpn_topk_decode")
        ck(L.dtc_nms_sorted(self.pre_boxes.data_ptr(), self.pre_counts.data_ptr(), S, self.kmax, self.rpn_thresh, self.P,
                            self.nms_ws.data_ptr(), self.nms_ws.numel(), self.keep.data_ptr(), self.P,
                            self.keep_cnt.data_ptr(), st), "nms_sorted")
        if self.fused_gather:
            ck(L.dtc_fpn_collect_distribute_kept(self.pre_boxes.data_ptr(), self.pre_scores.data_ptr(), self.kmax, self.keep.data_ptr(),
                                                 self.keep_cnt.data_ptr(), self.P, B, 5, T, 2, 5, self.rois5.data_ptr(),
                                                 self.roi_scores.data_ptr(), self.roi_levels.data_ptr(), self.n_rois.data_ptr(),
                                                 self.rois_by_level.data_ptr(), self.level_counts.data_ptr(), self.idx_restore.data_ptr(),
                                                 self.roi_order.data_ptr(), self.roi_desc.data_ptr(), st), "fpn_collect_kept")
        else:
            self._gather_kept(st)
            ck(L.dtc_fpn_collect_distribute(self._prop_boxes.data_ptr(), self._prop_scores.data_ptr(), self.keep_cnt.data_ptr(),
                                            B, 5, self.P, T, 2, 5, self.rois5.data_ptr(), self.roi_scores.data_ptr(),
                                            self.roi_levels.data_ptr(), self.n_rois.data_ptr(), self.rois_by_level.data_ptr(),
                                            self.level_counts.data_ptr(), self.idx_restore.data_ptr(),
                                            self.roi_order.data_ptr(), self.roi_desc.data_ptr(), 1, st), "fpn_collect")
        self._roi_align_box(st)

    def _gather_kept(self, st=None):
        st = st or hip.stream_ptr(self.dev)
        hip.check(hip.lib().dtc_gather_kept(self.pre_boxes.data_ptr(), self.pre_scores.data_ptr(), self.B * 5, self.kmax, self.keep.data_ptr(),
                                            self.keep_cnt.data_ptr(), self.P, self._prop_boxes.data_ptr(), self._prop_scores.data_ptr(), st),
                  "gather_kept")

    # per-level proposals after NMS (generate_proposals.py:119-120).  The fused step no longer materialises them (collect reads
    # proposals[keep] in place); readers -- the parity checks -- get them from the gather kernel on demand.
    @property
    def prop_boxes(self):
        if self.fused_gather:
            with torch.cuda.device(self.dev):
                self._gather_kept()
        return self._prop_boxes

    @property
    def prop_scores(self):
        if self.fused_gather:
            with torch.cuda.device(self.dev):
                self._gather_kept()
        return self._prop_scores

    def launch_detections(self, st=None):
        """cls_score (probabilities, or logits with cls_logits=True) + bbox_pred -> dets -> mask-head features."""
        L, B, ck = hip.lib(), self.B, hip.check
        st = st or hip.stream_ptr(self.dev)
        T, D = self.top_n, self.max_out
        if self.fused_mask_map:
            ck(L.dtc_postprocess_detections_fpn(self.rois5.data_ptr(), self.n_rois.data_ptr(), self.cls_score.data_ptr(),
                                                1 if self.cls_logits else 0, self.bbox_pred.data_ptr(), self.sf.data_ptr(),
                                                self.im_size.data_ptr(), B, T, self.n_cls, 10.0, 10.0, 5.0, 5.0, 0.05, 0.5, self.max_det,
                                                self.det_ws.data_ptr(), self.det_ws.numel(), self.dets.data_ptr(), self.det_roi.data_ptr(),
                                                self.det_scaled.data_ptr(), self.det_count.data_ptr(), D, self.m_map, st),
               "postprocess_detections_fpn")
            self._roi_align_mask(st)
            return
        post = L.dtc_postprocess_detections_logits if self.cls_logits else L.dtc_postprocess_detections
        ck(post(self.rois5.data_ptr(), self.n_rois.data_ptr(), self.cls_score.data_ptr(),
                self.bbox_pred.data_ptr(), self.sf.data_ptr(), self.im_size.data_ptr(), B, T,
                self.n_cls, 10.0, 10.0, 5.0, 5.0, 0.05, 0.5, self.max_det,
                self.det_ws.data_ptr(), self.det_ws.numel(), self.dets.data_ptr(),
                self.det_roi.data_ptr(), self.det_scaled.data_ptr(), self.det_count.data_ptr(), D, st),
           "postprocess_detections")
        # mask branch: level ids of the (scaled) detection boxes, multilevel_rois.py:19-39
        ck(L.dtc_fpn_collect_distribute(self.det_scaled.data_ptr(), None, self.det_count.data_ptr(), B, 1, D, D, 2, 5,
                                        self.m_rois5.data_ptr(), None, self.m_levels.data_ptr(), self.m_n.data_ptr(),
                                        self.m_by_level.data_ptr(), self.m_level_counts.data_ptr(),
                                        self.m_restore.data_ptr(), self.m_order.data_ptr(), self.m_desc.data_ptr(), 0, st),
           "fpn_map_levels")
        self._roi_align_mask(st)

    def launch_masks(self, st=None):
        """mask-head outputs [B*D, n_cls, M, M] -> binarised crops (+ COCO RLE strings on the device with with_rle=True)."""
        L, B, ck = hip.lib(), self.B, hip.check
        st = st or hip.stream_ptr(self.dev)
        D = self.max_out
        ck(L.dtc_mask_paste(self.masks.data_ptr(), None, self.n_cls, self.M, self.dets.data_ptr(), self.det_count.data_ptr(),
                            self.im_size.data_ptr(), B, D, 0.5, 1, self.crops.data_ptr(), self.crop_capacity,
                            self.mask_boxes.data_ptr(), self.mask_rects.data_ptr(), self.mask_offsets.data_ptr(),
                            self.mask_bytes.data_ptr(), st), "mask_paste")
        if self.with_rle:
            ck(L.dtc_mask_rle(self.crops.data_ptr(), self.crop_capacity, self.mask_rects.data_ptr(), self.mask_offsets.data_ptr(),
                              self.det_count.data_ptr(), self.im_size.data_ptr(), B, D, self.rle_counts.data_ptr(),
                              self.rle_runs_stride, self.rle_n_runs.data_ptr(), self.rle_str.data_ptr(), self.rle_str_stride,
                              self.rle_str_len.data_ptr(), st), "mask_rle")

    def _launch(self):
        st = hip.stream_ptr(self.dev)
        self.launch_proposals(st)
        self.launch_detections(st)
        self.launch_masks(st)

    def _roi_align_box(self, st=None):
        st = st or hip.stream_ptr(self.dev)
        hip.check(hip.lib().dtc_roi_align_forward_packed(self.feat_lv, 4, self.C, self.feat_code, self.roi_desc.data_ptr(),
                                                  self.B * self.top_n, self.box_p, self.box_p,
                                                  self.sr, self.box_feats.data_ptr(), self.out_code, st), "roi_align(box)")

    def _roi_align_mask(self, st=None):
        st = st or hip.stream_ptr(self.dev)
        hip.check(hip.lib().dtc_roi_align_forward_packed(self.feat_lv, 4, self.C, self.feat_code, self.m_desc.data_ptr(),
                                                  self.B * self.max_out, self.mask_p, self.mask_p,
                                                  self.sr, self.mask_feats.data_ptr(), self.out_code, st), "roi_align(mask)")

    def step(self, use_graph=True):
        """One pass over the bound batch on the current stream.  With use_graph the launch sequence is captured once into
        a hipGraph (after an eager warm-up call) and replayed: the path is ~17 short launches, i.e. launch-bound."""
        with torch.cuda.device(self.dev):
            if not use_graph:
                self._launch()
                return
            if self.graph is None:
                self._launch()                       # eager warm-up (sets function attributes, primes allocations)
                torch.cuda.synchronize(self.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch()
                self.graph = g
            self.graph.replay()

    # ---- algorithmic (compulsory) bytes, SURVEY.md section 8(d) ------------------------------------------------------
    def box_roialign_bytes(self):
        fb = sum(f.numel() * f.element_size() for f in self.feats)
        return fb + self.B * self.top_n * 5 * 4 + self.box_feats.numel() * self.box_feats.element_size()

    def results(self, on_overflow="raise"):
        """Host copy of the detections of the last step: list of dict(boxes, scores, classes) per image.

        The reference keeps every detection that ties at the image threshold (result_utils.py:159-163); the fixed-shape rows hold
        max_out of them.  More ties than rows must not disappear silently: on_overflow="raise" (default: an evaluation run must stop),
        or "truncate" for a serving loop that must not abort on a rare tie -- the first max_out rows (class-major order) are returned,
        a warning is issued and the image's dict carries truncated=True and n_detections = the true count."""
        if on_overflow not in ("raise", "truncate"):
            raise ValueError("on_overflow must be 'raise' or 'truncate'")
        cnt = self.det_count.cpu().numpy()
        dets = self.dets.cpu().numpy()
        out = []
        for b in range(self.B):
            n, over = int(cnt[b]), int(cnt[b]) > self.max_out
            if over:
                msg = ("image %d has %d detections (ties at the image threshold) but max_out = %d rows: raise max_out"
                       % (b, n, self.max_out))
                if on_overflow == "raise":
                    raise RuntimeError(msg)
                import warnings
                warnings.warn(msg + " -- truncated", RuntimeWarning)
                n = self.max_out
            d = dict(boxes=dets[b, :n, :4].copy(), scores=dets[b, :n, 4].copy(), classes=dets[b, :n, 5].astype(np.int32))
            if over:
                d["truncated"], d["n_detections"] = True, int(cnt[b])
            out.append(d)
        return out


class C4RegionPath:
    """The region-proposal hot path of the C4 flavour (BASELINE configs[1]: Faster R-CNN R-50-C4, "1000 RPN proposals,
    RoIAlign 7x7 + NMS only") for a batch of images, device-resident, one hipGraph:

        RPN outputs [B,15,50,84] / [B,60,50,84] --[rpn_topk_decode 63 000 -> 6000, nms_sorted 0.7 -> 1000, gather]-->  generate_proposals.py:31-122
          --[collect (1 level: rois5 + visiting order)]--> rois5 [B,1000,5]
          --[roi_align on res4 [B,1024,50,84], adaptive sampling (sampling_ratio 0)]--> [B*1000,1024,P,P]               detector.py:240-248
          (res5 head + cls/bbox: convs / GEMMs, not part of the hot path -- synthetic outputs)
          --[postprocess_detections]--> dets [B,128,6]                                                                  result_utils.py:76-168
    """

    def __init__(self, batch, device, channels=1024, n_cls=81, pre_nms_top_n=6000, post_nms_top_n=1000, rpn_nms_thresh=0.7,
                 pooled=7, sampling_ratio=0, max_det=100, max_out=128, im_h=synth.IM_H, im_w=synth.IM_W,
                 feat_dtype=torch.float32):
        self.B, self.dev, self.C, self.n_cls = batch, device, channels, n_cls
        self.pre, self.post, self.top_n = pre_nms_top_n, post_nms_top_n, post_nms_top_n
        self.thresh, self.pooled, self.sr = rpn_nms_thresh, pooled, sampling_ratio
        self.max_det, self.max_out, self.im_h, self.im_w = max_det, max_out, im_h, im_w
        self.H, self.W = synth.c4_shape(im_h, im_w)
        self.anchors = [generate_anchors(stride=16.0)]                                   # 15 anchors, detector.py:197-199
        self.feat_dtype = feat_dtype
        self.graph = None
        B, dev, f32, i32 = batch, device, torch.float32, torch.int32
        L = hip.lib()
        e = lambda *shape, dtype=f32: torch.empty(shape, dtype=dtype, device=dev)
        N = 15 * self.H * self.W
        self.kmax = min(self.pre, N)
        self.P = min(self.post, self.kmax)
        self.pre_boxes, self.pre_scores, self.pre_counts = e(B, self.kmax, 4), e(B, self.kmax), e(B, dtype=i32)
        self.keep, self.keep_cnt = e(B, self.P, dtype=i32), e(B, dtype=i32)
        self.prop_boxes, self.prop_scores = torch.zeros((B, self.P, 4), device=dev), torch.zeros((B, self.P), device=dev)
        self.nms_ws = hip.workspace(L.dtc_nms_sorted_workspace_bytes(B, self.kmax), dev)
        T = self.top_n
        self.rois5, self.roi_scores = e(B, T, 5), e(B, T)
        self.roi_levels, self.n_rois = e(B, T, dtype=i32), e(B, dtype=i32)
        self.rois_by_level, self.level_counts, self.idx_restore = e(B, T, 4), e(B, 1, dtype=i32), e(B, T, dtype=i32)
        self.roi_order, self.roi_desc = e(B, T, dtype=i32), e(B, T, 8)
        self.box_feats = e(B * T, self.C, pooled, pooled, dtype=feat_dtype)
        self.ra_ws = hip.workspace(L.dtc_roi_align_workspace_bytes(B * T), dev)
        D = max_out
        self.dets, self.det_roi = torch.zeros((B, D, 6), device=dev), torch.zeros((B, D), dtype=i32, device=dev)
        self.det_scaled, self.det_count = torch.zeros((B, D, 4), device=dev), e(B, dtype=i32)
        self.det_ws = hip.workspace(L.dtc_postprocess_detections_workspace_bytes(B, T, n_cls), dev)

    def bind(self, rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, scaling_factor, im_size):
        self.rpn_cls, self.rpn_bbox, self.feat = rpn_cls, rpn_bbox, feat
        self.cls_score, self.bbox_pred, self.sf, self.im_size = cls_score, bbox_pred, scaling_factor, im_size
        self.rpn_lv, self._alive = hip.make_rpn_levels([rpn_cls], [rpn_bbox], self.anchors, [16.0], [self.pre])
        self.rpn_ws = hip.workspace(hip.lib().dtc_rpn_topk_decode_workspace_bytes(self.rpn_lv, 1, self.B, self.kmax), self.dev)
        self.feat_lv, _, _ = hip.make_levels([feat], [1.0 / 16.0])
        self.feat_code, self.out_code = hip._dtype_code(feat.dtype), hip._dtype_code(self.feat_dtype)
        self.graph = None

    def _launch(self):
        L, B, st, ck = hip.lib(), self.B, hip.stream_ptr(self.dev), hip.check
        T, D = self.top_n, self.max_out
        ck(L.dtc_rpn_topk_decode(self.rpn_lv, 1, B, float(self.im_h), float(self.im_w), 0.0, self.rpn_ws.data_ptr(),
                                 self.rpn_ws.numel(), self.pre_boxes.data_ptr(), self.pre_scores.data_ptr(),
                                 self.pre_counts.data_ptr(), self.kmax, st), "rpn_topk_decode")
        ck(L.dtc_nms_sorted(self.pre_boxes.data_ptr(), self.pre_counts.data_ptr(), B, self.kmax, self.thresh, self.P,
                            self.nms_ws.data_ptr(), self.nms_ws.numel(), self.keep.data_ptr(), self.P,
                            self.keep_cnt.data_ptr(), st), "nms_sorted")
        ck(L.dtc_gather_kept(self.pre_boxes.data_ptr(), self.pre_scores.data_ptr(), B, self.kmax, self.keep.data_ptr(),
                             self.keep_cnt.data_ptr(), self.P, self.prop_boxes.data_ptr(), self.prop_scores.data_ptr(), st),
           "gather_kept")
        # one input list per image, already in score order: rois5 (b, box) + the RoIAlign visiting order; k_min == k_max -> level 0
        ck(L.dtc_fpn_collect_distribute(self.prop_boxes.data_ptr(), self.prop_scores.data_ptr(), self.keep_cnt.data_ptr(),
                                        B, 1, self.P, T, 4, 4, self.rois5.data_ptr(), self.roi_scores.data_ptr(),
                                        self.roi_levels.data_ptr(), self.n_rois.data_ptr(), self.rois_by_level.data_ptr(),
                                        self.level_counts.data_ptr(), self.idx_restore.data_ptr(),
                                        self.roi_order.data_ptr(), self.roi_desc.data_ptr(), 1, st), "collect")
        self._roi_align_box(st)
        ck(L.dtc_postprocess_detections(self.rois5.data_ptr(), self.n_rois.data_ptr(), self.cls_score.data_ptr(),
                                        self.bbox_pred.data_ptr(), self.sf.data_ptr(), self.im_size.data_ptr(), B, T,
                                        self.n_cls, 10.0, 10.0, 5.0, 5.0, 0.05, 0.5, self.max_det,
                                        self.det_ws.data_ptr(), self.det_ws.numel(), self.dets.data_ptr(),
                                        self.det_roi.data_ptr(), self.det_scaled.data_ptr(), self.det_count.data_ptr(), D, st),
           "postprocess_detections")

    def _roi_align_box(self, st=None):
        st = st or hip.stream_ptr(self.dev)
        # (workspace: the per-RoI records the map-stationary kernel's preparation pass writes, csrc/roi_align_map.hip)
        hip.check(hip.lib().dtc_roi_align_forward_packed_ws(self.feat_lv, 1, self.C, self.feat_code, self.roi_desc.data_ptr(),
                                                     self.B * self.top_n, self.pooled, self.pooled, self.sr,
                                                     self.box_feats.data_ptr(), self.out_code, self.ra_ws.data_ptr(),
                                                     self.ra_ws.numel(), st), "roi_align(c4)")

    step = FpnRegionPath.step

    def box_roialign_bytes(self):
        return (self.feat.numel() * self.feat.element_size() + self.B * self.top_n * 5 * 4 +
                self.box_feats.numel() * self.box_feats.element_size())


def synthetic_c4_batch(batch, device, seed, channels=1024, n_cls=81, top_n=1000, feat_dtype=torch.float32):
    """SURVEY.md section 8(d) cfg2 inputs on the device: argument tuple of C4RegionPath.bind()."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    H, W = synth.c4_shape()
    rn = lambda *s: torch.randn(s, generator=g, device=device)
    rpn_cls = torch.sigmoid(rn(batch, 15, H, W) * 2.0 - 2.0)
    rpn_bbox = rn(batch, 60, H, W) * 0.2
    feat = torch.relu(rn(batch, channels, H, W)).to(feat_dtype)
    cls_score = torch.softmax(rn(batch, top_n, n_cls) * 2.0, dim=2).contiguous()
    bbox_pred = (rn(batch, top_n, 4 * n_cls) * 0.1).contiguous()
    sf = torch.full((batch,), 1.6, device=device)
    im_size = torch.tensor([[500.0, 833.0]] * batch, device=device)
    return rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, sf, im_size


class OverlappedRegionPath:
    """The same hot path with the batch split into `n_split` sub-batches that run on separate HIP streams inside ONE
    hipGraph (fork/join).  The path alternates chip-filling RoIAlign launches with short latency-bound kernels (radix
    select, NMS reduce, collect, detection finalize) that occupy a handful of CUs; with two sub-batches in flight the
    small kernels of one overlap the RoIAlign of the other instead of serialising behind it."""

    def __init__(self, batch, device, n_split=2, **kw):
        assert batch % n_split == 0
        self.B, self.dev, self.n = batch, device, n_split
        self.sub = [FpnRegionPath(batch // n_split, device, **kw) for _ in range(n_split)]
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_split)]
        self.graph = None
        p0 = self.sub[0]
        self.max_out, self.top_n, self.pad_h, self.pad_w = p0.max_out, p0.top_n, p0.pad_h, p0.pad_w
        self.box_p, self.mask_p, self.C, self.feat_dtype = p0.box_p, p0.mask_p, p0.C, p0.feat_dtype

    def bind(self, rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, scaling_factor, im_size):
        k = self.B // self.n
        for i, p in enumerate(self.sub):
            sl = slice(i * k, (i + 1) * k)
            p.bind([t[sl] for t in rpn_cls], [t[sl] for t in rpn_bbox], [t[sl] for t in feats], cls_score[sl], bbox_pred[sl],
                   masks[i * k * p.max_out:(i + 1) * k * p.max_out], scaling_factor[sl], im_size[sl])
        self.graph = None

    def _launch(self):
        cur = torch.cuda.current_stream(self.dev)
        for st, p in zip(self.streams, self.sub):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                p._launch()
        for st in self.streams:
            cur.wait_stream(st)

    def step(self, use_graph=True):
        with torch.cuda.device(self.dev):
            if not use_graph:
                self._launch()
                return
            if self.graph is None:
                self._launch()
                torch.cuda.synchronize(self.dev)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._launch()
                self.graph = g
            self.graph.replay()

    @property
    def dets(self):
        return torch.cat([p.dets for p in self.sub], 0)

    @property
    def box_feats(self):
        return torch.cat([p.box_feats for p in self.sub], 0)

    @property
    def det_count(self):
        return torch.cat([p.det_count for p in self.sub], 0)

    def _roi_align_box(self):
        for p in self.sub:
            p._roi_align_box()

    def box_roialign_bytes(self):
        return sum(p.box_roialign_bytes() for p in self.sub)

    def results(self):
        out = []
        for p in self.sub:
            out += p.results()
        return out


class StepPipeline:
    """Consecutive steps in flight.  A step alternates two chip-filling RoIAlign launches (65 % of its time) with a dozen
    latency-bound kernels that occupy a handful of CUs (radix select, NMS reduce, collect, detection finalize, paste); the
    steps of a serving loop are independent batches, so step k+1 is issued on another HIP stream while step k is still
    running and the small kernels of one fill the gaps of the other (one MI355X, batch 8: 0.78 -> 0.60 ms per step).
    `paths` are bound region paths (FpnRegionPath / C4RegionPath), at least `n_inflight` of them: a path owns its workspaces
    and results, so a path is only ever replayed on ITS stream and steps on the same path serialise.  Sub-batches of one
    step on two streams (OverlappedRegionPath) do not give this: they run in phase, RoIAlign against RoIAlign."""

    def __init__(self, paths, device, n_inflight=2):
        assert len(paths) >= n_inflight >= 1
        self.paths, self.dev, self.n = list(paths), device, n_inflight
        self.streams = [torch.cuda.Stream(device=device) for _ in range(n_inflight)] if n_inflight > 1 else [None]
        self.count = 0

    def step(self, use_graph=True):
        """Issue the next step; returns (path, stream) so that the caller can enqueue work behind it on the same stream."""
        i = self.count
        self.count += 1
        p = self.paths[i % len(self.paths)]
        st = self.streams[(i % len(self.paths)) % self.n]        # path j always runs on stream j % n
        if st is None:
            p.step(use_graph=use_graph)
        else:
            # whatever the caller enqueued on ITS stream before this call (binding inputs at first, refilling the bound input
            # tensors between steps later) is ordered before the step; the converse -- do not overwrite the inputs of a step
            # that is still running -- is the caller's: enqueue the refill on the returned stream, or synchronize() first
            st.wait_stream(torch.cuda.current_stream(self.dev))
            with torch.cuda.stream(st):
                p.step(use_graph=use_graph)
        return p, st

    def synchronize(self):
        for st in self.streams:
            if st is not None:
                st.synchronize()


def synthetic_batch(batch, device, seed, channels=256, n_cls=81, top_n=1000, max_out=128, mask_res=28,
                    feat_dtype=torch.float32, channels_last=False):
    """COCO-shaped synthetic inputs of SURVEY.md section 8(d), generated on the device (random-init: there are no
    Detectron weights / COCO images offline).  Returns the argument tuple of FpnRegionPath.bind()."""
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    shapes = synth.fpn_level_shapes()
    rn = lambda *s: torch.randn(s, generator=g, device=device)
    rpn_cls = [torch.sigmoid(rn(batch, 3, h, w) * 2.0 - 2.0) for (h, w) in shapes]
    rpn_bbox = [rn(batch, 12, h, w) * 0.2 for (h, w) in shapes]
    feats = [torch.relu(rn(batch, channels, h, w)).to(feat_dtype) for (h, w) in shapes[:4]]
    if channels_last:
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    cls_score = torch.softmax(rn(batch, top_n, n_cls) * 2.0, dim=2).contiguous()
    bbox_pred = (rn(batch, top_n, 4 * n_cls) * 0.1).contiguous()
    masks = torch.sigmoid(rn(batch * max_out, n_cls, mask_res, mask_res) * 1.5)
    sf = torch.full((batch,), 1.6, device=device)
    im_size = torch.tensor([[500.0, 833.0]] * batch, device=device)
    return rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size
