"""Where the hot path sits inside the whole model (context for DESIGN.md; bench.py is the contract benchmark).

Times, with HIP events on one MI355X, a Mask R-CNN R-50-FPN `detector` (random weights, eval mode) on a 1x3x800x1344 input:
  backbone+FPN (MIOpen convs, NOT ours) | whole forward = backbone + RPN heads + [GenerateProposals, NMS, collect/distribute,
  RoIAlign]* + box head GEMMs + softmax | postprocess_output* | mask_head = [RoIAlign 14x14]* + 4 convs + deconv | segm_results*
(* = this repository's kernels).  Usage: python tools/bench_detector.py [--channels-last] [--iters 10]
  --batched [--batch 8] [--dtype fp32|bf16|fp16] [--optimize]: detector.forward_batched on a batch (one launch chain, one sync),
  under autocast or -- --optimize -- in the inference form of detector.optimize_for_inference (BatchNorm folded, fused epilogues).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    h0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    timed.host_ms = (time.perf_counter() - h0) * 1e3 / iters        # host time to ISSUE one call (no synchronisation inside)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--batched", action="store_true", help="time detector.forward_batched (backbone -> fused region path -> heads -> detections -> masks, one launch chain, no host round trip) instead of the reference-shaped per-image flow")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", choices=["fp32", "bf16", "fp16"], default="fp32", help="--batched: backbone_dtype = head_dtype")
    ap.add_argument("--miopen-benchmark", action="store_true", help="torch.backends.cudnn.benchmark = True: MIOpen times its solvers per convolution shape on first use (slow first call)")
    ap.add_argument("--optimize", action="store_true", help="--batched: detector.optimize_for_inference(dtype) -- BatchNorm folded, fused epilogues, weights stored in the compute type -- instead of autocast")
    a = ap.parse_args()
    if a.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    from detectorch_amd.model.detector import detector
    from detectorch_amd.utils import result_utils
    from detectorch_amd.utils.multilevel_rois import add_multilevel_rois_for_test
    torch.manual_seed(0)
    m = detector(arch='resnet50', conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                 conv_head_layers='two_layer_mlp', fpn_layers=['layer1', 'layer2', 'layer3', 'layer4'], fpn_extra_lvl=True,
                 roi_height=7, roi_width=7, roi_spatial_scale=[0.25, 0.125, 0.0625, 0.03125], roi_sampling_ratio=2,
                 use_rpn_head=True, use_mask_head=True, mask_head_type='1up4convs', channels_last=a.channels_last).cuda()
    if a.channels_last:
        m = m.to(memory_format=torch.channels_last)
    if a.batched:
        dt = {"fp32": None, "bf16": torch.bfloat16, "fp16": torch.float16}[a.dtype]
        if a.optimize:
            m.optimize_for_inference(dt)
        else:
            m.backbone_dtype = m.head_dtype = dt
        m.classif_head.weight.data *= 60.0                                  # random weights: make some detections exist
        images = torch.randn(a.batch, 3, 800, 1344, device="cuda")
        sfb = torch.full((a.batch,), 1.6, device="cuda")
        szb = torch.tensor([[500.0, 833.0]] * a.batch, device="cuda")
        with torch.no_grad():
            low = torch.autocast("cuda", dtype=dt) if dt is not None else None
            def body():
                if a.optimize:
                    x = images.to(dt) if dt is not None else images
                    return m.conv_body(x.contiguous(memory_format=torch.channels_last) if a.channels_last else x)
                if low is None:
                    return m.conv_body(images)
                with torch.autocast("cuda", dtype=dt):
                    return m.conv_body(images)
            t_body, _ = timed(body, a.iters)
            t_all, path = timed(lambda: m.forward_batched(images, sfb, szb), a.iters)
        print({"mode": "forward_batched" + (" optimized" if a.optimize else ""), "batch": a.batch, "dtype": a.dtype, "layout": "NHWC" if a.channels_last else "NCHW",
               "backbone_fpn_ms": round(t_body, 3), "forward_batched_ms": round(t_all, 3),
               "host_issue_ms": round(timed.host_ms, 3),
               "ms_per_image": round(t_all / a.batch, 3), "images_per_sec": round(a.batch / t_all * 1e3, 1),
               "not_backbone_ms_per_image": round((t_all - t_body) / a.batch, 3),
               "detections": path.det_count.tolist()})
        return
    image = torch.randn(1, 3, 800, 1344, device="cuda")
    sf = torch.tensor([1.6], device="cuda")
    im_size = torch.tensor([500.0, 833.0, 3.0])
    with torch.no_grad():
        t_body, feats = timed(lambda: m.conv_body(image.contiguous(memory_format=torch.channels_last) if a.channels_last else image), a.iters)
        t_fwd, (cls_score, bbox_pred, rois, feats) = timed(lambda: m(image, scaling_factor=sf), a.iters)
        boosted = torch.softmax(torch.log(cls_score) * 40.0, dim=1)          # random weights: make some detections exist
        t_post, (scores_final, boxes_final, cls_boxes) = timed(
            lambda: result_utils.postprocess_output(rois, sf, im_size, boosted, bbox_pred), a.iters)
        out = {"layout": "NHWC" if a.channels_last else "NCHW", "backbone_fpn_ms": round(t_body, 3), "forward_ms": round(t_fwd, 3),
               "forward_minus_backbone_ms": round(t_fwd - t_body, 3), "postprocess_output_ms": round(t_post, 3),
               "rois": int(rois.shape[0]), "detections": int(boxes_final.shape[0])}
        if boxes_final.shape[0]:
            blobs = add_multilevel_rois_for_test({'rois': boxes_final * 1.6}, 'rois')
            per_level = [torch.from_numpy(blobs[k]).cuda() if len(blobs[k]) else None
                         for k in ['rois_fpn2', 'rois_fpn3', 'rois_fpn4', 'rois_fpn5']]
            restore = torch.from_numpy(blobs['rois_idx_restore_int32']).cuda().long()
            t_mask, masks = timed(lambda: m.mask_head(feats, per_level, restore), a.iters)
            t_segm, _ = timed(lambda: result_utils.segm_results(cls_boxes, masks, boxes_final, 500, 833, M=28), a.iters)
            out["mask_head_ms"], out["segm_results_ms"] = round(t_mask, 3), round(t_segm, 3)
    print(out)


if __name__ == "__main__":
    main()
