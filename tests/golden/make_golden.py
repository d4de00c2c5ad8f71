"""Generate tests/golden/*.npz from the REFERENCE ITSELF (build container only; needs /root/reference + `make -C oracle ref`).

    python tests/golden/make_golden.py

Every array stored here is either a seeded synthetic input or the output of the reference's own code on it:
  * lib/cppcuda_cffi/src/cpp/roi_align_cpu_loop.cpp (compiled unmodified into oracle/_ref/libref_roialign.so)
  * lib/utils_cython/cython_nms.pyx                 (built into oracle/_ref/utils_cython)
  * lib/model/generate_proposals.py, lib/model/collect_and_distribute_fpn_rpn_proposals.py,
    lib/utils/{boxes,generate_anchors,multilevel_rois,result_utils}.py imported in place on CPU tensors.
The fixtures are small (KBs) so they can be committed; they are what pins oracle/oracle.c (tests/test_oracle_golden.py)
and what the HIP path is compared with on the GPU box, where /root/reference does not exist.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_harness as rh  # noqa: E402

sys.path.insert(0, ROOT)
from detectorch_amd import synth  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("%-28s %7.1f KB  %s" % (name, os.path.getsize(path) / 1024.0, sorted(arrs)))


def bbox_vote(ns):
    """(f)-4: cython_bbox.bbox_overlaps (cython_bbox.pyx:32) and boxes.box_voting (boxes.py:280) run by the reference."""
    rs = synth.rng(7, 0)
    n = 600
    centres = synth.make_rois(rs, 12, im_h=500, im_w=833, min_side=20, max_side=300)
    a = centres[rs.randint(0, 12, n)] + rs.standard_normal((n, 4)).astype(np.float32) * 4
    a[:, 2:] = np.maximum(a[:, 2:], a[:, :2] + 1)
    all_dets = np.ascontiguousarray(np.hstack([a, rs.uniform(0.05, 1, (n, 1))]), np.float32)
    top = np.ascontiguousarray(all_dets[rs.choice(n, 40, replace=False)])
    q = synth.make_rois(rs, 37, im_h=500, im_w=833, min_side=4, max_side=500)
    arrs = dict(all_dets=all_dets, top_dets=top, query=q,
                overlaps=ns.cython_bbox.bbox_overlaps(np.ascontiguousarray(all_dets[:, :4]), q),
                overlaps_top=ns.cython_bbox.bbox_overlaps(np.ascontiguousarray(top[:, :4]), np.ascontiguousarray(all_dets[:, :4])))
    for m in ('ID', 'TEMP_AVG', 'AVG', 'IOU_AVG', 'GENERALIZED_AVG', 'QUASI_SUM'):
        for beta in (1.0, 0.5):
            arrs["vote_%s_b%d" % (m, int(beta * 10))] = ns.boxes.box_voting(top, all_dets, 0.6, scoring_method=m, beta=beta)
    # the voting branch of box_results_with_nms_and_limit on the postprocess fixture's inputs
    g = np.load(os.path.join(HERE, "postprocess.npz"))
    sc, bx, cb = ns.result_utils.box_results_with_nms_and_limit(g["cls"], g["pred_clipped"].copy(), do_bbox_vote=True,
                                                                bbox_vote_thresh=0.8)
    arrs["pp_vote_scores"], arrs["pp_vote_boxes"] = sc, bx
    arrs["pp_vote_cls_id"] = np.concatenate([np.full(len(cb[j]), j, np.int32) for j in range(1, 81)])
    save("bbox_vote", **arrs)


def postprocess_logits(ns):
    """the box head's softmax + postprocess_output.  Logits -> torch CPU F.softmax (what lib/model/detector.py:281
    runs) -> the reference's postprocess_output, imported in place.  The fused HIP path gets the LOGITS."""
    import torch.nn.functional as F
    rs = synth.rng(7, 0)
    R = 300
    rois = synth.make_rois(rs, R)
    logits = (rs.standard_normal((R, 81)) * 3.0).astype(np.float32)
    logits[:, 0] -= 1.0
    deltas = (rs.standard_normal((R, 324)) * 0.1).astype(np.float32)
    im_size = np.array([500.0, 833.0, 3.0], np.float32)
    sf = np.float32(1.6)
    prob = F.softmax(torch.from_numpy(logits), dim=1)
    scores_final, boxes_final, cls_boxes = ns.result_utils.postprocess_output(
        torch.from_numpy(rois), float(sf), torch.from_numpy(im_size), prob, torch.from_numpy(deltas))
    cls_id = np.concatenate([np.full(len(cls_boxes[j]), j, np.int32) for j in range(1, 81)])
    save("postprocess_logits", rois=rois, logits=logits, deltas=deltas, im_size=im_size, sf=np.array([sf], np.float32),
         prob_torch=prob.numpy(), scores_final=scores_final, boxes_final=boxes_final, cls_id=cls_id)


def main():
    ns = rh.load_reference()
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "bbox_vote":     # add one fixture without rewriting the others
        return bbox_vote(ns)
    if len(sys.argv) > 1 and sys.argv[1] == "postprocess_logits":
        return postprocess_logits(ns)

    # ---- A2 anchors (generate_anchors.py:54) -------------------------------------------------------------------
    arrs = {}
    cases = [(16, (32, 64, 128, 256, 512), (0.5, 1, 2)), (16, (128, 256, 512), (0.5, 1, 2))]
    cases += [(float(s), (32.0 * 2 ** i,), (0.5, 1, 2)) for i, s in enumerate((4, 8, 16, 32, 64))]
    for k, (stride, sizes, ratios) in enumerate(cases):
        arrs["in%d" % k] = np.array([stride] + list(sizes) + [-1] + list(ratios), np.float64)
        arrs["out%d" % k] = ns.generate_anchors.generate_anchors(stride=stride, sizes=sizes, aspect_ratios=ratios)
    save("anchors", **arrs)

    # ---- A1 RoIAlign (roi_align_cpu_loop.cpp:118) ---------------------------------------------------------------
    rs = synth.rng(1, 0)
    feat = rs.standard_normal((2, 6, 20, 30)).astype(np.float32)
    rois = synth.make_rois(rs, 24, im_h=320, im_w=480, min_side=8, max_side=400)
    b = rs.randint(0, 2, (24, 1)).astype(np.float32)
    rois5 = np.hstack([b, rois])
    # edge cases the CPU path special-cases: outside the map, degenerate (x2<x1), exactly on the border, sub-pixel
    edge = np.array([[0, -50, -50, -20, -20], [1, 470, 310, 479, 319], [0, 100, 100, 90, 95], [1, 0, 0, 479, 319],
                     [0, 15.5, 15.5, 16.0, 16.0], [1, 600, 400, 700, 500], [0, -8, -8, 8, 8], [1, 472, 300, 500, 330]],
                    np.float32)
    rois5 = np.vstack([rois5, edge])
    arrs = {"features": feat, "rois5": rois5}
    for tag, (ph, pw, sr, scale) in {"p7s2": (7, 7, 2, 1 / 16.), "p14s0": (14, 14, 0, 1 / 16.), "p7s0": (7, 7, 0, 0.25),
                                     "p14s2": (14, 14, 2, 0.125), "p3x5s3": (3, 5, 3, 1 / 16.)}.items():
        arrs["cfg_" + tag] = np.array([ph, pw, sr, scale], np.float64)
        arrs["out_" + tag] = rh.ref_roi_align(feat, rois5, ph, pw, scale, sr)
    arrs["out4col_p7s2"] = rh.ref_roi_align(feat[:1], rois5[:, 1:], 7, 7, 1 / 16., 2)   # 4-column rois: batch 0
    save("roi_align", **arrs)

    # ---- A5 / A6 NMS, Soft-NMS (cython_nms.pyx:37,98) -----------------------------------------------------------
    rs = synth.rng(2, 0)
    boxes = synth.make_rois(rs, 400, im_h=300, im_w=400, min_side=10, max_side=200)
    sc = synth.dedupe_scores(rs.uniform(0.01, 1.0, 400).astype(np.float32))
    dets = np.hstack([boxes, sc[:, None]]).astype(np.float32)
    arrs = {"dets": dets}
    for t in (0.3, 0.5, 0.7):
        arrs["keep_%02d" % int(t * 10)] = np.asarray(ns.boxes.nms(dets, t), np.int64)
    for m in ("hard", "linear", "gaussian"):
        d, k = ns.boxes.soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method=m)
        arrs["soft_%s_dets" % m] = d
        arrs["soft_%s_keep" % m] = np.asarray(k, np.int64)
    d, k = ns.boxes.soft_nms(dets, sigma=0.5, overlap_thresh=0.5, score_thresh=0.0001, method="linear")  # result_utils.py:133-139
    arrs["soft_linear05_dets"], arrs["soft_linear05_keep"] = d, np.asarray(k, np.int64)
    save("nms", **arrs)

    # ---- A2-A5 GenerateProposals.forward (generate_proposals.py:31) ---------------------------------------------
    arrs = {}
    gp_cases = {
        # tag: (A, H, W, spatial_scale, anchor_sizes, pre, post, im_h, im_w)
        "c4": (15, 12, 20, 1 / 16., (32, 64, 128, 256, 512), 600, 100, 192, 320),
        "p3": (3, 25, 42, 1 / 8., (64,), 1000, 1000, 200, 336),
        "p6": (3, 13, 21, 1 / 64., (512,), 1000, 1000, 800, 1333),       # fewer anchors than pre_nms_top_n
    }
    for k, (tag, (A, H, W, ss, sizes, pre, post, im_h, im_w)) in enumerate(gp_cases.items()):
        rs = synth.rng(3, k)
        p, d = synth.make_rpn_outputs(rs, A, H, W)
        gp = ns.generate_proposals.GenerateProposals(spatial_scale=ss, anchor_sizes=sizes, rpn_pre_nms_top_n=pre,
                                                     rpn_post_nms_top_n=post)
        props, scores = gp(torch.from_numpy(p), torch.from_numpy(d), im_h, im_w, 1.6)
        arrs[tag + "_cfg"] = np.array([A, H, W, ss, pre, post, im_h, im_w, 0.7] + list(sizes), np.float64)
        arrs[tag + "_cls"], arrs[tag + "_bbox"] = p, d
        arrs[tag + "_props"], arrs[tag + "_scores"] = props.numpy().copy(), scores.numpy().copy()
    save("generate_proposals", **arrs)

    # ---- A7 collect + distribute (collect_and_distribute_fpn_rpn_proposals.py:84,108; multilevel_rois.py:41) -----
    rs = synth.rng(4, 0)
    roi_list, score_list = [], []
    for lvl in range(5):
        n = [300, 280, 200, 120, 40][lvl]
        roi_list.append(synth.make_rois(rs, n))
        score_list.append(rs.uniform(0, 1, (n, 1)).astype(np.float32))
    allsc = synth.dedupe_scores(np.concatenate(score_list))
    p = 0
    for lvl in range(5):
        n = score_list[lvl].shape[0]
        score_list[lvl] = allsc[p:p + n].copy()
        p += n
    cd = ns.collect.CollectAndDistributeFpnRpnProposals(spatial_scales=list(synth.FPN_ROI_SCALES))
    # the reference hard-codes post_nms_topN=1000 (:86); 940 inputs -> exercises n < topN. Use a second, larger case too.
    distr, restore = cd([torch.from_numpy(r) for r in roi_list], [torch.from_numpy(s) for s in score_list])
    arrs = {"restore": np.asarray(restore, np.int64)}
    for lvl in range(5):
        arrs["rois%d" % lvl], arrs["scores%d" % lvl] = roi_list[lvl], score_list[lvl]
    for i, r in enumerate(distr):
        arrs["distr%d" % i] = r.numpy().copy()
    big_r = [synth.make_rois(rs, 600) for _ in range(5)]
    big_s = synth.dedupe_scores(rs.uniform(0, 1, (3000, 1)).astype(np.float32))
    big_s = [big_s[i * 600:(i + 1) * 600].copy() for i in range(5)]
    distr, restore = cd([torch.from_numpy(r) for r in big_r], [torch.from_numpy(s) for s in big_s])
    arrs["big_restore"] = np.asarray(restore, np.int64)
    for lvl in range(5):
        arrs["big_rois%d" % lvl], arrs["big_scores%d" % lvl] = big_r[lvl], big_s[lvl]
    for i, r in enumerate(distr):
        arrs["big_distr%d" % i] = r.numpy().copy()
    # level mapping on boundary-dense boxes: sqrt(area) right at 224 * 2^k +- a few ulps
    bl = []
    for s0 in (56.0, 112.0, 224.0, 448.0, 896.0):
        for off in np.linspace(-0.01, 0.01, 41):
            side = s0 + off
            bl.append([10.0, 20.0, 10.0 + side - 1.0, 20.0 + side - 1.0])
            bl.append([3.0, 7.0, 3.0 + side * 2 - 1.0, 7.0 + side / 2 - 1.0])
    bl = np.asarray(bl, np.float32)
    arrs["lvl_boxes"] = bl
    arrs["lvl_out"] = ns.multilevel_rois.map_rois_to_fpn_levels(bl, 2, 5).astype(np.int32)
    save("collect_distribute", **arrs)

    # ---- A4 numpy decode + A8 postprocess (boxes.py:168,150; result_utils.py:76,96) -----------------------------
    rs = synth.rng(5, 0)
    R = 160
    rois = synth.make_rois(rs, R)
    cls, deltas = synth.make_head_outputs(rs, R)
    # concentrate mass so that >100 detections survive and the max_detections_per_img branch (:154-163) runs
    cls = synth.dedupe_scores(np.clip(cls * 3.0, 0, 0.999).astype(np.float32))
    im_size = np.array([500.0, 833.0, 3.0], np.float32)
    sf = np.float32(1.6)
    boxes = (torch.from_numpy(rois) / float(sf)).numpy()
    pred = ns.boxes.bbox_transform(boxes, deltas, (10.0, 10.0, 5.0, 5.0))
    pred_c = ns.boxes.clip_tiled_boxes(pred.copy(), im_size)
    scores_final, boxes_final, cls_boxes = ns.result_utils.postprocess_output(
        torch.from_numpy(rois), float(sf), torch.from_numpy(im_size), torch.from_numpy(cls), torch.from_numpy(deltas))
    cls_id = np.concatenate([np.full(len(cls_boxes[j]), j, np.int32) for j in range(1, 81)])
    arrs = dict(rois=rois, cls=cls, deltas=deltas, im_size=im_size, sf=np.array([sf], np.float32), pred=pred,
                pred_clipped=pred_c, scores_final=scores_final, boxes_final=boxes_final, cls_id=cls_id)
    # unlimited variant (max_detections_per_img=0) for the per-class NMS alone
    sc2, bx2, cb2 = ns.result_utils.box_results_with_nms_and_limit(cls, pred_c.copy(), max_detections_per_img=0)
    arrs["nolimit_scores"], arrs["nolimit_boxes"] = sc2, bx2
    arrs["nolimit_cls_id"] = np.concatenate([np.full(len(cb2[j]), j, np.int32) for j in range(1, 81)])
    save("postprocess", **arrs)

    # ---- A9 geometry only (expand_boxes + int32 truncation, boxes.py:245; result_utils.py:182-184) --------------
    rs = synth.rng(6, 0)
    ref_boxes = synth.make_rois(rs, 64, im_h=500, im_w=833, min_side=4, max_side=500)
    arrs = {"ref_boxes": ref_boxes}
    for M in (14, 28):
        arrs["exp_int_M%d" % M] = ns.boxes.expand_boxes(ref_boxes, (M + 2.0) / M).astype(np.int32)
    save("mask_geometry", **arrs)
    bbox_vote(ns)
    postprocess_logits(ns)


if __name__ == "__main__":
    main()
