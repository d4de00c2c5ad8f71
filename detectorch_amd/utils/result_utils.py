"""Detection post-processing -- same names / arguments / return layout as the reference's lib/utils/result_utils.py, with
the arithmetic done by the HIP kernels (detectorch_amd/csrc/detections.hip, nms.hip, mask_paste.hip).

    postprocess_output              result_utils.py:76-94
    box_results_with_nms_and_limit  result_utils.py:96-168   (hard NMS, Soft-NMS, optional bbox voting)
    segm_results                    result_utils.py:170-228  (RLE on the device: dtc_mask_rle)
    empty_results / extend_results  result_utils.py:32-60
    assemble_results                the eval loop's per-image empty_results / extend_results bookkeeping (eval_mask_FPN.ipynb
                                    cell 6) for a whole (gathered) batch of fixed-shape device results
    coco_bbox_results / coco_segm_results   the records of lib/utils/json_dataset_evaluator.py:67-113, 149-190
"""
import numpy as np
import torch

from .. import hip
from . import boxes as box_utils


def to_np(x):
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()


def empty_results(num_classes, num_images):
    all_boxes = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_segms = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    all_keyps = [[[] for _ in range(num_images)] for _ in range(num_classes)]
    return all_boxes, all_segms, all_keyps


def extend_results(index, all_res, im_res):
    for cls_idx in range(1, len(im_res)):
        all_res[cls_idx][index] = im_res[cls_idx]


def _dev(*xs):
    for x in xs:
        if torch.is_tensor(x) and x.is_cuda:
            return x.device
    if not torch.cuda.is_available():
        raise RuntimeError("detectorch_amd needs the MI355X HIP path (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _t(x, dev):
    if torch.is_tensor(x):
        return x.detach().to(device=dev, dtype=torch.float32)
    return torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev)


def _split_by_class(dets, n, num_classes):
    dets = dets[:n]
    cls = dets[:, 5].astype(np.int64)
    cls_boxes = [[] for _ in range(num_classes)]
    for j in range(1, num_classes):
        cls_boxes[j] = np.ascontiguousarray(dets[cls == j, :5])
    return dets[:, 4].copy(), np.ascontiguousarray(dets[:, :4]), cls_boxes


def postprocess_output(rois, scaling_factor, im_size, class_scores, bbox_deltas, bbox_reg_weights=(10.0, 10.0, 5.0, 5.0),
                       class_scores_are_logits=False):
    """result_utils.py:76-94 -> (scores_final [D], boxes_final [D,4], boxes_per_class list[81] of [d_j,5]).
    class_scores_are_logits=True (not in the reference): `class_scores` is the raw output of the cls_score layer and the
    F.softmax of lib/model/detector.py:281 is folded into the detection kernel."""
    dev = _dev(rois, class_scores, bbox_deltas)
    rois = _t(rois, dev)
    if rois.dim() == 3:
        rois = rois.squeeze(0)
    R = rois.shape[0]
    cls = _t(class_scores, dev).reshape(1, R, -1)
    dl = _t(bbox_deltas, dev).reshape(1, R, -1)
    n_cls = cls.shape[2]
    rois5 = torch.cat([torch.zeros((R, 1), device=dev), rois[:, -4:]], 1).reshape(1, R, 5)
    sf = _t(scaling_factor, dev).reshape(-1)[:1]
    sz = _t(im_size, dev).reshape(-1)[:2].reshape(1, 2)
    dets, _, _, cnt = hip.postprocess_detections(rois5, None, cls, dl, sf, sz, weights=bbox_reg_weights,
                                                 max_out=max(R * (n_cls - 1), 1) if R * (n_cls - 1) <= 4096 else 4096,
                                                 scores_are_logits=class_scores_are_logits)
    n = int(cnt[0].item())
    if n > dets.shape[1]:
        dets, _, _, cnt = hip.postprocess_detections(rois5, None, cls, dl, sf, sz, weights=bbox_reg_weights, max_out=n,
                                                     scores_are_logits=class_scores_are_logits)
    return _split_by_class(dets[0].cpu().numpy(), n, n_cls)


def box_results_with_nms_and_limit(scores, boxes, num_classes=81, score_thresh=0.05, overlap_thresh=0.5,
                                   do_soft_nms=False, soft_nms_sigma=0.5, soft_nms_method='linear', do_bbox_vote=False,
                                   bbox_vote_thresh=0.8, bbox_vote_method='ID', max_detections_per_img=100):
    """result_utils.py:96-168 on already decoded+clipped boxes [R,4*num_classes] (numpy in / numpy out).
    The default branch (hard NMS, no bbox vote) is ONE pass on the device -- threshold, 80-class segmented NMS, the top-100
    limit, one copy back (dtc_box_results_nms_limit); the reference runs 80 Python iterations, each a host NMS call.
    Soft-NMS and bbox voting keep the reference's per-class structure on their own kernels."""
    scores = np.ascontiguousarray(scores, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32)
    if not do_soft_nms and not do_bbox_vote and scores.shape[0] <= 4096 and scores.shape[0] > 0 and num_classes > 1:
        dev = _dev()
        R = scores.shape[0]
        sc = torch.from_numpy(scores[:, :num_classes]).to(dev).reshape(1, R, num_classes)
        bx = torch.from_numpy(boxes[:, :4 * num_classes]).to(dev).reshape(1, R, 4 * num_classes)
        max_out = 256 if max_detections_per_img > 0 else R * (num_classes - 1)
        dets, _, cnt = hip.box_results_nms_limit(sc, bx, None, score_thresh, overlap_thresh, max_detections_per_img, max_out)
        n = int(cnt[0].item())
        if n > dets.shape[1]:            # more than max_out rows tie at the limit score (:161 keeps them all)
            dets, _, cnt = hip.box_results_nms_limit(sc, bx, None, score_thresh, overlap_thresh, max_detections_per_img, n)
        return _split_by_class(dets[0].cpu().numpy(), n, num_classes)
    cls_boxes = [[] for _ in range(num_classes)]
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > score_thresh)[0]
        dets_j = np.hstack((boxes[inds, j * 4:(j + 1) * 4], scores[inds, j][:, np.newaxis])).astype(np.float32, copy=False)
        if do_soft_nms:
            nms_dets, _ = box_utils.soft_nms(dets_j, sigma=soft_nms_sigma, overlap_thresh=overlap_thresh,
                                             score_thresh=0.0001, method=soft_nms_method)
        else:
            keep = box_utils.nms(dets_j, overlap_thresh)
            nms_dets = dets_j[keep, :]
        if do_bbox_vote:                                       # result_utils.py:147-153
            nms_dets = box_utils.box_voting(nms_dets, dets_j, bbox_vote_thresh, scoring_method=bbox_vote_method)
        cls_boxes[j] = nms_dets
    if max_detections_per_img > 0:
        image_scores = np.hstack([cls_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > max_detections_per_img:
            image_thresh = np.sort(image_scores)[-max_detections_per_img]
            for j in range(1, num_classes):
                keep = np.where(cls_boxes[j][:, -1] >= image_thresh)[0]
                cls_boxes[j] = cls_boxes[j][keep, :]
    im_results = np.vstack([cls_boxes[j] for j in range(1, num_classes)])
    return im_results[:, -1], im_results[:, :-1], cls_boxes


# ---- COCO RLE of a dense host mask (utility for callers that hold numpy masks; segm_results encodes on the device) ----
def _rle_counts_to_string(cnts):
    out = []
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return "".join(out)


def rle_encode(mask):
    """mask [h,w] uint8 -> {'size': [h,w], 'counts': str}: column-major run lengths starting with a run of zeros."""
    h, w = mask.shape
    flat = np.asarray(mask, np.uint8).reshape(-1, order='F')
    change = np.flatnonzero(np.diff(flat)) + 1
    runs = np.diff(np.concatenate([[0], change, [flat.size]]))
    if flat.size and flat[0] == 1:
        runs = np.concatenate([[0], runs])
    return {'size': [int(h), int(w)], 'counts': _rle_counts_to_string(runs.tolist())}


def segm_results(cls_boxes, masks, ref_boxes, im_h, im_w, num_classes=81, M=14, cls_specific_mask=True,
                 thresh_binarize=0.5):
    """result_utils.py:170-228.  masks may be a CUDA tensor (preferred: nothing but the crops leaves the GPU) or ndarray."""
    dev = _dev(masks)
    masks_t = masks if torch.is_tensor(masks) else torch.from_numpy(np.ascontiguousarray(masks, np.float32))
    masks_t = masks_t.to(device=dev, dtype=torch.float32)
    ref = np.ascontiguousarray(ref_boxes, np.float32)
    D = ref.shape[0]
    cls_of = np.concatenate([np.full(len(cls_boxes[j]), j, np.float32) for j in range(1, num_classes)]) if D else np.zeros(0, np.float32)
    assert cls_of.shape[0] == masks_t.shape[0] == D                  # :227
    cls_segms = [[] for _ in range(num_classes)]
    if D == 0:
        return cls_segms
    dets = np.zeros((1, D, 6), np.float32)
    dets[0, :, :4] = ref
    dets[0, :, 5] = cls_of
    cap = int(im_h) * int(im_w) * D
    out = hip.mask_paste(masks_t, torch.from_numpy(dets).to(dev), torch.tensor([D], dtype=torch.int32, device=dev),
                         torch.tensor([[float(im_h), float(im_w)]], device=dev), M, cap,
                         mask_index=torch.arange(D, dtype=torch.int32, device=dev).reshape(1, D), thresh=thresh_binarize,
                         cls_specific=cls_specific_mask)
    # COCO RLE on the device (dtc_mask_rle): only the count strings cross PCIe.  If a mask has more runs than the first
    # guess holds (pathological noise), the kernel reports the sizes it needs and is re-run once with those.
    dcount = torch.tensor([D], dtype=torch.int32, device=dev)
    imsz = torch.tensor([[float(im_h), float(im_w)]], device=dev)
    runs_stride, str_stride = 2 * int(im_w) + 8, 4 * int(im_w) + 64
    for _ in range(3):
        rle = hip.mask_rle(out, dcount, imsz, runs_stride=runs_stride, str_stride=str_stride)
        nrun = rle["n_runs"][0].cpu().numpy()
        slen = rle["str_len"][0].cpu().numpy()
        if nrun.min() >= 0 and slen.min() >= 0:
            break
        runs_stride = max(runs_stride, int(-nrun.min()) + 8)
        str_stride = max(str_stride, int(-slen.min()) + 8, 7 * runs_stride)
    else:
        raise RuntimeError("dtc_mask_rle: buffers still too small")
    sbuf = rle["str"][0, :, :max(int(slen.max()), 1)].cpu().numpy()
    for d in range(D):
        cls_segms[int(cls_of[d])].append({'size': [int(im_h), int(im_w)],
                                          'counts': sbuf[d, :slen[d]].tobytes().decode('ascii')})
    return cls_segms


def assemble_results(dets, det_count, im_sizes=None, rle_str=None, rle_len=None, num_classes=81, all_boxes=None,
                     all_segms=None, first_image=0, on_overflow="raise"):
    """all_boxes / all_segms (result_utils.py:32-60) from the fixed-shape outputs of the batched path.

      dets [N, max_out, 6] = (x1,y1,x2,y2,score,class) class-major per image, det_count [N]   (dtc_postprocess_detections,
      or detectorch_amd.dist.ResultGatherer.finish() reshaped to [images, ...])
      rle_str uint8 [N, max_out, stride] + rle_len int32 [N, max_out] (dtc_mask_rle) and im_sizes [N,2] = (h, w): optional

    all_boxes[cls][image] = [k,5] array (x1,y1,x2,y2,score); all_segms[cls][image] = list of COCO RLE dicts in 1:1
    correspondence -- exactly what `extend_results(i, all_boxes, cls_boxes_i)` / `extend_results(i, all_segms, cls_segms_i)`
    leave behind image by image in the reference's eval loop.  Pass all_boxes / all_segms / first_image to fill a slice of
    existing lists (e.g. the shard of one rank).  on_overflow: an image with more detections (ties at the image threshold) than the
    max_out fixed rows raises ("raise", default) or keeps the first max_out rows with a RuntimeWarning ("truncate")."""
    if on_overflow not in ("raise", "truncate"):
        raise ValueError("on_overflow must be 'raise' or 'truncate'")
    dets, det_count = to_np(dets), to_np(det_count).reshape(-1)
    N, max_out = dets.shape[0], dets.shape[1]
    if all_boxes is None:
        all_boxes, all_segms, _ = empty_results(num_classes, first_image + N)
    want_segms = rle_str is not None
    if want_segms:
        rle_str, rle_len = to_np(rle_str), to_np(rle_len).reshape(N, -1)
        im_sizes = to_np(im_sizes).reshape(N, -1)
    for i in range(N):
        if int(det_count[i]) > max_out:      # ties at the image threshold beyond the fixed rows (result_utils.py:159-163 keeps them all)
            msg = "image %d: %d detections but only %d rows were kept: raise the path's max_out" % (first_image + i, int(det_count[i]), max_out)
            if on_overflow == "raise":
                raise RuntimeError(msg)
            import warnings
            warnings.warn(msg + " -- truncated", RuntimeWarning)
        n = min(int(det_count[i]), max_out)
        d = dets[i, :n]
        cls = d[:, 5].astype(np.int64)
        for j in range(1, num_classes):
            sel = np.flatnonzero(cls == j)
            all_boxes[j][first_image + i] = np.ascontiguousarray(d[sel, :5])
            if want_segms:
                h, w = int(im_sizes[i, 0]), int(im_sizes[i, 1])
                segs = []
                for k in sel:
                    if rle_len[i, k] < 0:
                        raise RuntimeError("device RLE buffer too small (or mask crop capacity exceeded) for image %d "
                                           "detection %d" % (first_image + i, k))
                    if rle_len[i, k] > rle_str.shape[2]:     # a gatherer that ships fewer bytes than the device stride
                        raise RuntimeError("RLE string of image %d detection %d was truncated in transit (%d > %d bytes)"
                                           % (first_image + i, k, rle_len[i, k], rle_str.shape[2]))
                    segs.append({'size': [h, w], 'counts': rle_str[i, k, :rle_len[i, k]].tobytes().decode('ascii')})
                all_segms[j][first_image + i] = segs
    return all_boxes, all_segms


def coco_bbox_results(all_boxes, image_ids, class_to_cat_id):
    """The records _write_coco_bbox_results_file dumps (json_dataset_evaluator.py:149-190): xywh boxes, one dict per detection.
    class_to_cat_id[j] = COCO category id of class index j (index 0 = background, ignored)."""
    res = []
    for j in range(1, len(all_boxes)):
        for i, image_id in enumerate(image_ids):
            d = all_boxes[j][i]
            if isinstance(d, list) and len(d) == 0:
                continue
            d = np.asarray(d, dtype=np.float64)
            xywh = box_utils.xyxy_to_xywh(d[:, 0:4])
            res.extend({'image_id': image_id, 'category_id': class_to_cat_id[j],
                        'bbox': [xywh[k, 0], xywh[k, 1], xywh[k, 2], xywh[k, 3]], 'score': d[k, -1]} for k in range(d.shape[0]))
    return res


def coco_segm_results(all_boxes, all_segms, image_ids, class_to_cat_id):
    """The records _write_coco_segms_results_file dumps (json_dataset_evaluator.py:67-113)."""
    res = []
    for j in range(1, len(all_boxes)):
        for i, image_id in enumerate(image_ids):
            d, rles = all_boxes[j][i], all_segms[j][i]
            if isinstance(d, list) and len(d) == 0:
                continue
            d = np.asarray(d, dtype=np.float64)
            assert len(rles) == d.shape[0]
            res.extend({'image_id': image_id, 'category_id': class_to_cat_id[j], 'segmentation': rles[k], 'score': d[k, -1]}
                       for k in range(d.shape[0]))
    return res
