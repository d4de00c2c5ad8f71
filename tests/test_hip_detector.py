"""The reference's eval_mask_FPN.ipynb flow (cells 4, 6) through the drop-in modules on the GPU, with random weights:
model(image, scaling_factor) -> postprocess_output -> add_multilevel_rois_for_test -> model.mask_head -> segm_results.
Checks the call surface / shapes / invariants (AP needs COCO + Detectron weights, unavailable offline).  -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _deterministic_convs():
    """MIOpen's default conv algorithms are not reproducible from call to call (feature maps differ by ~3e-7 between two
    forward passes of the same input, measured: tools/r02/dbg_det.py), which can swap near-tied proposals between the two
    flows these tests compare.  The deterministic algorithms reproduce bit for bit."""
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    yield
    torch.backends.cudnn.deterministic = old


def _fpn_model(arch="resnet50", channels_last=False):
    from detectorch_amd.model.detector import detector
    torch.manual_seed(0)
    m = detector(arch=arch, conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                 conv_head_layers='two_layer_mlp', fpn_layers=['layer1', 'layer2', 'layer3', 'layer4'], fpn_extra_lvl=True,
                 roi_height=7, roi_width=7, roi_spatial_scale=[0.25, 0.125, 0.0625, 0.03125], roi_sampling_ratio=2,
                 use_rpn_head=True, use_mask_head=True, mask_head_type='1up4convs', channels_last=channels_last)
    return m.cuda()


@pytest.mark.parametrize("channels_last", [False, True])
def test_mask_rcnn_fpn_eval_flow(channels_last):
    from detectorch_amd.utils import result_utils
    from detectorch_amd.utils.multilevel_rois import add_multilevel_rois_for_test
    model = _fpn_model(channels_last=channels_last)
    if channels_last:
        model = model.to(memory_format=torch.channels_last)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    image = torch.randn(1, 3, 320, 448, generator=g, device="cuda")
    sf = torch.tensor([1.6], device="cuda")
    cls_score, bbox_pred, rois, feats = model(image, scaling_factor=sf)
    R = rois.shape[0]
    assert 0 < R <= 1000 and tuple(cls_score.shape) == (R, 81) and tuple(bbox_pred.shape) == (R, 324)
    assert len(feats) == 4 and feats[0].shape[1] == 256 and tuple(feats[0].shape[2:]) == (80, 112)
    assert torch.allclose(cls_score.sum(1), torch.ones(R, device="cuda"), atol=1e-4)
    im_size = torch.tensor([200.0, 280.0, 3.0])
    # random weights give ~uniform class scores (1/81 < 0.05): lower the bar by boosting scores so detections exist
    boosted = torch.softmax(torch.log(cls_score) * 40.0, dim=1)
    scores_final, boxes_final, boxes_per_class = result_utils.postprocess_output(rois, sf, im_size, boosted, bbox_pred)
    assert len(boxes_per_class) == 81 and scores_final.shape[0] == boxes_final.shape[0]
    if len(boxes_final) == 0:
        pytest.skip("no detections with these random weights")
    assert boxes_final[:, 0::2].max() <= 279 and boxes_final[:, 1::2].max() <= 199 and boxes_final.min() >= 0
    blobs = add_multilevel_rois_for_test({'rois': boxes_final * 1.6}, 'rois')
    per_level = []
    for k in ['rois_fpn2', 'rois_fpn3', 'rois_fpn4', 'rois_fpn5']:
        per_level.append(torch.from_numpy(blobs[k]).cuda() if len(blobs[k]) > 0 else None)
    restore = torch.from_numpy(blobs['rois_idx_restore_int32']).cuda().long()
    masks = model.mask_head(feats, per_level, restore)
    D = boxes_final.shape[0]
    assert tuple(masks.shape) == (D, 81, 28, 28) and float(masks.min()) >= 0 and float(masks.max()) <= 1
    segms = result_utils.segm_results(boxes_per_class, masks, boxes_final, 200, 280, M=28)
    assert sum(len(s) for s in segms) == D and all(r['size'] == [200, 280] for s in segms for r in s)


def test_faster_rcnn_c4_flow(oracle):
    """eval_faster.ipynb flow (detector.py:240-248): shapes, AND the proposals the model returns == the oracle's GenerateProposals on
    the model's OWN RPN-head outputs (captured with a forward hook), AND the pooled features that reach the res5 head == the oracle's
    RoIAlign of those proposals on the model's own res4 map."""
    from detectorch_amd.model.detector import detector
    torch.manual_seed(0)
    model = detector(arch='resnet50', use_rpn_head=True).cuda()
    seen = {}
    h1 = model.rpn.register_forward_hook(lambda m, i, o: seen.__setitem__("rpn", (o[0].clone(), o[1].clone())))
    h2 = model.conv_head.register_forward_hook(lambda m, i, o: seen.__setitem__("pooled", i[0].clone()))
    image = torch.randn(1, 3, 256, 320, device="cuda")
    cls_score, bbox_pred, rois, feats = model(image, scaling_factor=1.0)
    h1.remove(); h2.remove()
    assert rois.shape[1] == 4 and cls_score.shape[0] == rois.shape[0] and bbox_pred.shape[1] == 324
    assert tuple(feats.shape) == (1, 1024, 16, 20)
    score, deltas = seen["rpn"]
    prob = oracle.rpn_sigmoid(score[0].cpu().numpy()) if model.fuse_rpn_sigmoid else score[0].cpu().numpy()
    want, _ = oracle.generate_proposals(prob, deltas[0].cpu().numpy(), oracle.generate_anchors(stride=16), 16.0, 256, 320, 6000, 1000, 0.7)
    assert want.shape[0] > 50 and np.array_equal(rois.cpu().numpy(), want)
    rois5 = np.hstack([np.zeros((want.shape[0], 1), np.float32), want])
    ref = oracle.roi_align_forward(feats.cpu().numpy(), rois5, 14, 14, 0.0625, 0)
    assert np.array_equal(seen["pooled"].cpu().numpy(), ref)


def test_fast_rcnn_c4_precomputed_rois_branch(oracle):
    """eval_fast.ipynb:227 (BASELINE configs[0]'s plumbing): detector without an RPN head, proposals handed in by the caller as a
    [1,R,4] tensor (the notebook's `rois.unsqueeze(0)`-style batch) -- detector.py:240-248 `rois=` branch.  The pooled features
    that reach the head are the oracle's RoIAlign of THOSE rows in THAT order; the returned rois are the caller's."""
    from detectorch_amd import synth
    from detectorch_amd.model.detector import detector
    torch.manual_seed(0)
    model = detector(arch='resnet50', use_rpn_head=False).cuda()
    seen = {}
    hk = model.conv_head.register_forward_hook(lambda m, i, o: seen.__setitem__("pooled", i[0].clone()))
    image = torch.randn(1, 3, 256, 320, device="cuda")
    boxes = synth.make_rois(synth.rng(9, 1), 37, im_h=256, im_w=320, min_side=12.0, max_side=200.0)
    cls_score, bbox_pred, rois, feats = model(image, rois=torch.from_numpy(boxes).cuda().unsqueeze(0), scaling_factor=1.0)
    hk.remove()
    assert tuple(cls_score.shape) == (37, 81) and tuple(bbox_pred.shape) == (37, 324)
    ref = oracle.roi_align_forward(feats.cpu().numpy(), np.hstack([np.zeros((37, 1), np.float32), boxes]), 14, 14, 0.0625, 0)
    assert np.array_equal(seen["pooled"].cpu().numpy(), ref)
    rows = rois[0] if rois.dim() == 3 else rois
    assert np.array_equal(rows.cpu().numpy()[:, -4:], boxes)


def test_fast_rcnn_fpn_precomputed_per_level_rois_and_restore_index(oracle):
    """eval_fast_FPN.ipynb:238: FPN detector without an RPN head; the caller distributes its proposals over the levels
    (add_multilevel_rois_for_test) and passes the per-level lists + `roi_original_idx` (detector.py:260-270).  Row i of the pooled
    features that reach fc6 must be the oracle's RoIAlign of ORIGINAL proposal i on ITS level's map -- a wrong level order or a
    wrong restore index fails this -- and the returned rois are the original proposals in the original order."""
    from detectorch_amd import synth
    from detectorch_amd.model.detector import detector
    from detectorch_amd.utils.multilevel_rois import add_multilevel_rois_for_test
    torch.manual_seed(0)
    model = detector(arch='resnet50', conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                     conv_head_layers='two_layer_mlp', fpn_layers=['layer1', 'layer2', 'layer3', 'layer4'], fpn_extra_lvl=True,
                     roi_height=7, roi_width=7, roi_spatial_scale=[0.25, 0.125, 0.0625, 0.03125], roi_sampling_ratio=2,
                     use_rpn_head=False).cuda()
    seen = {}
    hk = model.conv_head.register_forward_hook(lambda m, i, o: seen.__setitem__("pooled", i[0].clone()))
    image = torch.randn(1, 3, 480, 640, device="cuda")
    boxes = synth.make_rois(synth.rng(9, 2), 83, im_h=480, im_w=640, min_side=10.0, max_side=900.0)
    blobs = add_multilevel_rois_for_test({'rois': boxes}, 'rois')
    keys = ['rois_fpn2', 'rois_fpn3', 'rois_fpn4', 'rois_fpn5']
    assert all(len(blobs[k]) > 0 for k in keys)                       # every level is populated: a level swap cannot hide
    per_level = [torch.from_numpy(np.ascontiguousarray(blobs[k])).cuda() for k in keys]
    restore = torch.from_numpy(blobs['rois_idx_restore_int32']).cuda().long()
    assert not np.array_equal(blobs['rois_idx_restore_int32'], np.arange(83))
    cls_score, bbox_pred, rois, feats = model(image, rois=per_level, scaling_factor=1.0, roi_original_idx=restore)
    hk.remove()
    assert tuple(cls_score.shape) == (83, 81) and np.array_equal(rois.cpu().numpy(), boxes)
    lv = oracle.map_rois_to_fpn_levels(boxes, 2, 5) - 2
    ref = np.zeros((83, 256, 7, 7), np.float32)
    rois5 = np.hstack([np.zeros((83, 1), np.float32), boxes])
    for l in range(4):
        m = lv == l
        ref[m] = oracle.roi_align_forward(feats[l].cpu().numpy(), rois5[m], 7, 7, [0.25, 0.125, 0.0625, 0.03125][l], 2)
    assert np.array_equal(seen["pooled"].cpu().numpy().reshape(83, 256, 7, 7), ref)


def _boost(model, k=60.0):
    # random weights give ~uniform class scores (1/81 < the 0.05 detection threshold): sharpen the classifier so that
    # detections exist and the NMS / top-100 / mask branches run
    model.classif_head.weight.data *= k
    return model


@pytest.mark.parametrize("optimized", [False, True])
def test_forward_batched_equals_reference_shaped_forward(optimized):
    """detector.forward_batched (backbone(B) -> FpnRegionPath stages -> heads -> detections -> mask branch, no host round
    trip) against the reference-shaped batch-1 calls: forward() + postprocess_output + add_multilevel_rois_for_test +
    mask_head + segm_results (eval_mask_FPN.ipynb cells 4, 6).  optimized: the same on the float32 inference form of the model
    (optimize_for_inference: BatchNorm folded, fused epilogues, 104 detection rows per image) -- both flows run the fused modules."""
    from detectorch_amd.model.detector import detector
    from detectorch_amd.utils import result_utils
    from detectorch_amd.utils.multilevel_rois import add_multilevel_rois_for_test
    model = _boost(_fpn_model())
    if optimized:
        model.optimize_for_inference()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    image = torch.randn(1, 3, 320, 448, generator=g, device="cuda")
    sf, im_size = torch.tensor([1.6], device="cuda"), torch.tensor([[200.0, 280.0]], device="cuda")
    path = model.forward_batched(image, sf, im_size)
    torch.cuda.synchronize()
    assert path.max_out == (104 if optimized else 128)
    cls_b, bbox_b, rois_b, feats_b = detector.per_image(path, 0)
    cls_score, bbox_pred, rois, feats = model(image, scaling_factor=sf)
    # two passes through MIOpen's convs are not bit-reproducible (algorithm selection warms up): same proposals up to conv rounding
    assert rois.shape == rois_b.shape and torch.allclose(rois, rois_b, atol=1e-2)
    assert torch.allclose(bbox_pred, bbox_b, rtol=1e-3, atol=1e-3) and torch.allclose(cls_score, cls_b, rtol=1e-2, atol=1e-4)
    scores_final, boxes_final, boxes_per_class = result_utils.postprocess_output(rois, sf, im_size[0], cls_score, bbox_pred)
    D = boxes_final.shape[0]
    assert D > 0 and D == min(int(path.det_count[0]), path.max_out)
    dets = path.dets[0, :D].cpu().numpy()
    assert np.array_equal(dets[:, 5].astype(np.int64), np.concatenate([np.full(len(boxes_per_class[j]), j) for j in range(1, 81)]))
    assert np.allclose(dets[:, 4], scores_final, rtol=1e-2, atol=1e-4) and np.allclose(dets[:, :4], boxes_final, atol=5e-2)
    # mask branch: the reference-shaped calls on the same detections
    blobs = add_multilevel_rois_for_test({'rois': boxes_final * 1.6}, 'rois')
    per_level = [torch.from_numpy(blobs[k]).cuda() if len(blobs[k]) > 0 else None for k in ['rois_fpn2', 'rois_fpn3', 'rois_fpn4', 'rois_fpn5']]
    masks = model.mask_head(feats, per_level, torch.from_numpy(blobs['rois_idx_restore_int32']).cuda().long())
    assert torch.allclose(masks, path.masks[:D], rtol=1e-2, atol=1e-3)
    segms = result_utils.segm_results(boxes_per_class, masks, boxes_final, 200, 280, M=28)
    _, got_segms = result_utils.assemble_results(path.dets, path.det_count, path.im_size, path.rle_str, path.rle_str_len)
    n_same = sum(a == b for j in range(1, 81) for a, b in zip(segms[j], got_segms[j][0]))
    assert sum(len(s) for s in segms) == D and n_same >= int(0.8 * D)     # identical unless conv rounding moves a 0.5-contour pixel


def test_forward_batched_batch2_and_bf16_head():
    """B = 2 runs both images at once (image 0 == its batch-1 result up to conv-batch rounding: same proposals); the bf16 head
    option (RoIAlign writes bf16, fc6/fc7 as bf16 MFMA GEMMs) stays within bf16 rounding of the fp32 head."""
    from detectorch_amd.model.detector import detector
    model = _boost(_fpn_model())
    g = torch.Generator(device="cuda"); g.manual_seed(6)
    images = torch.randn(2, 3, 320, 448, generator=g, device="cuda")
    sf, im_size = torch.tensor([1.6, 1.6], device="cuda"), torch.tensor([[200.0, 280.0], [200.0, 280.0]], device="cuda")
    p2 = model.forward_batched(images, sf, im_size)
    n2 = p2.n_rois.tolist()
    rois2 = p2.rois5[0, :n2[0], 1:].clone()
    d2 = p2.det_count.tolist()
    p1 = model.forward_batched(images[:1], sf[:1], im_size[:1])
    assert n2[0] == int(p1.n_rois[0]) and min(d2) > 0
    # conv outputs at batch 2 and batch 1 differ in the last bits (MIOpen picks per-shape algorithms), which can swap near-tied
    # proposals: compare as sets
    dmin = (rois2[:, None, :] - p1.rois5[0, :n2[0], 1:][None, :, :]).abs().amax(2).amin(1)
    assert float((dmin < 5e-2).float().mean()) > 0.95
    assert float(p2.rois5[1, :, 0].min()) == 1.0                       # image index carried in column 0
    ref_logits = p1.cls_logits_out.clone()
    model.head_dtype = torch.bfloat16
    model._paths.clear()
    pb = model.forward_batched(images[:1], sf[:1], im_size[:1])
    assert pb.box_feats.dtype == torch.bfloat16
    # rows are RoIs in proposal order; the conv outputs of two calls can differ in the last bits and swap near-tied proposals,
    # so match the RoIs of the two runs by their boxes before comparing the head outputs row by row
    nb = int(pb.n_rois[0])
    d = (pb.rois5[0, :nb, 1:][:, None, :] - p1.rois5[0, :n2[0], 1:][None, :, :]).abs().amax(2)
    dmin, j = d.min(1)
    ok = dmin < 1e-3
    assert float(ok.float().mean()) > 0.9
    err = (pb.cls_logits_out[0, :nb][ok] - ref_logits[0][j[ok]]).abs().max() / ref_logits.abs().max()
    assert float(err) < 5e-2


def test_forward_batched_bf16_backbone_feeds_roialign_and_heads_without_a_cast():
    """SURVEY 8f-2 remainder: backbone_dtype = head_dtype = bfloat16 -- ResNet / FPN / RPN convs under bf16 autocast, the bf16
    feature maps go straight into RoIAlign (fp32 accumulate), the pooled bf16 features straight into fc6.  The region path on
    those maps is exact: its bf16 output is the fp32-output launch of the same descriptors rounded once."""
    from detectorch_amd import hip
    model = _boost(_fpn_model())
    model.backbone_dtype = model.head_dtype = torch.bfloat16
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    images = torch.randn(2, 3, 320, 448, generator=g, device="cuda")
    sf, im_size = torch.tensor([1.6, 1.6], device="cuda"), torch.tensor([[200.0, 280.0], [200.0, 280.0]], device="cuda")
    p = model.forward_batched(images, sf, im_size)
    torch.cuda.synchronize()
    assert all(f.dtype == torch.bfloat16 for f in p.feats) and p.box_feats.dtype == torch.bfloat16
    assert p.rpn_cls[0].dtype == torch.float32 and min(p.det_count.tolist()) > 0
    assert bool(torch.isfinite(p.cls_logits_out).all()) and bool(torch.isfinite(p.dets).all())
    R = p.B * p.top_n
    out32 = torch.empty((R, p.C, 7, 7), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().dtc_roi_align_forward_packed(p.feat_lv, 4, p.C, hip.DTC_BF16, p.roi_desc.data_ptr(), R, 7, 7, 2,
                                                     out32.data_ptr(), hip.DTC_F32, hip.stream_ptr()), "packed bf16 -> fp32")
    torch.cuda.synchronize()
    assert torch.equal(out32.to(torch.bfloat16), p.box_feats)
