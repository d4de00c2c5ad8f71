"""End-to-end parity of the fused, batched, graph-replayed hot path (detectorch_amd.pipeline.FpnRegionPath) against the
oracle chain, image by image: every intermediate bit-exact (proposals, NMS survivors, collected rois, level ids, restore
permutation, pooled features, detections, mask-branch features, binarised mask crops).  -m gpu."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_graph", [False, True])
def test_fpn_region_path_vs_oracle_chain(oracle, use_graph):
    import chain
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C = 2, 8
    path = FpnRegionPath(B, dev, channels=C)
    inputs = synthetic_batch(B, dev, seed=3000, channels=C)
    path.bind(*inputs)
    path.step(use_graph=use_graph)
    if use_graph:
        path.step(use_graph=True)          # replay twice: counters / histograms must be reset inside the graph
    torch.cuda.synchronize()
    rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = [
        [t.cpu().numpy() for t in x] if isinstance(x, list) else x.cpu().numpy() for x in inputs]
    for b in range(B):
        ref = chain.fpn_hot_path([c[b] for c in rpn_cls], [d[b] for d in rpn_bbox], [f[b:b + 1] for f in feats],
                                 cls_score[b], bbox_pred[b], masks[b * path.max_out:(b + 1) * path.max_out], sf[b],
                                 im_size[b], path.pad_h, path.pad_w)
        assert chain.compare_with_gpu(path, b, ref, int(im_size[b, 0]), int(im_size[b, 1]))
        assert ref["dets"].shape[0] >= 100


def test_cfg5_shape_2000_proposals_fp16_features(oracle):
    """BASELINE cfg5: collect top-N = 2000 (the train-time constant of collect...py:86 / detector.py:207) and fp16 feature
    maps.  Oracle = CPU loop on the fp16->fp32 up-cast maps; fp32-accumulated features stored as fp16 -> rel 1e-3."""
    import chain
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C = 1, 8
    path = FpnRegionPath(B, dev, channels=C, collect_top_n=2000, feat_dtype=torch.float16)
    inputs = synthetic_batch(B, dev, seed=5000, channels=C, top_n=2000, feat_dtype=torch.float16)
    path.bind(*inputs)
    path.step(use_graph=False)
    torch.cuda.synchronize()
    rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = [
        [t.float().cpu().numpy() for t in x] if isinstance(x, list) else x.float().cpu().numpy() for x in inputs]
    ref = chain.fpn_hot_path([c[0] for c in rpn_cls], [d[0] for d in rpn_bbox], [f[0:1] for f in feats], cls_score[0],
                             bbox_pred[0], masks[:path.max_out], sf[0], im_size[0], path.pad_h, path.pad_w, top_n=2000)
    n = int(path.n_rois[0])
    assert n == ref["rois"].shape[0] and n > 1000
    assert np.array_equal(path.rois5[0, :n, 1:].cpu().numpy(), ref["rois"])
    assert np.array_equal(path.roi_levels[0, :n].cpu().numpy(), ref["roi_levels"])
    got = path.box_feats[:n].float().cpu().numpy()
    assert np.allclose(got, ref["box_feats"], rtol=1e-3, atol=1e-3)
    D = min(int(path.det_count[0]), path.max_out)
    assert np.array_equal(path.dets[0, :D].cpu().numpy(), ref["dets"][:D])


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_cfg5_real_shape_batch8_c256_fp16(oracle, layout):
    """BASELINE cfg5 at its REAL shape: B = 8 images x 2000 RoIs, C = 256, fp16 feature maps, hipGraph replay -- in BOTH layouts
    bench.py times: channels_last (cfg5's default since round 4: roi_align_fwd_nhwc16<__half,__half,64>, bins of four RoIs dealt over
    a workgroup) and NCHW (the cluster kernel with the 16-bit LDS image).  Image 0 against the oracle chain on the up-cast maps
    (proposals, levels, detections exact); the float32 output of the SAME 16 000 descriptors bit-equal to the oracle for EVERY image,
    and the fp16 output of the fused path == that, rounded once."""
    import chain
    from detectorch_amd import hip, synth
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C, T = 8, 256, 2000
    path = FpnRegionPath(B, dev, channels=C, collect_top_n=T, feat_dtype=torch.float16)
    inputs = synthetic_batch(B, dev, seed=5000, channels=C, top_n=T, feat_dtype=torch.float16, channels_last=layout == "nhwc")
    path.bind(*inputs)
    path.step(use_graph=True)
    path.step(use_graph=True)
    torch.cuda.synchronize()
    rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = inputs
    assert feats[0].is_contiguous(memory_format=torch.channels_last) == (layout == "nhwc")
    host = lambda t: np.ascontiguousarray(t.float().cpu().numpy())
    b = 0
    ref = chain.fpn_hot_path([host(c[b]) for c in rpn_cls], [host(d[b]) for d in rpn_bbox], [host(f[b:b + 1]) for f in feats],
                             host(cls_score[b]), host(bbox_pred[b]), host(masks[b * path.max_out:(b + 1) * path.max_out]),
                             float(sf[b]), host(im_size[b]), path.pad_h, path.pad_w, top_n=T)
    n = int(path.n_rois[b])
    assert n == ref["rois"].shape[0] and n > 1000
    assert np.array_equal(path.rois5[b, :n, 1:].cpu().numpy(), ref["rois"])
    assert np.array_equal(path.roi_levels[b, :n].cpu().numpy(), ref["roi_levels"])
    assert np.allclose(path.box_feats[:n].float().cpu().numpy(), ref["box_feats"], rtol=1e-3, atol=1e-3)
    D = min(int(path.det_count[b]), path.max_out)
    assert np.array_equal(path.dets[b, :D].cpu().numpy(), ref["dets"][:D])
    # fp32 output of the SAME 16 000-descriptor launch (fp16 maps, fp32 accumulate, no final rounding): bit-equal, every image
    out32 = torch.empty((B * T, C, 7, 7), dtype=torch.float32, device=dev)
    hip.check(hip.lib().dtc_roi_align_forward_packed(path.feat_lv, 4, C, hip.DTC_F16, path.roi_desc.data_ptr(), B * T, 7, 7, 2,
                                                     out32.data_ptr(), hip.DTC_F32, hip.stream_ptr(dev)), "packed fp16->fp32")
    torch.cuda.synchronize()
    assert np.array_equal(out32[:n].cpu().numpy(), ref["box_feats"])
    maps = [host(f) for f in feats]
    rois5, lv = path.rois5.reshape(-1, 5).cpu().numpy(), path.roi_levels.reshape(-1).cpu().numpy()
    got = out32.cpu().numpy()
    for l in range(4):
        m = lv == l
        assert np.array_equal(got[m], oracle.roi_align_forward(maps[l], np.ascontiguousarray(rois5[m]), 7, 7, synth.FPN_ROI_SCALES[l], 2))
    assert not got[lv < 0].any()                                      # padding rows: zeros
    assert torch.equal(out32.to(torch.float16), path.box_feats)       # every image: the fp16 output is that, rounded once


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_cfg5_real_shape_contract_mode_within_tolerance(oracle, layout, dtype):
    """CONTRACT mode on 16-bit maps (dtc_roi_align_set_exact(0), round 6): the pooling is one fused convert-multiply-accumulate per
    element (v_fma_mix_f32 / v_pk_fma_f32) instead of the reference's separate multiply and add.  The reference is float-only
    (roi_align_forward_cuda.cu:199-208); SURVEY cfg5 / north_star ask for <= 1e-4 on the float32-accumulated result.  At the real
    shape (8 x 2000 RoIs, C = 256, both layouts): the float32 output within 1e-4 (observed ~1e-6) of the exact-mode output, which the
    test above pins bit-equal to the oracle; the 16-bit output within ONE ulp of the exact mode's 16-bit output; the mode is read at
    launch time, and switching back restores bit-equality."""
    from detectorch_amd import hip
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C, T = 8, 256, 2000
    tdt, code = (torch.float16, hip.DTC_F16) if dtype == "f16" else (torch.bfloat16, hip.DTC_BF16)
    path = FpnRegionPath(B, dev, channels=C, collect_top_n=T, feat_dtype=tdt)
    inputs = list(synthetic_batch(B, dev, seed=5000, channels=C, top_n=T, feat_dtype=torch.float16, channels_last=layout == "nhwc"))
    if dtype == "bf16":
        inputs[2] = [f.to(torch.bfloat16) for f in inputs[2]]
    path.bind(*inputs)
    path.step(use_graph=False)
    torch.cuda.synchronize()

    def launch(out, ocode):
        hip.check(hip.lib().dtc_roi_align_forward_packed(path.feat_lv, 4, C, code, path.roi_desc.data_ptr(), B * T, 7, 7, 2,
                                                         out.data_ptr(), ocode, hip.stream_ptr(dev)), "packed")
        torch.cuda.synchronize()
        return out.clone()

    o32, o16 = torch.empty((B * T, C, 7, 7), device=dev), torch.empty((B * T, C, 7, 7), dtype=tdt, device=dev)
    exact32, exact16 = launch(o32, hip.DTC_F32), launch(o16, code)
    assert hip.lib().dtc_roi_align_get_exact() == 1
    hip.roi_align_set_exact(False)
    try:
        fused32, fused16 = launch(o32, hip.DTC_F32), launch(o16, code)
    finally:
        hip.roi_align_set_exact(True)
    d = (fused32 - exact32).abs()
    assert float(d.max()) <= 1e-4, float(d.max())
    assert float(d.max()) > 0.0                       # the fused kernels really ran (one rounding instead of two differs somewhere)
    # 16-bit output: at most one ulp apart, and only where the float32 value sits next to a rounding boundary
    a, b = exact16.view(torch.int16).int(), fused16.view(torch.int16).int()
    assert int((a - b).abs().max()) <= 1
    assert float(((a - b) != 0).float().mean()) < 1e-3
    assert torch.equal(launch(o32, hip.DTC_F32), exact32)   # back in exact mode: bit-equal again


def test_overlapped_split_equals_single(oracle):
    """Two sub-batches on two streams inside one hipGraph give exactly the single-stream result."""
    from detectorch_amd.pipeline import FpnRegionPath, OverlappedRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C = 4, 8
    inputs = synthetic_batch(B, dev, seed=3100, channels=C)
    one = FpnRegionPath(B, dev, channels=C)
    one.bind(*inputs)
    one.step(use_graph=True)
    two = OverlappedRegionPath(B, dev, n_split=2, channels=C)
    two.bind(*inputs)
    two.step(use_graph=True)
    two.step(use_graph=True)
    torch.cuda.synchronize()
    assert torch.equal(one.det_count, two.det_count)
    for b in range(B):
        n = min(int(one.det_count[b]), one.max_out)
        assert torch.equal(one.dets[b, :n], two.dets[b, :n])


@pytest.mark.parametrize("use_graph", [False, True])
def test_steps_in_flight_equal_steps_alone(oracle, use_graph):
    """StepPipeline: consecutive steps on two HIP streams (their kernels interleave on the device) leave exactly the results
    every path produces when it runs alone -- proposals, pooled features, detections, crops."""
    from detectorch_amd.pipeline import FpnRegionPath, StepPipeline, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C = 2, 16
    paths, want = [], []
    for s in range(2):
        p = FpnRegionPath(B, dev, channels=C)
        p.bind(*synthetic_batch(B, dev, seed=3300 + 17 * s, channels=C))
        p.crops.zero_()                      # capacity buffer: only the pasted rectangles are written
        p.step(use_graph=False)
        torch.cuda.synchronize()
        want.append([t.clone() for t in (p.rois5, p.n_rois, p.box_feats, p.dets, p.det_count, p.mask_feats, p.crops)])
        for t in (p.rois5, p.box_feats, p.dets, p.mask_feats, p.crops):
            t.zero_()
        paths.append(p)
    pipe = StepPipeline(paths, dev, n_inflight=2)
    for _ in range(7):
        pipe.step(use_graph=use_graph)
    pipe.synchronize()
    torch.cuda.synchronize()
    for p, w in zip(paths, want):
        got = (p.rois5, p.n_rois, p.box_feats, p.dets, p.det_count, p.mask_feats, p.crops)
        for g, e in zip(got, w):
            assert torch.equal(g, e)


def test_bench_configuration_vs_oracle_chain(oracle):
    """The exact configuration bench.py times -- seed 3000, batch 8, C = 256, hipGraph replay -- against the oracle chain for
    the first and the last image of the batch (every intermediate bit-exact, incl. the 8000-RoI box-head RoIAlign launch)."""
    import chain
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B = 8
    path = FpnRegionPath(B, dev)
    inputs = synthetic_batch(B, dev, seed=3000)
    path.bind(*inputs)
    path.step(use_graph=True)
    path.step(use_graph=True)
    torch.cuda.synchronize()
    rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = inputs
    host = lambda t: t.float().cpu().numpy()
    for b in (0, B - 1):
        ref = chain.fpn_hot_path([host(c[b]) for c in rpn_cls], [host(d[b]) for d in rpn_bbox], [host(f[b:b + 1]) for f in feats],
                                 host(cls_score[b]), host(bbox_pred[b]), host(masks[b * path.max_out:(b + 1) * path.max_out]),
                                 float(sf[b]), host(im_size[b]), path.pad_h, path.pad_w)
        assert chain.compare_with_gpu(path, b, ref, int(im_size[b, 0]), int(im_size[b, 1]))


def test_c4_region_path_vs_oracle_chain(oracle):
    """BASELINE configs[1] flavour (63 000 anchors -> 6000 -> NMS 0.7 -> 1000 -> RoIAlign on res4, adaptive sampling ->
    detections), batched + graph-replayed, against the oracle chain: every intermediate bit-exact.  C = 32 keeps the CPU side
    short; the true C = 1024 RoIAlign shape is covered by test_c4_true_shape_vs_reference_compiled."""
    import chain
    from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch
    dev = torch.device("cuda", 0)
    B, C = 2, 32
    for pooled in (7, 14):
        path = C4RegionPath(B, dev, channels=C, pooled=pooled)
        inputs = synthetic_c4_batch(B, dev, seed=2000, channels=C)
        path.bind(*inputs)
        path.step(use_graph=True)
        path.step(use_graph=True)
        torch.cuda.synchronize()
        rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, sf, im_size = [x.cpu().numpy() for x in inputs]
        for b in range(B):
            ref = chain.c4_hot_path(rpn_cls[b], rpn_bbox[b], feat[b:b + 1], cls_score[b], bbox_pred[b], sf[b], im_size[b],
                                    path.im_h, path.im_w, pooled=pooled)
            assert chain.compare_c4_with_gpu(path, b, ref)
            assert ref["rois"].shape[0] == 1000 and ref["dets"].shape[0] >= 100


@pytest.mark.parametrize("pooled", [7, 14])
def test_c4_region_path_true_channel_count(oracle, pooled):
    """BASELINE cfg2 at its REAL channel count (round-2 VERDICT: the C4RegionPath test ran C = 32): res4 [1,1024,50,84], 1000
    proposals, adaptive sampling, 7x7 (as cfg2 names it) and 14x14 (the reference's default): the whole chain for one image,
    every intermediate bit-exact (the map-stationary kernel's 128 channel-group workgroups all included)."""
    import chain
    from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch
    dev = torch.device("cuda", 0)
    B, C = 1, 1024
    path = C4RegionPath(B, dev, channels=C, pooled=pooled)
    inputs = synthetic_c4_batch(B, dev, seed=2100, channels=C)
    path.bind(*inputs)
    path.step(use_graph=True)
    path.step(use_graph=True)
    torch.cuda.synchronize()
    rpn_cls, rpn_bbox, feat, cls_score, bbox_pred, sf, im_size = [x.cpu().numpy() for x in inputs]
    ref = chain.c4_hot_path(rpn_cls[0], rpn_bbox[0], feat[0:1], cls_score[0], bbox_pred[0], sf[0], im_size[0], path.im_h,
                            path.im_w, pooled=pooled)
    assert chain.compare_c4_with_gpu(path, 0, ref)
    assert ref["rois"].shape[0] == 1000


@pytest.mark.parametrize("use_graph", [False, True])
def test_channels_last_maps_equal_nchw_maps(oracle, use_graph):
    """The same inputs with channels_last float32 feature maps (C = 128): box-head features through the grouped direct kernel
    (roi_align_fwd_nhwc16<float>: the default for float32 launches with <= 64 bins since round 4; the LDS-DMA kernel it displaced
    runs in the DTC_RA_NHWC_DIRECT32=0 child processes of test_hip_roi_align.py), mask-branch features (14 x 14 bins, packed
    descriptors with padding rows) through the pipelined kernel roi_align_fwd_nhwc_pipe -- every result equal, bit for bit, to the
    NCHW path's (which the oracle-chain tests pin)."""
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    dev = torch.device("cuda", 0)
    B, C = 3, 128
    res = []
    for cl in (False, True):
        p = FpnRegionPath(B, dev, channels=C)
        p.bind(*synthetic_batch(B, dev, seed=3100, channels=C, channels_last=cl))
        p.crops.zero_()
        p.step(use_graph=use_graph)
        if use_graph:
            p.step(use_graph=True)
        torch.cuda.synchronize()
        res.append([t.clone() for t in (p.rois5, p.n_rois, p.box_feats, p.dets, p.det_count, p.mask_feats, p.crops)])
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert int(res[0][4].min()) > 0 and float(res[0][5].abs().sum()) > 0


def test_more_ties_than_rows_raise_or_truncate(oracle):
    """More detections than the fixed max_out rows (ties at the image threshold, result_utils.py:159-163) never disappear silently:
    results() / assemble_results raise by default; on_overflow="truncate" keeps the first max_out rows, warns and flags the image."""
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    from detectorch_amd.utils import result_utils
    dev = torch.device("cuda", 0)
    path = FpnRegionPath(2, dev, channels=8, max_out=104)
    path.bind(*synthetic_batch(2, dev, seed=3000, channels=8, max_out=104))
    path.step(use_graph=False)
    torch.cuda.synchronize()
    ok = path.results()
    assert len(ok) == 2 and "truncated" not in ok[0]
    path.det_count[1] = path.max_out + 5             # what the kernel reports when 109 detections tie into the top 100
    with pytest.raises(RuntimeError):
        path.results()
    with pytest.raises(RuntimeError):
        result_utils.assemble_results(path.dets, path.det_count)
    with pytest.warns(RuntimeWarning):
        got = path.results(on_overflow="truncate")
    assert got[1]["truncated"] and got[1]["n_detections"] == path.max_out + 5 and got[1]["boxes"].shape[0] == path.max_out
    assert np.array_equal(got[0]["boxes"], ok[0]["boxes"])
    with pytest.warns(RuntimeWarning):
        boxes, _ = result_utils.assemble_results(path.dets, path.det_count, on_overflow="truncate")
    # (the rows past the true count are zero rows of class 0 here: the count was faked)
    assert sum(len(boxes[j][1]) for j in range(1, 81)) == int((path.dets[1, :, 5] >= 1).sum())
