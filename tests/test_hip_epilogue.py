"""dtc_bias_act (the fused convolution epilogue) against the same float32 arithmetic in torch, and
detector.optimize_for_inference (BatchNorm folded, fused epilogues, weights stored in the compute type) against the unoptimised
model.  -m gpu."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

# OUTSIDE SURVEY section 8 (round-3 backbone work, frozen): `-m "gpu and not offscope"` is the hot-path suite
pytestmark = [pytest.mark.gpu, pytest.mark.offscope]


def _ref(x, bias, res, relu, up2):
    """x + bias + residual in float32 in that order, ReLU, ONE rounding to x's type (what the kernel documents)."""
    y = x.float()
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    if res is not None:
        r = res.float()
        if up2:
            r = F.interpolate(r, scale_factor=2, mode="nearest")
        y = y + r
    if relu:
        y = torch.where(y < 0, torch.zeros_like(y), y)
    return y.to(x.dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("shape", [(2, 64, 24, 40), (3, 6, 10, 14), (1, 256, 50, 84), (2, 8, 6, 2), (2, 16, 26, 42), (4, 8, 14, 14),
                                   (1, 3, 6, 10), (5, 2, 2, 2)])
def test_bias_act_every_variant_equals_float32_reference(dtype, layout, shape):
    """bias / residual / x2-upsampled residual / ReLU in every combination, vectorisable and odd shapes, both dense layouts,
    three types: bit-identical to x.float() + bias + residual -> relu -> one rounding."""
    from detectorch_amd import hip
    g = torch.Generator(device="cuda"); g.manual_seed(hash((str(dtype), layout, shape)) % (2 ** 31))
    n, c, h, w = shape
    fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
    mk = lambda *s: (torch.randn(*s, generator=g, device="cuda") * 2).to(dtype).contiguous(memory_format=fmt)
    bias = torch.randn(c, generator=g, device="cuda")
    for use_bias in (True, False):
        for res_kind in (None, "same", "up2"):
            for relu in (True, False):
                x = mk(n, c, h, w)
                res = None if res_kind is None else mk(n, c, h, w) if res_kind == "same" else mk(n, c, h // 2, w // 2)
                want = _ref(x, bias if use_bias else None, res, relu, res_kind == "up2")
                got = hip.bias_act_(x, bias if use_bias else None, res, relu=relu, residual_up2=res_kind == "up2")
                assert got is x and got.is_contiguous(memory_format=fmt)
                assert torch.equal(got, want), (use_bias, res_kind, relu)


def test_bias_act_nan_propagates_and_bad_arguments_raise():
    from detectorch_amd import hip
    x = torch.tensor([[[[float("nan"), -1.0, 2.0, float("-inf")]]]], device="cuda")
    y = hip.bias_act_(x.clone(), torch.zeros(1, device="cuda"))
    assert torch.isnan(y[0, 0, 0, 0]) and y[0, 0, 0, 1:].tolist() == [0.0, 2.0, 0.0]
    z = torch.zeros(2, 4, 6, 6, device="cuda")
    with pytest.raises(ValueError):
        hip.bias_act_(z[:, :, ::2], torch.zeros(4, device="cuda"))                   # not dense
    with pytest.raises(ValueError):
        hip.bias_act_(z, torch.zeros(5, device="cuda"))                               # bias length
    with pytest.raises(ValueError):
        hip.bias_act_(z, None, torch.zeros(2, 4, 3, 3, device="cuda"))                # half-size residual without up2
    with pytest.raises(ValueError):
        hip.bias_act_(z, None, torch.zeros(2, 4, 6, 6, device="cuda", dtype=torch.float16))
    with pytest.raises(RuntimeError):
        hip.bias_act_(torch.zeros(1, 1, 2, 2), None)                                  # CPU tensor: no fallback
    assert hip.bias_act_(torch.zeros(0, 4, 6, 6, device="cuda"), torch.zeros(4, device="cuda")).numel() == 0


def _fpn_model(channels_last=False):
    from detectorch_amd.model.detector import detector
    torch.manual_seed(0)
    m = detector(arch='resnet50', conv_body_layers=['conv1', 'bn1', 'relu', 'maxpool', 'layer1', 'layer2', 'layer3', 'layer4'],
                 conv_head_layers='two_layer_mlp', fpn_layers=['layer1', 'layer2', 'layer3', 'layer4'], fpn_extra_lvl=True,
                 roi_height=7, roi_width=7, roi_spatial_scale=[0.25, 0.125, 0.0625, 0.03125], roi_sampling_ratio=2,
                 use_rpn_head=True, use_mask_head=True, mask_head_type='1up4convs', channels_last=channels_last)
    g = torch.Generator(); g.manual_seed(1)
    for mod in m.modules():                                   # non-trivial AffineChannel constants (a fresh BatchNorm is the identity)
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.weight.data = 0.5 + torch.rand(mod.num_features, generator=g)
            mod.bias.data = torch.randn(mod.num_features, generator=g) * 0.1
            mod.running_mean.data = torch.randn(mod.num_features, generator=g) * 0.1
            mod.running_var.data = 0.5 + torch.rand(mod.num_features, generator=g)
    m = m.cuda()
    if channels_last:
        m = m.to(memory_format=torch.channels_last)
    m.classif_head.weight.data *= 60.0                        # random weights: make some detections exist
    return m


@pytest.mark.parametrize("channels_last", [False, True])
def test_optimize_for_inference_float32_matches_the_eager_model(channels_last):
    """BatchNorm folded into the convs + fused epilogues, float32: FPN levels, RPN outputs, box-head and mask-head outputs of the
    same inputs agree with the eager modules to float32 rounding (the scale moves from the conv output to the weights)."""
    import copy
    eager = _fpn_model(channels_last)
    fast = copy.deepcopy(eager).optimize_for_inference()
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x = torch.randn(2, 3, 256, 320, generator=g, device="cuda")
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        fe, ff = eager.conv_body(x), fast.conv_body(x)
        for a, b in zip(fe, ff):
            assert a.shape == b.shape
            assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()), float((a - b).abs().max() / a.abs().max())
        (ce, be), (cf, bf) = eager.rpn(fe[0], logits=True), fast.rpn(fe[0], logits=True)
        assert torch.allclose(ce, cf, rtol=1e-4, atol=1e-5) and torch.allclose(be, bf, rtol=1e-4, atol=1e-5)
        r = torch.relu(torch.randn(6, 256, 14, 14, generator=g, device="cuda"))
        me, mf = eager.mask_head.conv_head(r), fast.mask_head.conv_head(r.clone())
        assert float((me - mf).abs().max()) <= 1e-4 * float(me.abs().max())
    with pytest.raises(RuntimeError):
        fast.optimize_for_inference()                          # one-way


@pytest.mark.parametrize("dtype,channels_last", [(None, False), (torch.bfloat16, True), (torch.float16, False)])
def test_forward_batched_on_an_optimised_model(dtype, channels_last):
    """The whole batched flow on the inference form: float32 -> the same detections as the eager model wherever the proposals
    coincide (they do: deterministic convs, differences ~1e-6); 16-bit -> 16-bit feature maps and pooled features without
    autocast, finite results, detections present; forward() refuses a 16-bit-optimised model."""
    import copy
    old = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    try:
        eager = _fpn_model(channels_last)
        fast = copy.deepcopy(eager).optimize_for_inference(dtype)
        g = torch.Generator(device="cuda"); g.manual_seed(9)
        images = torch.randn(2, 3, 320, 448, generator=g, device="cuda")
        sf, im_size = torch.tensor([1.6, 1.6], device="cuda"), torch.tensor([[200.0, 280.0], [200.0, 280.0]], device="cuda")
        p = fast.forward_batched(images, sf, im_size)
        torch.cuda.synchronize()
        want = dtype or torch.float32
        assert all(f.dtype == want for f in p.feats) and p.box_feats.dtype == want
        assert min(p.det_count.tolist()) > 0 and bool(torch.isfinite(p.dets).all()) and bool(torch.isfinite(p.cls_logits_out).all())
        if dtype is None:
            # same function up to float32 rounding: the RPN outputs agree closely; the proposals they select coincide except
            # where two scores were within that rounding of each other (random weights: many near-ties)
            cls_f, box_f, n_rois, rois = [t.clone() for t in p.rpn_cls], [t.clone() for t in p.rpn_bbox], p.n_rois.clone(), p.rois5.clone()
            q = eager.forward_batched(images, sf, im_size)
            torch.cuda.synchronize()
            for a, b in zip(cls_f + box_f, list(q.rpn_cls) + list(q.rpn_bbox)):
                assert float((a - b).abs().max()) <= 1e-3 * max(1.0, float(b.abs().max()))
            assert torch.equal(n_rois, q.n_rois)
            close = float(((rois - q.rois5).abs().amax(dim=2) < 0.05).float().mean())
            assert close > 0.5, close                           # position by position: one swapped near-tie shifts what follows it
        else:
            with pytest.raises(NotImplementedError):
                fast(images[:1], scaling_factor=sf[:1])
    finally:
        torch.backends.cudnn.deterministic = old
