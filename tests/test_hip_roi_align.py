"""A1 parity: HIP RoIAlign (through the C ABI) vs the committed golden vectors and vs the oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden
from detectorch_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-4   # north_star tolerance on RoIAlign-pooled features; the kernel is written to be bit-exact (asserted too)


@pytest.fixture(scope="module")
def hip():
    from detectorch_amd import hip as h
    h.lib()
    return h


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("tag", ["p7s2", "p14s0", "p7s0", "p14s2", "p3x5s3"])
def test_golden(hip, tag):
    g = golden("roi_align")
    ph, pw, sr, scale = g["cfg_" + tag]
    out = hip.roi_align_forward(cu(g["features"]), float(scale), cu(g["rois5"]), int(ph), int(pw), int(sr)).cpu().numpy()
    ref = g["out_" + tag]
    assert np.abs(out - ref).max() <= TOL
    assert np.array_equal(out, ref)


def test_golden_4col_and_reference_abi(hip):
    g = golden("roi_align")
    f = cu(g["features"][:1])
    out = hip.roi_align_forward(f, 1 / 16., cu(g["rois5"][:, 1:]), 7, 7, 2).cpu().numpy()
    assert np.array_equal(out, g["out4col_p7s2"])
    # the reference-shaped entry point (launch_roi_align_forward_cuda's argument list), caller-allocated output
    rois = cu(g["rois5"])
    feat = cu(g["features"])
    R, C, H, W = rois.shape[0], feat.shape[1], feat.shape[2], feat.shape[3]
    top = torch.zeros(R, C, 7, 7, device="cuda")
    rc = hip.lib().launch_roi_align_forward_hip(top.numel(), feat.data_ptr(), rois.data_ptr(), 1 / 16., C, H, W, 7, 7, 2,
                                                top.data_ptr(), hip.stream_ptr())
    assert rc == 1
    assert np.array_equal(top.cpu().numpy(), g["out_p7s2"])


def test_module_surface(hip):
    from detectorch_amd.model.roi_align import RoIAlign, RoIAlignFunction, preprocess_rois
    g = golden("roi_align")
    feat = cu(g["features"])
    m = RoIAlign(7, 7, 1 / 16., 2)
    assert np.array_equal(m(feat, cu(g["rois5"])).cpu().numpy(), g["out_p7s2"])
    out = RoIAlignFunction.apply(feat[:1], preprocess_rois([cu(g["rois5"][:10, 1:]), cu(g["rois5"][10:, 1:])]), 7, 7,
                                 1 / 16., 2)
    assert np.array_equal(out.cpu().numpy(), g["out4col_p7s2"])
    with pytest.raises(TypeError):
        RoIAlignFunction.apply(feat, torch.from_numpy(g["rois5"]), 7, 7, 1 / 16., 2)
    with pytest.raises(RuntimeError):
        RoIAlignFunction.apply(feat.cpu(), torch.from_numpy(g["rois5"]), 7, 7, 1 / 16., 2)


def test_empty(hip):
    out = hip.roi_align_forward(torch.zeros(1, 4, 8, 8, device="cuda"), 0.25, torch.zeros(0, 5, device="cuda"), 7, 7, 2)
    assert tuple(out.shape) == (0, 4, 7, 7)


def _fpn_case(oracle, R, C, ph, sr, seed, batch=1):
    rs = synth.rng(3, seed)
    shapes = synth.fpn_level_shapes()[:4]
    feats = [synth.make_features(rs, (batch, C, h, w)) for (h, w) in shapes]
    rois = synth.make_rois(rs, R)
    lv = oracle.map_rois_to_fpn_levels(rois, 2, 5) - 2
    bidx = rs.randint(0, batch, (R, 1)).astype(np.float32)
    rois5 = np.hstack([bidx, rois])
    ref = np.zeros((R, C, ph, ph), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(feats[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], sr)
    return feats, rois5, lv.astype(np.int32), ref


@pytest.mark.parametrize("ph,sr,R", [(7, 2, 300), (14, 2, 100)])
def test_fpn_multilevel_vs_oracle(hip, oracle, ph, sr, R):
    feats, rois5, lv, ref = _fpn_case(oracle, R, 32, ph, sr, ph, batch=2)
    out = hip.roi_align_forward([cu(f) for f in feats], synth.FPN_ROI_SCALES, cu(rois5), ph, ph, sr, roi_levels=cu(lv))
    out = out.cpu().numpy()
    assert np.abs(out - ref).max() <= TOL
    assert np.array_equal(out, ref)


def test_channels_last_and_fp16(hip, oracle):
    feats, rois5, lv, ref = _fpn_case(oracle, 120, 64, 7, 2, 77)
    tf = [cu(f).contiguous(memory_format=torch.channels_last) for f in feats]
    out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv)).cpu().numpy()
    assert np.array_equal(out, ref)
    # fp16 feature maps (BASELINE cfg5): oracle on the up-cast maps; fp32 accumulate -> exact; fp16 store -> rel 1e-3
    h16 = [cu(f).half() for f in feats]
    ref16 = np.zeros_like(ref)
    for l in range(4):
        m = lv == l
        if m.any():
            ref16[m] = oracle.roi_align_forward(h16[l].float().cpu().numpy(), rois5[m], 7, 7, synth.FPN_ROI_SCALES[l], 2)
    out32 = hip.roi_align_forward(h16, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv)).cpu().numpy()
    assert np.array_equal(out32, ref16)
    out16 = hip.roi_align_forward(h16, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv),
                                  out_dtype=torch.float16).float().cpu().numpy()
    assert np.allclose(out16, ref16, rtol=1e-3, atol=1e-3)


def test_c4_adaptive_sampling_full_size(hip, oracle):
    # BASELINE cfg2 shape class: [1,C,50,84], 14x14, sampling_ratio=0 (adaptive grid up to 4x6), scale 1/16
    rs = synth.rng(2, 5)
    feat = synth.make_features(rs, (1, 48, 50, 84))
    rois = np.vstack([synth.make_rois(rs, 200), [[0, 0, 1332, 799]], [[5, 5, 6, 6]]]).astype(np.float32)
    rois5 = np.hstack([np.zeros((rois.shape[0], 1), np.float32), rois])
    for ph in (14, 7):
        out = hip.roi_align_forward(cu(feat), 1 / 16., cu(rois5), ph, ph, 0).cpu().numpy()
        assert np.array_equal(out, oracle.roi_align_forward(feat, rois5, ph, ph, 1 / 16., 0))


def test_linearity_full_size(hip):
    # size-independent property at BASELINE cfg3's full size (R=1000, C=256, 4 levels): RoIAlign is linear in the
    # features, roi_align(a*F) == a*roi_align(F) exactly for a power of two.
    rs = synth.rng(3, 99)
    shapes = synth.fpn_level_shapes()[:4]
    feats = [torch.from_numpy(synth.make_features(rs, (1, 256, h, w))).cuda() for (h, w) in shapes]
    rois = torch.from_numpy(np.hstack([np.zeros((1000, 1), np.float32), synth.make_rois(rs, 1000)])).cuda()
    lv = torch.from_numpy(rs.randint(0, 4, 1000).astype(np.int32)).cuda()
    a = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois, 7, 7, 2, roi_levels=lv)
    b = hip.roi_align_forward([f * 4.0 for f in feats], synth.FPN_ROI_SCALES, rois, 7, 7, 2, roi_levels=lv)
    assert torch.equal(a * 4.0, b)
    assert torch.isfinite(a).all() and float(a.abs().max()) > 0


@pytest.mark.parametrize("C", [1, 3, 5, 33, 65, 130])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_ragged_channel_counts_and_shapes(hip, oracle, C, layout):
    """Channel tails (C not a multiple of 4 / 8 / 64), odd pooled sizes, sampling ratios 1..3 and adaptive, tiny maps, both
    layouts: every kernel variant (LDS-staged, channels_last direct, general) against the oracle, bit-exact."""
    rs = synth.rng(1, 100 + C)
    for (H, W, ph, pw, sr, scale) in [(20, 30, 7, 7, 2, 1 / 16.), (9, 5, 1, 1, 1, 0.5), (13, 21, 2, 3, 3, 1 / 64.),
                                      (25, 42, 14, 14, 2, 1 / 32.), (50, 84, 7, 7, 0, 1 / 16.), (1, 1, 3, 3, 2, 0.25)]:
        feat = rs.standard_normal((2, C, H, W)).astype(np.float32)
        im_w, im_h = W / scale, H / scale
        rois = synth.make_rois(rs, 40, im_h=int(im_h) + 1, im_w=int(im_w) + 1, min_side=2, max_side=max(im_w, im_h) * 1.2)
        rois[::7] += 30.0                                   # some rois partly / fully outside the map
        rois5 = np.hstack([rs.randint(0, 2, (40, 1)).astype(np.float32), rois]).astype(np.float32)
        ref = oracle.roi_align_forward(feat, rois5, ph, pw, scale, sr)
        f = cu(feat)
        if layout == "nhwc":
            f = f.contiguous(memory_format=torch.channels_last)
        out = hip.roi_align_forward(f, scale, cu(rois5), ph, pw, sr).cpu().numpy()
        assert np.array_equal(out, ref), (C, layout, H, W, ph, pw, sr)


def test_ordered_and_packed_entry_points(hip, oracle):
    feats, rois5, lv, ref = _fpn_case(oracle, 200, 16, 7, 2, 4242, batch=2)
    tf = [cu(f) for f in feats]
    order = torch.from_numpy(np.random.RandomState(0).permutation(200).astype(np.int32)).cuda()
    out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv), roi_order=order)
    assert np.array_equal(out.cpu().numpy(), ref)
    # packed descriptors (batch, x1, y1, x2, y2, level, output row, 0) in visiting order, incl. a padding row (level -1)
    o = order.cpu().numpy()
    desc = np.zeros((201, 8), np.float32)
    desc[:200, :5] = rois5[o]
    desc[:200, 5] = lv[o]
    desc[:200, 6] = o
    desc[200] = [0, 0, 0, 10, 10, -1, 200, 0]
    outp = torch.full((201, 16, 7, 7), 7.0, device="cuda")
    lvs, ch, dt = hip.make_levels(tf, synth.FPN_ROI_SCALES)
    rc = hip.lib().dtc_roi_align_forward_packed(lvs, 4, ch, 0, cu(desc).data_ptr(), 201, 7, 7, 2, outp.data_ptr(), 0,
                                                hip.stream_ptr())
    assert rc == 0
    res = outp.cpu().numpy()
    assert np.array_equal(res[:200], ref) and not res[200].any()


def _c4_rois(rs, n, im_h=800, im_w=1333):
    """proposals of all sizes on a stride-16 map: from 2 x 2 feature pixels to most of the image (adaptive grids 1 .. 12)"""
    side = np.exp(rs.uniform(np.log(24), np.log(1100), (n, 2)))
    cx, cy = rs.uniform(0, im_w, n), rs.uniform(0, im_h, n)
    x1, y1 = np.clip(cx - side[:, 0] / 2, 0, im_w - 1), np.clip(cy - side[:, 1] / 2, 0, im_h - 1)
    x2, y2 = np.clip(cx + side[:, 0] / 2, 0, im_w - 1), np.clip(cy + side[:, 1] / 2, 0, im_h - 1)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)


@pytest.mark.parametrize("case", ["p7_slab", "p14_direct", "tail_c50", "fp16", "shuffled", "sr3", "col4", "rect_5x8", "rect_8x6", "rect_9x4"])
def test_map_stationary_kernel_vs_oracle(hip, oracle, case):
    """The single-level (C4) kernel roi_align_fwd_map: whole map of 8 channels in LDS, one RoI per wavefront.  Packed
    descriptors as dtc_fpn_collect_distribute emits them (image-major, with padding rows), adaptive sampling; 7x7 bins (LDS
    slab + 16-byte stores), 14x14 bins (direct stores, 4 bin chunks per RoI), a channel count that is not a multiple of 4
    (single quads, clamped tail), fp16 features, a NOT image-major order (every change of image re-stages the map: slow,
    still exact), a fixed non-2 sampling ratio, the 4-column (single image) plain entry, and rectangular bin grids (round 6: the
    lane <-> bin assignment of 7 x 7-like grids puts two whole bin rows into each ds_read_b128 lane group)."""
    rs = synth.rng(8, 77)
    B, H, W = 3, 50, 84
    C = 50 if case == "tail_c50" else 128
    ph = pw = 14 if case == "p14_direct" else 7
    if case.startswith("rect_"):      # rectangular bins: both sides of the two-bin-rows-per-lane-group assignment (pooled_h, pooled_w <= 8) and past it
        ph, pw = (int(v) for v in case[5:].split("x"))
    sr = 3 if case == "sr3" else 0
    n = 700 if C == 128 else 1500           # R * C >= 64 K: below that the dispatcher keeps the RoI-stationary kernel
    feat = synth.make_features(rs, (B, C, H, W))
    if case == "fp16":
        feat = feat.astype(np.float16).astype(np.float32)
    rois = _c4_rois(rs, n)
    img = np.sort(rs.randint(0, B, n)).astype(np.float32)
    if case == "col4":
        img[:] = 0
    rois5 = np.hstack([img[:, None], rois]).astype(np.float32)
    ref = oracle.roi_align_forward(feat, rois5, ph, pw, 1 / 16., sr)
    tf = cu(feat).half() if case == "fp16" else cu(feat)
    if case == "col4":
        out = hip.roi_align_forward(tf[:1].contiguous(), 1 / 16., cu(rois), ph, pw, sr).cpu().numpy()
        assert np.array_equal(out, ref)
        return
    order = np.arange(n)
    if case == "shuffled":
        order = rs.permutation(n)
    desc = np.zeros((n + 5, 8), np.float32)
    desc[:n, :5] = rois5[order]
    desc[:n, 6] = order
    pad_rows = np.arange(n, n + 5)
    desc[n:, 5] = -1
    desc[n:, 6] = pad_rows
    # padding rows interleaved at the image boundaries, like the fixed-shape batches of the fused path
    perm = np.argsort(np.concatenate([np.arange(n), np.linspace(0, n - 1, 5)]), kind="stable")
    desc = desc[perm]
    out = torch.full((n + 5, C, ph, pw), 3.0, device="cuda")
    lvs, ch, dt = hip.make_levels([tf], [1 / 16.])
    rc = hip.lib().dtc_roi_align_forward_packed(lvs, 1, ch, hip._dtype_code(dt), cu(desc).data_ptr(), n + 5, ph, pw, sr,
                                                out.data_ptr(), 0, hip.stream_ptr())
    assert rc == 0
    res = out.cpu().numpy()
    assert not res[n:].any()
    assert np.abs(res[:n] - ref).max() <= TOL
    assert np.array_equal(res[:n], ref)
    # the same launch with the per-launch preparation pass (dtc_roi_align_forward_packed_ws: geometry and axis samples formed
    # once per RoI by map_prep_kernel instead of once per 8-channel workgroup): bit-identical
    ws = hip.workspace(hip.lib().dtc_roi_align_workspace_bytes(n + 5), "cuda")
    out2 = torch.full((n + 5, C, ph, pw), 5.0, device="cuda")
    rc = hip.lib().dtc_roi_align_forward_packed_ws(lvs, 1, ch, hip._dtype_code(dt), cu(desc).data_ptr(), n + 5, ph, pw, sr,
                                                   out2.data_ptr(), 0, ws.data_ptr(), ws.numel(), hip.stream_ptr())
    assert rc == 0 and torch.equal(out2, out)
    # fast mode (dtc_roi_align_set_exact(0)): merged taps with separable weight sums -- the same sums in exact arithmetic, within
    # 1e-5 of the bit-exact result in float32 (BASELINE.json allows 1e-4 on pooled features); padding rows still zero; and the
    # switch back restores bit-identity
    try:
        hip.roi_align_set_exact(False)
        out3 = torch.full((n + 5, C, ph, pw), 7.0, device="cuda")
        rc = hip.lib().dtc_roi_align_forward_packed_ws(lvs, 1, ch, hip._dtype_code(dt), cu(desc).data_ptr(), n + 5, ph, pw, sr,
                                                       out3.data_ptr(), 0, ws.data_ptr(), ws.numel(), hip.stream_ptr())
        assert rc == 0
        d = (out3 - out).abs().max().item()
        assert d <= 1e-5 * max(1.0, float(out.abs().max())), d
        if sr == 0 and case not in ("fp16",):
            assert not torch.equal(out3, out)          # the fast path really ran (a different association moves some last bits)
    finally:
        hip.roi_align_set_exact(True)
    out4 = torch.full((n + 5, C, ph, pw), 9.0, device="cuda")
    rc = hip.lib().dtc_roi_align_forward_packed_ws(lvs, 1, ch, hip._dtype_code(dt), cu(desc).data_ptr(), n + 5, ph, pw, sr,
                                                   out4.data_ptr(), 0, ws.data_ptr(), ws.numel(), hip.stream_ptr())
    assert rc == 0 and torch.equal(out4, out)


_VARIANT_CHILD = r"""
import sys, os, numpy as np, torch
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import oracle as orc
from detectorch_amd import hip, synth
import test_hip_roi_align as T
for (C, ph, sr, R, seed) in [(32, 7, 2, 200, 5), (130, 7, 2, 60, 6), (24, 14, 2, 60, 8), (20, 7, 0, 60, 9), (256, 7, 2, 1600, 1007)]:
    feats, rois5, lv, ref = T._fpn_case(orc, R, C, ph, sr, seed, batch=2)
    for nhwc in (False, True):
        tf = [T.cu(f) for f in feats]
        if nhwc:
            tf = [t.contiguous(memory_format=torch.channels_last) for t in tf]
        out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, T.cu(rois5), ph, ph, sr, roi_levels=T.cu(lv)).cpu().numpy()
        assert np.array_equal(out, ref), (C, ph, sr, nhwc)
    h16 = [T.cu(f).half() for f in feats]
    ref16 = np.zeros_like(ref)
    for l in range(4):
        m = lv == l
        if m.any():
            ref16[m] = orc.roi_align_forward(h16[l].float().cpu().numpy(), rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], sr)
    out16 = hip.roi_align_forward(h16, synth.FPN_ROI_SCALES, T.cu(rois5), ph, ph, sr, roi_levels=T.cu(lv)).cpu().numpy()
    assert np.array_equal(out16, ref16), ("fp16", C, ph, sr)
    # the (image, level, row band, x) visiting order of dtc_fpn_collect_distribute: neighbours overlap, clusters merge
    yc, xc = (rois5[:, 2] + rois5[:, 4]) * 0.5, (rois5[:, 1] + rois5[:, 3]) * 0.5
    band = (yc / (4 * 2 ** lv.astype(np.float32) * 16)).astype(np.int32)
    order = np.lexsort((xc, band, lv, rois5[:, 0]))
    outs = hip.roi_align_forward([T.cu(f) for f in feats], synth.FPN_ROI_SCALES, T.cu(rois5[order]), ph, ph, sr,
                                 roi_levels=T.cu(lv[order])).cpu().numpy()
    assert np.array_equal(outs, ref[order]), ("visiting order", C, ph, sr)
print("ok")
"""


@pytest.mark.parametrize("env", ["DTC_ROIALIGN_TILE=0", "DTC_ROIALIGN_GENERAL=1", "DTC_ROIALIGN_TILE=0 DTC_RA_NO_CTS64=1",
                                 "DTC_ROIALIGN_MAP=0",
                                 "DTC_ROIALIGN_NO_NHWC_DIRECT=1", "DTC_RA_TILE_CHBLOCK=128", "DTC_RA_TILE_CHBLOCK=32",
                                 "DTC_RA_TILE_CBMAJOR=0", "DTC_RA_NO_XCD=1", "DTC_RA_MAP_PREP=0", "DTC_RA_MAP_PITCH=0", "DTC_RA_TILE_LDS16_KB=52"])
def test_kernel_variants_bit_exact_in_child_process(hip, oracle, env):
    """Every RoIAlign kernel that stays in the library -- the cluster-stationary default in its three workgroup shapes, the
    RoI-stationary LDS kernel with its stager options, the channels_last direct kernel, the per-output gather kernel -- does
    the same float32 arithmetic in the same order: bit-identical to the oracle, fp32 / fp16, NCHW / channels_last, channel
    tails, adaptive sampling, and the 1600 x 256 many-workgroup configuration.  The selecting knobs are resolved once per
    process, so each setting runs in a child process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for kv in env.split():
        k, v = kv.split("=")
        e[k] = v
    r = subprocess.run([sys.executable, "-c", _VARIANT_CHILD % root], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, env + "\n" + r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_bfloat16_features_and_output(hip, oracle, layout):
    """bf16 feature maps (what an MI355X bf16 backbone emits) / bf16 pooled output for the fc6 GEMM: fp32 accumulation, so
    bf16-in -> fp32-out equals the oracle on the up-cast maps bit for bit; bf16-out is that result rounded to nearest even."""
    feats, rois5, lv, _ = _fpn_case(oracle, 150, 72, 7, 2, 31, batch=2)
    tb = [cu(f).to(torch.bfloat16) for f in feats]
    up = [t.float().cpu().numpy() for t in tb]
    ref = np.zeros((150, 72, 7, 7), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(up[l], rois5[m], 7, 7, synth.FPN_ROI_SCALES[l], 2)
    if layout == "nhwc":
        tb = [t.contiguous(memory_format=torch.channels_last) for t in tb]
    out32 = hip.roi_align_forward(tb, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv))
    assert out32.dtype == torch.float32 and np.array_equal(out32.cpu().numpy(), ref)
    out16 = hip.roi_align_forward(tb, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv), out_dtype=torch.bfloat16)
    assert out16.dtype == torch.bfloat16
    assert torch.equal(out16.cpu(), torch.from_numpy(ref).to(torch.bfloat16))          # same round-to-nearest-even
    f32in = hip.roi_align_forward([cu(u) for u in up], synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv),
                                  out_dtype=torch.bfloat16)
    assert torch.equal(f32in.cpu(), out16.cpu())


# ---- the BENCHMARKED dispatch, against the oracle (round-1 VERDICT: no parity test reached these shapes) ---------------
@pytest.mark.parametrize("ph,R", [(7, 1600), (14, 320)])
def test_full_channel_count_multilevel_batched_vs_oracle(hip, oracle, ph, R):
    """C = 256 (the real FPN channel count), all four levels, B = 2, enough RoIs x channel blocks that the launch takes the
    many-workgroup / 128-channel-block configuration bench.py runs (>= 3072 workgroups for the round-1 kernel), both pooled
    sizes, for the default cluster-stationary kernel (the other kernels: test_kernel_variants_bit_exact_in_child_process).
    Bit-exact, in the plain order and in the (image, level, band, x) visiting order where clusters actually merge."""
    feats, rois5, lv, ref = _fpn_case(oracle, R, 256, ph, 2, 1000 + ph, batch=2)
    # visit in (image, level, y band, x) order like dtc_fpn_collect_distribute does, so clusters actually merge
    yc, xc = (rois5[:, 2] + rois5[:, 4]) * 0.5, (rois5[:, 1] + rois5[:, 3]) * 0.5
    band = (yc / (4 * 2 ** lv.astype(np.float32) * 16)).astype(np.int32)
    order = np.lexsort((xc, band, lv, rois5[:, 0])).astype(np.int32)
    for o in (None, order):
        out = hip.roi_align_forward([cu(f) for f in feats], synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv),
                                    roi_order=None if o is None else cu(o)).cpu().numpy()
        assert np.abs(out - ref).max() <= TOL
        assert np.array_equal(out, ref)


def test_tile_kernel_edge_cases(hip, oracle):
    """Cluster kernel corner cases: a RoI covering the whole map (window larger than the LDS image -> per-output gather),
    RoIs hanging over every border, degenerate (x2 < x1) boxes, identical RoIs (full overlap), RoIs alternating between
    images / levels (no merge), a RoI count that is not a multiple of the RoIs-per-workgroup, C not a multiple of 4."""
    rs = synth.rng(9, 1)
    shapes = synth.fpn_level_shapes()[:4]
    for C in (8, 6):
        feats = [synth.make_features(rs, (2, C, h, w)) - 0.3 for (h, w) in shapes]       # negative values too
        base = synth.make_rois(rs, 40, max_side=64)
        rois = np.vstack([base, base[:7], [[0, 0, 1343, 799]], [[-50, -50, 30, 30]], [[1300, 760, 1500, 900]],
                          [[100, 100, 90, 90]], [[5000, 5000, 6000, 6000]], [[0, 0, 0, 0]], [[1343, 799, 1343, 799]]]).astype(np.float32)
        R = rois.shape[0]
        bidx = (np.arange(R) % 2).astype(np.float32)[:, None]
        rois5 = np.hstack([bidx, rois])
        lv = (np.arange(R) % 4).astype(np.int32)
        lv[:40] = 0
        lv[47] = 0     # the whole-image RoI on the finest level: 200 x 336 window
        for ph in (7, 14):
            ref = np.zeros((R, C, ph, ph), np.float32)
            for l in range(4):
                m = lv == l
                ref[m] = oracle.roi_align_forward(feats[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], 2)
            out = hip.roi_align_forward([cu(f) for f in feats], synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv))
            assert np.array_equal(out.cpu().numpy(), ref), (C, ph)


def test_c4_true_shape_vs_reference_compiled(hip, oracle):
    """BASELINE cfg1/cfg2's real shape: features [1,1024,50,84], R = 1000 proposals, 14x14 bins, sampling_ratio 0, scale 1/16
    -- against the reference's own roi_align_cpu_loop.cpp compiled unmodified (oracle/_ref/libref_roialign.so) when it has
    travelled to this box, else against the oracle restatement (itself pinned to that .so in the CPU suite)."""
    import os
    import ref_harness
    rs = synth.rng(2, 11)
    feat = synth.make_features(rs, (1, 1024, 50, 84))
    rois5 = np.hstack([np.zeros((1000, 1), np.float32), synth.make_rois(rs, 1000)]).astype(np.float32)
    if os.path.exists(os.path.join(ref_harness.REF_BUILD, "libref_roialign.so")):
        ref = ref_harness.ref_roi_align(feat, rois5, 14, 14, 1 / 16., 0)
    else:
        ref = oracle.roi_align_forward(feat, rois5, 14, 14, 1 / 16., 0)
    out = hip.roi_align_forward(cu(feat), 1 / 16., cu(rois5), 14, 14, 0).cpu().numpy()
    assert np.abs(out - ref).max() <= TOL
    assert np.array_equal(out, ref)


# ---- channels_last maps: the LDS-DMA staged kernel (csrc/roi_align_nhwc.hip) -------------------------------------------------------
def _nhwc_case(oracle, C, seed, R=260):
    rs = synth.rng(11, seed)
    shapes = synth.fpn_level_shapes()[:4]
    feats = [synth.make_features(rs, (2, C, h, w)) - 0.3 for (h, w) in shapes]
    small = synth.make_rois(rs, R - 60, max_side=90.0)
    big = synth.make_rois(rs, 52, max_side=900.0)
    odd = np.array([[0, 0, 1343, 799], [-40, -30, 25, 20], [1300, 770, 1500, 900], [100, 100, 90, 90], [0, 0, 0, 0],
                    [1343, 799, 1343, 799], [0, 300, 1343, 330], [600, 0, 640, 799]], np.float32)
    rois = np.vstack([small, big, odd]).astype(np.float32)
    lv = (oracle.map_rois_to_fpn_levels(rois, 2, 5) - 2).astype(np.int32)
    lv[-8:] = [0, 0, 3, 1, 2, 3, 0, 0]           # the whole image / full-width / full-height boxes on the FINEST level: no strip fits
    rois5 = np.hstack([rs.randint(0, 2, (rois.shape[0], 1)).astype(np.float32), rois]).astype(np.float32)
    return feats, rois5, lv


@pytest.mark.parametrize("dtype,C,ph", [("f16", 96, 14), ("bf16", 64, 14), ("bf16", 160, 7), ("f16", 72, 14)])
def test_nhwc_16bit_direct_kernel_wide_lanes_vs_oracle(hip, oracle, dtype, C, ph):
    """channels_last 16-bit maps, sampling ratio 2: the direct-gather kernel with 8 channels per lane (16-byte tap loads) -- 64-channel
    blocks for 7x7 bins, 32-channel blocks for the 14x14 bins of the mask head; C = 160 / 72 leave a partial last block on the
    4-channel lanes (72 is not a multiple of 32 at 14x14: the LDS kernel takes it).  Small, large and border boxes on all four
    levels: bit-exact against the oracle on the up-cast maps; the 16-bit output is that result rounded once."""
    feats, rois5, lv = _nhwc_case(oracle, C, C + ph, R=120)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    tf = [cu(f).to(tdt).contiguous(memory_format=torch.channels_last) for f in feats]
    up = [t.float().contiguous().cpu().numpy() for t in tf]
    ref = np.zeros((rois5.shape[0], C, ph, ph), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(up[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], 2)
    out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv))
    assert out.dtype == torch.float32 and np.array_equal(out.cpu().numpy(), ref)
    out16 = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv), out_dtype=tdt)
    assert torch.equal(out16.cpu(), torch.from_numpy(ref).to(tdt))


@pytest.mark.parametrize("dtype,ph,far", [("f16", 7, False), ("bf16", 7, True), ("f16", 14, True), ("bf16", 14, False)])
def test_nhwc16_grouped_kernel_padding_rows_tail_groups_and_far_levels(hip, oracle, dtype, ph, far):
    """roi_align_nhwc16.hip (16-bit channels_last maps, the bins of up to four RoIs dealt over one workgroup): padding rows
    (level -1 -> zeros) inside and at the end of a group, a RoI count that leaves a partial last group, a permuted visiting order,
    and -- far -- levels carved from the two ends of one 5 GB allocation, so that the upper ones lie more than 4 GB above the
    launch's base address and their workgroups take the 64-bit address loop.  Bit-exact against the oracle on the up-cast maps."""
    C = 128
    feats, rois5, lv = _nhwc_case(oracle, C, 77 + ph, R=117)
    lv = lv.copy()
    lv[[3, 4, 5, 50, rois5.shape[0] - 1]] = -1
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    if far:
        big = torch.empty(int(2.6e9), dtype=tdt, device="cuda")          # 5.2 GB
        tf, pos = [], 0
        for i, f in enumerate(feats):
            n, c, h, w = f.shape
            cnt = n * c * h * w
            if i < 2:
                flat = big[pos:pos + cnt]; pos += (cnt + 63) // 64 * 64
            else:
                end = big.numel() - (i - 2) * 8 * 1024 * 1024
                flat = big[end - cnt:end]
            v = flat.view(n, h, w, c).permute(0, 3, 1, 2)
            v.copy_(cu(f).to(tdt))
            assert v.stride(1) == 1
            tf.append(v)
        assert tf[3].data_ptr() - tf[0].data_ptr() > (1 << 32)
    else:
        tf = [cu(f).to(tdt).contiguous(memory_format=torch.channels_last) for f in feats]
    up = [t.float().contiguous().cpu().numpy() for t in tf]
    ref = np.zeros((rois5.shape[0], C, ph, ph), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(up[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], 2)
    out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv))
    assert out.dtype == torch.float32 and np.array_equal(out.cpu().numpy(), ref)
    order = torch.randperm(rois5.shape[0], generator=torch.Generator().manual_seed(5)).to(torch.int32)
    out16 = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv), roi_order=order.cuda(), out_dtype=tdt)
    assert torch.equal(out16.cpu(), torch.from_numpy(ref).to(tdt))


@pytest.mark.parametrize("dtype,C", [("f32", 64), ("f32", 256), ("f16", 128), ("bf16", 256)])
def test_nhwc_lds_dma_kernel_vs_oracle(hip, oracle, dtype, C):
    """channels_last feature maps, 7x7 bins, sampling ratio 2: window staged with LDS-DMA, lane <-> channel chunk.  Small boxes
    (one strip), large ones (several strips of bin rows), boxes no strip fits (per-bin-row straight from global), every border case,
    the 42-column P5 map; float32 maps take the new kernel, 16-bit maps the direct-gather kernel by default (the new one in the
    child-process test below): bit-exact against the oracle on the up-cast maps, and the 16-bit outputs are that result rounded once."""
    feats, rois5, lv = _nhwc_case(oracle, C, C + len(dtype))
    tdt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    tf = [cu(f).to(tdt).contiguous(memory_format=torch.channels_last) for f in feats]
    up = [t.float().contiguous().cpu().numpy() for t in tf]
    ref = np.zeros((rois5.shape[0], C, 7, 7), np.float32)
    for l in range(4):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(up[l], rois5[m], 7, 7, synth.FPN_ROI_SCALES[l], 2)
    out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv))
    assert out.dtype == torch.float32 and np.array_equal(out.cpu().numpy(), ref)
    if dtype != "f32":
        o16 = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv), out_dtype=tdt)
        assert torch.equal(o16.cpu(), torch.from_numpy(ref).to(tdt))
    # the same maps in NCHW through the cluster-stationary kernel: the two layouts agree bit for bit
    nchw = hip.roi_align_forward([t.contiguous() for t in tf], synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv))
    assert torch.equal(nchw, out)


_NHWC_CHILD = r"""
import sys, os, numpy as np, torch
root = %r
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import oracle as orc
from detectorch_amd import hip, synth
import test_hip_roi_align as T
sixteen = (os.environ.get("DTC_RA_NHWC_LDS_16BIT") or os.environ.get("DTC_RA_NHWC_PIPE16")
           or ("DTC_RA_NHWC16" in os.environ and "DTC_RA_NHWC_DIRECT32" not in os.environ))
for ph in (7, 14):
    for tdt, C in ((torch.float32, 64),) + (((torch.float16, 128), (torch.bfloat16, 256)) if sixteen else ()):
        feats, rois5, lv = T._nhwc_case(orc, C, 5 + ph, R=260 if ph == 7 else 120)
        tf = [T.cu(f).to(tdt).contiguous(memory_format=torch.channels_last) for f in feats]
        up = [t.float().contiguous().cpu().numpy() for t in tf]
        ref = np.zeros((rois5.shape[0], C, ph, ph), np.float32)
        for l in range(4):
            m = lv == l
            if m.any():
                ref[m] = orc.roi_align_forward(up[l], rois5[m], ph, ph, synth.FPN_ROI_SCALES[l], 2)
        out = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, T.cu(rois5), ph, ph, 2, roi_levels=T.cu(lv)).cpu().numpy()
        assert np.array_equal(out, ref), (str(tdt), ph)
        # visiting order + padding rows through the packed-descriptor entry (the planner's scalar descriptor loads)
        order = torch.randperm(rois5.shape[0]).to(torch.int32)
        out2 = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, T.cu(rois5), ph, ph, 2, roi_levels=T.cu(lv), roi_order=order.cuda()).cpu().numpy()
        assert np.array_equal(out2, ref), ("ordered", str(tdt), ph)
        if tdt != torch.float32:
            o16 = hip.roi_align_forward(tf, synth.FPN_ROI_SCALES, T.cu(rois5), ph, ph, 2, roi_levels=T.cu(lv), out_dtype=tdt)
            assert torch.equal(o16.cpu(), torch.from_numpy(ref).to(tdt))
print("ok")
"""


_D0 = "DTC_RA_NHWC_DIRECT32=0 DTC_RA_NHWC16=0 "        # float32 and 16-bit maps past the grouped direct kernel: the kernels of roi_align_nhwc.hip / roi_align.hip


@pytest.mark.parametrize("env", [_D0 + "DTC_RA_NHWC_LDS_KB=24", _D0 + "DTC_RA_NHWC_LDS_KB=78", _D0 + "DTC_RA_NHWC_LDS_KB=156", _D0 + "DTC_RA_NHWC_LDS=0",
                                 _D0 + "DTC_RA_NHWC_LDS_16BIT=1", _D0 + "DTC_RA_NHWC_LDS_16BIT=1 DTC_RA_NHWC_LDS_KB=24",
                                 _D0 + "DTC_RA_NHWC_PIPE=2", _D0 + "DTC_RA_NHWC_PIPE=2 DTC_RA_NHWC_LDS_KB=30", _D0 + "DTC_RA_NHWC_PIPE=2 DTC_RA_NHWC_LDS_KB=156",
                                 _D0 + "DTC_RA_NHWC_PIPE=2 DTC_RA_NHWC_PIPE16=1", _D0 + "DTC_RA_NHWC_PIPE=2 DTC_RA_NHWC_PIPE16=1 DTC_RA_NHWC_LDS_KB=40",
                                 _D0 + "DTC_RA_NHWC_PIPE=0", "DTC_RA_NHWC16=0", "DTC_RA_NHWC_DIRECT32=0", "DTC_RA_NHWC16=1"])
def test_nhwc_lds_image_sizes_in_child_process(hip, oracle, env):
    """The LDS image size decides how many strips a window takes (24 KB: nearly every RoI in several strips or straight from
    global; 156 KB: one workgroup per CU, one strip) -- and must not change a bit; DTC_RA_NHWC_LDS=0 is the direct-gather kernel.
    DTC_RA_NHWC_PIPE=2: the pipelined kernel (round 4) for 7 x 7 bins too (by default it takes the 14 x 14 launches only), at
    image sizes from a handful of pixels to one workgroup per CU, float32 and 16-bit maps; =0: the round-3 kernels everywhere.
    DTC_RA_NHWC16=0: 16-bit maps through the one-RoI-per-workgroup direct kernel instead of the grouped one (roi_align_nhwc16.hip).
    DTC_RA_NHWC_DIRECT32=0: float32 maps with <= 64 bins on the LDS-DMA kernels instead of the grouped direct kernel (4-channel lanes);
    the LDS-kernel settings above carry both switches so that they reach the kernels they size.  DTC_RA_NHWC16=1: the defaults, 16-bit maps included.
    Every case: 7 x 7 and 14 x 14 bins, identity and permuted visiting order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for kv in env.split():
        k, v = kv.split("=")
        e[k] = v
    r = subprocess.run([sys.executable, "-c", _NHWC_CHILD % root], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, env + "\n" + r.stdout[-1500:] + r.stderr[-3000:]


# ---- REAL-SHAPE dispatches of the round-4 / round-5 kernels: B = 8 images, C = 256, 8000 / 16 000 RoIs ----------
def _oracle_levels(oracle, maps, rois5, lv, ph, sr=2):
    """Oracle RoIAlign of a multi-level launch: `maps` float32 numpy [B,C,H,W] per level, level id -1 (padding row) -> zeros."""
    ref = np.zeros((rois5.shape[0], maps[0].shape[1], ph, ph), np.float32)
    for l in range(len(maps)):
        m = lv == l
        if m.any():
            ref[m] = oracle.roi_align_forward(maps[l], np.ascontiguousarray(rois5[m]), ph, ph, synth.FPN_ROI_SCALES[l], sr)
    return ref


def _bench_population(B, T, C, dev, feat_dtype, channels_last, seed):
    """The bench's OWN RoIs: the fused path (GenerateProposals -> NMS -> collect / distribute) on the bench's synthetic batch; returns
    the path (rois5, level ids, visiting order, packed descriptors, box_feats of the real launch) and the feature maps."""
    from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
    path = FpnRegionPath(B, dev, channels=C, collect_top_n=T, feat_dtype=feat_dtype, max_out=104)
    inputs = synthetic_batch(B, dev, seed=seed, channels=C, top_n=T, feat_dtype=feat_dtype, channels_last=channels_last, max_out=104)
    path.bind(*inputs)
    path.step(use_graph=False)
    torch.cuda.synchronize()
    return path, inputs[2]


@pytest.mark.parametrize("population", ["bench", "harder"])
@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
def test_real_shape_box_head_fp32_c256_8000_rois(hip, oracle, layout, population):
    """The float32 box head at the bench's real shape -- 8 images x 1000 RoIs, C = 256, 7 x 7 bins, sampling ratio 2 -- on
    channels_last maps (roi_align_fwd_nhwc16<float>, G = 2) AND NCHW maps (the cluster kernel), on BOTH RoI populations bench.py times:
    the path's own proposals (seed 3000) and the log-uniform 16-600 px `harder_set`.  All 8000 RoIs bit-equal to the oracle."""
    dev = torch.device("cuda", 0)
    B, T, C = 8, 1000, 256
    path, feats = _bench_population(B, T, C, dev, torch.float32, layout == "nhwc", 3000)
    maps = [f.float().cpu().numpy() for f in feats]          # .cpu().numpy() of a channels_last tensor keeps the logical NCHW index order
    maps = [np.ascontiguousarray(m) for m in maps]
    if population == "bench":
        rois5 = path.rois5.reshape(-1, 5).cpu().numpy()
        lv = path.roi_levels.reshape(-1).cpu().numpy()
        got = path.box_feats.cpu().numpy()                   # the launch of the fused path itself (packed descriptors, restored order)
        assert int(path.n_rois.min()) > 900
    else:
        rois5, lv, order = synth.harder_roi_set(B, T)
        assert len(np.unique(lv)) == 4
        got = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, cu(rois5), 7, 7, 2, roi_levels=cu(lv), roi_order=cu(order)).cpu().numpy()
    ref = _oracle_levels(oracle, maps, rois5, lv, 7)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("layout", ["nhwc", "nchw"])
@pytest.mark.parametrize("population", ["bench", "harder"])
def test_real_shape_bf16_maps_and_output_c256_8000_rois(hip, oracle, layout, population):
    """bf16 maps -> bf16 pooled features at the real shape (8 x 1000 RoIs, C = 256): float32 accumulation of the up-cast maps equals the
    oracle bit for bit; the bf16 output (v_cvt_pk_bf16_f32 in the kernels' store path) equals torch's round-to-nearest-even of it."""
    dev = torch.device("cuda", 0)
    B, T, C = 8, 1000, 256
    path, feats32 = _bench_population(B, T, C, dev, torch.float32, False, 3000)
    feats = [f.to(torch.bfloat16) for f in feats32]
    maps = [np.ascontiguousarray(f.float().cpu().numpy()) for f in feats]
    if layout == "nhwc":
        feats = [f.contiguous(memory_format=torch.channels_last) for f in feats]
    if population == "bench":
        rois5, lv, order = path.rois5.reshape(-1, 5), path.roi_levels.reshape(-1), path.roi_order.reshape(-1)
        rois5_np, lv_np = rois5.cpu().numpy(), lv.cpu().numpy()
    else:
        rois5_np, lv_np, order_np = synth.harder_roi_set(B, T)
        rois5, lv, order = cu(rois5_np), cu(lv_np), cu(order_np)
    ref = _oracle_levels(oracle, maps, rois5_np, lv_np, 7)
    out32 = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois5, 7, 7, 2, roi_levels=lv, roi_order=order)
    assert np.array_equal(out32.cpu().numpy(), ref)
    out16 = hip.roi_align_forward(feats, synth.FPN_ROI_SCALES, rois5, 7, 7, 2, roi_levels=lv, roi_order=order, out_dtype=torch.bfloat16)
    assert torch.equal(out16.cpu(), torch.from_numpy(ref).to(torch.bfloat16))


def test_bf16_output_rounding_special_values(hip, oracle):
    """the bf16 store is an inline v_cvt_pk_bf16_f32.  Channels that are CONSTANT maps of special float32 values, pooled by
    RoIs whose samples fall on pixel centres (weights exactly (1, 0, 0, 0), 4 v / 4 == v): ties to even in both directions, the largest
    value below a tie, denormals, -0, values that round up into the next binade, NaN.  bf16 output == torch's .to(bfloat16) of the
    oracle's float32 result (NaN compared as NaN)."""
    bits = [0x3F808000, 0x3F818000, 0x3F807FFF, 0x3F808001, 0x3FFF8000, 0x3FFFFFFF, 0x00000001, 0x00008000, 0x00018000, 0x007FFFFF,
            0x80000000, 0x80008000, 0xBF808000, 0x7E7F8000, 0x7E7FFFFF, 0x00800000, 0x7FC00001, 0x3F800000, 0x477FE000, 0x0000FFFF]
    vals = np.array(bits, np.uint32).view(np.float32)
    C, H, W = len(bits), 24, 24
    f = np.broadcast_to(vals[None, :, None, None], (1, C, H, W)).astype(np.float32).copy()
    # 7 x 7 bins over 14 px, sampling ratio 2: sample k of an axis sits at start + 0.5 + k -> integer coordinates for start = n + 0.5
    rois5 = np.array([[0, 3.5, 2.5, 17.5, 16.5], [0, 1.5, 4.5, 15.5, 18.5]], np.float32)
    ref = oracle.roi_align_forward(f, rois5, 7, 7, 1.0, 2)
    same = ~np.isnan(vals) & (vals.view(np.uint32) != 0x80000000)      # (the accumulator starts at +0: 0 + (-0) = +0)
    assert np.array_equal(ref[:, same].view(np.uint32), np.broadcast_to(vals[same].view(np.uint32)[None, :, None, None], (2, int(same.sum()), 7, 7)))
    want = torch.from_numpy(ref).to(torch.bfloat16)
    for layout in ("nchw", "nhwc"):
        t = cu(f)
        if layout == "nhwc":
            t = t.contiguous(memory_format=torch.channels_last)
        got = hip.roi_align_forward(t, 1.0, cu(rois5), 7, 7, 2, out_dtype=torch.bfloat16).cpu()
        nan = torch.isnan(want)
        assert torch.equal(torch.isnan(got), nan)
        assert torch.equal(got.view(torch.int16)[~nan], want.view(torch.int16)[~nan])


@pytest.mark.parametrize("out_dtype", ["f32", "f16", "bf16"])
def test_cluster_kernel_window_shapes(hip, oracle, out_dtype):
    """Window shapes of the cluster kernel against the oracle, bit-exact (written for the round-5 restructured kernel, commit 1e29752,
    kept for the shipped one): wide and tall single windows, a box over most of the coarsest map, a box over the whole finest map
    (per-output gather), windows on the right / bottom edge of every level (the clamped taps of roi_align_cpu_loop.cpp:78-90; P5's 42
    columns = unaligned pieces), clusters that merge and clusters that cannot, 7 x 7 and 14 x 14 bins, float32 / fp16 / bf16 output."""
    rs = synth.rng(11, 5)
    shapes = synth.fpn_level_shapes()[:4]
    C = 16
    feats = [synth.make_features(rs, (2, C, h, w)) - 0.25 for (h, w) in shapes]
    W, H = 1344.0, 800.0
    rows = []
    for l, s in enumerate((4.0, 8.0, 16.0, 32.0)):
        rows += [[l, 0, 0, W - 1, H - 1]]                                        # the whole map of the level
        rows += [[l, W - 1 - 30 * s, H - 1 - 12 * s, W - 1, H - 1]]              # bottom-right corner: duplicate column AND row
        rows += [[l, W - 1 - 20 * s, 3 * s, W + 40, 15 * s]]                     # hangs over the right edge
        rows += [[l, 5 * s, H - 1 - 9 * s, 17 * s, H + 100]]                     # hangs over the bottom edge
        rows += [[l, 2 * s, 2 * s, 2 * s + 35 * s, 2 * s + 20 * s]]              # 35 x 20 px window: 3 x 6 = 18 blocks -> six units
        rows += [[l, 1 * s, 1 * s, 1 * s + 41 * s, 1 * s + 23.5 * s]]            # ~ 43 x 26: more than one image on 7 x 7 -> big path
        for k in range(12):                                                      # neighbours that merge into clusters
            rows += [[l, (10 + 3 * k) * s, 6 * s, (22 + 3 * k) * s, 17 * s]]
    rows = np.array(rows, np.float32)
    lv = rows[:, 0].astype(np.int32)
    R = rows.shape[0]
    rois5 = np.hstack([(np.arange(R) % 2).astype(np.float32)[:, None], rows[:, 1:]])
    # clusters only form between consecutive RoIs of one image: visit image 0's rows, then image 1's
    order = np.concatenate([np.arange(0, R, 2), np.arange(1, R, 2)]).astype(np.int32)
    odt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[out_dtype]
    for ph in (7, 14):
        ref = _oracle_levels(oracle, feats, rois5, lv, ph)
        for od in (None, order):
            out = hip.roi_align_forward([cu(f) for f in feats], synth.FPN_ROI_SCALES, cu(rois5), ph, ph, 2, roi_levels=cu(lv),
                                        roi_order=None if od is None else cu(od), out_dtype=odt)
            if out_dtype == "f32":
                assert np.array_equal(out.cpu().numpy(), ref), ph
            else:
                assert torch.equal(out.cpu(), torch.from_numpy(ref).to(odt)), ph
