"""A2-A4 (+A5) parity: HIP GenerateProposals vs the golden vectors produced by the reference's own Python, and vs the
oracle at full BASELINE sizes.  Survivor identity/order (scores are unique keys) must match exactly; box coordinates
within 1e-4 / 1 ulp of the reference (its exp is torch-CPU), and BIT-EXACT vs the oracle.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden, ulp_close
from detectorch_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from detectorch_amd import hip as h
    h.lib()
    return h


@pytest.mark.parametrize("tag", ["c4", "p3", "p6"])
def test_golden_module_surface(hip, tag):
    from detectorch_amd.model.generate_proposals import GenerateProposals
    g = golden("generate_proposals")
    cfg = g[tag + "_cfg"]
    ss, pre, post, im_h, im_w = cfg[3], int(cfg[4]), int(cfg[5]), cfg[6], cfg[7]
    gp = GenerateProposals(spatial_scale=ss, anchor_sizes=tuple(cfg[9:]), rpn_pre_nms_top_n=pre, rpn_post_nms_top_n=post)
    props, scores = gp(torch.from_numpy(g[tag + "_cls"]).cuda(), torch.from_numpy(g[tag + "_bbox"]).cuda(), im_h, im_w, 1.6)
    assert props.is_cuda and scores.is_cuda and scores.dim() == 2 and scores.shape[1] == 1
    assert np.array_equal(scores.cpu().numpy(), g[tag + "_scores"])
    assert ulp_close(props.cpu().numpy(), g[tag + "_props"])


def _oracle_levels(oracle, cls, bbox, sizes, strides, im_h, im_w, pre, post, thr):
    out = []
    for l in range(len(cls)):
        anchors = oracle.generate_anchors(strides[l], (sizes[l],) if np.isscalar(sizes[l]) else sizes[l], (0.5, 1, 2))
        per_img = []
        for b in range(cls[l].shape[0]):
            per_img.append(oracle.generate_proposals(cls[l][b], bbox[l][b], anchors, strides[l], im_h, im_w, pre[l], post,
                                                     thr, return_pre_nms=True))
        out.append(per_img)
    return out


def test_fpn_batched_full_size_vs_oracle(hip, oracle):
    # BASELINE cfg3: 5 FPN levels of a 800x1344 input, pre 1000 / post 1000 per level, 2 images in one call
    from detectorch_amd.utils.generate_anchors import generate_anchors
    shapes = synth.fpn_level_shapes()
    B = 2
    cls, bbox = [], []
    for l, (h, w) in enumerate(shapes):
        cs, bs = [], []
        for b in range(B):
            c, d = synth.make_rpn_outputs(synth.rng(3, 10 * l + b), 3, h, w)
            cs.append(c); bs.append(d)
        cls.append(np.concatenate(cs)); bbox.append(np.concatenate(bs))
    strides = [float(s) for s in synth.FPN_STRIDES]
    sizes = [32.0 * 2 ** l for l in range(5)]
    anchors = [generate_anchors(stride=strides[l], sizes=(sizes[l],), aspect_ratios=(0.5, 1, 2)) for l in range(5)]
    boxes, scores, counts, pb, ps, pc = hip.generate_proposals(
        [torch.from_numpy(c).cuda() for c in cls], [torch.from_numpy(d).cuda() for d in bbox], anchors, strides,
        synth.FPN_PAD_H, synth.FPN_PAD_W, [1000] * 5, 1000, 0.7)
    ref = _oracle_levels(oracle, cls, bbox, sizes, strides, synth.FPN_PAD_H, synth.FPN_PAD_W, [1000] * 5, 1000, 0.7)
    boxes, scores, counts = boxes.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy()
    pb, ps, pc = pb.cpu().numpy(), ps.cpu().numpy(), pc.cpu().numpy()
    for l in range(5):
        for b in range(B):
            rb, rs, rpb, rps = ref[l][b]
            s = b * 5 + l
            assert pc[s] == rps.shape[0]
            assert np.array_equal(ps[s, :pc[s]], rps)            # pre-NMS: same top-k, same order
            assert np.array_equal(pb[s, :pc[s]], rpb)            # decode/clip bit-exact vs the oracle
            assert counts[b, l] == rs.shape[0]
            assert np.array_equal(scores[b, l, :counts[b, l]], rs)
            assert np.array_equal(boxes[b, l, :counts[b, l]], rb)


def test_c4_full_size_vs_oracle(hip, oracle):
    # BASELINE cfg2: A=15 on 50x84 (63 000 anchors), pre-NMS 6000, post 1000, thresh 0.7
    from detectorch_amd.model.generate_proposals import GenerateProposals
    c, d = synth.make_rpn_outputs(synth.rng(2, 0), 15, 50, 84)
    gp = GenerateProposals()
    props, scores = gp(torch.from_numpy(c).cuda(), torch.from_numpy(d).cuda(), 800, 1333, 1.6)
    rb, rs = oracle.generate_proposals(c[0], d[0], oracle.generate_anchors(16.0), 16.0, 800, 1333, 6000, 1000, 0.7)
    assert np.array_equal(scores.cpu().numpy().reshape(-1), rs)
    assert np.array_equal(props.cpu().numpy(), rb)


def test_ties_and_constant_maps(hip, oracle):
    # (1) heavy exact ties (quantised scores) (2) a constant score map: every element ties at the threshold -> the
    # canonical rule (lowest (h,w,a) index first) must hold, including through the massive-tie fallback path.
    from detectorch_amd.utils.generate_anchors import generate_anchors
    rs = synth.rng(3, 777)
    A, H, W = 3, 50, 84
    c, d = synth.make_rpn_outputs(rs, A, H, W, tie_free=False)
    quant = (np.round(c * 16) / 16).astype(np.float32)
    const = np.full_like(c, 0.25)
    anchors = generate_anchors(stride=16.0, sizes=(128.0,), aspect_ratios=(0.5, 1, 2))
    for sc in (quant, const):
        _, _, _, pb, ps, pc = hip.generate_proposals([torch.from_numpy(sc).cuda()], [torch.from_numpy(d).cuda()], [anchors],
                                                     [16.0], 800, 1344, [1000], 1000, 0.7)
        _, _, rpb, rps = oracle.generate_proposals(sc[0], d[0], anchors, 16.0, 800, 1344, 1000, 1000, 0.7,
                                                   return_pre_nms=True)
        n = int(pc.cpu().numpy()[0])
        assert n == rps.shape[0]
        assert np.array_equal(ps.cpu().numpy()[0, :n], rps)
        assert np.array_equal(pb.cpu().numpy()[0, :n], rpb)


def test_sortedness_property(hip):
    # size-independent property: pre-NMS scores are non-increasing and every box is inside the image
    from detectorch_amd.utils.generate_anchors import generate_anchors
    c, d = synth.make_rpn_outputs(synth.rng(3, 5), 3, 200, 336)
    anchors = generate_anchors(stride=4.0, sizes=(32.0,), aspect_ratios=(0.5, 1, 2))
    _, _, _, pb, ps, pc = hip.generate_proposals([torch.from_numpy(c).cuda()], [torch.from_numpy(d).cuda()], [anchors], [4.0],
                                                 800, 1344, [1000], 1000, 0.7)
    n = int(pc[0].item())
    s = ps[0, :n]
    assert n == 1000 and bool((s[:-1] >= s[1:]).all())
    b = pb[0, :n]
    assert float(b.min()) >= 0 and float(b[:, 2].max()) <= 1343 and float(b[:, 3].max()) <= 799
    top = torch.topk(torch.from_numpy(c).reshape(-1), 1000).values
    assert torch.equal(s.cpu(), top)


def test_logit_input_equals_materialised_sigmoid(hip, oracle):
    """the RPN head's sigmoid (detector.py:125) folded into the top-k.  Passing LOGITS must give exactly what
    the reference flow gives for probabilities = sigmoid(logits): same top-k, same order (equal probabilities tie-break by
    index even when they come from different logits), same scores, same boxes, same NMS survivors.  Cases: ordinary
    logits, saturated logits (thousands of logits collapse onto 1.0f and its neighbours), quantised logits (exact
    ties), and a level smaller than pre_nms_top_n (take-all path)."""
    from detectorch_amd.utils.generate_anchors import generate_anchors
    rs = synth.rng(3, 4242)
    cases = []
    for (A, H, W, stride, size, mu, sd, quant) in [(3, 200, 336, 4.0, 32.0, -2.0, 2.0, 0), (3, 100, 168, 8.0, 64.0, 14.0, 4.0, 0),
                                                   (3, 50, 84, 16.0, 128.0, 9.0, 3.0, 8), (3, 13, 21, 64.0, 512.0, 0.0, 3.0, 0),
                                                   (15, 50, 84, 16.0, (32, 64, 128, 256, 512), 1.0, 5.0, 0)]:
        lg = (rs.standard_normal((2, A, H, W)) * sd + mu).astype(np.float32)
        if quant:
            lg = (np.round(lg * quant) / quant).astype(np.float32)
        d = (rs.standard_normal((2, 4 * A, H, W)) * 0.2).astype(np.float32)
        cases.append((lg, d, stride, size))
    for (lg, d, stride, size) in cases:
        sizes = size if isinstance(size, tuple) else (size,)
        anchors = generate_anchors(stride=stride, sizes=sizes, aspect_ratios=(0.5, 1, 2))
        pre = 6000 if len(sizes) > 1 else 1000
        prob = oracle.rpn_sigmoid(lg)
        tl, td = torch.from_numpy(lg).cuda(), torch.from_numpy(d).cuda()
        fused = hip.generate_proposals([tl], [td], [anchors], [stride], 800, 1344, [pre], 1000, 0.7, scores_are_logits=True)
        plain = hip.generate_proposals([torch.from_numpy(prob).cuda()], [td], [anchors], [stride], 800, 1344, [pre], 1000, 0.7)
        fb, fs, fc, fpb, fps, fpc = [x.cpu().numpy() for x in fused]
        pb_, ps_, pc_, ppb, pps, ppc = [x.cpu().numpy() for x in plain]
        assert np.array_equal(fc, pc_) and np.array_equal(fpc, ppc)
        for b in range(2):
            n, m = int(fpc[b]), int(fc[b, 0])
            assert np.array_equal(fps[b, :n], pps[b, :n]) and np.array_equal(fpb[b, :n], ppb[b, :n])
            assert np.array_equal(fs[b, 0, :m], ps_[b, 0, :m]) and np.array_equal(fb[b, 0, :m], pb_[b, 0, :m])
            # and against the CPU oracle run on the materialised probabilities
            oa = oracle.generate_anchors(stride, sizes, (0.5, 1, 2))
            rb, rsc, rpb, rps = oracle.generate_proposals(prob[b], d[b], oa, stride, 800, 1344, pre, 1000, 0.7, return_pre_nms=True)
            assert n == rps.shape[0] and np.array_equal(fps[b, :n], rps) and np.array_equal(fpb[b, :n], rpb)
            assert m == rsc.shape[0] and np.array_equal(fs[b, 0, :m], rsc) and np.array_equal(fb[b, 0, :m], rb)


# ---- long segments: the LSD radix sort of rpn_sort (more than 2048 keys to order), with TIED scores ---------------------------
# (round-5 review: every test that reached block_radix_sort_u64 used tie-free scores, for which the index digits never decide an
# ordering.)  generate_proposals.py:77-93 with pre_nms_top_n = 6000: pre-NMS boxes, scores and counts bit-equal to the oracle.
def _c4_tie_maps(rs, A, H, W):
    c, d = synth.make_rpn_outputs(rs, A, H, W, tie_free=False)
    sparse = np.zeros_like(c)                                   # fewer than 6000 elements above the floor value: the rest of the
    pick = rs.permutation(c.size)[:3500]                        # top-6000 are ties AT the floor, taken in index order
    sparse.reshape(-1)[pick] = c.reshape(-1)[pick]
    two = np.where(c > np.median(c), np.float32(0.75), np.float32(0.25)).astype(np.float32)
    return d, [("q16", (np.round(c * 16) / 16).astype(np.float32)),
               ("q4096", (np.round(c * 4096) / 4096).astype(np.float32)),
               ("const", np.full_like(c, 0.25)),
               ("saturated", np.ones_like(c)),
               ("sparse", sparse),
               ("two_values", two)]


def _check_pre_nms(hip, oracle, sc, d, anchors, stride, pre, im_h=800, im_w=1333):
    _, _, _, pb, ps, pc = hip.generate_proposals([torch.from_numpy(sc).cuda()], [torch.from_numpy(d).cuda()], [anchors],
                                                 [stride], im_h, im_w, [pre], 1000, 0.7)
    _, _, rpb, rps = oracle.generate_proposals(sc[0], d[0], anchors, stride, im_h, im_w, pre, 1000, 0.7, return_pre_nms=True)
    n = int(pc.cpu().numpy()[0])
    assert n == rps.shape[0]
    assert np.array_equal(ps.cpu().numpy()[0, :n], rps)
    assert np.array_equal(pb.cpu().numpy()[0, :n], rpb)
    return n


@pytest.mark.parametrize("pre", [6000, 3000, 8192, 10000])
def test_c4_size_ties_reach_radix_sort(hip, oracle, pre):
    # A = 15 on 50 x 84; pre = 6000 / 3000 / 8192 -> the radix branch (sort_cap 8192 / 4096 / 8192), 10000 -> the 16-keys-per-thread
    # bitonic branch; "const" / "saturated" / "two_values" also go through the in-workgroup radix select that precedes the sort
    from detectorch_amd.utils.generate_anchors import generate_anchors
    anchors = generate_anchors(stride=16.0)
    d, maps = _c4_tie_maps(synth.rng(2, 4100 + pre), 15, 50, 84)
    for name, sc in maps:
        n = _check_pre_nms(hip, oracle, sc, d, anchors, 16.0, pre)
        assert n > 2048, name


@pytest.mark.parametrize("A,H,W", [(16, 64, 64), (1, 257, 256), (16, 64, 65)])
def test_radix_sort_index_digit_boundary(hip, oracle, A, H, W):
    # idx_bits = bits of N - 1 decides which index digits the sort visits: N = 65 536 (16 bits: two index digits), 65 792 and
    # 66 560 (17 bits: three).  Quantised scores -> thousands of exact ties ordered by the index digits alone.
    from detectorch_amd.utils.generate_anchors import generate_anchors
    sizes = (32.0,) if A == 1 else (32, 64, 128, 256)
    ratios = (1.0,) if A == 1 else (0.5, 1, 2, 4)
    anchors = generate_anchors(stride=16.0, sizes=sizes, aspect_ratios=ratios)
    assert anchors.shape[0] == A
    rs = synth.rng(2, 5200 + W)
    c, d = synth.make_rpn_outputs(rs, A, H, W, tie_free=False)
    for q in (8, 512):
        sc = (np.round(c * q) / q).astype(np.float32)
        assert _check_pre_nms(hip, oracle, sc, d, anchors, 16.0, 6000, im_h=16 * H, im_w=16 * W) > 2048
    assert _check_pre_nms(hip, oracle, np.full_like(c, 0.5), d, anchors, 16.0, 6000, im_h=16 * H, im_w=16 * W) > 2048


def test_radix_sort_24bit_index(hip, oracle):
    # N = 2^24 + 4096 anchors (25 index bits: all four index digits in use); two score values
    from detectorch_amd.utils.generate_anchors import generate_anchors
    A, H, W = 1, 4097, 4096
    anchors = generate_anchors(stride=4.0, sizes=(32.0,), aspect_ratios=(1.0,))
    rs = synth.rng(2, 6300)
    sc = np.full((1, A, H, W), 0.25, np.float32)
    hot = rs.permutation(A * H * W)[:2500]
    sc.reshape(-1)[hot] = 0.75
    d = (rs.standard_normal((1, 4 * A, H, W)).astype(np.float32) * np.float32(0.2))
    assert _check_pre_nms(hip, oracle, sc, d, anchors, 4.0, 6000, im_h=4 * H, im_w=4 * W) > 2048


def test_c4_path_graph_replay_with_ties(hip, oracle):
    # the same tie cases through C4RegionPath's captured hipGraph, replayed twice with different score maps in the bound tensors
    from detectorch_amd.pipeline import C4RegionPath, synthetic_c4_batch
    dev = torch.device("cuda", 0)
    B = 2
    path = C4RegionPath(B, dev, channels=8)
    inputs = list(synthetic_c4_batch(B, dev, seed=2600, channels=8))
    path.bind(*inputs)
    anchors = oracle.generate_anchors(16.0)
    for q in (16, 4096, 0):
        c = inputs[0]
        c.copy_(torch.round(c * q) / q if q else torch.full_like(c, 0.25))
        path.step(use_graph=True)
        torch.cuda.synchronize()
        sc, d = inputs[0].cpu().numpy(), inputs[1].cpu().numpy()
        pb, ps, pc = path.pre_boxes.cpu().numpy(), path.pre_scores.cpu().numpy(), path.pre_counts.cpu().numpy()
        kc = path.keep_cnt.cpu().numpy()
        for b in range(B):
            rb, rsc, rpb, rps = oracle.generate_proposals(sc[b], d[b], anchors, 16.0, path.im_h, path.im_w, 6000, 1000, 0.7,
                                                          return_pre_nms=True)
            n = int(pc[b])
            assert n == rps.shape[0] and n > 2048
            assert np.array_equal(ps[b, :n], rps) and np.array_equal(pb[b, :n], rpb)
            m = int(kc[b])
            assert m == rb.shape[0]
            assert np.array_equal(path.prop_boxes[b, :m].cpu().numpy(), rb)
            assert np.array_equal(path.prop_scores[b, :m].cpu().numpy(), rsc)
