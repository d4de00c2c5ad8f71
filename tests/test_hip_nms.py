"""A5 parity: HIP NMS keep indices must equal the oracle / golden vectors BIT-EXACTLY.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden
from detectorch_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from detectorch_amd import hip as h
    h.lib()
    return h


def _dets(seed, n, **kw):
    rs = synth.rng(7, seed)
    b = synth.make_rois(rs, n, **kw)
    s = synth.dedupe_scores(rs.uniform(0.0, 1.0, n).astype(np.float32))
    return np.ascontiguousarray(np.hstack([b, s[:, None]]), np.float32)


@pytest.mark.parametrize("t", [0.3, 0.5, 0.7])
def test_golden_keep_indices(hip, t):
    from detectorch_amd.utils import boxes as box_utils
    g = golden("nms")
    keep = box_utils.nms(g["dets"], t)
    assert keep.dtype == np.int64
    assert np.array_equal(keep, g["keep_%02d" % int(t * 10)])


def test_empty_and_single(hip):
    from detectorch_amd.utils import boxes as box_utils
    assert box_utils.nms(np.zeros((0, 5), np.float32), 0.5) == []
    assert np.array_equal(box_utils.nms(np.array([[1, 2, 30, 40, 0.9]], np.float32), 0.5), [0])


@pytest.mark.parametrize("n,thr", [(2, 0.5), (63, 0.5), (64, 0.7), (65, 0.3), (129, 0.5), (1000, 0.7), (2000, 0.5),
                                   (6000, 0.7), (4097, 0.7)])
def test_vs_oracle_sizes(hip, oracle, n, thr):
    d = _dets(n, n)
    keep = hip.nms(torch.from_numpy(d).cuda(), thr).cpu().numpy()
    assert np.array_equal(keep, oracle.nms(d, thr))


@pytest.mark.parametrize("n", [127, 128, 200, 255, 256, 257, 300, 511, 512, 513, 777, 1023, 1024, 1025, 1500, 2047, 2048, 2049])
def test_sort_size_sweep_through_the_drop_in(hip, oracle, n):
    """dtc_nms = segment sort (score desc, index asc) + mask + reduce + a second sort of the kept indices.  Sizes on both sides of every
    branch of block_sort.h (round 6): <= 128 keys and more than one key per thread on the bitonic network with DPP / ds_swizzle steps,
    256 <= n_pow2 <= 1024 on the wave-local sort + merge-path rounds; runs of EQUAL scores (quantised to 1/32) so that the index half of
    the key decides -- keep indices bit-equal to the oracle."""
    d = _dets(9000 + n, n)
    d[:, 4] = np.round(d[:, 4] * 32) / 32
    keep = hip.nms(torch.from_numpy(d).cuda(), 0.5).cpu().numpy()
    assert np.array_equal(keep, oracle.nms(d, 0.5))


def test_heavy_overlap_and_ties(hip, oracle):
    # dense cluster: long suppression chains inside one 64-row block
    rs = synth.rng(7, 1234)
    base = np.array([100, 100, 300, 260], np.float32)
    b = base + rs.uniform(-12, 12, (500, 4)).astype(np.float32)
    s = synth.dedupe_scores(rs.uniform(0, 1, 500).astype(np.float32))
    d = np.hstack([b, s[:, None]]).astype(np.float32)
    for thr in (0.5, 0.7, 0.9):
        assert np.array_equal(hip.nms(torch.from_numpy(d).cuda(), thr).cpu().numpy(), oracle.nms(d, thr))
    # exact score ties: canonical rule (score desc, index asc) -- same rule in oracle and HIP
    d[:, 4] = np.float32(0.5)
    d[::7, 4] = np.float32(0.75)
    assert np.array_equal(hip.nms(torch.from_numpy(d).cuda(), 0.6).cpu().numpy(), oracle.nms(d, 0.6))


def test_quotient_on_the_threshold_and_degenerate_boxes(hip, oracle):
    """The mask kernel decides `inter / union >= thresh` from the sign of inter - thresh * union and divides only inside a
    2^-21 band around zero (and for union <= 0 / thresh <= 0): pairs whose quotient equals the threshold exactly or misses it by
    a hair, thresholds that are not representable, non-positive thresholds, zero-area and inverted boxes.  (The oracle is pinned
    to the reference's Cython on the same set: tests/test_oracle_ref.py.)"""
    from conftest import BOUNDARY_THRESHOLDS, threshold_boundary_dets
    d = threshold_boundary_dets()
    for thr in BOUNDARY_THRESHOLDS:
        keep = hip.nms(torch.from_numpy(d).cuda(), thr).cpu().numpy()
        assert np.array_equal(keep, oracle.nms(d, thr)), thr


def test_segmented_sorted_with_max_keep(hip, oracle):
    # 5 "levels" with ragged counts, already score-sorted (the RPN case), keep[:post_nms_top_n]
    counts = [1000, 700, 64, 1, 0]
    S, N = len(counts), 1000
    boxes = np.zeros((S, N, 4), np.float32)
    refs = []
    for s, c in enumerate(counts):
        d = _dets(100 + s, max(c, 1))[:c]
        d = d[np.argsort(-d[:, 4], kind="stable")]
        boxes[s, :c] = d[:, :4]
        k = oracle.nms(d, 0.7, max_keep=300) if c else np.zeros(0, np.int64)
        refs.append(np.sort(k))      # sorted input: ascending original index == score order
    keep, cnt = hip.nms_sorted(torch.from_numpy(boxes).cuda(), torch.tensor(counts, dtype=torch.int32).cuda(), 0.7,
                               max_keep=300)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for s in range(S):
        assert cnt[s] == len(refs[s])
        assert np.array_equal(keep[s, :cnt[s]], refs[s])


def _c4_like_sorted_dets(seed, n, spread):
    """n score-sorted boxes like the C4 RPN call's: anchors of a few shapes on a stride-16 grid, decoded with small deltas.  `spread`:
    grid cells the boxes spread over -- few cells = heavy overlap (few survivors), many cells = most boxes survive."""
    rs = synth.rng(41, seed)
    shapes = np.array([[32, 32], [64, 64], [128, 128], [256, 256], [512, 512], [46, 23], [90, 45], [181, 90], [23, 46], [45, 90]], np.float32)
    sh = shapes[rs.randint(0, len(shapes), n)] * np.exp(rs.normal(0, 0.2, (n, 2))).astype(np.float32)
    cx = (rs.randint(0, spread, n) * 16 + rs.normal(0, 3, n)).astype(np.float32)
    cy = (rs.randint(0, max(spread * 3 // 5, 1), n) * 16 + rs.normal(0, 3, n)).astype(np.float32)
    b = np.stack([cx - sh[:, 0] / 2, cy - sh[:, 1] / 2, cx + sh[:, 0] / 2, cy + sh[:, 1] / 2], 1)
    b = np.clip(b, 0, [1332, 799, 1332, 799]).astype(np.float32)
    sc = np.sort(synth.dedupe_scores(rs.uniform(0, 1, n).astype(np.float32)))[::-1]
    return np.ascontiguousarray(np.hstack([b, sc[:, None]]), np.float32)


def test_two_phase_keep_max_keep_of_long_segments(hip, oracle):
    """keep[:max_keep] of LONG sorted segments (the C4 RPN call, generate_proposals.py:114-117: 6000 boxes, 1000 kept): dtc_nms_sorted
    first walks only the leading 2048 x 2048 corner of the suppression matrix and redoes a segment in full only if it has not kept
    max_keep boxes by then.  One launch with every case side by side, keep lists bit-equal to the oracle's greedy walk:
    a segment that reaches 1000 early (finished by the first phase), one whose 1000th survivor lies beyond row 2048 (second phase),
    one that never reaches 1000 (second phase, fewer kept), a short one (<= 2048 rows: finished by the first phase whatever it
    keeps) and an empty one."""
    N, cap = 6000, 1000
    segs = [_c4_like_sorted_dets(1, 6000, 84), _c4_like_sorted_dets(2, 6000, 10), _c4_like_sorted_dets(3, 6000, 6),
            _c4_like_sorted_dets(4, 1500, 84), np.zeros((0, 5), np.float32), _c4_like_sorted_dets(5, 4100, 84)]
    counts = [d.shape[0] for d in segs]
    boxes = np.zeros((len(segs), N, 4), np.float32)
    refs, last_rows = [], []
    for s_, d in enumerate(segs):
        boxes[s_, :d.shape[0]] = d[:, :4]
        k = oracle.nms(d, 0.7, max_keep=cap) if d.shape[0] else np.zeros(0, np.int64)
        refs.append(np.sort(k))
        last_rows.append(int(refs[-1][-1]) if len(k) else -1)
    # the cases this test exists for are really there
    assert len(refs[0]) == cap and last_rows[0] < 2048
    assert len(refs[1]) == cap and last_rows[1] >= 2048
    assert len(refs[2]) < cap and counts[2] > 2048
    keep, cnt = hip.nms_sorted(torch.from_numpy(boxes).cuda(), torch.tensor(counts, dtype=torch.int32).cuda(), 0.7, max_keep=cap)
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for s_ in range(len(segs)):
        assert cnt[s_] == len(refs[s_]), (s_, cnt[s_], len(refs[s_]))
        assert np.array_equal(keep[s_, :cnt[s_]], refs[s_]), s_
    # counts == NULL (every row valid) takes the same two phases
    full = np.stack([segs[0][:, :4], segs[1][:, :4]])
    keep, cnt = hip.nms_sorted(torch.from_numpy(full).cuda(), None, 0.7, max_keep=cap)
    for s_ in range(2):
        assert int(cnt[s_]) == cap and np.array_equal(keep[s_].cpu().numpy(), refs[s_])


def test_negative_count_leaves_segment_untouched(hip, oracle):
    """include/detectorch_hip.h: a NEGATIVE count marks a segment that is already reduced -- its keep / keep_count stay as they are
    (what the second phase of the long-segment path relies on).  Mixed negative and positive counts, short segments (the LDS walk)
    and long ones (the one-wave walk)."""
    for N in (1000, 2500):
        d0, d1 = _c4_like_sorted_dets(7, N, 84), _c4_like_sorted_dets(8, N - 100, 84)
        boxes = np.zeros((3, N, 4), np.float32)
        boxes[0] = d0[:, :4]; boxes[1] = d0[:, :4]; boxes[2, :N - 100] = d1[:, :4]
        L = hip.lib()
        dev = torch.device("cuda", 0)
        ws = hip.workspace(L.dtc_nms_sorted_workspace_bytes(3, N), dev)
        keep = torch.full((3, N), -7, dtype=torch.int32, device=dev)
        cnt = torch.full((3,), -7, dtype=torch.int32, device=dev)
        counts = torch.tensor([N, -1, N - 100], dtype=torch.int32, device=dev)
        bt = torch.from_numpy(boxes).cuda()
        hip.check(L.dtc_nms_sorted(bt.data_ptr(), counts.data_ptr(), 3, N, 0.7, 0, ws.data_ptr(), ws.numel(), keep.data_ptr(), N,
                                   cnt.data_ptr(), hip.stream_ptr(dev)), "nms_sorted")
        torch.cuda.synchronize()
        keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
        r0, r2 = np.sort(oracle.nms(d0, 0.7)), np.sort(oracle.nms(d1, 0.7))
        assert cnt[0] == len(r0) and np.array_equal(keep[0, :cnt[0]], r0)
        assert cnt[2] == len(r2) and np.array_equal(keep[2, :cnt[2]], r2)
        assert cnt[1] == -7 and (keep[1] == -7).all()


def test_negative_count_in_two_phase_path(hip, oracle):
    """The same protocol through the keep[:max_keep] path of LONG segments (N = 6000, max_keep = 1000: two phases, the second one
    runs on a per-segment count array the first one writes).  A caller-supplied negative count must survive both phases: the
    workspace is pre-filled with a POSITIVE garbage pattern, so a phase-1 early-out that forgets to forward the -1 shows up as an
    overwritten keep list (round-5 advisor finding)."""
    N, cap = 6000, 1000
    d0, d1 = _c4_like_sorted_dets(11, N, 84), _c4_like_sorted_dets(12, N, 10)
    boxes = np.zeros((4, N, 4), np.float32)
    boxes[0] = d0[:, :4]; boxes[1] = d1[:, :4]; boxes[2] = d1[:, :4]; boxes[3] = d0[:, :4]
    L = hip.lib()
    dev = torch.device("cuda", 0)
    nbytes = L.dtc_nms_sorted_workspace_bytes(4, N)
    ws = torch.full((nbytes // 4,), 5000, dtype=torch.int32, device=dev)          # stale positive "counts" everywhere
    keep = torch.full((4, cap), -7, dtype=torch.int32, device=dev)
    cnt = torch.full((4,), -7, dtype=torch.int32, device=dev)
    counts = torch.tensor([N, -1, N, -1], dtype=torch.int32, device=dev)
    bt = torch.from_numpy(boxes).cuda()
    hip.check(L.dtc_nms_sorted(bt.data_ptr(), counts.data_ptr(), 4, N, 0.7, cap, ws.data_ptr(), nbytes, keep.data_ptr(), cap,
                               cnt.data_ptr(), hip.stream_ptr(dev)), "nms_sorted")
    torch.cuda.synchronize()
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    r0, r2 = np.sort(oracle.nms(d0, 0.7, max_keep=cap)), np.sort(oracle.nms(d1, 0.7, max_keep=cap))
    assert int(r2[-1]) >= 2048                                                     # segment 2 really needs the second phase
    assert cnt[0] == len(r0) and np.array_equal(keep[0, :cnt[0]], r0)
    assert cnt[2] == len(r2) and np.array_equal(keep[2, :cnt[2]], r2)
    for s_ in (1, 3):
        assert cnt[s_] == -7 and (keep[s_] == -7).all(), s_


def test_idempotence_full_size(hip):
    # size-independent property at BASELINE cfg2's full size (6000 pre-NMS boxes): NMS of the survivors keeps them all
    d = _dets(4242, 6000)
    dc = torch.from_numpy(d).cuda()
    keep = hip.nms(dc, 0.7)
    again = hip.nms(dc[keep], 0.7)
    assert again.numel() == keep.numel() and torch.equal(again, torch.arange(keep.numel(), device="cuda"))


# ---------------------------------------------------------------- A6 Soft-NMS ----------------------------------------
@pytest.mark.parametrize("method,tag,ot,st", [("hard", "hard", 0.3, 0.001), ("linear", "linear", 0.3, 0.001),
                                               ("gaussian", "gaussian", 0.3, 0.001), ("linear", "linear05", 0.5, 0.0001)])
def test_soft_nms_golden(hip, method, tag, ot, st):
    from detectorch_amd.utils import boxes as box_utils
    g = golden("nms")
    d, k = box_utils.soft_nms(g["dets"], sigma=0.5, overlap_thresh=ot, score_thresh=st, method=method)
    assert np.array_equal(k, g["soft_%s_keep" % tag])
    assert np.array_equal(d, g["soft_%s_dets" % tag])


@pytest.mark.parametrize("method", ["hard", "linear", "gaussian"])
def test_soft_nms_vs_oracle_dense(hip, oracle, method):
    from detectorch_amd.utils import boxes as box_utils
    d = _dets(99, 1500, min_side=40, max_side=300)      # dense overlaps: many scores decay below the threshold
    rd, rk = oracle.soft_nms(d, 0.5, 0.3, 0.01, method)
    gd, gk = box_utils.soft_nms(d, 0.5, 0.3, 0.01, method)
    assert rk.shape[0] < 1500
    assert np.array_equal(gk, rk) and np.array_equal(gd, rd)
    dd, kk = box_utils.soft_nms(np.zeros((0, 5), np.float32))
    assert kk == [] and dd.shape == (0, 5)


def test_bbox_transform_golden(hip):
    from conftest import ulp_close
    from detectorch_amd.utils import boxes as box_utils
    g = golden("postprocess")
    boxes = g["rois"] / g["sf"][0]
    pred = box_utils.bbox_transform(boxes, g["deltas"], (10.0, 10.0, 5.0, 5.0))
    assert ulp_close(pred, g["pred"])
    assert ulp_close(box_utils.clip_tiled_boxes(pred, g["im_size"]), g["pred_clipped"])


def test_box_results_soft_nms_branch(hip, oracle):
    from detectorch_amd.utils import result_utils
    g = golden("postprocess")
    sc, bx, cb = result_utils.box_results_with_nms_and_limit(g["cls"], g["pred_clipped"].copy(), do_soft_nms=True)
    # the same loop on the oracle (result_utils.py:126-141 with soft_nms)
    n = 0
    for j in range(1, 81):
        inds = np.where(g["cls"][:, j] > 0.05)[0]
        dj = np.hstack((g["pred_clipped"][inds, j * 4:(j + 1) * 4], g["cls"][inds, j][:, None])).astype(np.float32)
        rd, _ = oracle.soft_nms(dj, 0.5, 0.5, 0.0001, "linear") if len(inds) else (dj, [])
        n += rd.shape[0]
    assert n >= sc.shape[0] >= 100


def _greedy_nms(dets, thresh, order):
    """cython_nms.pyx:37-87 with an explicit visiting order (float32 arithmetic, `>=`)."""
    x1, y1, x2, y2 = (dets[:, k] for k in range(4))
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    sup = np.zeros(len(dets), bool)
    for _i, i in enumerate(order):
        if sup[i]:
            continue
        for j in order[_i + 1:]:
            if sup[j]:
                continue
            w = max(np.float32(0), min(x2[i], x2[j]) - max(x1[i], x1[j]) + np.float32(1))
            h = max(np.float32(0), min(y2[i], y2[j]) - max(y1[i], y1[j]) + np.float32(1))
            inter = np.float32(w * h)
            if inter / np.float32(areas[i] + areas[j] - inter) >= np.float32(thresh):
                sup[j] = True
    return np.where(~sup)[0]


def test_tie_break_orders(hip):
    """the reference visits scores.argsort()[::-1] (cython_nms.pyx:45) -- equal scores in descending index order
    when the argsort is stable.  utils.boxes.nms(tie_break='index_desc') reproduces that reading, the default keeps the
    canonical (score desc, index asc); with tied scores the two keep sets differ, each equals a greedy NMS run in its order."""
    from detectorch_amd.utils import boxes as bu
    rs = synth.rng(7, 11)
    n = 300
    dets = np.zeros((n, 5), np.float32)
    dets[:, :4] = synth.make_rois(rs, n)
    dets[:, 4] = rs.randint(0, 6, n).astype(np.float32) / 8          # six distinct scores: ~50 boxes tie on each
    dets[::3, :4] = dets[1::3, :4][:len(dets[::3])]                  # identical boxes with (often) identical scores
    asc = np.lexsort((np.arange(n), -dets[:, 4]))                    # score desc, index asc
    desc = np.argsort(dets[:, 4], kind='stable')[::-1]               # what a stable argsort makes of cython_nms.pyx:45
    k_asc = bu.nms(dets, 0.5)
    k_desc = bu.nms(dets, 0.5, tie_break='index_desc')
    assert np.array_equal(k_asc, _greedy_nms(dets, 0.5, list(asc)))
    assert np.array_equal(k_desc, _greedy_nms(dets, 0.5, list(desc)))
    assert not np.array_equal(k_asc, k_desc)
    # tie-free input: the option changes nothing
    dets[:, 4] = rs.permutation(n).astype(np.float32)
    assert np.array_equal(bu.nms(dets, 0.5), bu.nms(dets, 0.5, tie_break='index_desc'))
