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


def test_idempotence_full_size(hip):
    # size-independent property at BASELINE cfg2's full size (6000 pre-NMS boxes): NMS of the survivors keeps them all
    d = _dets(4242, 6000)
    dc = torch.from_numpy(d).cuda()
    keep = hip.nms(dc, 0.7)
    again = hip.nms(dc[keep], 0.7)
    assert again.numel() == keep.numel() and torch.equal(again, torch.arange(keep.numel(), device="cuda"))
