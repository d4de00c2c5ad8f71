"""(f)-4 parity: HIP bbox_overlaps / box_voting vs the reference-generated golden vectors (tests/golden/bbox_vote.npz) and vs
the oracle at larger sizes; the bbox-vote branch of box_results_with_nms_and_limit end to end.  Bit-exact.  -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden
from detectorch_amd import synth

pytestmark = pytest.mark.gpu


def test_bbox_overlaps_golden_and_oracle(oracle):
    from detectorch_amd.utils import boxes as box_utils
    g = golden("bbox_vote")
    assert np.array_equal(box_utils.bbox_overlaps(g["all_dets"][:, :4], g["query"]), g["overlaps"])
    assert np.array_equal(box_utils.bbox_overlaps(g["top_dets"][:, :4], g["all_dets"][:, :4]), g["overlaps_top"])
    rs = synth.rng(14, 0)
    for (n, k) in [(1, 1), (3, 5), (1000, 1000), (257, 64), (64, 1023)]:      # k % 4 != 0 takes the scalar-store path
        b, q = synth.make_rois(rs, n), synth.make_rois(rs, k)
        assert np.array_equal(box_utils.bbox_overlaps(b, q), oracle.bbox_overlaps(b, q)), (n, k)
    assert box_utils.bbox_overlaps(np.zeros((0, 4), np.float32), g["query"]).shape == (0, 37)


@pytest.mark.parametrize("method", ['ID', 'TEMP_AVG', 'AVG', 'IOU_AVG', 'GENERALIZED_AVG', 'QUASI_SUM'])
def test_box_voting_golden(method):
    from detectorch_amd.utils import boxes as box_utils
    g = golden("bbox_vote")
    for beta in (1.0, 0.5):
        out = box_utils.box_voting(g["top_dets"], g["all_dets"], 0.6, scoring_method=method, beta=beta)
        ref = g["vote_%s_b%d" % (method, int(beta * 10))]
        assert out.dtype == np.float32 and np.array_equal(out[:, :4], ref[:, :4])            # voted boxes: bit-exact
        assert np.array_equal(out[:, 4], ref[:, 4]) if method in ('ID', 'AVG', 'QUASI_SUM', 'IOU_AVG') else \
            np.allclose(out[:, 4], ref[:, 4], rtol=1e-6, atol=0)                             # host exp/log/pow statistics


def test_box_voting_many_voters_vs_oracle(oracle):
    from detectorch_amd import hip
    rs = synth.rng(14, 1)
    base = np.array([[50, 60, 200, 220], [300, 100, 420, 300], [10, 10, 600, 400]], np.float32)
    for n in (5, 130, 1000, 4000, 8192):
        a = base[rs.randint(0, 3, n)] + rs.standard_normal((n, 4)).astype(np.float32) * 3
        all_d = np.ascontiguousarray(np.hstack([a, rs.uniform(0, 1, (n, 1))]), np.float32)
        top = np.ascontiguousarray(all_d[rs.choice(n, min(n, 16), replace=False)])
        out, nv = hip.box_voting(torch.from_numpy(top).cuda(), torch.from_numpy(all_d).cuda(), 0.5)
        assert np.array_equal(out.cpu().numpy(), oracle.box_voting(top, all_d, 0.5)), n
        assert int(nv.min()) >= 1
    with pytest.raises(RuntimeError):          # more than 8192 all_dets: DTC_EUNSUPPORTED, reported loudly
        hip.box_voting(torch.zeros((1, 5), device="cuda"), torch.zeros((8193, 5), device="cuda"), 0.5)


def test_postprocess_with_bbox_vote_golden():
    from detectorch_amd.utils import result_utils
    g, gv = golden("postprocess"), golden("bbox_vote")
    sc, bx, cb = result_utils.box_results_with_nms_and_limit(g["cls"], g["pred_clipped"].copy(), do_bbox_vote=True,
                                                             bbox_vote_thresh=0.8)
    assert np.array_equal(sc, gv["pp_vote_scores"]) and np.array_equal(bx, gv["pp_vote_boxes"])
    assert np.array_equal(np.concatenate([np.full(len(cb[j]), j, np.int32) for j in range(1, 81)]), gv["pp_vote_cls_id"])
