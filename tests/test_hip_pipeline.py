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
