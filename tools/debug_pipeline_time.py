import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
t0 = time.time()
def T(msg):
    print("%7.2f %s" % (time.time() - t0, msg), flush=True)
import numpy as np, torch
T("import torch")
import oracle as orc, chain
orc.lib(); T("oracle lib")
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch
dev = torch.device("cuda", 0)
B, C = 2, 8
path = FpnRegionPath(B, dev, channels=C); T("path")
inputs = synthetic_batch(B, dev, seed=3000, channels=C); T("inputs")
path.bind(*inputs); T("bind")
path.step(use_graph=False); torch.cuda.synchronize(); T("step eager")
path.step(use_graph=True); torch.cuda.synchronize(); T("step graph capture")
path.step(use_graph=True); torch.cuda.synchronize(); T("step graph replay")
rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, sf, im_size = [[t.cpu().numpy() for t in x] if isinstance(x, list) else x.cpu().numpy() for x in inputs]
T("to host")
for b in range(B):
    tm = {}
    ref = chain.fpn_hot_path([c[b] for c in rpn_cls], [d[b] for d in rpn_bbox], [f[b:b + 1] for f in feats], cls_score[b], bbox_pred[b], masks[b * path.max_out:(b + 1) * path.max_out], sf[b], im_size[b], path.pad_h, path.pad_w, timings=tm)
    T("chain %d %s" % (b, {k: round(v, 3) for k, v in tm.items()}))
    chain.compare_with_gpu(path, b, ref, int(im_size[b, 0]), int(im_size[b, 1])); T("compare %d" % b)
