"""Development aid: RoIs (x1,y1,x2,y2,level) of the FPN box head for a few synthetic images, in the RoIAlign visiting order,
produced on the CPU by the oracle (same distributions as the bench inputs).  Output: /tmp/rois_<img>.npy  [n,6] (b,x1,y1,x2,y2,lvl)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle as orc
from detectorch_amd import synth

def image_rois(seed, top_n=1000):
    rs = synth.rng(3, seed)
    props, scores = [], []
    for l, (h, w) in enumerate(synth.fpn_level_shapes()):
        p, d = synth.make_rpn_outputs(rs, 3, h, w, tie_free=False)
        anchors = orc.generate_anchors(synth.FPN_STRIDES[l], (32.0 * 2 ** l,), (0.5, 1, 2))
        b, s = orc.generate_proposals(p[0], d[0], anchors, float(synth.FPN_STRIDES[l]), 800, 1344, 1000, 1000, 0.7)
        props.append(b); scores.append(s)
    rois, rsc, _ = orc.collect(np.concatenate(props), np.concatenate(scores), top_n)
    _, _, lv = orc.distribute(rois, 2, 5)
    lvl = (lv - 2).astype(np.int64)
    # visiting order of fpn.hip: (level, band of 16 feature rows of the centre, x centre in feature pixels, rank)
    fs = lvl + 2
    xc = ((rois[:, 0] + rois[:, 2]) * 0.5).astype(np.int64); yc = ((rois[:, 1] + rois[:, 3]) * 0.5).astype(np.int64)
    band = np.minimum((yc >> fs) >> 4, 63); xf = np.minimum(xc >> fs, 4095)
    order = np.lexsort((np.arange(len(rois)), xf, band, lvl))
    out = np.zeros((len(rois), 6), np.float32)
    out[:, 1:5] = rois[order]; out[:, 5] = lvl[order]
    return out

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    for i in range(n):
        r = image_rois(i)
        np.save("/tmp/rois_%d.npy" % i, r)
        print(i, r.shape, np.bincount(r[:, 5].astype(int), minlength=4))
