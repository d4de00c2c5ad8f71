import sys, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/oracle')
import oracle as orc
from detectorch_amd import synth
shapes = synth.fpn_level_shapes()
strides = (4.0, 8.0, 16.0, 32.0, 64.0)
for img in range(2):
    rs = synth.rng(3, img)
    props, scores = [], []
    for l in range(5):
        H, W = shapes[l]
        p, d = synth.make_rpn_outputs(rs, 3, H, W)
        anchors = orc.generate_anchors(strides[l], (32.0 * 2 ** l,), (0.5, 1, 2))
        b, s = orc.generate_proposals(p[0], d[0], anchors, strides[l], 800, 1344, 1000, 1000, 0.7)
        props.append(b); scores.append(s)
    rois, rsc, _ = orc.collect(np.concatenate(props), np.concatenate(scores), 1000)
    per_level, restore, lv = orc.distribute(rois, 2, 5)
    out = np.hstack([np.zeros((len(rois), 1), np.float32), rois, (lv - 2).astype(np.float32)[:, None]])
    np.save('/tmp/sim/rois_%d.npy' % img, out)
    print(img, out.shape, np.bincount((lv - 2).astype(int)))
# C4 proposals
H, W = synth.c4_shape()
rs = synth.rng(2, 0)
p, d = synth.make_rpn_outputs(rs, 15, H, W)
anchors = orc.generate_anchors(16.0)
rois, sc = orc.generate_proposals(p[0], d[0], anchors, 16.0, 800, 1333, 6000, 1000, 0.7)
np.save('/tmp/sim/c4_rois.npy', rois); print('c4', rois.shape)
