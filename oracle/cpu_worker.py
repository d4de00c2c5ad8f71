"""Worker of bench.py's all-cores CPU baseline leg (TEST INFRASTRUCTURE, like everything under oracle/).

    python oracle/cpu_worker.py <dir> <image_index> <worker_id>

Loads one image's inputs from <dir>/img<k>_*.npy (memory-mapped), warms up liboracle.so, writes <dir>/ready<w>, waits for
<dir>/go, runs the oracle chain once (oracle/chain.py:fpn_hot_path) and writes <dir>/done<w> = "<start> <end>" (time.time()).
No torch import: start-up stays cheap when dozens of workers are launched at once.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


def main():
    d, k, w = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    import chain
    ld = lambda name: np.load(os.path.join(d, "img%d_%s.npy" % (k, name)), mmap_mode="r")
    meta = np.load(os.path.join(d, "img%d_meta.npy" % k))
    rpn_cls = [np.ascontiguousarray(ld("cls%d" % l)) for l in range(5)]
    rpn_bbox = [np.ascontiguousarray(ld("bbox%d" % l)) for l in range(5)]
    feats = [np.ascontiguousarray(ld("feat%d" % l)) for l in range(4)]
    cls_score, bbox_pred, masks, im_size = (np.ascontiguousarray(ld(n)) for n in ("score", "pred", "masks", "imsize"))
    chain.orc.lib()                                        # load liboracle.so before the timed region
    open(os.path.join(d, "ready" + w), "w").close()
    go = os.path.join(d, "go")
    while not os.path.exists(go):
        time.sleep(0.001)
    t0 = time.time()
    chain.fpn_hot_path(rpn_cls, rpn_bbox, feats, cls_score, bbox_pred, masks, float(meta[0]), im_size, int(meta[1]), int(meta[2]))
    t1 = time.time()
    with open(os.path.join(d, "done" + w + ".tmp"), "w") as f:
        f.write("%r %r" % (t0, t1))
    os.replace(os.path.join(d, "done" + w + ".tmp"), os.path.join(d, "done" + w))


if __name__ == "__main__":
    main()
