"""BASELINE configs[1] (Faster R-CNN R-50-C4: 63 000 anchors -> 6000 -> NMS 0.7 -> 1000; RoIAlign on [1,1024,50,84]) stage
timings on the GPU (development aid; bench.py is the contract benchmark on configs[2])."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorch_amd import hip, synth
from detectorch_amd.utils.generate_anchors import generate_anchors


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(2000)
    cls = torch.sigmoid(torch.randn((B, 15, 50, 84), generator=g, device=dev) * 2 - 2)
    bbox = torch.randn((B, 60, 50, 84), generator=g, device=dev) * 0.2
    feat = torch.relu(torch.randn((B, 1024, 50, 84), generator=g, device=dev))
    anchors = [generate_anchors(stride=16.0)]
    res = {}
    res["generate_proposals(6000->1000)"] = timeit(lambda: hip.generate_proposals([cls], [bbox], anchors, [16.0], 800, 1333, [6000], 1000, 0.7))
    boxes, scores, counts, _, _, _ = hip.generate_proposals([cls], [bbox], anchors, [16.0], 800, 1333, [6000], 1000, 0.7)
    rois = torch.cat([torch.cat([torch.full((1000, 1), float(b), device=dev), boxes[b, 0]], 1) for b in range(B)])
    for ph in (14, 7):
        out = torch.empty((B * 1000, 1024, ph, ph), device=dev)
        t = timeit(lambda: hip.roi_align_forward(feat, 1 / 16., rois, ph, ph, 0, out=out), iters=5)
        alg = feat.numel() * 4 + out.numel() * 4
        res["roi_align %dx%d sr0 (adaptive)" % (ph, ph)] = t
        print("roi_align %dx%d sr=0 C=1024 R=%d: %.3f ms, algorithmic %.0f MB -> %.2f TB/s" % (ph, ph, B * 1000, t, alg / 1e6, alg / t / 1e9))
        out2 = torch.empty_like(out)
        t2 = timeit(lambda: hip.roi_align_forward(feat, 1 / 16., rois, ph, ph, 2, out=out2), iters=5)
        print("roi_align %dx%d sr=2 (LDS kernel) : %.3f ms -> %.2f TB/s" % (ph, ph, t2, alg / t2 / 1e9))
        del out, out2
    for k, v in res.items():
        print("%-40s %.3f ms per %d images" % (k, v, B))


if __name__ == "__main__":
    main()
