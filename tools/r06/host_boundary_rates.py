"""The PCIe-inclusive cost of the drop-ins whose reference signature hands over HOST buffers (numpy in, numpy out:
lib/utils/boxes.py:332 nms, lib/utils/result_utils.py:76 postprocess_output, :170 segm_results), next to the same work on
device-resident tensors.  bench.py's `value` is device-resident by contract; this is the note DESIGN.md section 5 refers to.

    python tools/r06/host_boundary_rates.py
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip, synth  # noqa: E402
from detectorch_amd.utils import boxes as box_utils, result_utils  # noqa: E402


def wall(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    rs = synth.rng(6, 1)
    # hard NMS, N = 6000 (the C4 RPN call of generate_proposals.py:115), host arrays in and out against device tensors in and out
    b = synth.make_rois(rs, 6000)
    dets = np.hstack([b, rs.uniform(0, 1, (6000, 1)).astype(np.float32)]).astype(np.float32)
    t_host = wall(lambda: box_utils.nms(dets, 0.7))
    d_dev = torch.from_numpy(dets).to(dev)
    t_dev = wall(lambda: hip.nms(d_dev, 0.7))
    print("nms N=6000: numpy in / numpy out %.3f ms per call; CUDA tensor in / out %.3f ms (its one D2H of the count included)" % (t_host, t_dev))
    # detection post-processing of one image, R = 1000 (result_utils.py:76): numpy in / numpy out
    R = 1000
    rois = synth.make_rois(rs, R)
    cls, dl = synth.make_head_outputs(rs, R)
    t_pp = wall(lambda: result_utils.postprocess_output(rois, 1.6, (500, 833), cls, dl), n=20)
    ct, dt, rt = (torch.from_numpy(x).to(dev) for x in (cls, dl, rois))
    t_pp_dev = wall(lambda: result_utils.postprocess_output(rt, 1.6, (500, 833), ct, dt), n=20)
    print("postprocess_output R=1000: numpy in %.3f ms per image; CUDA tensors in (numpy lists out, as the reference returns) %.3f ms" % (t_pp, t_pp_dev))
    print("(the fused path -- FpnRegionPath / bench.py -- keeps every stage on the device: 0.50 ms per 8 images, no host round trip)")


if __name__ == "__main__":
    main()
