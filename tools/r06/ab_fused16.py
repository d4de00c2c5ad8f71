"""Round 6, VERDICT r05 item 2: the cfg5 box-head RoIAlign launch (8 x 2000 RoIs, fp16 maps) in exact mode and in CONTRACT mode
(dtc_roi_align_set_exact(0): fused convert-multiply-accumulate on 16-bit maps), interleaved in one process; max deviation between
the two and against the CPU oracle on the first image.   python tools/r06/ab_fused16.py [--channels-last] [--bf16] [--f32-out]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from detectorch_amd import hip  # noqa: E402
from detectorch_amd.pipeline import FpnRegionPath, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--top-n", type=int, default=2000)
    ap.add_argument("--mode", choices=["ab", "exact", "contract"], default="ab", help="ab: both modes interleaved; exact / contract: only that mode (counter runs)")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fdt = torch.bfloat16 if a.bf16 else torch.float16
    if a.mode != "ab":
        hip.roi_align_set_exact(a.mode == "exact")      # before the first launch: a counter run sees one kernel only
    path = FpnRegionPath(8, dev, feat_dtype=fdt, collect_top_n=a.top_n)
    inp = synthetic_batch(8, dev, seed=5000, feat_dtype=fdt, top_n=a.top_n, channels_last=a.channels_last)
    path.bind(*inp)
    path.step(use_graph=False)
    torch.cuda.synchronize()

    def timed():
        for _ in range(3):
            path._roi_align_box()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            path._roi_align_box()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    if a.mode != "ab":
        hip.roi_align_set_exact(a.mode == "exact")
        print("%s: %.4f ms" % (a.mode, timed()))
        return
    res = {0: [], 1: []}
    feats = {}
    for _ in range(a.rounds):
        for ex in (1, 0):
            hip.roi_align_set_exact(bool(ex))
            res[ex].append(timed())
            feats[ex] = path.box_feats.float().clone()
    hip.roi_align_set_exact(True)
    alg = path.box_roialign_bytes()
    for ex in (1, 0):
        ms = float(np.mean(res[ex]))
        print("%s: %s ms  mean %.4f  frac %.4f" % ("exact" if ex else "contract", ["%.4f" % v for v in res[ex]], ms, alg / ms / 1e6 / 8000))
    d = (feats[0] - feats[1]).abs()
    print("max |contract - exact| on the %s output: %.3e  (values up to %.2f); differing elements %.4f %%" %
          (str(fdt), float(d.max()), float(feats[1].abs().max()), 100.0 * float((d > 0).float().mean())))


if __name__ == "__main__":
    main()
