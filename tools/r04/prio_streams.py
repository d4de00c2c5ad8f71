"""Experiment: the short kernels of a step on a HIGH-priority stream, the two chip-filling RoIAlign launches on a LOW-priority one
(events in between), two steps in flight, eager launches.  Against the product's scheme (each step entirely on its own stream).
    python tools/r04/prio_streams.py [--steps 400] [--fp16 --channels-last --top-n 2000]"""
import argparse
import os
import sys
import time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from detectorch_amd import hip
from detectorch_amd.pipeline import FpnRegionPath, StepPipeline, synthetic_batch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--fp16", action="store_true")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--top-n", type=int, default=1000)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    fdt = torch.float16 if a.fp16 else torch.float32
    paths = []
    for i in range(2):
        p = FpnRegionPath(a.batch, dev, feat_dtype=fdt, collect_top_n=a.top_n, max_out=104)
        p.bind(*synthetic_batch(a.batch, dev, seed=100 + i, feat_dtype=fdt, top_n=a.top_n, channels_last=a.channels_last, max_out=104))
        p.step(use_graph=False)
        paths.append(p)
    torch.cuda.synchronize()

    def timed(fn, sync):
        for _ in range(20): fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(a.steps): fn()
        sync()
        return (time.perf_counter() - t0) / a.steps * 1e3

    # A. the product's scheme, eager and graph replay
    for graph in (False, True):
        pipe = StepPipeline(paths, dev, 2)
        ms = timed(lambda: pipe.step(use_graph=graph), lambda: (pipe.synchronize(), torch.cuda.synchronize()))
        print("product scheme, %s: %.4f ms/step = %.0f images/s" % ("graph replay" if graph else "eager", ms, a.batch / ms * 1e3))

    # B. priorities
    lo_pri, hi_pri = 0, -1
    for shared_lo in (True, False):
        his = [torch.cuda.Stream(device=dev, priority=hi_pri) for _ in paths]
        los = [torch.cuda.Stream(device=dev, priority=lo_pri)] * 2 if shared_lo else [torch.cuda.Stream(device=dev, priority=lo_pri) for _ in paths]
        for p, h, l in zip(paths, his, los):
            def wrap(orig, h=h, l=l):
                def f(st=None):
                    e1 = torch.cuda.Event(); e1.record(h); l.wait_event(e1)
                    orig(l.cuda_stream)
                    e2 = torch.cuda.Event(); e2.record(l); h.wait_event(e2)
                return f
            p._rb, p._rm = p._roi_align_box, p._roi_align_mask
            p._roi_align_box, p._roi_align_mask = wrap(p._rb), wrap(p._rm)
        cnt = [0]

        def step():
            i = cnt[0] & 1; cnt[0] += 1
            with torch.cuda.stream(his[i]):
                paths[i].step(use_graph=False)
        ms = timed(step, torch.cuda.synchronize)
        print("short kernels on high-priority streams, RoIAlign on %s low-priority stream(s), eager: %.4f ms/step = %.0f images/s"
              % ("ONE shared" if shared_lo else "per-path", ms, a.batch / ms * 1e3))
        for p in paths:
            p._roi_align_box, p._roi_align_mask = p._rb, p._rm


if __name__ == "__main__":
    main()
