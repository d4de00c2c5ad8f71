"""HBM rate of the fused convolution epilogue (dtc_bias_act) on backbone-sized tensors (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detectorch_amd import hip


def run(shape, dtype, nhwc, res, iters=20):
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    x = torch.randn(*shape, device="cuda").to(dtype).contiguous(memory_format=fmt)
    r = torch.randn(*shape, device="cuda").to(dtype).contiguous(memory_format=fmt) if res else None
    b = torch.randn(shape[1], device="cuda")
    for _ in range(3):
        hip.bias_act_(x, b, r)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        hip.bias_act_(x, b, r)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = x.numel() * x.element_size() * (3 if res else 2)
    # the torch ops it replaces: x + b (broadcast), + r, relu
    bb = b.view(1, -1, 1, 1).to(dtype)
    def eager():
        y = x + bb
        if r is not None:
            y = y + r
        return torch.relu_(y)
    for _ in range(3):
        eager()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        eager()
    e1.record()
    torch.cuda.synchronize()
    ms_t = e0.elapsed_time(e1) / iters
    print("%-22s %-9s %s res=%d: %.1f us = %.2f TB/s   (torch add[+add]+relu: %.1f us)" %
          (shape, str(dtype).split(".")[-1], "NHWC" if nhwc else "NCHW", int(res), ms * 1e3, nbytes / ms / 1e9, ms_t * 1e3))


if __name__ == "__main__":
    for shape in [(8, 64, 400, 672), (8, 256, 200, 336), (8, 64, 200, 336), (8, 1024, 50, 84), (8, 2048, 25, 42), (1024, 256, 14, 14)]:
        for dtype in (torch.bfloat16, torch.float32):
            for nhwc in (True, False):
                for res in (False, True):
                    run(shape, dtype, nhwc, res)
