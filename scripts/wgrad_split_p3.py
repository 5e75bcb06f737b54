"""usage (GPU box): python scripts/wgrad_split_p3.py -- the tower weight gradient (csrc/conv_wgrad.hip, 128 x 64 tiles) under forced
position splits at the P3 shape (2 x 128^2) and a packed-canvas-sized map (2 x 80^2), 256 -> 256; us per call, 25 warm-up launches"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd.ops import conv_igemm as CI  # noqa: E402


def timeit(fn, iters=30, warm=25):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for n, c, h, w in ((2, 256, 128, 128), (2, 256, 80, 80), (2, 256, 64, 64)):
    x = torch.randn(n, h, w, c, device="cuda")
    g = torch.randn(n, h, w, c, device="cuda")
    out = torch.zeros(c, 3, 3, c, device="cuda")
    row = []
    for ks in (0, 8, 16, 24, 32, 40, 48, 64, 96, 0):
        row.append("%d: %.1f" % (ks, timeit(lambda: CI.conv3x3_wgrad_nhwc(x, g, out=out, ksplit=ks))))
    print("%dx%dx%d  " % (n, h, w) + "  ".join(row), flush=True)
