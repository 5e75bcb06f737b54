"""usage (GPU box): python scripts/conv_wgrad_timing.py [ksplit ...] -- csrc/conv_wgrad.hip against the library's weight
gradient (autotuned, incl. its zero fill) at the 3x3 / stride 1 shapes of the S2ANet step; one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd.ops import conv_igemm as CI  # noqa: E402

torch.backends.cudnn.benchmark = True
splits = [int(v) for v in sys.argv[1:]] or [0]


def timeit(fn, iters=30, warm=8):
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


SHAPES = ((2, 256, 256, 128, 128), (2, 256, 256, 64, 64), (2, 256, 256, 57, 32), (2, 128, 128, 128, 128),
          (2, 512, 512, 32, 32), (2, 256, 256, 32, 32), (2, 256, 256, 16, 16), (2, 64, 64, 256, 256))
for n, ci, co, h, w in SHAPES:
    x = torch.randn(n, ci, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(co, ci, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    g = torch.randn(n, co, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    xn, gn = x.permute(0, 2, 3, 1), g.permute(0, 2, 3, 1)
    lib = lambda: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                      [False, True, False])
    ref = lib()[1]
    flop = 2.0 * n * h * w * ci * co * 9
    row = dict(shape=[n, ci, co, h, w], lib_us=round(timeit(lib), 1))
    row["lib_tflops"] = round(flop / row["lib_us"] / 1e6, 1)
    out = torch.zeros(co, 3, 3, ci, device="cuda")
    for ks in splits:
        out.zero_()
        CI.conv3x3_wgrad_nhwc(xn, gn, out=out, ksplit=ks)
        err = float((out.permute(0, 3, 1, 2) - ref).abs().max() / ref.abs().max())
        t = timeit(lambda: CI.conv3x3_wgrad_nhwc(xn, gn, out=out, ksplit=ks))
        row["own_us_ks%d" % ks] = round(t, 1)
        row["own_tflops_ks%d" % ks] = round(flop / t / 1e6, 1)
        row["rel_err_ks%d" % ks] = float("%.2e" % err)
    print(json.dumps(row), flush=True)
