"""1x1 convolutions of the ResNet-50 / FPN stack at the S2ANet bench size (2 x 1024^2): library convolution (MIOpen, what
the model uses) against the same contraction as a plain GEMM on the channels-last activation matrix (hipBLASLt through
torch.matmul): forward, data gradient, weight gradient.  python scripts/conv1x1_probe.py"""
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
# (N, Cin, H, W, Cout) of the stride-1 1x1 convs (bottleneck conv1 / conv3, FPN laterals), 2 images of 1024^2
SHAPES = [(2, 64, 256, 256, 64), (2, 64, 256, 256, 256), (2, 256, 256, 256, 64), (2, 256, 256, 256, 128),
          (2, 128, 128, 128, 512), (2, 512, 128, 128, 128), (2, 512, 128, 128, 256), (2, 256, 64, 64, 1024),
          (2, 1024, 64, 64, 256), (2, 1024, 64, 64, 512), (2, 512, 32, 32, 2048), (2, 2048, 32, 32, 512),
          (2, 512, 128, 128, 256), (2, 1024, 64, 64, 256), (2, 2048, 32, 32, 256)]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = {"conv": [0, 0, 0], "gemm": [0, 0, 0]}
for (N, Ci, H, W, Co) in SHAPES:
    x = torch.randn(N, Ci, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, Ci, 1, 1, device=dev)
    gy = torch.randn(N, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    xm, wm, gm = x.permute(0, 2, 3, 1).reshape(-1, Ci), w.view(Co, Ci), gy.permute(0, 2, 3, 1).reshape(-1, Co)
    c = [timeit(lambda: F.conv2d(x, w)),
         timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])),
         timeit(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [False, True, False]))]
    g = [timeit(lambda: xm @ wm.t()), timeit(lambda: gm @ wm), timeit(lambda: gm.t() @ xm)]
    for i in range(3):
        tot["conv"][i] += c[i]
        tot["gemm"][i] += g[i]
    print("%-26s conv fwd/dgrad/wgrad %6.1f %6.1f %6.1f us | gemm %6.1f %6.1f %6.1f us" % ((N, Ci, H, W, Co), *c, *g), flush=True)
print("sum: conv %s gemm %s" % ([round(v) for v in tot["conv"]], [round(v) for v in tot["gemm"]]))
