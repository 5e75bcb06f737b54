"""usage (GPU box): python scripts/wgrad_bar.py -- what the library's weight-gradient costs per call (autotuned, incl. its
zero fill) at the tower shapes of the S2ANet step: the bar a hand-written wgrad kernel has to clear."""
import json
import torch

torch.backends.cudnn.benchmark = True


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


for n, c, h, w in ((2, 256, 128, 128), (2, 256, 64, 64), (2, 256, 57, 32), (2, 128, 128, 128), (2, 512, 32, 32)):
    x = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(c, c, 3, 3, device="cuda") * 0.02).contiguous(memory_format=torch.channels_last)
    g = torch.randn(n, c, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    f = lambda m: torch.ops.aten.convolution_backward(g, x, wt, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, m)
    flop = 2.0 * n * h * w * c * c * 9
    t_w, t_x = timeit(lambda: f([False, True, False])), timeit(lambda: f([True, False, False]))
    print(json.dumps(dict(shape=[n, c, h, w], wgrad_us=round(t_w, 1), wgrad_tflops=round(flop / t_w / 1e6, 1),
                          dgrad_us=round(t_x, 1), dgrad_tflops=round(flop / t_x / 1e6, 1))), flush=True)
