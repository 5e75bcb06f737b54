"""usage (GPU box): python scripts/conv_igemm_timing.py  -- fused 3x3 implicit GEMM vs the library conv, per FPN level
of S2ANet / RetinaNet at 1024^2 batch 2 (256 -> 256 channels), and the fused deformable conv vs im2col + GEMM."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from jdet_amd.ops import conv_igemm as CI   # noqa: E402
from jdet_amd.ops import dcn_v1             # noqa: E402


def timeit(fn, iters=30, warm=5):
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


rows = []
for hw in (128, 64, 32, 16, 8):
    N, C = 2, 256
    x = torch.randn(N, C, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(C, C, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last) * 0.02
    b = torch.randn(C, device="cuda")
    off = torch.randn(N, 18, hw, hw, device="cuda") * 2
    xn, wk = x.permute(0, 2, 3, 1), CI.weight_krsc(w)
    assert xn.is_contiguous()
    flop = 2.0 * N * hw * hw * C * C * 9
    t_lib = timeit(lambda: torch.relu_(F.conv2d(x, w, b, padding=1)))
    t_conv = timeit(lambda: F.conv2d(x, w, None, padding=1))
    row = dict(hw=hw, lib_conv_only_us=round(t_conv, 1), lib_conv_bias_relu_us=round(t_lib, 1))
    for tile in (64, 65, 66, 128, 129, 130):
        row["igemm_t%d_us" % tile] = round(timeit(lambda: CI.conv3x3_nhwc(xn, wk, b, True, tile=tile)), 1)
    row["igemm_auto_us"] = t = round(timeit(lambda: CI.conv3x3_nhwc(xn, wk, b, True)), 1)
    row["igemm_auto_tflops"] = round(flop / t / 1e6, 1)
    row["deform_cols_gemm_us"] = round(timeit(lambda: torch.mm(
        dcn_v1.deformable_im2col_nhwc(xn, off, 3, 3, (1, 1), (1, 1), (1, 1)), wk.view(C, -1).t())), 1)
    for tile in (64, 65, 128, 129):
        row["deform_t%d_us" % tile] = round(timeit(lambda: CI.conv3x3_nhwc(xn, wk, None, False, None, off, tile)), 1)
    rows.append(row)
    print(json.dumps(rows[-1]), flush=True)
