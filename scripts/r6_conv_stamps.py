#!/usr/bin/env python3
"""Round 6: where the time of a conv_bn 64 x 64-tile launch goes ACROSS the chip -- workgroup time stamps.

JDET_CONV_BN_ABL=16 launches conv_bn_kernel<64, 32, 1, 2, 16>: every workgroup writes wall_clock64() at its start, when it
enters / leaves the K loop and at its end, plus HW_ID / XCC_ID / blockIdx, over the first 12 words of its tile's first
output row (a profiling build: those words are garbage afterwards).  Per layer: kernel span, workgroup lifetimes, the split
prologue | K loop | epilogue, workgroups per CU (placement), and what the K loop alone would take at 64 cycles per MFMA.

    JDET_CONV_BN_ABL=16 python scripts/r6_conv_stamps.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JDET_CONV_BN_ABL", "16")
from jdet_amd.ops import conv_bn as CB  # noqa: E402

dev = torch.device("cuda:0")
# (name, N, H, W, Cin, Cout, R): stride-1 layers whose plan is the 64 x 64 tile with 32-deep steps and no K split over workgroups
LAYERS = [("l1.conv2", 2, 256, 256, 64, 64, 3), ("l2.conv2", 2, 128, 128, 128, 128, 3), ("l3.conv2", 2, 64, 64, 256, 256, 3),
          ("l3.conv1", 2, 64, 64, 1024, 256, 1), ("l2.conv1", 2, 128, 128, 512, 128, 1)]
TICK = 1e-2          # wall_clock64: 100 MHz -> 0.01 us


def main():
    for name, N, H, W, Ci, Co, R in LAYERS:
        x = torch.randn(N, H, W, Ci, device=dev)
        w = torch.randn(Co, R, R, Ci, device=dev) / (R * Ci ** 0.5)
        bn = torch.nn.BatchNorm2d(Co).to(dev).eval()
        for _ in range(10):
            y = CB.conv_bn_nhwc(x, w, 1, bn, None, True)
        torch.cuda.synchronize()
        M = N * H * W
        rows = y.reshape(M, Co)[::64]                                   # first row of every M tile
        words = rows.reshape(rows.shape[0], Co // 64, 64)[:, :, :16].contiguous().view(torch.int32).cpu().numpy().astype(np.int64)
        words = words.reshape(-1, 16)
        ok = (words[:, 11] == 0x5741)
        assert ok.all(), "%s: %d of %d tiles without stamps (is JDET_CONV_BN_ABL=16 set?)" % (name, (~ok).sum(), len(ok))
        t = [(words[:, 2 * k] & 0xffffffff) | (words[:, 2 * k + 1] << 32) for k in range(4)]
        base = t[0].min()
        s, l0, l1, e = [(v - base) * TICK for v in t]
        hw, xcc = words[:, 8], words[:, 9] & 0xf
        cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
        place = xcc * 1024 + se * 64 + sh * 16 + cu
        per_cu = np.bincount(np.unique(place, return_inverse=True)[1])
        steps = R * R * Ci // 32
        ideal = steps * 16 * 64 / 2.35e3                                 # us of matrix-pipe time per wave at 2.35 GHz
        print("%-9s %4d workgroups, %3d K steps | span %.1f us | lifetime mean %.1f p10 %.1f p90 %.1f max %.1f | "
              "prologue %.1f  K loop %.1f  epilogue %.1f us (means) | last start %.1f us"
              % (name, len(s), steps, e.max(), (e - s).mean(), *np.percentile(e - s, [10, 90]), (e - s).max(),
                 (l0 - s).mean(), (l1 - l0).mean(), (e - l1).mean(), s.max()))
        print("          CUs used %d; workgroups per CU: %s | one wave's MFMAs alone %.1f us -> K loop = %.2f x (waves per SIMD "
              "sharing the pipe: workgroups per CU)" % (len(per_cu), dict(zip(*np.unique(per_cu, return_counts=True))), ideal,
                                                        (l1 - l0).mean() / ideal))
        print("          prologue: index arithmetic %.2f | first tile in registers %.2f | in LDS + barrier %.2f us from the start; "
              "epilogue: BatchNorm parameters folded %.2f us after the K loop (of %.1f)"
              % ((words[:, 12] * TICK).mean(), (words[:, 13] * TICK).mean(), (l0 - s).mean(), (words[:, 14] * TICK).mean(),
                 (e - l1).mean()))
        byx = "  ".join("%d: %.1f" % (k, (e - s)[xcc == k].mean()) for k in range(8))
        print("          lifetime by XCD: " + byx)
        # K-loop time against the number of workgroups that shared the CU
        inv = np.unique(place, return_inverse=True)[1]
        share = per_cu[inv]
        print("          K loop by workgroups on the CU: " +
              "  ".join("%d: %.1f us (n=%d)" % (k, (l1 - l0)[share == k].mean(), (share == k).sum()) for k in np.unique(share)),
              flush=True)


if __name__ == "__main__":
    main()
