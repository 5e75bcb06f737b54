#!/usr/bin/env python3
"""Round 6: where the time of the product RoIAlign forward goes ACROSS the chip -- workgroup start / end stamps.

JDET_ROI_FWD_GRAN=128 launches roi_align_fwd_merged_kernel<..., ABL = 128>: every workgroup writes wall_clock64() at its
start and end, HW_ID, XCC_ID and blockIdx into the first words of its RoI's first output row (a profiling build: that
row is garbage afterwards).  This script runs the north-star launch a few times and prints, from the last one:
kernel span, per-XCD first start / last end, the busy fraction of the (XCD, CU) workgroup slots over the span, the
distribution of workgroup durations, and how much of the span is the ragged end (time after the LAST workgroup start).

    JDET_ROI_FWD_GRAN=128 python scripts/r6_fwd_stamps.py [R]
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("JDET_ROI_FWD_GRAN", "128")
from jdet_amd import _lib as L  # noqa: E402
from tests import inputs as I  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    dev = torch.device("cuda:0")
    lib = L.lib()
    rng = np.random.default_rng(1000)
    g = torch.Generator(device="cpu").manual_seed(0)
    feat = torch.randn((1, 256, 256, 256), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    rois_np = I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))
    rois = torch.from_numpy(rois_np).to(dev)
    out = torch.empty((R, 256, 7, 7), device=dev).contiguous(memory_format=torch.channels_last)
    wsb = lib.jdet_roi_align_forward_cl_workspace(R, 7, 7)
    ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
    for _ in range(20):
        L.check(lib.jdet_roi_align_forward_cl(0, feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), R, 7, 7, 0.25, 2, 1,
                                              out.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(feat)), "fwd_cl")
    torch.cuda.synchronize()
    # memory of `out` is (R, 7, 7, 256): the stamps are the first 8 words of row (r, bin 0)
    words = out.permute(0, 2, 3, 1).reshape(R, -1)[:, :8].contiguous().view(torch.int32).cpu().numpy().astype(np.int64)
    assert ((words[:, 7] & 0xffff) == 0x5741).all(), "no stamps: is JDET_ROI_FWD_GRAN=128 set and the library rebuilt?"
    t0 = (words[:, 0] & 0xffffffff) | (words[:, 1] << 32)
    t1 = (words[:, 2] & 0xffffffff) | (words[:, 3] << 32)
    hw, xcc, blk = words[:, 4], words[:, 5] & 0xf, words[:, 6] & 7
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    tick = 1e-2          # wall_clock64: 100 MHz -> 10 ns = 0.01 us
    base = t0.min()
    s, e = (t0 - base) * tick, (t1 - base) * tick
    span = e.max()
    dur = e - s
    print("R = %d: span %.1f us (first workgroup start -> last end); workgroup duration mean %.1f  p10 %.1f  p50 %.1f  "
          "p90 %.1f  max %.1f us" % (R, span, dur.mean(), *np.percentile(dur, [10, 50, 90]), dur.max()))
    pro = (words[:, 5] >> 8) * tick
    print("prologue (workgroup start -> wave 0 enters the tap loop): mean %.2f  p10 %.2f  p90 %.2f us" %
          (pro.mean(), *np.percentile(pro, [10, 90])))
    if (words[:, 6] >> 3).any():      # the rolling-window build records the prologue's stages too (10 ns ticks from the start)
        a, b, c = ((words[:, 6] >> 3) & 0x3ff) * tick, ((words[:, 6] >> 13) & 0x3ff) * tick, ((words[:, 6] >> 23) & 0x1ff) * tick
        d = ((words[:, 7] >> 16) & 0xffff) * tick
        print("prologue stages, mean us from the workgroup's start: RoI row read %.2f | trig + barrier %.2f | sample geometry "
              "%.2f | taps merged %.2f | lists + group table %.2f" % (a.mean(), b.mean(), c.mean(), d.mean(), pro.mean()))
    print("blockIdx %% 8 == XCC_ID for %.1f %% of the workgroups" % (100.0 * np.mean((blk % 8) == xcc)))
    print("last workgroup START at %.1f us; ragged end = %.1f us = %.0f %% of the span" %
          (s.max(), span - s.max(), 100 * (span - s.max()) / span))
    # busy slot-time: sum of durations / (span * concurrent slots); slots = max concurrency observed
    ev = sorted([(x, 1) for x in s] + [(x, -1) for x in e])
    cur = peak = 0
    area = 0.0
    last = 0.0
    half_t = None
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        peak = max(peak, cur)
    print("peak concurrent workgroups %d (= %.2f per CU of 256); mean concurrency over the span %.1f = %.0f %% of the peak"
          % (peak, peak / 256.0, area / span, 100 * area / span / peak))
    for frac in (0.5, 0.75, 0.9):
        # time at which that fraction of the total workgroup-time has been spent
        acc, lastt, curc = 0.0, 0.0, 0
        tot = dur.sum()
        for t, d in ev:
            acc += curc * (t - lastt)
            if acc >= frac * tot:
                print("  %.0f %% of the workgroup-time is spent by %.1f us (%.0f %% of the span)" % (100 * frac, t, 100 * t / span))
                break
            lastt, curc = t, curc + d
    print("per XCD: workgroups, first start, last end, sum of durations / (span x 128 slots)")
    for x in range(8):
        m = xcc == x
        if m.any():
            print("  XCD %d: %4d  %6.1f  %6.1f  %.2f" % (x, m.sum(), s[m].min(), e[m].max(), dur[m].sum() / (span * 128)))
    # concurrency histogram over time in 5 us bins
    edges = np.arange(0, span + 5, 5.0)
    conc = [np.sum((s < b + 5) & (e > b)) for b in edges[:-1]]
    print("workgroups alive per 5 us bin:", " ".join("%d" % c for c in conc))
    # durations by RoI size class
    area_px = rois_np[:, 3] * rois_np[:, 4] * 0.25 * 0.25
    for lo, hi in ((0, 50), (50, 200), (200, 800), (800, 1e9)):
        m = (area_px >= lo) & (area_px < hi)
        if m.any():
            print("  RoI area [%g, %g) px: %4d RoIs, duration mean %.1f us" % (lo, hi, m.sum(), dur[m].mean()))


if __name__ == "__main__":
    main()
