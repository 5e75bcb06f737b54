#!/usr/bin/env python3
"""Experiment: does the processing order inside an XCD's run matter?  Times jdet_roi_align_forward_cl_roi at the
north-star point under custom `order` arrays (size classes first / last, Hilbert vs Morton, random)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jdet_amd import _lib as L  # noqa: E402
from tests import inputs as I  # noqa: E402


def morton(ix, iy):
    def sp(v):
        v = v & 0xffff
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        v = (v | (v << 1)) & 0x55555555
        return v
    return sp(ix) | (sp(iy) << 1)


def deal(sorted_idx):
    R = len(sorted_idx)
    order = np.zeros(R, np.int32)
    starts = [sum((R - y + 7) >> 3 for y in range(x)) for x in range(8)]
    for b in range(R):
        order[b] = sorted_idx[starts[b & 7] + (b >> 3)]
    return order


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "morton"
    dev = torch.device("cuda:0")
    lib = L.lib()
    rng = np.random.default_rng(0)
    R = 2000
    g = torch.Generator(device="cpu").manual_seed(0)
    feat = torch.randn((1, 256, 256, 256), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    rois = I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))
    cx, cy = rois[:, 1] * 0.25 / 256, rois[:, 2] * 0.25 / 256
    key = np.array([morton(int(min(max(x, 0), 0.999999) * 32), int(min(max(y, 0), 0.999999) * 32)) for x, y in zip(cx, cy)])
    srt = np.argsort(key, kind="stable")
    region = np.zeros(R, np.int64)
    starts = [sum((R - y + 7) >> 3 for y in range(x)) for x in range(9)]
    for x in range(8):
        region[srt[starts[x]:starts[x + 1]]] = x
    big = (np.maximum(rois[:, 3], rois[:, 4]) * 0.25 / 7 >= 2.0).astype(np.int64)
    size = (rois[:, 3] * rois[:, 4])
    if which == "morton":
        s2 = srt
    elif which == "bigfirst":
        s2 = np.lexsort((key, 1 - big, region))
    elif which == "biglast":
        s2 = np.lexsort((key, big, region))
    elif which == "bysize":
        s2 = np.lexsort((size, region))
    elif which == "random":
        s2 = np.concatenate([rng.permutation(srt[starts[x]:starts[x + 1]]) for x in range(8)])
    elif which == "fine":   # 64x64 cells instead of 32x32
        key2 = np.array([morton(int(min(max(x, 0), 0.999999) * 128), int(min(max(y, 0), 0.999999) * 128)) for x, y in zip(cx, cy)])
        s2 = np.lexsort((key2, region))
    order = torch.from_numpy(deal(s2)).to(dev)
    rt = torch.from_numpy(rois).to(dev)
    out = torch.empty((R, 256, 7, 7), device=dev, memory_format=torch.channels_last)
    st = L.stream_ptr(feat)

    def step():
        L.check(lib.jdet_roi_align_forward_cl_roi(0, feat.data_ptr(), 1, 256, 256, 256, rt.data_ptr(), R, 7, 7, 0.25, 2, 1,
                                                  order.data_ptr(), out.data_ptr(), st), "fwd")
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        step()
    e1.record()
    torch.cuda.synchronize()
    print("%s: %.1f us (big RoIs: %d)" % (which, e0.elapsed_time(e1) * 10, int(big.sum())))


if __name__ == "__main__":
    main()
