#!/usr/bin/env python3
"""Round 6: delivered bandwidth of 1 KiB LDS-DMA gathers (buffer_load_dwordx4 ... lds) next to the same gathers into
registers (jdet_debug_gather_probe), by the cache level that serves them and by the shape of an instruction (one map row /
4 x 256 B / 8 x 128 B).  2000 workgroups x 4 waves x 128 KiB = 1.05 GB, the forward's tap volume.
    python scripts/dma_probe.py > gpurun_out/r6_dma_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd import _experimental as X  # noqa: E402
from jdet_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
ROWS = 65536
buf = torch.randn((ROWS, 256), device=dev)
sink = torch.zeros((1 << 20,), device=dev)
lib = X.lib()


def timed(go, reps=40):
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def dma(window, local, unroll, seg, read, blocks=2000, per_wave=128):
    def go():
        L.check(lib.jdet_debug_dma_probe(L.ptr(buf), ROWS, window, per_wave, local, blocks, unroll, seg, read,
                                         L.ptr(sink), L.stream_ptr(buf)), "dma probe")
    us = timed(go)
    return us, blocks * 4 * per_wave * 1024 / 1e9 / (us * 1e-6) / 1e3


def reg(window, local, unroll, blocks=2000, per_wave=128):
    def go():
        L.check(lib.jdet_debug_gather_probe(L.ptr(buf), ROWS, window, per_wave, local, blocks, unroll, L.ptr(sink),
                                            L.stream_ptr(buf)), "probe")
    us = timed(go)
    return us, blocks * 4 * per_wave * 1024 / 1e9 / (us * 1e-6) / 1e3


print("# 1 KiB gathers, 2000 workgroups x 4 waves x 128 KiB (1.05 GB); us / TB/s")
print("%-46s %16s %16s %16s %16s %16s" % ("rows drawn from", "registers u16", "dma row u8", "dma row u16", "dma 4x256 u8",
                                           "dma 8x128 u8"))
for window, local, what in ((16, 0, "shared 16 KiB window (L1 hits)"),
                            (256, 0, "shared 256 KiB window (L2 hits)"),
                            (1024, 0, "shared 1 MiB window (L2)"),
                            (4096, 0, "shared 4 MiB window"),
                            (65536, 0, "whole 64 MiB map (beyond the L2)"),
                            (32, 1, "32 KiB window per workgroup"),
                            (128, 1, "128 KiB window per workgroup"),
                            (512, 1, "512 KiB window per workgroup")):
    cells = [reg(window, local, 16), dma(window, local, 8, 1, 1), dma(window, local, 16, 1, 1),
             dma(window, local, 8, 4, 1), dma(window, local, 8, 8, 1)]
    print("%-46s " % what + " ".join("%8.1f %7.2f" % c for c in cells))
print("# 1 MiB shared window: in-flight depth and the read-back")
for unroll, seg, read in ((4, 1, 1), (8, 1, 1), (16, 1, 1), (8, 1, 0), (16, 1, 0), (4, 4, 1), (8, 4, 1), (16, 4, 1),
                          (16, 8, 1)):
    us, tb = dma(1024, 0, unroll, seg, read)
    print("unroll %2d  seg %d  read %d   %8.1f us %7.2f TB/s" % (unroll, seg, read, us, tb))
