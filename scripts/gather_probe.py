#!/usr/bin/env python3
"""Delivered bandwidth of 1 KiB row gathers on one MI355X, by the cache level that serves them
(jdet_debug_gather_probe, csrc/gather_probe.hip).  Geometry of the RoIAlign forward launch at the bench point: 2000
workgroups x 4 waves, 128 row loads per wave (1.02 M rows = 1.05 GB through the vector L1), 16 loads in flight.
    python scripts/gather_probe.py > gpurun_out/gather_probe.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd import _experimental as X  # noqa: E402
from jdet_amd import _lib as L  # noqa: E402

dev = torch.device("cuda:0")
ROWS = 65536                                      # 64 MiB: the bench map
buf = torch.randn((ROWS, 256), device=dev)
sink = torch.zeros((1 << 20,), device=dev)
lib = X.lib()   # the probes live in libjdet_experimental.so (include/jdet_experimental.h)


def run(window, local, blocks=2000, per_wave=128, unroll=16, reps=40):
    def go():
        L.check(lib.jdet_debug_gather_probe(L.ptr(buf), ROWS, window, per_wave, local, blocks, unroll, L.ptr(sink),
                                            L.stream_ptr(buf)), "probe")
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    gb = blocks * 4 * per_wave * 1024 / 1e9
    return us, gb / (us * 1e-6) / 1e3


print("# 1 KiB row gathers, 2000 workgroups x 4 waves x 128 rows (1.05 GB), 16 loads in flight per wave")
print("%-58s %9s %9s" % ("rows drawn from", "us", "TB/s"))
for window, local, what in (
        (16, 0, "one shared 16 KiB window (vector L1 hits)"),
        (256, 0, "one shared 256 KiB window (L2 hits, L1 thrash)"),
        (1024, 0, "one shared 1 MiB window (L2)"),
        (4096, 0, "one shared 4 MiB window (one XCD's L2 size)"),
        (16384, 0, "one shared 16 MiB window"),
        (65536, 0, "the whole 64 MiB map (L2 misses: Infinity Cache)"),
        (32, 1, "a 32 KiB window per workgroup (L1-sized, cold)"),
        (128, 1, "a 128 KiB window per workgroup"),
        (512, 1, "a 512 KiB window per workgroup (a RoI's footprint)"),
        (2048, 1, "a 2 MiB window per workgroup")):
    us, tbs = run(window, local)
    print("%-58s %9.1f %9.2f" % (what, us, tbs))
for unroll in (4, 8, 16):
    us, tbs = run(512, 1, unroll=unroll)
    print("%-58s %9.1f %9.2f" % ("512 KiB per workgroup, %d loads in flight" % unroll, us, tbs))
for blocks, per in ((4000, 64), (8000, 32), (1000, 256)):
    us, tbs = run(512, 1, blocks=blocks, per_wave=per)
    print("%-58s %9.1f %9.2f" % ("512 KiB per workgroup, %d workgroups x %d rows" % (blocks, per), us, tbs))


out = torch.empty((2000, 49, 256), device=dev)


def run_acc(window, per_wave, pairs, reps=40):
    def go():
        L.check(lib.jdet_debug_gather_accumulate_probe(L.ptr(buf), ROWS, window, per_wave, pairs, 2000, L.ptr(out),
                                                       L.stream_ptr(buf)), "probe")
    for _ in range(5):
        go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("# gather + LDS accumulate (ds_add_f32) + 100 MB output stream, 2000 workgroups x 4 waves, 512 KiB window each")
print("%-58s %9s" % ("rows per wave x pairs per row", "us"))
for per_wave, pairs in ((59, 0), (59, 1), (59, 2), (59, 3), (122, 0), (122, 1), (30, 2), (30, 4)):
    print("%-58s %9.1f" % ("%d x %d  (%.2f M rows, %.2f M pairs)" % (per_wave, pairs, per_wave * 8e-3, per_wave * 8e-3 * pairs),
                           run_acc(512, per_wave, pairs)))


print("# the same 1 KiB rows as dword / dwordx2 / dwordx4 loads (4 rows in flight per wave), shared window")
print("%-58s %9s %9s" % ("window, load width", "us", "TB/s"))
for window in (16, 1024, 65536):
    for w in (1, 2, 4):
        def go():
            L.check(lib.jdet_debug_gather_width_probe(L.ptr(buf), ROWS, window, 128, w, 2000, L.ptr(sink),
                                                      L.stream_ptr(buf)), "probe")
        for _ in range(5):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            go()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 40 * 1e3
        print("%-58s %9.1f %9.2f" % ("%d KiB window, %d B per lane" % (window, 4 * w), us, 2000 * 4 * 128 * 1024 / 1e9 / (us * 1e-6) / 1e3))
