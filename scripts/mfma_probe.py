"""usage (GPU box): python scripts/mfma_probe.py -- shader cycles per v_mfma_f32_32x32x2_f32 of one wave's stream as the
pieces of the conv_wgrad.hip K loop are added (csrc/experimental/mfma_probe.hip), at 1 / 2 / 3 / 4 workgroups per CU."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd import _experimental as X  # noqa: E402
from jdet_amd import _lib as L  # noqa: E402

src = torch.randn(1 << 16, device="cuda")
sink = torch.zeros(4, device="cuda")
steps = 256
names = ["bare MFMAs", "+ LDS fragment fetches", "+ LDS tile writes + barrier", "+ buffer loads",
         "64-tile step: ONE accumulator, 8 ds_read_b128 per 16 MFMAs", "128-tile step: FOUR accumulators, 4 ds_read_b128 per 16 MFMAs",
         "one accumulator, operands in registers", "two accumulators (even / odd slices), 8 ds_read_b128 per 16 MFMAs"]
# (variants 0-3 count 32 MFMAs per step, 4-7 count 16)
for variant in range(8):
    row = []
    for per_cu in (1, 2, 3, 4):
        n = 256 * per_cu
        cyc = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
        for _ in range(2):
            L.check(X.lib().jdet_debug_mfma_probe(variant, L.ptr(src), n, steps, L.ptr(cyc), L.ptr(sink),
                                                  L.stream_ptr(src)), "probe")
        torch.cuda.synchronize()
        c = cyc.double() * (2.0 if variant >= 4 else 1.0)
        # per wave: cycles per own MFMA; per SIMD the pipe serves `per_cu` waves -> cycles of pipe time per MFMA
        row.append("%d/CU: %.0f cyc/MFMA/wave = %.0f pipe cyc/MFMA (max wave %.0f)"
                   % (per_cu, c.mean() / (steps * 32), c.mean() / (steps * 32) / per_cu, c.max() / (steps * 32)))
    print("variant %d (%s)\n   " % (variant, names[variant]) + "\n   ".join(row), flush=True)


# the clock under the matrix pipe's load: a long run of the two inner structures (4 workgroups per CU, random operands), shader
# cycles of the slowest wave (s_memtime) over the launch's HIP-event time
steps = 8192
for variant in (4, 5, 6):
    n = 1024
    cyc = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        L.check(X.lib().jdet_debug_mfma_probe(variant, L.ptr(src), n, steps, L.ptr(cyc), L.ptr(sink), L.stream_ptr(src)), "probe")
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    c = cyc.double()
    tf = n * 4 * steps * 16 * 4096 / (ms * 1e-3) / 1e12
    print("variant %d: %d steps, launch %.3f ms, slowest wave %.3e cycles (mean %.3e) -> %.2f GHz; %.1f TFLOP/s"
          % (variant, steps, ms, c.max(), c.mean(), c.max() / (ms * 1e-3) / 1e9, tf), flush=True)
