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
names = ["bare MFMAs", "+ LDS fragment fetches", "+ LDS tile writes + barrier", "+ buffer loads"]
for variant in range(4):
    row = []
    for per_cu in (1, 2, 3, 4):
        n = 256 * per_cu
        cyc = torch.zeros(n * 4, dtype=torch.int64, device="cuda")
        for _ in range(2):
            L.check(X.lib().jdet_debug_mfma_probe(variant, L.ptr(src), n, steps, L.ptr(cyc), L.ptr(sink),
                                                  L.stream_ptr(src)), "probe")
        torch.cuda.synchronize()
        c = cyc.double()
        # per wave: cycles per own MFMA; per SIMD the pipe serves `per_cu` waves -> cycles of pipe time per MFMA
        row.append("%d/CU: %.0f cyc/MFMA/wave = %.0f pipe cyc/MFMA (max wave %.0f)"
                   % (per_cu, c.mean() / (steps * 32), c.mean() / (steps * 32) / per_cu, c.max() / (steps * 32)))
    print("variant %d (%s)\n   " % (variant, names[variant]) + "\n   ".join(row), flush=True)
