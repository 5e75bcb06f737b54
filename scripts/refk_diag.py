"""usage (GPU box): python scripts/refk_diag.py [--time] -- how far the CPU restatement and the product kernels sit from the
reference's own RoIAlign kernels (oracle/_ref/libjdet_ref_hip.so), per dialect: max abs difference and bit-equal share.
--time: the reference's rotated RoIAlign kernels against the product's on the north-star workload (1 x 256 x 256 x 256 map,
2000 RoIs, 7 x 7 bins, 2 x 2 samples) on the same device -- wall time per call incl. the reference wrapper's device sync."""
import sys

import numpy as np
import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O          # noqa: E402
from oracle import ref_hip as RH        # noqa: E402
from tests.test_gpu_reference_kernels import KINDS, _case, _product   # noqa: E402

dev = torch.device("cuda:0")

if "--time" in sys.argv:
    import time
    from tests import inputs as I
    from tests.test_gpu_roi_align import _layer
    rng = np.random.default_rng(1000)
    feat = torch.randn((1, 256, 256, 256), device=dev)
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, 2000), np.zeros(2000))).to(dev)
    grad = torch.randn((2000, 256, 7, 7), device=dev)
    fcl = feat.contiguous(memory_format=torch.channels_last)

    def wall(fn, n=20):
        fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e6

    layer = _layer(O.V_ROT, (7, 7), 0.25, 2)
    print("forward : reference kernel %.1f us, product %.1f us" % (
        wall(lambda: RH.roi_align_forward("rot", feat, rois, (7, 7), 0.25, 2)), wall(lambda: layer(fcl, rois))))
    x = fcl.clone().requires_grad_(True)
    y = layer(x, rois)
    print("backward: reference kernel %.1f us, product %.1f us (autograd call)" % (
        wall(lambda: RH.roi_align_backward("rot", grad, rois, feat.shape, 0.25, 2)),
        wall(lambda: torch.autograd.grad(y, x, grad, retain_graph=True))))
    sys.exit(0)


def d(a, b):
    return "%.2e/%.3f" % (np.abs(a.astype(np.float64) - b).max(), (a == b).mean())


for kind, variant in KINDS:
    for hw, s in (((7, 7), 2), ((3, 5), 0), ((2, 2), 3)):
        rng = np.random.default_rng(7 + variant + hw[0])
        feat, rois, scale = _case(rng, kind)
        grad = rng.standard_normal((rois.shape[0], feat.shape[1]) + hw).astype(np.float32)
        tf, tr, tg = (torch.from_numpy(v).to(dev) for v in (feat, rois, grad))
        ry = RH.roi_align_forward(kind, tf, tr, hw, scale, s).cpu().numpy()
        rg = RH.roi_align_backward(kind, tg, tr, feat.shape, scale, s).cpu().numpy()
        fy = RH.roi_align_forward(kind, tf, tr, hw, scale, s, fma=True).cpu().numpy()
        fg = RH.roi_align_backward(kind, tg, tr, feat.shape, scale, s, fma=True).cpu().numpy()
        oy = O.roi_align_forward(variant, feat, rois, hw, scale, s, 8)
        og = O.roi_align_backward(variant, grad, rois, feat.shape, scale, s, 8)
        y1, g1 = _product(variant, feat, rois, hw, scale, s, grad, dev, 1)
        y0, g0 = _product(variant, feat, rois, hw, scale, s, grad, dev, 0)
        print("%-7s %s s%d | fwd oracle %s prod-ref-order %s prod-merged %s fma %s | bwd oracle %s prod %s fma %s | scale y %.2f g %.2f"
              % (kind, hw, s, d(oy, ry), d(y1, ry), d(y0, ry), d(fy, ry), d(og, rg), d(g1, rg), d(fg, rg),
                 np.abs(ry).max(), np.abs(rg).max()), flush=True)
