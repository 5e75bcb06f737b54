"""usage (GPU box): python scripts/refk_diag.py -- how far the CPU restatement and the product kernels sit from the
reference's own RoIAlign kernels (oracle/_ref/libjdet_ref_hip.so), per dialect: max abs difference and bit-equal share"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oracle import oracle as O          # noqa: E402
from oracle import ref_hip as RH        # noqa: E402
from tests.test_gpu_reference_kernels import KINDS, _case, _product   # noqa: E402

dev = torch.device("cuda:0")


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
