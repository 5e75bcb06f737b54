#!/usr/bin/env python3
"""Round 6: the footprint-staged RoIAlign forward (csrc/roi_align_stage.h, LDS-DMA) next to the product kernel.
    python scripts/r6_stage.py parity            small dialect x shape cases against the CPU oracle + the product kernel
    python scripts/r6_stage.py time [reps]       north-star point: order kernel + forward, HIP events
Environment: JDET_ROI_STAGE_CPP=64|32 (channels per pass)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jdet_amd import _experimental as X  # noqa: E402
from jdet_amd import _lib as L  # noqa: E402
from tests import inputs as I  # noqa: E402

dev = torch.device("cuda:0")
lib, xl = L.lib(), X.lib()


def staged(variant, x, rois, hw, scale, order=None):
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
    L.check(xl.jdet_roi_align_forward_cl_mode(4, variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1],
                                              scale, 2, 1, order.data_ptr() if order is not None else None,
                                              out.data_ptr(), None, 0, L.stream_ptr(x)), "fwd_cl_mode 4")
    return out


def product(variant, x, rois, hw, scale, order=None):
    N, C, H, W = x.shape
    R = rois.shape[0]
    out = torch.full((R, C) + tuple(hw), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
    L.check(lib.jdet_roi_align_forward_cl_roi(variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0], hw[1], scale,
                                              2, 1, order.data_ptr() if order is not None else None, out.data_ptr(),
                                              L.stream_ptr(x)), "fwd_cl_roi")
    return out


def parity():
    from oracle import oracle as O
    worst = 0.0
    for variant, C in ((O.V_ROT, 256), (O.V_ROT, 64), (O.V_ROT_V1, 128), (O.V_HBB0, 64), (O.V_HBB1, 192)):
        for hw in ((7, 7), (4, 4), (5, 8), (8, 3)):
            rng = np.random.default_rng(300 + variant * 7 + C + hw[1])
            N, H, W, scale = 3, 40, 56, 0.25
            feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
            R = 203
            rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(4.0, 200.0)),
                                                    rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
            rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0
            if variant in (O.V_HBB0, O.V_HBB1):
                rois = I.obb_to_hbb_rois(rois)
            x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
            r = torch.from_numpy(rois).to(dev)
            a = staged(variant, x, r, hw, scale).cpu().numpy()
            b = product(variant, x, r, hw, scale).cpu().numpy()
            masked = rois[:, 0] < 0
            ok_mask = np.isnan(a[masked]).all() and not np.isnan(a[~masked]).any()
            ref = O.roi_align_forward(variant, feat, rois[~masked], hw, scale, 2)
            e_ref = np.abs(a[~masked] - ref).max() if ok_mask else float("nan")
            e_prod = np.abs(a[~masked] - b[~masked]).max() if ok_mask else float("nan")
            worst = max(worst, e_ref if ok_mask else 1e9)
            print("variant %d C %3d hw %s: masked rows untouched / no NaN %s, |staged - oracle| %.2e, |staged - product| %.2e"
                  % (variant, C, hw, ok_mask, e_ref, e_prod), flush=True)
    # big RoIs: the in-kernel fallback (line bitmaps that do not fit)
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((1, 64, 200, 300)).astype(np.float32)
    obbs = I.random_obbs(rng, 64, extent=1000.0, wh=(300.0, 1200.0))
    rois = I.rois_from_obbs(obbs, np.zeros(64))
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    a = staged(O.V_ROT, x, r, (7, 7), 0.25).cpu().numpy()
    ref = O.roi_align_forward(O.V_ROT, feat, rois, (7, 7), 0.25, 2)
    print("huge RoIs (fallback path): |staged - oracle| %.2e" % np.abs(a - ref).max())
    worst = max(worst, np.abs(a - ref).max())
    print("WORST %.3e %s" % (worst, "OK" if worst <= 2e-6 else "FAIL"))


def north_star(seed=0, R=2000):
    g = torch.Generator(device="cpu").manual_seed(seed)
    feat = torch.randn((1, 256, 256, 256), generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    rng = np.random.default_rng(seed)
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
    return feat, rois


def time(reps=200):
    feat, rois = north_star()
    R = rois.shape[0]
    obuf = torch.empty((2, R), dtype=torch.int32, device=dev)

    def order():
        L.check(lib.jdet_roi_spatial_order(rois.data_ptr(), R, 6, 0.25, 1, 256, 256, obuf[0].data_ptr(),
                                           obuf[1].data_ptr(), L.stream_ptr(feat)), "order")
    order()
    a = staged(0, feat, rois, (7, 7), 0.25, obuf[0])
    b = product(0, feat, rois, (7, 7), 0.25, obuf[0])
    torch.cuda.synchronize()
    print("north star: NaN in staged %d, |staged - product| max %.3e, share bit-equal %.3f" %
          (int(torch.isnan(a).sum()), float((a - b).abs().max()), float((a == b).float().mean())), flush=True)
    outs = torch.empty((R, 256, 7, 7), device=dev).contiguous(memory_format=torch.channels_last)

    def run(kind, with_order):
        def go():
            if with_order:
                order()
            if kind == "staged":
                L.check(xl.jdet_roi_align_forward_cl_mode(4, 0, feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), R, 7,
                                                          7, 0.25, 2, 1, obuf[0].data_ptr(), outs.data_ptr(), None, 0,
                                                          L.stream_ptr(feat)), "staged")
            else:
                L.check(lib.jdet_roi_align_forward_cl_roi(0, feat.data_ptr(), 1, 256, 256, 256, rois.data_ptr(), R, 7, 7,
                                                          0.25, 2, 1, obuf[0].data_ptr(), outs.data_ptr(),
                                                          L.stream_ptr(feat)), "product")
        for _ in range(10):
            go()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            go()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for rnd in range(2):
        for kind in ("product", "staged"):
            print("%-8s kernel alone %.1f us   with the order kernel %.1f us" % (kind, run(kind, False), run(kind, True)),
                  flush=True)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "parity"
    if what == "parity":
        parity()
    elif what == "stamps":
        pass
    else:
        time(int(sys.argv[2]) if len(sys.argv) > 2 else 200)


def stamps():
    """JDET_ROI_STAGE_ABL=16 (+ other bits): per-workgroup shader-clock stamps -> mean phase durations"""
    feat, rois = north_star()
    R = rois.shape[0]
    obuf = torch.empty((2, R), dtype=torch.int32, device=dev)
    L.check(lib.jdet_roi_spatial_order(rois.data_ptr(), R, 6, 0.25, 1, 256, 256, obuf[0].data_ptr(), obuf[1].data_ptr(),
                                       L.stream_ptr(feat)), "order")
    for _ in range(3):
        a = staged(0, feat, rois, (7, 7), 0.25, obuf[0])
    torch.cuda.synchronize()
    raw = a.permute(0, 2, 3, 1).contiguous().view(R, -1)[:, :12].contiguous().view(torch.int64).cpu().numpy()  # (R, 6)
    t0, t1, t2, t3, steps, blk = raw.T
    print("workgroups %d  mean steps %.1f" % (R, steps.mean()))
    print("trig+geom %.0f  prologue %.0f  main loop %.0f  (shader clocks, mean per workgroup); per step %.0f" %
          ((t1 - t0).mean(), (t2 - t1).mean(), (t3 - t2).mean(), ((t3 - t2) / np.maximum(steps, 1)).mean()))
    span = t3.max() - t0.min()
    print("kernel span %.0f clocks; sum of workgroup lifetimes / span = %.1f resident workgroups on average (/256 CUs = %.2f per CU)"
          % (span, (t3 - t0).sum() / span, (t3 - t0).sum() / span / 256))


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "stamps":
    stamps()
