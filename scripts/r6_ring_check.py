#!/usr/bin/env python3
"""Round 6: the rolling-window tap loop of the RoIAlign forward against the per-bin loop -- BIT equality.

The two loops fold the same entries in the same order with the same fmaf, so every output word must be equal.  The
switch (JDET_ROI_FWD_GRAN) is read once per process: this script runs itself once per setting, each run writes its
outputs to a file, and the parent compares them.

    python scripts/r6_ring_check.py [gran_a gran_b]         (default 4 256; on the GPU box)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cases():
    from tests import inputs as I
    out = []
    # (variant, n_orient, C, N, H, W, hw, R)
    shapes = [(0, 1, 256, 1, 256, 256, (7, 7), 2000), (0, 1, 64, 3, 40, 56, (7, 7), 203), (1, 1, 128, 2, 40, 56, (4, 4), 150),
              (0, 1, 192, 2, 33, 47, (5, 8), 180), (3, 1, 64, 3, 40, 56, (8, 3), 120), (4, 1, 256, 2, 64, 64, (7, 7), 300),
              (2, 8, 256, 2, 64, 64, (7, 7), 300), (2, 4, 128, 2, 40, 56, (7, 7), 200), (0, 1, 256, 1, 16, 16, (7, 7), 64),
              (0, 1, 256, 1, 256, 256, (7, 7), 3), (0, 1, 256, 1, 256, 256, (2, 2), 50), (0, 1, 256, 1, 256, 256, (1, 1), 50)]
    for k, (variant, no, C, N, H, W, hw, R) in enumerate(shapes):
        rng = np.random.default_rng(900 + k)
        feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
        scale = 0.25
        rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(2.0, 300.0)),
                                                rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
        rois[:, 0] = np.minimum(rois[:, 0], N - 1)              # (edge_rois name image 1)
        rois[rng.random(rois.shape[0]) < 0.15, 0] = -1.0        # masked RoIs (another pyramid level's)
        if variant in (3, 4):
            rois = I.obb_to_hbb_rois(rois)
        out.append((variant, no, feat, rois.astype(np.float32), hw, scale))
    return out


def child(path):
    from jdet_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:0")
    res = []
    for variant, no, feat, rois, hw, scale in cases():
        x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
        r = torch.from_numpy(rois).to(dev)
        N, C, H, W = x.shape
        R = r.shape[0]
        out = torch.full((R, C) + tuple(hw), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
        wsb = lib.jdet_roi_align_forward_cl_workspace(R, hw[0], hw[1])
        ws = torch.empty((wsb,), dtype=torch.uint8, device=dev)
        L.check(lib.jdet_roi_align_forward_cl(variant, x.data_ptr(), N, C, H, W, r.data_ptr(), R, hw[0], hw[1], scale, 2, no,
                                              out.data_ptr(), ws.data_ptr(), wsb, L.stream_ptr(x)), "fwd_cl")
        torch.cuda.synchronize()
        res.append(out.cpu())
    torch.save(res, path)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        return child(sys.argv[2])
    a, b = (sys.argv[1], sys.argv[2]) if len(sys.argv) > 2 else ("4", "256")
    files = []
    for gran in (a, b):
        f = tempfile.mktemp(suffix=".pt")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", f],
                              env=dict(os.environ, JDET_ROI_FWD_GRAN=gran))
        files.append(f)
    ra, rb = torch.load(files[0]), torch.load(files[1])
    ok = True
    for k, (x, y) in enumerate(zip(ra, rb)):
        same = torch.equal(x.view(torch.int32), y.view(torch.int32))
        diff = float((x - y).abs().max())
        nan = bool(torch.isnan(y).any())
        print("case %2d  shape %-22s bit-equal %s  max|diff| %.3e  nan %s" % (k, tuple(x.shape), same, diff, nan))
        ok = ok and same
    for f in files:
        os.remove(f)
    print("ALL BIT-EQUAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
