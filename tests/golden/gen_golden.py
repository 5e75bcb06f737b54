#!/usr/bin/env python3
"""Generate tests/golden/*.npz.

Run in the build container only (needs /root/reference):
    python oracle/build_ref.py && python tests/golden/gen_golden.py
Each .npz holds seeded inputs and expected outputs.  The fixtures are data: no reference source text is stored.
Two kinds of expected outputs, said per file:
  * box_iou_rotated.npz (v0, v1), nms_rotated.npz, arf.npz: outputs of the reference's OWN CPU sources compiled here
    (oracle/_ref, see oracle/build_ref.py).  While generating, the restatement oracle/jdet_oracle.cpp is checked
    against them bit for bit -- that is how those parts of the oracle are pinned.  (`iou_cudasort` in
    box_iou_rotated.npz, the reference's CUDA exchange-sort ordering, is restatement output: CUDA-only source.)
  * roi_align.npz, riroi_align.npz, deform_conv.npz, dcn_v2.npz, convex_ops.npz: the reference has these operators as
    CUDA kernels only: nothing to run them on in this container (no GPU; on the GPU box the reference's kernel text,
    compiled by oracle/build_ref_hip.py, checks both the restatement and the HIP kernels: tests/test_gpu_reference_kernels.py).  The expected
    outputs are those of the restatement oracle/jdet_oracle.cpp, written only AFTER the restatement has passed the
    closed-form pins (tests/closed_form.py: affine-map RoIAlign for all five dialects, integer-offset DeformConv;
    tests/test_dcn_v2_oracle.py, tests/test_convex_oracle.py for the round-3 operators -- none involves the
    restatement's own arithmetic).  They serve the GPU box, where they are the regression
    vectors of the HIP kernels.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from tests import inputs as I  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **kw):
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **kw)
    print("%-28s %7.1f kB" % (name + ".npz", os.path.getsize(p) / 1e3))


def exact(a, b, what):
    if not np.array_equal(a, b):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise SystemExit("oracle != reference (%s): max abs diff %g at %d elements"
                         % (what, d.max(), (d > 0).sum()))


def close(a, b, what, tol=1e-5):
    d = np.abs(a.astype(np.float64) - b.astype(np.float64)).max() if a.size else 0.0
    if not d <= tol:
        raise SystemExit("oracle != reference (%s): max abs diff %g" % (what, d))


def gen_roi_align():
    rng = np.random.default_rng(100)
    N, C, H, W = 2, 6, 24, 32
    scale = 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, 14, extent=W / scale, wh=(4.0, 90.0))
    obbs[:, 1] *= H / W
    rois = np.concatenate([I.rois_from_obbs(obbs, rng.integers(0, N, 14)), I.edge_rois(H, W, scale)], 0)
    hrois = I.obb_to_hbb_rois(rois)
    out = {"feat": feat, "rois": rois, "hrois": hrois, "scale": np.float32(scale)}
    for variant, nm in ((O.V_ROT, "rot"), (O.V_ROT_V1, "rot_v1"), (O.V_HBB0, "hbb0"), (O.V_HBB1, "hbb1")):
        rr = hrois if variant in (O.V_HBB0, O.V_HBB1) else rois
        for (ph, pw), s in (((7, 7), 2), ((3, 5), 0), ((2, 2), 3)):
            key = "%s_%dx%d_s%d" % (nm, ph, pw, s)
            y = O.roi_align_forward(variant, feat, rr, (ph, pw), scale, s)
            g = rng.standard_normal(y.shape).astype(np.float32)
            gi = O.roi_align_backward(variant, g, rr, feat.shape, scale, s)
            out["y_" + key] = y
            out["g_" + key] = g
            out["gi_" + key] = gi
    save("roi_align", **out)

    # RiRoIAlign: C=3 channels x 8 orientations
    nO = 8
    feat = rng.standard_normal((N, 3 * nO, H, W)).astype(np.float32)
    rois_ri = rois.copy()
    rois_ri[:, 5] = rng.uniform(-3.5, 3.5, rois.shape[0]).astype(np.float32)  # all orientation bins
    out = {"feat": feat, "rois": rois_ri, "scale": np.float32(scale), "nO": np.int32(nO)}
    for (ph, pw), s in (((7, 7), 2), ((3, 5), 0)):
        key = "ri_%dx%d_s%d" % (ph, pw, s)
        y = O.roi_align_forward(O.V_RI, feat, rois_ri, (ph, pw), scale, s, nO)
        g = rng.standard_normal(y.shape).astype(np.float32)
        gi = O.roi_align_backward(O.V_RI, g, rois_ri, feat.shape, scale, s, nO)
        out["y_" + key] = y
        out["g_" + key] = g
        out["gi_" + key] = gi
    save("riroi_align", **out)


def gen_iou_nms():
    rng = np.random.default_rng(200)
    b1 = np.concatenate([I.clustered_obbs(rng, 40), I.special_obbs()], 0)
    b2 = np.concatenate([I.clustered_obbs(rng, 30), I.special_obbs()[::-1]], 0)
    iou = O.ref_box_iou_rotated(b1, b2, 0)
    iou_v1 = O.ref_box_iou_rotated(b1, b2, 1)
    iou_cs = O.box_iou_rotated(b1, b2, 0, 1)     # CUDA exchange-sort ordering: restatement (CUDA-only source)
    exact(O.box_iou_rotated(b1, b2, 0, 0), iou, "iou v0")
    exact(O.box_iou_rotated(b1, b2, 1, 0), iou_v1, "iou v1")
    # reference literal (box_iou_rotated.py:L513-516): analytically [[1,0.2],[0.2,1]]
    lit = np.asarray([[0, 0, 1, 1, 0], [0.5, 0.5, 1, 2, 0]], np.float32)
    iou_lit = O.ref_box_iou_rotated(lit, lit, 0)
    assert np.allclose(iou_lit, [[1, 0.2], [0.2, 1]], atol=1e-6), iou_lit
    # larger seeded case kept as a checksum only
    big1, big2 = I.clustered_obbs(rng, 300, 12, 512.0), I.clustered_obbs(rng, 280, 12, 512.0)
    iou_big = O.ref_box_iou_rotated(big1, big2, 0)
    exact(O.box_iou_rotated(big1, big2, 0, 0), iou_big, "iou big")
    save("box_iou_rotated", b1=b1, b2=b2, iou=iou, iou_v1=iou_v1, iou_cudasort=iou_cs, lit=lit,
         iou_lit=iou_lit, big1=big1, big2=big2, iou_big_sum=np.float64(iou_big.astype(np.float64).sum()),
         iou_big_nnz=np.int64((iou_big > 0).sum()), iou_big_diag=iou_big[np.arange(280), np.arange(280)])

    out = {}
    # reference literal nms_rotated.py:L599-603 -> keeps index [2]
    dets = np.asarray([[0, 0, 1, 1, 0], [0, 0, 0.5, 0.5, 0.3], [0, 0, 0.9, 0.9, 0]], np.float32)
    scores = np.asarray([0.1, 0.2, 0.3], np.float32)
    order = np.argsort(-scores, kind="stable").astype(np.int32)
    k5 = O.ref_nms_rotated_keep(dets, order, 0.3)
    d6 = np.concatenate([dets, np.ones((3, 1), np.float32)], 1)
    k6 = O.ref_nms_rotated_keep(d6, order, 0.3)
    assert list(np.nonzero(k5)[0]) == [2] and list(np.nonzero(k6)[0]) == [2]
    out.update(lit_dets=dets, lit_scores=scores, lit_keep5=k5, lit_keep6=k6)
    for n, nm in ((64, "a"), (200, "b"), (517, "c")):
        dets = I.clustered_obbs(rng, n, max(4, n // 16), 384.0)
        scores = (rng.uniform(0, 1, n) + np.arange(n) * 1e-7).astype(np.float32)
        labels = rng.integers(0, 3, n).astype(np.float32)
        order = np.argsort(-scores, kind="stable").astype(np.int32)
        d6 = np.concatenate([dets, labels[:, None]], 1)
        out["dets_" + nm], out["scores_" + nm], out["labels_" + nm] = dets, scores, labels
        for thr in (0.1, 0.5):
            k5 = O.ref_nms_rotated_keep(dets, order, thr)
            k6 = O.ref_nms_rotated_keep(d6, order, thr)
            assert np.array_equal(O.nms_rotated_keep(dets, order, thr, 1, 0), k5)
            assert np.array_equal(O.nms_rotated_keep(d6, order, thr, 1, 0), k6)
            out["keep5_%s_%g" % (nm, thr)] = k5
            out["keep6_%s_%g" % (nm, thr)] = k6
    save("nms_rotated", **out)


def gen_dcn_arf():
    rng = np.random.default_rng(300)
    out = {}
    for nm, (B, C, H, W, k, pad, stride, dil, dg) in {
        "a": (2, 4, 9, 11, 3, 1, 1, 1, 1),
        "b": (1, 6, 10, 8, 3, 1, 2, 1, 2),
        "c": (2, 4, 7, 9, 3, 2, 1, 2, 1),
    }.items():
        im = rng.standard_normal((B, C, H, W)).astype(np.float32)
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        off = (rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 2.0).astype(np.float32)
        off.flat[::7] = np.round(off.flat[::7])  # integer offsets hit the floor()/edge branches
        a = (k, k, (pad, pad), (stride, stride), (dil, dil), dg)
        col = O.deform_im2col(im, off, *a)
        gcol = rng.standard_normal(col.shape).astype(np.float32)
        gim = O.deform_col2im(gcol, off, im.shape, *a)
        goff = O.deform_col2im_coord(gcol, im, off, *a)
        out.update({"im_" + nm: im, "off_" + nm: off, "cfg_" + nm: np.asarray([k, pad, stride, dil, dg]),
                    "col_" + nm: col, "gcol_" + nm: gcol, "gim_" + nm: gim, "goff_" + nm: goff})
    save("deform_conv", **out)

    idx = I.arf_indices(8, 8, 3)
    w = rng.standard_normal((4, 3, 8, 3, 3)).astype(np.float32)
    y = O.ref_arf_forward(w, idx)
    exact(O.arf_forward(w, idx), y, "arf fwd")
    g = rng.standard_normal(y.shape).astype(np.float32)
    gw = O.ref_arf_backward(idx, g)
    exact(O.arf_backward(idx, g), gw, "arf bwd")
    idx1 = I.arf_indices(8, 8, 1)
    w1 = rng.standard_normal((2, 2, 8, 1, 1)).astype(np.float32)
    y1 = O.ref_arf_forward(w1, idx1)
    exact(O.arf_forward(w1, idx1), y1, "arf1 fwd")
    save("arf", idx=idx, w=w, y=y, g=g, gw=gw, idx1=idx1, w1=w1, y1=y1)


def gen_dcn_v2():
    rng = np.random.default_rng(400)
    out = {}
    for nm, (B, C, Cout, H, W, k, pad, stride, dil, dg) in {
        "a": (2, 8, 6, 9, 11, 3, (1, 1), 1, 1, 2),
        "b": (1, 6, 4, 10, 8, 3, (2, 1), 1, 1, 3),          # asymmetric padding: the (pad_h, pad_h) input gradient
        "c": (2, 4, 5, 12, 9, 3, (2, 2), 2, 2, 1),
    }.items():
        Ho = (H + 2 * pad[0] - (dil * (k - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad[1] - (dil * (k - 1) + 1)) // stride + 1
        x = rng.standard_normal((B, C, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cout, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
        bias = rng.standard_normal((Cout,)).astype(np.float32)
        off = (rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 2.0).astype(np.float32)
        off.flat[::7] = np.round(off.flat[::7])
        mask = rng.uniform(0, 1, size=(B, dg * k * k, Ho, Wo)).astype(np.float32)
        g = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
        a = (pad, (stride, stride), (dil, dil), dg)
        y = O.dcn_v2_forward(x, off, mask, w, bias, *a)
        gi, go, gm, gw, gb = O.dcn_v2_backward(x, off, mask, w, g, *a)
        out.update({k_ + "_" + nm: v for k_, v in dict(x=x, w=w, bias=bias, off=off, mask=mask, g=g, y=y, gi=gi, go=go,
                                                       gm=gm, gw=gw, gb=gb,
                                                       cfg=np.asarray([k, pad[0], pad[1], stride, dil, dg])).items()})
    # deformable PSRoI pooling: test_pool's shape class (dcn_v2.py:L1470-1508), scaled down
    R, N, H, W, od, G, P, part, ncls, spp, tstd, scale = 12, 2, 20, 24, 8, 2, 4, 4, 2, 3, 0.2, 0.25
    x = rng.standard_normal((N, od * G * G, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rng.integers(0, N, R)
    x1, y1 = rng.uniform(-12, W / scale, R), rng.uniform(-12, H / scale, R)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, W / scale / 2, R), y1 + rng.uniform(0, H / scale / 2, R)
    trans = rng.standard_normal((R, 2 * ncls, part, part)).astype(np.float32)
    g = rng.standard_normal((R, od, P, P)).astype(np.float32)
    for nm, no_trans in (("plain", True), ("deform", False)):
        y, cnt = O.deform_psroi_forward(x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd)
        gi, gt = O.deform_psroi_backward(g, cnt, x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd)
        out.update({"ps_y_" + nm: y, "ps_cnt_" + nm: cnt, "ps_gi_" + nm: gi, "ps_gt_" + nm: gt})
    out.update(ps_x=x, ps_rois=rois, ps_trans=trans, ps_g=g,
               ps_cfg=np.asarray([od, G, P, part, spp], np.int32), ps_f=np.asarray([scale, tstd], np.float32))
    save("dcn_v2", **out)


def gen_convex():
    rng = np.random.default_rng(500)
    c = rng.uniform(40, 160, size=(40, 1, 2))
    ps = (c + rng.normal(0, 14, size=(40, 9, 2))).reshape(40, 18).astype(np.float32)
    ps[0] = np.tile(ps[0, :2], 9)                                           # degenerate: one point nine times
    ps[1] = np.stack([np.linspace(10, 90, 9), np.linspace(20, 60, 9)], 1).reshape(18)    # collinear
    quads = []
    for _ in range(11):
        cx, cy, w, h, t = rng.uniform(40, 160), rng.uniform(40, 160), rng.uniform(10, 80), rng.uniform(6, 50), \
            rng.uniform(-np.pi, np.pi)
        d = np.asarray([[-.5, -.5], [.5, -.5], [.5, .5], [-.5, .5]]) * [w, h]
        quads.append((d @ np.asarray([[np.cos(t), np.sin(t)], [-np.sin(t), np.cos(t)]]) + [cx, cy]).reshape(8))
    quads = np.asarray(quads, np.float32)
    quads[::3] = quads[::3].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)
    pts = rng.uniform(0, 50, size=(30, 12, 2)).astype(np.float32)
    pts[:, -1] = pts[:, 0]
    masks = (rng.uniform(size=(30, 12)) > 0.3).astype(np.float32)
    masks[:, 0] = 1
    save("convex_ops", pointsets=ps, quads=quads, ious=O.convex_iou(ps, quads), boxes=O.min_area_bbox(ps), pts=pts,
         masks=masks, sort_circular=O.convex_sort(pts, masks, True), sort_open=O.convex_sort(pts, masks, False))


if __name__ == "__main__":
    if not O.have_ref():
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import build_ref
        build_ref.build()
    # the restatement-generated fixtures are written only behind the closed-form pins
    import subprocess
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x"] +
                       [os.path.join(ROOT, "tests", f) for f in ("test_closed_form_cpu.py", "test_dcn_v2_oracle.py",
                                                                 "test_convex_oracle.py")], cwd=ROOT)
    if r.returncode != 0:
        raise SystemExit("closed-form pins failed: not writing restatement fixtures")
    gen_roi_align()
    gen_iou_nms()
    gen_dcn_arf()
    gen_dcn_v2()
    gen_convex()
    print("all checks passed")
