"""GPU: the reference's OWN kernel text, compiled for gfx950 by oracle/build_ref_hip.py (oracle/_ref/libjdet_ref_hip.so),
run on the same device as the HIP kernels: pins BOTH the product kernels and the CPU restatement (oracle/) by reference
execution for the operators the reference has as CUDA text only -- the five RoIAligns, DeformConv v1 sampling, feature
refinement, the RepPoints geometry, convex_sort.

Tolerances (measured: profiles/r03_reference_kernels_parity.txt).  The build with contraction off runs the text's own
operation order.  The horizontal dialects involve no trigonometry: the restatement and the product's reference-order
forward equal the reference kernel BIT FOR BIT.  In the rotated dialects the kernel text calls `cos(theta)` on a float,
which device code resolves to the single-precision routine of the platform's math library (cosf: an ulp or two from the
correctly rounded value, and not the same function on any two platforms), while the restatement and the product round
the double-precision cosine -- they agree with EACH OTHER bit for bit (tests/test_gpu_roi_align.py) and with the
reference kernel to 4e-6 on N(0,1) maps (82-96 % of the elements bit-equal): 1e-5.  Backward results sum float atomics
in arbitrary order: 3e-5.  The twin built with the compiler's default contraction (what a CUDA toolchain does to the
same text) stays within 2e-5 of the contraction-free build: the parity claims do not hinge on the flag."""
import numpy as np
import pytest
import torch

from oracle import fr_oracle as FO
from oracle import oracle as O
from oracle import ref_hip as RH
from tests import inputs as I

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not RH.available(), reason="oracle/_ref/libjdet_ref_hip.so not built")]

KINDS = [("rot", O.V_ROT), ("rot_v1", O.V_ROT_V1), ("hbb0", O.V_HBB0), ("hbb1", O.V_HBB1), ("riroi", O.V_RI)]


def _case(rng, kind, C=16, nO=8):
    N, H, W, scale = 2, 24, 32, 0.25
    Ct = C * nO if kind == "riroi" else C
    feat = rng.standard_normal((N, Ct, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, 24, extent=W / scale, wh=(4.0, 90.0))
    obbs[:, 1] *= H / W
    rois = np.concatenate([I.rois_from_obbs(obbs, rng.integers(0, N, 24)), I.edge_rois(H, W, scale)], 0)
    if kind in ("hbb0", "hbb1"):
        rois = I.obb_to_hbb_rois(rois)
    return feat, rois.astype(np.float32), scale


def _product(variant, feat, rois, hw, scale, s, grad, dev, mode, nO=8):
    from jdet_amd import _lib as L
    from tests.test_gpu_roi_align import _layer
    from jdet_amd.ops import _roi_common as RC
    prev = RC.set_arithmetic("reference" if mode == 1 else "merged")
    try:
        x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        y = _layer(variant, hw, scale, s, nO)(x, torch.from_numpy(rois).to(dev))
        y.backward(torch.from_numpy(grad).to(dev))
        return y.detach().cpu().numpy(), x.grad.detach().cpu().contiguous().numpy()
    finally:
        RC.set_arithmetic(prev)


@pytest.mark.parametrize("kind,variant", KINDS)
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0), ((2, 2), 3)])
def test_roi_align_against_the_reference_kernels(dev, kind, variant, hw, s):
    rng = np.random.default_rng(7 + variant + hw[0])
    feat, rois, scale = _case(rng, kind)
    grad = rng.standard_normal((rois.shape[0], feat.shape[1]) + hw).astype(np.float32)
    tf, tr, tg = (torch.from_numpy(v).to(dev) for v in (feat, rois, grad))
    ref_y = RH.roi_align_forward(kind, tf, tr, hw, scale, s).cpu().numpy()
    ref_g = RH.roi_align_backward(kind, tg, tr, feat.shape, scale, s).cpu().numpy()
    trig = kind not in ("hbb0", "hbb1")
    fwd_tol = 1e-5 if trig else 0.0
    # 1. the CPU restatement is the reference kernel's arithmetic
    o_y = O.roi_align_forward(variant, feat, rois, hw, scale, s, 8)
    o_g = O.roi_align_backward(variant, grad, rois, feat.shape, scale, s, 8)
    np.testing.assert_allclose(o_y, ref_y, rtol=0, atol=fwd_tol)
    assert (o_y == ref_y).mean() > 0.75
    np.testing.assert_allclose(o_g, ref_g, rtol=0, atol=3e-5)
    # 2. the product kernels: reference-order arithmetic and the default merged-tap arithmetic
    y1, g1 = _product(variant, feat, rois, hw, scale, s, grad, dev, 1)
    np.testing.assert_allclose(y1, ref_y, rtol=0, atol=fwd_tol)
    np.testing.assert_allclose(g1, ref_g, rtol=0, atol=3e-5)
    y0, _ = _product(variant, feat, rois, hw, scale, s, grad, dev, 0)
    np.testing.assert_allclose(y0, ref_y, rtol=0, atol=1e-5)
    # 3. the same text under the compiler's default contraction
    fma_y = RH.roi_align_forward(kind, tf, tr, hw, scale, s, fma=True).cpu().numpy()
    np.testing.assert_allclose(fma_y, ref_y, rtol=0, atol=2e-5)


@pytest.mark.parametrize("R", [512, 2000], ids=["cfg0", "north_star"])
def test_roi_align_full_size_against_the_reference_kernel(dev, R):
    """BASELINE configs[0] (512 RoIs) and the north-star point (2000 RoIs) on the 1 x 256 x 256 x 256 map -- the shape
    the roofline kernel is timed on: product forward (reference-order and default merged-tap arithmetic; the
    channel-sliced kernels of forward mode 2 as well) and backward against the reference's OWN kernels on this
    device.  Forward tolerance 1e-4: the kernel text's `cos(float)` is the device's cosf, an ulp or two (6e-8 relative)
    from the rounded double-precision cosine the product and the restatement use; the sample position moves by that
    times its distance from the RoI centre (up to 45 map pixels here against 12 in the small cases), the value by the
    map's slope (N(0,1) pixels) times that: 3e-5 measured at the worst element of 6.4 M, most elements bit-equal.
    Backward 3e-5 x the gradient's scale (float atomics in arbitrary order there, sorted-gather sums here)."""
    from jdet_amd import _lib as L
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    rng = np.random.default_rng(R)
    feat = torch.from_numpy(rng.standard_normal((1, 256, 256, 256)).astype(np.float32)).to(dev)
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
    grad = torch.from_numpy(rng.standard_normal((R, 256, 7, 7)).astype(np.float32)).to(dev)
    ref_y = RH.roi_align_forward("rot", feat, rois, (7, 7), 0.25, 2)
    ref_g = RH.roi_align_backward("rot", grad, rois, tuple(feat.shape), 0.25, 2)
    layer = ROIAlignRotated(7, 0.25, 2)
    x = feat.contiguous(memory_format=torch.channels_last)
    from jdet_amd.ops import _roi_common as RC
    y_twin = None
    for mode in (1, 0):
        prev = RC.set_arithmetic("reference" if mode == 1 else "merged")
        try:
            xg = x.detach().clone(memory_format=torch.preserve_format).requires_grad_(True)
            y = layer(xg, rois)
            assert float((y - ref_y).abs().max()) <= 1e-4, mode
            if mode == 1:
                assert float((y == ref_y).float().mean()) > 0.5       # bit-equal wherever cos / sin round alike
                y_twin = y.detach()
            if mode == 0:
                # the DEFAULT (merged-tap) arithmetic at full size (round 6): within 2e-6 of the reference-order twin on
                # every element, and bit-equal shares with a floor -- measured 0.356 of the elements equal to the twin
                # and 0.337 to the reference kernel at both sizes (the merge re-associates a bin's sum; where no tap
                # merges the chain is the reference's).  A regression in the tap merge moves these shares first.
                assert float((y - y_twin).abs().max()) <= 2e-6
                assert float((y == y_twin).float().mean()) > 0.30
                assert float((y == ref_y).float().mean()) > 0.28
                y.backward(grad.contiguous(memory_format=torch.channels_last))
                scale = max(1.0, float(ref_g.abs().max()))
                assert float((xg.grad - ref_g).abs().max()) <= 1e-4 * scale
        finally:
            RC.set_arithmetic(prev)


@pytest.mark.parametrize("B,C,H,W,k,pad,stride,dil,dg", [(2, 4, 9, 11, 3, 1, 1, 1, 1), (1, 6, 10, 8, 3, 1, 2, 1, 2),
                                                          (2, 8, 7, 9, 3, 2, 1, 2, 1),
                                                          # S2ANet's AlignConv at P3 of a 1024 tile
                                                          (2, 256, 128, 128, 3, 1, 1, 1, 1)])
def test_deform_conv_sampling_against_the_reference_kernels(dev, B, C, H, W, k, pad, stride, dil, dg):
    from jdet_amd.ops import dcn_v1
    rng = np.random.default_rng(B * 10 + C)
    a = (k, k, (pad, pad), (stride, stride), (dil, dil), dg)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    im = rng.standard_normal((B, C, H, W)).astype(np.float32)
    off = (rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 2.0).astype(np.float32)
    off.flat[::7] = np.round(off.flat[::7])
    tim, toff = torch.from_numpy(im).to(dev), torch.from_numpy(off).to(dev)
    ref_col = RH.deform_im2col(tim, toff, *a)
    gcol = torch.from_numpy(rng.standard_normal(tuple(ref_col.shape)).astype(np.float32)).to(dev)
    ref_gim = RH.deform_col2im(gcol, toff, im.shape, *a).cpu().numpy()
    ref_goff = RH.deform_col2im_coord(gcol, tim, toff, *a).cpu().numpy()
    if C >= 256:
        # full-size case: product against the reference kernel on the device (the restatement is pinned by the small
        # cases; a 302 MB column matrix per comparison stays off the host)
        assert torch.equal(dcn_v1.deformable_im2col(tim, toff, *a), ref_col)
        ref_gim_t, ref_goff_t = torch.from_numpy(ref_gim).to(dev), torch.from_numpy(ref_goff).to(dev)
        g1 = dcn_v1.deformable_col2im(gcol, toff, im.shape, *a)
        assert float((g1 - ref_gim_t).abs().max()) <= 1e-5 * max(1.0, float(ref_gim_t.abs().max()))
        g2 = dcn_v1.deformable_col2im_coord(gcol, tim, toff, *a)
        assert float((g2 - ref_goff_t).abs().max()) <= 1e-5 * max(1.0, float(ref_goff_t.abs().max()))
        return
    # restatement
    np.testing.assert_array_equal(O.deform_im2col(im, off, *a), ref_col.cpu().numpy())
    np.testing.assert_allclose(O.deform_col2im(gcol.cpu().numpy(), off, im.shape, *a), ref_gim, rtol=0, atol=1e-5)
    np.testing.assert_allclose(O.deform_col2im_coord(gcol.cpu().numpy(), im, off, *a), ref_goff, rtol=0, atol=1e-5)
    # product (general NCHW path)
    np.testing.assert_array_equal(dcn_v1.deformable_im2col(tim, toff, *a).cpu().numpy(), ref_col.cpu().numpy())
    np.testing.assert_allclose(dcn_v1.deformable_col2im(gcol, toff, im.shape, *a).cpu().numpy(), ref_gim, rtol=0, atol=1e-5)
    np.testing.assert_allclose(dcn_v1.deformable_col2im_coord(gcol, tim, toff, *a).cpu().numpy(), ref_goff, rtol=0,
                               atol=1e-5)


@pytest.mark.parametrize("points", [1, 5])
def test_feature_refine_against_the_reference_kernels(dev, points):
    from jdet_amd.ops.fr import FR
    from tests.test_gpu_fr import _boxes
    rng = np.random.default_rng(points)
    N, C, H, W, stride = 2, 32, 12, 10, 8.0
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    boxes = _boxes(rng, N, H, W, stride)
    grad = rng.standard_normal((N, C, H, W)).astype(np.float32)
    tf, tb, tg = (torch.from_numpy(v).to(dev) for v in (feat, boxes, grad))
    ref_y = RH.feature_refine(tf, tb, 1.0 / stride, points).cpu().numpy()
    ref_g = RH.feature_refine(tf, tb, 1.0 / stride, points, grad=tg).cpu().numpy()
    tol = 2e-5 * max(1.0, np.abs(ref_y).max())          # float sin / cos of two math libraries move a sample by an ulp
    np.testing.assert_allclose(FO.feature_refine_forward(feat, boxes, 1.0 / stride, points), ref_y, rtol=0, atol=tol)
    np.testing.assert_allclose(FO.feature_refine_backward(grad, boxes, 1.0 / stride, points), ref_g, rtol=0, atol=1e-4)
    x = tf.clone().requires_grad_(True)
    y = FR(1.0 / stride, points)(x, tb)
    y.backward(tg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref_y, rtol=0, atol=tol)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref_g, rtol=0, atol=1e-4)


def test_convex_geometry_against_the_reference_kernels(dev):
    from jdet_amd.ops.convex_sort import convex_sort
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_iou
    from jdet_amd.ops.reppoints_min_area_bbox import reppoints_min_area_bbox
    from tests.test_gpu_convex_ops import _pointsets, _quads
    rng = np.random.default_rng(21)
    ps, q = _pointsets(rng, 200), _quads(rng, 13)
    q[::3] = q[::3].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)
    tp, tq = torch.from_numpy(ps).to(dev), torch.from_numpy(q).to(dev)
    # GIoU + point gradients of aligned pairs: the product's dual-number kernel against the reference's hand-derived
    # gradients (convex_giou_kernel.cu:L725-821) executed on this device
    from jdet_amd.ops.reppoints_convex_iou import reppoints_convex_giou
    m = ps.shape[0]
    qa = _quads(np.random.default_rng(22), m)
    qa[::5] = qa[::5].reshape(-1, 4, 2)[:, ::-1].reshape(-1, 8)          # some clockwise quadrilaterals
    qa[: m // 2] += (ps[: m // 2].reshape(-1, 9, 2).mean(1) - qa[: m // 2].reshape(-1, 4, 2).mean(1))[:, None, :].repeat(4, 1).reshape(-1, 8)
    ref19 = RH.convex_giou(tp[:m], torch.from_numpy(qa).to(dev)).cpu().numpy()
    giou, pg = reppoints_convex_giou(tp[:m], torch.from_numpy(qa).to(dev))
    np.testing.assert_allclose(giou.cpu().numpy(), ref19[:, 18], rtol=0, atol=1e-5)
    np.testing.assert_allclose(pg.cpu().numpy(), ref19[:, :18], rtol=0, atol=1e-4 * max(1e-3, float(np.abs(ref19[:, :18]).max())))
    ref_iou = RH.convex_iou(tp, tq).cpu().numpy()
    np.testing.assert_allclose(O.convex_iou(ps, q), ref_iou, rtol=0, atol=1e-6)
    np.testing.assert_allclose(reppoints_convex_iou(tp, tq).cpu().numpy(), ref_iou, rtol=0, atol=1e-6)
    ref_box = RH.min_area_bbox(tp).cpu().numpy()
    area = lambda b: np.linalg.norm(b[:, 0:2] - b[:, 2:4], axis=1) * np.linalg.norm(b[:, 4:6] - b[:, 2:4], axis=1)
    got = reppoints_min_area_bbox(tp).cpu().numpy()
    np.testing.assert_allclose(area(got), area(ref_box), rtol=1e-4)
    assert (np.abs(got - ref_box).max(1) < 2e-3).mean() > 0.97       # equal-area rectangles may pick another edge
    np.testing.assert_allclose(area(O.min_area_bbox(ps)), area(ref_box), rtol=1e-4)
    pts = rng.uniform(0, 50, size=(150, 12, 2)).astype(np.float32)
    pts[:, -1] = pts[:, 0]
    masks = (rng.uniform(size=(150, 12)) > 0.3).astype(np.float32)
    masks[:, 0] = 1
    for circular in (True, False):
        ref_idx = RH.convex_sort(torch.from_numpy(pts).to(dev), torch.from_numpy(masks).to(dev), circular).cpu().numpy()
        np.testing.assert_array_equal(convex_sort(torch.from_numpy(pts).to(dev), torch.from_numpy(masks).to(dev),
                                                  circular).cpu().numpy(), ref_idx)
        np.testing.assert_array_equal(O.convex_sort(pts, masks, circular), ref_idx)


@pytest.mark.parametrize("C,H,W,k,pad,stride,dil,dg", [(8, 9, 11, 3, (1, 1), 1, 1, 2), (6, 10, 8, 3, (2, 2), 2, 2, 3)])
def test_dcn_v2_sampling_against_the_reference_kernels(dev, C, H, W, k, pad, stride, dil, dg):
    """the three modulated sampling kernels as the reference's backward loop calls them: one image at a time"""
    from jdet_amd import _lib as L
    from jdet_amd.ops import dcn_v2
    from jdet_amd.ops.dcn_v1 import _geom_args
    rng = np.random.default_rng(C + H)
    Ho = (H + 2 * pad[0] - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad[1] - (dil * (k - 1) + 1)) // stride + 1
    a = (k, k, pad, (stride, stride), (dil, dil), dg)
    im = torch.from_numpy(rng.standard_normal((C, H, W)).astype(np.float32)).to(dev)
    off_np = (rng.standard_normal((dg * 2 * k * k, Ho, Wo)) * 2.0).astype(np.float32)
    off_np.flat[::7] = np.round(off_np.flat[::7])
    off = torch.from_numpy(off_np).to(dev)
    mask = torch.from_numpy(rng.uniform(0, 1, (dg * k * k, Ho, Wo)).astype(np.float32)).to(dev)
    ref_col = RH.dcn2_im2col(im, off, mask, *a)
    gcol = torch.from_numpy(rng.standard_normal(tuple(ref_col.shape)).astype(np.float32)).to(dev)
    ref_gim = RH.dcn2_col2im(gcol, off, mask, (C, H, W), *a)
    ref_goff, ref_gmask = RH.dcn2_col2im_coord(gcol, im, off, mask, *a)
    # product kernels on the same single image (B = 1: the two column layouts coincide)
    col = dcn_v2._im2col(im[None], off[None], mask[None], *a)
    np.testing.assert_allclose(col.view_as(ref_col).cpu().numpy(), ref_col.cpu().numpy(), rtol=0, atol=1e-6)
    lib, st = L.lib(), L.stream_ptr(im)
    geom = _geom_args(1, C, H, W, k, k, pad, (stride, stride), (dil, dil), dg)
    gim, goff, gmask = torch.empty_like(im), torch.empty_like(off), torch.empty_like(mask)
    L.check(lib.jdet_modulated_deform_col2im(L.ptr(gcol), L.ptr(off), L.ptr(mask), *geom, L.ptr(gim), st), "col2im")
    L.check(lib.jdet_modulated_deform_col2im_coord(L.ptr(gcol), L.ptr(im), L.ptr(off), L.ptr(mask), *geom, L.ptr(goff),
                                                   L.ptr(gmask), st), "coord")
    np.testing.assert_allclose(gim.cpu().numpy(), ref_gim.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(goff.cpu().numpy(), ref_goff.cpu().numpy(), rtol=0, atol=2e-5)
    np.testing.assert_allclose(gmask.cpu().numpy(), ref_gmask.cpu().numpy(), rtol=0, atol=2e-5)


@pytest.mark.parametrize("no_trans", [True, False])
def test_deform_psroi_pooling_against_the_reference_kernels(dev, no_trans):
    from jdet_amd.ops import dcn_v2
    rng = np.random.default_rng(17)
    R, N, H, W, od, G, P, part, ncls, spp, tstd, scale = 14, 2, 20, 24, 8, 2, 4, 4, 2, 3, 0.2, 0.25
    x = rng.standard_normal((N, od * G * G, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rng.integers(0, N, R)
    x1, y1 = rng.uniform(-12, W / scale, R), rng.uniform(-12, H / scale, R)
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, W / scale / 2, R), y1 + rng.uniform(0, H / scale / 2, R)
    trans = rng.standard_normal((R, 2 * ncls, part, part)).astype(np.float32)
    g = rng.standard_normal((R, od, P, P)).astype(np.float32)
    tx, tr, tt, tg = (torch.from_numpy(v).to(dev) for v in (x, rois, trans, g))
    cfg = (no_trans, scale, od, G, P, part, spp, tstd)
    ref_y, ref_cnt = RH.psroi_forward(tx, tr, tt, *cfg)
    ref_gi, ref_gt = RH.psroi_backward(tg, ref_cnt, tx, tr, tt, *cfg)
    # restatement
    o_y, o_cnt = O.deform_psroi_forward(x, rois, trans, *cfg)
    np.testing.assert_array_equal(o_cnt, ref_cnt.cpu().numpy())
    np.testing.assert_allclose(o_y, ref_y.cpu().numpy(), rtol=0, atol=2e-6)
    o_gi, o_gt = O.deform_psroi_backward(g, o_cnt, x, rois, trans, *cfg)
    np.testing.assert_allclose(o_gi, ref_gi.cpu().numpy(), rtol=0, atol=2e-5)
    # product
    xt, ttt = tx.clone().requires_grad_(True), tt.clone().requires_grad_(True)
    y = dcn_v2.dcn_v2_pooling(xt, tr, ttt, scale, P, od, no_trans, G, part, spp, tstd)
    y.backward(tg)
    np.testing.assert_allclose(y.detach().cpu().numpy(), ref_y.cpu().numpy(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(xt.grad.cpu().numpy(), ref_gi.cpu().numpy(), rtol=0, atol=2e-5)
    if not no_trans:
        scale_t = max(1.0, float(ref_gt.abs().max()))
        np.testing.assert_allclose(o_gt, ref_gt.cpu().numpy(), rtol=0, atol=1e-4 * scale_t)
        np.testing.assert_allclose(ttt.grad.cpu().numpy(), ref_gt.cpu().numpy(), rtol=0, atol=1e-4 * scale_t)


def test_poly_nms_against_the_reference_kernel(dev):
    """nms_poly.py's mask kernel (float arithmetic, eps sign tests) + its greedy scan against the product's polygon NMS
    and the float64 restatement: the same polygons survive, in the same order (no pair within 1e-3 of the threshold)"""
    import math
    from jdet_amd.ops.nms_poly import poly_nms
    from oracle import poly_oracle as PO
    rng = np.random.default_rng(5)
    base = np.array([0, 0, 30, 0, 30, 12, 0, 12], np.float64).reshape(4, 2)
    polys = []
    for _ in range(300):
        ang = rng.uniform(-0.5, 0.5)
        c, s = math.cos(ang), math.sin(ang)
        p = base @ np.array([[c, s], [-s, c]]) + rng.uniform(0, 120, 2)
        p[rng.integers(0, 4)] += rng.uniform(-2, 2, 2)
        polys.append(p.reshape(8))
    polys = np.stack(polys)
    scores = rng.uniform(0, 1, 300)
    thr = 0.25
    iou = PO.poly_iou_matrix(polys, polys, 0)
    ok = (np.abs(iou - thr) > 1e-3).all(1)                      # drop polygons with a borderline partner
    polys, scores = polys[ok], scores[ok]
    boxes = torch.from_numpy(np.concatenate([polys, scores[:, None]], 1).astype(np.float32)).to(dev)
    ref = RH.poly_nms(boxes, thr).cpu().numpy().tolist()
    assert 20 < len(ref) < len(polys)
    assert poly_nms(boxes, thr).cpu().numpy().tolist() == ref
    assert PO.poly_nms(polys, scores.astype(np.float32), thr) == ref


@pytest.mark.parametrize("version", [0, 1])
def test_rotated_iou_cuda_variant_against_the_reference_kernel(dev, version):
    """the CUDA text of the rotated IoU (hull ordered by an exchange sort, device cosf / sinf) against the product in
    its `sort_mode = 1` and the restatement's: IoU of well-conditioned random boxes to 1e-4, mean 1e-6"""
    from tests.test_gpu_iou_nms import _iou
    rng = np.random.default_rng(31 + version)
    b1, b2 = I.random_obbs(rng, 300, extent=300.0, wh=(10.0, 120.0)), I.random_obbs(rng, 280, extent=300.0, wh=(10.0, 120.0))
    ref = RH.box_iou_rotated(torch.from_numpy(b1).to(dev), torch.from_numpy(b2).to(dev), version).cpu().numpy()
    assert (ref > 0.05).mean() > 0.02
    for got in (_iou(b1, b2, dev, version, 1), O.box_iou_rotated(b1, b2, version=version, sort_mode=1)):
        d = np.abs(got.astype(np.float64) - ref)
        assert d.max() < 1e-4 and d.mean() < 1e-6, (d.max(), d.mean())


@pytest.mark.parametrize("box_len", [5, 6])
def test_rotated_nms_cuda_variant_against_the_reference_kernel(dev, box_len):
    """nms_rotated.py's CUDA kernel (`iou > thr`) + the scan of its launch snippet against the product's "cuda" rule"""
    from jdet_amd.ops.nms_rotated import nms_rotated_keep_mask
    rng = np.random.default_rng(41 + box_len)
    boxes = I.clustered_obbs(rng, 600, 12, 300.0)
    scores = rng.uniform(0, 1, 600).astype(np.float32)
    thr = 0.3
    iou = O.box_iou_rotated(boxes, boxes, sort_mode=1)
    ok = (np.abs(iou - thr) > 1e-3).all(1)                     # no decision within reach of the threshold
    boxes, scores = boxes[ok], scores[ok]
    dets = boxes if box_len == 5 else np.concatenate([boxes, rng.integers(0, 3, (boxes.shape[0], 1)).astype(np.float32)], 1)
    td = torch.from_numpy(dets.astype(np.float32)).to(dev)
    order = torch.argsort(torch.from_numpy(scores).to(dev), descending=True, stable=True)
    ref = RH.nms_rotated(td, order, thr).cpu().numpy()
    got = nms_rotated_keep_mask(td, order, thr, rule="cuda").cpu().numpy()
    assert 10 < ref.sum() < len(ref)
    np.testing.assert_array_equal(got, ref)
