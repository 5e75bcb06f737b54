"""CPU: the oracle restatement against closed forms that involve neither the restatement nor any build of the
reference text (tests/closed_form.py): affine-map RoIAlign for the five dialects, integer-offset DeformConv.
These are the pins of oracle/jdet_oracle.cpp for the operators whose reference source is CUDA-only."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import closed_form as CF

DIALECTS = [(O.V_ROT, "rot"), (O.V_ROT_V1, "rot_v1"), (O.V_HBB0, "hbb0"), (O.V_HBB1, "hbb1"), (O.V_RI, "riroi")]


@pytest.mark.parametrize("variant,nm", DIALECTS)
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 1), ((4, 4), 0)])
def test_oracle_roi_align_on_affine_map(variant, nm, hw, s):
    rng = np.random.default_rng(11 + variant)
    N, C, H, W, scale, nO = 2, 16, 48, 56, 0.25, 8
    feat, a, b, d = CF.affine_map(rng, N, C, H, W)
    rois = CF.interior_rois(rng, 40, N, H, W, scale, variant, max_wh=18.0)
    y = O.roi_align_forward(variant, feat, rois, hw, scale, s, nO)
    ref = CF.roi_align_expected(variant, (a, b, d), rois, scale, hw[0], hw[1], nO)
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-4)


def test_dialects_are_distinguishable():
    """the closed form separates the conventions: v0 vs v1 rotated (rotation sense, -0.5), hbb v0 vs v1 (+1 px)"""
    rng = np.random.default_rng(3)
    _, a, b, d = CF.affine_map(rng, 1, 8, 40, 40)
    roi = np.asarray([[0, 80, 72, 30, 12, 0.6]], np.float32)
    e0 = CF.roi_align_expected(O.V_ROT, (a, b, d), roi, 0.25, 7, 7)
    e1 = CF.roi_align_expected(O.V_ROT_V1, (a, b, d), roi, 0.25, 7, 7)
    assert np.abs(e0 - e1).max() > 0.1
    hroi = np.asarray([[0, 60, 60, 100, 90]], np.float32)
    h0 = CF.roi_align_expected(O.V_HBB0, (a, b, d), hroi, 0.25, 7, 7)
    h1 = CF.roi_align_expected(O.V_HBB1, (a, b, d), hroi, 0.25, 7, 7)
    assert np.abs(h0 - h1).max() > 0.01


@pytest.mark.parametrize("k,pad,stride,dil", [(3, 1, 1, 1), (3, 2, 2, 2), (1, 0, 1, 1)])
def test_oracle_deform_im2col_integer_offsets(k, pad, stride, dil):
    rng = np.random.default_rng(k + pad)
    B, C, Cout, H, W = 2, 6, 5, 13, 15
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((Cout, C, k, k)).astype(np.float32)
    dy = rng.integers(-3, 4, size=(k, k))
    dx = rng.integers(-3, 4, size=(k, k))
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = CF.integer_offsets(dy, dx, B, Ho, Wo)
    col = O.deform_im2col(x, off, k, k, (pad, pad), (stride, stride), (dil, dil), 1)
    y = (w.reshape(Cout, -1).astype(np.float64) @ col.reshape(C * k * k, -1).astype(np.float64))
    y = y.reshape(Cout, B, Ho, Wo).transpose(1, 0, 2, 3)
    np.testing.assert_allclose(y, CF.deform_conv_integer_expected(x, w, dy, dx, pad, stride, dil), rtol=0, atol=1e-5)


# ---- pins the affine map cannot give: sample positions, sampling grid, divisor, boundary rules ----
@pytest.mark.parametrize("variant,nm", DIALECTS)
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 1), ((4, 4), 0), ((2, 3), 3)])
def test_oracle_roi_align_on_quadratic_map(variant, nm, hw, s):
    """quadratic map: the pooled value depends on every sample position (bilinear of x^2 at fraction t = x^2 + t(1-t))"""
    rng = np.random.default_rng(5 + variant)
    N, C, H, W, scale, nO = 2, 16, 48, 56, 0.25, 8
    feat, co = CF.quadratic_map(rng, N, C, H, W)
    rois = CF.interior_rois(rng, 40, N, H, W, scale, variant, max_wh=18.0)
    y = O.roi_align_forward(variant, feat, rois, hw, scale, s, nO)
    ref = CF.roi_align_expected_quadratic(variant, co, rois, scale, hw[0], hw[1], s, nO)
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-3)          # |f| up to ~900: 1e-6 relative


def test_quadratic_pin_separates_sampling_grids():
    """what the pin is for: a different sampling grid, or the fixed grid instead of the adaptive one, moves the
    expectation by far more than the tolerance"""
    rng = np.random.default_rng(9)
    _, co = CF.quadratic_map(rng, 1, 8, 48, 56)
    rois = CF.interior_rois(rng, 30, 1, 48, 56, 0.25, O.V_ROT, max_wh=30.0)
    e = {s: CF.roi_align_expected_quadratic(O.V_ROT, co, rois, 0.25, 7, 7, s) for s in (0, 1, 2, 3)}
    for s1, s2 in ((1, 2), (2, 3), (0, 2)):
        assert np.abs(e[s1] - e[s2]).max() > 0.02


@pytest.mark.parametrize("variant,nm", DIALECTS[:4])
def test_oracle_roi_align_boundary_literals(variant, nm):
    """hand-derived values on a 3x3 map: dropped samples (< -1, > extent), coordinates moved to 0, the last pixel"""
    for sx, sy, exp in CF.BOUNDARY_POINTS:
        roi = np.asarray([CF.boundary_roi(variant, sx, sy)], np.float32)
        y = O.roi_align_forward(variant, CF.BOUNDARY_MAP, roi, (1, 1), 1.0, 1)
        assert float(y.ravel()[0]) == pytest.approx(exp, abs=1e-6), (sx, sy)
    cx, cy, w, h, exp = CF.BOUNDARY_STRADDLE
    roi = np.asarray([CF.boundary_roi(variant, cx, cy, w, h)], np.float32)
    assert float(O.roi_align_forward(variant, CF.BOUNDARY_MAP, roi, (1, 1), 1.0, 2).ravel()[0]) == pytest.approx(exp, abs=1e-6)


@pytest.mark.parametrize("variant,nm", DIALECTS)
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0)])
def test_oracle_roi_align_backward_is_the_adjoint(variant, nm, hw, s):
    """<forward(x), g> == <x, backward(g)>: the backward restatement is pinned to the (pinned) forward.  RoIs include
    the boundary cases; ROIAlignRotated_v1's forward-only count >= 1 cannot differ from the backward's count here
    (sizes are floored at 1, so the adaptive grid is never empty)."""
    from tests import inputs as I
    rng = np.random.default_rng(40 + variant)
    N, C, H, W, scale, nO = 2, 8, 20, 24, 0.5, 4
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, 30, extent=W / scale, wh=(2.0, 40.0))
    rois = np.concatenate([I.rois_from_obbs(obbs, rng.integers(0, N, 30)), I.edge_rois(H, W, scale)], 0)
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    g = rng.standard_normal((rois.shape[0], C) + hw).astype(np.float32)
    y = O.roi_align_forward(variant, x, rois, hw, scale, s, nO).astype(np.float64)
    gx = O.roi_align_backward(variant, g, rois, x.shape, scale, s, nO).astype(np.float64)
    lhs, rhs = float((y * g).sum()), float((x.astype(np.float64) * gx).sum())
    assert lhs == pytest.approx(rhs, rel=2e-5, abs=1e-3)


@pytest.mark.parametrize("k,pad,stride,dil,dg", [(3, 1, 1, 1, 1), (3, 2, 2, 2, 2), (1, 0, 1, 1, 1)])
def test_oracle_deform_col2im_is_the_adjoint_and_coord_is_the_derivative(k, pad, stride, dil, dg):
    """dcn_v1.py:L185-306: col2im = adjoint of im2col in the image; col2im_coord = derivative of <im2col, c> in the
    offsets (central differences; bilinear sampling is piecewise linear, so an offset whose +-eps stays inside its
    cell gives the exact derivative -- offsets are drawn away from integer sample positions)."""
    rng = np.random.default_rng(7 * k + pad)
    B, C, H, W = 2, 4, 9, 11
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = (rng.integers(-2, 3, size=(B, dg * 2 * k * k, Ho, Wo)) + rng.uniform(0.2, 0.8, size=(B, dg * 2 * k * k, Ho, Wo))
           ).astype(np.float32)
    args = (k, k, (pad, pad), (stride, stride), (dil, dil), dg)
    col = O.deform_im2col(x, off, *args).astype(np.float64)
    c = rng.standard_normal(col.shape).astype(np.float32)
    gx = O.deform_col2im(c, off, x.shape, *args).astype(np.float64)
    assert float((col * c).sum()) == pytest.approx(float((x.astype(np.float64) * gx).sum()), rel=2e-5, abs=1e-3)
    goff = O.deform_col2im_coord(c, x, off, *args)
    eps = 1.0 / 64
    idx = [tuple(rng.integers(0, n) for n in off.shape) for _ in range(40)]
    for i in idx:
        op, om = off.copy(), off.copy()
        op[i] += eps
        om[i] -= eps
        fd = ((O.deform_im2col(x, op, *args).astype(np.float64) - O.deform_im2col(x, om, *args).astype(np.float64)) * c
              ).sum() / (2 * eps)
        assert float(goff[i]) == pytest.approx(fd, rel=2e-3, abs=2e-3), i
