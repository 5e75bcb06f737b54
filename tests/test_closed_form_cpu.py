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
