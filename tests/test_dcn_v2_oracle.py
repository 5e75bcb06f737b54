"""CPU: pins of the DCN v2 restatement (oracle/jdet_oracle.cpp: jo_dcn_v2_*, jo_deform_psroi_*).  The reference text is
CUDA + cuBLAS only (not buildable here), so the restatement is held to statements that involve neither it nor any
reference build: a mask of ones on the v1 sampling that IS pinned, integer offsets = a shifted convolution scaled by
the mask, adjoint identities of every linear argument, central differences for the offsets, and the affine-map closed
form of the pooling (tests/closed_form.py)."""
import numpy as np
import pytest

from oracle import oracle as O
from tests import closed_form as CF


def _setup(rng, B, C, Cout, H, W, k, pad, stride, dil, dg, frac=True):
    Ho = (H + 2 * pad[0] - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad[1] - (dil * (k - 1) + 1)) // stride + 1
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = rng.standard_normal((Cout, C, k, k)).astype(np.float32)
    bias = rng.standard_normal((Cout,)).astype(np.float32)
    off = rng.integers(-2, 3, size=(B, dg * 2 * k * k, Ho, Wo)).astype(np.float32)
    if frac:
        off = off + rng.uniform(0.2, 0.8, size=off.shape).astype(np.float32)
    mask = rng.uniform(0.1, 1.0, size=(B, dg * k * k, Ho, Wo)).astype(np.float32)
    return x, w, bias, off, mask, Ho, Wo


@pytest.mark.parametrize("k,pad,stride,dil,dg", [(3, 1, 1, 1, 1), (3, 2, 2, 2, 2), (1, 0, 1, 1, 1)])
def test_mask_of_ones_is_the_v1_sampling(k, pad, stride, dil, dg):
    rng = np.random.default_rng(k * 10 + dg)
    x, w, bias, off, mask, Ho, Wo = _setup(rng, 2, 4, 5, 9, 11, k, (pad, pad), stride, dil, dg)
    y = O.dcn_v2_forward(x, off, np.ones_like(mask), w, bias, (pad, pad), (stride, stride), (dil, dil), dg)
    col = O.deform_im2col(x, off, k, k, (pad, pad), (stride, stride), (dil, dil), dg).astype(np.float64)
    ref = (w.reshape(5, -1).astype(np.float64) @ col.reshape(4 * k * k, -1)).reshape(5, 2, Ho, Wo).transpose(1, 0, 2, 3)
    np.testing.assert_allclose(y, ref + bias[None, :, None, None], rtol=0, atol=2e-5)


@pytest.mark.parametrize("k,pad,stride,dil", [(3, 1, 1, 1), (3, 2, 2, 2), (1, 0, 1, 1)])
def test_integer_offsets_are_a_shifted_convolution_scaled_by_the_mask(k, pad, stride, dil):
    rng = np.random.default_rng(k + pad)
    x, w, bias, _, mask, Ho, Wo = _setup(rng, 2, 6, 5, 13, 15, k, (pad, pad), stride, dil, 1)
    dy, dx = rng.integers(-3, 4, size=(k, k)), rng.integers(-3, 4, size=(k, k))
    off = CF.integer_offsets(dy, dx, 2, Ho, Wo)
    y = O.dcn_v2_forward(x, off, mask, w, bias, (pad, pad), (stride, stride), (dil, dil), 1)
    ref = CF.deform_conv_integer_expected(x, w, dy, dx, pad, stride, dil, mask=mask) + bias[None, :, None, None]
    np.testing.assert_allclose(y, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("k,pad,stride,dil,dg", [(3, (1, 1), 1, 1, 1), (3, (2, 2), 2, 2, 2), (1, (0, 0), 1, 1, 1)])
def test_backward_is_the_adjoint_in_every_linear_argument(k, pad, stride, dil, dg):
    """y is linear in x, in the weight, in the mask and in the bias: <y(arg) - y(0), g> = <arg, grad_arg>"""
    rng = np.random.default_rng(100 + k + dg)
    x, w, bias, off, mask, Ho, Wo = _setup(rng, 2, 4, 3, 9, 11, k, pad, stride, dil, dg)
    g = rng.standard_normal((2, 3, Ho, Wo)).astype(np.float32)
    args = (pad, (stride, stride), (dil, dil), dg)
    zero_b = np.zeros_like(bias)
    y = O.dcn_v2_forward(x, off, mask, w, zero_b, *args).astype(np.float64)
    gi, goff, gm, gw, gb = O.dcn_v2_backward(x, off, mask, w, g, *args)
    lhs = float((y * g).sum())
    for name, arg, grad in (("input", x, gi), ("weight", w, gw), ("mask", mask, gm)):
        assert lhs == pytest.approx(float((arg.astype(np.float64) * grad).sum()), rel=3e-5, abs=2e-3), name
    np.testing.assert_allclose(gb, g.astype(np.float64).sum((0, 2, 3)), rtol=1e-5, atol=1e-4)
    # offsets: central differences, offsets drawn away from integer sample positions (piecewise-linear sampling)
    eps = 1.0 / 64
    for _ in range(30):
        i = tuple(int(rng.integers(0, n)) for n in off.shape)
        op, om = off.copy(), off.copy()
        op[i] += eps
        om[i] -= eps
        fd = ((O.dcn_v2_forward(x, op, mask, w, zero_b, *args).astype(np.float64)
               - O.dcn_v2_forward(x, om, mask, w, zero_b, *args).astype(np.float64)) * g).sum() / (2 * eps)
        assert float(goff[i]) == pytest.approx(fd, rel=3e-3, abs=3e-3), i


def test_input_gradient_uses_pad_h_on_both_axes():
    """dcn_v2.py:L651-653 passes (pad_h, pad_h) to the col2im kernel: with pad_h != pad_w the reference's input
    gradient is NOT the adjoint; it equals the adjoint of the operator whose x offsets are shifted by pad_w - pad_h"""
    rng = np.random.default_rng(5)
    pad = (2, 1)
    x, w, bias, off, mask, Ho, Wo = _setup(rng, 1, 3, 2, 8, 9, 3, pad, 1, 1, 1)
    g = rng.standard_normal((1, 2, Ho, Wo)).astype(np.float32)
    gi = O.dcn_v2_backward(x, off, mask, w, g, pad, (1, 1), (1, 1), 1)[0]
    off2 = off.copy()
    off2[:, 1::2] += pad[1] - pad[0]
    y2 = O.dcn_v2_forward(x, off2, mask, w, np.zeros_like(bias), pad, (1, 1), (1, 1), 1).astype(np.float64)
    assert float((y2 * g).sum()) == pytest.approx(float((x.astype(np.float64) * gi).sum()), rel=3e-5, abs=2e-3)
    y = O.dcn_v2_forward(x, off, mask, w, np.zeros_like(bias), pad, (1, 1), (1, 1), 1).astype(np.float64)
    assert abs(float((y * g).sum()) - float((x.astype(np.float64) * gi).sum())) > 1e-2


# ------------------------------------------------------------------------------------------------ PSRoI pooling
def _psroi_case(rng, R, N, H, W, scale, output_dim, G, P, part, ncls, no_trans):
    C = output_dim * G * G
    feat, a, b, d = CF.affine_map(rng, N, C, H, W)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rng.integers(0, N, R)
    # boxes well inside the image so that no sample (incl. the trans shift) is skipped or clamped
    cx, cy = rng.uniform(0.35, 0.65, R) * W / scale, rng.uniform(0.35, 0.65, R) * H / scale
    bw, bh = rng.uniform(2, 0.3 * W, R) / scale, rng.uniform(2, 0.3 * H, R) / scale
    rois[:, 1], rois[:, 2], rois[:, 3], rois[:, 4] = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
    trans = None if no_trans else rng.uniform(-1, 1, size=(R, 2 * ncls, part, part)).astype(np.float32)
    return feat, (a, b, d), rois, trans


@pytest.mark.parametrize("G,P,part,ncls,no_trans,spp", [(1, 7, 7, 1, True, 4), (1, 7, 7, 1, False, 4),
                                                        (3, 6, 3, 2, False, 2), (2, 4, 4, 4, False, 3)])
def test_psroi_pooling_on_affine_map(G, P, part, ncls, no_trans, spp):
    rng = np.random.default_rng(G * 10 + P)
    output_dim, N, H, W, scale, trans_std = 4, 2, 40, 48, 0.25, 0.1
    feat, (a, b, d), rois, trans = _psroi_case(rng, 12, N, H, W, scale, output_dim, G, P, part, ncls, no_trans)
    out, cnt = O.deform_psroi_forward(feat, rois, trans, no_trans, scale, output_dim, G, P, part, spp, trans_std)
    assert (cnt == spp * spp).all()
    bi = rois[:, 0].astype(int)
    for n in range(rois.shape[0]):
        exp, _ = CF.psroi_expected((a[bi[n]], b[bi[n]], d[bi[n]]), rois[n:n + 1],
                                   None if no_trans else trans[n:n + 1], scale, output_dim, G, P, part, spp, trans_std)
        np.testing.assert_allclose(out[n], exp[0], rtol=0, atol=3e-4)


@pytest.mark.parametrize("G,P,part,ncls,spp", [(1, 7, 7, 1, 4), (3, 6, 3, 2, 2)])
def test_psroi_backward_adjoint_and_trans_gradient(G, P, part, ncls, spp):
    rng = np.random.default_rng(G + P)
    output_dim, N, H, W, scale, trans_std = 4, 2, 40, 48, 0.25, 0.1
    feat, (a, b, d), rois, trans = _psroi_case(rng, 10, N, H, W, scale, output_dim, G, P, part, ncls, False)
    g = rng.standard_normal((10, output_dim, P, P)).astype(np.float32)
    out, cnt = O.deform_psroi_forward(feat, rois, trans, False, scale, output_dim, G, P, part, spp, trans_std)
    gi, gt = O.deform_psroi_backward(g, cnt, feat, rois, trans, False, scale, output_dim, G, P, part, spp, trans_std)
    # linear in the input: adjoint (on a random map, same sampling geometry)
    x2 = rng.standard_normal(feat.shape).astype(np.float32)
    out2, cnt2 = O.deform_psroi_forward(x2, rois, trans, False, scale, output_dim, G, P, part, spp, trans_std)
    gi2, _ = O.deform_psroi_backward(g, cnt2, x2, rois, trans, False, scale, output_dim, G, P, part, spp, trans_std)
    assert float((out2.astype(np.float64) * g).sum()) == pytest.approx(
        float((x2.astype(np.float64) * gi2).sum()), rel=3e-5, abs=2e-3)
    # affine map: d out / d trans is the closed form; grad_trans sums it over the outputs that share a trans cell
    bi = rois[:, 0].astype(int)
    exp = np.zeros_like(gt, dtype=np.float64)
    cec = output_dim // ncls
    for n in range(rois.shape[0]):
        _, dtr = CF.psroi_expected((a[bi[n]], b[bi[n]], d[bi[n]]), rois[n:n + 1], trans[n:n + 1], scale, output_dim,
                                   G, P, part, spp, trans_std)
        for ct in range(output_dim):
            for ph in range(P):
                for pw in range(P):
                    p_h, p_w = int(np.floor(ph / P * part)), int(np.floor(pw / P * part))
                    exp[n, 2 * (ct // cec), p_h, p_w] += dtr[0, ct, ph, pw, 0] * g[n, ct, ph, pw]
                    exp[n, 2 * (ct // cec) + 1, p_h, p_w] += dtr[0, ct, ph, pw, 1] * g[n, ct, ph, pw]
    np.testing.assert_allclose(gt, exp, rtol=2e-4, atol=2e-3)


def test_psroi_boundary_rules():
    """samples beyond [-0.5, W - 0.5] are skipped (count drops), samples inside the half-pixel rim are clamped to the
    border pixel; a bin with no counted sample returns 0 and passes no gradient"""
    H = W = 8
    feat = np.arange(H * W, dtype=np.float32).reshape(1, 1, H, W)
    rois = np.asarray([[0, -8, -8, 7, 7],       # top-left corner partly outside
                       [0, 40, 40, 60, 60]], np.float32)   # entirely outside (scale 1)
    out, cnt = O.deform_psroi_forward(feat, rois, None, True, 1.0, 1, 1, 2, 2, 2, 0.0)
    assert cnt[1].max() == 0 and (out[1] == 0).all()
    # frame [-8.5, 7.5), bins of 8, samples every 4: bin 0 samples at -8.5, -4.5 (skipped); bin 1 at -0.5 (counted,
    # clamped to 0) and 3.5: value = mean of 8 y + x over {0, 3.5}^2 = 15.75
    assert cnt[0, 0, 0, 0] == 0 and cnt[0, 0, 0, 1] == 0 and cnt[0, 0, 1, 1] == 4
    assert out[0, 0, 1, 1] == pytest.approx(15.75, abs=1e-5)
    g = np.ones_like(out)
    gi, _ = O.deform_psroi_backward(g, cnt, feat, rois, None, True, 1.0, 1, 1, 2, 2, 2, 0.0)
    assert gi.sum() == pytest.approx(float((cnt > 0).sum()), abs=1e-5)   # each counted bin spreads exactly its 1.0
