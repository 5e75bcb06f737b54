"""GPU: jdet_amd.ops.dcn_v2 (HIP sampling kernels + library GEMMs) against the CPU oracle's operator-level restatement
of ops/dcn_v2.py, forward and all gradients, plus the module interfaces.

Tolerances: sampling arithmetic is the same fp32 expression on both sides; the GEMM reductions differ in order
(library GEMM vs float64 accumulation in the oracle): 2e-5 of the output scale forward, 1e-4 on gradients, which sum
thousands of fp32 products with atomics."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _close(a, b, rel, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, what
    scale = max(np.abs(b).max(), 1e-6)
    assert np.abs(a - b).max() <= rel * scale, (what, np.abs(a - b).max(), scale)


CONV_CASES = [
    # B, C, Cout, H, W, k, pad, stride, dil, dg
    (2, 8, 6, 14, 17, 3, (1, 1), 1, 1, 1),
    (2, 8, 5, 15, 13, 3, (2, 2), 2, 2, 2),
    (1, 4, 3, 9, 9, 1, (0, 0), 1, 1, 1),
    (1, 6, 4, 12, 10, 3, (2, 1), 1, 1, 3),      # asymmetric padding: the reference's pad_h-for-both input gradient
    (2, 64, 64, 32, 32, 3, (1, 1), 1, 1, 2),    # the reference's own smoke shape class (test_conv L1458-1468)
]


@pytest.mark.parametrize("B,C,Cout,H,W,k,pad,stride,dil,dg", CONV_CASES)
def test_dcn_v2_conv_forward_and_gradients(B, C, Cout, H, W, k, pad, stride, dil, dg):
    from jdet_amd.ops import dcn_v2
    rng = np.random.default_rng(B * 100 + C + k)
    Ho = (H + 2 * pad[0] - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad[1] - (dil * (k - 1) + 1)) // stride + 1
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    w = (rng.standard_normal((Cout, C, k, k)) / np.sqrt(C * k * k)).astype(np.float32)
    bias = rng.standard_normal((Cout,)).astype(np.float32)
    off = (rng.standard_normal((B, dg * 2 * k * k, Ho, Wo)) * 2.5).astype(np.float32)    # some samples leave the image
    mask = rng.uniform(0, 1, size=(B, dg * k * k, Ho, Wo)).astype(np.float32)
    g = rng.standard_normal((B, Cout, Ho, Wo)).astype(np.float32)
    t = [torch.from_numpy(v).cuda().requires_grad_(True) for v in (x, off, mask, w, bias)]
    y = dcn_v2.dcn_v2_conv(*t, (stride, stride), pad, (dil, dil), dg)
    y.backward(torch.from_numpy(g).cuda())
    args = (pad, (stride, stride), (dil, dil), dg)
    _close(y.detach().cpu().numpy(), O.dcn_v2_forward(x, off, mask, w, bias, *args), 2e-5, "output")
    for name, got, ref in zip(("grad_input", "grad_offset", "grad_mask", "grad_weight", "grad_bias"),
                              [v.grad.cpu().numpy() for v in t], O.dcn_v2_backward(x, off, mask, w, g, *args)):
        _close(got, ref, 1e-4, name)


def test_dcn_modules_and_registry():
    from jdet_amd.ops import dcn_v2
    from jdet_amd.utils.registry import HEADS
    torch.manual_seed(0)
    x = torch.randn(2, 16, 20, 20, device="cuda", requires_grad=True)
    dcn = HEADS.get("DCN")(16, 12, kernel_size=(3, 3), stride=1, padding=1, deformable_groups=2).cuda()
    y = dcn(x)
    assert y.shape == (2, 12, 20, 20)
    # zero-initialised offset conv: offsets 0, mask sigmoid(0) = 0.5 -> half the plain convolution
    ref = torch.nn.functional.conv2d(x, dcn.weight, None, padding=1) * 0.5 + dcn.bias.view(1, -1, 1, 1)
    assert (y - ref).abs().max().item() < 1e-4
    y.sum().backward()
    assert x.grad is not None and dcn.conv_offset_mask.weight.grad is not None
    v1 = dcn_v2.DeformConv(16, 8, 3, padding=1).cuda()
    assert "bias" not in dict(v1.named_parameters())
    off = torch.zeros(2, 18, 20, 20, device="cuda")
    ref = torch.nn.functional.conv2d(x.detach(), v1.weight, None, padding=1)
    assert (v1(x.detach(), off) - ref).abs().max().item() < 1e-4
    with pytest.raises(ValueError):
        dcn_v2.dcn_v2_conv(x, off[:, :16], torch.ones(2, 9, 20, 20, device="cuda"), v1.weight, None, 1, 1, 1, 1)


POOL_CASES = [
    # R, N, H, W, output_dim, G, P, part, classes, no_trans, spp, trans_std
    (20, 2, 64, 64, 32, 1, 7, 7, 1, True, 4, 0.1),       # test_pool L1470-1508 (plain)
    (20, 2, 64, 64, 32, 1, 7, 7, 1, False, 4, 0.1),      # test_pool (deformable)
    (15, 3, 30, 41, 8, 3, 6, 3, 2, False, 2, 0.2),
    (9, 1, 17, 23, 4, 2, 4, 4, 4, False, 3, 0.3),
]


@pytest.mark.parametrize("R,N,H,W,od,G,P,part,ncls,no_trans,spp,tstd", POOL_CASES)
def test_deform_psroi_pooling_forward_and_gradients(R, N, H, W, od, G, P, part, ncls, no_trans, spp, tstd):
    from jdet_amd.ops import dcn_v2
    rng = np.random.default_rng(R + H + G)
    scale = 0.25
    C = od * G * G
    x = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = rng.integers(0, N, R)
    x1, y1 = rng.uniform(-20, W / scale, R), rng.uniform(-20, H / scale, R)       # some boxes hang over the border
    rois[:, 1], rois[:, 2] = x1, y1
    rois[:, 3], rois[:, 4] = x1 + rng.uniform(0, W / scale / 2, R), y1 + rng.uniform(0, H / scale / 2, R)
    trans = rng.standard_normal((R, 2 * ncls, part, part)).astype(np.float32)
    g = rng.standard_normal((R, od, P, P)).astype(np.float32)
    xt = torch.from_numpy(x).cuda().requires_grad_(True)
    tt = torch.from_numpy(trans).cuda().requires_grad_(True)
    y = dcn_v2.dcn_v2_pooling(xt, torch.from_numpy(rois).cuda(), tt, scale, P, od, no_trans, G, part, spp, tstd)
    y.backward(torch.from_numpy(g).cuda())
    out, cnt = O.deform_psroi_forward(x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd)
    _close(y.detach().cpu().numpy(), out, 2e-6, "output")
    gi, gt = O.deform_psroi_backward(g, cnt, x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd)
    _close(xt.grad.cpu().numpy(), gi, 2e-5, "grad_input")
    if no_trans:
        assert tt.grad is None
    else:
        _close(tt.grad.cpu().numpy(), gt, 1e-4, "grad_trans")


def test_pooling_modules():
    from jdet_amd.ops import dcn_v2
    torch.manual_seed(1)
    x = torch.randn(2, 32, 64, 64, device="cuda")
    rois = torch.tensor([[0, 10, 12, 90, 100], [1, 50, 40, 200, 180], [0, 0, 0, 255, 255]], device="cuda",
                        dtype=torch.float32)
    plain = dcn_v2.DCNv2Pooling(0.25, 7, 32, no_trans=True, trans_std=0.1)
    deform = dcn_v2.DCNv2Pooling(0.25, 7, 32, no_trans=False, trans_std=0.1)
    off = torch.zeros(3, 2, 7, 7, device="cuda")
    assert torch.equal(plain(x, rois, off), deform(x, rois, off))          # zero shift = plain pooling
    full = dcn_v2.DCNPooling(0.25, 7, 32, no_trans=False, trans_std=0.1, deform_fc_dim=64).cuda()
    y = full(x, rois)
    # zero-initialised last layer: offset 0, mask sigmoid(0) = 0.5
    assert torch.allclose(y, plain(x, rois, off) * 0.5, atol=1e-6)
    assert dcn_v2.DCNPooling(0.25, 7, 32, no_trans=True)(x, rois).shape == (3, 32, 7, 7)
    assert plain(x, rois[:0], off[:0]).shape == (0, 32, 7, 7)


def test_dcn_v2_vs_golden(golden):
    """the committed regression vectors (oracle outputs written behind its closed-form pins)"""
    from jdet_amd.ops import dcn_v2
    g = golden("dcn_v2")
    for nm in "abc":
        k, ph, pw, stride, dil, dg = [int(v) for v in g["cfg_" + nm]]
        t = [torch.from_numpy(g[key + "_" + nm]).cuda().requires_grad_(True) for key in ("x", "off", "mask", "w", "bias")]
        y = dcn_v2.dcn_v2_conv(*t, (stride, stride), (ph, pw), (dil, dil), dg)
        y.backward(torch.from_numpy(g["g_" + nm]).cuda())
        _close(y.detach().cpu().numpy(), g["y_" + nm], 2e-5, "y_" + nm)
        for v, key in zip(t, ("gi", "go", "gm", "gw", "gb")):
            _close(v.grad.cpu().numpy(), g[key + "_" + nm], 1e-4, key + "_" + nm)
    od, G, P, part, spp = [int(v) for v in g["ps_cfg"]]
    scale, tstd = [float(v) for v in g["ps_f"]]
    for nm, no_trans in (("plain", True), ("deform", False)):
        xt = torch.from_numpy(g["ps_x"]).cuda().requires_grad_(True)
        tt = torch.from_numpy(g["ps_trans"]).cuda().requires_grad_(True)
        y = dcn_v2.dcn_v2_pooling(xt, torch.from_numpy(g["ps_rois"]).cuda(), tt, scale, P, od, no_trans, G, part, spp, tstd)
        y.backward(torch.from_numpy(g["ps_g"]).cuda())
        _close(y.detach().cpu().numpy(), g["ps_y_" + nm], 2e-6, "ps_y_" + nm)
        _close(xt.grad.cpu().numpy(), g["ps_gi_" + nm], 2e-5, "ps_gi_" + nm)
        if not no_trans:
            _close(tt.grad.cpu().numpy(), g["ps_gt_" + nm], 1e-4, "ps_gt_" + nm)
