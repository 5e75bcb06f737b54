"""fp32-MFMA weight gradient of the 3x3 / stride 1 / pad 1 convolution (csrc/conv_wgrad.hip) against the float64
weight gradient autograd derives on the host and, for the deformable form, against the column-matrix path
(deformable im2col + GEMM, ops/dcn_v1.py) whose weight gradient it replaces.

Tolerance: fp32 products, fp32 accumulation over K = N*H*W positions in an order that differs from run to run (the
position chunks meet by float atomics): |err| <= 2e-5 * max|gw| + 1e-6 relative to the gradient scale."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref64(x, gy, cout):
    w = torch.zeros(cout, x.shape[1], 3, 3, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x.double().cpu(), w, None, padding=1)
    (gw,) = torch.autograd.grad(y, w, gy.double().cpu())
    return gw


CASES = [
    # N, H, W, Cin, Cout
    (2, 16, 16, 256, 256),
    (1, 7, 9, 32, 128),         # ragged position count (63: one partial K step)
    (2, 13, 5, 64, 16),         # both tiles narrow
    (1, 32, 32, 256, 8),        # Cout below a tile
    (3, 20, 12, 96, 200),       # channels straddle tiles (96 = 64 + 32, 200 = 128 + 72)
    (1, 6, 6, 48, 40),
    (1, 1, 1, 32, 32),          # a single position: every tap but the centre is padding
    (2, 3, 40, 128, 64),        # rows shorter / longer than a K step
    (1, 128, 128, 256, 256),    # S2ANet P3 tower conv at 1024^2 (one image)
]


@pytest.mark.parametrize("N,H,W,Cin,Cout", CASES)
@pytest.mark.parametrize("ksplit", [0, 1, 3, 8])
def test_wgrad_matches_float64(N, H, W, Cin, Cout, ksplit):
    from jdet_amd.ops import conv_igemm as CI
    if ksplit in (1, 3) and N * H * W > 4096:
        pytest.skip("forced splits are exercised on the small cases")
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    gy = torch.randn(N, Cout, H, W, generator=g)
    gw = CI.conv3x3_wgrad(x.cuda(), gy.cuda(), ksplit=ksplit)
    assert gw.shape == (Cout, Cin, 3, 3)
    ref = _ref64(x, gy, Cout)
    err = (gw.double().cpu() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err


def test_wgrad_accumulates_into_the_given_buffer_and_empty_batch():
    from jdet_amd.ops import conv_igemm as CI
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 10, 6, 64, generator=g).cuda()
    gy = torch.randn(2, 10, 6, 128, generator=g).cuda()
    base = torch.randn(128, 3, 3, 64, generator=g).cuda()
    one = CI.conv3x3_wgrad_nhwc(x, gy)
    acc = base.clone()
    CI.conv3x3_wgrad_nhwc(x, gy, out=acc)
    CI.conv3x3_wgrad_nhwc(x, gy, out=acc)
    assert torch.allclose(acc, base + 2 * one, rtol=1e-5, atol=1e-4)
    assert CI.conv3x3_wgrad_nhwc(x[:0], gy[:0]).abs().max().item() == 0.0


def test_wgrad_unsupported_shapes_raise():
    from jdet_amd.ops import conv_igemm as CI
    x = torch.zeros(1, 4, 4, 30).cuda()
    with pytest.raises(RuntimeError):
        CI.conv3x3_wgrad_nhwc(x, torch.zeros(1, 4, 4, 32).cuda())
    with pytest.raises(ValueError):
        CI.conv3x3_wgrad_nhwc(torch.zeros(1, 4, 4, 32).cuda(), torch.zeros(1, 4, 5, 32).cuda())


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 12, 10, 64, 128), (1, 33, 17, 256, 256), (2, 8, 8, 32, 16)])
def test_deformable_wgrad_matches_the_column_matrix_path(N, H, W, Cin, Cout):
    """gw = gy^T . im2col_deform(x, offset): the column matrix from the pinned sampling kernel (bit-equal to the
    reference's deformable_im2col, tests/test_gpu_reference_kernels.py), the product in float64"""
    from jdet_amd.ops import conv_igemm as CI
    from jdet_amd.ops import dcn_v1
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(N, Cin, H, W, generator=g).cuda()
    gy = torch.randn(N, Cout, H, W, generator=g).cuda()
    offset = (torch.randn(N, 18, H, W, generator=g) * 2.5).cuda()     # many samples leave the map: zero rule exercised
    gw = CI.conv3x3_wgrad(x, gy, offset)
    cols = dcn_v1.deformable_im2col_nhwc(x.permute(0, 2, 3, 1).contiguous(), offset, 3, 3, (1, 1), (1, 1), (1, 1))   # (N*H*W, (tap, c))
    ref = (gy.permute(0, 2, 3, 1).reshape(-1, Cout).double().t() @ cols.double()).view(Cout, 3, 3, Cin)
    err = (gw.permute(0, 2, 3, 1).double() - ref).abs().max().item()
    assert err <= 2e-5 * ref.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("channels_last", [True, False])
def test_shared_tower_weight_gradient_through_autograd(channels_last, monkeypatch):
    """`_ConvBiasAct` (the ConvModule path) with the own weight gradient: one weight used on two maps in one backward
    pass (second use adds into the first use's buffer), then a second pass -- against float64 autograd on the host."""
    from jdet_amd.ops import conv_igemm as CI
    monkeypatch.setattr(CI, "WGRAD", True)
    g = torch.Generator().manual_seed(21)
    w = (torch.randn(64, 32, 3, 3, generator=g) * 0.1).cuda()
    if channels_last:
        w = w.contiguous(memory_format=torch.channels_last)
    w.requires_grad_(True)
    b = torch.randn(64, generator=g).cuda().requires_grad_(True)
    xs = [torch.randn(2, 32, 12, 9, generator=g), torch.randn(1, 32, 5, 30, generator=g)]
    for rounds in range(2):
        w.grad = b.grad = None
        loss, ref_loss = 0, 0
        w64 = w.detach().double().cpu().requires_grad_(True)
        b64 = b.detach().double().cpu().requires_grad_(True)
        for x in xs:
            xc = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
            loss = loss + CI.conv3x3_bias_act(xc, w, b, relu=True).square().sum()
            ref_loss = ref_loss + torch.relu(F.conv2d(x.double(), w64, b64, padding=1)).square().sum()
        loss.backward()
        ref_loss.backward()
        assert (w.grad.double().cpu() - w64.grad).abs().max().item() <= 2e-5 * w64.grad.abs().max().item() + 1e-6
        assert (b.grad.double().cpu() - b64.grad).abs().max().item() <= 2e-5 * b64.grad.abs().max().item() + 1e-6
