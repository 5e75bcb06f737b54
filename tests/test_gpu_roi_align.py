"""GPU parity: HIP RoIAlign family (through the C ABI / jdet_amd.ops) vs golden fixtures and the
CPU oracle.  Forward tolerances (inputs ~N(0,1)): reference-order arithmetic (mode 1) = 0 ulp (same operation order,
contraction off); merged-tap arithmetic (default) = 2e-6 abs.
Backward: 2e-5 abs (summation order of the gather / atomics)."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu

FWD_ATOL = 0.0           # the *_reference entry points (set_arithmetic("reference"))
FWD_MERGED_ATOL = 2e-6   # default mode
BWD_ATOL = 2e-5


PATHS = [("roi", 1, FWD_ATOL), ("roi", 0, FWD_MERGED_ATOL), ("roi_cl", 1, FWD_ATOL), ("roi_cl", 0, FWD_MERGED_ATOL)]


@pytest.fixture(params=PATHS, ids=["roi-reforder", "roi-merged", "roicl-reforder", "roicl-merged"],
                autouse=True)
def fwd_mode(request):
    """every forward path: (R,C,PH,PW)-contiguous / channels-last result x reference-order / merged-tap arithmetic.
    Yields the forward tolerance of the path."""
    from jdet_amd import _lib as L
    from jdet_amd.ops import _roi_common as RC
    path, mode, atol = request.param
    prev_path = RC.set_forward_path(path)
    prev = RC.set_arithmetic("reference" if mode == 1 else "merged")
    yield atol
    RC.set_arithmetic(prev)
    RC.set_forward_path(prev_path)


def _layer(variant, hw, scale, s, nO=8):
    from jdet_amd.ops.riroi_align import RiRoIAlign
    from jdet_amd.ops.roi_align import ROIAlign
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    from jdet_amd.ops.roi_align_rotated_v1 import ROIAlignRotated_v1
    if variant == O.V_ROT:
        return ROIAlignRotated(hw, scale, s)
    if variant == O.V_ROT_V1:
        return ROIAlignRotated_v1(hw, scale, s)
    if variant == O.V_RI:
        return RiRoIAlign(hw, scale, s, nO)
    return ROIAlign(hw, scale, s, version=1 if variant == O.V_HBB1 else 0)


def _run(variant, feat, rois, hw, scale, s, grad, dev, channels_last, nO=8):
    x = torch.from_numpy(feat).to(dev)
    if channels_last:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    y = _layer(variant, hw, scale, s, nO)(x, torch.from_numpy(rois).to(dev))
    y.backward(torch.from_numpy(grad).to(dev))
    torch.cuda.synchronize()
    return y.detach().cpu().numpy(), x.grad.detach().cpu().contiguous().numpy()


@pytest.mark.parametrize("variant,nm", [(O.V_ROT, "rot"), (O.V_ROT_V1, "rot_v1"), (O.V_HBB0, "hbb0"), (O.V_HBB1, "hbb1")])
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0), ((2, 2), 3)])
@pytest.mark.parametrize("channels_last", [False, True])
def test_roi_align_vs_golden(golden, dev, variant, nm, hw, s, channels_last, fwd_mode):
    g = golden("roi_align")
    rois = g["hrois"] if variant in (O.V_HBB0, O.V_HBB1) else g["rois"]
    key = "%s_%dx%d_s%d" % (nm, hw[0], hw[1], s)
    y, gi = _run(variant, g["feat"], rois, hw, float(g["scale"]), s, g["g_" + key], dev, channels_last)
    np.testing.assert_allclose(y, g["y_" + key], rtol=0, atol=fwd_mode)
    np.testing.assert_allclose(gi, g["gi_" + key], rtol=0, atol=BWD_ATOL)


@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0)])
def test_riroi_align_vs_golden(golden, dev, hw, s, fwd_mode):
    g = golden("riroi_align")
    key = "ri_%dx%d_s%d" % (hw[0], hw[1], s)
    y, gi = _run(O.V_RI, g["feat"], g["rois"], hw, float(g["scale"]), s, g["g_" + key], dev, True, int(g["nO"]))
    np.testing.assert_allclose(y, g["y_" + key], rtol=0, atol=fwd_mode)
    np.testing.assert_allclose(gi, g["gi_" + key], rtol=0, atol=BWD_ATOL)


@pytest.mark.parametrize("variant", [O.V_ROT, O.V_ROT_V1])
@pytest.mark.parametrize("C", [256, 260, 5, 512])
def test_rotated_vs_oracle_random(dev, variant, C, fwd_mode):
    """seeded random maps incl. channel counts off the 4-vector / 256-chunk fast path"""
    rng = np.random.default_rng(7 + C)
    N, H, W, scale = 2, 40, 48, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, 40, extent=W / scale, wh=(4.0, 160.0))
    rois = np.concatenate([I.rois_from_obbs(obbs, rng.integers(0, N, 40)), I.edge_rois(H, W, scale)], 0)
    for hw, s in (((7, 7), 2), ((7, 7), 0)):
        grad = rng.standard_normal((rois.shape[0], C) + hw).astype(np.float32)
        y, gi = _run(variant, feat, rois, hw, scale, s, grad, dev, True)
        np.testing.assert_allclose(y, O.roi_align_forward(variant, feat, rois, hw, scale, s), rtol=0, atol=fwd_mode)
        ref = O.roi_align_backward(variant, grad, rois, feat.shape, scale, s)
        np.testing.assert_allclose(gi, ref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(ref).max()))


def test_cfg0_micro_vs_oracle(dev, fwd_mode):
    """BASELINE configs[0]: 1x256x256x256 fmap, 512 random OBBs, 7x7, sampling 2, scale 0.25."""
    rng = np.random.default_rng(0)
    feat = rng.standard_normal((1, 256, 256, 256)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 512), np.zeros(512))
    grad = rng.standard_normal((512, 256, 7, 7)).astype(np.float32)
    y, gi = _run(O.V_ROT, feat, rois, (7, 7), 0.25, 2, grad, dev, True)
    O.set_threads(8)
    np.testing.assert_allclose(y, O.roi_align_forward(O.V_ROT, feat, rois, (7, 7), 0.25, 2), rtol=0, atol=fwd_mode)
    ref = O.roi_align_backward(O.V_ROT, grad, rois, feat.shape, 0.25, 2)
    # measured (round 5): max |err| 2.9e-6 on gradients up to 7.1 (mean 0.16) -- summation order of up to ~40 entries per
    # pixel, fp32; the bar is the backward tolerance of the rest of this file, 2e-5 x the gradient scale
    np.testing.assert_allclose(gi, ref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("variant", [O.V_ROT, O.V_ROT_V1, O.V_HBB0])
def test_channels_last_result_and_gradient(dev, variant, fwd_mode):
    """the channels-last paths return the (R,C,PH,PW) tensor with channels-last strides; a channels-last gradient goes to
    jdet_roi_align_backward_cl (no transpose pass) and must equal the oracle's scatter"""
    from jdet_amd.ops import _roi_common as RC
    rng = np.random.default_rng(21 + variant)
    N, C, H, W, scale = 2, 64, 40, 48, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, 80, extent=W / scale, wh=(4.0, 160.0)),
                                            rng.integers(0, N, 80)), I.edge_rois(H, W, scale)], 0)
    if variant == O.V_HBB0:
        rois = I.obb_to_hbb_rois(rois)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = _layer(variant, (7, 7), scale, 2)(x, torch.from_numpy(rois).to(dev))
    assert y.shape == (rois.shape[0], C, 7, 7)
    if RC._FORWARD_PATH[0] != "roi":
        assert y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous()
    grad = rng.standard_normal(y.shape).astype(np.float32)
    y.backward(torch.from_numpy(grad).to(dev).contiguous(memory_format=torch.channels_last))
    np.testing.assert_allclose(y.detach().cpu().numpy(), O.roi_align_forward(variant, feat, rois, (7, 7), scale, 2),
                               rtol=0, atol=fwd_mode)
    ref = O.roi_align_backward(variant, grad, rois, feat.shape, scale, 2)
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("nO,C", [(8, 264), (4, 36), (8, 16), (2, 12)])
@pytest.mark.parametrize("hw,s", [((7, 7), 2), ((3, 5), 0)])
def test_riroi_vector_path_vs_oracle(dev, nO, C, hw, s):
    """RiRoIAlign on the vector kernels: orientation planes mixed in registers (nO = 8: a plane group spans a lane
    pair; nO = 4: one lane), 264 channels = two channel chunks with a partial second one; nO = 2 stays on the scalar
    kernel.  Forward bit-identical to the oracle (reference accumulation order), backward (mixed rows + sorted gather
    for fixed sampling, atomics for adaptive) to accumulation-order tolerance.  Angles cover every orientation index
    incl. negative theta."""
    rng = np.random.default_rng(100 * nO + C)
    N, H, W, scale, R = 2, 28, 36, 0.25, 60
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    obbs = I.random_obbs(rng, R, extent=W / scale, wh=(6.0, 120.0))
    obbs[:, 4] = np.linspace(-2 * np.pi, 2 * np.pi, R).astype(np.float32)
    rois = I.rois_from_obbs(obbs, rng.integers(0, N, R))
    grad = rng.standard_normal((R, C) + hw).astype(np.float32)
    from jdet_amd import _lib as L
    ref = O.roi_align_forward(O.V_RI, feat, rois, hw, scale, s, nO)
    gref = O.roi_align_backward(O.V_RI, grad, rois, feat.shape, scale, s, nO)
    from jdet_amd.ops import _roi_common as RC
    prev = RC.set_arithmetic("reference")                # reference accumulation order: bit-identical
    try:
        y, gi = _run(O.V_RI, feat, rois, hw, scale, s, grad, dev, True, nO)
    finally:
        RC.set_arithmetic(prev)
    assert np.array_equal(y, ref)
    np.testing.assert_allclose(gi, gref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(gref).max()))
    # default mode: merged taps, one orientation mix per bin, channels-last result and gradient
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y2 = _layer(O.V_RI, hw, scale, s, nO)(x, torch.from_numpy(rois).to(dev))
    y2.backward(torch.from_numpy(grad).to(dev).contiguous(memory_format=torch.channels_last))
    np.testing.assert_allclose(y2.detach().cpu().numpy(), ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(x.grad.cpu().numpy(), gref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(gref).max()))


def test_backward_cl_kept_workspace_contract(dev):
    """jdet_roi_align_backward_cl with workspace_clean=1 on ONE kept workspace, three different RoI sets in a row:
    each result equals the oracle and the first jdet_roi_align_backward_clean_bytes() bytes are zero again after every
    call; workspace_clean=0 on a workspace full of garbage gives the same result.  Odd map sizes: the 2x2 patch /
    XCD stripe mapping of the gather has to cover partial patches and stripes."""
    from jdet_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(5)
    N, C, H, W, scale, R = 2, 32, 37, 45, 0.25, 150
    wsb = lib.jdet_roi_align_backward_workspace(O.V_ROT, R, N, C, H, W, 7, 7, 2)
    clean = lib.jdet_roi_align_backward_clean_bytes(O.V_ROT, R, N, C, H, W, 7, 7, 2)
    assert 0 < clean < wsb
    ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)
    gin = torch.empty((N, H, W, C), device=dev)
    for rep in range(3):
        rois = I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(4.0, 160.0)), rng.integers(0, N, R))
        grad = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
        live = np.ones(R, dtype=bool)
        live[::11] = False
        ref = O.roi_align_backward(O.V_ROT, grad[live], rois[live], (N, C, H, W), scale, 2)
        rois[~live, 0] = -1         # masked RoIs (padding rows of the fixed-shape heads) contribute nothing
        g_cl = torch.from_numpy(np.ascontiguousarray(grad.transpose(0, 2, 3, 1))).to(dev)     # (R, PH, PW, C)
        r = torch.from_numpy(rois).to(dev)
        tol = BWD_ATOL * max(1.0, np.abs(ref).max())
        for ws_call, flag in ((ws, 1), (torch.full((wsb,), 0xA5, dtype=torch.uint8, device=dev), 0)):
            gin.fill_(float("nan"))
            L.check(lib.jdet_roi_align_backward_cl(O.V_ROT, L.ptr(g_cl), L.ptr(r), R, N, C, H, W, 7, 7, scale, 2, 1,
                                                   L.ptr(gin), L.ptr(ws_call), wsb, flag, L.stream_ptr(gin)), "bwd_cl")
            got = gin.permute(0, 3, 1, 2).cpu().numpy()
            np.testing.assert_allclose(got, ref, rtol=0, atol=tol)
            assert int(ws_call[:clean].max()) == 0


def test_backward_rows_past_the_direct_cap(dev):
    """Patches whose entry count exceeds the direct rows' cap (csr_gather.h: at most 256 entries of a patch have a fixed
    home; a tiny RoI puts up to 49 entries -- one per bin -- into one 2x2 patch): 80 near-identical 5-pixel RoIs on top of
    random ones send several thousand entries through the overflow CSR.  Both entry points, the kept workspace handed
    back clean, then a call WITHOUT overflow on the same kept workspace (the overflow total was reset)."""
    from jdet_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(11)
    N, C, H, W, scale = 1, 16, 40, 48, 0.25
    tiny = np.tile(np.array([[0, 61.0, 83.0, 5.0, 6.0, 0.3]], np.float32), (80, 1))
    tiny[:, 1:3] += rng.uniform(-0.5, 0.5, size=(80, 2)).astype(np.float32)
    other = I.rois_from_obbs(I.random_obbs(rng, 70, extent=W / scale, wh=(8.0, 120.0)), np.zeros(70))
    rois = np.concatenate([tiny, other]).astype(np.float32)
    R = rois.shape[0]
    grad = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
    ref = O.roi_align_backward(O.V_ROT, grad, rois, (N, C, H, W), scale, 2)
    tol = BWD_ATOL * max(1.0, np.abs(ref).max())
    wsb = lib.jdet_roi_align_backward_workspace(O.V_ROT, R, N, C, H, W, 7, 7, 2)
    g_nchw, r_dev = torch.from_numpy(grad).to(dev), torch.from_numpy(rois).to(dev)
    scratch = torch.full((wsb,), 0x5A, dtype=torch.uint8, device=dev)
    gin0 = torch.full((N, H, W, C), float("nan"), device=dev)
    L.check(lib.jdet_roi_align_backward(O.V_ROT, L.ptr(g_nchw), L.ptr(r_dev), R, N, C, H, W, 7, 7, scale, 2, 1, None,
                                        L.ptr(gin0), L.ptr(scratch), wsb, L.stream_ptr(gin0)), "bwd")
    np.testing.assert_allclose(gin0.permute(0, 3, 1, 2).cpu().numpy(), ref, rtol=0, atol=tol)
    clean = lib.jdet_roi_align_backward_clean_bytes(O.V_ROT, R, N, C, H, W, 7, 7, 2)
    ws = torch.zeros((wsb,), dtype=torch.uint8, device=dev)
    gin = torch.empty((N, H, W, C), device=dev)
    for rr, gg, want in ((rois, grad, ref), (rois[80:], grad[80:], None), (rois, grad, ref)):
        if want is None:
            want = O.roi_align_backward(O.V_ROT, gg, rr, (N, C, H, W), scale, 2)
        g_cl = torch.from_numpy(np.ascontiguousarray(gg.transpose(0, 2, 3, 1))).to(dev)
        r = torch.from_numpy(np.ascontiguousarray(rr)).to(dev)
        gin.fill_(float("nan"))
        # (the workspace was sized for the larger R: a smaller call on a kept workspace is the FPN heads' normal case)
        L.check(lib.jdet_roi_align_backward_cl(O.V_ROT, L.ptr(g_cl), L.ptr(r), rr.shape[0], N, C, H, W, 7, 7, scale, 2, 1,
                                               L.ptr(gin), L.ptr(ws), wsb, 1, L.stream_ptr(gin)), "bwd_cl")
        np.testing.assert_allclose(gin.permute(0, 3, 1, 2).cpu().numpy(), want, rtol=0,
                                   atol=BWD_ATOL * max(1.0, np.abs(want).max()))
        assert int(ws[:clean].max()) == 0


def test_full_size_properties(dev):
    """north-star size (1024x1024 tile -> 256x256x256 map, 2000 RoIs): size-independent properties.
    (a) linearity in the feature map, (b) a constant map pools to the constant for RoIs whose samples
    are all inside, (c) backward of ones conserves mass: sum(grad_in) == #valid-sample-weight."""
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    rng = np.random.default_rng(1)
    R = 2000
    rois_np = I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))
    rois = torch.from_numpy(rois_np).to(dev)
    layer = ROIAlignRotated(7, 0.25, 2)
    a = torch.randn(1, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.randn(1, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    ya, yb, yab = layer(a, rois), layer(b, rois), layer(2.0 * a - 0.5 * b, rois)
    assert torch.allclose(yab, 2.0 * ya - 0.5 * yb, atol=2e-5)
    ones = torch.ones(1, 8, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    ones.requires_grad_(True)
    yo = layer(ones, rois)
    # RoIs fully inside the map (centre +- half diagonal) must pool to exactly ~1
    half = 0.5 * np.hypot(rois_np[:, 3], rois_np[:, 4])
    inside = (rois_np[:, 1] - half > 4) & (rois_np[:, 1] + half < 1016) & (rois_np[:, 2] - half > 4) & (rois_np[:, 2] + half < 1016)
    assert inside.sum() > 100
    assert torch.allclose(yo[torch.from_numpy(inside).to(dev)], torch.ones((), device=dev), atol=1e-5)
    yo.sum().backward()
    # bilinear weights of a valid sample sum to 1, so sum(grad_in) == sum(forward(ones))
    assert abs(ones.grad.sum().item() - yo.sum().item()) < 1e-3 * yo.sum().item()


def test_empty_and_errors(dev):
    from jdet_amd.ops.roi_align_rotated import ROIAlignRotated
    x = torch.randn(1, 8, 16, 16, device=dev, requires_grad=True)
    y = ROIAlignRotated(7, 0.25, 2)(x, torch.zeros((0, 6), device=dev))
    assert y.shape == (0, 8, 7, 7)
    y.sum().backward()
    assert x.grad.abs().sum().item() == 0
    with pytest.raises(AssertionError):
        ROIAlignRotated(7, 0.25, 2)(x, torch.zeros((3, 5), device=dev))


def test_layout_helpers(dev):
    from jdet_amd import _lib as L
    x = torch.randn(3, 37, 19, 23, device=dev)
    y = torch.empty(3, 19, 23, 37, device=dev)
    L.check(L.lib().jdet_nchw_to_nhwc(x.data_ptr(), 3, 37, 19, 23, y.data_ptr(), L.stream_ptr(x)), "t")
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())
    z = torch.empty_like(x)
    L.check(L.lib().jdet_nhwc_to_nchw(y.data_ptr(), 3, 37, 19, 23, z.data_ptr(), L.stream_ptr(x)), "t")
    assert torch.equal(z, x)


@pytest.mark.parametrize("R,N", [(1, 1), (7, 1), (64, 2), (2000, 1), (2003, 3), (9001, 2), (20000, 16)])
def test_spatial_order_is_a_permutation(dev, R, N):
    """jdet_roi_spatial_order is a pure scheduling hint: it must emit a permutation of [0,R) for any R
    (LDS path R <= 8192, global-scratch path above), any image count, 6- and 5-column RoIs."""
    from jdet_amd.ops._roi_common import spatial_order
    rng = np.random.default_rng(R)
    rois = I.rois_from_obbs(I.random_obbs(rng, R), rng.integers(0, N, R))
    o = spatial_order(torch.from_numpy(rois).to(dev), 0.25, N, 256, 256).cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(R))
    h = I.obb_to_hbb_rois(rois)
    o5 = spatial_order(torch.from_numpy(h).to(dev), 0.25, N, 256, 256).cpu().numpy()
    assert np.array_equal(np.sort(o5), np.arange(R))
    if R >= 2000 and N == 1:
        # consecutive workgroups of one XCD (b, b+8) are spatial neighbours: mean centre distance far
        # below that of a random order
        c = rois[o][:, 1:3]
        d_sched = np.linalg.norm(c[8:] - c[:-8], axis=1).mean()
        d_rand = np.linalg.norm(rois[8:, 1:3] - rois[:-8, 1:3], axis=1).mean()
        assert d_sched < 0.25 * d_rand


def _fwd_cl_both(variant, x, rois, hw, scale, nO, rois_legacy=None):
    """(channel-sliced product path, RoI-stationary kernels of rounds 1-3) on the same inputs, merged-tap arithmetic.
    Output buffers start as NaN so that rows a kernel skips (masked RoIs) stay recognisable."""
    if rois_legacy is None:
        rois_legacy = rois
    from jdet_amd import _lib as L
    lib = L.lib()
    N, C, H, W = x.shape
    R = rois.shape[0]
    outs = []
    for sliced in (True, False):
        out = torch.full((R, C) + hw, float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
        if sliced:
            from jdet_amd import _experimental as X        # the channel-sliced kernels: libjdet_experimental.so
            xl = X.lib()
            wsb = xl.jdet_roi_align_forward_cl_mode_workspace(2, R, hw[0], hw[1])
            ws = torch.empty((wsb,), dtype=torch.uint8, device=x.device)
            L.check(xl.jdet_roi_align_forward_cl_mode(2, variant, x.data_ptr(), N, C, H, W, rois.data_ptr(), R, hw[0],
                                                      hw[1], scale, 2, nO, None, out.data_ptr(), ws.data_ptr(), wsb,
                                                      L.stream_ptr(x)), "fwd_cl_mode 2")
        else:
            L.check(lib.jdet_roi_align_forward_cl_roi(variant, x.data_ptr(), N, C, H, W, rois_legacy.data_ptr(), R, hw[0],
                                                      hw[1], scale, 2, nO, None, out.data_ptr(), L.stream_ptr(x)),
                    "fwd_cl_roi")
        outs.append(out)
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("variant,nO,C", [(O.V_ROT, 1, 256), (O.V_ROT, 1, 96), (O.V_ROT_V1, 1, 64), (O.V_HBB0, 1, 32),
                                          (O.V_HBB1, 1, 128), (O.V_RI, 8, 64), (O.V_RI, 4, 32)])
@pytest.mark.parametrize("hw", [(7, 7), (4, 4), (5, 9)])
def test_sliced_forward_equals_roi_stationary_kernels(dev, variant, nO, C, hw):
    """jdet_roi_align_forward_cl in forward mode 2 (channel-sliced: plan + 8-lane groups per item) runs the same
    geometry functions, the same tap merge and the same fma chain as the RoI-stationary merged-tap kernel: bit-equal.
    Covers several images, masked RoIs (negative batch index: rows untouched), RoIs hanging over the map border, a RoI
    of an image that does not exist (zeros), R not a multiple of anything, slices counts 1 / 2 / 3 / 4 / 8."""
    from jdet_amd import _lib as L
    if fwd_mode_is_reference():
        pytest.skip("merged-tap arithmetic only")
    rng = np.random.default_rng(100 + variant * 7 + C + hw[0])
    N, H, W, scale = 3, 40, 56, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    R = 203
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(4.0, 200.0)),
                                            rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
    rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0        # masked
    rois[5, 0] = 7.0                                        # no such image
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    rois_l = rois.copy()
    rois_l[5, 0] = -1.0      # (the RoI-stationary kernel would read image 7 of 3 through its per-image descriptor)
    a, b = _fwd_cl_both(variant, x, r, hw, scale, nO, torch.from_numpy(rois_l).to(dev))
    a, b = a.cpu().numpy(), b.cpu().numpy()
    masked = rois[:, 0] < 0
    assert np.isnan(a[masked]).all() and np.isnan(b[masked]).all()
    assert not np.isnan(a[~masked]).any()
    assert np.array_equal(a[5], np.zeros_like(a[5]))
    keep = ~masked
    keep[5] = False
    assert np.array_equal(a[keep], b[keep])
    ref = O.roi_align_forward(variant, feat, rois[keep], hw, scale, 2, nO) if variant == O.V_RI else \
        O.roi_align_forward(variant, feat, rois[keep], hw, scale, 2)
    np.testing.assert_allclose(a[keep], ref, rtol=0, atol=FWD_MERGED_ATOL)


def fwd_mode_is_reference():
    from jdet_amd.ops import _roi_common as RC
    return RC._ARITHMETIC[0] == "reference"


def test_sliced_forward_north_star_equals_roi_stationary(dev):
    """the roofline shape: 1 x 256 x 256 x 256 map, 2000 RoIs, 7 x 7: sliced == RoI-stationary bit for bit, and a
    second call on a dirty workspace gives the same result (the workspace carries no state between calls)"""
    if fwd_mode_is_reference():
        pytest.skip("merged-tap arithmetic only")
    rng = np.random.default_rng(1)
    R = 2000
    rois = torch.from_numpy(I.rois_from_obbs(I.random_obbs(rng, R), np.zeros(R))).to(dev)
    x = torch.randn(1, 256, 256, 256, device=dev).contiguous(memory_format=torch.channels_last)
    a, b = _fwd_cl_both(O.V_ROT, x, rois, (7, 7), 0.25, 1)
    assert torch.equal(a, b)
    a2, _ = _fwd_cl_both(O.V_ROT, x, rois, (7, 7), 0.25, 1)
    assert torch.equal(a, a2)


@pytest.mark.parametrize("variant,C", [(O.V_ROT, 256), (O.V_ROT, 96), (O.V_ROT_V1, 64), (O.V_HBB0, 32), (O.V_HBB1, 128)])
@pytest.mark.parametrize("hw", [(7, 7), (4, 4), (5, 8)])
def test_line_forward_matches_the_oracle(dev, variant, C, hw):
    """The line kernel (csrc/experimental/roi_align_line.h, libjdet_experimental.so): every distinct pixel row of a LINE of bins loaded once, per-bin weights
    from an LDS table.  Same geometry and the same within-bin merge as mode 0; a bin's sum runs in another order, so
    the bar is the merged-tap tolerance against the oracle (and the same distance from mode 0).  Several images, masked
    RoIs (rows untouched), RoIs over the border, thin and fat RoIs (lines along rows / along columns)."""
    from jdet_amd import _lib as L
    if fwd_mode_is_reference():
        pytest.skip("merged-tap arithmetic only")
    lib = L.lib()
    rng = np.random.default_rng(300 + variant * 7 + C + hw[1])
    N, H, W, scale = 3, 40, 56, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    R = 203
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, R, extent=W / scale, wh=(4.0, 200.0)),
                                            rng.integers(0, N, R)), I.edge_rois(H, W, scale)], 0)
    rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0        # masked
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last)
    r = torch.from_numpy(rois).to(dev)
    outs = []
    from jdet_amd import _experimental as X
    for mode in (3, 0):
        out = torch.full((rois.shape[0], C) + hw, float("nan"), device=x.device).contiguous(memory_format=torch.channels_last)
        if mode == 3:       # the line kernel: libjdet_experimental.so
            L.check(X.lib().jdet_roi_align_forward_cl_mode(3, variant, x.data_ptr(), N, C, H, W, r.data_ptr(),
                                                           rois.shape[0], hw[0], hw[1], scale, 2, 1, None, out.data_ptr(),
                                                           None, 0, L.stream_ptr(x)), "fwd_cl_mode 3")
        else:
            L.check(lib.jdet_roi_align_forward_cl_roi(variant, x.data_ptr(), N, C, H, W, r.data_ptr(), rois.shape[0],
                                                      hw[0], hw[1], scale, 2, 1, None, out.data_ptr(),
                                                      L.stream_ptr(x)), "fwd_cl_roi")
        outs.append(out.cpu().numpy())
    a, b = outs
    masked = rois[:, 0] < 0
    assert np.isnan(a[masked]).all()
    assert not np.isnan(a[~masked]).any()
    ref = O.roi_align_forward(variant, feat, rois[~masked], hw, scale, 2)
    np.testing.assert_allclose(a[~masked], ref, rtol=0, atol=FWD_MERGED_ATOL)
    np.testing.assert_allclose(a[~masked], b[~masked], rtol=0, atol=FWD_MERGED_ATOL)


@pytest.mark.parametrize("variant", [O.V_ROT, O.V_ROT_V1, O.V_HBB0, O.V_HBB1])
def test_planned_backward_equals_the_self_contained_one(dev, variant, fwd_mode):
    """round 6: jdet_roi_align_backward_plan + jdet_roi_align_backward_cl_planned (the plan built once, the gather alone per
    gradient) against the oracle's scatter and the self-contained jdet_roi_align_backward_cl; the plan survives a gather
    (second gradient, other channel count, same plan); masked RoIs and RoIs over the border included."""
    from jdet_amd import _lib as L
    if fwd_mode != FWD_MERGED_ATOL:
        pytest.skip("one arithmetic is enough: the backward has a single one")
    lib = L.lib()
    rng = np.random.default_rng(77 + variant)
    N, H, W, scale = 2, 40, 56, 0.25
    rois = np.concatenate([I.rois_from_obbs(I.random_obbs(rng, 150, extent=W / scale, wh=(4.0, 200.0)),
                                            rng.integers(0, N, 150)), I.edge_rois(H, W, scale)], 0)
    rois[rng.random(rois.shape[0]) < 0.2, 0] = -1.0
    if variant in (O.V_HBB0, O.V_HBB1):
        rois = I.obb_to_hbb_rois(rois)
    R = rois.shape[0]
    r = torch.from_numpy(rois).to(dev)
    pb = lib.jdet_roi_align_backward_plan_bytes(variant, R, N, H, W, 7, 7, 2)
    assert pb > 0
    plan = torch.empty((pb,), dtype=torch.uint8, device=dev).fill_(0xA5)        # any content on entry
    L.check(lib.jdet_roi_align_backward_plan(variant, r.data_ptr(), R, N, H, W, 7, 7, scale, 2, plan.data_ptr(), pb,
                                             L.stream_ptr(r)), "plan")
    for C in (64, 8, 260):
        grad = rng.standard_normal((R, C, 7, 7)).astype(np.float32)
        g = torch.from_numpy(grad).to(dev).contiguous(memory_format=torch.channels_last)
        gin = torch.full((N, C, H, W), float("nan"), device=dev).contiguous(memory_format=torch.channels_last)
        L.check(lib.jdet_roi_align_backward_cl_planned(variant, g.data_ptr(), R, N, C, H, W, 7, 7, 2, gin.data_ptr(),
                                                       plan.data_ptr(), pb, L.stream_ptr(g)), "planned")
        live = rois[:, 0] >= 0
        ref = O.roi_align_backward(variant, grad[live], rois[live], (N, C, H, W), scale, 2)
        np.testing.assert_allclose(gin.cpu().numpy(), ref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(ref).max()))
    # too small a plan buffer / RiRoIAlign are refused
    assert lib.jdet_roi_align_backward_cl_planned(variant, g.data_ptr(), R, N, C, H, W, 7, 7, 2, gin.data_ptr(),
                                                  plan.data_ptr(), pb - 1, L.stream_ptr(g)) == -3
    assert lib.jdet_roi_align_backward_cl_planned(O.V_RI, g.data_ptr(), R, N, 256, H, W, 7, 7, 2, gin.data_ptr(),
                                                  plan.data_ptr(), pb, L.stream_ptr(g)) == -2


def test_autograd_uses_the_plan_and_matches_without_it(dev, fwd_mode):
    """the layer builds the plan at the forward (side stream) and gathers from it in backward; JDET_ROI_BWD_PLAN off gives
    the same gradient (summation order aside); two backward passes through one graph are refused by autograd as usual"""
    from jdet_amd.ops import _roi_common as RC
    if fwd_mode != FWD_MERGED_ATOL or RC._FORWARD_PATH[0] != "roi_cl":
        pytest.skip("channels-last result only")
    rng = np.random.default_rng(5)
    N, C, H, W, scale = 2, 64, 48, 48, 0.25
    feat = rng.standard_normal((N, C, H, W)).astype(np.float32)
    rois = I.rois_from_obbs(I.random_obbs(rng, 300, extent=W / scale, wh=(4.0, 160.0)), rng.integers(0, N, 300))
    grad = torch.from_numpy(rng.standard_normal((300, C, 7, 7)).astype(np.float32)).to(dev).contiguous(
        memory_format=torch.channels_last)
    outs = []
    for on in (True, False):
        prev = RC.set_backward_plan(on)
        try:
            x = torch.from_numpy(feat).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = _layer(O.V_ROT, (7, 7), scale, 2)(x, torch.from_numpy(rois).to(dev))
            y.backward(grad)
            outs.append(x.grad.cpu().numpy())
        finally:
            RC.set_backward_plan(prev)
    ref = O.roi_align_backward(O.V_ROT, grad.cpu().numpy(), rois, feat.shape, scale, 2)
    for g in outs:
        np.testing.assert_allclose(g, ref, rtol=0, atol=BWD_ATOL * max(1.0, np.abs(ref).max()))

