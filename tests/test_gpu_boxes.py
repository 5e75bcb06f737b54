"""GPU parity: fused delta codec, fused MaxIoUAssigner, anchor generator, AlignConv offsets and
anchor_target against the numpy / C++ oracles.  Integer outputs (gt_inds, labels, pos/neg sets)
bit-exact; codec outputs within 2e-5 relative (device libm vs numpy for cos/sin/exp/log)."""
import math

import numpy as np
import pytest
import torch

from oracle import box_oracle as B
from oracle import oracle as O
from tests import inputs as I

pytestmark = pytest.mark.gpu


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_codec_vs_oracle(dev):
    from jdet_amd.models.boxes.box_ops import bbox2delta_rotated, delta2bbox_rotated, norm_angle
    rng = np.random.default_rng(0)
    p, g = I.random_obbs(rng, 21824), I.random_obbs(rng, 21824)
    m, s = (0.1, -0.2, 0.3, 0.0, 0.05), (0.1, 0.2, 0.2, 0.1, 0.5)
    for means, stds in (((0,) * 5, (1,) * 5), (m, s)):
        d = bbox2delta_rotated(_t(p, dev), _t(g, dev), means, stds).cpu().numpy()
        np.testing.assert_allclose(d, B.bbox2delta_rotated(p, g, means, stds), rtol=2e-5, atol=2e-5)
        for clip in (16 / 1000, 1e-6):
            dd = (rng.standard_normal((21824, 15)) * 0.5).astype(np.float32)
            out = delta2bbox_rotated(_t(p, dev), _t(dd, dev), means, stds, None, clip).cpu().numpy()
            ref = B.delta2bbox_rotated(p, dd, means, stds, clip)
            np.testing.assert_allclose(out[:, [0, 1, 2, 3, 5, 6, 7, 8]], ref[:, [0, 1, 2, 3, 5, 6, 7, 8]], rtol=3e-5, atol=2e-3)
            da = np.abs(out[:, 4::5] - ref[:, 4::5])   # angles: equal modulo the wrap at the range edge
            assert np.all((da < 1e-4) | (np.abs(da - math.pi) < 1e-4))
    a = torch.tensor([-10., -3.2, -math.pi / 4, 0., 2.3, 9.7], device=dev)
    np.testing.assert_allclose(norm_angle(a).cpu().numpy(), B.norm_angle(a.cpu().numpy()), atol=1e-6)
    # differentiable torch form == fused form
    pt, gt_ = _t(p[:100], dev), _t(g[:100], dev)
    dt = _t((rng.standard_normal((100, 5)) * 0.3).astype(np.float32), dev).requires_grad_(True)
    o1 = delta2bbox_rotated(pt, dt)
    o2 = delta2bbox_rotated(pt, dt.detach())
    assert o1.requires_grad and torch.allclose(o1, o2, rtol=1e-5, atol=1e-3)
    o1.sum().backward()
    assert torch.isfinite(dt.grad).all()


@pytest.mark.parametrize("K,A", [(64, 21824), (1, 7), (3, 300), (200, 1000)])
def test_assigner_vs_oracle(dev, K, A):
    from jdet_amd.models.boxes.assigner import MaxIoUAssigner
    rng = np.random.default_rng(K * 1000 + A)
    gts = I.random_obbs(rng, K, wh=(16.0, 256.0))
    anchors = B.grid_anchors_s2anet(8, [4], [1.0], (128, 128), 8)[rng.permutation(16384)[:A]] if A <= 16384 else \
        np.concatenate([B.grid_anchors_s2anet(s, [4], [1.0], (1024 // s, 1024 // s), s) for s in (8, 16, 32, 64, 128)])
    assert anchors.shape[0] == A
    gl = rng.integers(1, 16, K).astype(np.int32)
    ov = O.box_iou_rotated(gts, anchors)
    for kw in (dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0),
               dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.3, match_low_quality=True, assigned_labels_filled=-1),
               dict(pos_iou_thr=0.7, neg_iou_thr=(0.1, 0.3), min_pos_iou=0.3, match_low_quality=False),
               dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0, gt_max_assign_all=False)):
        asg = MaxIoUAssigner(iou_calculator=dict(type="BboxOverlaps2D_rotated"), **kw)
        res = asg.assign(_t(anchors, dev), _t(gts, dev), None, _t(gl, dev))
        gi, mo, lab = B.assign_wrt_overlaps(ov, kw["pos_iou_thr"], kw["neg_iou_thr"], kw.get("min_pos_iou", 0.0),
                                            kw.get("match_low_quality", True), kw.get("gt_max_assign_all", True), gl,
                                            kw.get("assigned_labels_filled", 0))
        np.testing.assert_array_equal(res.gt_inds.cpu().numpy(), gi)
        np.testing.assert_array_equal(res.max_overlaps.cpu().numpy(), mo)
        np.testing.assert_array_equal(res.labels.cpu().numpy(), lab)
        assert res.num_gts == K
    asg = MaxIoUAssigner(0.5, 0.4, iou_calculator=dict(type="BboxOverlaps2D_rotated"))
    r2 = asg.assign(_t(anchors, dev), _t(gts, dev))
    assert r2.labels is None
    with pytest.raises(ValueError):
        asg.assign(_t(anchors, dev), torch.zeros((0, 5), device=dev))


def test_hand_built_overlaps_on_device(dev):
    from jdet_amd.models.boxes.assigner import assign_wrt_overlaps_device
    ov = np.asarray([[0.9, 0.45, 0.3, 0.0, 0.2, 0.6], [0.1, 0.45, 0.6, 0.0, 0.2, 0.6], [0.0, 0.10, 0.1, 0.0, 0.2, 0.1]],
                    np.float32)
    gl = torch.tensor([7, 8, 9], dtype=torch.int32, device=dev)
    gi, mo, lab = assign_wrt_overlaps_device(_t(ov, dev), 0.5, 0.4, 0.0, True, True, gl, 0)
    assert gi.tolist() == [1, -1, 2, 0, 3, 2] and lab.tolist() == [7, 0, 8, 0, 9, 8]
    gi, _, _ = assign_wrt_overlaps_device(_t(ov, dev), 0.5, 0.4, 0.0, True, False, None, 0)
    assert gi.tolist() == [1, -1, 2, 0, 3, 1]
    gi, _, _ = assign_wrt_overlaps_device(_t(ov, dev), 0.5, 0, 0.0, False, True, None, 0)   # int thr: no negatives
    assert gi.tolist() == [1, -1, 2, -1, -1, 1]


def test_anchor_generator_and_alignconv_offsets(dev):
    from jdet_amd.models.boxes.anchor_generator import AnchorGeneratorRotatedRetinaNet, AnchorGeneratorRotatedS2ANet
    from jdet_amd.models.roi_heads.s2anet_head import AlignConv
    g = AnchorGeneratorRotatedS2ANet(16, [4], [1.0])
    a = g.grid_anchors((5, 7), 16, device=dev).cpu().numpy()
    np.testing.assert_array_equal(a, B.grid_anchors_s2anet(16, [4], [1.0], (5, 7), 16))
    f = g.valid_flags((5, 7), (4, 6), device=dev).cpu().numpy().reshape(5, 7)
    assert f[:4, :6].all() and not f[4].any() and not f[:, 6].any()
    r = AnchorGeneratorRotatedRetinaNet(8, None, [0.5, 1.0, 2.0], octave_base_scale=4, scales_per_octave=3)
    assert r.num_base_anchors == 9
    b = r.base_anchors.numpy()
    np.testing.assert_allclose(b[:, 2] * b[:, 3], np.tile((8 * 4 * 2 ** (np.arange(3) / 3)) ** 2, 3), rtol=1e-5)
    np.testing.assert_allclose(b[:3, 3] / b[:3, 2], 0.5, rtol=1e-5)     # ratio-major, then scale
    rng = np.random.default_rng(1)
    H, W, s = 9, 11, 16
    anc = I.random_obbs(rng, 2 * H * W, extent=W * s, wh=(16.0, 128.0)).reshape(2, H * W, 5)
    off = AlignConv(4, 4).to(dev).get_offset(_t(anc, dev), (H, W), s).cpu().numpy()
    for i in range(2):
        np.testing.assert_allclose(off[i], B.align_conv_offsets(anc[i], (H, W), s), rtol=1e-5, atol=2e-5)


def test_anchor_target_vs_oracle(dev):
    """S2ANet FAM targets for one 1024x1024 image (21,824 anchors, 64 gts): labels / weights / pos / neg sets
    bit-exact, regression targets 2e-5."""
    from jdet_amd.models.boxes.anchor_target import anchor_target
    from jdet_amd.models.roi_heads.s2anet_head import _DEFAULT_ASSIGN, _cfg
    rng = np.random.default_rng(2)
    strides = (8, 16, 32, 64, 128)
    lv = [B.grid_anchors_s2anet(s, [4], [1.0], (1024 // s, 1024 // s), s) for s in strides]
    gts = [I.random_obbs(rng, 64, wh=(16.0, 256.0)), I.random_obbs(rng, 30, wh=(16.0, 256.0))]
    gls = [rng.integers(1, 16, 64).astype(np.int32), rng.integers(1, 16, 30).astype(np.int32)]
    anchor_list = [[_t(a, dev) for a in lv] for _ in range(2)]
    valid = [[torch.ones(a.shape[0], dtype=torch.bool, device=dev) for a in lv] for _ in range(2)]
    metas = [dict(img_shape=(1024, 1024), pad_shape=(1024, 1024)) for _ in range(2)]
    out = anchor_target(anchor_list, valid, [_t(g, dev) for g in gts], metas, (0,) * 5, (1,) * 5, _cfg(_DEFAULT_ASSIGN),
                        gt_labels_list=[_t(g, dev) for g in gls], label_channels=15, sampling=False)
    labels_list, lw_list, bt_list, bw_list, npos, nneg = out
    flat = np.concatenate(lv)
    tot_pos = tot_neg = 0
    for i in range(2):
        lab, lw, bt, bw, pos, neg = B.anchor_target_single(flat, gts[i], gls[i])
        tot_pos += max(len(pos), 1)
        tot_neg += max(len(neg), 1)
        np.testing.assert_array_equal(torch.cat([l[i] for l in labels_list]).cpu().numpy(), lab)
        np.testing.assert_array_equal(torch.cat([l[i] for l in lw_list]).cpu().numpy(), lw)
        np.testing.assert_array_equal(torch.cat([l[i] for l in bw_list]).cpu().numpy(), bw)
        np.testing.assert_allclose(torch.cat([l[i] for l in bt_list]).cpu().numpy(), bt, rtol=2e-5, atol=2e-5)
    assert (npos, nneg) == (tot_pos, tot_neg)
    assert [tuple(l.shape) for l in labels_list] == [(2, (1024 // s) ** 2) for s in strides]


def test_dense_anchor_targets_equal_index_path(dev):
    """fixed-shape target path (fused kernel, no nonzero / host sync) == the reference-shaped index path"""
    from jdet_amd.models.boxes.anchor_target import anchor_target
    rng = np.random.default_rng(11)
    sizes = [(16, 16), (8, 8), (4, 4)]
    strides = [8, 16, 32]
    cfg = dict(assigner=dict(type="MaxIoUAssigner", pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0,
                             ignore_iof_thr=-1, iou_calculator=dict(type="BboxOverlaps2D_rotated")),
               bbox_coder=dict(type="DeltaXYWHABBoxCoder", target_means=(0., 0., 0., 0., 0.),
                               target_stds=(0.5, 0.5, 1., 1., 2.), clip_border=True),
               allowed_border=-1, pos_weight=-1, debug=False)

    def inputs():
        anchors, flags = [], []
        for (h, w), s in zip(sizes, strides):
            yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
            a = np.stack([xx.ravel() * s + s / 2, yy.ravel() * s + s / 2, np.full(h * w, 4.0 * s), np.full(h * w, 4.0 * s),
                          rng.uniform(-0.6, 0.6, h * w)], 1).astype(np.float32)
            anchors.append(torch.from_numpy(a).to(dev))
            flags.append(torch.ones(h * w, dtype=torch.bool, device=dev))
        return anchors, flags
    a0, f0 = inputs()
    gts, labels = [], []
    for k in (7, 1):
        g = np.concatenate([rng.uniform(10, 118, (k, 2)), rng.uniform(20, 70, (k, 2)), rng.uniform(-1.5, 1.5, (k, 1))], 1)
        gts.append(torch.from_numpy(g.astype(np.float32)).to(dev))
        labels.append(torch.from_numpy(rng.integers(1, 16, k).astype(np.int32)).to(dev))
    metas = [dict(img_shape=(128, 128), pad_shape=(128, 128), _all_valid=True) for _ in range(2)]
    outs = []
    for dense in (True, False):
        al = [list(a0), list(a0)]
        fl = [list(f0), list(f0)]
        outs.append(anchor_target(al, fl, gts, [dict(m) for m in metas], None, None, cfg, gt_labels_list=labels,
                                  label_channels=15, sampling=False, dense=dense))
    d, s = outs
    assert torch.is_tensor(d[4]) and d[4].dim() == 0 and isinstance(s[4], int)
    assert float(d[4]) == float(s[4]) and s[4] >= 2
    for k in range(4):
        for lv in range(3):
            assert d[k][lv].shape == s[k][lv].shape
            assert torch.equal(d[k][lv].float(), s[k][lv].float()), (k, lv)


def test_fused_focal_loss_matches_tensor_program(dev):
    """fused sigmoid focal loss (value and gradient) == the reference-shaped tensor-op chain"""
    from jdet_amd.models.losses.focal_loss import FocalLoss, sigmoid_focal_loss
    rng = np.random.default_rng(2)
    for M, C in ((1000, 15), (37, 1), (4096, 15)):
        x = torch.from_numpy((rng.standard_normal((M, C)) * 3).astype(np.float32)).to(dev)
        x[0, 0], x[1, 0] = 40.0, -40.0     # saturated logits
        lab = torch.from_numpy(rng.integers(0, C + 1, M).astype(np.int32)).to(dev)
        w = torch.from_numpy((rng.uniform(0, 1, M) > 0.2).astype(np.float32)).to(dev)
        x1 = x.clone().requires_grad_(True)
        x2 = x.clone().requires_grad_(True)
        avg = torch.tensor(17.0, device=dev)
        l1 = FocalLoss(loss_weight=0.7)(x1, lab, w, avg_factor=avg)
        l2 = 0.7 * sigmoid_focal_loss(x2, lab, w, gamma=2.0, alpha=0.25, reduction="mean", avg_factor=avg)
        assert l1.dim() == 0
        torch.testing.assert_close(l1, l2, rtol=2e-5, atol=1e-6)
        l1.backward()
        l2.backward()
        torch.testing.assert_close(x1.grad, x2.grad, rtol=2e-4, atol=1e-7)
    assert torch.isfinite(x1.grad).all()


def test_fused_align_conv_offset_matches_restatement(dev):
    from jdet_amd.models.roi_heads.s2anet_head import AlignConv
    from oracle import box_oracle as B
    rng = np.random.default_rng(4)
    H, W, stride = 9, 13, 16
    anchors = np.concatenate([rng.uniform(0, 200, (2, H * W, 2)), np.exp(rng.uniform(np.log(8), np.log(200), (2, H * W, 2))),
                              rng.uniform(-1.6, 1.6, (2, H * W, 1))], -1).astype(np.float32)
    ac = AlignConv(8, 8, 3).to(dev)
    got = ac.get_offset(torch.from_numpy(anchors).to(dev), (H, W), stride).cpu().numpy()
    assert got.shape == (2, 18, H, W)
    for n in range(2):
        np.testing.assert_allclose(got[n], B.align_conv_offsets(anchors[n], (H, W), stride), rtol=1e-5, atol=2e-5)
    # an axis-aligned anchor of size 3*stride centred on the pixel samples the regular grid: zero offsets
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    reg = np.stack([xx.ravel() * stride, yy.ravel() * stride, np.full(H * W, 3.0 * stride), np.full(H * W, 3.0 * stride),
                    np.zeros(H * W)], 1).astype(np.float32)[None]
    z = ac.get_offset(torch.from_numpy(reg).to(dev), (H, W), stride)
    assert float(z.abs().max()) < 1e-5


def test_fused_smooth_l1_matches_tensor_program(dev):
    from jdet_amd.models.losses.smooth_l1_loss import L1Loss, SmoothL1Loss, l1_loss, smooth_l1_loss
    rng = np.random.default_rng(6)
    for M, D, beta in ((5000, 5, 1.0 / 9.0), (300, 4, 1.0), (64, 5, 0.0)):
        p = torch.from_numpy(rng.standard_normal((M, D)).astype(np.float32)).to(dev)
        t = torch.from_numpy((rng.standard_normal((M, D)) * 0.5).astype(np.float32)).to(dev)
        p[0, 0] = t[0, 0]                                   # zero difference: sign(0) = 0
        w = torch.from_numpy((rng.uniform(0, 1, (M, D)) > 0.7).astype(np.float32)).to(dev)
        for weight in (w, w[:, 0].contiguous(), None):
            p1, p2 = p.clone().requires_grad_(True), p.clone().requires_grad_(True)
            avg = torch.tensor(23.0, device=dev)
            if beta == 0.0:
                l1 = L1Loss(loss_weight=1.3)(p1, t, weight, avg_factor=avg)
                l2 = 1.3 * l1_loss(p2, t, weight, avg_factor=avg)
            else:
                l1 = SmoothL1Loss(beta=beta, loss_weight=1.3)(p1, t, weight, avg_factor=avg)
                l2 = 1.3 * smooth_l1_loss(p2, t, weight, beta=beta, avg_factor=avg)
            torch.testing.assert_close(l1, l2, rtol=2e-5, atol=1e-6)
            l1.backward()
            l2.backward()
            torch.testing.assert_close(p1.grad, p2.grad, rtol=1e-5, atol=1e-7)
    # default normaliser (no avg_factor) = number of rows
    q = torch.randn(10, 5, device=dev, requires_grad=True)
    a = SmoothL1Loss(beta=1.0)(q, torch.zeros(10, 5, device=dev))
    torch.testing.assert_close(a, smooth_l1_loss(q.detach(), torch.zeros(10, 5, device=dev), beta=1.0), rtol=1e-5, atol=1e-6)


def test_level_loss_nodes_equal_the_composition_bit_for_bit(dev):
    """Round 6: one autograd node per (pyramid level, loss) -- targets read in place through their column windows of the
    per-image arrays, `/ avg_factor` and `* loss_weight` inside the finishing launch, one scaling launch in backward
    (jdet_*_loss_level, jdet_loss_grad_scale) -- against what it replaces: window.reshape(-1) copies + the fused sum +
    two scalar ops (+ three in backward).  Same operations in the same order: loss and gradient EQUAL."""
    from jdet_amd.models.losses import focal_loss as FL
    from jdet_amd.models.losses.smooth_l1_loss import L1Loss, SmoothL1Loss
    rng = np.random.default_rng(11)
    N, A, C = 3, 700, 15
    labels = torch.from_numpy(rng.integers(0, C + 1, (N, A)).astype(np.int32)).to(dev)
    lw = torch.from_numpy((rng.uniform(0, 1, (N, A)) > 0.2).astype(np.float32)).to(dev)
    bt = torch.from_numpy((rng.standard_normal((N, A, 5)) * 0.5).astype(np.float32)).to(dev)
    bw = torch.from_numpy((rng.uniform(0, 1, (N, A, 5)) > 0.7).astype(np.float32)).to(dev)
    avg = torch.tensor(37.0, device=dev)
    up = torch.tensor(0.625, device=dev)                  # the gradient arriving from parse_losses' sum
    for s, e in ((0, 300), (300, 650), (650, 700), (0, 700)):
        M = N * (e - s)
        x = torch.from_numpy((rng.standard_normal((M, C)) * 3).astype(np.float32)).to(dev)
        p = torch.from_numpy(rng.standard_normal((M, 5)).astype(np.float32)).to(dev)
        got = {}
        for nodes in (True, False):
            FL.LEVEL_NODES = nodes
            try:
                x1, p1 = x.clone().requires_grad_(True), p.clone().requires_grad_(True)
                lc = FL.FocalLoss(loss_weight=0.7)(x1, labels[:, s:e], lw[:, s:e], avg_factor=avg)
                lb = SmoothL1Loss(beta=1.0 / 9.0, loss_weight=1.3)(p1, bt[:, s:e], bw[:, s:e], avg_factor=avg)
                l1 = L1Loss(loss_weight=0.9)(p1, bt[:, s:e], bw[:, s:e], avg_factor=avg)
                ((lc + lb + l1) * up).backward()
                got[nodes] = (lc.detach(), lb.detach(), l1.detach(), x1.grad, p1.grad)
            finally:
                FL.LEVEL_NODES = True
        for a, b in zip(got[True], got[False]):
            assert torch.equal(a, b)
    # the node is what ran: no clone of a window, no scalar op (one node per loss)
    x1 = x.clone().requires_grad_(True)
    assert type(FL.FocalLoss()(x1, labels, lw, avg_factor=avg).grad_fn).__name__ == "_FocalLevelBackward"
    # a host avg_factor keeps the composition (the framework divides by a host scalar through its reciprocal)
    assert type(FL.FocalLoss()(x1, labels, lw, avg_factor=37.0).grad_fn).__name__ != "_FocalLevelBackward"


# ---- fused Oriented R-CNN codecs (csrc/box_codec_oriented.hip) vs the numpy restatement -----------------------
def _obbs(n, seed, regular=True):
    rng = np.random.default_rng(seed)
    c = rng.uniform(50, 950, (n, 2))
    wh = np.exp(rng.uniform(np.log(8), np.log(300), (n, 2)))
    if regular:
        wh = np.stack([wh.max(1), wh.min(1)], 1)
    th = rng.uniform(-math.pi / 2, math.pi / 2, (n, 1)) if regular else rng.uniform(-4, 4, (n, 1))
    return np.concatenate([c, wh, th], 1).astype(np.float32)


def test_fused_oriented_codecs_vs_oracle(dev):
    """MidpointOffsetCoder / OrientedDeltaXYWHTCoder on the device = one launch each; values = the numpy restatement
    of coder.py:L332-518 (tolerances: libm differences of cos / sin / exp / log / atan2 between numpy and HIP)"""
    from jdet_amd.models.boxes.coder import MidpointOffsetCoder, OrientedDeltaXYWHTCoder
    from oracle import box_oracle as BO
    n = 4000
    g = _obbs(n, 1)
    anchors = BO.obb2hbb(g) + np.random.default_rng(2).uniform(-12, 12, (n, 4)).astype(np.float32)
    m6, s6 = [0.] * 6, [1., 1., 1., 1., .5, .5]
    mc = MidpointOffsetCoder(target_means=m6, target_stds=s6)
    t = lambda a: torch.from_numpy(a).to(dev)
    e = mc.encode(t(anchors), t(g)).cpu().numpy()
    np.testing.assert_allclose(e, BO.midpoint_offset_encode(anchors, g, m6, s6), rtol=2e-5, atol=2e-5)
    d6 = np.random.default_rng(3).normal(0, 0.5, (n, 6)).astype(np.float32)
    dec = mc.decode(t(anchors), t(d6)).cpu().numpy()
    ref = BO.midpoint_offset_decode(anchors, d6, m6, s6)
    assert dec.shape == (n, 5)
    np.testing.assert_allclose(dec[:, :4], ref[:, :4], rtol=1e-4, atol=5e-3)
    assert np.abs(np.sin(dec[:, 4] - ref[:, 4])).max() < 2e-4          # angles equal modulo pi (wrap at the interval end)
    assert (dec[:, 2] >= dec[:, 3]).all() and (dec[:, 4] >= -math.pi / 2 - 1e-6).all() and (dec[:, 4] < math.pi / 2 + 1e-6).all()
    m5, s5 = [0.] * 5, [0.1, 0.1, 0.2, 0.2, 0.1]
    oc = OrientedDeltaXYWHTCoder(target_means=m5, target_stds=s5)
    p = _obbs(n, 4, regular=False)
    e5 = oc.encode(t(p), t(g)).cpu().numpy()
    r5 = BO.oriented_delta_encode(p, g, m5, s5)
    # rows where |dtheta1| and |dtheta2| tie within rounding may pick the other branch: exclude exact ties only
    d1 = np.abs(BO.regular_theta(g[:, 4] - p[:, 4]))
    d2 = np.abs(BO.regular_theta(g[:, 4] - p[:, 4] + np.float32(math.pi / 2)))
    ok = np.abs(d1 - d2) > 1e-4
    assert ok.mean() > 0.999
    np.testing.assert_allclose(e5[ok], r5[ok], rtol=2e-4, atol=2e-4)
    d15 = np.random.default_rng(5).normal(0, 1, (n, 15)).astype(np.float32)
    dec15 = oc.decode(t(p), t(d15)).cpu().numpy().reshape(n, 3, 5)
    ref15 = BO.oriented_delta_decode(p, d15, m5, s5).reshape(n, 3, 5)
    np.testing.assert_allclose(dec15[..., :4], ref15[..., :4], rtol=1e-4, atol=5e-3)
    assert np.abs(np.sin(dec15[..., 4] - ref15[..., 4])).max() < 2e-4
    # round trip on the device: decode(encode(gt)) is the gt again (both regular)
    back = oc.decode(t(p), oc.encode(t(p), t(g)), wh_ratio_clip=1e-6).cpu().numpy()
    np.testing.assert_allclose(back[:, :4], g[:, :4], rtol=2e-4, atol=2e-2)
    assert np.abs(np.sin(back[:, 4] - g[:, 4])).max() < 2e-4
    assert mc.decode(t(anchors[:0]), t(d6[:0])).shape == (0, 5)


@pytest.mark.parametrize("version", [0, 1])
@pytest.mark.parametrize("mode", ["iou", "iof"])
def test_fused_hbb_overlaps_equal_the_tensor_program(dev, version, mode):
    """jdet_bbox_overlaps_hbb = `bbox_overlaps` (iou_calculator.py:L235-350) bit for bit: same operation order, incl.
    the +1 convention, the eps clamp (degenerate / identical boxes), 5-column candidates (score column ignored) and
    the dead-column mask of the fixed-shape heads; the assigner on top (row maxima in chunks) is unchanged."""
    from jdet_amd.models.boxes.iou_calculator import bbox_overlaps, bbox_overlaps_fused
    g = torch.Generator().manual_seed(3 + version)
    K, A = 37, 5003
    c = torch.rand((K, 2), generator=g) * 500
    wh = torch.rand((K, 2), generator=g) * 120
    gts = torch.cat([c - wh / 2, c + wh / 2], 1)
    gts[3, 2:] = gts[3, :2]                       # zero-area gt
    c2 = torch.rand((A, 2), generator=g) * 500
    wh2 = torch.rand((A, 2), generator=g) * 150
    boxes = torch.cat([c2 - wh2 / 2, c2 + wh2 / 2, torch.rand((A, 1), generator=g)], 1)
    boxes[:K, :4] = gts                           # exact duplicates
    boxes[100, 2:4] = boxes[100, :2]              # zero-area box
    gts, boxes = gts.to(dev), boxes.to(dev)
    ref = bbox_overlaps(gts, boxes[:, :4], mode, version=version)
    got = bbox_overlaps_fused(gts, boxes, mode, version)
    assert torch.equal(got, ref)
    alive = (torch.rand((A,), generator=g) > 0.3).to(dev)
    got = bbox_overlaps_fused(gts, boxes, mode, version, alive=alive)
    assert torch.equal(got, torch.where(alive[None, :], ref, torch.full_like(ref, -1.0)))
