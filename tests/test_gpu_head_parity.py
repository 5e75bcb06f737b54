"""GPU: head-level parity.  S2ANetHead.loss and get_bboxes on FIXED network outputs against the
numpy composition in oracle/head_oracle.py (s2anet_head.py:L322-428, L510-601; anchor_target.py:L60-102; focal /
smooth-L1 losses; multiclass_nms_rotated).  What this pins beyond the per-kernel tests: level order, the
`images_to_levels` regrouping, averaging factors (sum over images of max(#pos, 1)), the 1-based label convention of
the one-stage heads, score threshold / nms_pre / max_per_img and the order of the returned detections."""
import numpy as np
import pytest
import torch

from oracle import head_oracle as HO
from tests import inputs as I

pytestmark = pytest.mark.gpu

STRIDES = [8, 16, 32, 64, 128]


def _fixed_outputs(rng, N, size, num_cls):
    sizes = [(max(size // s, 1), max(size // s, 1)) for s in STRIDES]
    from oracle import box_oracle as B
    fam_cls = [rng.normal(-2.0, 1.5, size=(N, num_cls) + sz).astype(np.float32) for sz in sizes]
    odm_cls = [rng.normal(-2.0, 1.5, size=(N, num_cls) + sz).astype(np.float32) for sz in sizes]
    fam_box = [rng.normal(0, 0.15, size=(N, 5) + sz).astype(np.float32) for sz in sizes]
    odm_box = [rng.normal(0, 0.15, size=(N, 5) + sz).astype(np.float32) for sz in sizes]
    refine = []
    for s, sz, fb in zip(STRIDES, sizes, fam_box):
        init = B.grid_anchors_s2anet(s, [4], [1.0], sz, s)                                     # (H*W, 5)
        dec = np.stack([B.delta2bbox_rotated(init, np.transpose(fb[i], (1, 2, 0)).reshape(-1, 5), wh_ratio_clip=1e-6)
                        for i in range(N)], 0)
        refine.append(dec.reshape((N,) + sz + (5,)).astype(np.float32))
    return sizes, fam_cls, fam_box, refine, odm_cls, odm_box


def _head(dev):
    import jdet_amd.models  # noqa: F401  (registries)
    from jdet_amd.models.roi_heads.s2anet_head import S2ANetHead
    return S2ANetHead(num_classes=16, in_channels=256).to(dev)


def test_s2anet_head_loss_vs_restatement(dev):
    rng = np.random.default_rng(31)
    N, size = 2, 256
    sizes, fam_cls, fam_box, refine, odm_cls, odm_box = _fixed_outputs(rng, N, size, 15)
    gts = [I.random_obbs(rng, k, extent=float(size), wh=(12.0, 120.0)) for k in (9, 5)]
    labels = [rng.integers(1, 16, size=g.shape[0]).astype(np.int32) for g in gts]
    ref = HO.s2anet_loss(fam_cls, fam_box, refine, odm_cls, odm_box, gts, labels, STRIDES)
    head = _head(dev).train()
    t = lambda a: torch.from_numpy(a).to(dev)
    metas = [dict(img_shape=(size, size), scale_factor=1.0, pad_shape=(size, size)) for _ in range(N)]
    out = head.loss([t(a) for a in fam_cls], [t(a) for a in fam_box], [t(a) for a in refine], [t(a) for a in odm_cls],
                    [t(a) for a in odm_box], [t(g) for g in gts], [t(l) for l in labels], metas)
    assert set(out) == set(ref)
    for k in ref:
        got = np.asarray([float(v) for v in out[k]])
        assert got.shape == (len(STRIDES),)
        np.testing.assert_allclose(got, np.asarray(ref[k]), rtol=2e-5, atol=1e-7, err_msg=k)
    # the assignment is not degenerate: both modules see positives on several levels
    assert sum(v > 0 for v in ref["loss_fam_bbox"]) >= 2 and sum(v > 0 for v in ref["loss_odm_bbox"]) >= 2


def _corner_sets(polys):
    p = np.asarray(polys, np.float64).reshape(-1, 4, 2)
    order = np.lexsort((p[:, :, 1], p[:, :, 0]), axis=1)
    return np.take_along_axis(p, order[:, :, None], 1)


def _rect_corners(b):
    """the four corners c + R(theta) (+-w/2, +-h/2) (box_ops.py:L556-563), as (k, 8)"""
    b = np.asarray(b, np.float64)
    c, s = np.cos(b[:, 4]), np.sin(b[:, 4])
    out = []
    for px, py in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
        x, y = px * b[:, 2] / 2, py * b[:, 3] / 2
        out += [b[:, 0] + c * x - s * y, b[:, 1] + s * x + c * y]
    return np.stack(out, 1)


def test_s2anet_head_get_bboxes_vs_restatement(dev):
    rng = np.random.default_rng(32)
    N, size = 2, 256
    sizes, fam_cls, fam_box, refine, odm_cls, odm_box = _fixed_outputs(rng, N, size, 15)
    for a in odm_cls:       # a few confident detections per level, clustered so that NMS has work to do
        a += (rng.uniform(size=a.shape) < 0.03) * rng.uniform(2.0, 6.0, size=a.shape).astype(np.float32)
    head = _head(dev).eval()
    t = lambda a: torch.from_numpy(a).to(dev)
    metas = [dict(img_shape=(size, size), scale_factor=1.0, pad_shape=(size, size)) for _ in range(N)]
    with torch.no_grad():
        res = head.get_bboxes([t(a) for a in fam_cls], [t(a) for a in fam_box], [t(a) for a in refine],
                              [t(a) for a in odm_cls], [t(a) for a in odm_box], metas)
    from jdet_amd.ops import nms_rotated as NR
    cmp_ge = 1 if NR.REFERENCE_RULE == "cpu" else 0
    for i in range(N):
        polys, scores, labels = (v.cpu().numpy() for v in res[i])
        eb, es, el = HO.s2anet_get_bboxes_single([a[i] for a in odm_cls], [a[i] for a in odm_box],
                                                 [r[i].reshape(-1, 5) for r in refine], cmp_ge=cmp_ge)
        assert len(es) > 20 and len(es) == len(scores)
        np.testing.assert_allclose(scores, es, rtol=1e-5, atol=1e-6)
        assert np.array_equal(labels.astype(np.int64), el.astype(np.int64))
        np.testing.assert_allclose(_corner_sets(polys), _corner_sets(_rect_corners(eb)), rtol=0, atol=2e-3)


def test_oriented_rpn_proposals_vs_restatement(dev):
    """OrientedRPNHead.get_bboxes on fixed head outputs: per-level top-k, midpoint-offset decode, per-level horizontal
    NMS, best nms_post overall (oriented_rpn_head.py:L128-226) -- the fixed-shape proposal table's valid rows must be
    the restatement's proposals, in order"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.models.roi_heads.oriented_rpn_head import INVALID_SCORE, OrientedRPNHead
    rng = np.random.default_rng(33)
    N, size = 2, 256
    head = OrientedRPNHead(in_channels=256, nms_pre=300, nms_post=200, nms_thresh=0.8).to(dev).eval()
    strides = [4, 8, 16, 32, 64]
    sizes = [(size // s, size // s) for s in strides]
    A = head.num_anchors
    cls = [rng.normal(-1.0, 2.0, size=(N, A) + sz).astype(np.float32) for sz in sizes]
    reg = [rng.normal(0, 0.25, size=(N, A * 6) + sz).astype(np.float32) for sz in sizes]
    t = lambda a: torch.from_numpy(a).to(dev)
    targets = [dict(img_size=(size, size), pad_shape=(size, size)) for _ in range(N)]
    with torch.no_grad():
        tables = head.get_bboxes([t(a) for a in cls], [t(a) for a in reg], targets)
        anchors = [a.cpu().numpy() for a in head.anchor_generator.grid_anchors(sizes, device=dev)]
    for i in range(N):
        tab = tables[i].cpu().numpy()
        assert tab.shape == (200, 6)
        valid = tab[:, 5] > INVALID_SCORE
        ref = HO.oriented_rpn_proposals_single([c[i] for c in cls], [r[i] for r in reg], anchors, nms_pre=300,
                                               nms_post=200, nms_thresh=0.8)
        assert valid.sum() == len(ref) > 50
        assert np.all(valid[:len(ref)])                      # valid rows first, padding after
        np.testing.assert_allclose(tab[:len(ref), 5], ref[:, 5], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(tab[:len(ref), :4], ref[:, :4], rtol=1e-5, atol=2e-3)
        np.testing.assert_allclose(np.cos(2 * tab[:len(ref), 4]), np.cos(2 * ref[:, 4]), rtol=0, atol=1e-4)


# ------------------------------------------------------------------------------------------- Oriented R-CNN RoI head
def _trunk_pair(dev):
    """a fixed smooth map box -> (cls_score, bbox_pred) standing in for RoI features + FC layers, as numpy and torch"""
    rng = np.random.default_rng(77)
    wc = rng.standard_normal((5, 16)).astype(np.float32)
    wr = rng.standard_normal((5, 5)).astype(np.float32)
    sc = np.asarray([1 / 200.0, 1 / 200.0, 1 / 60.0, 1 / 60.0, 1.0], np.float32)

    def f_np(boxes):
        b = boxes.astype(np.float32) * sc
        return (np.sin(b @ wc) * 2).astype(np.float32), (np.sin(b @ wr) * 0.3).astype(np.float32)

    twc, twr, tsc = (torch.from_numpy(v).to(dev) for v in (wc, wr, sc))

    def f_t(feats, rois):
        b = rois[:, 1:] * tsc
        return torch.sin(b @ twc) * 2, torch.sin(b @ twr) * 0.3
    return f_np, f_t


def _proposal_table(rng, gts, n_random, n_pad, size):
    """gt jitters (some above, some below IoU 0.5), random boxes and padding rows (score < 0), as the RPN hands over"""
    jit = []
    for g in gts:
        for s in (0.03, 0.12, 0.4):
            j = g.copy()
            j[:2] += rng.normal(0, s, 2) * j[2:4]
            j[2:4] *= np.exp(rng.normal(0, s, 2))
            j[4] = -j[4] + rng.normal(0, s * 0.5)         # proposals live in the head's (negated) angle convention
            jit.append(j)
    rnd = I.random_obbs(rng, n_random, extent=float(size), wh=(10.0, 90.0))
    boxes = np.concatenate([np.asarray(jit, np.float32), rnd], 0)
    scores = rng.uniform(0.05, 1.0, size=(boxes.shape[0], 1)).astype(np.float32)
    table = np.concatenate([boxes, scores], 1)
    pad = np.concatenate([np.zeros((n_pad, 5), np.float32), -np.ones((n_pad, 1), np.float32)], 1)
    return np.concatenate([table, pad], 0).astype(np.float32)


def _oriented_head(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.models.roi_heads.oriented_head import OrientedHead
    return OrientedHead(num_classes=15, in_channels=256).to(dev)


def test_oriented_head_targets_and_loss_vs_restatement(dev):
    rng = np.random.default_rng(41)
    size = 512
    gts = [I.random_obbs(rng, k, extent=float(size), wh=(20.0, 110.0)) for k in (7, 4)]
    labels = [rng.integers(1, 16, size=g.shape[0]).astype(np.int32) for g in gts]
    tables = [_proposal_table(rng, g, 120, 40, size) for g in gts]
    f_np, f_t = _trunk_pair(dev)
    per_img = [HO.oriented_head_targets(t[t[:, 5] >= 0], g, l) for t, g, l in zip(tables, gts, labels)]
    for t in per_img:      # the premise of the restatement: the sampler keeps every candidate
        assert t[0].shape[0] <= 512 and int((t[1] < 15).sum()) <= 128 and int((t[1] < 15).sum()) >= 3
    ref = HO.oriented_head_loss(per_img, f_np)
    head = _oriented_head(dev).train()
    head._trunk = f_t
    t = lambda a: torch.from_numpy(a).to(dev)
    targets = [dict(rboxes=t(g), labels=t(l), img_size=(size, size), scale_factor=1.0) for g, l in zip(gts, labels)]
    out = head.forward_train(None, [t(tb) for tb in tables], targets)
    assert set(out) == set(ref)
    for k in ref:
        assert float(out[k]) == pytest.approx(ref[k], rel=3e-5, abs=1e-6), k
    assert ref["orcnn_bbox_loss"] > 0


def test_oriented_head_detections_vs_restatement(dev):
    rng = np.random.default_rng(43)
    size = 512
    gts = I.random_obbs(rng, 6, extent=float(size), wh=(20.0, 110.0))
    table = _proposal_table(rng, gts, 60, 25, size)
    f_np, f_t = _trunk_pair(dev)
    alive = table[:, 5] >= 0
    cls, reg = f_np(table[alive, :5])
    polys, scores, labels = HO.oriented_head_detections(table[alive], cls, reg, scale_factor=2.0)
    head = _oriented_head(dev).eval()
    head._trunk = f_t
    target = dict(img_size=(size, size), scale_factor=2.0)
    with torch.no_grad():
        (p, s, l), = head.forward_test(None, [torch.from_numpy(table).to(dev)], [target])
    assert l.shape[0] == labels.shape[0] > 20
    assert np.array_equal(l.cpu().numpy(), labels)
    np.testing.assert_allclose(s.cpu().numpy(), scores, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(p.cpu().numpy(), polys, rtol=0, atol=2e-3)


# --------------------------------------------------------------------------------------- RoI-Transformer R-CNN stages
def _roitrans_trunks(dev):
    rng = np.random.default_rng(91)
    w1c, w1r = rng.standard_normal((4, 16)).astype(np.float32), rng.standard_normal((4, 5)).astype(np.float32)
    w2c, w2r = rng.standard_normal((5, 16)).astype(np.float32), rng.standard_normal((5, 80)).astype(np.float32)
    s1 = np.full((4,), 1 / 150.0, np.float32)
    s2 = np.asarray([1 / 150.0, 1 / 150.0, 1 / 60.0, 1 / 60.0, 1.0], np.float32)

    def t1(rois):
        b = rois[:, 1:].astype(np.float32) * s1
        return (np.sin(b @ w1c) * 2).astype(np.float32), (np.sin(b @ w1r) * 0.5).astype(np.float32)

    def t2(rrois):
        b = rrois[:, 1:].astype(np.float32) * s2
        return (np.sin(b @ w2c) * 2).astype(np.float32), (np.sin(b @ w2r) * 0.5).astype(np.float32)

    tt = {k: torch.from_numpy(v).to(dev) for k, v in dict(w1c=w1c, w1r=w1r, w2c=w2c, w2r=w2r, s1=s1, s2=s2).items()}

    def g1(rois):
        b = rois[:, 1:] * tt["s1"]
        return torch.sin(b @ tt["w1c"]) * 2, torch.sin(b @ tt["w1r"]) * 0.5

    def g2(rrois):
        b = rrois[:, 1:] * tt["s2"]
        return torch.sin(b @ tt["w2c"]) * 2, torch.sin(b @ tt["w2r"]) * 0.5
    return t1, t2, g1, g2


def test_roi_transformer_rcnn_stages_vs_restatement(dev):
    """RoITransformer.execute_train with the network replaced by fixed maps box -> (scores, deltas): both R-CNN stages
    (assignment, gt-as-proposal, targets, losses, the stage-1 -> stage-2 refinement and its gt filtering)"""
    import jdet_amd.models  # noqa: F401
    from jdet_amd.config.named import roitrans_train_cfg
    from jdet_amd.utils.registry import MODELS, build_from_cfg
    rng = np.random.default_rng(47)
    size = 512
    gt_obbs = [I.random_obbs(rng, k, extent=float(size), wh=(24.0, 120.0)) for k in (6, 4)]
    from oracle import box_oracle as B
    gt_hbbs = [B.obb2hbb(g).astype(np.float32) for g in gt_obbs]
    labels = [rng.integers(1, 16, size=g.shape[0]).astype(np.int64) for g in gt_obbs]
    props = []
    for gh in gt_hbbs:
        jit = np.concatenate([gh + rng.normal(0, s, gh.shape).astype(np.float32) * (gh[:, 2:] - gh[:, :2]).mean()
                              for s in (0.03, 0.1, 0.3)], 0)
        c = rng.uniform(0, size, (90, 2))
        wh = rng.uniform(10, 100, (90, 2))
        rnd = np.concatenate([c - wh / 2, c + wh / 2], 1)
        props.append(np.concatenate([jit, rnd], 0).astype(np.float32))
    t1, t2, g1, g2 = _roitrans_trunks(dev)
    ref = HO.roitrans_rcnn_losses(props, gt_hbbs, gt_obbs, labels, t1, t2)

    model = build_from_cfg(roitrans_train_cfg("Resnet50")["model"], MODELS)
    t = lambda a: torch.from_numpy(a).to(dev)
    tables = [torch.cat([t(p), torch.ones((p.shape[0], 1), device=dev)], 1) for p in props]
    tables = [torch.cat([tb, torch.cat([tb.new_zeros((30, 4)), -tb.new_ones((30, 1))], 1)]) for tb in tables]   # padding

    class Rpn(torch.nn.Module):
        def forward(self, feats):
            return ()

        def loss(self, *a, **k):
            return {}

        def get_bboxes(self, *a, **k):
            return tables

    class Extractor(torch.nn.Module):
        num_inputs, w_enlarge, h_enlarge = 4, 1.2, 1.4

        def forward(self, feats, rois):
            return rois

    class Backbone(torch.nn.Module):
        def forward(self, im):
            return [im]

    model.backbone, model.neck, model.rpn_head = Backbone(), None, Rpn()
    model.bbox_roi_extractor, model.rbbox_roi_extractor = Extractor(), Extractor()
    model.bbox_head.forward, model.rbbox_head.forward = g1, g2
    model.to(dev).train()
    targets = [dict(ori_img_size=(size, size), img_size=(size, size), pad_shape=(size, size), scale_factor=1.0,
                    hboxes=t(gh), rboxes=t(go), labels=t(gl)) for gh, go, gl in zip(gt_hbbs, gt_obbs, labels)]
    out = model.execute_train(torch.zeros(2, 3, 8, 8, device=dev), targets)
    for k, v in ref.items():
        assert float(out[k]) == pytest.approx(v, rel=5e-5, abs=1e-6), k
    assert ref["s0.rbbox_loss_bbox"] > 0 and ref["s1.rbbox_loss_bbox"] > 0


# ------------------------------------------------------------------------------------------------ RotatedRetinaHead
def _retina_outputs(rng, N, size):
    sizes = [(max(size // s, 1), max(size // s, 1)) for s in STRIDES]
    cls = [rng.normal(-2.5, 1.5, size=(N, 9 * 15) + sz).astype(np.float32) for sz in sizes]
    box = [rng.normal(0, 0.15, size=(N, 9 * 5) + sz).astype(np.float32) for sz in sizes]
    return cls, box


def _retina_head(dev):
    import jdet_amd.models  # noqa: F401
    from jdet_amd.config.named import RETINANET_CFG
    from jdet_amd.utils.registry import HEADS, build_from_cfg
    return build_from_cfg(RETINANET_CFG["model"]["bbox_head"], HEADS).to(dev)


def test_retina_head_loss_vs_restatement(dev):
    """9 anchors per location (3 octave scales x 3 ratios, anchor-fastest order), focal + L1 losses"""
    rng = np.random.default_rng(51)
    N, size = 2, 256
    cls, box = _retina_outputs(rng, N, size)
    gts = [I.random_obbs(rng, k, extent=float(size), wh=(16.0, 120.0)) for k in (8, 5)]
    for g in gts:          # axis-aligned-ish boxes so that the horizontal anchors reach IoU 0.5
        g[:, 4] = rng.normal(0, 0.08, g.shape[0])
    labels = [rng.integers(1, 16, size=g.shape[0]).astype(np.int32) for g in gts]
    ref = HO.retina_loss(cls, box, gts, labels, STRIDES)
    head = _retina_head(dev).train()
    t = lambda a: torch.from_numpy(a).to(dev)
    metas = [dict(img_shape=(size, size), scale_factor=1.0, pad_shape=(size, size)) for _ in range(N)]
    out = head.loss([t(a) for a in cls], [t(a) for a in box], [t(g) for g in gts], [t(l) for l in labels], metas)
    for k in ref:
        got = np.asarray([float(v) for v in out[k]])
        np.testing.assert_allclose(got, np.asarray(ref[k]), rtol=3e-5, atol=1e-7, err_msg=k)
    assert sum(v > 0 for v in ref["loss_bbox"]) >= 2


def test_retina_head_get_bboxes_vs_restatement(dev):
    rng = np.random.default_rng(52)
    N, size = 2, 256
    cls, box = _retina_outputs(rng, N, size)
    for a in cls:
        a += (rng.uniform(size=a.shape) < 0.01) * rng.uniform(2.0, 6.0, size=a.shape).astype(np.float32)
    head = _retina_head(dev).eval()
    t = lambda a: torch.from_numpy(a).to(dev)
    metas = [dict(img_shape=(size, size), scale_factor=1.0, pad_shape=(size, size)) for _ in range(N)]
    with torch.no_grad():
        res = head.get_bboxes([t(a) for a in cls], [t(a) for a in box], metas)
    from jdet_amd.ops import nms_rotated as NR
    cmp_ge = 1 if NR.REFERENCE_RULE == "cpu" else 0
    for i in range(N):
        polys, scores, labels = (v.cpu().numpy() for v in res[i])
        eb, es, el = HO.retina_get_bboxes_single([a[i] for a in cls], [a[i] for a in box], STRIDES, cmp_ge=cmp_ge)
        assert len(es) > 20 and len(es) == len(scores)
        np.testing.assert_allclose(scores, es, rtol=1e-5, atol=1e-6)
        assert np.array_equal(labels.astype(np.int64), el.astype(np.int64))
        np.testing.assert_allclose(_corner_sets(polys), _corner_sets(_rect_corners(eb)), rtol=0, atol=2e-3)
