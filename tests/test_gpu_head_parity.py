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
