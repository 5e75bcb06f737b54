"""numpy restatement of the reference's Python-level box programs.  TEST INFRASTRUCTURE ONLY
(imported by tests/ and bench.py's cpu_baseline leg; never by jdet_amd).

These reference functions are Jittor tensor programs that cannot be imported here (Jittor is not
installable), so unlike oracle/jdet_oracle.cpp this file is pinned only by closed-form cases and by
cross-checks listed in SURVEY.md 8(c): decode(encode(x)) = x up to norm_angle, zero AlignConv offsets
for an axis-aligned 3*stride anchor, hand-built overlap matrices covering every assigner branch.
Jittor-internal conventions it has to assume (floor-mod `%`, `safe_log` clamp, argmax ties ->
first index) are "parity unpinned" and stated at each use.
"""
import math

import numpy as np


def norm_angle(angle, rng=(-math.pi / 4, math.pi)):
    """models/boxes/box_ops.py:L176-178; `%` taken as floor-mod (numpy semantics)"""
    return (angle - np.float32(rng[0])) % np.float32(rng[1]) + np.float32(rng[0])


def bbox2delta_rotated(proposals, gt, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.)):
    """box_ops.py:L180-226"""
    p, g = proposals.astype(np.float32), gt.astype(np.float32)
    cosa, sina = np.cos(p[..., 4]), np.sin(p[..., 4])
    coord = g[..., 0:2] - p[..., 0:2]
    dx = (cosa * coord[..., 0] + sina * coord[..., 1]) / p[..., 2]
    dy = (-sina * coord[..., 0] + cosa * coord[..., 1]) / p[..., 3]
    dw = np.log(np.clip(g[..., 2] / p[..., 2], 1e-30, 1e30))  # jt.safe_log
    dh = np.log(np.clip(g[..., 3] / p[..., 3], 1e-30, 1e30))
    da = norm_angle(g[..., 4] - p[..., 4]) / np.float32(math.pi)
    deltas = np.stack((dx, dy, dw, dh, da), -1).astype(np.float32)
    return ((deltas - np.asarray(means, np.float32)[None]) / np.asarray(stds, np.float32)[None]).astype(np.float32)


def delta2bbox_rotated(rois, deltas, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.), wh_ratio_clip=16 / 1000):
    """box_ops.py:L229-285 (max_shape / clip_border never applied there)"""
    rois, deltas = rois.astype(np.float32), deltas.astype(np.float32)
    k = deltas.shape[1] // 5
    d = deltas * np.tile(np.asarray(stds, np.float32), k)[None] + np.tile(np.asarray(means, np.float32), k)[None]
    dx, dy, dw, dh, da = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
    mr = np.float32(abs(math.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    rx, ry, rw, rh, ra = [rois[:, i:i + 1] for i in range(5)]
    gx = dx * rw * np.cos(ra) - dy * rh * np.sin(ra) + rx
    gy = dx * rw * np.sin(ra) + dy * rh * np.cos(ra) + ry
    gw, gh = rw * np.exp(dw), rh * np.exp(dh)
    ga = norm_angle(np.float32(math.pi) * da + ra)
    return np.stack([gx, gy, gw, gh, ga], -1).reshape(deltas.shape).astype(np.float32)


def assign_wrt_overlaps(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou=0.0, match_low_quality=True,
                        gt_max_assign_all=True, gt_labels=None, labels_filled=0):
    """models/boxes/assigner.py:L160-219, literally (the per-gt loop included).
    argmax ties -> first index (Jittor's rule is unpinned)."""
    K, A = overlaps.shape
    assigned = np.full((A,), -1, np.int32)
    argmax_overlaps, max_overlaps = overlaps.argmax(0), overlaps.max(0)
    gt_argmax_overlaps, gt_max_overlaps = overlaps.argmax(1), overlaps.max(1)
    if isinstance(neg_iou_thr, float):
        assigned[(max_overlaps >= 0) & (max_overlaps < neg_iou_thr)] = 0
    elif isinstance(neg_iou_thr, tuple):
        assigned[(max_overlaps >= neg_iou_thr[0]) & (max_overlaps < neg_iou_thr[1])] = 0
    pos = max_overlaps >= pos_iou_thr
    assigned[pos] = argmax_overlaps[pos] + 1
    if match_low_quality:
        for i in range(K):
            if gt_max_overlaps[i] >= min_pos_iou:
                if gt_max_assign_all:
                    assigned[overlaps[i, :] == gt_max_overlaps[i]] = i + 1
                else:
                    assigned[gt_argmax_overlaps[i]] = i + 1
    labels = None
    if gt_labels is not None:
        labels = np.full((A,), labels_filled, np.int32)
        pi = np.nonzero(assigned > 0)[0]
        labels[pi] = gt_labels[assigned[pi] - 1]
    return assigned, max_overlaps.astype(np.float32), labels


def grid_anchors_s2anet(base_size, scales, ratios, featmap_size, stride, angles=(0.,)):
    """models/boxes/anchor_generator.py:L127-183"""
    w = h = base_size
    xc, yc = 0.5 * (w - 1), 0.5 * (h - 1)
    ratios, scales, angles = (np.asarray(v, np.float32) for v in (ratios, scales, angles))
    hr = np.sqrt(ratios)
    wr = 1 / hr
    ws = (w * wr[:, None, None] * scales[None, :, None] * np.ones_like(angles)[None, None, :]).reshape(-1)
    hs = (h * hr[:, None, None] * scales[None, :, None] * np.ones_like(angles)[None, None, :]).reshape(-1)
    an = np.tile(angles, len(scales) * len(ratios))
    base = np.stack([xc + 0 * ws, yc + 0 * ws, ws, hs, an], -1).astype(np.float32)
    fh, fw = featmap_size
    sx, sy = np.arange(fw) * stride, np.arange(fh) * stride
    xx, yy = np.tile(sx, fh), np.repeat(sy, fw)
    shifts = np.stack([xx, yy, 0 * xx, 0 * xx, 0 * xx], -1).astype(np.float32)
    return (base[None] + shifts[:, None]).reshape(-1, 5)


def align_conv_offsets(anchors, featmap_size, stride, kernel_size=3):
    """models/roi_heads/s2anet_head.py:L676-713 for one image: anchors (H*W,5) -> (2*k*k, H, W)"""
    a = anchors.astype(np.float32)
    fh, fw = featmap_size
    pad = (kernel_size - 1) // 2
    idx = np.arange(-pad, pad + 1, dtype=np.float32)
    yy, xx = np.meshgrid(idx, idx, indexing="ij")
    xx, yy = xx.reshape(-1), yy.reshape(-1)
    yc, xc = np.meshgrid(np.arange(fh, dtype=np.float32), np.arange(fw, dtype=np.float32), indexing="ij")
    xc, yc = xc.reshape(-1), yc.reshape(-1)
    x_conv, y_conv = xc[:, None] + xx, yc[:, None] + yy
    x_ctr, y_ctr, w, h, ang = (a[:, i] for i in range(5))
    x_ctr, y_ctr, w, h = x_ctr / stride, y_ctr / stride, w / stride, h / stride
    cos, sin = np.cos(ang), np.sin(ang)
    dw, dh = w / kernel_size, h / kernel_size
    x, y = dw[:, None] * xx, dh[:, None] * yy
    xr = cos[:, None] * x - sin[:, None] * y
    yr = sin[:, None] * x + cos[:, None] * y
    ox = xr + x_ctr[:, None] - x_conv
    oy = yr + y_ctr[:, None] - y_conv
    off = np.stack([oy, ox], -1)
    return off.reshape(a.shape[0], -1).transpose(1, 0).reshape(-1, fh, fw).astype(np.float32)


def anchor_target_single(anchors, gt_bboxes, gt_labels, pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0.0,
                         means=(0,) * 5, stds=(1,) * 5):
    """models/boxes/anchor_target.py:L105-180 for the S2ANet config (sampling=False -> PseudoSampler,
    allowed_border=-1, pos_weight=-1, all anchors valid).  IoU from the C++ oracle."""
    from oracle import oracle as O
    ov = O.box_iou_rotated(gt_bboxes, anchors)
    gt_inds, _, _ = assign_wrt_overlaps(ov, pos_iou_thr, neg_iou_thr, min_pos_iou, True, True, None)
    pos, neg = np.nonzero(gt_inds > 0)[0], np.nonzero(gt_inds == 0)[0]
    A = anchors.shape[0]
    bbox_targets, bbox_weights = np.zeros((A, 5), np.float32), np.zeros((A, 5), np.float32)
    labels, label_weights = np.zeros((A,), np.int32), np.zeros((A,), np.float32)
    if len(pos):
        bbox_targets[pos] = bbox2delta_rotated(anchors[pos], gt_bboxes[gt_inds[pos] - 1], means, stds)
        bbox_weights[pos] = 1.0
        labels[pos] = gt_labels[gt_inds[pos] - 1]
        label_weights[pos] = 1.0
    if len(neg):
        label_weights[neg] = 1.0
    return labels, label_weights, bbox_targets, bbox_weights, pos, neg
