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


# ---- RoI-Transformer codecs (python/jdet/ops/bbox_transforms.py) -- numpy restatement, test-only ----------
def hbb2obb_v2(boxes):
    """L34-44"""
    ex_h = boxes[:, 2] - boxes[:, 0] + 1.0
    ex_w = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * (ex_h - 1.0)
    cy = boxes[:, 1] + 0.5 * (ex_w - 1.0)
    return np.stack([cx, cy, ex_w, ex_h, -np.ones_like(cx) * np.pi / 2], 1).astype(np.float32)


def dbbox2delta_v3(p, g, means, stds):
    """L7-32"""
    p, g = p.astype(np.float32), g.astype(np.float32)
    co = g[:, 0:2] - p[:, 0:2]
    dx = (np.cos(p[:, 4]) * co[:, 0] + np.sin(p[:, 4]) * co[:, 1]) / p[:, 2]
    dy = (-np.sin(p[:, 4]) * co[:, 0] + np.cos(p[:, 4]) * co[:, 1]) / p[:, 3]
    d = np.stack([dx, dy, np.log(g[:, 2] / p[:, 2]), np.log(g[:, 3] / p[:, 3]), g[:, 4] - p[:, 4]], -1)
    return ((d - np.asarray(means, np.float32)[None]) / np.asarray(stds, np.float32)[None]).astype(np.float32)


def delta2dbbox(rrois, deltas, means, stds, angle_scale, wh_ratio_clip=16 / 1000):
    """v3 (angle_scale 1, L279-321) and v2 (angle_scale pi/2, L323-360); deltas (n, 5*k)"""
    k = deltas.shape[1] // 5
    d = deltas * np.tile(np.asarray(stds, np.float32), k)[None] + np.tile(np.asarray(means, np.float32), k)[None]
    dx, dy, dw, dh, da = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
    mr = abs(np.log(wh_ratio_clip))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    rx, ry, rw, rh, ra = [rrois[:, i:i + 1] for i in range(5)]
    gx = dx * rw * np.cos(ra) - dy * rh * np.sin(ra) + rx
    gy = dx * rw * np.sin(ra) + dy * rh * np.cos(ra) + ry
    out = np.stack([gx, gy, rw * np.exp(dw), rh * np.exp(dh), angle_scale * da + ra], -1)
    return out.reshape(deltas.shape).astype(np.float32)


def choose_best_Rroi_batch(r):
    """L444-463 (returns a copy)"""
    r = r.copy()
    w, h = r[:, 2].copy(), r[:, 3].copy()
    idx = w < h
    r[idx, 2], r[idx, 3] = h[idx], w[idx]
    r[idx, 4] = r[idx, 4] + np.pi / 2.
    r[:, 4] = r[:, 4] % np.pi
    return r


def choose_best_obb_batch(g0):
    """L465-479"""
    g = g0.copy()
    w, h = g0[:, 2], g0[:, 3]
    g[:, 4] = (g[:, 4] - np.pi / 4.) % np.pi
    idx = g[:, 4] >= np.pi / 2
    g[idx, 2], g[idx, 3] = h[idx], w[idx]
    g[idx, 4] = g[idx, 4] - np.pi / 2.
    g[:, 4] = g[:, 4] - np.pi * 3. / 4.
    return g


def best_match_dbbox2delta(rrois, gt, means, stds):
    """choose_best_match_batch L237-266 (row loop) + dbbox2delta_v2 L206-235"""
    new = np.zeros_like(gt)
    for i in range(gt.shape[0]):
        x, y, w, h, a = gt[i]
        ext = [(x, y, w, h, a), (x, y, h, w, a + np.pi / 2.), (x, y, w, h, a + np.pi), (x, y, h, w, a + np.pi * 3 / 2.)]
        dist = [(rrois[i, 4] - e[4]) % (2 * np.pi) for e in ext]
        dist = [min(d, 2 * np.pi - d) for d in dist]
        new[i] = ext[int(np.argmin(dist))]
    new[:, 4] = new[:, 4] % (2 * np.pi)
    co = new[:, 0:2] - rrois[:, 0:2]
    ra = rrois[:, 4]
    dx = (np.cos(ra) * co[:, 0] + np.sin(ra) * co[:, 1]) / rrois[:, 2]
    dy = (-np.sin(ra) * co[:, 0] + np.cos(ra) * co[:, 1]) / rrois[:, 3]
    da = new[:, 4] - ra
    dist = da % (2 * np.pi)
    dist = np.minimum(dist, 2 * np.pi - dist)
    dist = np.where(np.sin(da) < 0, -dist, dist) / (np.pi / 2.)
    d = np.stack([dx, dy, np.log(new[:, 2] / rrois[:, 2]), np.log(new[:, 3] / rrois[:, 3]), dist], -1)
    return ((d - np.asarray(means)[None]) / np.asarray(stds)[None]).astype(np.float32)


def bbox2delta(p, g, means, stds):
    """L179-204"""
    px, py = (p[:, 0] + p[:, 2]) * 0.5, (p[:, 1] + p[:, 3]) * 0.5
    pw, ph = p[:, 2] - p[:, 0] + 1.0, p[:, 3] - p[:, 1] + 1.0
    gx, gy = (g[:, 0] + g[:, 2]) * 0.5, (g[:, 1] + g[:, 3]) * 0.5
    gw, gh = g[:, 2] - g[:, 0] + 1.0, g[:, 3] - g[:, 1] + 1.0
    d = np.stack([(gx - px) / pw, (gy - py) / ph, np.log(gw / pw), np.log(gh / ph)], -1)
    return ((d - np.asarray(means)[None]) / np.asarray(stds)[None]).astype(np.float32)


def delta2bbox(rois, deltas, means, stds, max_shape=None, wh_ratio_clip=16 / 1000):
    """L362-396, deltas (n,4)"""
    d = deltas * np.asarray(stds, np.float32)[None] + np.asarray(means, np.float32)[None]
    mr = abs(np.log(wh_ratio_clip))
    dw, dh = np.clip(d[:, 2], -mr, mr), np.clip(d[:, 3], -mr, mr)
    px, py = (rois[:, 0] + rois[:, 2]) * 0.5, (rois[:, 1] + rois[:, 3]) * 0.5
    pw, ph = rois[:, 2] - rois[:, 0] + 1.0, rois[:, 3] - rois[:, 1] + 1.0
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    gx, gy = px + pw * d[:, 0], py + ph * d[:, 1]
    x1, y1, x2, y2 = gx - gw * 0.5 + 0.5, gy - gh * 0.5 + 0.5, gx + gw * 0.5 - 0.5, gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1, x2 = np.clip(x1, 0, max_shape[1] - 1), np.clip(x2, 0, max_shape[1] - 1)
        y1, y2 = np.clip(y1, 0, max_shape[0] - 1), np.clip(y2, 0, max_shape[0] - 1)
    return np.stack([x1, y1, x2, y2], -1).astype(np.float32)


# ---- Oriented R-CNN box algebra and codecs ---------------------------------------------------------------------
def regular_theta(theta, start=-math.pi / 2, cycle=math.pi):
    """ops/bbox_transforms.py:L499-505 (mode '180'); `%` = floor-mod"""
    theta = np.asarray(theta, np.float32)
    return ((theta - np.float32(start)) % np.float32(cycle) + np.float32(start)).astype(np.float32)


def regular_obb(b):
    """ops/bbox_transforms.py:L507-517: w >= h (else swap and add pi/2), theta wrapped into [-pi/2, pi/2)"""
    b = np.asarray(b, np.float32)
    x, y, w, h, t = [b[..., i] for i in range(5)]
    m = (w > h).astype(np.float32)
    wr = w * m + h * (1 - m)
    hr = h * m + w * (1 - m)
    tr = regular_theta(t * m + (t + np.float32(math.pi / 2)) * (1 - m))
    return np.stack([x, y, wr, hr, tr], -1).astype(np.float32)


def obb2poly(b):
    """ops/bbox_transforms.py:L626-637: v1 = (w/2 cos, -w/2 sin), v2 = (-h/2 sin, -h/2 cos); c+v1+v2, c+v1-v2, c-v1-v2, c-v1+v2"""
    b = np.asarray(b, np.float32)
    c, w, h, t = b[..., :2], b[..., 2:3], b[..., 3:4], b[..., 4:5]
    Cos, Sin = np.cos(t), np.sin(t)
    v1 = np.concatenate([w / 2 * Cos, -w / 2 * Sin], -1)
    v2 = np.concatenate([-h / 2 * Sin, -h / 2 * Cos], -1)
    return np.concatenate([c + v1 + v2, c + v1 - v2, c - v1 - v2, c - v1 + v2], -1).astype(np.float32)


def obb2hbb(b):
    """ops/bbox_transforms.py:L640-646"""
    b = np.asarray(b, np.float32)
    c, w, h, t = b[..., :2], b[..., 2:3], b[..., 3:4], b[..., 4:5]
    Cos, Sin = np.cos(t), np.sin(t)
    bias = np.concatenate([np.abs(w / 2 * Cos) + np.abs(h / 2 * Sin), np.abs(w / 2 * Sin) + np.abs(h / 2 * Cos)], -1)
    return np.concatenate([c - bias, c + bias], -1).astype(np.float32)


def rectpoly2obb(p):
    """ops/bbox_transforms.py:L575-597"""
    p = np.asarray(p, np.float32)
    theta = np.arctan2(-(p[..., 3] - p[..., 1]), p[..., 2] - p[..., 0])
    Cos, Sin = np.cos(theta), np.sin(theta)
    M = np.stack([Cos, -Sin, Sin, Cos], -1).reshape(p.shape[:-1] + (2, 2))
    x, y = p[..., 0::2].mean(-1), p[..., 1::2].mean(-1)
    cp = p.reshape(p.shape[:-1] + (4, 2)) - np.stack([x, y], -1)[..., None, :]
    rp = cp @ np.swapaxes(M, -1, -2)
    w = rp[..., 0].max(-1) - rp[..., 0].min(-1)
    h = rp[..., 1].max(-1) - rp[..., 1].min(-1)
    return regular_obb(np.stack([x, y, w, h, theta], -1))


def midpoint_offset_encode(anchors, gt, means, stds):
    """MidpointOffsetCoder.encode, models/boxes/coder.py:L332-372"""
    a, g = np.asarray(anchors, np.float32), np.asarray(gt, np.float32)
    px, py = (a[:, 0] + a[:, 2]) * 0.5, (a[:, 1] + a[:, 3]) * 0.5
    pw, ph = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    hbb, poly = obb2hbb(g), obb2poly(g)
    gx, gy = (hbb[:, 0] + hbb[:, 2]) * 0.5, (hbb[:, 1] + hbb[:, 3]) * 0.5
    gw, gh = hbb[:, 2] - hbb[:, 0], hbb[:, 3] - hbb[:, 1]
    xc, yc = poly[:, 0::2], poly[:, 1::2]
    y_min, x_max = yc.min(1, keepdims=True), xc.max(1, keepdims=True)
    _x = xc.copy()
    _x[np.abs(yc - y_min) > 0.1] = -1000
    ga = _x.max(1)
    _y = yc.copy()
    _y[np.abs(xc - x_max) > 0.1] = -1000
    gb = _y.max(1)
    d = np.stack([(gx - px) / pw, (gy - py) / ph, np.log(gw / pw), np.log(gh / ph), (ga - gx) / gw, (gb - gy) / gh], -1)
    return ((d - np.asarray(means, np.float32)[None]) / np.asarray(stds, np.float32)[None]).astype(np.float32)


def midpoint_offset_decode(anchors, deltas, means, stds, wh_ratio_clip=16 / 1000):
    """MidpointOffsetCoder.decode, models/boxes/coder.py:L374-437 (one box per row)"""
    a = np.asarray(anchors, np.float32)
    d = np.asarray(deltas, np.float32) * np.asarray(stds, np.float32)[None] + np.asarray(means, np.float32)[None]
    dx, dy, dw, dh, da, db = [d[:, i] for i in range(6)]
    mr = np.float32(abs(math.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    px, py = (a[:, 0] + a[:, 2]) * 0.5, (a[:, 1] + a[:, 3]) * 0.5
    pw, ph = a[:, 2] - a[:, 0], a[:, 3] - a[:, 1]
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    da, db = np.clip(da, -0.5, 0.5), np.clip(db, -0.5, 0.5)
    ga, _ga, gb, _gb = gx + da * gw, gx - da * gw, gy + db * gh, gy - db * gh
    polys = np.stack([ga, y1, x2, gb, _ga, y2, x1, _gb], -1)
    center = np.stack([gx, gy] * 4, -1)
    cp = polys - center
    diag = np.sqrt(cp[:, 0::2] ** 2 + cp[:, 1::2] ** 2)
    scale = diag.max(-1, keepdims=True) / diag
    cp = cp * np.repeat(scale, 2, axis=-1)
    return rectpoly2obb((cp + center).astype(np.float32))


def oriented_delta_encode(rois, gt, means, stds):
    """OrientedDeltaXYWHTCoder.encode, models/boxes/coder.py:L449-479"""
    p, g = np.asarray(rois, np.float32), np.asarray(gt, np.float32)
    px, py, pw, ph, pt = [p[:, i] for i in range(5)]
    gx, gy, gw, gh, gt_ = [g[:, i] for i in range(5)]
    d1, d2 = regular_theta(gt_ - pt), regular_theta(gt_ - pt + np.float32(math.pi / 2))
    m = (np.abs(d1) < np.abs(d2)).astype(np.float32)
    gwr, ghr, dt = gw * m + gh * (1 - m), gh * m + gw * (1 - m), d1 * m + d2 * (1 - m)
    dx = (np.cos(-pt) * (gx - px) + np.sin(-pt) * (gy - py)) / pw
    dy = (-np.sin(-pt) * (gx - px) + np.cos(-pt) * (gy - py)) / ph
    d = np.stack([dx, dy, np.log(gwr / pw), np.log(ghr / ph), dt], -1)
    return ((d - np.asarray(means, np.float32)[None]) / np.asarray(stds, np.float32)[None]).astype(np.float32)


def oriented_delta_decode(rois, deltas, means, stds, wh_ratio_clip=16 / 1000):
    """OrientedDeltaXYWHTCoder.decode, models/boxes/coder.py:L481-518; deltas (n, ncls*5) -> (n, ncls*5)"""
    r, d = np.asarray(rois, np.float32), np.asarray(deltas, np.float32)
    k = d.shape[1] // 5
    d = d * np.tile(np.asarray(stds, np.float32), k)[None] + np.tile(np.asarray(means, np.float32), k)[None]
    dx, dy, dw, dh, dt = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
    mr = np.float32(abs(math.log(wh_ratio_clip)))
    dw, dh = np.clip(dw, -mr, mr), np.clip(dh, -mr, mr)
    px, py, pw, ph, pt = [r[:, i:i + 1] for i in range(5)]
    gx = dx * pw * np.cos(-pt) - dy * ph * np.sin(-pt) + px
    gy = dx * pw * np.sin(-pt) + dy * ph * np.cos(-pt) + py
    gw, gh = pw * np.exp(dw), ph * np.exp(dh)
    out = regular_obb(np.stack([gx, gy, gw, gh, regular_theta(dt + pt)], -1))
    return out.reshape(d.shape).astype(np.float32)
