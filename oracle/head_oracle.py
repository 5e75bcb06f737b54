"""TEST INFRASTRUCTURE (checker, never shipped, never a fallback): numpy restatement of the head-level tensor programs
of the reference -- what S2ANetHead does between the network outputs and the loss dictionary / the detections.

Composes oracle/box_oracle.py (anchors, assigner, codecs) and the C++ oracle (rotated IoU, rotated NMS) with the
losses and the multi-class NMS stated here; each function cites the reference lines it follows.  Parity status:
unpinned by reference execution (Jittor is not importable) -- the pieces are held to closed forms in
tests/test_box_oracle.py; this module pins how the heads COMPOSE them (level order, averaging factors, label
conventions, score thresholds, keep order).
"""
import numpy as np

from oracle import box_oracle as B
from oracle import oracle as O


def sigmoid_focal_loss(pred, target, weight, gamma=2.0, alpha=0.25, avg_factor=None):
    """models/losses/focal_loss.py:L5-56 (reduction "mean").  pred (n, C) logits, target (n,) with 0 = background and
    class c stored as c (one-hot column c - 1: `(index + 1) == target`, L37-38), weight (n,) broadcast over classes."""
    pred = pred.astype(np.float64)
    n, C = pred.shape
    t = (np.arange(C)[None, :] + 1 == target[:, None]).astype(np.float64)
    max_val = np.clip(-pred, 0, None)
    ce = (1 - t) * pred + max_val + np.log(np.maximum(np.exp(-max_val) + np.exp(-pred - max_val), 1e-10))   # L8-13
    ce = ce * weight[:, None]                                                                                 # L14-15
    p = 1.0 / (1.0 + np.exp(-pred))
    p_t = p * t + (1 - p) * (1 - t)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * t + (1 - alpha) * (1 - t)) * loss
    return loss.sum() / (loss.size if avg_factor is None else avg_factor)


def smooth_l1_loss(pred, target, weight, beta=1.0 / 9.0, avg_factor=None):
    """models/losses/smooth_l1_loss.py:L5-26 (reduction "mean"): flag*0.5*d^2/beta + (1-flag)*(d - 0.5*beta)"""
    d = np.abs(pred.astype(np.float64) - target.astype(np.float64))
    if beta != 0:
        flag = (d < beta).astype(np.float64)
        loss = flag * 0.5 * d * d / beta + (1 - flag) * (d - 0.5 * beta)
    else:
        loss = d
    w = weight[:, None] if weight.ndim == 1 else weight
    loss = loss * w
    return loss.sum() / (max(loss.shape[0], 1) if avg_factor is None else avg_factor)


def anchor_targets(anchors_per_image, gts, labels, cfg, num_level_anchors):
    """models/boxes/anchor_target.py:L60-102 with sampling=False: per image the single-image targets, then
    `images_to_levels`; num_total_pos = sum over images of max(#positives, 1) (L79-80)."""
    per_img = [B.anchor_target_single(a, g, l, cfg["pos_iou_thr"], cfg["neg_iou_thr"], cfg["min_pos_iou"])
               for a, g, l in zip(anchors_per_image, gts, labels)]
    num_total_pos = sum(max(len(t[4]), 1) for t in per_img)
    out = []
    for k in range(4):
        stacked = np.stack([t[k] for t in per_img], 0)                 # (N, A[, 5])
        lv, start = [], 0
        for n in num_level_anchors:
            lv.append(stacked[:, start:start + n])
            start += n
        out.append(lv)
    return out[0], out[1], out[2], out[3], num_total_pos


def _nhwc_rows(t, C):
    """(N, C, H, W) -> (N*H*W, C): `permute(0, 2, 3, 1).reshape(-1, C)` (s2anet_head.py:L443-444)"""
    return np.transpose(t, (0, 2, 3, 1)).reshape(-1, C)


def s2anet_loss(fam_cls, fam_box, refine_anchors, odm_cls, odm_box, gts, labels, strides, anchor_scale=4,
                fam_cfg=None, odm_cfg=None, gamma=2.0, alpha=0.25, beta=1.0 / 9.0):
    """models/roi_heads/s2anet_head.py:L322-428 (+ L430-507): per-level loss lists of the FAM and the ODM.
    fam_* / odm_*: lists over levels of (N, C, H, W); refine_anchors: list over levels of (N, H, W, 5)."""
    dflt = dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0)
    fam_cfg, odm_cfg = fam_cfg or dflt, odm_cfg or dflt
    N = fam_cls[0].shape[0]
    Ccls = fam_cls[0].shape[1]
    sizes = [tuple(t.shape[-2:]) for t in odm_cls]
    init = [B.grid_anchors_s2anet(s, [anchor_scale], [1.0], sz, s) for s, sz in zip(strides, sizes)]    # L339
    nla = [a.shape[0] for a in init]
    out = {}
    for name, cls, box, anchors_img, cfg in (
            ("fam", fam_cls, fam_box, [np.concatenate(init, 0)] * N, fam_cfg),
            ("odm", odm_cls, odm_box,
             [np.concatenate([refine_anchors[l][i].reshape(-1, 5) for l in range(len(sizes))], 0) for i in range(N)],
             odm_cfg)):
        lab, lw, bt, bw, num_pos = anchor_targets(anchors_img, gts, labels, cfg, nla)
        out["loss_%s_cls" % name] = [
            sigmoid_focal_loss(_nhwc_rows(cls[l], Ccls), lab[l].reshape(-1), lw[l].reshape(-1), gamma, alpha, num_pos)
            for l in range(len(sizes))]
        out["loss_%s_bbox" % name] = [
            smooth_l1_loss(_nhwc_rows(box[l], 5), bt[l].reshape(-1, 5), bw[l].reshape(-1, 5), beta, num_pos)
            for l in range(len(sizes))]
    return out


def multiclass_nms_rotated(boxes, scores_with_bg, score_thr, iou_thr, max_num, cmp_ge=1):
    """ops/nms_rotated.py:L552-596 for (n, 5) boxes: every (box, class) pair above the threshold, NMS inside each
    class (ml_nms_rotated L512-525: boxes of different labels never suppress each other, L283-286), survivors in
    descending score order, cut at max_num.  Returns (boxes (k,5), scores (k,), labels (k,) 0-based)."""
    scores = scores_with_bg[:, 1:]
    valid = scores > score_thr
    bi, ci = np.nonzero(valid)                      # row-major: the order of `bboxes[valid_mask]`
    if len(bi) == 0:
        return np.zeros((0, 5), np.float32), np.zeros((0,), np.float32), np.zeros((0,), np.int64)
    b, s, l = boxes[bi], scores[bi, ci], ci
    keep = np.zeros(len(b), bool)
    for c in np.unique(l):
        idx = np.nonzero(l == c)[0]
        idx = idx[np.argsort(-s[idx], kind="stable")]                 # the class's boxes, best first
        k = O.nms_rotated_keep(b[idx], np.arange(len(idx), dtype=np.int32), iou_thr, cmp_ge=cmp_ge)
        keep[idx[np.asarray(k, bool)]] = True
    kept = np.nonzero(keep)[0]
    kept = kept[np.argsort(-s[kept], kind="stable")]
    if len(kept) > max_num:
        kept = kept[:max_num]
    return b[kept], s[kept], l[kept]


def s2anet_get_bboxes_single(odm_cls, odm_box, refine_anchors, score_thr=0.05, iou_thr=0.1, max_per_img=2000,
                             nms_pre=2000, scale_factor=1.0, cmp_ge=1):
    """s2anet_head.py:L546-601 for one image: per level sigmoid scores, top nms_pre by the best class, decode against
    the refined anchors (delta2bbox_rotated, unit stds), concatenate, rescale, background column, multi-class NMS.
    odm_cls / odm_box: lists over levels of (C, H, W); refine_anchors: list over levels of (H*W, 5)."""
    mb, ms = [], []
    for cls, box, anchors in zip(odm_cls, odm_box, refine_anchors):
        C = cls.shape[0]
        sc = 1.0 / (1.0 + np.exp(-np.transpose(cls, (1, 2, 0)).reshape(-1, C).astype(np.float64)))
        bp = np.transpose(box, (1, 2, 0)).reshape(-1, 5)
        if 0 < nms_pre < sc.shape[0]:
            top = np.argsort(-sc.max(1), kind="stable")[:nms_pre]
            anchors, bp, sc = anchors[top], bp[top], sc[top]
        mb.append(B.delta2bbox_rotated(anchors, bp))
        ms.append(sc.astype(np.float32))
    mb, ms = np.concatenate(mb, 0), np.concatenate(ms, 0)
    mb = mb.copy()
    mb[:, :4] /= scale_factor
    ms = np.concatenate([np.zeros((ms.shape[0], 1), np.float32), ms], 1)
    return multiclass_nms_rotated(mb.astype(np.float32), ms, score_thr, iou_thr, max_per_img, cmp_ge)


def hbb_nms(boxes, scores, thr):
    """`jt.nms(dets (n,5), thr)` as SURVEY 8c restates it (Jittor-internal, unpinned): greedy in descending score order
    (stable), suppress at IoU > thr, areas without the +1 px convention; returns the kept indices in score order."""
    order = np.argsort(-scores, kind="stable")
    b = boxes.astype(np.float64)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    dead = np.zeros(len(b), bool)
    keep = []
    for i in order:
        if dead[i]:
            continue
        keep.append(i)
        iw = np.clip(np.minimum(b[i, 2], b[:, 2]) - np.maximum(b[i, 0], b[:, 0]), 0, None)
        ih = np.clip(np.minimum(b[i, 3], b[:, 3]) - np.maximum(b[i, 1], b[:, 1]), 0, None)
        inter = iw * ih
        iou = inter / np.maximum(area[i] + area - inter, 1e-300)
        dead |= iou > thr
    return np.asarray(keep, np.int64)


def oriented_rpn_proposals_single(cls_scores, bbox_preds, mlvl_anchors, nms_pre=2000, nms_post=2000, nms_thresh=0.8,
                                  min_bbox_size=0, means=(0.,) * 6, stds=(1., 1., 1., 1., 0.5, 0.5)):
    """models/roi_heads/oriented_rpn_head.py:L128-226 for one image.  cls_scores: list over levels of (A, H, W);
    bbox_preds: (A*6, H, W); mlvl_anchors: (H*W*A, 4) in grid order.  Returns (k, 6) [obb, score], best first."""
    sc, dl, an, ids = [], [], [], []
    for idx, (cls, reg, anchors) in enumerate(zip(cls_scores, bbox_preds, mlvl_anchors)):
        s = 1.0 / (1.0 + np.exp(-np.transpose(cls, (1, 2, 0)).reshape(-1).astype(np.float64)))      # L167-169
        d = np.transpose(reg, (1, 2, 0)).reshape(-1, 6)                                                 # L177
        if 0 < nms_pre < s.shape[0]:                                                                    # L180-187
            top = np.argsort(-s, kind="stable")[:nms_pre]
            s, d, anchors = s[top], d[top], anchors[top]
        sc.append(s.astype(np.float32))
        dl.append(d)
        an.append(anchors)
        ids.append(np.full(s.shape[0], idx, np.int64))
    sc, dl, an, ids = np.concatenate(sc), np.concatenate(dl), np.concatenate(an), np.concatenate(ids)
    prop = B.midpoint_offset_decode(an, dl, means, stds)                                               # L198
    if min_bbox_size >= 0:                                                                              # L201-207
        ok = (prop[:, 2] > min_bbox_size) & (prop[:, 3] > min_bbox_size)
        prop, sc, ids = prop[ok], sc[ok], ids[ok]
    h = B.obb2hbb(prop).astype(np.float64)                                                              # L209-212
    h = h + (ids.astype(np.float64) * (h.max() - h.min() + 1))[:, None]
    keep = hbb_nms(h, sc, nms_thresh)                                                                   # L214-215
    dets = np.concatenate([prop, sc[:, None]], 1)[keep]
    return dets[:nms_post]
