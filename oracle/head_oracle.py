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
                             nms_pre=2000, scale_factor=1.0, cmp_ge=1, num_classes=None):
    """s2anet_head.py:L546-601 for one image: per level sigmoid scores, top nms_pre by the best class, decode against
    the refined anchors (delta2bbox_rotated, unit stds), concatenate, rescale, background column, multi-class NMS.
    odm_cls / odm_box: lists over levels of (C, H, W); refine_anchors: list over levels of (H*W, 5)."""
    mb, ms = [], []
    for cls, box, anchors in zip(odm_cls, odm_box, refine_anchors):
        C = cls.shape[0] if num_classes is None else num_classes       # (A*C, H, W): A anchors per location, A-fastest
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


def softmax_cross_entropy(cls_score, labels, weights, avg_factor):
    """models/losses/cross_entropy_loss.py:L6-13 (weighted_cross_entropy): sum(w * CE) / avg_factor"""
    z = cls_score.astype(np.float64)
    z = z - z.max(1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(1, keepdims=True))
    raw = -logp[np.arange(z.shape[0]), labels]
    return float((raw * weights).sum() / avg_factor)


def oriented_head_targets(proposals, gt_rboxes, gt_labels_1based, num_classes=15, pos_iou_thr=0.5, neg_iou_thr=0.5,
                          min_pos_iou=0.5, means=(0.,) * 5, stds=(0.1, 0.1, 0.2, 0.2, 0.1)):
    """models/roi_heads/oriented_head.py:L444-501 + L339-392 for ONE image when the sampler keeps every candidate
    (fewer than `num` candidates, positives within the quota): angle negation (L459-466), 0-based labels (L472),
    MaxIoUAssigner on the v1 rotated IoU with match_low_quality=False, gts prepended as proposals matched to
    themselves (sampler.py:L92-99), then get_bboxes_target_single: positives first, negatives after.
    Returns (boxes (n,5), labels (n,), label_weights (n,), bbox_targets (n,5), bbox_weights (n,5))."""
    gt = np.asarray(gt_rboxes, np.float32).copy()
    gt[:, -1] *= -1
    gl = np.asarray(gt_labels_1based, np.int64) - 1
    props = np.asarray(proposals, np.float32)[:, :5]
    overlaps = O.box_iou_rotated(gt, props, version=1)
    gt_inds, _, _ = B.assign_wrt_overlaps(overlaps, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality=False,
                                          gt_labels=gl.astype(np.int32), labels_filled=-1)
    boxes = np.concatenate([gt, props], 0)
    gt_inds = np.concatenate([np.arange(1, gt.shape[0] + 1), gt_inds.astype(np.int64)])
    pos, neg = np.nonzero(gt_inds > 0)[0], np.nonzero(gt_inds == 0)[0]
    n = len(pos) + len(neg)
    labels = np.full((n,), num_classes, np.int64)
    labels[:len(pos)] = gl[gt_inds[pos] - 1]
    label_weights = np.ones((n,), np.float32)
    bbox_targets = np.zeros((n, 5), np.float32)
    bbox_weights = np.zeros((n, 5), np.float32)
    if len(pos):
        bbox_targets[:len(pos)] = B.oriented_delta_encode(boxes[pos], gt[gt_inds[pos] - 1], means, stds)
        bbox_weights[:len(pos)] = 1
    return np.concatenate([boxes[pos], boxes[neg]], 0), labels, label_weights, bbox_targets, bbox_weights


def oriented_head_loss(per_image_targets, trunk, num_classes=15, beta=1.0):
    """oriented_head.py:L318-343: per_image_targets = list of oriented_head_targets(...) tuples; trunk(boxes (n,5)) ->
    (cls_score (n, C+1), bbox_pred (n, 5)) stands for RoI extraction + the FC layers (class-agnostic regression)."""
    boxes, labels, lw, bt, bw = [np.concatenate([t[k] for t in per_image_targets], 0) for k in range(5)]
    cls_score, bbox_pred = trunk(boxes)
    avg = max(float((lw > 0).sum()), 1.0)
    loss_cls = softmax_cross_entropy(cls_score, labels, lw, avg)
    pos = (labels >= 0) & (labels < num_classes)
    loss_bbox = 0.0
    if pos.any():
        loss_bbox = float(smooth_l1_loss(bbox_pred[pos], bt[pos], bw[pos], beta, avg_factor=bt.shape[0]))
    return dict(loss_cls=loss_cls, orcnn_bbox_loss=loss_bbox)


def oriented_head_detections(proposals, cls_score, bbox_pred, scale_factor=1.0, score_thresh=0.05, means=(0.,) * 5,
                             stds=(0.1, 0.1, 0.2, 0.2, 0.1)):
    """oriented_head.py:L404-440 + get_results L242-268 for one image: softmax, class-agnostic decode, rescale the
    first four box parameters, every (row, class) pair above the threshold in row-major order -> (polys (k,8),
    scores (k,), labels (k,) 0-based).  (No NMS here: the network's merge step applies it.)"""
    z = cls_score.astype(np.float64)
    z = z - z.max(1, keepdims=True)
    scores = (np.exp(z) / np.exp(z).sum(1, keepdims=True)).astype(np.float32)
    dec = B.oriented_delta_decode(np.asarray(proposals, np.float32)[:, :5], bbox_pred, means, stds).reshape(-1, 5).copy()
    dec[:, :4] /= scale_factor
    fg = scores[:, :-1]
    ri, ci = np.nonzero(fg > score_thresh)
    return B.obb2poly(dec[ri]), fg[ri, ci], ci


def hbb_overlaps(b1, b2, version=0, eps=1e-6):
    """models/boxes/iou_calculator.py:L302-343 (mode "iou", not aligned): `version` is the +1 px convention"""
    b1, b2 = np.asarray(b1, np.float32), np.asarray(b2, np.float32)
    v = np.float32(version)
    a1 = (b1[:, 2] - b1[:, 0] + v) * (b1[:, 3] - b1[:, 1] + v)
    a2 = (b2[:, 2] - b2[:, 0] + v) * (b2[:, 3] - b2[:, 1] + v)
    lt = np.maximum(b1[:, None, :2], b2[None, :, :2])
    rb = np.minimum(b1[:, None, 2:], b2[None, :, 2:])
    wh = np.clip(rb - lt + v, 0, None)
    overlap = wh[..., 0] * wh[..., 1]
    union = np.maximum(a1[:, None] + a2[None, :] - overlap, np.float32(eps))
    return (overlap / union).astype(np.float32)


def _sampled_all(boxes, gt_inds, is_gt):
    """RandomSampler when every candidate fits (sampler.py:L114-168): positives first, then negatives"""
    pos, neg = np.nonzero(gt_inds > 0)[0], np.nonzero(gt_inds == 0)[0]
    order = np.concatenate([pos, neg])
    return boxes[order], gt_inds[order], is_gt[order], len(pos)


def roitrans_rcnn_losses(proposals, gt_hbbs, gt_obbs, gt_labels, trunk1, trunk2, w_enlarge=1.2, h_enlarge=1.4,
                         stds1=(0.1, 0.1, 0.2, 0.2, 0.1), stds2=(0.05, 0.05, 0.1, 0.1, 0.05), num_classes=16):
    """models/networks/roi_transformer.py:L71-134 with rbbox_head.py (targets L9-121, loss L398-423, refinement
    L425-448), for samplers that keep every candidate.  proposals: per image (P, 4) horizontal boxes; labels 1-based
    (0 = background).  trunk1(rois (n,5) [img, hbb]) -> (cls (n,C), reg (n,5)); trunk2(rrois (n,6) [img, obb]) ->
    (cls (n,C), reg (n,5*C)).  Returns {"s0.rbbox_loss_cls", "s0.rbbox_loss_bbox", "s1...."}."""
    zeros5 = (0.,) * 5
    # ---- stage 1: horizontal proposals -> rotated RoIs
    rows = []
    for i, (p, gh, go, gl) in enumerate(zip(proposals, gt_hbbs, gt_obbs, gt_labels)):
        ov = hbb_overlaps(gh, p, version=1)                                    # BboxOverlaps2D_v1
        gt_inds, _, _ = B.assign_wrt_overlaps(ov, 0.5, 0.5, 0.5, match_low_quality=True, gt_labels=None)
        boxes = np.concatenate([gh.astype(np.float32), p.astype(np.float32)], 0)
        gt_inds = np.concatenate([np.arange(1, gh.shape[0] + 1), gt_inds.astype(np.int64)])
        is_gt = np.concatenate([np.ones(gh.shape[0], bool), np.zeros(p.shape[0], bool)])
        boxes, gt_inds, is_gt, npos = _sampled_all(boxes, gt_inds, is_gt)
        labels = np.zeros(boxes.shape[0], np.int64)
        labels[:npos] = gl[gt_inds[:npos] - 1]
        bt, bw = np.zeros((boxes.shape[0], 5), np.float32), np.zeros((boxes.shape[0], 5), np.float32)
        bt[:npos] = B.dbbox2delta_v3(B.hbb2obb_v2(boxes[:npos]), B.choose_best_obb_batch(go[gt_inds[:npos] - 1]),
                                     zeros5, stds1)
        bw[:npos] = 1
        rows.append((np.concatenate([np.full((boxes.shape[0], 1), i, np.float32), boxes], 1), labels, bt, bw, is_gt))
    rois, labels, bt, bw = [np.concatenate([r[k] for r in rows], 0) for k in range(4)]
    cls, reg = trunk1(rois)
    n = float(rois.shape[0])
    out = {"s0.rbbox_loss_cls": softmax_cross_entropy(cls, labels, np.ones(len(labels), np.float32), n),
           "s0.rbbox_loss_bbox": float(smooth_l1_loss(reg[labels > 0], bt[labels > 0], bw[labels > 0], 1.0,
                                                      avg_factor=n))}
    # ---- refinement: regress every sampled row, keep the ones that were not gts (L425-448)
    droi = np.concatenate([rois[:, :1], B.hbb2obb_v2(rois[:, 1:])], 1)
    refined = B.choose_best_Rroi_batch(B.delta2dbbox(droi[:, 1:], reg, zeros5, stds1, 1.0))
    # ---- stage 2: rotated RoIs -> detections
    rows2, start = [], 0
    for i, (r, go, gl) in enumerate(zip(rows, gt_obbs, gt_labels)):
        k = r[0].shape[0]
        cand = refined[start:start + k][~r[4]]
        start += k
        gbest = B.choose_best_Rroi_batch(go.astype(np.float32))
        ov = O.box_iou_rotated(gbest, cand, version=0)                         # BboxOverlaps2D_rotated
        gt_inds, _, _ = B.assign_wrt_overlaps(ov, 0.5, 0.5, 0.5, match_low_quality=True, gt_labels=None)
        boxes = np.concatenate([gbest, cand], 0)
        gt_inds = np.concatenate([np.arange(1, gbest.shape[0] + 1), gt_inds.astype(np.int64)])
        boxes, gt_inds, _, npos = _sampled_all(boxes, gt_inds, np.zeros(boxes.shape[0], bool))
        labels = np.zeros(boxes.shape[0], np.int64)
        labels[:npos] = gl[gt_inds[:npos] - 1]
        bt, bw = np.zeros((boxes.shape[0], 5), np.float32), np.zeros((boxes.shape[0], 5), np.float32)
        bt[:npos] = B.best_match_dbbox2delta(boxes[:npos], gbest[gt_inds[:npos] - 1], zeros5, stds2)
        bw[:npos] = 1
        rows2.append((np.concatenate([np.full((boxes.shape[0], 1), i, np.float32), boxes], 1), labels, bt, bw))
    rrois, labels, bt, bw = [np.concatenate([r[k] for r in rows2], 0) for k in range(4)]
    enl = rrois.copy()
    enl[:, 3] *= w_enlarge
    enl[:, 4] *= h_enlarge
    cls, reg = trunk2(enl)
    n = float(rrois.shape[0])
    pos = labels > 0
    pred = reg.reshape(reg.shape[0], num_classes, 5)[np.nonzero(pos)[0], labels[pos]]
    out["s1.rbbox_loss_cls"] = softmax_cross_entropy(cls, labels, np.ones(len(labels), np.float32), n)
    out["s1.rbbox_loss_bbox"] = float(smooth_l1_loss(pred, bt[pos], bw[pos], 1.0, avg_factor=n))
    return out


def l1_loss(pred, target, weight, avg_factor):
    """models/losses/l1_loss.py (reduction "mean" with avg_factor): sum(|pred - target| * weight) / avg_factor"""
    return float((np.abs(pred.astype(np.float64) - target.astype(np.float64)) * weight).sum() / avg_factor)


def retina_anchors(strides, sizes, octave_base_scale=4, scales_per_octave=3, ratios=(1.0, 0.5, 2.0)):
    """AnchorGeneratorRotatedRetinaNet (models/boxes/anchor_generator.py:L7-91): scales = base * 2^(i / per_octave)"""
    scales = [octave_base_scale * 2 ** (i / scales_per_octave) for i in range(scales_per_octave)]
    return [B.grid_anchors_s2anet(s, scales, list(ratios), sz, s) for s, sz in zip(strides, sizes)]


def retina_loss(cls_scores, bbox_preds, gts, labels, strides, cfg=None, gamma=2.0, alpha=0.25, **anchor_kw):
    """models/roi_heads/rotated_retina_head.py:L132-215 with the config's L1Loss: per-level loss lists.
    cls_scores: list over levels of (N, A*C, H, W); bbox_preds: (N, A*5, H, W)."""
    cfg = cfg or dict(pos_iou_thr=0.5, neg_iou_thr=0.4, min_pos_iou=0)
    N = cls_scores[0].shape[0]
    sizes = [tuple(t.shape[-2:]) for t in cls_scores]
    anchors = retina_anchors(strides, sizes, **anchor_kw)
    nla = [a.shape[0] for a in anchors]
    A = nla[0] // (sizes[0][0] * sizes[0][1])
    C = cls_scores[0].shape[1] // A
    lab, lw, bt, bw, num_pos = anchor_targets([np.concatenate(anchors, 0)] * N, gts, labels, cfg, nla)
    out = dict(loss_cls=[], loss_bbox=[])
    for l in range(len(sizes)):
        out["loss_cls"].append(sigmoid_focal_loss(_nhwc_rows(cls_scores[l], C), lab[l].reshape(-1), lw[l].reshape(-1),
                                                  gamma, alpha, num_pos))
        out["loss_bbox"].append(l1_loss(_nhwc_rows(bbox_preds[l], 5), bt[l].reshape(-1, 5), bw[l].reshape(-1, 5),
                                        num_pos))
    return out


def retina_get_bboxes_single(cls_scores, bbox_preds, strides, num_classes=15, **kw):
    """rotated_retina_head.py:L342-398 for one image (cls (A*C, H, W), box (A*5, H, W) per level)"""
    sizes = [tuple(t.shape[-2:]) for t in cls_scores]
    anchors = retina_anchors(strides, sizes)
    return s2anet_get_bboxes_single(cls_scores, bbox_preds, anchors, num_classes=num_classes, **kw)
