"""Rotated NMS.  Mirrors python/jdet/ops/nms_rotated.py: `nms_rotated` (L527-538), `ml_nms_rotated`
(L515-525), `multiclass_nms_rotated` (L540-596).

The suppression rule follows the reference's CPU path by default (`iou >= thr`, L444); set
`REFERENCE_RULE = "cuda"` for its CUDA rule (`iou > thr`, L403).  Everything up to the final
index extraction runs on the device without a host sync (the reference synchronises and scans the
bitmask on the host, L475-491); the returned index tensor has a data-dependent shape, so
`nonzero` is the one unavoidable sync -- exactly where `jt.where(keep)[0]` has it.
"""
import torch

from .. import _lib as L

__all__ = ["nms_rotated", "ml_nms_rotated", "multiclass_nms_rotated", "nms_rotated_keep_mask"]

REFERENCE_RULE = "cpu"


def nms_rotated_keep_mask(dets, order, iou_threshold, rule=None, n_labels=1):
    """dets (n,5|6) fp32, order (n,) visiting order (descending score; for 6-column dets any order that is
    descending in score inside each label) -> bool keep mask over original indices.  Device-only, fixed shapes.
    `rule`: "cpu" (suppress at iou >= thr) | "cuda" (iou > thr); None = the module-level REFERENCE_RULE.
    `n_labels` > 1: column 5 holds integer labels 0 .. n_labels-1 and `order` is sorted by label -- one scan
    workgroup per label."""
    L.need_device(dets, order)
    d = L.f32c(dets)
    n, bl = d.shape
    o = order.to(torch.int32).contiguous()
    keep = (torch.zeros if n_labels > 1 else torch.empty)((n,), dtype=torch.uint8, device=d.device)
    ws_bytes = L.lib().jdet_nms_rotated_workspace(n)
    ws = torch.empty((max(ws_bytes, 8),), dtype=torch.uint8, device=d.device)
    L.check(L.lib().jdet_nms_labeled(L.ptr(d), n, bl, L.ptr(o), float(iou_threshold),
                                     1 if (rule or REFERENCE_RULE) == "cpu" else 0, L.REFERENCE_SORT, 0,
                                     int(n_labels) if bl == 6 else 1, L.ptr(keep), L.ptr(ws), ws_bytes,
                                     L.stream_ptr(d)), "jdet_nms_labeled")
    return keep.bool()


def _order(scores):
    # jt.argsort(descending) tie order is a Jittor internal ("parity unpinned"): a stable
    # descending sort (lowest index first among equal scores) is the documented choice here.
    return torch.argsort(scores, dim=0, descending=True, stable=True)


def ml_nms_rotated(dets, scores, labels, iou_threshold, num_classes=None):
    assert dets.numel() > 0 and dets.dim() == 2
    assert dets.dtype == scores.dtype
    dets6 = torch.cat([dets, labels.to(dets.dtype).unsqueeze(1)], dim=1)
    # boxes of different labels never suppress each other (nms_rotated.py:L283-286), so visiting them class by
    # class (descending score inside a class) keeps exactly the same set as the reference's global score order,
    # and makes the 64x64 tiles label-homogeneous: the kernel skips every tile whose label ranges are disjoint
    order = _order(scores)
    order = order[torch.argsort(labels[order], stable=True)]
    # num_classes given (labels are 0 .. num_classes-1): every class is scanned by its own workgroup
    keep = nms_rotated_keep_mask(dets6, order, iou_threshold, n_labels=num_classes or 1)
    return torch.nonzero(keep)[:, 0]


def nms_rotated(dets, scores, iou_threshold):
    if dets.numel() == 0:
        return torch.zeros((0,), dtype=torch.long, device=dets.device)  # reference: jt.array([])
    assert dets.dim() == 2
    assert dets.dtype == scores.dtype
    keep = nms_rotated_keep_mask(dets, _order(scores), iou_threshold)
    return torch.nonzero(keep)[:, 0]


def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, nms_cfg, max_num=-1, score_factors=None):
    """(n, #class*5 | 5), (n, #class+1 with background in column 0) -> ((k,6) [box,score], (k,) labels 0-based)."""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 5)[:, 1:]
    else:
        bboxes = multi_bboxes[:, None].expand(multi_bboxes.shape[0], num_classes, 5)
    scores = multi_scores[:, 1:]
    valid_mask = scores > score_thr
    bboxes = bboxes[valid_mask]
    if score_factors is not None:
        scores = scores * score_factors[:, None]
    scores = scores[valid_mask]
    labels = valid_mask.nonzero()[:, 1]
    if bboxes.numel() == 0:
        return (torch.zeros((0, 6), device=multi_bboxes.device),
                torch.zeros((0,), dtype=torch.int32, device=multi_bboxes.device))
    nms_cfg_ = dict(nms_cfg)
    nms_cfg_.pop("type", "nms")
    iou_thr = nms_cfg_.pop("iou_thr", 0.1)
    keep = ml_nms_rotated(bboxes, scores, labels, iou_thr, num_classes=num_classes)
    bboxes, scores, labels = bboxes[keep], scores[keep], labels[keep]
    inds = _order(scores)
    if keep.size(0) > max_num:
        inds = inds[:max_num]  # literal reference behaviour (max_num=-1 drops the last one, L588-594)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    return torch.cat([bboxes, scores[:, None]], 1), labels
