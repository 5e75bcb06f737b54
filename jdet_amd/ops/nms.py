"""Horizontal NMS.  Mirrors python/jdet/ops/nms.py:L4-9 (`nms(boxes, scores, thresh)` -> `jt.nms`).

`jt.nms` is a Jittor builtin with no source in the reference tree: **parity unpinned** (SURVEY 8c).
Documented assumption: standard greedy NMS, suppress when IoU > thresh, no "+1" pixel convention,
kept indices returned in descending-score order.  It runs on the same device path as rotated NMS
(tile bitmask kernel + on-device scan) with the rectangle overlap formula in the tile kernel.
"""
import torch

from .. import _lib as L


def nms_keep_mask(boxes, scores, thresh, labels=None, n_labels=None, visit_order=None):
    """greedy horizontal NMS -> bool keep mask over the input order; device-only, fixed shapes (no host sync).
    `labels` (optional, e.g. FPN level ids): boxes with different labels never suppress each other -- the effect of
    the reference's "add level_id * (max_coordinate + 1) to the boxes" trick (oriented_rpn_head.py:L214-219), obtained
    by skipping the cross-label 64x64 tiles instead of computing their zero IoUs.  `n_labels`: the labels are the
    integers 0 .. n_labels-1 (every label gets its own scan workgroup); None: any labels, one scan.
    `visit_order`: the caller's visiting order (label by label, descending score inside a label) when the boxes
    already come that way -- saves the two device sorts; `order` is then returned as given.  Returns (keep, order)
    with `order` = indices by descending score (stable)."""
    assert boxes.shape[-1] == 4 and len(scores) == len(boxes)
    if scores.dim() == 2:
        scores = scores[:, 0]
    n = boxes.shape[0]
    L.need_device(boxes, scores)
    b = boxes.float()
    cols = [(b[:, 0] + b[:, 2]) * 0.5, (b[:, 1] + b[:, 3]) * 0.5, b[:, 2] - b[:, 0], b[:, 3] - b[:, 1],
            torch.zeros_like(b[:, 0])]
    if labels is not None:
        cols.append(labels.to(b.dtype))
    if visit_order is not None:
        order = visit = visit_order
    else:
        order = torch.argsort(scores.float(), descending=True, stable=True)
        visit = order
        if labels is not None:
            visit = order[torch.argsort(labels[order], stable=True)]   # class by class, descending score inside
    obb = torch.stack(cols, dim=1).contiguous()
    o32 = visit.to(torch.int32).contiguous()
    # one scan workgroup per label writes the flags of ITS boxes: a label outside 0 .. n_labels-1 is never visited and
    # stays "suppressed" (zeros) instead of uninitialised
    keep = torch.zeros((n,), dtype=torch.uint8, device=b.device)
    wsb = L.lib().jdet_nms_rotated_workspace(n)
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=b.device)
    L.check(L.lib().jdet_nms_labeled(L.ptr(obb), n, obb.shape[1], L.ptr(o32), float(thresh), 0, 0, 1,
                                     int(n_labels) if (labels is not None and n_labels) else 1, L.ptr(keep),
                                     L.ptr(ws), wsb, L.stream_ptr(b)), "jdet_nms_labeled (horizontal)")
    return keep.bool(), order


def nms(boxes, scores, thresh, labels=None):
    """kept indices in descending-score order (data-dependent length: one host sync, as `jt.nms`)"""
    if boxes.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device)
    keep, order = nms_keep_mask(boxes, scores, thresh, labels)
    return order[keep[order]]


def nms_dets(dets, thresh, labels=None):
    """`jt.nms(dets (n,5) [x1,y1,x2,y2,score], thresh)`"""
    return nms(dets[:, :4], dets[:, 4], thresh, labels)
