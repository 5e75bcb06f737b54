"""Polygon IoU / polygon NMS.  Mirrors python/jdet/ops/nms_poly.py: `poly_nms` (L187-232), `multiclass_poly_nms`
(L234-245), `iou_poly` (L247-252); the kernels are csrc/poly_iou.hip (4-point polygons, any orientation, convex or
not).  `poly_iou_matrix` is the batched form the evaluation / merging code here uses instead of the reference's
per-pair Python loops.  Everything up to the final index extraction stays on the device (the reference synchronises
and scans the bit mask on the host, L207-229).
"""
import numpy as np
import torch

from .. import _lib as L

__all__ = ["poly_iou_matrix", "iou_poly", "poly_nms", "poly_nms_keep_mask", "multiclass_poly_nms"]


def poly_iou_matrix(polys1, polys2, mode=1):
    """(n1, 8+) x (n2, 8+) device tensors -> (n1, n2) IoU.  mode 1: `iou_poly`'s rule (inter / max(union, 0.01));
    mode 0: the NMS kernel's (a zero union counts as IoU 1)."""
    L.need_device(polys1, polys2)
    a, b = L.f32c(polys1), L.f32c(polys2)
    assert a.dim() == 2 and b.dim() == 2 and a.shape[1] >= 8 and b.shape[1] >= 8
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    L.check(L.lib().jdet_poly_iou(L.ptr(a), a.shape[0], a.shape[1], L.ptr(b), b.shape[0], b.shape[1], int(mode),
                                  L.ptr(out), L.stream_ptr(a)), "jdet_poly_iou")
    return out


def iou_poly(poly1, poly2, device=None):
    """two 8-vectors (numpy / tensors) -> python float, the reference's per-pair call (shapely there)"""
    dev = torch.device("cuda") if device is None else torch.device(device)
    a = torch.as_tensor(np.asarray(poly1, np.float32).reshape(1, 8)).to(dev)
    b = torch.as_tensor(np.asarray(poly2, np.float32).reshape(1, 8)).to(dev)
    return float(poly_iou_matrix(a, b, 1)[0, 0])


def poly_nms_keep_mask(polys, order, thresh, n_labels=1):
    """polys (n, 8) or (n, 9) with an integer label in column 8; order: visiting order (descending score; label by
    label when n_labels > 1) -> bool keep mask over original indices; device only, fixed shapes"""
    L.need_device(polys, order)
    p = L.f32c(polys)
    n, rl = p.shape
    assert rl in (8, 9)
    o = order.to(torch.int32).contiguous()
    keep = (torch.zeros if n_labels > 1 else torch.empty)((n,), dtype=torch.uint8, device=p.device)
    wsb = L.lib().jdet_nms_rotated_workspace(n)
    ws = torch.empty((max(wsb, 8),), dtype=torch.uint8, device=p.device)
    L.check(L.lib().jdet_nms_poly(L.ptr(p), n, rl, L.ptr(o), float(thresh), int(n_labels) if rl == 9 else 1,
                                  L.ptr(keep), L.ptr(ws), wsb, L.stream_ptr(p)), "jdet_nms_poly")
    return keep.bool()


def poly_nms(boxes, nms_overlap_thresh):
    """boxes (n, 9) [8 coordinates, score] -> kept indices in descending-score order (L187-232)"""
    assert boxes.dim() == 2 and boxes.shape[1] == 9
    if boxes.shape[0] == 0:
        return torch.zeros((0,), dtype=torch.long, device=boxes.device)
    order = torch.argsort(boxes[:, 8], descending=True, stable=True)
    keep = poly_nms_keep_mask(boxes[:, :8], order, nms_overlap_thresh)
    return order[keep[order]]


def multiclass_poly_nms(bboxes, scores, labels, thresh):
    """(n, 8), (n,), (n,) -> (dets (k, 9), labels (k,)) in descending-score order.  The reference separates the
    classes by adding label * (coordinate range + 1) to the polygons (L235-237); here the label rides along as a
    ninth column and the kernel skips cross-label pairs -- same keep set without the fp32 cost of large offsets."""
    if bboxes.shape[0] == 0:
        return torch.zeros((0, 9), device=bboxes.device), labels[:0]
    order = torch.argsort(scores, descending=True, stable=True)
    visit = order[torch.argsort(labels[order], stable=True)]
    polys9 = torch.cat([bboxes[:, :8].float(), labels.to(torch.float32)[:, None]], 1)
    keep = poly_nms_keep_mask(polys9, visit, thresh)
    sel = order[keep[order]]
    return torch.cat([bboxes[sel], scores[sel, None]], dim=1), labels[sel]
