"""Pairwise IoU of rotated boxes [xc,yc,w,h,theta(rad)].  Mirrors python/jdet/ops/box_iou_rotated.py:
L502-509 and box_iou_rotated_v1.py:L507-525 (the `_v1` flavour uses the Oriented R-CNN vertex
convention and zeroes rows/cols of boxes with min(w,h) < 1e-3).

`jdet_amd._lib.REFERENCE_SORT`: 0 reproduces the reference's CPU path (std::sort), 1 its CUDA exchange
sort; they differ only on degenerate hulls.  Default 0 = "the Jittor CPU reference" of BASELINE.json.
"""
import torch

from .. import _lib as L

__all__ = ["box_iou_rotated", "box_iou_rotated_v1"]


def _iou(boxes1, boxes2, version):
    assert boxes1.dtype == boxes2.dtype  # box_iou_rotated.py:L503
    L.need_device(boxes1, boxes2)
    b1, b2 = L.f32c(boxes1), L.f32c(boxes2)
    assert b1.dim() == 2 and b2.dim() == 2 and b1.shape[1] >= 5 and b1.shape[1] == b2.shape[1]
    n1, n2 = b1.shape[0], b2.shape[0]
    ious = torch.empty((n1, n2), dtype=torch.float32, device=b1.device)
    L.check(L.lib().jdet_box_iou_rotated(L.ptr(b1), n1, L.ptr(b2), n2, b1.shape[1], version, L.REFERENCE_SORT,
                                         L.ptr(ious), L.stream_ptr(b1)), "jdet_box_iou_rotated")
    return ious


def box_iou_rotated(boxes1, boxes2):
    return _iou(boxes1, boxes2, 0)


def box_iou_rotated_v1(boxes1, boxes2):
    ious = _iou(boxes1, boxes2, 1)
    # box_iou_rotated_v1.py:L515-523.  (The reference writes `.min(1)[0] < 0.001`; its intent --
    # per-box min(w,h) -- is what is implemented, without the `any_()` host sync.)
    small1 = boxes1[:, 2:4].min(dim=1).values < 0.001
    small2 = boxes2[:, 2:4].min(dim=1).values < 0.001
    ious = ious.masked_fill(small1[:, None], 0.0).masked_fill(small2[None, :], 0.0)
    return ious
