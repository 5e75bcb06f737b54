"""RepPoints convex IoU.  Drop-in for `jdet.ops.reppoints_convex_iou.reppoints_convex_iou`
(python/jdet/ops/reppoints_convex_iou/convex_iou.py:L29-45): IoU between the convex hull of each 9-point set and each
quadrilateral, as the RepPoints assigner / `ConvexOverlaps` use it (models/boxes/iou_calculator.py:L3).
`reppoints_convex_giou` (convex_giou.py:L29-47): GIoU of ALIGNED pairs and its gradient w.r.t. the 9 points, as the
RepPoints losses use it (the reference returns the gradient from the op instead of defining a backward)."""
import torch

from jdet_amd import _lib as L


def reppoints_convex_iou(pointsets, gt_bboxes):
    """pointsets (N, 18), gt_bboxes (M, 8) -> ious (N, M)"""
    assert pointsets.dtype == gt_bboxes.dtype
    assert pointsets.dim() == 2 and pointsets.shape[1] == 18
    assert gt_bboxes.dim() == 2 and gt_bboxes.shape[1] == 8
    L.need_device(pointsets, gt_bboxes)
    ps, gt = L.f32c(pointsets), L.f32c(gt_bboxes)
    N, M = ps.shape[0], gt.shape[0]
    ious = torch.zeros((N, M), dtype=torch.float32, device=ps.device)     # (an empty side: zeros, L14)
    L.check(L.lib().jdet_convex_iou(L.ptr(ps), N, L.ptr(gt), M, L.ptr(ious), L.stream_ptr(ps)), "jdet_convex_iou")
    return ious.to(pointsets.dtype)


def reppoints_convex_giou(pointsets, gt_bboxes):
    """pointsets (N, 18), gt_bboxes (N, 8) -> (giou (N,), point_grad (N, 18)) -- convex_giou.py:L29-47"""
    assert pointsets.dtype == gt_bboxes.dtype
    assert pointsets.dim() == 2 and pointsets.shape[1] == 18
    assert gt_bboxes.dim() == 2 and gt_bboxes.shape[1] == 8
    assert gt_bboxes.shape[0] == pointsets.shape[0]
    L.need_device(pointsets, gt_bboxes)
    ps, gt = L.f32c(pointsets), L.f32c(gt_bboxes)
    N = ps.shape[0]
    out = torch.zeros((N, 19), dtype=torch.float32, device=ps.device)
    L.check(L.lib().jdet_convex_giou(L.ptr(ps), L.ptr(gt), N, L.ptr(out), L.stream_ptr(ps)), "jdet_convex_giou")
    out = out.to(pointsets.dtype)
    return out[:, -1], out[:, :-1]
