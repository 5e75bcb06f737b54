"""RepPoints minimum-area rectangle.  Drop-in for `jdet.ops.reppoints_min_area_bbox.reppoints_min_area_bbox`
(python/jdet/ops/reppoints_min_area_bbox/min_area_bbox.py:L22-34)."""
import torch

from jdet_amd import _lib as L


def reppoints_min_area_bbox(pointsets):
    """pointsets (N, 18) -> bboxes (N, 8): the four corners of the smallest rectangle around the hull of the 9 points"""
    assert pointsets.shape[1] == 18
    L.need_device(pointsets)
    ps = L.f32c(pointsets)
    out = torch.empty((ps.shape[0], 8), dtype=torch.float32, device=ps.device)
    L.check(L.lib().jdet_min_area_bbox(L.ptr(ps), ps.shape[0], L.ptr(out), L.stream_ptr(ps)), "jdet_min_area_bbox")
    return out.to(pointsets.dtype)
