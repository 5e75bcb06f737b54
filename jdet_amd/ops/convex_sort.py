"""Graham scan of masked point sets.  Drop-in for `jdet.ops.convex_sort.convex_sort` (python/jdet/ops/convex_sort.py:
L196-201), the hull step of the polygon IoU loss (models/losses/poly_iou_loss.py:L19-27).  One kernel does what the
reference spreads over tensor ops + a scan kernel: start point (first lowest unmasked), angular order, scan."""
import torch

from jdet_amd import _lib as L


def convex_sort(pts, masks, circular=True):
    """pts (nbs, npts, 2), masks (nbs, npts) bool / 0-1 -> (nbs, npts + circular) int32 hull indices, -1 padded"""
    assert pts.size(0) == masks.size(0) and pts.size(1) == masks.size(1)
    L.need_device(pts, masks)
    p, m = L.f32c(pts), L.f32c(masks)
    nbs, npts = p.shape[0], p.shape[1]
    out = torch.full((nbs, npts + (1 if circular else 0)), -1, dtype=torch.int32, device=p.device)
    L.check(L.lib().jdet_convex_sort(L.ptr(p), L.ptr(m), nbs, npts, int(bool(circular)), L.ptr(out), L.stream_ptr(p)),
            "jdet_convex_sort")
    return out
