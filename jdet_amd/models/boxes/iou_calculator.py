"""IoU calculators resolved by name through the BOXES registry.  Mirrors
python/jdet/models/boxes/iou_calculator.py: `BboxOverlaps2D_rotated` L121-159 (-> box_iou_rotated),
`BboxOverlaps2D_rotated_v1` L161-198 (-> box_iou_rotated_v1), `bbox_overlaps_rotated` L228-233."""
from jdet_amd.ops import box_iou_rotated, box_iou_rotated_v1
from jdet_amd.utils.registry import BOXES


def bbox_overlaps_rotated(rboxes1, rboxes2, version=0):
    if version == 0:
        return box_iou_rotated(rboxes1.float(), rboxes2.float())
    return box_iou_rotated_v1(rboxes1.float(), rboxes2.float())


class _RotatedOverlaps:
    version = 0

    def __call__(self, bboxes1, bboxes2, mode="iou", is_aligned=False):
        assert bboxes1.size(-1) in [0, 5, 6]
        assert bboxes2.size(-1) in [0, 5, 6]
        if bboxes2.size(-1) == 6:
            bboxes2 = bboxes2[..., :5]
        if bboxes1.size(-1) == 6:
            bboxes1 = bboxes1[..., :5]
        assert mode == "iou" and is_aligned is False
        return bbox_overlaps_rotated(bboxes1, bboxes2, version=self.version)

    def __repr__(self):
        return self.__class__.__name__ + "()"


@BOXES.register_module()
class BboxOverlaps2D_rotated(_RotatedOverlaps):
    version = 0


@BOXES.register_module()
class BboxOverlaps2D_rotated_v1(_RotatedOverlaps):
    version = 1


def bbox_overlaps(bboxes1, bboxes2, mode="iou", is_aligned=False, eps=1e-6, version=0):
    """Axis-aligned overlaps, <x1,y1,x2,y2>; `version=1` is the legacy +1 pixel convention.
    Mirrors iou_calculator.py:L235-350 (elementwise tensor program -> torch)."""
    import torch
    assert mode in ["iou", "iof", "giou"], f"Unsupported mode {mode}"
    assert bboxes1.size(-1) == 4 or bboxes1.size(0) == 0
    assert bboxes2.size(-1) == 4 or bboxes2.size(0) == 0
    assert bboxes1.shape[:-2] == bboxes2.shape[:-2]
    batch_shape = tuple(bboxes1.shape[:-2])
    rows, cols = bboxes1.size(-2), bboxes2.size(-2)
    if is_aligned:
        assert rows == cols
    if rows * cols == 0:
        return bboxes1.new_zeros(batch_shape + ((rows,) if is_aligned else (rows, cols)))
    area1 = (bboxes1[..., 2] - bboxes1[..., 0] + version) * (bboxes1[..., 3] - bboxes1[..., 1] + version)
    area2 = (bboxes2[..., 2] - bboxes2[..., 0] + version) * (bboxes2[..., 3] - bboxes2[..., 1] + version)
    if is_aligned:
        lt = torch.maximum(bboxes1[..., :2], bboxes2[..., :2])
        rb = torch.minimum(bboxes1[..., 2:], bboxes2[..., 2:])
        wh = (rb - lt + version).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1 + area2 - overlap if mode in ["iou", "giou"] else area1
        if mode == "giou":
            enclosed_lt = torch.minimum(bboxes1[..., :2], bboxes2[..., :2])
            enclosed_rb = torch.maximum(bboxes1[..., 2:], bboxes2[..., 2:])
    else:
        lt = torch.maximum(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
        rb = torch.minimum(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
        wh = (rb - lt + version).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = area1[..., None] + area2[..., None, :] - overlap if mode in ["iou", "giou"] else area1[..., None]
        if mode == "giou":
            enclosed_lt = torch.minimum(bboxes1[..., :, None, :2], bboxes2[..., None, :, :2])
            enclosed_rb = torch.maximum(bboxes1[..., :, None, 2:], bboxes2[..., None, :, 2:])
    union = torch.clamp(union, min=eps)
    ious = overlap / union
    if mode in ["iou", "iof"]:
        return ious
    enclose_wh = (enclosed_rb - enclosed_lt).clamp(min=0)
    enclose_area = torch.clamp(enclose_wh[..., 0] * enclose_wh[..., 1], min=eps)
    return ious - (enclose_area - union) / enclose_area


def bbox_overlaps_fused(bboxes1, bboxes2, mode="iou", version=0, eps=1e-6, alive=None):
    """`bbox_overlaps` (not aligned, iou / iof) of (K,4) x (A,4+) device boxes as ONE launch (jdet_bbox_overlaps_hbb:
    the same operation order, bit-identical to the ~12-pass tensor program).  `alive` (A,) bool: columns of dead
    boxes come back as -1 (what the fixed-shape heads would otherwise write with a torch.where pass)."""
    import torch
    from jdet_amd import _lib as L
    b1, b2 = L.f32c(bboxes1), L.f32c(bboxes2)
    K, A = b1.shape[0], b2.shape[0]
    out = torch.empty((K, A), dtype=torch.float32, device=b1.device)
    al = alive.to(torch.uint8).contiguous() if alive is not None else None
    L.check(L.lib().jdet_bbox_overlaps_hbb(L.ptr(b1), K, L.ptr(b2), A, b2.shape[1], 1 if mode == "iof" else 0,
                                           int(version), float(eps), L.ptr(al), L.ptr(out), L.stream_ptr(b1)),
            "jdet_bbox_overlaps_hbb")
    return out


def _fusable(b1, b2, mode, is_aligned):
    import torch
    return (not is_aligned and mode in ("iou", "iof") and b1.dim() == 2 and b2.dim() == 2 and b1.is_cuda
            and b1.dtype == torch.float32 and b2.dtype == torch.float32 and b1.shape[0] > 0 and b2.shape[0] > 0
            and b1.shape[1] == 4 and not (torch.is_grad_enabled() and (b1.requires_grad or b2.requires_grad)))


class _HbbOverlaps:
    version = 0

    def __call__(self, bboxes1, bboxes2, mode="iou", is_aligned=False, version=None, alive=None):
        """`alive` (columns mask, fixed-shape heads only): dead columns are -1"""
        assert bboxes1.size(-1) in [0, 4, 5]
        assert bboxes2.size(-1) in [0, 4, 5]
        version = self.version if version is None else version
        if bboxes1.size(-1) == 5:
            bboxes1 = bboxes1[..., :4]
        if _fusable(bboxes1, bboxes2, mode, is_aligned):
            return bbox_overlaps_fused(bboxes1, bboxes2, mode, version, alive=alive)
        if bboxes2.size(-1) == 5:
            bboxes2 = bboxes2[..., :4]
        ious = bbox_overlaps(bboxes1, bboxes2, mode, is_aligned, version=version)
        if alive is not None:
            import torch
            ious = torch.where(alive[None, :], ious, torch.full_like(ious, -1.0))
        return ious

    def __repr__(self):
        return self.__class__.__name__ + "()"


@BOXES.register_module()
class BboxOverlaps2D(_HbbOverlaps):
    version = 0


@BOXES.register_module()
class BboxOverlaps2D_v1(_HbbOverlaps):
    version = 1
