"""BBox coders of the named configs.  Mirrors python/jdet/models/boxes/coder.py:
`DeltaXYWHABBoxCoder` L76-141."""
import torch

from jdet_amd import _lib as L
from jdet_amd.utils.registry import BOXES

from .box_ops import bbox2delta_rotated, delta2bbox_rotated


def _fused_ok(*ts):
    """the fused HIP codecs produce values, not graphs: used whenever no gradient is asked for"""
    return all(t.is_cuda and t.dim() == 2 for t in ts) and not (torch.is_grad_enabled() and
                                                                  any(t.requires_grad for t in ts))


@BOXES.register_module()
class DeltaXYWHABBoxCoder:
    """encodes (x,y,w,h,a) into (dx,dy,dw,dh,da) relative to a base box and back."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.), clip_border=True):
        self.means = target_means
        self.stds = target_stds
        self.clip_border = clip_border

    def encode(self, bboxes, gt_bboxes):
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 5
        return bbox2delta_rotated(bboxes, gt_bboxes, self.means, self.stds)

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        assert pred_bboxes.size(0) == bboxes.size(0)
        return delta2bbox_rotated(bboxes, pred_bboxes, self.means, self.stds, max_shape, wh_ratio_clip,
                                  self.clip_border)


@BOXES.register_module()
class MidpointOffsetCoder:
    """Oriented RPN 6-parameter codec (coder.py:L322-437): hbb anchor -> (dx,dy,dw,dh,da,db) where a/b are
    the offsets of the top-most / right-most vertex from the mid-points of the enclosing box."""

    def __init__(self, target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.)):
        self.means = target_means
        self.stds = target_stds

    def encode(self, bboxes, gt_bboxes):
        from jdet_amd.ops.bbox_transforms import obb2hbb, obb2poly
        assert bboxes.size(0) == gt_bboxes.size(0)
        if _fused_ok(bboxes, gt_bboxes) and bboxes.shape[1] == 4 and gt_bboxes.shape[1] == 5:
            a, g = L.f32c(bboxes), L.f32c(gt_bboxes)
            out = torch.empty((a.shape[0], 6), dtype=torch.float32, device=a.device)
            L.check(L.lib().jdet_midpoint_offset_encode(L.ptr(a), L.ptr(g), a.shape[0], L.vecn(self.means, 6),
                                                        L.vecn(self.stds, 6), L.ptr(out), L.stream_ptr(a)),
                    "jdet_midpoint_offset_encode")
            return out
        pred_bboxes, gt = bboxes.float(), gt_bboxes.float()
        px = (pred_bboxes[..., 0] + pred_bboxes[..., 2]) * 0.5
        py = (pred_bboxes[..., 1] + pred_bboxes[..., 3]) * 0.5
        pw = pred_bboxes[..., 2] - pred_bboxes[..., 0]
        ph = pred_bboxes[..., 3] - pred_bboxes[..., 1]
        hbb, poly = obb2hbb(gt), obb2poly(gt)
        gx = (hbb[..., 0] + hbb[..., 2]) * 0.5
        gy = (hbb[..., 1] + hbb[..., 3]) * 0.5
        gw = hbb[..., 2] - hbb[..., 0]
        gh = hbb[..., 3] - hbb[..., 1]
        x_coor, y_coor = poly[:, 0::2], poly[:, 1::2]
        y_min = y_coor.min(dim=1, keepdim=True).values
        x_max = x_coor.max(dim=1, keepdim=True).values
        _x_coor = x_coor.clone()
        _x_coor[torch.abs(y_coor - y_min) > 0.1] = -1000
        ga = _x_coor.max(dim=1).values
        _y_coor = y_coor.clone()
        _y_coor[torch.abs(x_coor - x_max) > 0.1] = -1000
        gb = _y_coor.max(dim=1).values
        dx = (gx - px) / pw
        dy = (gy - py) / ph
        dw = torch.log(gw / pw)
        dh = torch.log(gh / ph)
        da = (ga - gx) / gw
        db = (gb - gy) / gh
        deltas = torch.stack([dx, dy, dw, dh, da, db], dim=-1)
        means = deltas.new_tensor(self.means).unsqueeze(0)
        stds = deltas.new_tensor(self.stds).unsqueeze(0)
        return (deltas - means) / stds

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        import math
        from jdet_amd.ops.bbox_transforms import rectpoly2obb
        assert pred_bboxes.size(0) == bboxes.size(0)
        if _fused_ok(bboxes, pred_bboxes) and bboxes.shape[1] == 4 and pred_bboxes.shape[1] == 6:
            a, d = L.f32c(bboxes), L.f32c(pred_bboxes)
            out = torch.empty((a.shape[0], 5), dtype=torch.float32, device=a.device)
            L.check(L.lib().jdet_midpoint_offset_decode(L.ptr(a), L.ptr(d), a.shape[0], L.vecn(self.means, 6),
                                                        L.vecn(self.stds, 6), float(wh_ratio_clip), L.ptr(out),
                                                        L.stream_ptr(a)), "jdet_midpoint_offset_decode")
            return out
        means = pred_bboxes.new_tensor(self.means).repeat(1, pred_bboxes.size(1) // 6)
        stds = pred_bboxes.new_tensor(self.stds).repeat(1, pred_bboxes.size(1) // 6)
        d = pred_bboxes * stds + means
        dx, dy, dw, dh, da, db = d[:, 0::6], d[:, 1::6], d[:, 2::6], d[:, 3::6], d[:, 4::6], d[:, 5::6]
        max_ratio = abs(math.log(wh_ratio_clip))
        dw = dw.clamp(min=-max_ratio, max=max_ratio)
        dh = dh.clamp(min=-max_ratio, max=max_ratio)
        px = ((bboxes[:, 0] + bboxes[:, 2]) * 0.5).unsqueeze(1)
        py = ((bboxes[:, 1] + bboxes[:, 3]) * 0.5).unsqueeze(1)
        pw = (bboxes[:, 2] - bboxes[:, 0]).unsqueeze(1)
        ph = (bboxes[:, 3] - bboxes[:, 1]).unsqueeze(1)
        gw, gh = pw * dw.exp(), ph * dh.exp()
        gx, gy = px + pw * dx, py + ph * dy
        x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
        da = da.clamp(min=-0.5, max=0.5)
        db = db.clamp(min=-0.5, max=0.5)
        ga, _ga = gx + da * gw, gx - da * gw
        gb, _gb = gy + db * gh, gy - db * gh
        polys = torch.stack([ga, y1, x2, gb, _ga, y2, x1, _gb], dim=-1)
        center = torch.stack([gx, gy, gx, gy, gx, gy, gx, gy], dim=-1)
        center_polys = polys - center
        diag_len = torch.sqrt(center_polys[..., 0::2] ** 2 + center_polys[..., 1::2] ** 2)
        max_diag_len = diag_len.max(dim=-1, keepdim=True).values
        diag_scale_factor = max_diag_len / diag_len
        center_polys = center_polys * diag_scale_factor.repeat_interleave(2, dim=-1)
        rectpolys = center_polys + center
        return rectpoly2obb(rectpolys).flatten(-2)


@BOXES.register_module()
class OrientedDeltaXYWHTCoder:
    """Oriented R-CNN 5-parameter codec (coder.py:L439-518): picks between dtheta and dtheta + pi/2 (with a
    w/h swap) by smaller magnitude, using arithmetic masks."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.)):
        self.means = target_means
        self.stds = target_stds

    def encode(self, bboxes, gt_bboxes):
        import math
        from jdet_amd.ops.bbox_transforms import regular_theta
        assert bboxes.size(0) == gt_bboxes.size(0)
        assert bboxes.size(-1) == gt_bboxes.size(-1) == 5
        if _fused_ok(bboxes, gt_bboxes):
            p, g = L.f32c(bboxes), L.f32c(gt_bboxes)
            out = torch.empty_like(p)
            L.check(L.lib().jdet_oriented_delta_encode(L.ptr(p), L.ptr(g), p.shape[0], L.vecn(self.means, 5),
                                                       L.vecn(self.stds, 5), L.ptr(out), L.stream_ptr(p)),
                    "jdet_oriented_delta_encode")
            return out
        px, py, pw, ph, ptheta = bboxes.float().unbind(dim=-1)
        gx, gy, gw, gh, gtheta = gt_bboxes.float().unbind(dim=-1)
        dtheta1 = regular_theta(gtheta - ptheta)
        dtheta2 = regular_theta(gtheta - ptheta + math.pi / 2)
        m = (torch.abs(dtheta1) < torch.abs(dtheta2)).to(px.dtype)
        gw_regular = gw * m + gh * (1 - m)
        gh_regular = gh * m + gw * (1 - m)
        dtheta = dtheta1 * m + dtheta2 * (1 - m)
        dx = (torch.cos(-ptheta) * (gx - px) + torch.sin(-ptheta) * (gy - py)) / pw
        dy = (-torch.sin(-ptheta) * (gx - px) + torch.cos(-ptheta) * (gy - py)) / ph
        dw = torch.log(gw_regular / pw)
        dh = torch.log(gh_regular / ph)
        deltas = torch.stack([dx, dy, dw, dh, dtheta], dim=-1)
        means = deltas.new_tensor(self.means).unsqueeze(0)
        stds = deltas.new_tensor(self.stds).unsqueeze(0)
        return (deltas - means) / stds

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        import math
        from jdet_amd.ops.bbox_transforms import regular_obb, regular_theta
        assert pred_bboxes.size(0) == bboxes.size(0)
        if _fused_ok(bboxes, pred_bboxes) and bboxes.shape[1] == 5 and pred_bboxes.shape[1] % 5 == 0:
            r, d = L.f32c(bboxes), L.f32c(pred_bboxes)
            out = torch.empty_like(d)
            L.check(L.lib().jdet_oriented_delta_decode(L.ptr(r), L.ptr(d), d.shape[0], d.shape[1] // 5,
                                                       L.vecn(self.means, 5), L.vecn(self.stds, 5),
                                                       float(wh_ratio_clip), L.ptr(out), L.stream_ptr(d)),
                    "jdet_oriented_delta_decode")
            return out
        means = pred_bboxes.new_tensor(self.means).repeat(1, pred_bboxes.size(1) // 5)
        stds = pred_bboxes.new_tensor(self.stds).repeat(1, pred_bboxes.size(1) // 5)
        d = pred_bboxes * stds + means
        dx, dy, dw, dh, dtheta = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
        max_ratio = abs(math.log(wh_ratio_clip))
        dw = dw.clamp(min=-max_ratio, max=max_ratio)
        dh = dh.clamp(min=-max_ratio, max=max_ratio)
        px, py, pw, ph, ptheta = bboxes.unbind(dim=-1)
        px, py = px.unsqueeze(1).expand_as(dx), py.unsqueeze(1).expand_as(dy)
        pw, ph = pw.unsqueeze(1).expand_as(dw), ph.unsqueeze(1).expand_as(dh)
        ptheta = ptheta.unsqueeze(1).expand_as(dtheta)
        gx = dx * pw * torch.cos(-ptheta) - dy * ph * torch.sin(-ptheta) + px
        gy = dx * pw * torch.sin(-ptheta) + dy * ph * torch.cos(-ptheta) + py
        gw, gh = pw * dw.exp(), ph * dh.exp()
        gtheta = regular_theta(dtheta + ptheta)
        new_bboxes = regular_obb(torch.stack([gx, gy, gw, gh, gtheta], dim=-1))
        return new_bboxes.view_as(pred_bboxes)
