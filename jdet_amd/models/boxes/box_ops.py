"""Rotated box algebra used by the named configs.  Mirrors python/jdet/models/boxes/box_ops.py:
norm_angle L176-178, bbox2delta_rotated L180-226, delta2bbox_rotated L229-285,
rotated_box_to_poly L592-613.

On a HIP device and outside autograd the two coders run as ONE fused kernel each
(csrc/box_codec_assign.hip); the torch expressions below are the differentiable form (needed when
a loss is applied to decoded boxes) and define the same arithmetic.
"""
import math

import torch

from jdet_amd import _lib as L


def norm_angle(angle, range=(float(-math.pi / 4), float(math.pi))):
    # Python-style floor mod (Jittor's float % is unpinned; floor-mod is what makes the
    # documented [-pi/4, 3pi/4) range come out)
    return torch.remainder(angle - range[0], range[1]) + range[0]


def _fused_ok(*ts):
    return all(t.is_cuda for t in ts) and not (torch.is_grad_enabled() and any(t.requires_grad for t in ts))


def _vec5(v):
    import ctypes
    return (ctypes.c_float * 5)(*[float(x) for x in v])


def bbox2delta_rotated(proposals, gt, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.)):
    assert proposals.size() == gt.size()
    if _fused_ok(proposals, gt) and proposals.dim() == 2:
        p, g = L.f32c(proposals), L.f32c(gt)
        out = torch.empty_like(p)
        L.check(L.lib().jdet_bbox2delta_rotated(L.ptr(p), L.ptr(g), p.shape[0], _vec5(means), _vec5(stds),
                                                L.ptr(out), L.stream_ptr(p)), "jdet_bbox2delta_rotated")
        return out
    gt_widths, gt_heights, gt_angle = gt[..., 2], gt[..., 3], gt[..., 4]
    pw, ph, pa = proposals[..., 2], proposals[..., 3], proposals[..., 4]
    cosa, sina = torch.cos(pa), torch.sin(pa)
    coord = gt[..., 0:2] - proposals[..., 0:2]
    dx = (cosa * coord[..., 0] + sina * coord[..., 1]) / pw
    dy = (-sina * coord[..., 0] + cosa * coord[..., 1]) / ph
    dw = torch.log(torch.clamp(gt_widths / pw, 1e-30, 1e30))   # jt.safe_log
    dh = torch.log(torch.clamp(gt_heights / ph, 1e-30, 1e30))
    da = norm_angle(gt_angle - pa) / math.pi
    deltas = torch.stack((dx, dy, dw, dh, da), -1)
    means = deltas.new_tensor(means).unsqueeze(0)
    stds = deltas.new_tensor(stds).unsqueeze(0)
    return (deltas - means) / stds


def delta2bbox_rotated(rois, deltas, means=(0., 0., 0., 0., 0.), stds=(1., 1., 1., 1., 1.), max_shape=None,
                       wh_ratio_clip=16 / 1000, clip_border=True):
    """rois (N,5), deltas (N, 5*num_classes) -> (N, 5*num_classes).  `max_shape` / `clip_border` are
    accepted but never applied, exactly as in the reference (box_ops.py:L229-285)."""
    if _fused_ok(rois, deltas) and deltas.dim() == 2 and deltas.shape[1] % 5 == 0:
        r, d = L.f32c(rois), L.f32c(deltas)
        out = torch.empty_like(d)
        L.check(L.lib().jdet_delta2bbox_rotated(L.ptr(r), L.ptr(d), d.shape[0], d.shape[1] // 5, _vec5(means),
                                                _vec5(stds), float(wh_ratio_clip), L.ptr(out),
                                                L.stream_ptr(d)), "jdet_delta2bbox_rotated")
        return out
    means = deltas.new_tensor(means).repeat(1, deltas.size(1) // 5)
    stds = deltas.new_tensor(stds).repeat(1, deltas.size(1) // 5)
    denorm = deltas * stds + means
    dx, dy, dw, dh, dangle = denorm[:, 0::5], denorm[:, 1::5], denorm[:, 2::5], denorm[:, 3::5], denorm[:, 4::5]
    max_ratio = abs(math.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    roi_x = rois[:, 0].unsqueeze(1).expand_as(dx)
    roi_y = rois[:, 1].unsqueeze(1).expand_as(dy)
    roi_w = rois[:, 2].unsqueeze(1).expand_as(dw)
    roi_h = rois[:, 3].unsqueeze(1).expand_as(dh)
    roi_angle = rois[:, 4].unsqueeze(1).expand_as(dangle)
    gx = dx * roi_w * torch.cos(roi_angle) - dy * roi_h * torch.sin(roi_angle) + roi_x
    gy = dx * roi_w * torch.sin(roi_angle) + dy * roi_h * torch.cos(roi_angle) + roi_y
    gw = roi_w * dw.exp()
    gh = roi_h * dh.exp()
    ga = norm_angle(math.pi * dangle + roi_angle)
    return torch.stack([gx, gy, gw, gh, ga], dim=-1).view_as(deltas)


def rotated_box_to_poly(rrects):
    """(n,5) [xc,yc,w,h,theta] -> (n,8) [x0,y0,...,x3,y3], corners tl, tr, br, bl of the unrotated
    box rotated by theta: x' = c*x - s*y + xc, y' = s*x + c*y + yc (box_ops.py:L592-613)."""
    n = rrects.shape[0]
    if n == 0:
        return rrects.new_zeros((0, 8))
    x_ctr, y_ctr, width, height, angle = rrects[:, 0], rrects[:, 1], rrects[:, 2], rrects[:, 3], rrects[:, 4]
    tl_x, tl_y, br_x, br_y = -width / 2, -height / 2, width / 2, height / 2
    xs = torch.stack([tl_x, br_x, br_x, tl_x], 1)
    ys = torch.stack([tl_y, tl_y, br_y, br_y], 1)
    c, s_ = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    px = c * xs - s_ * ys + x_ctr[:, None]
    py = s_ * xs + c * ys + y_ctr[:, None]
    return torch.stack([px, py], dim=2).reshape(n, 8)
