"""Tensor box algebra of the named configs.  Mirrors python/jdet/ops/bbox_transforms.py:
regular_theta / regular_obb L499-517, get_bbox_type / get_bbox_dim L519-545, rectpoly2obb L575-597,
poly2hbb L600-607, obb2poly L610-637, obb2hbb L640-646, hbb2poly L649-651, hbb2obb L654-666,
bbox2type L677-687, get_bbox_areas L689-702.  (cv2-based poly2obb / mask helpers are out of scope.)

Oriented R-CNN negates gt angles on entry, hence obb2poly's (+w/2 cos, -w/2 sin) convention here.
"""
import math

import torch

from jdet_amd.utils.general import const_like


def regular_theta(theta, mode="180", start=-math.pi / 2):
    assert mode in ["360", "180"]
    cycle = 2 * math.pi if mode == "360" else math.pi
    return torch.remainder(theta - start, cycle) + start   # floor-mod (Jittor % on floats: unpinned)


def regular_obb(obboxes):
    x, y, w, h, theta = obboxes.unbind(dim=-1)
    m = (w > h).to(w.dtype)                # arithmetic masks, as the reference
    w_regular = w * m + h * (1 - m)
    h_regular = h * m + w * (1 - m)
    theta_regular = theta * m + (theta + math.pi / 2) * (1 - m)
    theta_regular = regular_theta(theta_regular)
    return torch.stack([x, y, w_regular, h_regular, theta_regular], dim=-1)


def get_bbox_type(bboxes, with_score=False):
    dim = bboxes.size(-1)
    if with_score:
        dim -= 1
    return {4: "hbb", 5: "obb", 8: "poly"}.get(dim, "notype")


def get_bbox_dim(bbox_type, with_score=False):
    if bbox_type not in ("hbb", "obb", "poly"):
        raise ValueError(f"don't know {bbox_type} bbox dim")
    return {"hbb": 4, "obb": 5, "poly": 8}[bbox_type] + (1 if with_score else 0)


def rectpoly2obb(polys):
    theta = torch.atan2(-(polys[..., 3] - polys[..., 1]), polys[..., 2] - polys[..., 0])
    Cos, Sin = torch.cos(theta), torch.sin(theta)
    Matrix = torch.stack([Cos, -Sin, Sin, Cos], dim=-1)
    Matrix = Matrix.view(*Matrix.shape[:-1], 2, 2)
    x = polys[..., 0::2].mean(-1)
    y = polys[..., 1::2].mean(-1)
    center = torch.stack([x, y], dim=-1).unsqueeze(-2)
    center_polys = polys.reshape(*polys.shape[:-1], 4, 2) - center
    rotate_polys = torch.matmul(center_polys, Matrix.transpose(-1, -2))
    xmin, xmax = rotate_polys[..., :, 0].min(dim=-1).values, rotate_polys[..., :, 0].max(dim=-1).values
    ymin, ymax = rotate_polys[..., :, 1].min(dim=-1).values, rotate_polys[..., :, 1].max(dim=-1).values
    return regular_obb(torch.stack([x, y, xmax - xmin, ymax - ymin, theta], dim=-1))


def poly2hbb(polys):
    polys = polys.view(*polys.shape[:-1], polys.size(-1) // 2, 2)
    return torch.cat([polys.min(dim=-2).values, polys.max(dim=-2).values], dim=-1)


def obb2poly(obboxes):
    center, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=-1)
    Cos, Sin = torch.cos(theta), torch.sin(theta)
    vector1 = torch.cat([w / 2 * Cos, -w / 2 * Sin], dim=-1)
    vector2 = torch.cat([-h / 2 * Sin, -h / 2 * Cos], dim=-1)
    return torch.cat([center + vector1 + vector2, center + vector1 - vector2, center - vector1 - vector2,
                      center - vector1 + vector2], dim=-1)


def obb2hbb(obboxes):
    center, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=-1)
    Cos, Sin = torch.cos(theta), torch.sin(theta)
    x_bias = torch.abs(w / 2 * Cos) + torch.abs(h / 2 * Sin)
    y_bias = torch.abs(w / 2 * Sin) + torch.abs(h / 2 * Cos)
    bias = torch.cat([x_bias, y_bias], dim=-1)
    return torch.cat([center - bias, center + bias], dim=-1)


def hbb2poly(hbboxes):
    l, t, r, b = hbboxes.unbind(-1)
    return torch.stack([l, t, r, t, r, b, l, b], dim=-1)


def hbb2obb(hbboxes):
    x = (hbboxes[..., 0] + hbboxes[..., 2]) * 0.5
    y = (hbboxes[..., 1] + hbboxes[..., 3]) * 0.5
    w = hbboxes[..., 2] - hbboxes[..., 0]
    h = hbboxes[..., 3] - hbboxes[..., 1]
    theta = torch.zeros_like(x)
    obboxes1 = torch.stack([x, y, w, h, theta], dim=-1)
    obboxes2 = torch.stack([x, y, h, w, theta - math.pi / 2], dim=-1)
    flag = (w >= h)[..., None].to(x.dtype)
    return flag * obboxes1 + (1 - flag) * obboxes2


_type_func_map = {("poly", "hbb"): poly2hbb, ("obb", "poly"): obb2poly, ("obb", "hbb"): obb2hbb,
                  ("hbb", "poly"): hbb2poly, ("hbb", "obb"): hbb2obb}


def bbox2type(bboxes, to_type):
    assert to_type in ["hbb", "obb", "poly"]
    ori_type = get_bbox_type(bboxes)
    if ori_type == "notype":
        raise ValueError("Not a bbox type")
    if ori_type == to_type:
        return bboxes
    if (ori_type, to_type) not in _type_func_map:
        raise NotImplementedError("poly -> obb needs cv2.minAreaRect in the reference (out of scope)")
    return _type_func_map[(ori_type, to_type)](bboxes)


def get_bbox_areas(bboxes):
    btype = get_bbox_type(bboxes)
    if btype == "hbb":
        wh = bboxes[..., 2:] - bboxes[..., :2]
        return wh[..., 0] * wh[..., 1]
    if btype == "obb":
        return bboxes[..., 2] * bboxes[..., 3]
    if btype == "poly":
        pts = bboxes.view(*bboxes.size()[:-1], 4, 2)
        roll_pts = torch.roll(pts, 1, dims=-2)
        xyxy = torch.sum(pts[..., 0] * roll_pts[..., 1] - roll_pts[..., 0] * pts[..., 1], dim=-1)
        return 0.5 * torch.abs(xyxy)
    raise ValueError("The type of bboxes is notype")


# ------------------------------------------------------------------------------------------------
# RoI-Transformer codecs and roi helpers.  Mirror python/jdet/ops/bbox_transforms.py:
# dbbox2delta_v3 L7-32, hbb2obb_v2 L34-44, bbox2delta L179-204, dbbox2delta_v2 L206-235,
# choose_best_match_batch L237-266, best_match_dbbox2delta L268-272, dbbox2result L274-277,
# delta2dbbox_v3 L279-321, delta2dbbox_v2 L323-360, delta2bbox L362-396, bbox2roi L398-417,
# roi2droi L434-442, choose_best_Rroi_batch L444-463, choose_best_obb_batch L465-479, dbbox2roi L481-497.
# `%` on float tensors is Python-style floor-mod in the reference = torch.remainder.
# ------------------------------------------------------------------------------------------------
def _row(v, like):
    return const_like(v, like)[None, :]      # cached on the device: no host -> device copy inside a step


def dbbox2delta_v3(proposals, gt, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    proposals, gt = proposals.float(), gt.float()
    pw, ph, pa = proposals[..., 2], proposals[..., 3], proposals[..., 4]
    coord = gt[..., 0:2] - proposals[..., 0:2]
    dx = (torch.cos(pa) * coord[..., 0] + torch.sin(pa) * coord[..., 1]) / pw
    dy = (-torch.sin(pa) * coord[..., 0] + torch.cos(pa) * coord[..., 1]) / ph
    dw = torch.log(gt[..., 2] / pw)
    dh = torch.log(gt[..., 3] / ph)
    deltas = torch.stack((dx, dy, dw, dh, gt[..., 4] - pa), -1)
    return (deltas - _row(means, deltas)) / _row(stds, deltas)


def hbb2obb_v2(boxes):
    """(x1,y1,x2,y2) -> (xc, yc, y-extent, x-extent, -pi/2): the reference's naming is swapped but
    consistent with the -90 degree angle (L34-44)."""
    ex_heights = boxes[..., 2] - boxes[..., 0] + 1.0
    ex_widths = boxes[..., 3] - boxes[..., 1] + 1.0
    ex_ctr_x = boxes[..., 0] + 0.5 * (ex_heights - 1.0)
    ex_ctr_y = boxes[..., 1] + 0.5 * (ex_widths - 1.0)
    angles = torch.full_like(ex_ctr_x, -math.pi / 2)
    return torch.stack((ex_ctr_x, ex_ctr_y, ex_widths, ex_heights, angles), 1)


def bbox2delta(proposals, gt, means=(0, 0, 0, 0), stds=(1, 1, 1, 1)):
    assert proposals.size() == gt.size()
    proposals, gt = proposals.float(), gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0] + 1.0
    ph = proposals[..., 3] - proposals[..., 1] + 1.0
    gx = (gt[..., 0] + gt[..., 2]) * 0.5
    gy = (gt[..., 1] + gt[..., 3]) * 0.5
    gw = gt[..., 2] - gt[..., 0] + 1.0
    gh = gt[..., 3] - gt[..., 1] + 1.0
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph)], dim=-1)
    return (deltas - _row(means, deltas)) / _row(stds, deltas)


def dbbox2delta_v2(proposals, gt, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    rw, rh, ra = proposals[..., 2], proposals[..., 3], proposals[..., 4]
    coord = gt[..., 0:2] - proposals[..., 0:2]
    dx = (torch.cos(ra) * coord[..., 0] + torch.sin(ra) * coord[..., 1]) / rw
    dy = (-torch.sin(ra) * coord[..., 0] + torch.cos(ra) * coord[..., 1]) / rh
    dw = torch.log(gt[..., 2] / rw)
    dh = torch.log(gt[..., 3] / rh)
    dangle = gt[..., 4] - ra
    dist = torch.remainder(dangle, 2 * math.pi)
    dist = torch.minimum(dist, math.pi * 2 - dist)
    dist = torch.where(torch.sin(dangle) < 0, -dist, dist)
    dist = dist / (math.pi / 2.)
    deltas = torch.stack((dx, dy, dw, dh, dist), -1)
    return (deltas - _row(means, deltas)) / _row(stds, deltas)


def choose_best_match_batch(Rrois, gt_rois):
    """of the 4 equivalent (w/h swapped, angle + k*pi/2) forms of each gt pick the one whose angle is
    closest to the RoI's; a gather here, a Python loop over rows in the reference (L262-263)."""
    ga = gt_rois[:, 4]
    ext = torch.stack((ga, ga + math.pi / 2., ga + math.pi, ga + math.pi * 3 / 2.), 1)
    dist = torch.remainder(Rrois[:, 4:5] - ext, 2 * math.pi)
    dist = torch.minimum(dist, math.pi * 2 - dist)
    k = torch.argmin(dist, 1)
    swap = (k % 2) == 1
    w = torch.where(swap, gt_rois[:, 3], gt_rois[:, 2])
    h = torch.where(swap, gt_rois[:, 2], gt_rois[:, 3])
    ang = torch.remainder(ext.gather(1, k[:, None])[:, 0], 2 * math.pi)
    return torch.stack((gt_rois[:, 0], gt_rois[:, 1], w, h, ang), 1)


def best_match_dbbox2delta(Rrois, gt, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1)):
    return dbbox2delta_v2(Rrois, choose_best_match_batch(Rrois, gt), means, stds)


def dbbox2result(dbboxes, labels, num_classes):
    return dbboxes[:, :8], dbboxes[:, -1].flatten(), labels


def _delta2dbbox(Rrois, deltas, means, stds, wh_ratio_clip, angle_scale):
    reps = deltas.size(1) // 5
    means = _row(means, deltas).repeat(1, reps)
    stds = _row(stds, deltas).repeat(1, reps)
    d = deltas * stds + means
    dx, dy, dw, dh, dangle = d[:, 0::5], d[:, 1::5], d[:, 2::5], d[:, 3::5], d[:, 4::5]
    max_ratio = abs(math.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    rx, ry, rw, rh, ra = [Rrois[:, i:i + 1].expand_as(dx) for i in range(5)]
    gx = dx * rw * torch.cos(ra) - dy * rh * torch.sin(ra) + rx
    gy = dx * rw * torch.sin(ra) + dy * rh * torch.cos(ra) + ry
    gw = rw * dw.exp()
    gh = rh * dh.exp()
    gangle = angle_scale * dangle + ra
    return torch.stack([gx, gy, gw, gh, gangle], dim=-1).view_as(deltas)


def delta2dbbox_v3(Rrois, deltas, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1), max_shape=None,
                   wh_ratio_clip=16 / 1000):
    return _delta2dbbox(Rrois, deltas, means, stds, wh_ratio_clip, 1.0)


def delta2dbbox_v2(Rrois, deltas, means=(0, 0, 0, 0, 0), stds=(1, 1, 1, 1, 1), max_shape=None,
                   wh_ratio_clip=16 / 1000):
    return _delta2dbbox(Rrois, deltas, means, stds, wh_ratio_clip, math.pi / 2.)


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None, wh_ratio_clip=16 / 1000):
    reps = deltas.size(1) // 4
    d = deltas * _row(stds, deltas).repeat(1, reps) + _row(means, deltas).repeat(1, reps)
    dx, dy, dw, dh = d[:, 0::4], d[:, 1::4], d[:, 2::4], d[:, 3::4]
    max_ratio = abs(math.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1).expand_as(dx)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1).expand_as(dy)
    pw = (rois[:, 2] - rois[:, 0] + 1.0).unsqueeze(1).expand_as(dw)
    ph = (rois[:, 3] - rois[:, 1] + 1.0).unsqueeze(1).expand_as(dh)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1 = gx - gw * 0.5 + 0.5
    y1 = gy - gh * 0.5 + 0.5
    x2 = gx + gw * 0.5 - 0.5
    y2 = gy + gh * 0.5 - 0.5
    if max_shape is not None:
        x1 = x1.clamp(min=0, max=max_shape[1] - 1)
        y1 = y1.clamp(min=0, max=max_shape[0] - 1)
        x2 = x2.clamp(min=0, max=max_shape[1] - 1)
        y2 = y2.clamp(min=0, max=max_shape[0] - 1)
    return torch.stack([x1, y1, x2, y2], dim=-1).view_as(deltas)


def _to_roi(box_list, ncol):
    rois_list = []
    for img_id, b in enumerate(box_list):
        if b.size(0) > 0:
            rois_list.append(torch.cat([torch.full((b.size(0), 1), float(img_id), dtype=b.dtype, device=b.device),
                                        b[:, :ncol]], dim=-1))
        else:
            rois_list.append(b.new_zeros((0, ncol + 1)))
    return torch.cat(rois_list, 0)


def bbox2roi(bbox_list):
    """list of (n,4+) boxes -> (sum n, 5) [batch_ind, x1, y1, x2, y2]"""
    return _to_roi(bbox_list, 4)


def dbbox2roi(dbbox_list):
    """list of (n,5+) rotated boxes -> (sum n, 6) [batch_ind, xc, yc, w, h, angle]"""
    return _to_roi(dbbox_list, 5)


def roi2droi(rois):
    return torch.cat((rois[:, 0:1], hbb2obb_v2(rois[:, 1:])), 1)


def choose_best_Rroi_batch(Rroi):
    """long side first (w >= h), angle % pi.  The reference edits its argument in place (L455-461); this
    returns a new tensor -- no caller on the path reads the argument afterwards except through the
    return value (roi_transformer.py:L114)."""
    w, h, a = Rroi[:, 2], Rroi[:, 3], Rroi[:, 4]
    swap = w < h
    out = Rroi.clone()
    out[:, 2] = torch.where(swap, h, w)
    out[:, 3] = torch.where(swap, w, h)
    out[:, 4] = torch.remainder(torch.where(swap, a + math.pi / 2., a), math.pi)
    return out


def choose_best_obb_batch(ori_gt_obbs):
    """angle brought into [-3pi/4, -pi/4) (closest to -90 degrees), swapping w/h when rotating by 90"""
    w, h = ori_gt_obbs[:, 2], ori_gt_obbs[:, 3]
    a = torch.remainder(ori_gt_obbs[:, 4] - math.pi / 4., math.pi)
    swap = a >= math.pi / 2
    out = ori_gt_obbs.clone()
    out[:, 2] = torch.where(swap, h, w)
    out[:, 3] = torch.where(swap, w, h)
    out[:, 4] = torch.where(swap, a - math.pi / 2., a) - math.pi * 3. / 4.
    return out
