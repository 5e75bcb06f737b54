"""Tensor box algebra of the named configs.  Mirrors python/jdet/ops/bbox_transforms.py:
regular_theta / regular_obb L499-517, get_bbox_type / get_bbox_dim L519-545, rectpoly2obb L575-597,
poly2hbb L600-607, obb2poly L610-637, obb2hbb L640-646, hbb2poly L649-651, hbb2obb L654-666,
bbox2type L677-687, get_bbox_areas L689-702.  (cv2-based poly2obb / mask helpers are out of scope.)

Oriented R-CNN negates gt angles on entry, hence obb2poly's (+w/2 cos, -w/2 sin) convention here.
"""
import math

import torch


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
