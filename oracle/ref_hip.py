"""ctypes bindings for oracle/_ref/libjdet_ref_hip.so -- the reference's own GPU kernel text compiled for gfx950 by
oracle/build_ref_hip.py.  TEST INFRASTRUCTURE ONLY (tests/, never jdet_amd): device memory comes from torch tensors,
every call synchronises the device.  `fma=True` loads the twin built with the compiler's default contraction."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = {False: os.path.join(_HERE, "_ref", "libjdet_ref_hip.so"), True: os.path.join(_HERE, "_ref", "libjdet_ref_hip_fma.so")}
_libs = {}
_i, _f, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def available():
    return os.path.exists(_SO[False]) and os.path.exists(_SO[True])


def lib(fma=False):
    if fma not in _libs:
        if not os.path.exists(_SO[fma]):
            raise RuntimeError("%s missing: run `python oracle/build_ref_hip.py` in the build container" % _SO[fma])
        _libs[fma] = ctypes.CDLL(_SO[fma])
    return _libs[fma]


def _t(x):
    assert x.is_cuda and x.is_contiguous()
    return _p(x.data_ptr() if x.numel() else 0)


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s: hipError %d" % (what, rc))


_ROI = {"rot": "refhip_roi_align_rotated", "rot_v1": "refhip_roi_align_rotated_v1", "hbb0": "refhip_roi_align_v0",
        "hbb1": "refhip_roi_align_v1"}


def roi_align_forward(kind, feat, rois, out_hw, spatial_scale, sampling_ratio, n_orient=8, fma=False):
    """feat (N, C, H, W) NCHW float32 cuda, rois (R, 6 | 5) -> (R, C, PH, PW)"""
    feat, rois = feat.float().contiguous(), rois.float().contiguous()
    N, C, H, W = feat.shape
    R, (PH, PW) = rois.shape[0], out_hw
    out = torch.zeros((R, C, PH, PW), dtype=torch.float32, device=feat.device)
    if kind == "riroi":
        _check(lib(fma).refhip_riroi_align_forward(_t(feat), _t(rois), _i(R), _i(C // n_orient), _i(H), _i(W), _i(PH),
                                                   _i(PW), _f(spatial_scale), _i(int(sampling_ratio)), _i(n_orient),
                                                   _t(out)), "riroi forward")
    else:
        _check(getattr(lib(fma), _ROI[kind] + "_forward")(_t(feat), _t(rois), _i(R), _i(C), _i(H), _i(W), _i(PH), _i(PW),
                                                          _f(spatial_scale), _f(float(sampling_ratio)), _t(out)),
               kind + " forward")
    return out


def roi_align_backward(kind, grad, rois, feat_shape, spatial_scale, sampling_ratio, n_orient=8, fma=False):
    grad, rois = grad.float().contiguous(), rois.float().contiguous()
    N, C, H, W = feat_shape
    R, _, PH, PW = grad.shape
    gin = torch.empty((N, C, H, W), dtype=torch.float32, device=grad.device)
    if kind == "riroi":
        _check(lib(fma).refhip_riroi_align_backward(_t(grad), _t(rois), _i(R), _i(N), _i(C // n_orient), _i(H), _i(W),
                                                    _i(PH), _i(PW), _f(spatial_scale), _i(int(sampling_ratio)),
                                                    _i(n_orient), _t(gin)), "riroi backward")
    else:
        _check(getattr(lib(fma), _ROI[kind] + "_backward")(_t(grad), _t(rois), _i(R), _i(N), _i(C), _i(H), _i(W), _i(PH),
                                                           _i(PW), _f(spatial_scale), _f(float(sampling_ratio)),
                                                           _t(gin)), kind + " backward")
    return gin


def _dcn_args(B, C, H, W, kh, kw, pad, stride, dil, dg):
    return [_i(v) for v in (B, C, H, W, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)]


def _dcn_out(H, W, kh, kw, pad, stride, dil):
    return ((H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1,
            (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1)


def deform_im2col(im, offset, kh, kw, pad, stride, dil, dg, fma=False):
    B, C, H, W = im.shape
    Ho, Wo = _dcn_out(H, W, kh, kw, pad, stride, dil)
    col = torch.empty((C * kh * kw, B, Ho, Wo), dtype=torch.float32, device=im.device)
    _check(lib(fma).refhip_deform_im2col(_t(im), _t(offset), *_dcn_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                         _t(col)), "deform_im2col")
    return col


def deform_col2im(col, offset, im_shape, kh, kw, pad, stride, dil, dg, fma=False):
    B, C, H, W = im_shape
    gim = torch.empty((B, C, H, W), dtype=torch.float32, device=col.device)
    _check(lib(fma).refhip_deform_col2im(_t(col), _t(offset), *_dcn_args(B, C, H, W, kh, kw, pad, stride, dil, dg),
                                         _t(gim)), "deform_col2im")
    return gim


def deform_col2im_coord(col, im, offset, kh, kw, pad, stride, dil, dg, fma=False):
    B, C, H, W = im.shape
    goff = torch.empty_like(offset)
    _check(lib(fma).refhip_deform_col2im_coord(_t(col), _t(im), _t(offset),
                                               *_dcn_args(B, C, H, W, kh, kw, pad, stride, dil, dg), _t(goff)),
           "deform_col2im_coord")
    return goff


def feature_refine(features, boxes, spatial_scale, points, grad=None, fma=False):
    """features (N, C, H, W), boxes (N, H, W, 5) -> refined features; with `grad`: the input gradient instead"""
    N, C, H, W = features.shape
    out = torch.empty_like(features)
    fn = lib(fma).refhip_feature_refine_backward if grad is not None else lib(fma).refhip_feature_refine_forward
    _check(fn(_t(grad if grad is not None else features), _t(boxes), _i(N), _i(C), _i(H), _i(W), _f(spatial_scale),
              _i(points), _t(out)), "feature_refine")
    return out


def convex_iou(pointsets, polygons, fma=False):
    N, M = pointsets.shape[0], polygons.shape[0]
    out = torch.zeros((N, M), dtype=torch.float32, device=pointsets.device)
    _check(lib(fma).refhip_convex_iou(_t(pointsets), _i(N), _t(polygons), _i(M), _t(out)), "convex_iou")
    return out


def convex_giou(pointsets, polygons, fma=False):
    """(N, 18), (N, 8) -> (N, 19): 18 point gradients + giou (convex_giou.py:L29-47)"""
    N = pointsets.shape[0]
    out = torch.zeros((N, 19), dtype=torch.float32, device=pointsets.device)
    _check(lib(fma).refhip_convex_giou(_t(pointsets), _t(polygons), _i(N), _t(out)), "convex_giou")
    return out


def min_area_bbox(pointsets, fma=False):
    out = torch.zeros((pointsets.shape[0], 8), dtype=torch.float32, device=pointsets.device)
    _check(lib(fma).refhip_min_area_bbox(_t(pointsets), _i(pointsets.shape[0]), _t(out)), "min_area_bbox")
    return out


def convex_sort(pts, masks, circular=True, fma=False):
    """convex_sort_gpu (convex_sort.py:L159-194): the tensor program in torch (first minimum, stable descending sort --
    Jittor's tie rules are unpinned), the scan by the reference kernel"""
    nbs, npts = pts.shape[:2]
    m = masks.float()
    x, y = pts[:, :, 0].contiguous(), pts[:, :, 1].contiguous()
    masked_y = m * y + (1 - m) * 10000000
    start = masked_y.argmin(1, keepdim=True)
    sx, sy = x.gather(1, start), y.gather(1, start)
    cosv = (x - sx) / torch.sqrt((x - sx) * (x - sx) + (y - sy) * (y - sy) + 0.000001)
    order = torch.sort(cosv, dim=1, descending=True, stable=True).indices.int().contiguous()
    out = torch.full((nbs, npts + (1 if circular else 0)), -1, dtype=torch.int32, device=pts.device)
    _check(lib(fma).refhip_convex_sort_scan(_t(x), _t(y), _t(m.contiguous()), _t(start.int().contiguous()), _t(order),
                                            _i(nbs), _i(npts), _i(int(circular)), _t(out)), "convex_sort")
    return out


def _dcn2_args(C, H, W, kh, kw, pad, stride, dil, dg):
    return [_i(v) for v in (C, H, W, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)]


def dcn2_im2col(im, offset, mask, kh, kw, pad, stride, dil, dg, fma=False):
    """ONE image: im (C, H, W), offset (dg*2*kk, Ho, Wo), mask (dg*kk, Ho, Wo) -> columns (C*kk, Ho, Wo)"""
    C, H, W = im.shape
    Ho, Wo = _dcn_out(H, W, kh, kw, pad, stride, dil)
    col = torch.empty((C * kh * kw, Ho, Wo), dtype=torch.float32, device=im.device)
    _check(lib(fma).refhip_dcn2_im2col(_t(im), _t(offset), _t(mask), *_dcn2_args(C, H, W, kh, kw, pad, stride, dil, dg),
                                       _t(col)), "dcn2_im2col")
    return col


def dcn2_col2im(col, offset, mask, im_shape, kh, kw, pad, stride, dil, dg, fma=False):
    C, H, W = im_shape
    gim = torch.empty((C, H, W), dtype=torch.float32, device=col.device)
    _check(lib(fma).refhip_dcn2_col2im(_t(col), _t(offset), _t(mask), *_dcn2_args(C, H, W, kh, kw, pad, stride, dil, dg),
                                       _t(gim)), "dcn2_col2im")
    return gim


def dcn2_col2im_coord(col, im, offset, mask, kh, kw, pad, stride, dil, dg, fma=False):
    C, H, W = im.shape
    goff, gmask = torch.zeros_like(offset), torch.zeros_like(mask)
    _check(lib(fma).refhip_dcn2_col2im_coord(_t(col), _t(im), _t(offset), _t(mask),
                                             *_dcn2_args(C, H, W, kh, kw, pad, stride, dil, dg), _t(goff), _t(gmask)),
           "dcn2_col2im_coord")
    return goff, gmask


def _ps_args(x, R, no_trans, scale, od, G, P, part, spp, tstd, tch):
    return [_i(x.shape[1]), _i(x.shape[2]), _i(x.shape[3]), _i(R), _i(int(no_trans)), _f(scale), _i(od), _i(G), _i(P),
            _i(part), _i(spp), _f(tstd), _i(tch)]


def psroi_forward(x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd, fma=False):
    R = rois.shape[0]
    out = torch.zeros((R, od, P, P), dtype=torch.float32, device=x.device)
    cnt = torch.zeros_like(out)
    tch = 2 if no_trans else trans.shape[1]
    _check(lib(fma).refhip_psroi_forward(_t(x), _t(rois), _t(trans), *_ps_args(x, R, no_trans, scale, od, G, P, part, spp,
                                                                                tstd, tch), _t(out), _t(cnt)),
           "psroi_forward")
    return out, cnt


def psroi_backward(grad, cnt, x, rois, trans, no_trans, scale, od, G, P, part, spp, tstd, fma=False):
    R = rois.shape[0]
    gi, gt = torch.empty_like(x), torch.zeros_like(trans)
    tch = 2 if no_trans else trans.shape[1]
    _check(lib(fma).refhip_psroi_backward(_t(grad), _t(cnt), _t(x), _t(rois), _t(trans), _i(x.shape[0]),
                                          *_ps_args(x, R, no_trans, scale, od, G, P, part, spp, tstd, tch), _t(gi),
                                          _t(gt)), "psroi_backward")
    return gi, gt


def poly_nms(boxes, thr, fma=False):
    """nms_poly.py:L187-232: boxes (n, 9) [8 coordinates, score] on the device -> kept indices in descending-score order
    (stable sort: Jittor's tie rule is unpinned)"""
    import numpy as np
    n = boxes.shape[0]
    order = torch.argsort(boxes[:, 8], descending=True, stable=True)
    srt = boxes[order].float().contiguous()
    keep = np.zeros((max(n, 1),), np.uint8)
    _check(lib(fma).refhip_poly_nms(_t(srt), _i(n), _f(thr), keep.ctypes.data_as(_p)), "poly_nms")
    return order[torch.from_numpy(keep[:n].astype(bool)).to(order.device)]


def box_iou_rotated(b1, b2, version=0, fma=False):
    """the CUDA variant of the rotated IoU (exchange-sort hull ordering, device trigonometry): (n1, 5), (n2, 5) -> (n1, n2)"""
    b1, b2 = b1.float().contiguous(), b2.float().contiguous()
    out = torch.zeros((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    fn = lib(fma).refhip_box_iou_rotated_v1 if version else lib(fma).refhip_box_iou_rotated
    _check(fn(_t(b1), _i(b1.shape[0]), _t(b2), _i(b2.shape[0]), _t(out)), "box_iou_rotated")
    return out


def nms_rotated(dets, order, thr, fma=False):
    """the CUDA rotated NMS (suppress at iou > thr): dets (n, 5 | 6) device, order (n,) visiting order -> bool keep mask
    over original indices"""
    import numpy as np
    n, bl = dets.shape
    srt = dets[order.long()].float().contiguous()
    keep = np.zeros((max(n, 1),), np.uint8)
    fn = lib(fma).refhip_nms_rotated5 if bl == 5 else lib(fma).refhip_nms_rotated6
    _check(fn(_t(srt), _i(n), _f(thr), keep.ctypes.data_as(_p)), "nms_rotated")
    mask = torch.zeros((n,), dtype=torch.bool, device=dets.device)
    mask[order.long()[torch.from_numpy(keep[:n].astype(bool)).to(dets.device)]] = True
    return mask
