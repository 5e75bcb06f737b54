"""ctypes bindings for the CPU oracle (oracle/libjdet_oracle.so) and, when present, the reference's own CPU sources
compiled for the host (oracle/_ref/libjdet_ref.so: rotated IoU, rotated NMS, ARF).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by jdet_amd.

All functions take / return numpy arrays (float32 unless noted).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_SO = os.path.join(_HERE, "libjdet_oracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libjdet_ref.so")

V_ROT, V_ROT_V1, V_RI, V_HBB0, V_HBB1 = 0, 1, 2, 3, 4

_f = ctypes.c_float
_i = ctypes.c_int
_p = ctypes.c_void_p


def build_oracle(force=False):
    src = os.path.join(_HERE, "jdet_oracle.cpp")
    if force or not os.path.exists(_ORACLE_SO) or os.path.getmtime(_ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libjdet_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _ORACLE_SO


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = ctypes.CDLL(_ORACLE_SO)
    return _lib


def have_ref():
    return os.path.exists(_REF_SO)


def ref():
    global _ref
    if _ref is None:
        if not have_ref():
            raise RuntimeError("oracle/_ref/libjdet_ref.so missing: run `python oracle/build_ref.py` "
                               "(needs /root/reference)")
        _ref = ctypes.CDLL(_REF_SO)
    return _ref


def _c(a, dt=np.float32):
    return np.ascontiguousarray(a, dtype=dt)


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


def set_threads(n):
    os.environ["OMP_NUM_THREADS"] = str(n)


# ------------------------------------------------------------------ oracle (own restatement)
def roi_align_forward(variant, feat, rois, out_hw, spatial_scale, sample_num, n_orient=1):
    feat, rois = _c(feat), _c(rois)
    N, Ct, H, W = feat.shape
    nO = n_orient if variant == V_RI else 1
    C = Ct // nO
    R = rois.shape[0]
    PH, PW = out_hw
    out = np.empty((R, Ct, PH, PW), np.float32)
    lib().jo_roi_align_forward(_i(variant), _ptr(feat), _i(N), _i(C), _i(H), _i(W), _ptr(rois), _i(R),
                               _i(PH), _i(PW), _f(spatial_scale), _i(int(sample_num)), _i(nO), _ptr(out))
    return out


def roi_align_backward(variant, grad_out, rois, feat_shape, spatial_scale, sample_num, n_orient=1):
    grad_out, rois = _c(grad_out), _c(rois)
    N, Ct, H, W = feat_shape
    nO = n_orient if variant == V_RI else 1
    C = Ct // nO
    R, _, PH, PW = grad_out.shape
    gin = np.empty((N, Ct, H, W), np.float32)
    lib().jo_roi_align_backward(_i(variant), _ptr(grad_out), _i(N), _i(C), _i(H), _i(W), _ptr(rois), _i(R),
                                _i(PH), _i(PW), _f(spatial_scale), _i(int(sample_num)), _i(nO), _ptr(gin))
    return gin


def box_iou_rotated(b1, b2, version=0, sort_mode=0):
    b1, b2 = _c(b1), _c(b2)
    n1, n2 = b1.shape[0], b2.shape[0]
    out = np.zeros((n1, n2), np.float32)
    if n1 and n2:
        lib().jo_box_iou_rotated(_ptr(b1), _i(n1), _ptr(b2), _i(n2), _i(b1.shape[1]), _i(version),
                                 _i(sort_mode), _ptr(out))
    return out


def nms_rotated_keep(dets, order, thr, cmp_ge=1, sort_mode=0):
    """dets (n, 5|6); order int32 (descending score).  Returns bool keep mask over original indices."""
    dets, order = _c(dets), _c(order, np.int32)
    n, bl = dets.shape
    keep = np.zeros((n,), np.uint8)
    if n:
        lib().jo_nms_rotated(_ptr(dets), _i(n), _i(bl), _ptr(order), _f(thr), _i(cmp_ge), _i(sort_mode),
                             _ptr(keep))
    return keep.astype(bool)


def _dcn_out(H, W, kh, kw, pad, stride, dil):
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return Ho, Wo


def deform_im2col(im, offset, kh, kw, pad, stride, dil, dg, _l=None, _name="jo_deform_im2col"):
    im, offset = _c(im), _c(offset)
    B, C, H, W = im.shape
    Ho, Wo = _dcn_out(H, W, kh, kw, pad, stride, dil)
    col = np.zeros((C * kh * kw, B, Ho, Wo), np.float32)
    getattr(_l or lib(), _name)(_ptr(im), _ptr(offset), _i(B), _i(C), _i(H), _i(W), _i(kh), _i(kw),
                                _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]), _i(dil[0]),
                                _i(dil[1]), _i(dg), _ptr(col))
    return col


def deform_col2im(col, offset, im_shape, kh, kw, pad, stride, dil, dg, _l=None, _name="jo_deform_col2im"):
    col, offset = _c(col), _c(offset)
    B, C, H, W = im_shape
    gim = np.zeros((B, C, H, W), np.float32)
    getattr(_l or lib(), _name)(_ptr(col), _ptr(offset), _i(B), _i(C), _i(H), _i(W), _i(kh), _i(kw),
                                _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]), _i(dil[0]),
                                _i(dil[1]), _i(dg), _ptr(gim))
    return gim


def deform_col2im_coord(col, im, offset, kh, kw, pad, stride, dil, dg, _l=None,
                        _name="jo_deform_col2im_coord"):
    col, im, offset = _c(col), _c(im), _c(offset)
    B, C, H, W = im.shape
    goff = np.zeros(offset.shape, np.float32)
    getattr(_l or lib(), _name)(_ptr(col), _ptr(im), _ptr(offset), _i(B), _i(C), _i(H), _i(W), _i(kh),
                                _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]), _i(dil[0]),
                                _i(dil[1]), _i(dg), _ptr(goff))
    return goff


def dcn_v2_forward(x, offset, mask, weight, bias, pad, stride, dil, dg):
    """ops/dcn_v2.py:L11-306: (B,C,H,W), (B,dg*2*kk,Ho,Wo), (B,dg*kk,Ho,Wo), (Cout,C,kh,kw), (Cout,) -> (B,Cout,Ho,Wo)"""
    x, offset, mask, weight, bias = _c(x), _c(offset), _c(mask), _c(weight), _c(bias)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho, Wo = _dcn_out(H, W, kh, kw, pad, stride, dil)
    out = np.zeros((B, Cout, Ho, Wo), np.float32)
    lib().jo_dcn_v2_forward(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(bias), _i(B), _i(C), _i(H), _i(W),
                            _i(Cout), _i(kh), _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]),
                            _i(dil[0]), _i(dil[1]), _i(dg), _ptr(out))
    return out


def dcn_v2_backward(x, offset, mask, weight, grad_out, pad, stride, dil, dg):
    """ops/dcn_v2.py:L308-781 -> (grad_input, grad_offset, grad_mask, grad_weight, grad_bias)"""
    x, offset, mask, weight, grad_out = _c(x), _c(offset), _c(mask), _c(weight), _c(grad_out)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    gi, go, gm = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(mask)
    gw, gb = np.zeros_like(weight), np.zeros((Cout,), np.float32)
    lib().jo_dcn_v2_backward(_ptr(x), _ptr(offset), _ptr(mask), _ptr(weight), _ptr(grad_out), _i(B), _i(C), _i(H),
                             _i(W), _i(Cout), _i(kh), _i(kw), _i(pad[0]), _i(pad[1]), _i(stride[0]), _i(stride[1]),
                             _i(dil[0]), _i(dil[1]), _i(dg), _ptr(gi), _ptr(go), _ptr(gm), _ptr(gw), _ptr(gb))
    return gi, go, gm, gw, gb


def _psroi_args(x, R, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, spp, trans_std, tch):
    return (_i(x.shape[1]), _i(x.shape[2]), _i(x.shape[3]), _i(R), _i(int(no_trans)), _f(spatial_scale),
            _i(output_dim), _i(group_size), _i(pooled_size), _i(part_size), _i(spp), _f(trans_std), _i(tch))


def deform_psroi_forward(x, rois, trans, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                         spp, trans_std):
    """ops/dcn_v2.py:L808-985 -> (out, top_count), both (R, output_dim, P, P)"""
    x, rois = _c(x), _c(rois)
    trans = _c(trans) if trans is not None and not no_trans else np.zeros((0, 2, part_size, part_size), np.float32)
    R = rois.shape[0]
    out = np.zeros((R, output_dim, pooled_size, pooled_size), np.float32)
    cnt = np.zeros_like(out)
    tch = 2 if no_trans else trans.shape[1]
    lib().jo_deform_psroi_forward(_ptr(x), _ptr(rois), _ptr(trans), _i(x.shape[0]), *_psroi_args(
        x, R, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, spp, trans_std, tch),
        _ptr(out), _ptr(cnt))
    return out, cnt


def deform_psroi_backward(grad_out, top_count, x, rois, trans, no_trans, spatial_scale, output_dim, group_size,
                          pooled_size, part_size, spp, trans_std):
    """ops/dcn_v2.py:L988-1175 -> (grad_input, grad_trans)"""
    grad_out, top_count, x, rois = _c(grad_out), _c(top_count), _c(x), _c(rois)
    trans = _c(trans) if trans is not None and not no_trans else np.zeros((0, 2, part_size, part_size), np.float32)
    R = rois.shape[0]
    gi, gt = np.zeros_like(x), np.zeros_like(trans)
    tch = 2 if no_trans else trans.shape[1]
    a = _psroi_args(x, R, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size, spp, trans_std, tch)
    lib().jo_deform_psroi_backward(_ptr(grad_out), _ptr(top_count), _ptr(x), _ptr(rois), _ptr(trans), _i(x.shape[0]),
                                   *a, _ptr(gi), _ptr(gt))
    return gi, gt


def convex_iou(pointsets, polygons):
    """reppoints_convex_iou (ops/reppoints_convex_iou/convex_iou.py:L29-45): (N,18), (M,8) -> (N,M)"""
    ps, pg = _c(pointsets), _c(polygons)
    out = np.zeros((ps.shape[0], pg.shape[0]), np.float32)
    if out.size:
        lib().jo_convex_iou(_ptr(ps), _i(ps.shape[0]), _ptr(pg), _i(pg.shape[0]), _ptr(out))
    return out


def convex_hull9(pointsets):
    """the Jarvis hull both RepPoints ops start from: list of (k, 2) arrays in the reference's vertex order"""
    ps = _c(pointsets)
    n = ps.shape[0]
    hull, cnt = np.zeros((n, 9, 2), np.float32), np.zeros((n,), np.int32)
    if n:
        lib().jo_convex_hull9(_ptr(ps), _i(n), _ptr(hull), _ptr(cnt))
    return [hull[i, :cnt[i]] for i in range(n)]


def min_area_bbox(pointsets):
    """reppoints_min_area_bbox (ops/reppoints_min_area_bbox/min_area_bbox.py:L22-34): (N,18) -> (N,8)"""
    ps = _c(pointsets)
    out = np.zeros((ps.shape[0], 8), np.float32)
    if ps.shape[0]:
        lib().jo_min_area_bbox(_ptr(ps), _i(ps.shape[0]), _ptr(out))
    return out


def convex_sort(pts, masks, circular=True):
    """convex_sort (ops/convex_sort.py:L196-201): (nbs,npts,2), (nbs,npts) -> (nbs, npts + circular) int32"""
    pts, masks = _c(pts), _c(masks)
    nbs, npts = pts.shape[:2]
    out = np.full((nbs, npts + (1 if circular else 0)), -1, np.int32)
    if nbs:
        lib().jo_convex_sort(_ptr(pts), _ptr(masks), _i(nbs), _i(npts), _i(int(circular)), _ptr(out))
    return out


def arf_forward(weight, indices, _l=None, _name="jo_arf_forward"):
    weight, indices = _c(weight), _c(indices, np.uint8)
    nOut, nIn, nOri, kH, kW = weight.shape
    nRot = indices.shape[3]
    out = np.zeros((nOut * nRot, nIn * nOri, kH, kW), np.float32)
    getattr(_l or lib(), _name)(_ptr(weight), _ptr(indices), _i(nOut), _i(nIn), _i(nOri), _i(kH), _i(kW),
                                _i(nRot), _ptr(out))
    return out


def arf_backward(indices, grad_out, _l=None, _name="jo_arf_backward"):
    indices, grad_out = _c(indices, np.uint8), _c(grad_out)
    nOri, kH, kW, nRot = indices.shape
    nOut = grad_out.shape[0] // nRot
    nIn = grad_out.shape[1] // nOri
    gw = np.zeros((nOut, nIn, nOri, kH, kW), np.float32)
    getattr(_l or lib(), _name)(_ptr(indices), _ptr(grad_out), _i(nOut), _i(nIn), _i(nOri), _i(kH), _i(kW),
                                _i(nRot), _ptr(gw))
    return gw


# ------------------------------------------------------------------ the reference's own CPU sources (_ref)
def ref_box_iou_rotated(b1, b2, version=0):
    b1, b2 = _c(b1), _c(b2)
    n1, n2 = b1.shape[0], b2.shape[0]
    out = np.zeros((n1, n2), np.float32)
    name = "ref_box_iou_rotated_v1" if version == 1 else "ref_box_iou_rotated"
    getattr(ref(), name)(_ptr(b1), _i(n1), _ptr(b2), _i(n2), _i(b1.shape[1]), _ptr(out))
    return out


def ref_nms_rotated_keep(dets, order, thr):
    dets, order = _c(dets), _c(order, np.int32)
    n, bl = dets.shape
    keep = np.zeros((n,), np.uint8)
    getattr(ref(), "ref_nms_rotated%d" % bl)(_ptr(dets), _i(n), _ptr(order), _f(thr), _ptr(keep))
    return keep.astype(bool)


def ref_arf_forward(*a, **k):
    return arf_forward(*a, _l=ref(), _name="ref_arf_forward", **k)


def ref_arf_backward(*a, **k):
    return arf_backward(*a, _l=ref(), _name="ref_arf_backward", **k)
