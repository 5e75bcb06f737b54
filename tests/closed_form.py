"""Closed-form expectations that do NOT pass through the oracle restatement or the host-compiled reference text.

RoIAlign on an AFFINE feature map f_c(x, y) = a_c x + b_c y + d_c: bilinear interpolation reproduces an affine
function exactly, and the sampling grid of a bin is symmetric about the bin centre, so (as long as no sample is
clamped or dropped) the pooled value is f_c evaluated at the map position of the bin centre.  That position is
written here straight from the five dialect definitions (roi_align_rotated.py:L84-118,
roi_align_rotated_v1.py:L89-134, riroi_align.py:L104-121,L157-158, roi_align.py:L105-132): rotation sense, the
-0.5 centre shift, the max(.,1) / max(.,0) size floors and the +1 px of ROIAlign v1 each move the expected value.

DeformConv with INTEGER offsets is an ordinary correlation on an integer-shifted image: the expected output is
computed with plain numpy slicing (dcn_v1.py:L132-166 offset layout: per tap (dy, dx), tap-major).
"""
import math

import numpy as np

V_ROT, V_ROT_V1, V_RI, V_HBB0, V_HBB1 = 0, 1, 2, 3, 4
RI_PI = 3.141592653  # riroi_align.py:L8


def affine_map(rng, N, C, H, W):
    """(N,C,H,W) map f = a x + b y + d with dyadic coefficients (exact in fp32) + the coefficient arrays."""
    a = rng.integers(-8, 9, size=(N, C)).astype(np.float64) / 8.0
    b = rng.integers(-8, 9, size=(N, C)).astype(np.float64) / 8.0
    d = rng.integers(-16, 17, size=(N, C)).astype(np.float64) / 4.0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    f = a[:, :, None, None] * xs + b[:, :, None, None] * ys + d[:, :, None, None]
    return f.astype(np.float32), a, b, d


def interior_rois(rng, n, N, H, W, scale, variant, max_wh):
    """RoIs (image coordinates) whose every sample stays strictly inside [1, W-2] x [1, H-2] on the map."""
    out = []
    while len(out) < n:
        b = int(rng.integers(0, N))
        w, h = rng.uniform(0.6, max_wh, 2) / scale        # includes sub-pixel boxes (size floor)
        th = rng.uniform(-math.pi, math.pi)
        half = 0.5 * math.hypot(max(w * scale, 1.0), max(h * scale, 1.0)) + 2.5
        cx = rng.uniform(half, W - 1 - half) / scale
        cy = rng.uniform(half, H - 1 - half) / scale
        if half * 2 >= min(H, W) - 2:
            continue
        if variant in (V_HBB0, V_HBB1):
            out.append([b, cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2])
        else:
            out.append([b, cx, cy, w, h, th])
    return np.asarray(out, np.float32)


def bin_centres(variant, roi, scale, PH, PW):
    """map positions (X, Y), each (PH, PW), of the bin centres of one RoI row"""
    roi = [float(v) for v in roi]
    ph = (np.arange(PH) + 0.5)[:, None]
    pw = (np.arange(PW) + 0.5)[None, :]
    if variant in (V_HBB0, V_HBB1):
        x1, y1, x2, y2 = roi[1:5]
        sw, sh = x1 * scale, y1 * scale
        if variant == V_HBB1:
            rw, rh = max((x2 + 1) * scale - sw, 0.0), max((y2 + 1) * scale - sh, 0.0)
        else:
            rw, rh = max(x2 * scale - sw, 1.0), max(y2 * scale - sh, 1.0)
        X = sw + pw * rw / PW + 0 * ph
        Y = sh + ph * rh / PH + 0 * pw
        return X, Y
    xc, yc, w, h, th = roi[1:6]
    cx, cy = xc * scale, yc * scale
    if variant == V_ROT_V1:
        cx, cy = cx - 0.5, cy - 0.5
    rw, rh = max(w * scale, 1.0), max(h * scale, 1.0)
    xx = -rw / 2 + pw * rw / PW + 0 * ph
    yy = -rh / 2 + ph * rh / PH + 0 * pw
    c, s = math.cos(th), math.sin(th)
    if variant == V_ROT_V1:
        return xx * c + yy * s + cx, yy * c - xx * s + cy
    return xx * c - yy * s + cx, xx * s + yy * c + cy


def roi_align_expected(variant, coefs, rois, scale, PH, PW, n_orient=1):
    """(R, C, PH, PW) float64 expectation on the affine map given by coefs = (a, b, d), each (N, C)."""
    a, b, d = coefs
    R, C = rois.shape[0], a.shape[1]
    out = np.zeros((R, C, PH, PW))
    for r in range(R):
        n = int(rois[r, 0])
        X, Y = bin_centres(variant, rois[r], scale, PH, PW)
        F = a[n][:, None, None] * X + b[n][:, None, None] * Y + d[n][:, None, None]   # (C, PH, PW)
        if variant != V_RI:
            out[r] = F
            continue
        nO = n_orient
        th = float(np.float32(rois[r, 5]))
        ind_f = float(np.float32(np.float64(np.float32(th * nO)) / (2 * RI_PI)))
        ind = math.floor(ind_f)
        l = ind_f - ind
        rr = 1.0 - l
        ind = (ind + nO) % nO
        Fc = F.reshape(C // nO, nO, PH, PW)
        o = np.arange(nO)
        i0 = (o - ind + nO) % nO
        i1 = (i0 + 1 + nO) % nO
        out[r] = (rr * Fc[:, i0] + l * Fc[:, i1]).reshape(C, PH, PW)
    return out


def deform_conv_integer_expected(x, w, dy, dx, pad, stride=1, dil=1, mask=None):
    """y[b,o,i,j] = sum_{c,ky,kx} w[o,c,ky,kx] * X[b,c, i*stride - pad + ky*dil + dy[ky,kx], j*stride - pad + kx*dil + dx[ky,kx]]
    with zeros outside the image; dy, dx integer arrays (kh, kw).  float64.  mask (B, kh*kw, Ho, Wo): the modulated
    (DCN v2) form, every tap's contribution scaled by its mask value at the output position."""
    B, C, H, W = x.shape
    O, _, kh, kw = w.shape
    Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1
    big = 64
    xp = np.zeros((B, C, H + 2 * big, W + 2 * big))
    xp[:, :, big:big + H, big:big + W] = x
    y = np.zeros((B, O, Ho, Wo))
    for ky in range(kh):
        for kx in range(kw):
            oy = big - pad + ky * dil + int(dy[ky, kx])
            ox = big - pad + kx * dil + int(dx[ky, kx])
            patch = xp[:, :, oy:oy + (Ho - 1) * stride + 1:stride, ox:ox + (Wo - 1) * stride + 1:stride]
            t = np.einsum("oc,bchw->bohw", w[:, :, ky, kx].astype(np.float64), patch)
            y += t if mask is None else t * mask[:, ky * kw + kx][:, None].astype(np.float64)
    return y


def integer_offsets(dy, dx, B, Ho, Wo):
    """offset tensor (B, 2*kh*kw, Ho, Wo) in the reference layout: channel 2*(ky*kw+kx) = dy, +1 = dx"""
    kh, kw = dy.shape
    off = np.zeros((B, 2 * kh * kw, Ho, Wo), np.float32)
    for ky in range(kh):
        for kx in range(kw):
            off[:, 2 * (ky * kw + kx)] = dy[ky, kx]
            off[:, 2 * (ky * kw + kx) + 1] = dx[ky, kx]
    return off


# ---------------------------------------------------------------------------------------------------------------
# Pins the affine map cannot give.  On f = a x + b y + d the pooled value does not depend on where the samples of a
# bin sit, on their number or on the divisor; on a QUADRATIC map it does: bilinear interpolation of (x - x0)^2 between
# integer abscissae i, i + 1 at fraction t returns (x - x0)^2 + t (1 - t), so the expected pooled value is
#     mean over the bin's samples of  F(x, y) + p tx (1 - tx) + q ty (1 - ty),      F = p (x-x0)^2 + q (y-y0)^2 + affine
# with the sample positions (x, y), the sampling grid and the divisor written here from the dialect definitions:
# grid = sample_num > 0 ? sample_num : ceil(roi_size / pooled_size)  (roi_align_rotated.py:L92-97, _v1.py:L104-109,
# roi_align.py:L119-124), sample (iy, ix) of bin (ph, pw) at start + ph*bin + (iy + .5)*bin/grid (L108-116), divisor
# grid_h*grid_w (L104).
# ---------------------------------------------------------------------------------------------------------------
def quadratic_map(rng, N, C, H, W):
    """(N,C,H,W) map p (x-x0)^2 + q (y-y0)^2 + a x + b y + d, dyadic coefficients, exact in fp32."""
    x0, y0 = W // 2, H // 2
    p = rng.choice([-1.0, -0.5, -0.25, 0.25, 0.5, 1.0], size=(N, C))
    q = rng.choice([-1.0, -0.5, -0.25, 0.25, 0.5, 1.0], size=(N, C))
    a = rng.integers(-8, 9, size=(N, C)).astype(np.float64) / 8.0
    b = rng.integers(-8, 9, size=(N, C)).astype(np.float64) / 8.0
    d = rng.integers(-16, 17, size=(N, C)).astype(np.float64) / 4.0
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float64)
    e = lambda v: v[:, :, None, None]
    f = e(p) * (xs - x0) ** 2 + e(q) * (ys - y0) ** 2 + e(a) * xs + e(b) * ys + e(d)
    assert np.array_equal(f.astype(np.float32).astype(np.float64), f)
    return f.astype(np.float32), (p, q, a, b, d, x0, y0)


def roi_frame(variant, roi, scale):
    """(origin_x, origin_y, roi_w, roi_h, cos, sin, start_x, start_y) of one RoI row, from the dialect definitions"""
    roi = [float(v) for v in roi]
    if variant in (V_HBB0, V_HBB1):
        x1, y1, x2, y2 = roi[1:5]
        sw, sh = x1 * scale, y1 * scale
        if variant == V_HBB1:
            rw, rh = max((x2 + 1) * scale - sw, 0.0), max((y2 + 1) * scale - sh, 0.0)
        else:
            rw, rh = max(x2 * scale - sw, 1.0), max(y2 * scale - sh, 1.0)
        return 0.0, 0.0, rw, rh, 1.0, 0.0, sw, sh
    xc, yc, w, h, th = roi[1:6]
    cx, cy = xc * scale, yc * scale
    if variant == V_ROT_V1:
        cx, cy = cx - 0.5, cy - 0.5
    rw, rh = max(w * scale, 1.0), max(h * scale, 1.0)
    return cx, cy, rw, rh, math.cos(th), math.sin(th), -rw / 2, -rh / 2


def sample_positions(variant, roi, scale, PH, PW, sample_num):
    """X, Y of shape (PH, PW, gh, gw): map coordinates of every sample of every bin; and the divisor."""
    cx, cy, rw, rh, c, s, sx, sy = roi_frame(variant, roi, scale)
    gh = sample_num if sample_num > 0 else int(math.ceil(rh / PH))
    gw = sample_num if sample_num > 0 else int(math.ceil(rw / PW))
    bh, bw = rh / PH, rw / PW
    ph = np.arange(PH)[:, None, None, None]
    pw = np.arange(PW)[None, :, None, None]
    iy = np.arange(gh)[None, None, :, None]
    ix = np.arange(gw)[None, None, None, :]
    yy = sy + ph * bh + (iy + 0.5) * bh / gh + 0 * (pw + ix)
    xx = sx + pw * bw + (ix + 0.5) * bw / gw + 0 * (ph + iy)
    if variant in (V_HBB0, V_HBB1):
        return xx, yy, gh * gw
    if variant == V_ROT_V1:
        return xx * c + yy * s + cx, yy * c - xx * s + cy, gh * gw
    return xx * c - yy * s + cx, xx * s + yy * c + cy, gh * gw


def roi_align_expected_quadratic(variant, coefs, rois, scale, PH, PW, sample_num, n_orient=1):
    """(R, C, PH, PW) float64 expectation on quadratic_map() for RoIs whose samples all stay in [1, W-2] x [1, H-2]."""
    p, q, a, b, d, x0, y0 = coefs
    R, C = rois.shape[0], p.shape[1]
    out = np.zeros((R, C, PH, PW))
    e = lambda v, n: v[n][:, None, None, None, None]
    for r in range(R):
        n = int(rois[r, 0])
        X, Y, count = sample_positions(variant if variant != V_RI else V_ROT, rois[r], scale, PH, PW, sample_num)
        tx, ty = X - np.floor(X), Y - np.floor(Y)
        F = (e(p, n) * ((X - x0) ** 2 + tx * (1 - tx)) + e(q, n) * ((Y - y0) ** 2 + ty * (1 - ty))
             + e(a, n) * X + e(b, n) * Y + e(d, n))
        F = F.sum(axis=(3, 4)) / count                        # (C, PH, PW)
        if variant != V_RI:
            out[r] = F
            continue
        nO = n_orient
        th = float(np.float32(rois[r, 5]))
        ind_f = float(np.float32(np.float64(np.float32(th * nO)) / (2 * RI_PI)))
        ind = math.floor(ind_f)
        l = ind_f - ind
        rr = 1.0 - l
        ind = (ind + nO) % nO
        Fc = F.reshape(C // nO, nO, PH, PW)
        o = np.arange(nO)
        i0 = (o - ind + nO) % nO
        i1 = (i0 + 1 + nO) % nO
        out[r] = (rr * Fc[:, i0] + l * Fc[:, i1]).reshape(C, PH, PW)
    return out


# ---------------------------------------------------------------------------------------------------------------
# Boundary literals: a 3x3 one-channel map v[y][x] = 1 + 3 y + x, scale 1, ONE bin, ONE sample at (sx, sy) unless
# stated.  Expected values derived by hand from the bilinear routine of the reference (roi_align_rotated.py:L21-59,
# identical in the other dialects): a coordinate below -1 or above the extent contributes 0 (L26) -- but the divisor
# stays --, a coordinate in [-1, 0] is moved to 0 (L29-32), floor >= extent-1 pins both corners to the last pixel
# (L35-46).
# ---------------------------------------------------------------------------------------------------------------
BOUNDARY_MAP = (1.0 + 3.0 * np.arange(3)[:, None] + np.arange(3)[None, :]).astype(np.float32).reshape(1, 1, 3, 3)

# (sample x, sample y, expected value)
BOUNDARY_POINTS = [
    (1.0, 1.0, 5.0),                 # a pixel centre
    (0.5, 0.5, 3.0),                 # (1 + 2 + 4 + 5) / 4
    (1.25, 0.0, 2.25),               # top row, 2 + 0.25
    (-0.5, 1.0, 4.0),                # x in [-1, 0] -> x = 0: v[1][0]
    (-1.0, 1.0, 4.0),                # x = -1 is not below -1: still valid, moved to 0
    (-1.5, 1.0, 0.0),                # x < -1: the sample is dropped
    (2.5, 1.0, 6.0),                 # floor(x) = 2 >= W - 1: both corners = last column: v[1][2]
    (3.0, 1.0, 6.0),                 # x = W is not above W: valid, pinned to the last column
    (3.5, 1.0, 0.0),                 # x > W: dropped
    (1.0, -0.75, 2.0),               # y in [-1, 0] -> y = 0: v[0][1]
    (1.0, 2.75, 8.0),                # last row: v[2][1]
    (1.0, 3.25, 0.0),                # y > H: dropped
    (2.0, 2.0, 9.0),                 # the last pixel itself
    (-0.25, -0.25, 1.0),             # both moved to 0: v[0][0]
]


def boundary_roi(variant, sx, sy, w=1.0, h=1.0):
    """the RoI row (scale 1) whose single sample (one 1x1 bin, sampling 1) sits at map position (sx, sy)"""
    if variant in (V_HBB0, V_HBB1):
        # v0: extent max(x2 - x1, 1); v1: (x2 + 1) - x1.  Sample at x1 + extent / 2.
        ext_w, ext_h = w, h
        x1, y1 = sx - ext_w / 2, sy - ext_h / 2
        if variant == V_HBB1:
            return [0.0, x1, y1, x1 + ext_w - 1.0, y1 + ext_h - 1.0]
        return [0.0, x1, y1, x1 + ext_w, y1 + ext_h]
    shift = 0.5 if variant == V_ROT_V1 else 0.0
    return [0.0, sx + shift, sy + shift, w, h, 0.0]


# a 2x2-sample bin straddling the left border: RoI 2x2 centred at (-1, 1): samples x in {-1.5, -0.5}, y in {0.5, 1.5};
# x = -1.5 is dropped, x = -0.5 moves to column 0: (1 + 4)/2 + (4 + 7)/2 = 8, divisor 4 -> 2.0
BOUNDARY_STRADDLE = (-1.0, 1.0, 2.0, 2.0, 2.0)   # (centre x, centre y, w, h, expected)


# ---------------------------------------------------------------------------------------------------------------
# Deformable PSRoI pooling (ops/dcn_v2.py:L855-932) on an affine map: bilinear interpolation reproduces an affine
# function exactly, so with every sample inside the image the pooled value is the map at the MEAN sample position:
#   roi_start = round(x1) * scale - 0.5,  roi_end = (round(x2) + 1) * scale - 0.5,  roi_w = max(end - start, 0.1)
#   bin (ph, pw) starts at pw * roi_w / P + roi_start_w + trans_x * trans_std * roi_w; samples step by bin / spp
#   channel of output channel ctop in that bin: (ctop * G + floor(ph * G / P)) * G + floor(pw * G / P)
#   trans_x = trans[n, 2 * class, part_h, part_w], trans_y = trans[n, 2 * class + 1, ...], class = ctop // (dim / classes)
def psroi_expected(coef, rois, trans, scale, output_dim, G, P, part, spp, trans_std):
    """coef = (a, b, d) per input channel (f_c(x, y) = a_c x + b_c y + d_c); returns (R, output_dim, P, P) float64 and
    the derivative of every output w.r.t. its own (trans_x, trans_y) pair, (R, output_dim, P, P, 2)"""
    a, b, d = [np.asarray(v, np.float64) for v in coef]
    R = rois.shape[0]
    ncls = 1 if trans is None else trans.shape[1] // 2
    cec = output_dim // ncls
    out = np.zeros((R, output_dim, P, P))
    dtr = np.zeros((R, output_dim, P, P, 2))
    for n in range(R):
        x1, y1, x2, y2 = [float(np.round(v)) for v in rois[n, 1:5]]
        sw, sh = x1 * scale - 0.5, y1 * scale - 0.5
        ew, eh = (x2 + 1.0) * scale - 0.5, (y2 + 1.0) * scale - 0.5
        rw, rh = max(ew - sw, 0.1), max(eh - sh, 0.1)
        bw, bh = rw / P, rh / P
        for ct in range(output_dim):
            cls = ct // cec
            for ph in range(P):
                for pw in range(P):
                    part_h, part_w = int(np.floor(ph / P * part)), int(np.floor(pw / P * part))
                    tx = ty = 0.0
                    if trans is not None:
                        tx = float(trans[n, 2 * cls, part_h, part_w]) * trans_std
                        ty = float(trans[n, 2 * cls + 1, part_h, part_w]) * trans_std
                    mw = pw * bw + sw + tx * rw + (spp - 1) / 2.0 * (bw / spp)
                    mh = ph * bh + sh + ty * rh + (spp - 1) / 2.0 * (bh / spp)
                    gw = min(max(int(np.floor(pw * G / P)), 0), G - 1)
                    gh = min(max(int(np.floor(ph * G / P)), 0), G - 1)
                    c = (ct * G + gh) * G + gw
                    out[n, ct, ph, pw] = a[c] * mw + b[c] * mh + d[c]
                    dtr[n, ct, ph, pw] = (a[c] * trans_std * rw, b[c] * trans_std * rh)
    return out, dtr
