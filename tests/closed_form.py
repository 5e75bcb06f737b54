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


def deform_conv_integer_expected(x, w, dy, dx, pad, stride=1, dil=1):
    """y[b,o,i,j] = sum_{c,ky,kx} w[o,c,ky,kx] * X[b,c, i*stride - pad + ky*dil + dy[ky,kx], j*stride - pad + kx*dil + dx[ky,kx]]
    with zeros outside the image; dy, dx integer arrays (kh, kw).  float64."""
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
            y += np.einsum("oc,bchw->bohw", w[:, :, ky, kx].astype(np.float64), patch)
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
