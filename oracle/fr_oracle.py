"""numpy restatement of R3Det's feature refinement (python/jdet/ops/fr.py).  TEST INFRASTRUCTURE ONLY.

`feature_refine_forward_kernel` (fr.py:L113-159) and `feature_refine_backward_kernel` (L161-215) are CUDA-only in the
reference (no CPU source, no fixtures: PARITY UNPINNED by reference execution); this file restates them operation by
operation in float32 -- including the reference's use of box column 0 as the ROW coordinate (L134-135) and the
double-precision `1. - ly` -- and is itself held to a closed form (tests/test_fr_oracle.py: on an affine map bilinear
sampling is exact, so the output is the map plus its values at the sampled points).
"""
import numpy as np

F = np.float32


def sample_points(boxes, spatial_scale, points):
    """boxes (N,H,W,5) -> py, px (points, N, H, W) float32 (fr.py:L134-148)"""
    b = boxes.astype(F)
    s = F(spatial_scale)
    roi_y, roi_x = b[..., 0] * s, b[..., 1] * s              # (sic)
    py, px = [roi_y], [roi_x]
    if points > 1:
        w_2, h_2 = b[..., 2] * s / F(2), b[..., 3] * s / F(2)
        cosa, sina = np.cos(b[..., 4]).astype(F), np.sin(b[..., 4]).astype(F)
        wx, wy, hx, hy = cosa * w_2, sina * w_2, -sina * h_2, cosa * h_2
        px += [roi_x + wx + hx, roi_x - wx + hx, roi_x - wx - hx, roi_x + wx - hx]
        py += [roi_y + wy + hy, roi_y - wy + hy, roi_y - wy - hy, roi_y + wy - hy]
    return np.stack(py).astype(F), np.stack(px).astype(F)


def _taps(y, x, H, W):
    """-> y_low, x_low, y_high, x_high (int), w1..w4 (float32), valid (bool); fr.py:L18-61"""
    valid = ~((y < -1.0) | (y > H) | (x < -1.0) | (x > W))
    y = np.where(y <= 0, F(0), y)
    x = np.where(x <= 0, F(0), x)
    y_low, x_low = y.astype(np.int64), x.astype(np.int64)
    top, right = y_low >= H - 1, x_low >= W - 1
    y_low = np.where(top, H - 1, y_low)
    x_low = np.where(right, W - 1, x_low)
    y_high = np.where(top, H - 1, y_low + 1)
    x_high = np.where(right, W - 1, x_low + 1)
    y = np.where(top, y_low.astype(F), y)
    x = np.where(right, x_low.astype(F), x)
    ly, lx = (y - y_low.astype(F)).astype(F), (x - x_low.astype(F)).astype(F)
    hy, hx = (1.0 - ly.astype(np.float64)).astype(F), (1.0 - lx.astype(np.float64)).astype(F)
    ws = [hy * hx, hy * lx, ly * hx, ly * lx]
    return y_low, x_low, y_high, x_high, ws, valid


def feature_refine_forward(feat, boxes, spatial_scale, points):
    """feat (N,C,H,W), boxes (N,H,W,5) -> (N,C,H,W)"""
    feat = feat.astype(F)
    N, C, H, W = feat.shape
    py, px = sample_points(boxes, spatial_scale, points)
    out = feat.copy()
    n_idx = np.arange(N)[:, None, None]
    for i in range(points):
        yl, xl, yh, xh, ws, valid = _taps(py[i].copy(), px[i].copy(), H, W)
        lt, rt = feat[n_idx, :, yl, xl], feat[n_idx, :, yl, xh]          # (N,H,W,C)
        lb, rb = feat[n_idx, :, yh, xl], feat[n_idx, :, yh, xh]
        val = (ws[0][..., None] * lt + ws[1][..., None] * rt + ws[2][..., None] * lb + ws[3][..., None] * rb)
        val = np.where(valid[..., None], val, F(0)).astype(F)
        out = (out + val.transpose(0, 3, 1, 2)).astype(F)
    return out


def feature_refine_backward(grad_out, boxes, spatial_scale, points):
    """scatter of fr.py:L161-215 (float64 accumulation here: the kernel's atomics have no defined order)"""
    g = grad_out.astype(np.float64)
    N, C, H, W = g.shape
    py, px = sample_points(boxes, spatial_scale, points)
    gin = g.copy()
    gt = g.transpose(0, 2, 3, 1)                                           # (N,H,W,C)
    for i in range(points):
        yl, xl, yh, xh, ws, valid = _taps(py[i].copy(), px[i].copy(), H, W)
        for (yy, xx, w) in ((yl, xl, ws[0]), (yl, xh, ws[1]), (yh, xl, ws[2]), (yh, xh, ws[3])):
            for n in range(N):
                m = valid[n]
                np.add.at(gin[n].transpose(1, 2, 0), (yy[n][m], xx[n][m]), gt[n][m] * w[n][m][:, None].astype(np.float64))
    return gin.astype(F)
