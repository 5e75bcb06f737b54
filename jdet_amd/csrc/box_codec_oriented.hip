// Box codecs of the Oriented R-CNN path for gfx950, one fused launch each.
//
// Reference semantics (Jittor tensor programs of 30-60 elementwise launches each, fp32):
//   MidpointOffsetCoder.encode / decode      python/jdet/models/boxes/coder.py:L332-437
//   OrientedDeltaXYWHTCoder.encode / decode  python/jdet/models/boxes/coder.py:L449-518
//   obb2poly / obb2hbb / rectpoly2obb / regular_theta / regular_obb
//                                            python/jdet/ops/bbox_transforms.py:L499-517, L575-597, L610-646
// These are tiny passes (<= 2000 x 5 rows per image in the RCNN stage, <= 10 k in the RPN): their cost in the
// reference is the launch count.  One lane per row (per (row, class) for the class-wise decode); the box
// algebra above is inlined, in the reference's order of operations.
#include "common.h"

namespace {

struct Vec6 {
  float v[6];
};

constexpr float kPi = (float)M_PI;
constexpr float kHalfPi = (float)(M_PI / 2);

// regular_theta(theta, '180', start = -pi/2): floor-mod into [-pi/2, pi/2)   (bbox_transforms.py:L499-505)
__device__ __forceinline__ float regular_theta(float theta) {
  const float start = -kHalfPi;
  const float x = theta - start;
  const float r = x - floorf(x / kPi) * kPi;
  return r + start;
}

// regular_obb: w >= h, theta wrapped (L507-517; arithmetic masks there, a select here: same values for finite input)
__device__ __forceinline__ void regular_obb(float& w, float& h, float& theta) {
  if (!(w > h)) {
    const float t = w;
    w = h;
    h = t;
    theta = theta + kHalfPi;
  }
  theta = regular_theta(theta);
}

__global__ __launch_bounds__(256) void midpoint_decode_kernel(const float* __restrict__ anchors,
                                                             const float* __restrict__ deltas, long n, Vec6 means,
                                                             Vec6 stds, float max_ratio, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* a = anchors + i * 4;
    const float* d = deltas + i * 6;
    float dd[6];
#pragma unroll
    for (int k = 0; k < 6; k++) dd[k] = d[k] * stds.v[k] + means.v[k];
    const float dw = fminf(fmaxf(dd[2], -max_ratio), max_ratio);
    const float dh = fminf(fmaxf(dd[3], -max_ratio), max_ratio);
    const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f;
    const float pw = a[2] - a[0], ph = a[3] - a[1];
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dd[0], gy = py + ph * dd[1];
    const float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
    const float da = fminf(fmaxf(dd[4], -0.5f), 0.5f), db = fminf(fmaxf(dd[5], -0.5f), 0.5f);
    // the parallelogram (top, right, bottom, left vertices), centred, stretched to equal diagonals (L404-413)
    float cx[4] = {gx + da * gw - gx, x2 - gx, gx - da * gw - gx, x1 - gx};
    float cy[4] = {y1 - gy, gy + db * gh - gy, y2 - gy, gy - db * gh - gy};
    float len[4], mx = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      len[k] = sqrtf(cx[k] * cx[k] + cy[k] * cy[k]);
      mx = k == 0 ? len[0] : fmaxf(mx, len[k]);
    }
    float rx[4], ry[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float s = mx / len[k];
      rx[k] = cx[k] * s + gx;
      ry[k] = cy[k] * s + gy;
    }
    // rectpoly2obb (L575-597)
    const float theta = atan2f(-(ry[1] - ry[0]), rx[1] - rx[0]);
    const float c = cosf(theta), s = sinf(theta);
    const float x = (rx[0] + rx[1] + rx[2] + rx[3]) / 4.f, y = (ry[0] + ry[1] + ry[2] + ry[3]) / 4.f;
    float xmin = 0, xmax = 0, ymin = 0, ymax = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float ux = rx[k] - x, uy = ry[k] - y;
      const float vx = ux * c + uy * (-s), vy = ux * s + uy * c;
      xmin = k == 0 ? vx : fminf(xmin, vx);
      xmax = k == 0 ? vx : fmaxf(xmax, vx);
      ymin = k == 0 ? vy : fminf(ymin, vy);
      ymax = k == 0 ? vy : fmaxf(ymax, vy);
    }
    float w = xmax - xmin, h = ymax - ymin, t = theta;
    regular_obb(w, h, t);
    float* o = out + i * 5;
    o[0] = x; o[1] = y; o[2] = w; o[3] = h; o[4] = t;
  }
}

__global__ __launch_bounds__(256) void midpoint_encode_kernel(const float* __restrict__ anchors,
                                                             const float* __restrict__ gt, long n, Vec6 means,
                                                             Vec6 stds, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* a = anchors + i * 4;
    const float* g = gt + i * 5;
    const float px = (a[0] + a[2]) * 0.5f, py = (a[1] + a[3]) * 0.5f;
    const float pw = a[2] - a[0], ph = a[3] - a[1];
    const float c = cosf(g[4]), s = sinf(g[4]);
    const float w2 = g[2] / 2, h2 = g[3] / 2;
    // obb2hbb (L640-646)
    const float xb = fabsf(w2 * c) + fabsf(h2 * s), yb = fabsf(w2 * s) + fabsf(h2 * c);
    const float hx0 = g[0] - xb, hy0 = g[1] - yb, hx1 = g[0] + xb, hy1 = g[1] + yb;
    const float gx = (hx0 + hx1) * 0.5f, gy = (hy0 + hy1) * 0.5f, gw = hx1 - hx0, gh = hy1 - hy0;
    // obb2poly (L626-637): v1 = (w/2 cos, -w/2 sin), v2 = (-h/2 sin, -h/2 cos)
    const float v1x = w2 * c, v1y = -w2 * s, v2x = -h2 * s, v2y = -h2 * c;
    const float qx[4] = {g[0] + v1x + v2x, g[0] + v1x - v2x, g[0] - v1x - v2x, g[0] - v1x + v2x};
    const float qy[4] = {g[1] + v1y + v2y, g[1] + v1y - v2y, g[1] - v1y - v2y, g[1] - v1y + v2y};
    const float y_min = fminf(fminf(qy[0], qy[1]), fminf(qy[2], qy[3]));
    const float x_max = fmaxf(fmaxf(qx[0], qx[1]), fmaxf(qx[2], qx[3]));
    // x of the top-most vertex, y of the right-most vertex (vertices further than 0.1 from the extreme are masked
    // to -1000, L350-356)
    float ga = 0.f, gb = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float xa = fabsf(qy[k] - y_min) > 0.1f ? -1000.f : qx[k];
      const float yb2 = fabsf(qx[k] - x_max) > 0.1f ? -1000.f : qy[k];
      ga = k == 0 ? xa : fmaxf(ga, xa);
      gb = k == 0 ? yb2 : fmaxf(gb, yb2);
    }
    float d[6] = {(gx - px) / pw, (gy - py) / ph, logf(gw / pw), logf(gh / ph), (ga - gx) / gw, (gb - gy) / gh};
    float* o = out + i * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) o[k] = (d[k] - means.v[k]) / stds.v[k];
  }
}

__global__ __launch_bounds__(256) void oriented_decode_kernel(const float* __restrict__ rois,
                                                             const float* __restrict__ deltas, long n, int ncls,
                                                             Vec6 means, Vec6 stds, float max_ratio,
                                                             float* __restrict__ out) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n * ncls; idx += (long)gridDim.x * 256) {
    const float* r = rois + (idx / ncls) * 5;
    const float* d = deltas + idx * 5;
    float dd[5];
#pragma unroll
    for (int k = 0; k < 5; k++) dd[k] = d[k] * stds.v[k] + means.v[k];
    const float dw = fminf(fmaxf(dd[2], -max_ratio), max_ratio);
    const float dh = fminf(fmaxf(dd[3], -max_ratio), max_ratio);
    const float px = r[0], py = r[1], pw = r[2], ph = r[3], pt = r[4];
    const float c = cosf(-pt), s = sinf(-pt);
    const float gx = dd[0] * pw * c - dd[1] * ph * s + px;
    const float gy = dd[0] * pw * s + dd[1] * ph * c + py;
    float gw = pw * expf(dw), gh = ph * expf(dh);
    float gt = regular_theta(dd[4] + pt);
    regular_obb(gw, gh, gt);
    float* o = out + idx * 5;
    o[0] = gx; o[1] = gy; o[2] = gw; o[3] = gh; o[4] = gt;
  }
}

__global__ __launch_bounds__(256) void oriented_encode_kernel(const float* __restrict__ rois,
                                                             const float* __restrict__ gt, long n, Vec6 means,
                                                             Vec6 stds, float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float* p = rois + i * 5;
    const float* g = gt + i * 5;
    const float px = p[0], py = p[1], pw = p[2], ph = p[3], pt = p[4];
    const float d1 = regular_theta(g[4] - pt), d2 = regular_theta(g[4] - pt + kHalfPi);
    const bool first = fabsf(d1) < fabsf(d2);          // keep (w, h) or swap them with a quarter turn (L458-465)
    const float gw = first ? g[2] : g[3], gh = first ? g[3] : g[2], dt = first ? d1 : d2;
    const float c = cosf(-pt), s = sinf(-pt);
    float d[5] = {(c * (g[0] - px) + s * (g[1] - py)) / pw, (-s * (g[0] - px) + c * (g[1] - py)) / ph,
                  logf(gw / pw), logf(gh / ph), dt};
    float* o = out + i * 5;
#pragma unroll
    for (int k = 0; k < 5; k++) o[k] = (d[k] - means.v[k]) / stds.v[k];
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 65536 ? 65536 : (g < 1 ? 1 : g));
}

inline Vec6 vec_of(const float* p, int k) {
  Vec6 v;
  for (int i = 0; i < 6; i++) v.v[i] = i < k ? p[i] : 0.f;
  return v;
}

}  // namespace

JDET_API int jdet_midpoint_offset_decode(const float* anchors_hbb, const float* deltas, long n, const float* means6,
                                         const float* stds6, float wh_ratio_clip, float* out_obb,
                                         jdet_stream_t stream) {
  if (n < 0 || !means6 || !stds6 || !(wh_ratio_clip > 0.f)) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!anchors_hbb || !deltas || !out_obb) return JDET_E_BADARG;
  hipLaunchKernelGGL(midpoint_decode_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, anchors_hbb, deltas,
                     n, vec_of(means6, 6), vec_of(stds6, 6), fabsf(logf(wh_ratio_clip)), out_obb);
  return jdet_launch_status();
}

JDET_API int jdet_midpoint_offset_encode(const float* anchors_hbb, const float* gt_obb, long n, const float* means6,
                                         const float* stds6, float* out6, jdet_stream_t stream) {
  if (n < 0 || !means6 || !stds6) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!anchors_hbb || !gt_obb || !out6) return JDET_E_BADARG;
  hipLaunchKernelGGL(midpoint_encode_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, anchors_hbb, gt_obb, n,
                     vec_of(means6, 6), vec_of(stds6, 6), out6);
  return jdet_launch_status();
}

JDET_API int jdet_oriented_delta_decode(const float* rois, const float* deltas, long n, int ncls, const float* means5,
                                        const float* stds5, float wh_ratio_clip, float* out, jdet_stream_t stream) {
  if (n < 0 || ncls <= 0 || !means5 || !stds5 || !(wh_ratio_clip > 0.f)) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!rois || !deltas || !out) return JDET_E_BADARG;
  hipLaunchKernelGGL(oriented_decode_kernel, dim3(grid_for(n * ncls)), dim3(256), 0, (hipStream_t)stream, rois, deltas, n,
                     ncls, vec_of(means5, 5), vec_of(stds5, 5), fabsf(logf(wh_ratio_clip)), out);
  return jdet_launch_status();
}

JDET_API int jdet_oriented_delta_encode(const float* rois, const float* gt, long n, const float* means5,
                                        const float* stds5, float* out, jdet_stream_t stream) {
  if (n < 0 || !means5 || !stds5) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!rois || !gt || !out) return JDET_E_BADARG;
  hipLaunchKernelGGL(oriented_encode_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, rois, gt, n,
                     vec_of(means5, 5), vec_of(stds5, 5), out);
  return jdet_launch_status();
}
