// Feature refinement of R3Det (python/jdet/ops/fr.py): every location of a feature map adds to its own feature
// vector the map bilinearly sampled at the centre (points = 1) or at the centre and the four corners (points = 5) of
// the refined rotated box predicted at that location:
//   out[n, :, h, w] = feat[n, :, h, w] + sum_i bilinear(feat[n], py_i, px_i)          (fr.py:L121-159)
// Reference kernels: one thread per (n, c, h, w) scalar -- the box geometry (cosf / sinf included) recomputed per
// channel, 4 scattered scalar reads per point, and a backward of 1 + 4 * points float atomics per scalar (L161-215).
//
// Here (channels-last maps): forward = one wave per location, lanes = channels (dwordx4): the geometry once per
// location, every tap one contiguous C-vector.  Backward = the sorted gather of csr_gather.h: every (location,
// point, corner) tap and the identity term are inverted into a CSR over input pixels (integer atomics only) and
// each pixel's gradient vector is accumulated in registers and stored once.
// The sampling convention is the reference's, quirk included: box column 0 (x centre) scaled is used as the ROW
// coordinate and column 1 as the COLUMN coordinate (L134-135, L137-148), cosf / sinf in single precision.
#include "csr_gather.h"

namespace {

using namespace jdet_csr;

struct FrGeo {
  float px[5], py[5];
};

__device__ __forceinline__ FrGeo fr_points(const float* __restrict__ b, float scale, int points) {
  FrGeo g;
  const float roi_y = b[0] * scale, roi_x = b[1] * scale;      // (sic) fr.py:L134-135
  g.px[0] = roi_x;
  g.py[0] = roi_y;
#pragma unroll
  for (int i = 1; i < 5; i++) g.px[i] = g.py[i] = 0.f;
  if (points > 1) {
    const float w_2 = b[2] * scale / 2, h_2 = b[3] * scale / 2;
    const float cosa = cosf(b[4]), sina = sinf(b[4]);
    const float wx = cosa * w_2, wy = sina * w_2, hx = -sina * h_2, hy = cosa * h_2;
    g.px[1] = roi_x + wx + hx; g.py[1] = roi_y + wy + hy;
    g.px[2] = roi_x - wx + hx; g.py[2] = roi_y - wy + hy;
    g.px[3] = roi_x - wx - hx; g.py[3] = roi_y - wy - hy;
    g.px[4] = roi_x + wx - hx; g.py[4] = roi_y + wy - hy;
  }
  return g;
}

struct FrTap {
  int y_low, x_low, y_high, x_high;
  float w1, w2, w3, w4;
  int valid;
};

// fr.py:L18-61 (value) / L63-105 (gradient weights): the same clamping for both
__device__ __forceinline__ FrTap fr_tap(float y, float x, int H, int W) {
  FrTap t;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    t.y_low = t.x_low = t.y_high = t.x_high = 0;
    t.w1 = t.w2 = t.w3 = t.w4 = 0.f;
    t.valid = 0;
    return t;
  }
  if (y <= 0) y = 0;
  if (x <= 0) x = 0;
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) { y_high = y_low = H - 1; y = (float)y_low; } else { y_high = y_low + 1; }
  if (x_low >= W - 1) { x_high = x_low = W - 1; x = (float)x_low; } else { x_high = x_low + 1; }
  const float ly = y - y_low, lx = x - x_low;
  const float hy = (float)(1. - (double)ly), hx = (float)(1. - (double)lx);   // `1. - ly`: double literal
  t.y_low = y_low; t.x_low = x_low; t.y_high = y_high; t.x_high = x_high;
  t.w1 = hy * hx; t.w2 = hy * lx; t.w3 = ly * hx; t.w4 = ly * lx;
  t.valid = 1;
  return t;
}

__global__ __launch_bounds__(256) void fr_forward_kernel(const float* __restrict__ feat,
                                                         const float* __restrict__ boxes, int N, int C, int H, int W,
                                                         float scale, int points, float* __restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const long pos = (long)blockIdx.x * 4 + wave;
  if (pos >= (long)N * H * W) return;
  const int n = (int)(pos / ((long)H * W));
  const FrGeo g = fr_points(boxes + pos * 5, scale, points);
  const float* __restrict__ img = feat + (size_t)n * H * W * C;
  FrTap taps[5];
  for (int i = 0; i < points; i++) taps[i] = fr_tap(g.py[i], g.px[i], H, W);
  for (int c = lane * 4; c < C; c += 256) {
    v4f acc = *reinterpret_cast<const v4f*>(feat + (size_t)pos * C + c);
    for (int i = 0; i < points; i++) {
      const FrTap& t = taps[i];
      if (!t.valid) continue;    // the reference adds 0
      const v4f lt = *reinterpret_cast<const v4f*>(img + ((size_t)t.y_low * W + t.x_low) * C + c);
      const v4f rt = *reinterpret_cast<const v4f*>(img + ((size_t)t.y_low * W + t.x_high) * C + c);
      const v4f lb = *reinterpret_cast<const v4f*>(img + ((size_t)t.y_high * W + t.x_low) * C + c);
      const v4f rb = *reinterpret_cast<const v4f*>(img + ((size_t)t.y_high * W + t.x_high) * C + c);
      acc += (t.w1 * lt + t.w2 * rt + t.w3 * lb + t.w4 * rb);
    }
    *reinterpret_cast<v4f*>(out + (size_t)pos * C + c) = acc;
  }
}

// one thread per location: its 4 * points corner taps and the identity tap (own pixel, weight 1)
__global__ __launch_bounds__(256) void fr_taps_kernel(const float* __restrict__ boxes, int N, int H, int W,
                                                      float scale, int points, int* __restrict__ tap_key,
                                                      int* __restrict__ tap_pos, float* __restrict__ tap_w,
                                                      int* __restrict__ counts) {
  const long pos = (long)blockIdx.x * 256 + threadIdx.x;
  if (pos >= (long)N * H * W) return;
  const int n = (int)(pos / ((long)H * W));
  const int tps = points * 4 + 1;
  const FrGeo g = fr_points(boxes + pos * 5, scale, points);
  const long t0 = pos * tps;
  for (int i = 0; i < points; i++) {
    const FrTap t = fr_tap(g.py[i], g.px[i], H, W);
    const int ys[4] = {t.y_low, t.y_low, t.y_high, t.y_high};
    const int xs[4] = {t.x_low, t.x_high, t.x_low, t.x_high};
    const float ws[4] = {t.w1, t.w2, t.w3, t.w4};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      int key = -1, place = 0;
      if (t.valid && ws[k] != 0.f) {
        key = (n * H + ys[k]) * W + xs[k];
        place = atomicAdd(&counts[key], 1);
      }
      tap_key[t0 + i * 4 + k] = key;
      tap_pos[t0 + i * 4 + k] = place;
      tap_w[t0 + i * 4 + k] = ws[k];
    }
  }
  const int self = (int)pos;
  tap_key[t0 + tps - 1] = self;
  tap_pos[t0 + tps - 1] = atomicAdd(&counts[self], 1);
  tap_w[t0 + tps - 1] = 1.f;
}

int check(int N, int C, int H, int W, int points) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || (points != 1 && points != 5)) return JDET_E_BADARG;
  if (C % 4 != 0 || (long)N * H * W >= (1L << 30) || (long)N * H * W * (points * 4 + 1) >= (1L << 31))
    return JDET_E_UNSUPPORTED;
  return JDET_OK;
}

}  // namespace

// feat / out: (N, H, W, C) channels-last; boxes: (N, H, W, 5) [x_ctr, y_ctr, w, h, angle] per location.
JDET_API int jdet_feature_refine_forward(const float* feat_nhwc, const float* boxes, int N, int C, int H, int W,
                                         float spatial_scale, int points, float* out_nhwc, jdet_stream_t stream) {
  int e = check(N, C, H, W, points);
  if (e) return e;
  if (N == 0) return JDET_OK;
  if (!feat_nhwc || !boxes || !out_nhwc) return JDET_E_BADARG;
  const long npos = (long)N * H * W;
  hipLaunchKernelGGL(fr_forward_kernel, dim3((unsigned)((npos + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     feat_nhwc, boxes, N, C, H, W, spatial_scale, points, out_nhwc);
  return jdet_launch_status();
}

JDET_API size_t jdet_feature_refine_backward_workspace(int N, int C, int H, int W, int points) {
  if (check(N, C, H, W, points) || N == 0) return 0;
  const long npos = (long)N * H * W;
  return csr_carve(nullptr, npos, npos * (points * 4 + 1)).bytes;
}

JDET_API int jdet_feature_refine_backward(const float* grad_out_nhwc, const float* boxes, int N, int C, int H, int W,
                                          float spatial_scale, int points, float* grad_in_nhwc, void* workspace,
                                          size_t workspace_bytes, jdet_stream_t stream) {
  int e = check(N, C, H, W, points);
  if (e) return e;
  if (N == 0) return JDET_OK;
  if (!grad_out_nhwc || !boxes || !grad_in_nhwc || !workspace) return JDET_E_BADARG;
  const long npos = (long)N * H * W, ntaps = npos * (points * 4 + 1);
  CsrWs w = csr_carve(workspace, npos, ntaps);
  if (workspace_bytes < w.bytes) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int he = jdet_zero_async(w.counts, csr_zero_bytes(npos), st);
  if (he) return he;
  hipLaunchKernelGGL(fr_taps_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, boxes, N, H, W,
                     spatial_scale, points, w.tap_key, w.tap_pos, w.tap_w, w.counts);
  return csr_finish_and_gather(w, npos, ntaps, points * 4 + 1, grad_out_nhwc, C, grad_in_nhwc, W, N * H, st);
}
