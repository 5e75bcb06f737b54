// Top-down step of a feature pyramid as ONE pass (round 6): out = (lateral + nearest_upsample(top)) / div on channels-last
// maps.  Reference: FPN.execute, python/jdet/models/necks/fpn.py:L160-171 -- `laterals[i-1] += nn.interpolate(laterals[i],
// size | scale_factor, mode="nearest")`, then `/= upsample_div_factor` -- two framework ops (an upsampled copy of the
// coarse map written and read back, 4x its bytes, then the add) and two more in backward.  Here one kernel reads the
// lateral once, the coarse map once per four fine pixels (L2 hits) and writes the result; the backward of the coarse
// operand is the sum over a coarse pixel's pre-image (the gradient of the lateral is the incoming gradient itself).
// Index rule = the framework's "nearest": src = min(floor(dst * in / out), in - 1) in float32 -- which is dst >> 1 for the
// exact 2x of every FPN here -- evaluated by one function in both directions, so forward and backward pair up for any size.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
  return min((int)floorf((float)dst * scale), in_size - 1);
}

// lat / out (N, H, W, C), top (N, Ht, Wt, C); C % 4 == 0.  One thread per (pixel, 4 channels).
__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(const float* __restrict__ lat,
                                                               const float* __restrict__ top, float* __restrict__ out,
                                                               int N, int H, int W, int Ht, int Wt, int C4,
                                                               float sy, float sx, float inv_div) {
  const long total = (long)N * H * W * C4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % C4);
    long p = t / C4;
    const int x = (int)(p % W);
    p /= W;
    const int y = (int)(p % H);
    const int n = (int)(p / H);
    const int yt = nearest_src(y, sy, Ht), xt = nearest_src(x, sx, Wt);
    const v4f a = reinterpret_cast<const v4f*>(lat)[t];
    const v4f b = reinterpret_cast<const v4f*>(top)[(((long)n * Ht + yt) * Wt + xt) * C4 + c];
    v4f o = a + b;
    if (inv_div != 1.f) o = o * inv_div;
    reinterpret_cast<v4f*>(out)[t] = o;
  }
}

// g (N, H, W, C) -> g_top (N, Ht, Wt, C): sum of g over the fine pixels whose nearest source is (yt, xt), times inv_div.
// The pre-image of a coarse index is a contiguous range around yt / scale: scanned with one index of slack on both sides.
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(const float* __restrict__ g, float* __restrict__ g_top,
                                                               int N, int H, int W, int Ht, int Wt, int C4, float sy,
                                                               float sx, float inv_div) {
  const long total = (long)N * Ht * Wt * C4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % C4);
    long p = t / C4;
    const int xt = (int)(p % Wt);
    p /= Wt;
    const int yt = (int)(p % Ht);
    const int n = (int)(p / Ht);
    const int y_lo = max(0, (int)floorf((float)yt / sy) - 1), y_hi = min(H - 1, (int)ceilf((float)(yt + 1) / sy) + 1);
    const int x_lo = max(0, (int)floorf((float)xt / sx) - 1), x_hi = min(W - 1, (int)ceilf((float)(xt + 1) / sx) + 1);
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int y = y_lo; y <= y_hi; y++) {
      if (nearest_src(y, sy, Ht) != yt) continue;
      for (int x = x_lo; x <= x_hi; x++)
        if (nearest_src(x, sx, Wt) == xt) acc += reinterpret_cast<const v4f*>(g)[(((long)n * H + y) * W + x) * C4 + c];
    }
    if (inv_div != 1.f) acc = acc * inv_div;
    reinterpret_cast<v4f*>(g_top)[t] = acc;
  }
}

int check(const void* a, const void* b, const void* c, int N, int C, int H, int W, int Ht, int Wt, float div) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || div == 0.f) return JDET_E_BADARG;
  if (C % 4 != 0) return JDET_E_UNSUPPORTED;
  if (N > 0 && (!a || !b || !c)) return JDET_E_BADARG;
  if ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) != 0) return JDET_E_BADARG;
  return JDET_OK;
}

unsigned grid_for(long total) {
  long blocks = (total + 255) / 256;
  return (unsigned)(blocks > 262144 ? 262144 : (blocks > 0 ? blocks : 1));
}

}  // namespace

JDET_API int jdet_upsample_add_nhwc_forward(const float* lateral, const float* top, int N, int C, int H, int W, int Ht,
                                            int Wt, float div_factor, float* out, jdet_stream_t stream) {
  const int e = check(lateral, top, out, N, C, H, W, Ht, Wt, div_factor);
  if (e || N == 0) return e;
  const long total = (long)N * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_add_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, lateral, top,
                     out, N, H, W, Ht, Wt, C / 4, (float)Ht / (float)H, (float)Wt / (float)W, 1.f / div_factor);
  return jdet_launch_status();
}

JDET_API int jdet_upsample_add_nhwc_backward(const float* grad_out, int N, int C, int H, int W, int Ht, int Wt,
                                             float div_factor, float* grad_top, jdet_stream_t stream) {
  const int e = check(grad_out, grad_out, grad_top, N, C, H, W, Ht, Wt, div_factor);
  if (e || N == 0) return e;
  const long total = (long)N * Ht * Wt * (C / 4);
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, grad_out,
                     grad_top, N, H, W, Ht, Wt, C / 4, (float)Ht / (float)H, (float)Wt / (float)W, 1.f / div_factor);
  return jdet_launch_status();
}
