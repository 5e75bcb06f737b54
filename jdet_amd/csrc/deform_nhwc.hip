// Deformable-conv v1 sampling for channels-last tensors (the layout the conv stack and the RoI kernels
// use on MI355X), with the input gradient as a sorted gather instead of atomics.
//
// Reference semantics are those of deform_arf.hip (dcn_v1.py:L25-56 bilinear with per-corner zero
// padding, L130-184 im2col, L185-241 col2im incl. get_gradient_weight L58-85); only the memory layout
// and the execution scheme differ:
//   * x is NHWC, the column matrix is [pos][tap][c] (pos = (b, ho, wo)): a (pos, tap) sample is 4
//     contiguous C-vectors in, one contiguous C-vector out -> one wave per (pos, tap), lanes = channels
//     (dwordx4), everything coalesced.  The consumer GEMM is  out[pos][co] = cols[pos][tap*C+c] . Wt,
//     which lands directly in NHWC; the backward GEMM grad_cols = grad_out . Wt^T needs no transposes.
//   * the NCHW reference kernel scatters every column element with up to 4 atomics (302 M fp32 atomics
//     at S2ANet P3, 1.28 ms measured).  Here the (pos, tap, corner) taps -- independent of the channel --
//     are inverted into a CSR over input pixels once (csr_gather.h) and the gradient is gathered:
//     one wave per pixel, one store per pixel, integer atomics only.
#include "csr_gather.h"

namespace {

using namespace jdet_csr;

struct DcnN {
  int B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, Ho, Wo;
};

struct Corner {
  int hl, wl;
  float h, w;
  bool inside;
};

// sample position of (pos, tap); offset is (B, 2*kh*kw, Ho, Wo), (dy, dx) per tap (dcn_v1.py:L163-170)
__device__ __forceinline__ Corner sample_pos(const DcnN& p, const float* __restrict__ offset, int b, int ho, int wo,
                                             int tap) {
  const int i = tap / p.kw, j = tap % p.kw;
  const size_t plane = (size_t)p.Ho * p.Wo;
  const float* off = offset + ((size_t)b * 2 * p.kh * p.kw) * plane + (size_t)ho * p.Wo + wo;
  const float offset_h = off[(size_t)(2 * tap) * plane];
  const float offset_w = off[(size_t)(2 * tap + 1) * plane];
  Corner c;
  c.h = (ho * p.stride_h - p.pad_h) + i * p.dil_h + offset_h;
  c.w = (wo * p.stride_w - p.pad_w) + j * p.dil_w + offset_w;
  c.inside = c.h > -1 && c.w > -1 && c.h < p.H && c.w < p.W;
  c.hl = (int)floorf(c.h);
  c.wl = (int)floorf(c.w);
  return c;
}

// one wave per (pos, tap); lanes own 4 consecutive channels of a 256-channel chunk
__global__ __launch_bounds__(256) void deform_im2col_nhwc_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ offset, DcnN p,
                                                                long nitems, float* __restrict__ cols) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int kk = p.kh * p.kw;
  for (long it = (long)blockIdx.x * 4 + wave; it < nitems; it += (long)gridDim.x * 4) {
    const int tap = (int)(it % kk);
    const long pos = it / kk;
    const int wo = (int)(pos % p.Wo);
    const int ho = (int)((pos / p.Wo) % p.Ho);
    const int b = (int)(pos / p.Wo / p.Ho);
    const Corner s = sample_pos(p, offset, b, ho, wo, tap);
    const int hh_i = s.hl + 1, wh_i = s.wl + 1;
    const float lh = s.h - s.hl, lw = s.w - s.wl;
    const float hh = 1 - lh, hw = 1 - lw;
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    const bool ok1 = s.inside && s.hl >= 0 && s.wl >= 0;
    const bool ok2 = s.inside && s.hl >= 0 && wh_i <= p.W - 1;
    const bool ok3 = s.inside && hh_i <= p.H - 1 && s.wl >= 0;
    const bool ok4 = s.inside && hh_i <= p.H - 1 && wh_i <= p.W - 1;
    const float* img = x + (size_t)b * p.H * p.W * p.C;
    float* dst = cols + (size_t)it * p.C;
    for (int c0 = 0; c0 < p.C; c0 += 256) {
      const int c = c0 + lane * 4;
      if (c >= p.C) continue;
      const v4f z = {0.f, 0.f, 0.f, 0.f};
      const v4f v1 = ok1 ? *reinterpret_cast<const v4f*>(img + ((size_t)s.hl * p.W + s.wl) * p.C + c) : z;
      const v4f v2 = ok2 ? *reinterpret_cast<const v4f*>(img + ((size_t)s.hl * p.W + wh_i) * p.C + c) : z;
      const v4f v3 = ok3 ? *reinterpret_cast<const v4f*>(img + ((size_t)hh_i * p.W + s.wl) * p.C + c) : z;
      const v4f v4 = ok4 ? *reinterpret_cast<const v4f*>(img + ((size_t)hh_i * p.W + wh_i) * p.C + c) : z;
      *reinterpret_cast<v4f*>(dst + c) = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
    }
  }
}

// taps of the input gradient: one lane per (pos, tap) -> 4 (pixel, weight) pairs (get_gradient_weight)
__global__ __launch_bounds__(256) void deform_taps_kernel(const float* __restrict__ offset, DcnN p, long nitems,
                                                         int* __restrict__ tap_key, int* __restrict__ tap_pos,
                                                         float* __restrict__ tap_w, int* __restrict__ counts) {
  const long it = (long)blockIdx.x * 256 + threadIdx.x;
  if (it >= nitems) return;
  const int kk = p.kh * p.kw;
  const int tap = (int)(it % kk);
  const long pos = it / kk;
  const int wo = (int)(pos % p.Wo);
  const int ho = (int)((pos / p.Wo) % p.Ho);
  const int b = (int)(pos / p.Wo / p.Ho);
  const Corner s = sample_pos(p, offset, b, ho, wo, tap);
  // the reference tests argmax <= -1 || argmax >= size (same open interval as `inside`)
  const int hs[4] = {s.hl, s.hl, s.hl + 1, s.hl + 1};
  const int ws[4] = {s.wl, s.wl + 1, s.wl, s.wl + 1};
  const float wts[4] = {(s.hl + 1 - s.h) * (s.wl + 1 - s.w), (s.hl + 1 - s.h) * (s.w + 1 - (s.wl + 1)),
                        (s.h + 1 - (s.hl + 1)) * (s.wl + 1 - s.w), (s.h + 1 - (s.hl + 1)) * (s.w + 1 - (s.wl + 1))};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int key = -1, pos = 0;
    if (s.inside && hs[k] >= 0 && hs[k] < p.H && ws[k] >= 0 && ws[k] < p.W && wts[k] != 0.f) {
      key = (b * p.H + hs[k]) * p.W + ws[k];
      pos = atomicAdd(&counts[key], 1);
    }
    tap_key[it * 4 + k] = key;
    tap_pos[it * 4 + k] = pos;
    tap_w[it * 4 + k] = wts[k];
  }
}

int fill(DcnN& p, int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
         int dil_h, int dil_w) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || pad_h < 0 || pad_w < 0 || stride_h <= 0 ||
      stride_w <= 0 || dil_h <= 0 || dil_w <= 0)
    return JDET_E_BADARG;
  if (C % 4 != 0) return JDET_E_UNSUPPORTED;
  p.B = B; p.C = C; p.H = H; p.W = W; p.kh = kh; p.kw = kw; p.pad_h = pad_h; p.pad_w = pad_w;
  p.stride_h = stride_h; p.stride_w = stride_w; p.dil_h = dil_h; p.dil_w = dil_w;
  p.Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  p.Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return JDET_E_BADARG;
  if ((long)B * H * W >= (1L << 30) || (long)B * p.Ho * p.Wo * kh * kw * 4 >= (1L << 31)) return JDET_E_UNSUPPORTED;
  return JDET_OK;
}

}  // namespace

JDET_API int jdet_deform_im2col_nhwc(const float* x_nhwc, const float* offset, int B, int C, int H, int W, int kh,
                                     int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                     float* cols, jdet_stream_t stream) {
  DcnN p;
  int e = fill(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!x_nhwc || !offset || !cols) return JDET_E_BADARG;
  const long nitems = (long)B * p.Ho * p.Wo * kh * kw;
  long grid = (nitems + 3) / 4;
  if (grid > 1048576) grid = 1048576;
  hipLaunchKernelGGL(deform_im2col_nhwc_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_nhwc,
                     offset, p, nitems, cols);
  return jdet_launch_status();
}

JDET_API size_t jdet_deform_col2im_nhwc_workspace(int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                                  int stride_h, int stride_w, int dil_h, int dil_w) {
  DcnN p;
  if (fill(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w) || B == 0) return 0;
  return csr_carve(nullptr, (long)B * H * W, (long)B * p.Ho * p.Wo * kh * kw * 4).bytes;
}

JDET_API int jdet_deform_col2im_nhwc(const float* grad_cols, const float* offset, int B, int C, int H, int W, int kh,
                                     int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                                     float* grad_x_nhwc, void* workspace, size_t workspace_bytes,
                                     jdet_stream_t stream) {
  DcnN p;
  int e = fill(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!grad_cols || !offset || !grad_x_nhwc || !workspace) return JDET_E_BADARG;
  const long npix = (long)B * H * W, nitems = (long)B * p.Ho * p.Wo * kh * kw, ntaps = nitems * 4;
  CsrWs w = csr_carve(workspace, npix, ntaps);
  if (workspace_bytes < w.bytes) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  int he = jdet_zero_async(w.counts, csr_zero_bytes(npix), st);
  if (he) return he;
  hipLaunchKernelGGL(deform_taps_kernel, dim3((unsigned)((nitems + 255) / 256)), dim3(256), 0, st, offset, p, nitems,
                     w.tap_key, w.tap_pos, w.tap_w, w.counts);
  return csr_finish_and_gather(w, npix, ntaps, 4, grad_cols, C, grad_x_nhwc, W, B * H, st);
}
