// Deformable-conv v1 sampling, general path (NCHW tensors, any number of deformable groups) for gfx950.
// The channels-last fast path (what the detectors run) is deform_nhwc.hip; this file serves NCHW callers,
// deformable_groups > 1 and the offset gradient.
//
// Semantics (what has to come out, dcn_v1.py): column matrix (C*kh*kw, B, Ho, Wo); offsets (B, dg*2*kh*kw, Ho, Wo)
// ordered (dy, dx) per tap (L163-166); a sample at (h, w) = (ho*stride - pad + i*dil + dy, ...) counts when
// -1 < h < H and -1 < w < W and interpolates the four neighbours with zero outside the image (L25-56, L171-177);
// the input gradient spreads a column element over those neighbours with the bilinear weights (L58-85, L219-238),
// the offset gradient is the derivative of the interpolation w.r.t. h resp. w summed over the channels of the
// deformable group (L87-128, L262-304).
//
// Execution scheme (not the reference's one-thread-per-element grid-stride decode):
//   * work item = (image b, deformable group g, tap, output row ho, 64 consecutive wo): the LANE is the output
//     column, so offsets, column reads / writes and (mostly) the image taps of a wave are contiguous.  The sample
//     geometry -- two offset loads, floor, four corner indices / validity flags, four weights -- is computed ONCE
//     per lane and re-used for every channel of the group (the reference recomputes it per (channel, tap) thread:
//     C / dg times).  The four waves of a workgroup stride over the channels.
//   * input gradient: the four corners are known from the geometry (floor, floor + 1), no 5 x 5 neighbourhood
//     search; hardware fp32 atomics (this path is the general one -- the hot path gathers instead, deform_nhwc.hip).
//   * offset gradient: one thread per (b, g, tap, ho, wo) walks the group's channels in order and produces BOTH
//     directions (d/dh, d/dw) from the same four pixel loads and the same column element; summation order per
//     output = the reference's (bit-identical results).
//
// Modulated form (DCN v2, ops/dcn_v2.py:L86-149, L506-558, L560-627): the same three kernels with a mask plane per
// (b, g, tap) -- (B, dg*kh*kw, Ho, Wo) -- multiplying the column element (im2col), the column gradient (col2im, offset
// gradient), and with the mask gradient = sum over the group's channels of column gradient x unmasked sample
// (produced next to the offset gradient from the same pixel loads).  mask == nullptr is the v1 arithmetic unchanged.
#include "common.h"

namespace {

struct DcnP {
  int B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, Ho, Wo;
};

// bilinear sample geometry of one (b, g, tap, ho, wo)
struct Geo {
  float h, w;          // sample position
  int hl, wl;          // floor
  bool inside;         // -1 < h < H && -1 < w < W
  bool c00, c01, c10, c11;   // corner (hl,wl) (hl,wl+1) (hl+1,wl) (hl+1,wl+1) lies in the image
};

__device__ __forceinline__ Geo geometry(const DcnP& p, const float* __restrict__ offset, int b, int g, int tap,
                                        int ho, int wo) {
  const int i = tap / p.kw, j = tap - i * p.kw;
  const size_t plane = (size_t)p.Ho * p.Wo;
  const float* off = offset + (((size_t)b * p.dg + g) * 2 * p.kh * p.kw + 2 * tap) * plane + (size_t)ho * p.Wo + wo;
  Geo q;
  q.h = (ho * p.stride_h - p.pad_h) + i * p.dil_h + off[0];
  q.w = (wo * p.stride_w - p.pad_w) + j * p.dil_w + off[plane];
  q.inside = q.h > -1 && q.w > -1 && q.h < p.H && q.w < p.W;
  q.hl = (int)floorf(q.h);
  q.wl = (int)floorf(q.w);
  const bool h0 = q.hl >= 0, h1 = q.hl + 1 <= p.H - 1, w0 = q.wl >= 0, w1 = q.wl + 1 <= p.W - 1;
  q.c00 = h0 && w0; q.c01 = h0 && w1; q.c10 = h1 && w0; q.c11 = h1 && w1;
  return q;
}

__device__ __forceinline__ float mask_at(const DcnP& p, const float* __restrict__ mask, int b, int g, int tap, int ho,
                                         int wo) {
  return mask[((((size_t)b * p.dg + g) * p.kh * p.kw + tap) * p.Ho + ho) * p.Wo + wo];
}

// block -> (b, g, tap, ho, wo): blockIdx.x = wo segment, blockIdx.y = ho, blockIdx.z = (b * dg + g) * kk + tap
struct Item {
  int b, g, tap, ho, wo;
  bool ok;
};
__device__ __forceinline__ Item item_of(const DcnP& p) {
  Item it;
  const int kk = p.kh * p.kw;
  int z = blockIdx.z;
  it.tap = z % kk;
  z /= kk;
  it.g = z % p.dg;
  it.b = z / p.dg;
  it.ho = blockIdx.y;
  it.wo = blockIdx.x * 64 + (threadIdx.x & 63);
  it.ok = it.wo < p.Wo;
  return it;
}

__global__ __launch_bounds__(256) void deform_im2col_nchw_kernel(const float* __restrict__ im,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ mask, DcnP p,
                                                                float* __restrict__ col) {
  const Item it = item_of(p);
  if (!it.ok) return;
  const int wave = threadIdx.x >> 6;
  const Geo q = geometry(p, offset, it.b, it.g, it.tap, it.ho, it.wo);
  const float m = mask ? mask_at(p, mask, it.b, it.g, it.tap, it.ho, it.wo) : 1.f;
  const float lh = q.h - q.hl, lw = q.w - q.wl;
  const float hh = 1 - lh, hw = 1 - lw;
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  const int o00 = q.hl * p.W + q.wl;
  const int cpg = p.C / p.dg, kk = p.kh * p.kw;
  const size_t hw_in = (size_t)p.H * p.W, hw_out = (size_t)p.Ho * p.Wo;
  const size_t at = (size_t)it.ho * p.Wo + it.wo;
  for (int cc = wave; cc < cpg; cc += 4) {
    const int c = it.g * cpg + cc;
    const float* plane = im + ((size_t)it.b * p.C + c) * hw_in;
    float val = 0.f;
    if (q.inside) {
      const float v1 = q.c00 ? plane[o00] : 0.f;
      const float v2 = q.c01 ? plane[o00 + 1] : 0.f;
      const float v3 = q.c10 ? plane[o00 + p.W] : 0.f;
      const float v4 = q.c11 ? plane[o00 + p.W + 1] : 0.f;
      val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
    }
    col[(((size_t)c * kk + it.tap) * p.B + it.b) * hw_out + at] = mask ? val * m : val;   // dcn_v2.py:L142
  }
}

__global__ __launch_bounds__(256) void deform_col2im_nchw_kernel(const float* __restrict__ col,
                                                                const float* __restrict__ offset,
                                                                const float* __restrict__ mask, DcnP p,
                                                                float* __restrict__ grad_im) {
  const Item it = item_of(p);
  if (!it.ok) return;
  const int wave = threadIdx.x >> 6;
  const Geo q = geometry(p, offset, it.b, it.g, it.tap, it.ho, it.wo);
  if (!q.inside) return;   // contributes nothing (L58-64)
  const float m = mask ? mask_at(p, mask, it.b, it.g, it.tap, it.ho, it.wo) : 1.f;
  // weights of the four neighbours: the factors (h + 1 - argmax) / (argmax + 1 - h) of L76-83
  const float a0 = q.hl + 1 - q.h, a1 = q.h + 1 - (q.hl + 1);
  const float b0 = q.wl + 1 - q.w, b1 = q.w + 1 - (q.wl + 1);
  const float g00 = a0 * b0, g01 = a0 * b1, g10 = a1 * b0, g11 = a1 * b1;
  const int o00 = q.hl * p.W + q.wl;
  const int cpg = p.C / p.dg, kk = p.kh * p.kw;
  const size_t hw_in = (size_t)p.H * p.W, hw_out = (size_t)p.Ho * p.Wo;
  const size_t at = (size_t)it.ho * p.Wo + it.wo;
  for (int cc = wave; cc < cpg; cc += 4) {
    const int c = it.g * cpg + cc;
    float top = col[(((size_t)c * kk + it.tap) * p.B + it.b) * hw_out + at];
    if (mask) top = top * m;                                                     // dcn_v2.py:L540
    float* plane = grad_im + ((size_t)it.b * p.C + c) * hw_in;
    if (q.c00 && g00 != 0.f) unsafeAtomicAdd(plane + o00, g00 * top);
    if (q.c01 && g01 != 0.f) unsafeAtomicAdd(plane + o00 + 1, g01 * top);
    if (q.c10 && g10 != 0.f) unsafeAtomicAdd(plane + o00 + p.W, g10 * top);
    if (q.c11 && g11 != 0.f) unsafeAtomicAdd(plane + o00 + p.W + 1, g11 * top);
  }
}

// one thread per (b, g, tap, ho, wo): both offset-gradient channels (2*tap: d/dh, 2*tap + 1: d/dw)
__global__ __launch_bounds__(256) void deform_col2im_coord_nchw_kernel(const float* __restrict__ col,
                                                                      const float* __restrict__ im,
                                                                      const float* __restrict__ offset,
                                                                      const float* __restrict__ mask, DcnP p,
                                                                      float* __restrict__ grad_offset,
                                                                      float* __restrict__ grad_mask) {
  const int kk = p.kh * p.kw;
  const size_t hw_out = (size_t)p.Ho * p.Wo, hw_in = (size_t)p.H * p.W;
  const long n = (long)p.B * p.dg * kk * hw_out;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int wo = (int)(idx % p.Wo);
    const int ho = (int)((idx / p.Wo) % p.Ho);
    long z = idx / hw_out;
    const int tap = (int)(z % kk);
    z /= kk;
    const int g = (int)(z % p.dg);
    const int b = (int)(z / p.dg);
    const Geo q = geometry(p, offset, b, g, tap, ho, wo);
    float dh = 0.f, dw = 0.f, mval = 0.f;
    const float m = mask ? mask_at(p, mask, b, g, tap, ho, wo) : 1.f;
    if (q.inside) {   // (outside: the reference evaluates the weight at (-2, -2) = 0, L277-280)
      const int hl = q.hl, wl = q.wl;
      const int o00 = hl * p.W + wl;
      const int cpg = p.C / p.dg;
      const size_t at = (size_t)ho * p.Wo + wo;
      for (int cc = 0; cc < cpg; cc++) {
        const int c = g * cpg + cc;
        const float* plane = im + ((size_t)b * p.C + c) * hw_in;
        const float top = col[(((size_t)c * kk + tap) * p.B + b) * hw_out + at];
        const float v00 = q.c00 ? plane[o00] : 0.f, v01 = q.c01 ? plane[o00 + 1] : 0.f;
        const float v10 = q.c10 ? plane[o00 + p.W] : 0.f, v11 = q.c11 ? plane[o00 + p.W + 1] : 0.f;
        // d(interp)/dh and d(interp)/dw, term order of L104-125
        float wh = 0.f, ww = 0.f;
        if (q.c00) { wh += -1 * (wl + 1 - q.w) * v00; ww += -1 * (hl + 1 - q.h) * v00; }
        if (q.c01) { wh += -1 * (q.w - wl) * v01;     ww += (hl + 1 - q.h) * v01; }
        if (q.c10) { wh += (wl + 1 - q.w) * v10;      ww += -1 * (q.h - hl) * v10; }
        if (q.c11) { wh += (q.w - wl) * v11;          ww += (q.h - hl) * v11; }
        if (mask) {          // dcn_v2.py:L611-617: val += weight * col * mask; mval += col * bilinear(im)
          dh += wh * top * m;
          dw += ww * top * m;
          const float lh = q.h - hl, lw = q.w - wl, hh = 1 - lh, hw = 1 - lw;
          mval += top * (hh * hw * v00 + hh * lw * v01 + lh * hw * v10 + lh * lw * v11);
        } else {
          dh += wh * top;
          dw += ww * top;
        }
      }
    }
    if (grad_mask) grad_mask[(((size_t)b * p.dg + g) * kk + tap) * hw_out + (size_t)ho * p.Wo + wo] = mval;
    float* dst = grad_offset + (((size_t)b * p.dg + g) * 2 * kk + 2 * tap) * hw_out + (size_t)ho * p.Wo + wo;
    dst[0] = dh;
    dst[hw_out] = dw;
  }
}

int fill_dcn(DcnP& p, int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
             int stride_w, int dil_h, int dil_w, int dg) {
  if (B < 0 || C <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0 || pad_h < 0 || pad_w < 0 ||
      stride_h <= 0 || stride_w <= 0 || dil_h <= 0 || dil_w <= 0 || dg <= 0 || C % dg != 0)
    return JDET_E_BADARG;
  p.B = B; p.C = C; p.H = H; p.W = W; p.kh = kh; p.kw = kw; p.pad_h = pad_h; p.pad_w = pad_w;
  p.stride_h = stride_h; p.stride_w = stride_w; p.dil_h = dil_h; p.dil_w = dil_w; p.dg = dg;
  p.Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  p.Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  if (p.Ho <= 0 || p.Wo <= 0) return JDET_E_BADARG;
  if ((long)H * W >= (1L << 30)) return JDET_E_UNSUPPORTED;
  return JDET_OK;
}

// grid of the per-(b, g, tap, ho, 64 x wo) kernels; y / z limits of the launch API
int item_grid(const DcnP& p, dim3& grid) {
  const long z = (long)p.B * p.dg * p.kh * p.kw;
  if (p.Ho > 65535 || z > 65535) return JDET_E_UNSUPPORTED;
  grid = dim3((unsigned)((p.Wo + 63) / 64), (unsigned)p.Ho, (unsigned)z);
  return JDET_OK;
}

}  // namespace

static int im2col_impl(const float* im, const float* offset, const float* mask, int B, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                       float* col, jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!im || !offset || !col) return JDET_E_BADARG;
  dim3 grid;
  if ((e = item_grid(p, grid))) return e;
  hipLaunchKernelGGL(deform_im2col_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, im, offset, mask, p, col);
  return jdet_launch_status();
}

static int col2im_impl(const float* col, const float* offset, const float* mask, int B, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                       float* grad_im, jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!col || !offset || !grad_im) return JDET_E_BADARG;
  dim3 grid;
  if ((e = item_grid(p, grid))) return e;
  int he = jdet_zero_async(grad_im, sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream);
  if (he) return he;
  hipLaunchKernelGGL(deform_col2im_nchw_kernel, grid, dim3(256), 0, (hipStream_t)stream, col, offset, mask, p, grad_im);
  return jdet_launch_status();
}

static int coord_impl(const float* col, const float* im, const float* offset, const float* mask, int B, int C, int H,
                      int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                      int dg, float* grad_offset, float* grad_mask, jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!col || !im || !offset || !grad_offset) return JDET_E_BADARG;
  const long n = (long)p.Ho * p.Wo * kh * kw * dg * B;
  long g = (n + 255) / 256;
  if (g > 262144) g = 262144;
  hipLaunchKernelGGL(deform_col2im_coord_nchw_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, col, im,
                     offset, mask, p, grad_offset, grad_mask);
  return jdet_launch_status();
}

#define JDET_DCN_GEOM_PARAMS int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, \
                             int dil_h, int dil_w, int dg
#define JDET_DCN_GEOM_ARGS B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg

JDET_API int jdet_deform_im2col(const float* im, const float* offset, JDET_DCN_GEOM_PARAMS, float* col,
                                jdet_stream_t stream) {
  return im2col_impl(im, offset, nullptr, JDET_DCN_GEOM_ARGS, col, stream);
}
JDET_API int jdet_deform_col2im(const float* col, const float* offset, JDET_DCN_GEOM_PARAMS, float* grad_im,
                                jdet_stream_t stream) {
  return col2im_impl(col, offset, nullptr, JDET_DCN_GEOM_ARGS, grad_im, stream);
}
JDET_API int jdet_deform_col2im_coord(const float* col, const float* im, const float* offset, JDET_DCN_GEOM_PARAMS,
                                      float* grad_offset, jdet_stream_t stream) {
  return coord_impl(col, im, offset, nullptr, JDET_DCN_GEOM_ARGS, grad_offset, nullptr, stream);
}

JDET_API int jdet_modulated_deform_im2col(const float* im, const float* offset, const float* mask,
                                          JDET_DCN_GEOM_PARAMS, float* col, jdet_stream_t stream) {
  if (B > 0 && !mask) return JDET_E_BADARG;
  return im2col_impl(im, offset, mask, JDET_DCN_GEOM_ARGS, col, stream);
}
JDET_API int jdet_modulated_deform_col2im(const float* col, const float* offset, const float* mask,
                                          JDET_DCN_GEOM_PARAMS, float* grad_im, jdet_stream_t stream) {
  if (B > 0 && !mask) return JDET_E_BADARG;
  return col2im_impl(col, offset, mask, JDET_DCN_GEOM_ARGS, grad_im, stream);
}
JDET_API int jdet_modulated_deform_col2im_coord(const float* col, const float* im, const float* offset,
                                                const float* mask, JDET_DCN_GEOM_PARAMS, float* grad_offset,
                                                float* grad_mask, jdet_stream_t stream) {
  if (B > 0 && (!mask || !grad_mask)) return JDET_E_BADARG;
  return coord_impl(col, im, offset, mask, JDET_DCN_GEOM_ARGS, grad_offset, grad_mask, stream);
}
