// Deformable-conv v1 sampling kernels and the active-rotating-filter gather for gfx950.
//
// Reference semantics:
//   python/jdet/ops/dcn_v1.py:L25-56   deformable_im2col_bilinear
//   python/jdet/ops/dcn_v1.py:L130-184 deformable_im2col_gpu_kernel
//   python/jdet/ops/dcn_v1.py:L185-241 deformable_col2im_gpu_kernel      (+ get_gradient_weight L58-85)
//   python/jdet/ops/dcn_v1.py:L243-306 deformable_col2im_coord_gpu_kernel (+ get_coordinate_weight L87-128)
//   python/jdet/ops/orn.py:L17-72      ARF forward/backward
//
// All three deformable kernels are HBM-bound gathers/scatters whose fast axis is the output
// x coordinate: lanes of a wave take consecutive w_col, so offset reads, column reads/writes
// and (mostly) the image taps of a wave fall in contiguous rows.  The column matrix keeps the
// reference's (C*kh*kw, B, Ho, Wo) layout because its consumer is a plain GEMM.
#include "common.h"

namespace {

__device__ __forceinline__ float dcn_bilinear(const float* __restrict__ data, int data_width, int height,
                                              int width, float h, float w) {
  const int h_low = (int)floorf(h);
  const int w_low = (int)floorf(w);
  const int h_high = h_low + 1;
  const int w_high = w_low + 1;
  const float lh = h - h_low;
  const float lw = w - w_low;
  const float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = data[h_low * data_width + w_low];
  float v2 = 0;
  if (h_low >= 0 && w_high <= width - 1) v2 = data[h_low * data_width + w_high];
  float v3 = 0;
  if (h_high <= height - 1 && w_low >= 0) v3 = data[h_high * data_width + w_low];
  float v4 = 0;
  if (h_high <= height - 1 && w_high <= width - 1) v4 = data[h_high * data_width + w_high];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

__device__ __forceinline__ float dcn_gradient_weight(float argmax_h, float argmax_w, int h, int w,
                                                     int height, int width) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  const int hl = (int)floorf(argmax_h), wl = (int)floorf(argmax_w);
  const int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - argmax_h) * (w + 1 - argmax_w);
  if (h == hl && w == wh) weight = (h + 1 - argmax_h) * (argmax_w + 1 - w);
  if (h == hh && w == wl) weight = (argmax_h + 1 - h) * (w + 1 - argmax_w);
  if (h == hh && w == wh) weight = (argmax_h + 1 - h) * (argmax_w + 1 - w);
  return weight;
}

__device__ __forceinline__ float dcn_coordinate_weight(float argmax_h, float argmax_w, int height,
                                                       int width, const float* __restrict__ im,
                                                       int data_width, int bp_dir) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  const int hl = (int)floorf(argmax_h), wl = (int)floorf(argmax_w);
  const int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - argmax_w) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += -1 * (argmax_w - wl) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - argmax_w) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_w - wl) * im[hh * data_width + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - argmax_h) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - argmax_h) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += -1 * (argmax_h - hl) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_h - hl) * im[hh * data_width + wh];
  }
  return weight;
}

struct DcnP {
  int B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, Ho, Wo;
};

__global__ __launch_bounds__(256) void deform_im2col_kernel(long n, const float* __restrict__ im,
                                                            const float* __restrict__ offset, DcnP p,
                                                            float* __restrict__ col) {
  const int cpg = p.C / p.dg;
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < n; index += (long)gridDim.x * 256) {
    const int w_col = index % p.Wo;
    const int h_col = (index / p.Wo) % p.Ho;
    const int b = (index / p.Wo / p.Ho) % p.B;
    const int c_im = index / p.Wo / p.Ho / p.B;
    const int g = c_im / cpg;
    const int h_in = h_col * p.stride_h - p.pad_h;
    const int w_in = w_col * p.stride_w - p.pad_w;
    const size_t plane = (size_t)p.B * p.Ho * p.Wo;
    float* col_ptr = col + ((size_t)c_im * p.kh * p.kw * p.B + b) * p.Ho * p.Wo + (size_t)h_col * p.Wo + w_col;
    const float* im_ptr = im + ((size_t)b * p.C + c_im) * p.H * p.W;
    const float* off_ptr = offset + ((size_t)b * p.dg + g) * 2 * p.kh * p.kw * p.Ho * p.Wo;
    for (int i = 0; i < p.kh; ++i)
      for (int j = 0; j < p.kw; ++j) {
        const float offset_h = off_ptr[((size_t)(2 * (i * p.kw + j)) * p.Ho + h_col) * p.Wo + w_col];
        const float offset_w = off_ptr[((size_t)(2 * (i * p.kw + j) + 1) * p.Ho + h_col) * p.Wo + w_col];
        float val = 0.f;
        const float h_im = h_in + i * p.dil_h + offset_h;
        const float w_im = w_in + j * p.dil_w + offset_w;
        if (h_im > -1 && w_im > -1 && h_im < p.H && w_im < p.W)
          val = dcn_bilinear(im_ptr, p.W, p.H, p.W, h_im, w_im);
        *col_ptr = val;
        col_ptr += plane;
      }
  }
}

__global__ __launch_bounds__(256) void deform_col2im_kernel(long n, const float* __restrict__ col,
                                                            const float* __restrict__ offset, DcnP p,
                                                            float* __restrict__ grad_im) {
  const int cpg = p.C / p.dg;
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < n; index += (long)gridDim.x * 256) {
    const int j = (index / p.Wo / p.Ho / p.B) % p.kw;
    const int i = (index / p.Wo / p.Ho / p.B / p.kw) % p.kh;
    const int c = index / p.Wo / p.Ho / p.B / p.kw / p.kh;
    const int g = c / cpg;
    const int w_out = index % p.Wo;
    const int h_out = (index / p.Wo) % p.Ho;
    const int b = (index / p.Wo / p.Ho) % p.B;
    const int w_in = w_out * p.stride_w - p.pad_w;
    const int h_in = h_out * p.stride_h - p.pad_h;
    const float* off_ptr = offset + ((size_t)b * p.dg + g) * 2 * p.kh * p.kw * p.Ho * p.Wo;
    const float offset_h = off_ptr[((size_t)(2 * (i * p.kw + j)) * p.Ho + h_out) * p.Wo + w_out];
    const float offset_w = off_ptr[((size_t)(2 * (i * p.kw + j) + 1) * p.Ho + h_out) * p.Wo + w_out];
    const float cur_inv_h = h_in + i * p.dil_h + offset_h;
    const float cur_inv_w = w_in + j * p.dil_w + offset_w;
    const float cur_top_grad = col[index];
    const int cur_h = (int)cur_inv_h;
    const int cur_w = (int)cur_inv_w;
    for (int dy = -2; dy <= 2; dy++)
      for (int dx = -2; dx <= 2; dx++)
        if (cur_h + dy >= 0 && cur_h + dy < p.H && cur_w + dx >= 0 && cur_w + dx < p.W &&
            fabsf(cur_inv_h - (cur_h + dy)) < 1 && fabsf(cur_inv_w - (cur_w + dx)) < 1) {
          const size_t pos = (((size_t)b * p.C + c) * p.H + cur_h + dy) * p.W + cur_w + dx;
          const float weight = dcn_gradient_weight(cur_inv_h, cur_inv_w, cur_h + dy, cur_w + dx, p.H, p.W);
          unsafeAtomicAdd(grad_im + pos, weight * cur_top_grad);
        }
  }
}

__global__ __launch_bounds__(256) void deform_col2im_coord_kernel(long n, const float* __restrict__ col,
                                                                  const float* __restrict__ im,
                                                                  const float* __restrict__ offset, DcnP p,
                                                                  float* __restrict__ grad_offset) {
  const int cpg = p.C * p.kh * p.kw / p.dg;
  const int offset_channels = 2 * p.kh * p.kw * p.dg;
  for (long index = (long)blockIdx.x * 256 + threadIdx.x; index < n; index += (long)gridDim.x * 256) {
    float val = 0;
    const int w = index % p.Wo;
    const int h = (index / p.Wo) % p.Ho;
    const int c = (index / p.Wo / p.Ho) % offset_channels;
    const int b = (index / p.Wo / p.Ho) / offset_channels;
    const int g = c / (2 * p.kh * p.kw);
    const int col_step = p.kh * p.kw;
    int cnt = 0;
    const float* col_ptr = col + (size_t)g * cpg * p.B * p.Wo * p.Ho;
    const float* im_ptr = im + ((size_t)b * p.dg + g) * cpg / p.kh / p.kw * p.H * p.W;
    const float* off_ptr = offset + ((size_t)b * p.dg + g) * 2 * p.kh * p.kw * p.Ho * p.Wo;
    const int offset_c = c - g * 2 * p.kh * p.kw;
    const int bp_dir = offset_c % 2;
    // the tap (i,j) is fixed by offset_c: col_c = offset_c/2 + m*kh*kw  ->  (col_c / 1) % (kh*kw)
    const int tap = offset_c / 2;
    const int i = tap / p.kw, j = tap % p.kw;
    const int w_in = w * p.stride_w - p.pad_w;
    const int h_in = h * p.stride_h - p.pad_h;
    const float offset_h = off_ptr[((size_t)(2 * (i * p.kw + j)) * p.Ho + h) * p.Wo + w];
    const float offset_w = off_ptr[((size_t)(2 * (i * p.kw + j) + 1) * p.Ho + h) * p.Wo + w];
    float inv_h = h_in + i * p.dil_h + offset_h;
    float inv_w = w_in + j * p.dil_w + offset_w;
    if (inv_h <= -1 || inv_w <= -1 || inv_h >= p.H || inv_w >= p.W) inv_h = inv_w = -2;
    for (int col_c = tap; col_c < cpg; col_c += col_step) {
      const size_t col_pos = ((((size_t)col_c * p.B + b) * p.Ho) + h) * p.Wo + w;
      const float weight =
          dcn_coordinate_weight(inv_h, inv_w, p.H, p.W, im_ptr + (size_t)cnt * p.H * p.W, p.W, bp_dir);
      val += weight * col_ptr[col_pos];
      cnt += 1;
    }
    grad_offset[index] = val;
  }
}

// ARF: weight (nOut, nIn, nEntry) -> out (nOut, nRot, nIn, nEntry) permuted by the index table
__global__ __launch_bounds__(256) void arf_forward_kernel(long n, const float* __restrict__ weight,
                                                          const uint8_t* __restrict__ indices, int nIn,
                                                          int nEntry, int nRot, float* __restrict__ out) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int l = idx % nEntry;
    const int j = (idx / nEntry) % nIn;
    const int i = idx / nEntry / nIn;
    const float val = weight[idx];
    for (int k = 0; k < nRot; k++) {
      const int index = (int)indices[l * nRot + k] - 1;
      out[(size_t)i * (nRot * nIn * nEntry) + (size_t)k * (nIn * nEntry) + (size_t)j * nEntry + index] = val;
    }
  }
}

__global__ __launch_bounds__(256) void arf_backward_kernel(long n, const float* __restrict__ grad_out,
                                                           const uint8_t* __restrict__ indices, int nIn,
                                                           int nEntry, int nRot, float* __restrict__ grad_w) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int l = idx % nEntry;
    const int j = (idx / nEntry) % nIn;
    const int i = idx / nEntry / nIn;
    float val = 0;
    for (int k = 0; k < nRot; k++) {
      const int index = (int)indices[l * nRot + k] - 1;
      val = val + grad_out[(size_t)i * (nRot * nIn * nEntry) + (size_t)k * (nIn * nEntry) +
                           (size_t)j * nEntry + index];
    }
    grad_w[idx] = val;
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
  return JDET_OK;
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 262144 ? 262144 : g);
}

}  // namespace

JDET_API int jdet_deform_im2col(const float* im, const float* offset, int B, int C, int H, int W, int kh,
                                int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                int dil_w, int dg, float* col, jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!im || !offset || !col) return JDET_E_BADARG;
  const long n = (long)C * p.Ho * p.Wo * B;
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, im,
                     offset, p, col);
  return jdet_launch_status();
}

JDET_API int jdet_deform_col2im(const float* col, const float* offset, int B, int C, int H, int W, int kh,
                                int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                int dil_w, int dg, float* grad_im, jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!col || !offset || !grad_im) return JDET_E_BADARG;
  int he = jdet_zero_async(grad_im, sizeof(float) * (size_t)B * C * H * W, (hipStream_t)stream);
  if (he) return he;
  const long n = (long)C * kh * kw * p.Ho * p.Wo * B;
  hipLaunchKernelGGL(deform_col2im_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, col,
                     offset, p, grad_im);
  return jdet_launch_status();
}

JDET_API int jdet_deform_col2im_coord(const float* col, const float* im, const float* offset, int B, int C,
                                      int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                      int stride_w, int dil_h, int dil_w, int dg, float* grad_offset,
                                      jdet_stream_t stream) {
  DcnP p;
  int e = fill_dcn(p, B, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg);
  if (e) return e;
  if (B == 0) return JDET_OK;
  if (!col || !im || !offset || !grad_offset) return JDET_E_BADARG;
  const long n = (long)p.Ho * p.Wo * 2 * kh * kw * dg * B;
  hipLaunchKernelGGL(deform_col2im_coord_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n,
                     col, im, offset, p, grad_offset);
  return jdet_launch_status();
}

JDET_API int jdet_arf_forward(const float* weight, const uint8_t* indices, int nOut, int nIn, int nOri,
                              int kH, int kW, int nRot, float* out, jdet_stream_t stream) {
  if (nOut < 0 || nIn < 0 || nOri <= 0 || kH <= 0 || kW <= 0 || nRot <= 0) return JDET_E_BADARG;
  const long n = (long)nOut * nIn * nOri * kH * kW;
  if (n == 0) return JDET_OK;
  if (!weight || !indices || !out) return JDET_E_BADARG;
  hipLaunchKernelGGL(arf_forward_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n, weight,
                     indices, nIn, nOri * kH * kW, nRot, out);
  return jdet_launch_status();
}

JDET_API int jdet_arf_backward(const uint8_t* indices, const float* grad_out, int nOut, int nIn, int nOri,
                               int kH, int kW, int nRot, float* grad_weight, jdet_stream_t stream) {
  if (nOut < 0 || nIn < 0 || nOri <= 0 || kH <= 0 || kW <= 0 || nRot <= 0) return JDET_E_BADARG;
  const long n = (long)nOut * nIn * nOri * kH * kW;
  if (n == 0) return JDET_OK;
  if (!indices || !grad_out || !grad_weight) return JDET_E_BADARG;
  hipLaunchKernelGGL(arf_backward_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, n,
                     grad_out, indices, nIn, nOri * kH * kW, nRot, grad_weight);
  return jdet_launch_status();
}

JDET_API int jdet_version(void) { return 1; }
