// Deformable position-sensitive RoI pooling (DCN v2 pooling), forward and backward, for gfx950.
//
// Reference: DeformablePSROIPoolForwardKernel (ops/dcn_v2.py:L855-932) and DeformablePSROIPoolBackwardAccKernel
// (L1007-1116).  input (N, H, W, C) channels-last with C = output_dim * group_size^2 (the reference: NCHW); rois (R, 5) [batch, x1, y1, x2, y2] in
// image pixels; trans (R, 2 * num_classes, part, part) or absent (no_trans); out / top_count (R, P, P, output_dim) = the
// reference's (R, output_dim, P, P) stored channels-last.
//   * RoI frame: corners rounded to integers, end + 1, x spatial_scale, shifted by -0.5; width / height >= 0.1
//   * bin (ph, pw) starts at ph * bin_h + roi_start_h + trans_y * roi_height; its sample_per_part^2 samples step by
//     bin / sample_per_part; a sample outside [-0.5, W - 0.5] x [-0.5, H - 0.5] is skipped, the others are clamped to
//     the image and interpolated between floor and CEIL (weights (1 - dx), dx: an integer coordinate reads one pixel)
//   * channel read by output channel ctop in bin (ph, pw): (ctop * group + gh) * group + gw, gh = floor(ph * group / P)
//   * out = sum / count (0 when no sample counted); top_count = count, kept for the backward
//   * backward: diff / count spread over the four corners (fp32 atomics, as the reference), and -- with trans -- the
//     derivative of the bilinear sample w.r.t. the shift, x trans_std x roi size, accumulated per (roi, class, part).
// Execution (round 6 redesign; rounds 3-5 ran the reference's decomposition -- one thread per output element, pw fastest,
// NCHW planes, four scattered fp32 atomics per sample and thread): channels-last input, ONE WAVE per (RoI, bin, class,
// chunk of output channels) with the output channels across the lanes.  The bin's frame, its samples' positions, clamps
// and bilinear weights depend only on (RoI, bin, class): they are wave-uniform, computed once per wave; a corner read is
// one coalesced access of the pixel's channel vector (contiguous lanes when group_size == 1 -- four channels per lane
// and dwordx4 then -- stride group^2 otherwise), the result row (r, ph, pw, :) one contiguous store.  Backward: the
// four corner updates of a sample are lane-contiguous atomic adds on the channels-last gradient (the coalesced form the
// RoIAlign atomic path uses), and the shift gradient of a (RoI, class, part cell) is summed over the wave's samples in
// registers and over its lanes by a wave reduction: two atomics per WAVE instead of two per thread and sample.
#include "common.h"

namespace {

struct PsP {
  int N, C, H, W, R, no_trans, output_dim, group, P, part, spp, num_classes, ch_each_class;
  float scale, trans_std;
};

struct Bin {
  int n, ctop, ph, pw, batch, class_id, part_h, part_w, gh, gw;
  float roi_w, roi_h, wstart, hstart, sub_w, sub_h;
};

__device__ __forceinline__ Bin bin_of(const PsP& p, long index, const float* __restrict__ rois,
                                      const float* __restrict__ trans) {
  Bin b;
  b.pw = (int)(index % p.P);
  b.ph = (int)((index / p.P) % p.P);
  b.ctop = (int)((index / p.P / p.P) % p.output_dim);
  b.n = (int)(index / p.P / p.P / p.output_dim);
  const float* r = rois + (size_t)b.n * 5;
  b.batch = (int)r[0];
  // L873-876: (float)round(x) * scale - 0.5 (the 0.5 is a double literal: the product is widened, then narrowed)
  const float roi_start_w = (float)((double)((float)round(r[1]) * p.scale) - 0.5);
  const float roi_start_h = (float)((double)((float)round(r[2]) * p.scale) - 0.5);
  const float roi_end_w = (float)((double)((float)(round(r[3]) + 1.) * p.scale) - 0.5);
  const float roi_end_h = (float)((double)((float)(round(r[4]) + 1.) * p.scale) - 0.5);
  b.roi_w = (float)fmax((double)(roi_end_w - roi_start_w), 0.1);
  b.roi_h = (float)fmax((double)(roi_end_h - roi_start_h), 0.1);
  const float bin_h = b.roi_h / (float)p.P, bin_w = b.roi_w / (float)p.P;
  b.sub_h = bin_h / (float)p.spp;
  b.sub_w = bin_w / (float)p.spp;
  b.part_h = (int)floorf((float)b.ph / p.P * p.part);
  b.part_w = (int)floorf((float)b.pw / p.P * p.part);
  b.class_id = b.ctop / p.ch_each_class;
  float tx = 0.f, ty = 0.f;
  if (!p.no_trans) {
    const size_t t = (((size_t)b.n * p.num_classes + b.class_id) * 2) * p.part;
    tx = trans[(t + b.part_h) * p.part + b.part_w] * p.trans_std;
    ty = trans[(t + p.part + b.part_h) * p.part + b.part_w] * p.trans_std;
  }
  b.wstart = (float)b.pw * bin_w + roi_start_w;
  b.wstart += tx * b.roi_w;
  b.hstart = (float)b.ph * bin_h + roi_start_h;
  b.hstart += ty * b.roi_h;
  int gw = (int)floorf((float)b.pw * p.group / p.P), gh = (int)floorf((float)b.ph * p.group / p.P);
  b.gw = min(max(gw, 0), p.group - 1);
  b.gh = min(max(gh, 0), p.group - 1);
  return b;
}

// sample (ih, iw) of a bin: false = skipped; else the clamped position (L906-915)
__device__ __forceinline__ bool sample_of(const PsP& p, const Bin& b, int ih, int iw, float& w, float& h) {
  w = b.wstart + iw * b.sub_w;
  h = b.hstart + ih * b.sub_h;
  if ((double)w < -0.5 || (double)w > p.W - 0.5 || (double)h < -0.5 || (double)h > p.H - 0.5) return false;
  w = (float)fmin(fmax((double)w, 0.), p.W - 1.);
  h = (float)fmin(fmax((double)h, 0.), p.H - 1.);
  return true;
}

// wave-uniform frame of (RoI n, bin ph / pw, class): bin_of() with the channel left out
__device__ __forceinline__ Bin frame_of(const PsP& p, int n, int ph, int pw, int class_id, const float* __restrict__ rois,
                                        const float* __restrict__ trans) {
  const long index = (((long)n * p.output_dim + (long)class_id * p.ch_each_class) * p.P + ph) * p.P + pw;
  return bin_of(p, index, rois, trans);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}

// input (N, H, W, C), out / top_count (R, P, P, output_dim): channels-last.  VEC output channels per lane (4: group == 1).
// One wave per (n, ph, pw, class, chunk of 64 * VEC channels of the class); 4 waves per workgroup.
template <int VEC>
__global__ __launch_bounds__(256) void psroi_fwd_kernel(const float* __restrict__ input, const float* __restrict__ rois,
                                                       const float* __restrict__ trans, PsP p, int chunks, long waves,
                                                       float* __restrict__ out, float* __restrict__ top_count) {
  typedef float vf __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < waves; wv += (long)gridDim.x * 4) {
    const int chunk = (int)(wv % chunks);
    long rest = wv / chunks;
    const int class_id = (int)(rest % p.num_classes);
    rest /= p.num_classes;
    const int pw = (int)(rest % p.P);
    rest /= p.P;
    const int ph = (int)(rest % p.P);
    const int n = (int)(rest / p.P);
    const Bin b = frame_of(p, n, ph, pw, class_id, rois, trans);
    const int cc = (chunk * 64 + lane) * VEC;                 // first of my channels inside the class
    const bool mine = cc < p.ch_each_class;
    const int ctop = class_id * p.ch_each_class + (mine ? cc : 0);
    const size_t o = (((size_t)n * p.P + ph) * p.P + pw) * p.output_dim + ctop;
    vf sum = 0.f;
    int cnt = 0;
    if (b.batch >= 0 && b.batch < p.N) {     // (the reference reads out of bounds here; a bad batch index pools nothing)
      const int c = (ctop * p.group + b.gh) * p.group + b.gw;
      const float* img = input + (size_t)b.batch * p.H * p.W * p.C + c;
      for (int ih = 0; ih < p.spp; ih++)
        for (int iw = 0; iw < p.spp; iw++) {
          float w, h;
          if (!sample_of(p, b, ih, iw, w, h)) continue;
          // bilinear_interp L832-854: floor / ceil corners
          const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
          const float dx = w - x1, dy = h - y1;
          cnt++;
          if (!mine) continue;
          const vf v11 = *reinterpret_cast<const vf*>(img + ((size_t)y1 * p.W + x1) * p.C);
          const vf v12 = *reinterpret_cast<const vf*>(img + ((size_t)y2 * p.W + x1) * p.C);
          const vf v21 = *reinterpret_cast<const vf*>(img + ((size_t)y1 * p.W + x2) * p.C);
          const vf v22 = *reinterpret_cast<const vf*>(img + ((size_t)y2 * p.W + x2) * p.C);
          sum += (1 - dx) * (1 - dy) * v11 + (1 - dx) * dy * v12 + dx * (1 - dy) * v21 + dx * dy * v22;
        }
    }
    if (mine) {
      vf res = 0.f, cv = (float)cnt;
      if (cnt) res = sum / (float)cnt;
      *reinterpret_cast<vf*>(out + o) = res;
      *reinterpret_cast<vf*>(top_count + o) = cv;
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(256) void psroi_bwd_kernel(const float* __restrict__ top_diff,
                                                       const float* __restrict__ top_count,
                                                       const float* __restrict__ input, const float* __restrict__ rois,
                                                       const float* __restrict__ trans, PsP p, int chunks, long waves,
                                                       float* __restrict__ grad_input, float* __restrict__ grad_trans) {
  typedef float vf __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63;
  for (long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6); wv < waves; wv += (long)gridDim.x * 4) {
    const int chunk = (int)(wv % chunks);
    long rest = wv / chunks;
    const int class_id = (int)(rest % p.num_classes);
    rest /= p.num_classes;
    const int pw = (int)(rest % p.P);
    rest /= p.P;
    const int ph = (int)(rest % p.P);
    const int n = (int)(rest / p.P);
    const Bin b = frame_of(p, n, ph, pw, class_id, rois, trans);
    if (b.batch < 0 || b.batch >= p.N) continue;                // pooled nothing: count 0
    const int cc = (chunk * 64 + lane) * VEC;
    const bool mine = cc < p.ch_each_class;
    const int ctop = class_id * p.ch_each_class + (mine ? cc : 0);
    const size_t o = (((size_t)n * p.P + ph) * p.P + pw) * p.output_dim + ctop;
    const float count = top_count[o];                           // the same for every channel of the class: wave-uniform
    if (count <= 0) continue;
    vf diff_val = 0.f;
    if (mine) diff_val = *reinterpret_cast<const vf*>(top_diff + o) / count;
    const int c = (ctop * p.group + b.gh) * p.group + b.gw;
    const size_t base = (size_t)b.batch * p.H * p.W * p.C + c;
    const float* img = input + base;
    float* gimg = grad_input + base;
    float gx = 0.f, gy = 0.f;                                   // my channels' share of the shift gradient
    for (int ih = 0; ih < p.spp; ih++)
      for (int iw = 0; iw < p.spp; iw++) {
        float w, h;
        if (!sample_of(p, b, ih, iw, w, h)) continue;
        if (!mine) continue;
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - x0, dy = h - y0;
        const float q00 = (1 - dx) * (1 - dy), q01 = (1 - dx) * dy, q10 = dx * (1 - dy), q11 = dx * dy;
        const size_t o00 = ((size_t)y0 * p.W + x0) * p.C, o01 = ((size_t)y1 * p.W + x0) * p.C;
        const size_t o10 = ((size_t)y0 * p.W + x1) * p.C, o11 = ((size_t)y1 * p.W + x1) * p.C;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          const float dv = diff_val[k];
          unsafeAtomicAdd(gimg + o00 + k, q00 * dv);
          unsafeAtomicAdd(gimg + o01 + k, q01 * dv);
          unsafeAtomicAdd(gimg + o10 + k, q10 * dv);
          unsafeAtomicAdd(gimg + o11 + k, q11 * dv);
        }
        if (p.no_trans) continue;
        const vf U00 = *reinterpret_cast<const vf*>(img + o00), U01 = *reinterpret_cast<const vf*>(img + o01);
        const vf U10 = *reinterpret_cast<const vf*>(img + o10), U11 = *reinterpret_cast<const vf*>(img + o11);
        vf diff_x = (U11 * dy + U10 * (1 - dy) - U01 * dy - U00 * (1 - dy)) * p.trans_std * diff_val;
        diff_x *= b.roi_w;
        vf diff_y = (U11 * dx + U01 * (1 - dx) - U10 * dx - U00 * (1 - dx)) * p.trans_std * diff_val;
        diff_y *= b.roi_h;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
          gx += diff_x[k];
          gy += diff_y[k];
        }
      }
    if (!p.no_trans) {
      gx = wave_sum(gx);
      gy = wave_sum(gy);
      if (lane == 0) {
        const size_t t = (((size_t)b.n * p.num_classes + b.class_id) * 2) * p.part;
        unsafeAtomicAdd(grad_trans + (t + b.part_h) * p.part + b.part_w, gx);
        unsafeAtomicAdd(grad_trans + (t + p.part + b.part_h) * p.part + b.part_w, gy);
      }
    }
  }
}

int fill(PsP& p, int N, int C, int H, int W, int R, int no_trans, float spatial_scale, int output_dim, int group_size,
         int pooled_size, int part_size, int sample_per_part, float trans_std, int trans_channels) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || output_dim <= 0 || group_size <= 0 || pooled_size <= 0 ||
      part_size <= 0 || sample_per_part <= 0)
    return JDET_E_BADARG;
  if (C != output_dim * group_size * group_size) return JDET_E_BADARG;
  p.N = N; p.C = C; p.H = H; p.W = W; p.R = R; p.no_trans = no_trans ? 1 : 0; p.output_dim = output_dim; p.group = group_size;
  p.P = pooled_size; p.part = part_size; p.spp = sample_per_part; p.scale = spatial_scale; p.trans_std = trans_std;
  // L951-952: num_classes = no_trans ? 1 : trans channels / 2; channels_each_class = output_dim / num_classes
  p.num_classes = no_trans ? 1 : trans_channels / 2;
  if (p.num_classes <= 0 || output_dim % p.num_classes != 0) return JDET_E_BADARG;
  p.ch_each_class = no_trans ? output_dim : output_dim / p.num_classes;
  return JDET_OK;
}

// four channels per lane (dwordx4) where a lane's channels are contiguous in memory and 16-byte aligned
int vec_of(const PsP& p) { return (p.group == 1 && p.ch_each_class % 4 == 0 && p.C % 4 == 0) ? 4 : 1; }

unsigned blocks_for(long count) {
  long g = (count + 255) / 256;
  return (unsigned)(g > 65536 ? 65536 : g);
}

}  // namespace

JDET_API int jdet_deform_psroi_pool_forward(const float* input, const float* rois, const float* trans, int N, int C,
                                            int H, int W, int R, int no_trans, float spatial_scale, int output_dim,
                                            int group_size, int pooled_size, int part_size, int sample_per_part,
                                            float trans_std, int trans_channels, float* out, float* top_count,
                                            jdet_stream_t stream) {
  PsP p;
  int e = fill(p, N, C, H, W, R, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
               sample_per_part, trans_std, trans_channels);
  if (e) return e;
  if (R == 0) return JDET_OK;
  if (!input || !rois || !out || !top_count || (!no_trans && !trans)) return JDET_E_BADARG;
  const int vec = vec_of(p);
  const int chunks = (p.ch_each_class + 64 * vec - 1) / (64 * vec);
  const long waves = (long)R * pooled_size * pooled_size * p.num_classes * chunks;
  if (vec == 4)
    hipLaunchKernelGGL(psroi_fwd_kernel<4>, dim3(blocks_for(waves * 64)), dim3(256), 0, (hipStream_t)stream, input, rois,
                       trans, p, chunks, waves, out, top_count);
  else
    hipLaunchKernelGGL(psroi_fwd_kernel<1>, dim3(blocks_for(waves * 64)), dim3(256), 0, (hipStream_t)stream, input, rois,
                       trans, p, chunks, waves, out, top_count);
  return jdet_launch_status();
}

JDET_API int jdet_deform_psroi_pool_backward(const float* grad_out, const float* top_count, const float* input,
                                             const float* rois, const float* trans, int N, int C, int H, int W, int R,
                                             int no_trans, float spatial_scale, int output_dim, int group_size,
                                             int pooled_size, int part_size, int sample_per_part, float trans_std,
                                             int trans_channels, float* grad_input, float* grad_trans,
                                             jdet_stream_t stream) {
  PsP p;
  int e = fill(p, N, C, H, W, R, no_trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
               sample_per_part, trans_std, trans_channels);
  if (e) return e;
  if (!grad_input && N > 0) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if ((e = jdet_zero_async(grad_input, sizeof(float) * (size_t)N * C * H * W, st))) return e;
  if (!no_trans && R > 0) {
    if (!grad_trans) return JDET_E_BADARG;
    if ((e = jdet_zero_async(grad_trans, sizeof(float) * (size_t)R * trans_channels * part_size * part_size, st)))
      return e;
  }
  if (R == 0 || N == 0) return JDET_OK;
  if (!grad_out || !top_count || !input || !rois || (!no_trans && !trans)) return JDET_E_BADARG;
  const int vec = vec_of(p);
  const int chunks = (p.ch_each_class + 64 * vec - 1) / (64 * vec);
  const long waves = (long)R * pooled_size * pooled_size * p.num_classes * chunks;
  if (vec == 4)
    hipLaunchKernelGGL(psroi_bwd_kernel<4>, dim3(blocks_for(waves * 64)), dim3(256), 0, st, grad_out, top_count, input,
                       rois, trans, p, chunks, waves, grad_input, grad_trans);
  else
    hipLaunchKernelGGL(psroi_bwd_kernel<1>, dim3(blocks_for(waves * 64)), dim3(256), 0, st, grad_out, top_count, input,
                       rois, trans, p, chunks, waves, grad_input, grad_trans);
  return jdet_launch_status();
}
