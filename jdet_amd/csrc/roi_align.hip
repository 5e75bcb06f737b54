// RoIAlign family for gfx950 (MI355X): ROIAlignRotated, ROIAlignRotated_v1, RiRoIAlign and the
// horizontal ROIAlign v0/v1, forward and backward.
//
// Reference semantics (per output element, fp32):
//   python/jdet/ops/roi_align_rotated.py:L21-127 (fwd), L128-255 (bwd)
//   python/jdet/ops/roi_align_rotated_v1.py:L71-145, L193-298
//   python/jdet/ops/riroi_align.py:L70-163, L228-358
//   python/jdet/ops/roi_align.py:L13-204
// The reference launches one CUDA thread per output element (n,c,ph,pw): every thread
// recomputes sin/cos + bin geometry and gathers 16 scattered floats from an NCHW map.
//
// MI355X design (not a translation):
//   * feature map is NHWC, so one bilinear tap is ONE contiguous C-vector: a wave64 reads a
//     256-channel tap with a single global_load_dwordx4 (64 lanes x 16 B = 1 KiB).
//   * one workgroup (4 waves) per (RoI, 256-channel chunk).  Sample geometry (position,
//     4 weights, 4 pixel offsets) is computed ONCE per sample, lane-parallel (lane = sample),
//     and broadcast to the wave with v_readlane -> all control flow in the tap loop is
//     wave-uniform and the weights live in SGPRs.
//   * per-lane accumulation order is exactly the reference's (w1*lt + w2*rt + w3*lb + w4*rb,
//     summed iy-major, then / count) with FMA contraction off -> forward is bit-identical to
//     the CPU oracle.
//   * results are staged in LDS as [channel][bin] and written out as one contiguous,
//     float4-coalesced (C_chunk*PH*PW) block in the reference's (R,C,PH,PW) layout.
//   * backward: grad_out chunk staged in LDS, same sample broadcast, hardware
//     global_atomic_add_f32 into the NHWC gradient (lane-contiguous 256 B per instruction).
#include "common.h"

namespace {

constexpr int kBlock = 256;   // 4 waves
constexpr int kChunkC = 256;  // channels per workgroup

struct RoiGeom {
  int batch;
  float center_w, center_h;
  float start_w, start_h;
  float bin_h, bin_w;
  int grid_h, grid_w;
  float cosT, sinT;
  float count;
  float l_var, r_var;
  int ind;
};

template <int VARIANT>
__device__ __forceinline__ RoiGeom roi_geom(const float* __restrict__ roi, float spatial_scale,
                                            int sample_num, int PH, int PW, int nO, bool backward) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.l_var = 0.f;
  g.r_var = 1.f;
  g.ind = 0;
  float roi_width, roi_height;
  if (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) {
    float roi_start_w = roi[1] * spatial_scale;
    float roi_start_h = roi[2] * spatial_scale;
    if (VARIANT == JDET_ROI_HBB_V1) {
      float roi_end_w = (roi[3] + 1) * spatial_scale;
      float roi_end_h = (roi[4] + 1) * spatial_scale;
      roi_width = fmaxf(roi_end_w - roi_start_w, 0.f);
      roi_height = fmaxf(roi_end_h - roi_start_h, 0.f);
    } else {
      float roi_end_w = roi[3] * spatial_scale;
      float roi_end_h = roi[4] * spatial_scale;
      roi_width = fmaxf(roi_end_w - roi_start_w, 1.f);
      roi_height = fmaxf(roi_end_h - roi_start_h, 1.f);
    }
    g.start_w = roi_start_w;
    g.start_h = roi_start_h;
    g.center_w = g.center_h = 0.f;
    g.cosT = 1.f;
    g.sinT = 0.f;
  } else {
    g.center_w = roi[1] * spatial_scale;
    g.center_h = roi[2] * spatial_scale;
    if (VARIANT == JDET_ROI_ROTATED_V1) {
      g.center_w = roi[1] * spatial_scale - 0.5f;
      g.center_h = roi[2] * spatial_scale - 0.5f;
    }
    roi_width = roi[3] * spatial_scale;
    roi_height = roi[4] * spatial_scale;
    const float theta = roi[5];
    roi_width = fmaxf(roi_width, 1.f);
    roi_height = fmaxf(roi_height, 1.f);
    g.start_h = -roi_height / 2.0f;
    g.start_w = -roi_width / 2.0f;
    // once per RoI: double-precision trig rounded to fp32 (what the host-compiled reference
    // text does; CUDA's cosf agrees to <= 1 ulp)
    g.cosT = (float)cos((double)theta);
    g.sinT = (float)sin((double)theta);
    if (VARIANT == JDET_ROI_RIROI) {
      // riroi_align.py:L105-113, PI literal L8
      float ind_float = (float)((double)(theta * nO) / (2 * 3.141592653));
      int ind = (int)floor(ind_float);
      g.l_var = ind_float - (float)ind;
      g.r_var = (float)(1.0 - (double)g.l_var);
      g.ind = (ind + nO) % nO;
    }
  }
  g.bin_h = roi_height / (float)PH;
  g.bin_w = roi_width / (float)PW;
  g.grid_h = (sample_num > 0) ? sample_num : (int)ceilf(roi_height / PH);
  g.grid_w = (sample_num > 0) ? sample_num : (int)ceilf(roi_width / PW);
  int cnt = g.grid_h * g.grid_w;
  if (VARIANT == JDET_ROI_ROTATED_V1 && !backward) cnt = max(cnt, 1);
  g.count = (float)cnt;
  return g;
}

// One bilinear sample: 4 weights + 4 pixel offsets (y*W+x), valid flag.
struct Sample {
  float w1, w2, w3, w4;
  int o1, o2, o3, o4;
  int valid;
};

template <int VARIANT>
__device__ __forceinline__ Sample make_sample(const RoiGeom& g, int ph, int pw, int iy, int ix,
                                              int H, int W) {
  const float yy = g.start_h + ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
  const float xx = g.start_w + pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
  float x, y;
  if (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) {
    x = xx;
    y = yy;
  } else if (VARIANT == JDET_ROI_ROTATED_V1) {
    x = xx * g.cosT + yy * g.sinT + g.center_w;
    y = yy * g.cosT - xx * g.sinT + g.center_h;
  } else {
    x = xx * g.cosT - yy * g.sinT + g.center_w;
    y = xx * g.sinT + yy * g.cosT + g.center_h;
  }
  Sample s;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    s.w1 = s.w2 = s.w3 = s.w4 = 0.f;
    s.o1 = s.o2 = s.o3 = s.o4 = 0;
    s.valid = 0;
    return s;
  }
  if (VARIANT == JDET_ROI_ROTATED_V1) {
    if (y < 0) y = 0;
    if (x < 0) x = 0;
  } else {
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
  }
  int y_low = (int)y, x_low = (int)x, y_high, x_high;
  if (y_low >= H - 1) {
    y_high = y_low = H - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= W - 1) {
    x_high = x_low = W - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  const float ly = y - y_low;
  const float lx = x - x_low;
  const float hy = (float)(1. - (double)ly);  // reference: `1. - ly` in double
  const float hx = (float)(1. - (double)lx);
  s.w1 = hy * hx;
  s.w2 = hy * lx;
  s.w3 = ly * hx;
  s.w4 = ly * lx;
  s.o1 = y_low * W + x_low;
  s.o2 = y_low * W + x_high;
  s.o3 = y_high * W + x_low;
  s.o4 = y_high * W + x_high;
  s.valid = 1;
  return s;
}

__device__ __forceinline__ Sample bcast(const Sample& s, int src_lane) {
  Sample r;
  r.w1 = jdet_readlane_f(s.w1, src_lane);
  r.w2 = jdet_readlane_f(s.w2, src_lane);
  r.w3 = jdet_readlane_f(s.w3, src_lane);
  r.w4 = jdet_readlane_f(s.w4, src_lane);
  r.o1 = jdet_readlane_i(s.o1, src_lane);
  r.o2 = jdet_readlane_i(s.o2, src_lane);
  r.o3 = jdet_readlane_i(s.o3, src_lane);
  r.o4 = jdet_readlane_i(s.o4, src_lane);
  r.valid = jdet_readlane_i(s.valid, src_lane);
  return r;
}

// Channel ownership of a lane inside a 256-channel chunk.
//   CHMAP 0: lane owns 4 consecutive channels (one dwordx4 per tap)   -- forward, C % 4 == 0
//   CHMAP 1: lane owns channels lane + 64*k (four dword accesses, each instruction covers a
//            contiguous 256 B)                                         -- atomics, RiRoI, odd C
template <int CHMAP>
__device__ __forceinline__ int chan_of(int lane, int k) {
  return CHMAP == 0 ? lane * 4 + k : lane + 64 * k;
}

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
template <int VARIANT, int CHMAP>
__global__ __launch_bounds__(kBlock) void roi_align_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num, int nO) {
  extern __shared__ __attribute__((aligned(16))) float s_out[];  // [cc][nbins]
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  const int r = blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  const RoiGeom g = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH,
                                      PW, nO, false);
  const float* __restrict__ img = feat + (size_t)g.batch * H * W * C;

  // per-lane source channel indices
  int src0[4], src1[4];
  bool cval[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cl = chan_of<CHMAP>(lane, k);
    cval[k] = cl < cc;
    const int ch = c0 + (cval[k] ? cl : 0);
    if (VARIANT == JDET_ROI_RIROI) {
      const int c = ch / nO, o = ch % nO;
      const int ind_rot = (o - g.ind + nO) % nO;
      const int ind_rot_plus = (ind_rot + 1 + nO) % nO;
      src0[k] = c * nO + ind_rot;
      src1[k] = c * nO + ind_rot_plus;
    } else {
      src0[k] = ch;
      src1[k] = ch;
    }
  }

  const int spb = g.grid_h * g.grid_w;                  // samples per bin
  const int nb = (nbins - wave + 3) >> 2;               // bins of this wave: wave, wave+4, ...
  const int bpc = spb <= 64 ? (spb > 0 ? 64 / spb : 64) : 1;  // whole bins per 64-lane pass
  const int passes = spb <= 64 ? 1 : (spb + 63) / 64;

  // lane-parallel sample geometry: lane = one sample of this wave's bins.  passes == 1: a
  // 64-lane pass covers `bpc` whole bins (computed once per group); passes > 1 (adaptive grids
  // with > 64 samples per bin): one bin at a time, 64 samples per pass.
  auto lane_sample = [&](int kg, int pass) -> Sample {
    int my_kb, my_r;
    if (passes == 1) {
      my_kb = spb > 0 ? lane / spb : 0;
      my_r = spb > 0 ? lane % spb : 0;
    } else {
      my_kb = 0;
      my_r = pass * 64 + lane;
    }
    const int my_bin = wave + 4 * (kg + my_kb);
    const bool ok = my_kb < bpc && my_bin < nbins && my_r < spb && g.grid_w > 0;
    const int iy = ok ? my_r / g.grid_w : 0;
    const int ix = ok ? my_r % g.grid_w : 0;
    const int bb = ok ? my_bin : 0;
    return make_sample<VARIANT>(g, bb / PW, bb % PW, iy, ix, H, W);
  };

  for (int kg = 0; kg < nb; kg += bpc) {
    Sample mine = lane_sample(kg, 0);
    for (int kb = 0; kb < bpc && kg + kb < nb; kb++) {
      const int bin = wave + 4 * (kg + kb);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int pass = 0; pass < passes; pass++) {
        if (passes > 1) mine = lane_sample(kg, pass);
        const int lane0 = passes == 1 ? kb * spb : 0;
        const int ns = passes == 1 ? spb : min(64, spb - pass * 64);
        for (int j = 0; j < ns; j++) {
          const Sample s = bcast(mine, lane0 + j);
          if (!s.valid) continue;  // reference returns 0 for out-of-map samples
          if (CHMAP == 0 && VARIANT != JDET_ROI_RIROI) {
            if (cval[0]) {
              const float4 lt = *reinterpret_cast<const float4*>(img + (size_t)s.o1 * C + src0[0]);
              const float4 rt = *reinterpret_cast<const float4*>(img + (size_t)s.o2 * C + src0[0]);
              const float4 lb = *reinterpret_cast<const float4*>(img + (size_t)s.o3 * C + src0[0]);
              const float4 rb = *reinterpret_cast<const float4*>(img + (size_t)s.o4 * C + src0[0]);
              acc[0] += (s.w1 * lt.x + s.w2 * rt.x + s.w3 * lb.x + s.w4 * rb.x);
              acc[1] += (s.w1 * lt.y + s.w2 * rt.y + s.w3 * lb.y + s.w4 * rb.y);
              acc[2] += (s.w1 * lt.z + s.w2 * rt.z + s.w3 * lb.z + s.w4 * rb.z);
              acc[3] += (s.w1 * lt.w + s.w2 * rt.w + s.w3 * lb.w + s.w4 * rb.w);
            }
          } else {
#pragma unroll
            for (int k = 0; k < 4; k++) {
              if (!cval[k]) continue;
              const float lt = img[(size_t)s.o1 * C + src0[k]];
              const float rt = img[(size_t)s.o2 * C + src0[k]];
              const float lb = img[(size_t)s.o3 * C + src0[k]];
              const float rb = img[(size_t)s.o4 * C + src0[k]];
              const float val = (s.w1 * lt + s.w2 * rt + s.w3 * lb + s.w4 * rb);
              if (VARIANT == JDET_ROI_RIROI) {
                const float lt1 = img[(size_t)s.o1 * C + src1[k]];
                const float rt1 = img[(size_t)s.o2 * C + src1[k]];
                const float lb1 = img[(size_t)s.o3 * C + src1[k]];
                const float rb1 = img[(size_t)s.o4 * C + src1[k]];
                const float val_plus = (s.w1 * lt1 + s.w2 * rt1 + s.w3 * lb1 + s.w4 * rb1);
                acc[k] += g.r_var * val + g.l_var * val_plus;
              } else {
                acc[k] += val;
              }
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (cval[k]) s_out[chan_of<CHMAP>(lane, k) * nbins + bin] = acc[k] / g.count;
    }
  }
  __syncthreads();
  // coalesced write-out of the contiguous [cc][nbins] block
  float* __restrict__ dst = out + ((size_t)r * C + c0) * nbins;
  const int total = cc * nbins;
  if (((total & 3) == 0) && ((((size_t)r * C + c0) * nbins) & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s_out);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < (total >> 2); i += kBlock) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < total; i += kBlock) dst[i] = s_out[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Backward (feature gradient)
// ---------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(kBlock) void roi_align_bwd_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float* __restrict__ grad_in,
    int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num, int nO) {
  extern __shared__ __attribute__((aligned(16))) float s_g[];  // [cc][nbins]
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr int CHMAP = 1;
  const int r = blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  // stage grad_out[r, c0:c0+cc, :, :] (contiguous) into LDS
  {
    const float* __restrict__ src = grad_out + ((size_t)r * C + c0) * nbins;
    const int total = cc * nbins;
    if (((total & 3) == 0) && ((((size_t)r * C + c0) * nbins) & 3) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(s_g);
      for (int i = threadIdx.x; i < (total >> 2); i += kBlock) d4[i] = s4[i];
    } else {
      for (int i = threadIdx.x; i < total; i += kBlock) s_g[i] = src[i];
    }
  }
  __syncthreads();

  const RoiGeom g = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH,
                                      PW, nO, true);
  float* __restrict__ img = grad_in + (size_t)g.batch * H * W * C;

  int dst0[4], dst1[4];
  bool cval[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cl = chan_of<CHMAP>(lane, k);
    cval[k] = cl < cc;
    const int ch = c0 + (cval[k] ? cl : 0);
    if (VARIANT == JDET_ROI_RIROI) {
      const int c = ch / nO, o = ch % nO;
      const int ind_rot = (o - g.ind + nO) % nO;
      const int ind_rot_plus = (ind_rot + 1 + nO) % nO;
      dst0[k] = c * nO + ind_rot;
      dst1[k] = c * nO + ind_rot_plus;
    } else {
      dst0[k] = ch;
      dst1[k] = ch;
    }
  }

  const int spb = g.grid_h * g.grid_w;
  if (spb <= 0) return;  // count == 0: the reference divides by zero -> inf*0; nothing sane to add
  const int nb = (nbins - wave + 3) >> 2;
  const int bpc = spb <= 64 ? 64 / spb : 1;
  const int passes = spb <= 64 ? 1 : (spb + 63) / 64;

  auto lane_sample = [&](int kg, int pass) -> Sample {
    int my_kb, my_r;
    if (passes == 1) {
      my_kb = lane / spb;
      my_r = lane % spb;
    } else {
      my_kb = 0;
      my_r = pass * 64 + lane;
    }
    const int my_bin = wave + 4 * (kg + my_kb);
    const bool ok = my_kb < bpc && my_bin < nbins && my_r < spb;
    const int iy = ok ? my_r / g.grid_w : 0;
    const int ix = ok ? my_r % g.grid_w : 0;
    const int bb = ok ? my_bin : 0;
    Sample m = make_sample<VARIANT>(g, bb / PW, bb % PW, iy, ix, H, W);
    // fold 1/count into the weights once per sample (reference: top*w/count per element)
    m.w1 /= g.count;
    m.w2 /= g.count;
    m.w3 /= g.count;
    m.w4 /= g.count;
    return m;
  };

  for (int kg = 0; kg < nb; kg += bpc) {
    Sample mine = lane_sample(kg, 0);
    for (int kb = 0; kb < bpc && kg + kb < nb; kb++) {
      const int bin = wave + 4 * (kg + kb);
      float top[4];
#pragma unroll
      for (int k = 0; k < 4; k++) top[k] = cval[k] ? s_g[chan_of<CHMAP>(lane, k) * nbins + bin] : 0.f;
      for (int pass = 0; pass < passes; pass++) {
        if (passes > 1) mine = lane_sample(kg, pass);
        const int lane0 = passes == 1 ? kb * spb : 0;
        const int ns = passes == 1 ? spb : min(64, spb - pass * 64);
        for (int j = 0; j < ns; j++) {
          const Sample s = bcast(mine, lane0 + j);
          if (!s.valid) continue;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (!cval[k]) continue;
            const float g1 = top[k] * s.w1, g2 = top[k] * s.w2, g3 = top[k] * s.w3, g4 = top[k] * s.w4;
            if (VARIANT == JDET_ROI_RIROI) {
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst0[k], g1 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst0[k], g2 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst0[k], g3 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst0[k], g4 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst1[k], g1 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst1[k], g2 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst1[k], g3 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst1[k], g4 * g.l_var);
            } else {
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst0[k], g1);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst0[k], g2);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst0[k], g3);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst0[k], g4);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC tiled transposes: per image a (C, HW) <-> (HW, C) matrix transpose.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, int rows, int cols) {
  // x: (batch, rows, cols) -> y: (batch, cols, rows); 32x32 tiles, +1 pad (conflict-free)
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int rr = r0 + ty + i, ccol = c0 + tx;
    if (rr < rows && ccol < cols) tile[ty + i][tx] = x[base + (size_t)rr * cols + ccol];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int ccol = c0 + ty + i, rr = r0 + tx;
    if (rr < rows && ccol < cols) y[base + (size_t)ccol * rows + rr] = tile[tx][ty + i];
  }
}

int launch_transpose(const float* x, float* y, int batch, int rows, int cols, hipStream_t st) {
  if (batch == 0 || rows == 0 || cols == 0) return JDET_OK;
  dim3 grid(jdet_cdiv(cols, 32), jdet_cdiv(rows, 32), batch);
  if (grid.y > 65535 || grid.z > 65535) return JDET_E_UNSUPPORTED;
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, x, y, rows, cols);
  return jdet_launch_status();
}

template <int VARIANT>
int launch_fwd(const float* feat, const float* rois, float* out, int R, int C, int H, int W, int PH,
               int PW, float scale, int sample_num, int nO, hipStream_t st) {
  const int chunks = jdet_cdiv(C, kChunkC);
  const size_t lds = (size_t)min(C, kChunkC) * PH * PW * sizeof(float);
  dim3 grid(R, chunks);
  const bool vec = (C % 4 == 0) && VARIANT != JDET_ROI_RIROI;
  if (vec)
    hipLaunchKernelGGL((roi_align_fwd_kernel<VARIANT, 0>), grid, dim3(kBlock), lds, st, feat, rois,
                       out, C, H, W, PH, PW, scale, sample_num, nO);
  else
    hipLaunchKernelGGL((roi_align_fwd_kernel<VARIANT, 1>), grid, dim3(kBlock), lds, st, feat, rois,
                       out, C, H, W, PH, PW, scale, sample_num, nO);
  return jdet_launch_status();
}

template <int VARIANT>
int launch_bwd(const float* gout, const float* rois, float* gin, int R, int C, int H, int W, int PH,
               int PW, float scale, int sample_num, int nO, hipStream_t st) {
  const int chunks = jdet_cdiv(C, kChunkC);
  const size_t lds = (size_t)min(C, kChunkC) * PH * PW * sizeof(float);
  hipLaunchKernelGGL((roi_align_bwd_kernel<VARIANT>), dim3(R, chunks), dim3(kBlock), lds, st, gout,
                     rois, gin, C, H, W, PH, PW, scale, sample_num, nO);
  return jdet_launch_status();
}

int check_common(int variant, const void* a, const void* b, const void* c, int N, int C, int H,
                 int W, int R, int PH, int PW, int n_orient) {
  if (variant < 0 || variant > 4) return JDET_E_BADARG;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || PH <= 0 || PW <= 0) return JDET_E_BADARG;
  if (R > 0 && (!a || !b || !c)) return JDET_E_BADARG;
  if (PH * PW > 256) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && (n_orient <= 0 || C % n_orient != 0)) return JDET_E_BADARG;
  if ((long)H * W >= (1L << 30)) return JDET_E_UNSUPPORTED;
  return JDET_OK;
}

}  // namespace

JDET_API int jdet_nchw_to_nhwc(const float* x, int N, int C, int H, int W, float* y,
                               jdet_stream_t stream) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || ((long)N * C * H * W > 0 && (!x || !y))) return JDET_E_BADARG;
  return launch_transpose(x, y, N, C, H * W, (hipStream_t)stream);
}

JDET_API int jdet_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y,
                               jdet_stream_t stream) {
  if (N < 0 || C < 0 || H < 0 || W < 0 || ((long)N * C * H * W > 0 && (!x || !y))) return JDET_E_BADARG;
  return launch_transpose(x, y, N, H * W, C, (hipStream_t)stream);
}

JDET_API int jdet_roi_align_forward(int variant, const float* feat, int N, int C, int H, int W,
                                    const float* rois, int R, int PH, int PW, float spatial_scale,
                                    int sample_num, int n_orient, float* out, jdet_stream_t stream) {
  int e = check_common(variant, feat, rois, out, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (R == 0) return JDET_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_fwd<JDET_ROI_ROTATED>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    case JDET_ROI_ROTATED_V1:
      return launch_fwd<JDET_ROI_ROTATED_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    case JDET_ROI_RIROI:
      return launch_fwd<JDET_ROI_RIROI>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, st);
    case JDET_ROI_HBB_V0:
      return launch_fwd<JDET_ROI_HBB_V0>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    default:
      return launch_fwd<JDET_ROI_HBB_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
  }
}

JDET_API int jdet_roi_align_backward(int variant, const float* grad_out, const float* rois, int R,
                                     int N, int C, int H, int W, int PH, int PW, float spatial_scale,
                                     int sample_num, int n_orient, float* grad_in,
                                     jdet_stream_t stream) {
  if (!grad_in && (long)N * C * H * W > 0) return JDET_E_BADARG;
  int e = check_common(variant, grad_out, rois, grad_in, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  hipStream_t st = (hipStream_t)stream;
  hipError_t he = hipMemsetAsync(grad_in, 0, sizeof(float) * (size_t)N * C * H * W, st);
  if (he != hipSuccess) return (int)he;
  if (R == 0) return JDET_OK;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_bwd<JDET_ROI_ROTATED>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    case JDET_ROI_ROTATED_V1:
      return launch_bwd<JDET_ROI_ROTATED_V1>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    case JDET_ROI_RIROI:
      return launch_bwd<JDET_ROI_RIROI>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, st);
    case JDET_ROI_HBB_V0:
      return launch_bwd<JDET_ROI_HBB_V0>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
    default:
      return launch_bwd<JDET_ROI_HBB_V1>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, st);
  }
}
