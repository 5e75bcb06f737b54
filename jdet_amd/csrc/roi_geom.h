// Geometry shared by the RoIAlign forward / backward kernels (roi_align.hip, roi_align_bwd.hip):
// per-RoI scalar prologue and per-sample bilinear setup of the five dialects, with the reference's
// literal fp32/fp64 mix.  Reference lines: roi_align_rotated.py:L21-59, L70-118 (v1: _v1.py:L89-134;
// RiRoI: riroi_align.py:L105-113; horizontal: roi_align.py:L105-132).
#pragma once
#include "common.h"

namespace jdet_roi {

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

// RiRoIAlign's per-RoI orientation constants (riroi_align.py:L105-113, PI literal L8)
__device__ __forceinline__ void ri_params(float theta, int nO, int& ind, float& l_var, float& r_var) {
  const float ind_float = (float)((double)(theta * nO) / (2 * 3.141592653));
  const int fl = (int)floor(ind_float);
  l_var = ind_float - (float)fl;
  r_var = (float)(1.0 - (double)l_var);
  ind = (fl + nO) % nO;
}

template <int VARIANT, bool TRIG = true>
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
    if (TRIG) {
      g.cosT = (float)cos((double)theta);
      g.sinT = (float)sin((double)theta);
    } else {
      g.cosT = g.sinT = 0.f;   // caller fills them in (computed once per workgroup)
    }
    if (VARIANT == JDET_ROI_RIROI) ri_params(theta, nO, g.ind, g.l_var, g.r_var);
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

// RoI frame -> map coordinates (the three rotation / translation conventions of the dialects).
template <int VARIANT>
__device__ __forceinline__ void roi_xform(const RoiGeom& g, float xx, float yy, float& x, float& y) {
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
}

// Position part of one bilinear sample: the four corner coordinates and the lerp fractions.
struct SamplePos {
  int y_low, x_low, y_high, x_high;
  float ly, lx;
  int valid;
};

template <int VARIANT>
__device__ __forceinline__ SamplePos sample_pos(const RoiGeom& g, int ph, int pw, int iy, int ix,
                                                int H, int W) {
  const float yy = g.start_h + ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
  const float xx = g.start_w + pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
  float x, y;
  roi_xform<VARIANT>(g, xx, yy, x, y);
  SamplePos p;
  if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) {
    p.y_low = p.x_low = p.y_high = p.x_high = 0;
    p.ly = p.lx = 0.f;
    p.valid = 0;
    return p;
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
  p.y_low = y_low;
  p.x_low = x_low;
  p.y_high = y_high;
  p.x_high = x_high;
  p.ly = y - y_low;
  p.lx = x - x_low;
  p.valid = 1;
  return p;
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
  const SamplePos p = sample_pos<VARIANT>(g, ph, pw, iy, ix, H, W);
  Sample s;
  if (!p.valid) {
    s.w1 = s.w2 = s.w3 = s.w4 = 0.f;
    s.o1 = s.o2 = s.o3 = s.o4 = 0;
    s.valid = 0;
    return s;
  }
  const float hy = (float)(1. - (double)p.ly);  // reference: `1. - ly` in double
  const float hx = (float)(1. - (double)p.lx);
  s.w1 = hy * hx;
  s.w2 = hy * p.lx;
  s.w3 = p.ly * hx;
  s.w4 = p.ly * p.lx;
  s.o1 = p.y_low * W + p.x_low;
  s.o2 = p.y_low * W + p.x_high;
  s.o3 = p.y_high * W + p.x_low;
  s.o4 = p.y_high * W + p.x_high;
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


}  // namespace jdet_roi
