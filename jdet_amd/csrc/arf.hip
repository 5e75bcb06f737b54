// Active rotating filters (ORConv2d) for gfx950: a weight (nOut, nIn, nOri, kH, kW) is expanded to the nRot
// rotated copies (nOut * nRot, nIn * nOri, kH, kW) by a byte index table (ops/orn.py:L17-72: `indices[l][k]` = 1-based
// destination entry of source entry l under rotation k); the backward sums the gradient of the copies back.
// 73 728 weights at S2ANet's or_conv: a launch-latency-sized gather.  One thread per SOURCE entry, the nRot
// destinations of a thread form a permutation orbit, so neither direction needs atomics.
#include "common.h"

namespace {

struct ArfShape {
  int nIn, nEntry, nRot;   // nEntry = nOri * kH * kW
  int ks, cl;              // kH * kW; cl: the expanded filter bank in channels-last memory (Cout, kH, kW, Cin)
};

// destination element of source element (i, j, l) under rotation k
__device__ __forceinline__ size_t arf_dst(const ArfShape& s, const uint8_t* __restrict__ indices, int i, int j, int l,
                                          int k) {
  const int m = (int)indices[l * s.nRot + k] - 1;
  if (s.cl) {   // (co, h, w, ci) with ci = j * nOri + orientation plane of m
    const int lo = m / s.ks, hw = m - lo * s.ks, nOri = s.nEntry / s.ks;
    return ((size_t)(i * s.nRot + k) * s.ks + hw) * ((size_t)s.nIn * nOri) + (size_t)j * nOri + lo;
  }
  return ((size_t)(i * s.nRot + k) * s.nIn + j) * s.nEntry + m;
}

template <bool BACKWARD>
__global__ __launch_bounds__(256) void arf_kernel(long n, const float* __restrict__ src,
                                                  const uint8_t* __restrict__ indices, ArfShape s,
                                                  float* __restrict__ dst) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n; idx += (long)gridDim.x * 256) {
    const int l = (int)(idx % s.nEntry);
    const int j = (int)((idx / s.nEntry) % s.nIn);
    const int i = (int)(idx / s.nEntry / s.nIn);
    if (BACKWARD) {
      float acc = 0;
      for (int k = 0; k < s.nRot; k++) acc = acc + src[arf_dst(s, indices, i, j, l, k)];
      dst[idx] = acc;
    } else {
      const float v = src[idx];
      for (int k = 0; k < s.nRot; k++) dst[arf_dst(s, indices, i, j, l, k)] = v;
    }
  }
}

int arf_launch(bool backward, const float* src, const uint8_t* indices, int nOut, int nIn, int nOri, int kH, int kW,
               int nRot, float* dst, hipStream_t st, bool cl = false) {
  if (nOut < 0 || nIn < 0 || nOri <= 0 || kH <= 0 || kW <= 0 || nRot <= 0) return JDET_E_BADARG;
  const long n = (long)nOut * nIn * nOri * kH * kW;
  if (n == 0) return JDET_OK;
  if (!src || !indices || !dst) return JDET_E_BADARG;
  ArfShape s{nIn, nOri * kH * kW, nRot, kH * kW, cl ? 1 : 0};
  long g = (n + 255) / 256;
  if (g > 262144) g = 262144;
  if (backward)
    hipLaunchKernelGGL((arf_kernel<true>), dim3((unsigned)g), dim3(256), 0, st, n, src, indices, s, dst);
  else
    hipLaunchKernelGGL((arf_kernel<false>), dim3((unsigned)g), dim3(256), 0, st, n, src, indices, s, dst);
  return jdet_launch_status();
}

}  // namespace

JDET_API int jdet_arf_forward(const float* weight, const uint8_t* indices, int nOut, int nIn, int nOri,
                              int kH, int kW, int nRot, float* out, jdet_stream_t stream) {
  return arf_launch(false, weight, indices, nOut, nIn, nOri, kH, kW, nRot, out, (hipStream_t)stream);
}

JDET_API int jdet_arf_backward(const uint8_t* indices, const float* grad_out, int nOut, int nIn, int nOri,
                               int kH, int kW, int nRot, float* grad_weight, jdet_stream_t stream) {
  return arf_launch(true, grad_out, indices, nOut, nIn, nOri, kH, kW, nRot, grad_weight, (hipStream_t)stream);
}

// the same with the expanded filter bank (forward: out, backward: grad_out) in channels-last memory (Cout, kH, kW, Cin):
// what the channels-last convolution wants -- the library otherwise converts the bank on every call
JDET_API int jdet_arf_forward_cl(const float* weight, const uint8_t* indices, int nOut, int nIn, int nOri, int kH,
                                 int kW, int nRot, float* out_cl, jdet_stream_t stream) {
  return arf_launch(false, weight, indices, nOut, nIn, nOri, kH, kW, nRot, out_cl, (hipStream_t)stream, true);
}

JDET_API int jdet_arf_backward_cl(const uint8_t* indices, const float* grad_out_cl, int nOut, int nIn, int nOri,
                                  int kH, int kW, int nRot, float* grad_weight, jdet_stream_t stream) {
  return arf_launch(true, grad_out_cl, indices, nOut, nIn, nOri, kH, kW, nRot, grad_weight, (hipStream_t)stream, true);
}

// ---------------------------------------------------------------------------------------------------------------
// RotationInvariantPooling (orn.py:L595-618): y[p, g] = max over the nO orientation channels g*nO .. g*nO+nO-1 of a
// channels-last row; backward: the gradient goes to the channels that equal the maximum, split evenly among ties (what
// autograd derives for amax).  One thread per (position, group): nO consecutive floats in, one float out.
namespace {

template <int NO>
__global__ __launch_bounds__(256) void rip_fwd_kernel(const float* __restrict__ x, long n, float* __restrict__ y) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* src = reinterpret_cast<const v4f*>(x + t * NO);
  float m = -INFINITY;
  bool nan = false;                 // amax propagates NaN (fmaxf drops it): a diverged map must surface as NaN losses
#pragma unroll
  for (int i = 0; i < NO / 4; i++) {
    const v4f v = src[i];
    m = fmaxf(fmaxf(fmaxf(m, v.x), fmaxf(v.y, v.z)), v.w);
    nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
  }
  y[t] = nan ? __builtin_nanf("") : m;
}

template <int NO>
__global__ __launch_bounds__(256) void rip_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gy, long n, float* __restrict__ gx) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  typedef float v4f __attribute__((ext_vector_type(4)));
  const v4f* src = reinterpret_cast<const v4f*>(x + t * NO);
  v4f* dst = reinterpret_cast<v4f*>(gx + t * NO);
  const float m = y[t], g = gy[t];
  v4f v[NO / 4];
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < NO / 4; i++) {
    v[i] = src[i];
    cnt += (v[i].x == m) + (v[i].y == m) + (v[i].z == m) + (v[i].w == m);
  }
  const float share = g / (float)(cnt > 0 ? cnt : 1);
  if (m != m) {                     // NaN maximum: the gradient is NaN on the whole group, as amax's backward gives
#pragma unroll
    for (int i = 0; i < NO / 4; i++) dst[i] = v4f{m, m, m, m};
    return;
  }
#pragma unroll
  for (int i = 0; i < NO / 4; i++)
    dst[i] = v4f{v[i].x == m ? share : 0.f, v[i].y == m ? share : 0.f, v[i].z == m ? share : 0.f,
                 v[i].w == m ? share : 0.f};
}

}  // namespace

// x (P, C) channels-last rows with C = groups * nO, nO in {4, 8}; y (P, groups).  16-byte aligned x / gx.
JDET_API int jdet_rip_forward(const float* x_nhwc, long P, int C, int nO, float* y_nhwc, jdet_stream_t stream) {
  if (P < 0 || C <= 0 || (nO != 4 && nO != 8) || C % nO != 0) return JDET_E_UNSUPPORTED;
  if (P == 0) return JDET_OK;
  if (!x_nhwc || !y_nhwc || (((uintptr_t)x_nhwc) & 15)) return JDET_E_BADARG;
  const long n = P * (C / nO);
  const unsigned g = (unsigned)((n + 255) / 256);
  if (nO == 8) hipLaunchKernelGGL(rip_fwd_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_nhwc, n, y_nhwc);
  else hipLaunchKernelGGL(rip_fwd_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_nhwc, n, y_nhwc);
  return jdet_launch_status();
}

JDET_API int jdet_rip_backward(const float* x_nhwc, const float* y_nhwc, const float* grad_y_nhwc, long P, int C,
                               int nO, float* grad_x_nhwc, jdet_stream_t stream) {
  if (P < 0 || C <= 0 || (nO != 4 && nO != 8) || C % nO != 0) return JDET_E_UNSUPPORTED;
  if (P == 0) return JDET_OK;
  if (!x_nhwc || !y_nhwc || !grad_y_nhwc || !grad_x_nhwc || ((((uintptr_t)x_nhwc) | ((uintptr_t)grad_x_nhwc)) & 15))
    return JDET_E_BADARG;
  const long n = P * (C / nO);
  const unsigned g = (unsigned)((n + 255) / 256);
  if (nO == 8)
    hipLaunchKernelGGL(rip_bwd_kernel<8>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_nhwc, y_nhwc, grad_y_nhwc, n,
                       grad_x_nhwc);
  else
    hipLaunchKernelGGL(rip_bwd_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_nhwc, y_nhwc, grad_y_nhwc, n,
                       grad_x_nhwc);
  return jdet_launch_status();
}

JDET_API int jdet_version(void) { return 2; }
