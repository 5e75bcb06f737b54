// Inference-mode BatchNorm fused with its neighbours, channels-last (NHWC).
//
// The reference trains its detectors with every backbone BatchNorm in eval mode (resnet.py:L177-185:
// `norm_eval`), affine parameters still trainable outside the frozen stages.  A bottleneck is therefore
//   conv -> bn -> relu,  conv -> bn -> relu,  conv -> bn -> (+identity) -> relu      (resnet.py:L61-93)
// with bn(x) = (x - mean) * rsqrt(var + eps) * w + b a per-channel affine map.  Framework ops run this as
// bn / relu / add kernels forward and relu-backward / per-channel reduce / scale kernels backward: 11 tensor
// passes per layer, the reduction at ~1.3 TB/s (profiles/r01_s2anet_train_steady_state_kernels.txt, 4.3 ms
// of a 41.6 ms S2ANet step).  Here:
//   forward   y = act(x * a + sh (+ res))                              1 read (+1), 1 write
//   backward  g = dy * [y > 0];  dx = g * a;  (dres = g);  dbeta = sum g;  dgamma = sum g * xhat
//             one pass: reads dy, y, x; writes dx (, g); the two per-channel sums accumulate in registers
//             (a thread keeps the same 4 channels for its whole grid-stride loop), are combined per
//             workgroup through LDS and finished by a second tiny launch -- deterministic, no atomics.
// Tensors are [P][C] with P = N*H*W, C % 4 == 0, every access a float4.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

struct BnParams {
  const float* w;
  const float* b;
  const float* mean;
  const float* var;
  float eps;
};

// a = w * rsqrt(var + eps) and sh = b - mean * a of the 4 channels starting at c
__device__ __forceinline__ void affine4(const BnParams& p, int c, v4f& a, v4f& sh, v4f& mean, v4f& invstd) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const float is = 1.0f / sqrtf(p.var[c + k] + p.eps);
    const float w = p.w ? p.w[c + k] : 1.f;
    const float b = p.b ? p.b[c + k] : 0.f;
    invstd[k] = is;
    mean[k] = p.mean[c + k];
    a[k] = w * is;
    sh[k] = b - p.mean[c + k] * (w * is);
  }
}

// launch geometry: blockDim = max(256, C/4) (<= 1024), gridDim*blockDim is a multiple of C/4, so the channel
// quad of a thread, (global thread id) % (C/4), is the same in every iteration of its grid-stride loop
template <bool RELU, bool RES>
__global__ __launch_bounds__(1024) void frozen_bn_fwd_kernel(const v4f* __restrict__ x, const v4f* __restrict__ res,
                                                             v4f* __restrict__ y, BnParams p, int cpt, size_t n4) {
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4f a, sh, mean, invstd;
  affine4(p, (int)(g % cpt) * 4, a, sh, mean, invstd);
  size_t i = g;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // 4 independent loads in flight per lane
    v4f v[4], r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = x[i + u * stride];
    if (RES) {
#pragma unroll
      for (int u = 0; u < 4; u++) r[u] = res[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v4f o = v[u] * a + sh;
      if (RES) o += r[u];
      if (RELU) o = __builtin_elementwise_max(o, (v4f){0.f, 0.f, 0.f, 0.f});
      y[i + u * stride] = o;
    }
  }
  for (; i < n4; i += stride) {
    v4f o = x[i] * a + sh;
    if (RES) o += res[i];
    if (RELU) o = __builtin_elementwise_max(o, (v4f){0.f, 0.f, 0.f, 0.f});
    y[i] = o;
  }
}

template <bool RELU, bool RES, bool AFFINE>
__global__ __launch_bounds__(1024) void frozen_bn_bwd_kernel(const v4f* __restrict__ dy, const v4f* __restrict__ y,
                                                             const v4f* __restrict__ x, v4f* __restrict__ dx,
                                                             v4f* __restrict__ dres, BnParams p, int cpt, size_t n4,
                                                             float* __restrict__ partial) {
  extern __shared__ float s_red[];   // [blockDim][8] when AFFINE
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int cq = (int)(g % cpt);
  v4f a, sh, mean, invstd;
  affine4(p, cq * 4, a, sh, mean, invstd);
  v4f sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](size_t i, const v4f& d, const v4f& yy, const v4f& xx) {
    v4f gq = d;
    if (RELU) {
#pragma unroll
      for (int k = 0; k < 4; k++) gq[k] = yy[k] > 0.f ? d[k] : 0.f;
    }
    dx[i] = gq * a;
    if (RES) dres[i] = gq;
    if (AFFINE) {
      sg += gq;
      sgx += gq * ((xx - mean) * invstd);
    }
  };
  size_t i = g;
  for (; i + stride < n4; i += 2 * stride) {
    v4f d[2], yy[2], xx[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      d[u] = dy[i + u * stride];
      if (RELU) yy[u] = y[i + u * stride];
      if (AFFINE) xx[u] = x[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) one(i + u * stride, d[u], yy[u], xx[u]);
  }
  for (; i < n4; i += stride) {
    v4f d = dy[i], yy = d, xx = d;
    if (RELU) yy = y[i];
    if (AFFINE) xx = x[i];
    one(i, d, yy, xx);
  }
  if (AFFINE) {
    float* mine = s_red + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      mine[k] = sg[k];
      mine[4 + k] = sgx[k];
    }
    __syncthreads();
    // threads 0..cpt-1 own one channel quad each; the others of the workgroup with the same quad sit at
    // threadIdx + m*cpt (blockDim is a multiple of cpt)
    if ((int)threadIdx.x < cpt) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int t = threadIdx.x; t < (int)blockDim.x; t += cpt)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += s_red[(size_t)t * 8 + k];
      // quad of thread t in this workgroup: (blockIdx*blockDim + t) % cpt == t because blockDim % cpt == 0
      float* out = partial + (size_t)blockIdx.x * 2 * cpt * 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        out[threadIdx.x * 4 + k] = acc[k];                 // dbeta partials
        out[cpt * 4 + threadIdx.x * 4 + k] = acc[4 + k];   // dgamma partials
      }
    }
  }
}

// Backward of y = relu(bn(c) + idn) FROM ITS OUTPUT: the fused convolution (conv_bn.hip) never stores c, and wherever
// y > 0 the normalised input is recoverable as xhat = (y - idn - beta) / gamma.  One pass: reads dy, y (, idn); writes
// dc = dy * [y > 0] * a; per-channel partial sums of g = dy * [y > 0] and g * (y - idn - beta) -- the second stage
// divides the latter by gamma (jdet_bn_sums_finish).  Same launch geometry / partial layout as frozen_bn_bwd_kernel.
// SRC selects the second sum's factor: 0: y - beta (a plain conv -> bn -> relu layer), 1: y - v - beta (v = the identity
// added before the ReLU), 2: u - beta (u = this BatchNorm's own output, e.g. the downsample branch bn_d(c_d) that was
// added into y: its xhat is (u - beta) / gamma everywhere, the mask still comes from y).
template <int SRC, bool AFFINE>
__global__ __launch_bounds__(1024) void bn_out_bwd_kernel(const v4f* __restrict__ dy, const v4f* __restrict__ y,
                                                          const v4f* __restrict__ uv, v4f* __restrict__ dc,
                                                          BnParams p, int cpt, size_t n4, float* __restrict__ partial) {
  extern __shared__ float s_red[];   // [blockDim][8] when AFFINE
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const int cq = (int)(g % cpt);
  v4f a, sh, mean, invstd, beta;
  affine4(p, cq * 4, a, sh, mean, invstd);
#pragma unroll
  for (int k = 0; k < 4; k++) beta[k] = p.b ? p.b[cq * 4 + k] : 0.f;
  v4f sg = {0.f, 0.f, 0.f, 0.f}, sgx = {0.f, 0.f, 0.f, 0.f};
  auto one = [&](size_t i, const v4f& d, const v4f& yy, const v4f& rr) {
    v4f gq;
#pragma unroll
    for (int k = 0; k < 4; k++) gq[k] = yy[k] > 0.f ? d[k] : 0.f;
    dc[i] = gq * a;
    if (AFFINE) {
      sg += gq;
      sgx += gq * (SRC == 1 ? (yy - rr) - beta : (SRC == 2 ? rr - beta : yy - beta));
    }
  };
  size_t i = g;
  for (; i + stride < n4; i += 2 * stride) {
    v4f d[2], yy[2], rr[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      d[u] = dy[i + u * stride];
      yy[u] = y[i + u * stride];
      if (SRC != 0 && AFFINE) rr[u] = uv[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) one(i + u * stride, d[u], yy[u], rr[u]);
  }
  for (; i < n4; i += stride) {
    v4f d = dy[i], yy = y[i], rr = d;
    if (SRC != 0 && AFFINE) rr = uv[i];
    one(i, d, yy, rr);
  }
  if (AFFINE) {
    float* mine = s_red + (size_t)threadIdx.x * 8;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      mine[k] = sg[k];
      mine[4 + k] = sgx[k];
    }
    __syncthreads();
    if ((int)threadIdx.x < cpt) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int t = threadIdx.x; t < (int)blockDim.x; t += cpt)
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += s_red[(size_t)t * 8 + k];
      float* out = partial + (size_t)blockIdx.x * 2 * cpt * 4;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        out[threadIdx.x * 4 + k] = acc[k];
        out[cpt * 4 + threadIdx.x * 4 + k] = acc[4 + k];
      }
    }
  }
}

// Second stage for up to 4 BatchNorm layers in ONE launch (a bottleneck's three or four): job j sums its partial rows
// [rows][2][C] in a fixed order; grad_beta = sum of the first halves, grad_gamma = (sum of the second halves) / gamma.
// gamma == 0 leaves xhat unrecoverable from the activation: the quotient is then inf / NaN on purpose (a silent 0 would
// be a wrong gradient), see DESIGN.md 3.6.
struct SumsJobs {
  jdet_bn_sums_job_t job[4];
  int first_block[5];
  int njobs;
};

__global__ __launch_bounds__(1024) void bn_sums_finish_kernel(SumsJobs js) {
  __shared__ float s_b[32][33];
  __shared__ float s_g[32][33];
  int j = 0;
  while (j + 1 < js.njobs && (int)blockIdx.x >= js.first_block[j + 1]) j++;
  const jdet_bn_sums_job_t job = js.job[j];
  const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = ((int)blockIdx.x - js.first_block[j]) * 32 + lane;
  const int C = job.C;
  float acc = 0.f, accg = 0.f;
  if (c < C) {
    long b = rg;
    for (; b + 96 < job.rows; b += 128) {          // 4 rows in flight per thread
      const float v0 = job.partial[(size_t)b * 2 * C + c], v1 = job.partial[(size_t)(b + 32) * 2 * C + c];
      const float v2 = job.partial[(size_t)(b + 64) * 2 * C + c], v3 = job.partial[(size_t)(b + 96) * 2 * C + c];
      const float w0 = job.partial[(size_t)b * 2 * C + C + c], w1 = job.partial[(size_t)(b + 32) * 2 * C + C + c];
      const float w2 = job.partial[(size_t)(b + 64) * 2 * C + C + c], w3 = job.partial[(size_t)(b + 96) * 2 * C + C + c];
      acc += (v0 + v1) + (v2 + v3);
      accg += (w0 + w1) + (w2 + w3);
    }
    for (; b < job.rows; b += 32) {
      acc += job.partial[(size_t)b * 2 * C + c];
      accg += job.partial[(size_t)b * 2 * C + C + c];
    }
  }
  s_b[rg][lane] = acc;
  s_g[rg][lane] = accg;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f, tg = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r++) {
      t += s_b[r][lane];
      tg += s_g[r][lane];
    }
    if (job.grad_beta) job.grad_beta[c] = t;
    if (job.grad_gamma) job.grad_gamma[c] = job.gamma ? tg / job.gamma[c] : tg;
  }
}

// The same pass for a convolution's bias: g = dy * [y > 0] (when the conv is followed by a ReLU), dbias = sum g.
// Replaces the framework's threshold_backward + per-channel reduce pair (two kernels, three tensor passes, the reduce
// at ~1.7 TB/s) behind every conv + bias [+ ReLU] of the heads: one read of dy (and y), one write of g.
template <bool RELU>
__global__ __launch_bounds__(1024) void bias_act_bwd_kernel(const v4f* __restrict__ dy, const v4f* __restrict__ y,
                                                            v4f* __restrict__ dx, int cpt, size_t n4,
                                                            float* __restrict__ partial) {
  extern __shared__ float s_red[];   // [blockDim][4]
  const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  v4f sg = {0.f, 0.f, 0.f, 0.f};
  size_t i = g;
  for (; i + stride < n4; i += 2 * stride) {
    v4f d[2], yy[2];
#pragma unroll
    for (int u = 0; u < 2; u++) {
      d[u] = dy[i + u * stride];
      if (RELU) yy[u] = y[i + u * stride];
    }
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (RELU) {
#pragma unroll
        for (int k = 0; k < 4; k++) d[u][k] = yy[u][k] > 0.f ? d[u][k] : 0.f;
        dx[i + u * stride] = d[u];
      }
      sg += d[u];
    }
  }
  for (; i < n4; i += stride) {
    v4f d = dy[i];
    if (RELU) {
      const v4f yy = y[i];
#pragma unroll
      for (int k = 0; k < 4; k++) d[k] = yy[k] > 0.f ? d[k] : 0.f;
      dx[i] = d;
    }
    sg += d;
  }
  float* mine = s_red + (size_t)threadIdx.x * 4;
#pragma unroll
  for (int k = 0; k < 4; k++) mine[k] = sg[k];
  __syncthreads();
  if ((int)threadIdx.x < cpt) {     // threads t, t + cpt, ... of the workgroup hold the same channel quad
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = threadIdx.x; t < (int)blockDim.x; t += cpt)
#pragma unroll
      for (int k = 0; k < 4; k++) acc[k] += s_red[(size_t)t * 4 + k];
    float* out = partial + (size_t)blockIdx.x * 2 * cpt * 4;      // row layout of sums_finish_kernel (dbeta half)
#pragma unroll
    for (int k = 0; k < 4; k++) out[threadIdx.x * 4 + k] = acc[k];
  }
}

// Per-channel sum of channels-last rows (P, C) for ANY channel count up to 256 (the classification / regression
// output convs of the heads: 15 and 5 channels): the kernels above want C % 4 == 0 with C / 4 dividing 256, and the
// framework's per-channel reduce those layers fell back to runs 25-60 us per call (12 calls per S2ANet step).  A
// workgroup has C * floor(256 / C) threads and the grid-stride is a multiple of C, so a thread keeps ONE channel and
// the loads of a wave are consecutive floats; fixed-order LDS combine; partial rows in sums_finish_kernel's layout.
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, size_t n, int C,
                                                          float* __restrict__ partial) {
  __shared__ float s_red[256];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {       // 4 independent loads in flight per lane
    a0 += x[i];
    a1 += x[i + stride];
    a2 += x[i + 2 * stride];
    a3 += x[i + 3 * stride];
  }
  for (; i < n; i += stride) a0 += x[i];
  s_red[threadIdx.x] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if ((int)threadIdx.x < C) {       // threads t, t + C, ... hold channel t (blockDim % C == 0)
    float acc = 0.f;
    for (int t = threadIdx.x; t < (int)blockDim.x; t += C) acc += s_red[t];
    partial[(size_t)blockIdx.x * 2 * C + threadIdx.x] = acc;
  }
}

// Second stage of the per-channel sums (BN: dbeta + dgamma; conv bias: dbeta only): workgroup = 32 channels x 32 row
// groups, every partial row of a thread in flight at once (nblocks <= 256 -> 8 rows per thread), fixed-order LDS
// combine (deterministic).  History: the first version (256 threads, 8 row groups, 512 partial rows) took 14.5 us per
// BN layer -- 74 launches = 1.07 ms of the S2ANet step (profiles/r03_s2anet_train_steady_state_kernels.txt) -- behind
// an 8-workgroup grid on a 256-CU chip; the first stages now write at most 256 rows from 1024-thread workgroups.
template <bool GAMMA>
__global__ __launch_bounds__(1024) void sums_finish_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta) {
  __shared__ float s_b[32][33];
  __shared__ float s_g[GAMMA ? 32 : 1][33];
  const int lane = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + lane;
  float v[8], w[8];
#pragma unroll
  for (int u = 0; u < 8; u++) {
    const int b = rg + 32 * u;
    const bool ok = c < C && b < nblocks;
    v[u] = ok ? partial[(size_t)b * 2 * C + c] : 0.f;
    w[u] = (GAMMA && ok) ? partial[(size_t)b * 2 * C + C + c] : 0.f;
  }
  float acc = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  float accg = ((w[0] + w[1]) + (w[2] + w[3])) + ((w[4] + w[5]) + (w[6] + w[7]));
  for (int b = rg + 256; b < nblocks; b += 32) {
    acc += (c < C) ? partial[(size_t)b * 2 * C + c] : 0.f;
    if (GAMMA) accg += (c < C) ? partial[(size_t)b * 2 * C + C + c] : 0.f;
  }
  s_b[rg][lane] = acc;
  if (GAMMA) s_g[rg][lane] = accg;
  __syncthreads();
  if (rg == 0 && c < C) {
    float t = 0.f, tg = 0.f;
#pragma unroll
    for (int r = 0; r < 32; r++) {
      t += s_b[r][lane];
      if (GAMMA) tg += s_g[r][lane];
    }
    if (dbeta) dbeta[c] = t;
    if (GAMMA && dgamma) dgamma[c] = tg;
  }
}

struct Geo {
  int cpt, block, grid;
};

int geometry(long P, int C, int max_grid, Geo& g) {
  if (P < 0 || C <= 0) return JDET_E_BADARG;
  if (C % 4 != 0) return JDET_E_UNSUPPORTED;
  g.cpt = C / 4;
  if (g.cpt > 1024) return JDET_E_UNSUPPORTED;
  // blockDim: multiple of cpt when cpt <= 256 needs 256 % cpt == 0 (power-of-two channel counts); otherwise one
  // workgroup = one pixel row of quads
  if (g.cpt <= 256) {
    if (256 % g.cpt != 0) return JDET_E_UNSUPPORTED;
    g.block = 256;
  } else {
    g.block = g.cpt;
  }
  const size_t n4 = (size_t)P * g.cpt;
  long want = (long)((n4 + (size_t)g.block * 4 - 1) / ((size_t)g.block * 4));
  if (want < 1) want = 1;
  g.grid = (int)(want > max_grid ? max_grid : want);
  return JDET_OK;
}

constexpr int kFwdGrid = 4096;
constexpr int kBwdGrid = 512;    // rows of the partial-sum workspace

}  // namespace

JDET_API int jdet_frozen_bn_act_forward(const float* x_nhwc, const float* residual_nhwc, long P, int C,
                                        const float* weight, const float* bias, const float* running_mean,
                                        const float* running_var, float eps, int relu, float* y_nhwc,
                                        jdet_stream_t stream) {
  Geo g;
  int e = geometry(P, C, kFwdGrid, g);
  if (e) return e;
  if (P == 0) return JDET_OK;
  if (!x_nhwc || !y_nhwc || !running_mean || !running_var) return JDET_E_BADARG;
  BnParams p{weight, bias, running_mean, running_var, eps};
  const size_t n4 = (size_t)P * g.cpt;
  hipStream_t st = (hipStream_t)stream;
  const v4f* x = (const v4f*)x_nhwc;
  const v4f* r = (const v4f*)residual_nhwc;
  v4f* y = (v4f*)y_nhwc;
#define JDET_BN_FWD(RELU, RES) \
  hipLaunchKernelGGL((frozen_bn_fwd_kernel<RELU, RES>), dim3(g.grid), dim3(g.block), 0, st, x, r, y, p, g.cpt, n4)
  if (relu && r) JDET_BN_FWD(true, true);
  else if (relu) JDET_BN_FWD(true, false);
  else if (r) JDET_BN_FWD(false, true);
  else JDET_BN_FWD(false, false);
#undef JDET_BN_FWD
  return jdet_launch_status();
}

JDET_API size_t jdet_frozen_bn_act_backward_workspace(long P, int C) {
  Geo g;
  if (geometry(P, C, kBwdGrid, g) || P == 0) return 0;
  return sizeof(float) * (size_t)g.grid * 2 * C;
}

JDET_API int jdet_frozen_bn_act_backward(const float* grad_y_nhwc, const float* y_nhwc, const float* x_nhwc, long P,
                                         int C, const float* weight, const float* bias, const float* running_mean,
                                         const float* running_var, float eps, int relu, float* grad_x_nhwc,
                                         float* grad_residual_nhwc, float* grad_weight, float* grad_bias,
                                         void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  Geo g;
  int e = geometry(P, C, kBwdGrid, g);
  if (e) return e;
  if (P == 0) return JDET_OK;
  const bool affine = grad_weight != nullptr || grad_bias != nullptr;
  if (!grad_y_nhwc || !grad_x_nhwc || !running_mean || !running_var || (relu && !y_nhwc) || (affine && !x_nhwc))
    return JDET_E_BADARG;
  if (affine && (!workspace || workspace_bytes < jdet_frozen_bn_act_backward_workspace(P, C))) return JDET_E_WORKSPACE;
  BnParams p{weight, bias, running_mean, running_var, eps};
  const size_t n4 = (size_t)P * g.cpt;
  hipStream_t st = (hipStream_t)stream;
  const v4f* dy = (const v4f*)grad_y_nhwc;
  const v4f* y = (const v4f*)y_nhwc;
  const v4f* x = (const v4f*)x_nhwc;
  v4f* dx = (v4f*)grad_x_nhwc;
  v4f* dr = (v4f*)grad_residual_nhwc;
  float* part = (float*)workspace;
  if (affine && 1024 % g.cpt == 0) {
    // affine pass: 1024-thread workgroups (a multiple of the channel-quad count, as the LDS combine needs) and at
    // most 256 of them, so that the second stage has every partial row of a thread in flight at once
    g.block = 1024;
    long want = (long)((n4 + (size_t)g.block * 4 - 1) / ((size_t)g.block * 4));
    g.grid = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  }
  const size_t lds = affine ? sizeof(float) * 8 * (size_t)g.block : 0;
#define JDET_BN_BWD(RELU, RES, AFF)                                                                            \
  hipLaunchKernelGGL((frozen_bn_bwd_kernel<RELU, RES, AFF>), dim3(g.grid), dim3(g.block), lds, st, dy, y, x, dx, dr, \
                     p, g.cpt, n4, part)
  const int key = (relu ? 4 : 0) | (dr ? 2 : 0) | (affine ? 1 : 0);
  switch (key) {
    case 0: JDET_BN_BWD(false, false, false); break;
    case 1: JDET_BN_BWD(false, false, true); break;
    case 2: JDET_BN_BWD(false, true, false); break;
    case 3: JDET_BN_BWD(false, true, true); break;
    case 4: JDET_BN_BWD(true, false, false); break;
    case 5: JDET_BN_BWD(true, false, true); break;
    case 6: JDET_BN_BWD(true, true, false); break;
    default: JDET_BN_BWD(true, true, true); break;
  }
#undef JDET_BN_BWD
  e = jdet_launch_status();
  if (e || !affine) return e;
  hipLaunchKernelGGL((sums_finish_kernel<true>), dim3((C + 31) / 32), dim3(1024), 0, st, part, g.grid, C, grad_weight,
                     grad_bias);
  return jdet_launch_status();
}

/* conv bias [+ ReLU] backward: grad_pre = grad_y * [y > 0] (relu != 0; grad_pre may be NULL without ReLU: the gradient
 * passes unchanged) and grad_bias = sum over positions.  Same workspace query as the BN backward. */
JDET_API int jdet_bias_act_backward(const float* grad_y_nhwc, const float* y_nhwc, long P, int C, int relu,
                                    float* grad_pre_nhwc, float* grad_bias, void* workspace, size_t workspace_bytes,
                                    jdet_stream_t stream) {
  Geo g;
  int e = geometry(P, C, kBwdGrid, g);
  if (e) return e;
  if (!grad_bias) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) return jdet_zero_async(grad_bias, sizeof(float) * (size_t)C, st);
  if (!grad_y_nhwc || (relu && (!y_nhwc || !grad_pre_nhwc))) return JDET_E_BADARG;
  if (!workspace || workspace_bytes < jdet_frozen_bn_act_backward_workspace(P, C)) return JDET_E_WORKSPACE;
  // 1024-thread workgroups (a multiple of the channel-quad count, as the LDS combine needs) and at most 256 of them:
  // a quarter of the partial rows of the BN kernels' geometry for the second stage to read
  if (1024 % g.cpt == 0) g.block = 1024;
  const size_t n4 = (size_t)P * g.cpt;
  long want = (long)((n4 + (size_t)g.block * 4 - 1) / ((size_t)g.block * 4));
  g.grid = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  const size_t lds = sizeof(float) * 4 * (size_t)g.block;
  float* part = (float*)workspace;
  if (relu)
    hipLaunchKernelGGL((bias_act_bwd_kernel<true>), dim3(g.grid), dim3(g.block), lds, st, (const v4f*)grad_y_nhwc,
                       (const v4f*)y_nhwc, (v4f*)grad_pre_nhwc, g.cpt, n4, part);
  else
    hipLaunchKernelGGL((bias_act_bwd_kernel<false>), dim3(g.grid), dim3(g.block), lds, st, (const v4f*)grad_y_nhwc,
                       (const v4f*)nullptr, (v4f*)nullptr, g.cpt, n4, part);
  if ((e = jdet_launch_status())) return e;
  hipLaunchKernelGGL((sums_finish_kernel<false>), dim3((C + 31) / 32), dim3(1024), 0, st, part, g.grid, C,
                     (float*)nullptr, grad_bias);
  return jdet_launch_status();
}

namespace {
// launch geometry of the output-based backward with sums: 1024-thread workgroups where the channel-quad count divides
// 1024 (as the LDS combine needs), at most 256 of them
int out_bwd_geometry(long P, int C, bool affine, Geo& g) {
  int e = geometry(P, C, kBwdGrid, g);
  if (e) return e;
  if (affine && 1024 % g.cpt == 0) {
    const size_t n4 = (size_t)P * g.cpt;
    g.block = 1024;
    long want = (long)((n4 + (size_t)g.block * 4 - 1) / ((size_t)g.block * 4));
    g.grid = (int)(want < 1 ? 1 : (want > 256 ? 256 : want));
  }
  return JDET_OK;
}
}  // namespace

/* rows of partial sums ([rows][2][C] floats) jdet_bn_act_backward_from_output writes for a (P, C) tensor; 0 = unsupported */
JDET_API size_t jdet_bn_act_backward_from_output_rows(long P, int C) {
  Geo g;
  if (out_bwd_geometry(P, C, true, g) || P == 0) return 0;
  return (size_t)g.grid;
}

/* Backward of a BatchNorm whose (possibly summed) output went through a ReLU, from the ACTIVATION y (conv_bn.hip's fused
 * forward stores no conv output): grad_c = grad_y * [y > 0] * a, and -- sums non-NULL -- partial per-channel sums
 * [rows][2][C] of g = grad_y * [y > 0] and g * t, rows = jdet_bn_act_backward_from_output_rows(P, C), to be finished by
 * jdet_bn_sums_finish (which divides the second sum by gamma: t / gamma = xhat).  t is
 *   own_output == NULL, identity == NULL :  y - beta                 y = relu(bn(c))
 *   identity != NULL                     :  y - identity - beta      y = relu(bn(c) + identity)
 *   own_output != NULL                   :  own_output - beta        y = relu(other + bn(c)), own_output = bn(c) */
JDET_API int jdet_bn_act_backward_from_output(const float* grad_y_nhwc, const float* y_nhwc, const float* identity_nhwc,
                                              const float* own_output_nhwc, long P, int C, const float* weight,
                                              const float* bias, const float* running_mean, const float* running_var,
                                              float eps, float* grad_c_nhwc, float* sums, size_t sums_bytes,
                                              jdet_stream_t stream) {
  Geo g;
  const bool affine = sums != nullptr;
  int e = out_bwd_geometry(P, C, affine, g);
  if (e) return e;
  if (P == 0) return JDET_OK;
  if (!grad_y_nhwc || !y_nhwc || !grad_c_nhwc || !running_mean || !running_var) return JDET_E_BADARG;
  if (identity_nhwc && own_output_nhwc) return JDET_E_BADARG;
  if (affine && sums_bytes < sizeof(float) * (size_t)g.grid * 2 * C) return JDET_E_WORKSPACE;
  BnParams p{weight, bias, running_mean, running_var, eps};
  const size_t n4 = (size_t)P * g.cpt;
  const size_t lds = affine ? sizeof(float) * 8 * (size_t)g.block : 0;
  hipStream_t st = (hipStream_t)stream;
  const v4f* dy = (const v4f*)grad_y_nhwc;
  const v4f* y = (const v4f*)y_nhwc;
  const v4f* r = (const v4f*)(identity_nhwc ? identity_nhwc : own_output_nhwc);
  v4f* dc = (v4f*)grad_c_nhwc;
#define JDET_BN_OUT_BWD(SRC, AFF) \
  hipLaunchKernelGGL((bn_out_bwd_kernel<SRC, AFF>), dim3(g.grid), dim3(g.block), lds, st, dy, y, r, dc, p, g.cpt, n4, sums)
  if (!affine) JDET_BN_OUT_BWD(0, false);
  else if (identity_nhwc) JDET_BN_OUT_BWD(1, true);
  else if (own_output_nhwc) JDET_BN_OUT_BWD(2, true);
  else JDET_BN_OUT_BWD(0, true);
#undef JDET_BN_OUT_BWD
  return jdet_launch_status();
}

/* Second stage of the per-channel sums of up to 4 BatchNorm layers in one launch: job = {partial [rows][2][C], rows, C,
 * gamma (C) or NULL, grad_gamma (C) or NULL, grad_beta (C) or NULL}: grad_beta = column sums of the first halves,
 * grad_gamma = column sums of the second halves / gamma.  `jobs` is a HOST array read during the call. */
JDET_API int jdet_bn_sums_finish(const jdet_bn_sums_job_t* jobs, int njobs, jdet_stream_t stream) {
  if (njobs < 0 || njobs > 4 || (njobs && !jobs)) return JDET_E_BADARG;
  if (njobs == 0) return JDET_OK;
  SumsJobs js;
  int blocks = 0;
  for (int j = 0; j < njobs; j++) {
    if (!jobs[j].partial || jobs[j].rows < 0 || jobs[j].C <= 0) return JDET_E_BADARG;
    js.job[j] = jobs[j];
    js.first_block[j] = blocks;
    blocks += (jobs[j].C + 31) / 32;
  }
  for (int j = njobs; j < 5; j++) js.first_block[j] = blocks;
  js.njobs = njobs;
  hipLaunchKernelGGL(bn_sums_finish_kernel, dim3(blocks), dim3(1024), 0, (hipStream_t)stream, js);
  return jdet_launch_status();
}

/* Per-channel sum over the P rows of a channels-last (P, C) tensor, any 1 <= C <= 256 -- the bias gradient of a
 * convolution without an activation (the heads' output convs; ConvModule, models/utils/modules.py:L91-175), where
 * jdet_bias_act_backward's channel counts do not apply.  Deterministic two-stage sum.  workspace:
 * jdet_channel_sum_workspace(P, C) bytes. */
JDET_API size_t jdet_channel_sum_workspace(long P, int C) {
  if (P <= 0 || C <= 0 || C > 256) return 0;
  const size_t n = (size_t)P * C;
  const int block = C * (256 / C);
  size_t grid = (n + (size_t)block * 16 - 1) / ((size_t)block * 16);
  if (grid > 256) grid = 256;
  if (grid < 1) grid = 1;
  return sizeof(float) * grid * 2 * C;
}

JDET_API int jdet_channel_sum(const float* x_nhwc, long P, int C, float* sums, void* workspace, size_t workspace_bytes,
                              jdet_stream_t stream) {
  if (P < 0 || C <= 0 || !sums) return JDET_E_BADARG;
  if (C > 256) return JDET_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  if (P == 0) return jdet_zero_async(sums, sizeof(float) * (size_t)C, st);
  if (!x_nhwc) return JDET_E_BADARG;
  const size_t need = jdet_channel_sum_workspace(P, C);
  if (!workspace || workspace_bytes < need) return JDET_E_WORKSPACE;
  const int grid = (int)(need / (sizeof(float) * 2 * C)), block = C * (256 / C);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(grid), dim3(block), 0, st, x_nhwc, (size_t)P * C, C, (float*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL((sums_finish_kernel<false>), dim3((C + 31) / 32), dim3(1024), 0, st, (const float*)workspace, grid, C,
                     (float*)nullptr, sums);
  return jdet_launch_status();
}
