// Two launch-count killers of the single-stage heads.
//
// 1. Sigmoid focal loss (models/losses/focal_loss.py:L5-96): BCE-with-logits in the max_val-stable form, one-hot
//    by (class index + 1) == label, per-row weight, (1 - p_t)^gamma, alpha weighting, summed.  The reference is a
//    chain of ~15 elementwise tensor ops (and ~20 more in autograd's backward) per level per stage: ~350 launches
//    per S2ANet step.  Here one pass produces the loss sum (deterministic two-stage reduction) AND d loss / d logit;
//    the autograd backward is a scale of that buffer.  The log(max(., 1e-10)) floor of the reference can never
//    trigger (the argument is exp(-m) + exp(-x-m) >= 1), so it is not reproduced.
// 2. AlignConv.get_offset (models/roi_heads/s2anet_head.py:L676-713): the 3x3 sampling grid of a refined anchor
//    minus the regular grid, ~25 elementwise ops per level per image there, one thread per (image, location) here,
//    writing the 2*k*k offset planes (dy, dx per tap) the deformable kernels read.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// rows = anchors, C classes; element (r, c): t = (labels[r] == c + 1)
__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ logits,
                                                        const int32_t* __restrict__ labels,
                                                        const float* __restrict__ weight, long M, int C, float alpha,
                                                        float gamma, float* __restrict__ grad,
                                                        float* __restrict__ partial) {
  __shared__ float s_part[4];
  const long n = M * C;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float x = logits[i];
    const bool t = labels[r] == c + 1;
    const float w = weight ? weight[r] : 1.f;
    // stable BCE with logits: (1 - t) * x + max(-x, 0) + log(exp(-m) + exp(-x - m))
    const float m = fmaxf(-x, 0.f);
    const float ce = (t ? 0.f : x) + m + logf(expf(-m) + expf(-x - m));
    const float p = 1.f / (1.f + expf(-x));
    const float pt = t ? p : 1.f - p;          // probability of the true outcome
    const float q = 1.f - pt;
    const float mod = powf(q, gamma);
    const float at = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    acc += at * w * ce * mod;
    // d/dx [ce * q^gamma]: ce = -log(pt);  d pt/dx = s * pt * q with s = +1 (t) / -1 (!t)
    //   = s * ( -q^(gamma+1) - gamma * q^gamma * pt * ce )  ... written without the division by q
    const float s = t ? 1.f : -1.f;
    const float d = -s * (mod * q + gamma * mod * pt * ce);
    grad[i] = at * w * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ partial, int n,
                                                           float* __restrict__ out) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

constexpr int kFocalGrid = 1024;

// ---- the per-level loss of a dense head as ONE node (round 6) ----------------------------------------------------------
// The head calls loss(pred_level, target_level, weight_level, avg_factor) once per pyramid level, with the level's
// targets a column window [:, s:e] of the per-image target arrays and avg_factor a 0-dim device tensor; composed from
// the kernels above that was, per level and loss: a clone per window, the loss, `/ avg_factor`, `* loss_weight`, and three
// scaling launches in backward (profiles/r06_glue.md: 40 clones + 80 scalar ops per S2ANet step).  Here the target /
// weight rows are addressed through their window (row r -> block r / rows_per_block, element r % rows_per_block), the
// finishing workgroup writes (sum / avg_factor) * loss_weight, and one kernel scales the unit gradient by
// (grad_out * loss_weight) / avg_factor -- the operations of the composition, in its order: bit-identical.
struct Blocked {
  long rows_per_block;   // rows of one block (one image's window); >= total rows when the array is contiguous
  long block_stride;     // distance between two blocks, in rows
};
__device__ __forceinline__ long blocked_row(const Blocked& b, long r) {
  const long blk = r / b.rows_per_block;
  return blk * b.block_stride + (r - blk * b.rows_per_block);
}

__global__ __launch_bounds__(256) void focal_fwd_blocked_kernel(const float* __restrict__ logits,
                                                                const int32_t* __restrict__ labels,
                                                                const float* __restrict__ weight, long M, int C,
                                                                float alpha, float gamma, Blocked lb, Blocked wb,
                                                                float* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float s_part[4];
  const long n = M * C;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / C;
    const int c = (int)(i - r * C);
    const float x = logits[i];
    const bool t = labels[blocked_row(lb, r)] == c + 1;
    const float w = weight ? weight[blocked_row(wb, r)] : 1.f;
    const float m = fmaxf(-x, 0.f);
    const float ce = (t ? 0.f : x) + m + logf(expf(-m) + expf(-x - m));
    const float p = 1.f / (1.f + expf(-x));
    const float pt = t ? p : 1.f - p;
    const float q = 1.f - pt;
    const float mod = powf(q, gamma);
    const float at = alpha >= 0.f ? (t ? alpha : 1.f - alpha) : 1.f;
    acc += at * w * ce * mod;
    const float s = t ? 1.f : -1.f;
    const float d = -s * (mod * q + gamma * mod * pt * ce);
    grad[i] = at * w * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// pred (rows, E) contiguous; target / weight rows of E contiguous values behind a blocked row index
__global__ __launch_bounds__(256) void smooth_l1_fwd_blocked_kernel(const float* __restrict__ pred,
                                                                    const float* __restrict__ target,
                                                                    const float* __restrict__ weight, long rows, int E,
                                                                    float beta, Blocked tb, Blocked wb,
                                                                    float* __restrict__ grad,
                                                                    float* __restrict__ partial) {
  __shared__ float s_part[4];
  const long n = rows * E;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / E;
    const int k = (int)(i - r * E);
    const float d = pred[i] - target[blocked_row(tb, r) * E + k];
    const float w = weight ? weight[blocked_row(wb, r) * E + k] : 1.f;
    const float ad = fabsf(d);
    float l, g;
    if (beta != 0.f && ad < beta) {
      l = 0.5f * d * d / beta;
      g = d / beta;
    } else {
      l = beta != 0.f ? ad - 0.5f * beta : ad;
      g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    acc += l * w;
    grad[i] = g * w;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

__global__ __launch_bounds__(256) void sum_partials_scaled_kernel(const float* __restrict__ partial, int n,
                                                                  const float* __restrict__ avg_factor,
                                                                  float loss_weight, float* __restrict__ out) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float total = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
    out[0] = (total / avg_factor[0]) * loss_weight;
  }
}

__global__ __launch_bounds__(256) void loss_grad_scale_kernel(const float* __restrict__ unit, long n,
                                                              const float* __restrict__ grad_out,
                                                              const float* __restrict__ avg_factor, float loss_weight,
                                                              float* __restrict__ out) {
  const float s = (grad_out[0] * loss_weight) / avg_factor[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = unit[i] * s;
}

// weighted smooth-L1 (models/losses/smooth_l1_loss.py:L5-27): per element w * (|d| < beta ? 0.5 d^2 / beta
// : |d| - 0.5 beta), d = pred - target; beta == 0 -> plain L1.  Sum + d/d pred in one pass.
__global__ __launch_bounds__(256) void smooth_l1_fwd_kernel(const float* __restrict__ pred,
                                                            const float* __restrict__ target,
                                                            const float* __restrict__ weight, long n, float beta,
                                                            float* __restrict__ grad, float* __restrict__ partial) {
  __shared__ float s_part[4];
  float acc = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float d = pred[i] - target[i];
    const float w = weight ? weight[i] : 1.f;
    const float ad = fabsf(d);
    float l, g;
    if (beta != 0.f && ad < beta) {
      l = 0.5f * d * d / beta;
      g = d / beta;
    } else {
      l = beta != 0.f ? ad - 0.5f * beta : ad;
      g = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    }
    acc += l * w;
    grad[i] = g * w;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (s_part[0] + s_part[1]) + (s_part[2] + s_part[3]);
}

// anchors (N, H*W, 5) image coordinates -> offsets (N, 2*k*k, H, W)
__global__ __launch_bounds__(256) void align_offset_kernel(const float* __restrict__ anchors, int N, int H, int W,
                                                           float stride, int ks, float* __restrict__ offset) {
  const long hw = (long)H * W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * hw) return;
  const long n = idx / hw, pos = idx - n * hw;
  const int yc = (int)(pos / W), xc = (int)(pos - (long)yc * W);
  const float* a = anchors + idx * 5;
  const float x_ctr = a[0] / stride, y_ctr = a[1] / stride, w = a[2] / stride, h = a[3] / stride;
  const float cs = cosf(a[4]), sn = sinf(a[4]);
  const float dw = w / (float)ks, dh = h / (float)ks;
  const int pad = (ks - 1) / 2;
  float* o = offset + n * 2 * ks * ks * hw + pos;
  for (int i = 0; i < ks; i++)
    for (int j = 0; j < ks; j++) {
      const float yy = (float)(i - pad), xx = (float)(j - pad);
      const float x = dw * xx, y = dh * yy;
      const float xr = cs * x - sn * y, yr = sn * x + cs * y;
      const float x_anchor = xr + x_ctr, y_anchor = yr + y_ctr;
      const float x_conv = (float)xc + xx, y_conv = (float)yc + yy;
      const int tap = i * ks + j;
      o[(long)(2 * tap) * hw] = y_anchor - y_conv;
      o[(long)(2 * tap + 1) * hw] = x_anchor - x_conv;
    }
}

}  // namespace

JDET_API size_t jdet_sigmoid_focal_loss_workspace(void) { return sizeof(float) * kFocalGrid; }

JDET_API int jdet_sigmoid_focal_loss(const float* logits, const int32_t* labels, const float* weight, long M, int C,
                                     float alpha, float gamma, float* loss_sum, float* grad_logits,
                                     void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  if (M < 0 || C <= 0 || !loss_sum) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (M == 0) return jdet_zero_async(loss_sum, sizeof(float), st);
  if (!logits || !labels || !grad_logits || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_sigmoid_focal_loss_workspace()) return JDET_E_WORKSPACE;
  const long n = M * C;
  int grid = (int)((n + 1023) / 1024);
  if (grid > kFocalGrid) grid = kFocalGrid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(focal_fwd_kernel, dim3(grid), dim3(256), 0, st, logits, labels, weight, M, C, alpha, gamma,
                     grad_logits, (float*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, loss_sum);
  return jdet_launch_status();
}

JDET_API int jdet_smooth_l1_loss(const float* pred, const float* target, const float* weight, long n, float beta,
                                 float* loss_sum, float* grad_pred, void* workspace, size_t workspace_bytes,
                                 jdet_stream_t stream) {
  if (n < 0 || beta < 0.f || !loss_sum) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) return jdet_zero_async(loss_sum, sizeof(float), st);
  if (!pred || !target || !grad_pred || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_sigmoid_focal_loss_workspace()) return JDET_E_WORKSPACE;
  int grid = (int)((n + 1023) / 1024);
  if (grid > kFocalGrid) grid = kFocalGrid;
  hipLaunchKernelGGL(smooth_l1_fwd_kernel, dim3(grid), dim3(256), 0, st, pred, target, weight, n, beta, grad_pred,
                     (float*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, loss_sum);
  return jdet_launch_status();
}

JDET_API int jdet_sigmoid_focal_loss_level(const float* logits, const int32_t* labels, long label_rows_per_block,
                                           long label_block_stride, const float* weight, long weight_rows_per_block,
                                           long weight_block_stride, long M, int C, float alpha, float gamma,
                                           const float* avg_factor, float loss_weight, float* loss,
                                           float* grad_logits, void* workspace, size_t workspace_bytes,
                                           jdet_stream_t stream) {
  if (M <= 0 || C <= 0 || !loss || !avg_factor || !logits || !labels || !grad_logits || !workspace) return JDET_E_BADARG;
  if (label_rows_per_block <= 0 || label_block_stride < 0 || (weight && (weight_rows_per_block <= 0 || weight_block_stride < 0)))
    return JDET_E_BADARG;
  if (workspace_bytes < jdet_sigmoid_focal_loss_workspace()) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long n = M * C;
  int grid = (int)((n + 1023) / 1024);
  if (grid > kFocalGrid) grid = kFocalGrid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(focal_fwd_blocked_kernel, dim3(grid), dim3(256), 0, st, logits, labels, weight, M, C, alpha, gamma,
                     Blocked{label_rows_per_block, label_block_stride},
                     Blocked{weight ? weight_rows_per_block : 1, weight ? weight_block_stride : 0}, grad_logits,
                     (float*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, avg_factor,
                     loss_weight, loss);
  return jdet_launch_status();
}

JDET_API int jdet_smooth_l1_loss_level(const float* pred, const float* target, long target_rows_per_block,
                                       long target_block_stride, const float* weight, long weight_rows_per_block,
                                       long weight_block_stride, long rows, int E, float beta, const float* avg_factor,
                                       float loss_weight, float* loss, float* grad_pred, void* workspace,
                                       size_t workspace_bytes, jdet_stream_t stream) {
  if (rows <= 0 || E <= 0 || beta < 0.f || !loss || !avg_factor || !pred || !target || !grad_pred || !workspace)
    return JDET_E_BADARG;
  if (target_rows_per_block <= 0 || target_block_stride < 0 || (weight && (weight_rows_per_block <= 0 || weight_block_stride < 0)))
    return JDET_E_BADARG;
  if (workspace_bytes < jdet_sigmoid_focal_loss_workspace()) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const long n = rows * E;
  int grid = (int)((n + 1023) / 1024);
  if (grid > kFocalGrid) grid = kFocalGrid;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(smooth_l1_fwd_blocked_kernel, dim3(grid), dim3(256), 0, st, pred, target, weight, rows, E, beta,
                     Blocked{target_rows_per_block, target_block_stride},
                     Blocked{weight ? weight_rows_per_block : 1, weight ? weight_block_stride : 0}, grad_pred,
                     (float*)workspace);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(sum_partials_scaled_kernel, dim3(1), dim3(256), 0, st, (const float*)workspace, grid, avg_factor,
                     loss_weight, loss);
  return jdet_launch_status();
}

JDET_API int jdet_loss_grad_scale(const float* unit_grad, long n, const float* grad_out, const float* avg_factor,
                                  float loss_weight, float* out, jdet_stream_t stream) {
  if (n < 0) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!unit_grad || !grad_out || !avg_factor || !out) return JDET_E_BADARG;
  int grid = (int)((n + 1023) / 1024);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(loss_grad_scale_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, unit_grad, n, grad_out,
                     avg_factor, loss_weight, out);
  return jdet_launch_status();
}

JDET_API int jdet_align_conv_offset(const float* anchors, int N, int H, int W, float stride, int kernel_size,
                                    float* offset, jdet_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || kernel_size <= 0 || kernel_size % 2 == 0 || stride <= 0.f) return JDET_E_BADARG;
  if (N == 0) return JDET_OK;
  if (!anchors || !offset) return JDET_E_BADARG;
  const long total = (long)N * H * W;
  hipLaunchKernelGGL(align_offset_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     anchors, N, H, W, stride, kernel_size, offset);
  return jdet_launch_status();
}
