// Rotated box delta codec and max-IoU assignment for gfx950.
//
// Reference semantics (Jittor tensor programs, ~30 elementwise launches / a per-gt Python loop):
//   python/jdet/models/boxes/box_ops.py:L176-178  norm_angle
//   python/jdet/models/boxes/box_ops.py:L180-226  bbox2delta_rotated
//   python/jdet/models/boxes/box_ops.py:L229-285  delta2bbox_rotated
//   python/jdet/models/boxes/assigner.py:L160-219 MaxIoUAssigner.assign_wrt_overlaps
//
// All three are tiny HBM-bound passes (20 B per box in, 20 B out; the assigner reads the K x A
// overlap matrix twice).  Their cost in the reference is launch count and host synchronisation
// (`jt.sync_all()` every 100 gts, one masked store per gt), so each is ONE launch here:
//   * codec kernels: one lane per (box, class) element, fully fused;
//   * assignment: kernel 1 reduces each gt row to its max (one workgroup per gt, wave shuffles);
//     kernel 2 owns one anchor per lane: column argmax (first maximum), neg / pos thresholds, the
//     low-quality pass as "last gt i with overlaps[i,j] == gt_max[i] >= min_pos_iou wins"
//     (equivalent to the reference's in-order overwrite loop), label gather.  No host sync.
#include "common.h"

namespace {

// floor-mod norm_angle: (a + pi/4) mod pi - pi/4   (box_ops.py:L176-178, range [-pi/4, pi])
__device__ __forceinline__ float norm_angle(float a) {
  const float lo = (float)(-M_PI / 4), span = (float)M_PI;
  const float x = a - lo;
  float r = x - floorf(x / span) * span;  // python-style % for a positive modulus
  return r + lo;
}

struct Vec5 {
  float v[5];
};

__global__ __launch_bounds__(256) void delta2bbox_rotated_kernel(const float* __restrict__ rois,
                                                                 const float* __restrict__ deltas, long n,
                                                                 int ncls, Vec5 means, Vec5 stds,
                                                                 float max_ratio, float* __restrict__ out) {
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n * ncls; idx += (long)gridDim.x * 256) {
    const long i = idx / ncls;
    const float* r = rois + i * 5;
    const float* d = deltas + idx * 5;
    const float dx = d[0] * stds.v[0] + means.v[0];
    const float dy = d[1] * stds.v[1] + means.v[1];
    float dw = d[2] * stds.v[2] + means.v[2];
    float dh = d[3] * stds.v[3] + means.v[3];
    const float da = d[4] * stds.v[4] + means.v[4];
    dw = fminf(fmaxf(dw, -max_ratio), max_ratio);
    dh = fminf(fmaxf(dh, -max_ratio), max_ratio);
    const float rx = r[0], ry = r[1], rw = r[2], rh = r[3], ra = r[4];
    const float c = cosf(ra), s = sinf(ra);
    float* o = out + idx * 5;
    o[0] = dx * rw * c - dy * rh * s + rx;
    o[1] = dx * rw * s + dy * rh * c + ry;
    o[2] = rw * expf(dw);
    o[3] = rh * expf(dh);
    o[4] = norm_angle((float)M_PI * da + ra);
  }
}

__device__ __forceinline__ void encode_one(const float* __restrict__ p, const float* __restrict__ g, const Vec5& means,
                                           const Vec5& stds, float* __restrict__ o) {
  const float pw = p[2], ph = p[3], pa = p[4];
  const float c = cosf(pa), s = sinf(pa);
  const float cx = g[0] - p[0], cy = g[1] - p[1];
  float d[5];
  d[0] = (c * cx + s * cy) / pw;
  d[1] = (-s * cx + c * cy) / ph;
  // jt.safe_log = log(clamp(x, 1e-30, 1e30))
  d[2] = logf(fminf(fmaxf(g[2] / pw, 1e-30f), 1e30f));
  d[3] = logf(fminf(fmaxf(g[3] / ph, 1e-30f), 1e30f));
  d[4] = norm_angle(g[4] - pa) / (float)M_PI;
#pragma unroll
  for (int k = 0; k < 5; k++) o[k] = (d[k] - means.v[k]) / stds.v[k];
}

__global__ __launch_bounds__(256) void bbox2delta_rotated_kernel(const float* __restrict__ proposals,
                                                                 const float* __restrict__ gt, long n,
                                                                 Vec5 means, Vec5 stds,
                                                                 float* __restrict__ out) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    encode_one(proposals + i * 5, gt + i * 5, means, stds, out + i * 5);
}

// ---- dense anchor targets ---------------------------------------------------------------------
// anchor_target_single (models/boxes/anchor_target.py:L105-180) for the PseudoSampler case, without
// the pos_inds / neg_inds index lists: every anchor writes its own label, label weight, encoded box
// target and box weight from its assignment (0 = negative, -1 = ignored, i+1 = gt i).  Fixed shapes
// in and out -- no nonzero(), no host sync; the number of positives stays on the device.
__global__ __launch_bounds__(256) void anchor_targets_rotated_kernel(
    const float* __restrict__ anchors, const float* __restrict__ gt, const int32_t* __restrict__ gt_labels,
    const int32_t* __restrict__ gt_inds, int A, Vec5 means, Vec5 stds, float pos_weight,
    int32_t* __restrict__ labels, float* __restrict__ label_weights, float* __restrict__ bbox_targets,
    float* __restrict__ bbox_weights, int32_t* __restrict__ num_pos) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int gi = j < A ? gt_inds[j] : 0;
  const bool pos = gi > 0;
  if (j < A) {
    float* t = bbox_targets + (size_t)j * 5;
    float* w = bbox_weights + (size_t)j * 5;
    if (pos) {
      encode_one(anchors + (size_t)j * 5, gt + (size_t)(gi - 1) * 5, means, stds, t);
#pragma unroll
      for (int k = 0; k < 5; k++) w[k] = 1.f;
      labels[j] = gt_labels ? gt_labels[gi - 1] : 1;
      label_weights[j] = pos_weight;
    } else {
#pragma unroll
      for (int k = 0; k < 5; k++) t[k] = w[k] = 0.f;
      labels[j] = 0;
      label_weights[j] = gi == 0 ? 1.f : 0.f;
    }
  }
  const unsigned long long m = __ballot(pos);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(num_pos, __popcll(m));
}

// ---- MaxIoUAssigner ------------------------------------------------------------------------
// Row maxima in two launches: (gt row, chunk of the anchors) partial maxima -- K workgroups alone (K = 64 gts,
// 262 k anchors: 67 MB) left 3/4 of the CUs idle, 86 us -- then one thread per row folds its chunks.  max is
// order-independent -> bit-exact.
constexpr int kRowMaxChunks = 32;

__global__ __launch_bounds__(256) void assign_row_max_kernel(const float* __restrict__ overlaps, int K, int A,
                                                             float* __restrict__ partial) {
  __shared__ float s_part[4];
  const int i = blockIdx.x, chunk = blockIdx.y;
  const int per = (A + kRowMaxChunks - 1) / kRowMaxChunks;
  const int lo = chunk * per, hi = min(lo + per, A);
  const float* row = overlaps + (size_t)i * A;
  float m = -INFINITY;
  for (int j = lo + threadIdx.x; j < hi; j += 256) m = fmaxf(m, row[j]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    partial[(size_t)i * kRowMaxChunks + chunk] = fmaxf(fmaxf(s_part[0], s_part[1]), fmaxf(s_part[2], s_part[3]));
}

__global__ __launch_bounds__(256) void assign_row_max_fold_kernel(const float* __restrict__ partial, int K,
                                                                  float* __restrict__ gt_max) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= K) return;
  float m = -INFINITY;
  for (int c = 0; c < kRowMaxChunks; c++) m = fmaxf(m, partial[(size_t)i * kRowMaxChunks + c]);
  gt_max[i] = m;
}

// Axis-aligned overlaps (K gts x A boxes) in one pass, the arithmetic of iou_calculator.py:L235-350 (`bbox_overlaps`,
// not aligned, modes iou / iof; `plus_one` = the legacy +1 pixel convention of BboxOverlaps2D_v1) in the same
// operation order -- bit-identical to the elementwise tensor program, which costs ~12 passes over (K, A) tensors.
// alive (A) != NULL: columns of dead boxes get -1 (fixed-shape heads: padding rows, anchors outside the image).
__global__ __launch_bounds__(256) void bbox_overlaps_hbb_kernel(const float* __restrict__ gts, int K,
                                                                const float* __restrict__ boxes, int A, int stride2,
                                                                int iof, float plus_one, float eps,
                                                                const uint8_t* __restrict__ alive,
                                                                float* __restrict__ out) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= A) return;
  const float* b = boxes + (size_t)j * stride2;
  const float bx1 = b[0], by1 = b[1], bx2 = b[2], by2 = b[3];
  const float area2 = (bx2 - bx1 + plus_one) * (by2 - by1 + plus_one);
  const bool dead = alive && !alive[j];
  for (int i = 0; i < K; i++) {
    const float* g = gts + (size_t)i * 4;            // wave-uniform -> scalar loads
    const float gx1 = g[0], gy1 = g[1], gx2 = g[2], gy2 = g[3];
    const float area1 = (gx2 - gx1 + plus_one) * (gy2 - gy1 + plus_one);
    const float w = fmaxf(fminf(gx2, bx2) - fmaxf(gx1, bx1) + plus_one, 0.f);
    const float h = fmaxf(fminf(gy2, by2) - fmaxf(gy1, by1) + plus_one, 0.f);
    const float overlap = w * h;
    float uni = iof ? area1 : area1 + area2 - overlap;
    uni = fmaxf(uni, eps);
    out[(size_t)i * A + j] = dead ? -1.f : overlap / uni;
  }
}

__global__ __launch_bounds__(256) void assign_anchor_kernel(
    const float* __restrict__ overlaps, int K, int A, const float* __restrict__ gt_max, float pos_iou_thr,
    float neg_lo, float neg_hi, float min_pos_iou, int match_low_quality, int gt_max_assign_all,
    const int32_t* __restrict__ gt_labels, int labels_filled, int32_t* __restrict__ gt_inds,
    float* __restrict__ max_overlaps, int32_t* __restrict__ labels) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= A) return;
  // column argmax over the K gts (first maximum) -- coalesced: consecutive lanes, consecutive j
  float best = -INFINITY;
  int arg = 0;
  int low = -1;  // last gt whose row maximum this anchor attains (step 4, gt_max_assign_all)
  for (int i = 0; i < K; i++) {
    const float v = overlaps[(size_t)i * A + j];
    if (v > best) {
      best = v;
      arg = i;
    }
    if (match_low_quality && gt_max_assign_all) {
      const float gm = gt_max[i];
      if (gm >= min_pos_iou && v == gm) low = i;
    }
  }
  int assigned = -1;                                        // 1. default -1
  if (best >= neg_lo && best < neg_hi) assigned = 0;         // 2. negatives (assigner.py:L187-193)
  if (best >= pos_iou_thr) assigned = arg + 1;               // 3. positives (L196-197)
  if (low >= 0) assigned = low + 1;                          // 4. low-quality matches (L200-207)
  gt_inds[j] = assigned;
  max_overlaps[j] = best;
  if (labels) {
    labels[j] = (assigned > 0 && gt_labels) ? gt_labels[assigned - 1] : labels_filled;  // L211-215
  }
}

// step 4 with gt_max_assign_all = False: only gt_argmax_overlaps[i] (first maximum of row i) is set,
// in gt order (later gts overwrite).  One lane per gt finds its first-maximum column, then a single
// lane applies the K updates in order (K is tiny).
__global__ __launch_bounds__(256) void assign_row_argmax_apply_kernel(
    const float* __restrict__ overlaps, int K, int A, const float* __restrict__ gt_max, float min_pos_iou,
    const int32_t* __restrict__ gt_labels, int32_t* __restrict__ row_arg, int32_t* __restrict__ gt_inds,
    int32_t* __restrict__ labels) {
  for (int i = threadIdx.x; i < K; i += 256) {
    const float gm = gt_max[i];
    int a = 0;
    for (int j = 0; j < A; j++)
      if (overlaps[(size_t)i * A + j] == gm) {
        a = j;
        break;
      }
    row_arg[i] = a;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < K; i++)
      if (gt_max[i] >= min_pos_iou) {
        gt_inds[row_arg[i]] = i + 1;
        if (labels && gt_labels) labels[row_arg[i]] = gt_labels[i];
      }
  }
}

inline int grid_for(long n) {
  long g = (n + 255) / 256;
  return (int)(g > 262144 ? 262144 : (g < 1 ? 1 : g));
}

}  // namespace

JDET_API int jdet_delta2bbox_rotated(const float* rois, const float* deltas, int n, int ncls,
                                     const float* means5, const float* stds5, float wh_ratio_clip,
                                     float* out, jdet_stream_t stream) {
  if (n < 0 || ncls <= 0 || !means5 || !stds5 || !(wh_ratio_clip > 0)) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!rois || !deltas || !out) return JDET_E_BADARG;
  Vec5 m, s;
  for (int k = 0; k < 5; k++) {
    m.v[k] = means5[k];
    s.v[k] = stds5[k];
  }
  const float max_ratio = fabsf(logf(wh_ratio_clip));
  hipLaunchKernelGGL(delta2bbox_rotated_kernel, dim3(grid_for((long)n * ncls)), dim3(256), 0,
                     (hipStream_t)stream, rois, deltas, (long)n, ncls, m, s, max_ratio, out);
  return jdet_launch_status();
}

JDET_API int jdet_bbox2delta_rotated(const float* proposals, const float* gt, int n, const float* means5,
                                     const float* stds5, float* out, jdet_stream_t stream) {
  if (n < 0 || !means5 || !stds5) return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!proposals || !gt || !out) return JDET_E_BADARG;
  Vec5 m, s;
  for (int k = 0; k < 5; k++) {
    m.v[k] = means5[k];
    s.v[k] = stds5[k];
  }
  hipLaunchKernelGGL(bbox2delta_rotated_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     proposals, gt, (long)n, m, s, out);
  return jdet_launch_status();
}

JDET_API int jdet_anchor_targets_rotated(const float* anchors, const float* gt, const int32_t* gt_labels,
                                         const int32_t* gt_inds, int A, int K, const float* means5,
                                         const float* stds5, float pos_weight, int32_t* labels,
                                         float* label_weights, float* bbox_targets, float* bbox_weights,
                                         int32_t* num_pos, jdet_stream_t stream) {
  if (A < 0 || K < 0) return JDET_E_BADARG;
  if (A == 0) return JDET_OK;
  if (!anchors || !gt_inds || !means5 || !stds5 || !labels || !label_weights || !bbox_targets || !bbox_weights ||
      !num_pos || (K > 0 && !gt))
    return JDET_E_BADARG;
  Vec5 m, s;
  for (int k = 0; k < 5; k++) {
    m.v[k] = means5[k];
    s.v[k] = stds5[k];
  }
  hipLaunchKernelGGL(anchor_targets_rotated_kernel, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     anchors, gt, gt_labels, gt_inds, A, m, s, pos_weight, labels, label_weights, bbox_targets,
                     bbox_weights, num_pos);
  return jdet_launch_status();
}

JDET_API size_t jdet_assign_max_iou_workspace(int K) {
  return K > 0 ? (size_t)K * (8 + 4 * kRowMaxChunks) : 0;   // row maxima, row argmax, partial maxima
}

JDET_API int jdet_bbox_overlaps_hbb(const float* gts, int K, const float* boxes, int A, int box_stride, int iof,
                                    int plus_one, float eps, const uint8_t* alive, float* out,
                                    jdet_stream_t stream) {
  if (K < 0 || A < 0 || box_stride < 4) return JDET_E_BADARG;
  if (K == 0 || A == 0) return JDET_OK;
  if (!gts || !boxes || !out) return JDET_E_BADARG;
  hipLaunchKernelGGL(bbox_overlaps_hbb_kernel, dim3((A + 255) / 256), dim3(256), 0, (hipStream_t)stream, gts, K,
                     boxes, A, box_stride, iof ? 1 : 0, plus_one ? 1.f : 0.f, eps, alive, out);
  return jdet_launch_status();
}

JDET_API int jdet_assign_max_iou(const float* overlaps, int K, int A, float pos_iou_thr, float neg_iou_lo,
                                 float neg_iou_hi, float min_pos_iou, int match_low_quality,
                                 int gt_max_assign_all, const int32_t* gt_labels, int labels_filled,
                                 int32_t* gt_inds, float* max_overlaps, int32_t* labels, void* workspace,
                                 size_t workspace_bytes, jdet_stream_t stream) {
  if (K <= 0 || A <= 0) return JDET_E_BADARG;  // the reference raises ValueError('No gt or proposals')
  if (!overlaps || !gt_inds || !max_overlaps || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_assign_max_iou_workspace(K)) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* gt_max = (float*)workspace;
  int32_t* row_arg = (int32_t*)(gt_max + K);
  float* partial = (float*)(row_arg + K);
  hipLaunchKernelGGL(assign_row_max_kernel, dim3(K, kRowMaxChunks), dim3(256), 0, st, overlaps, K, A, partial);
  hipLaunchKernelGGL(assign_row_max_fold_kernel, dim3((K + 255) / 256), dim3(256), 0, st, partial, K, gt_max);
  int e = jdet_launch_status();
  if (e) return e;
  hipLaunchKernelGGL(assign_anchor_kernel, dim3((A + 255) / 256), dim3(256), 0, st, overlaps, K, A, gt_max,
                     pos_iou_thr, neg_iou_lo, neg_iou_hi, min_pos_iou, match_low_quality, gt_max_assign_all,
                     gt_labels, labels_filled, gt_inds, max_overlaps, labels);
  e = jdet_launch_status();
  if (e) return e;
  if (match_low_quality && !gt_max_assign_all) {
    hipLaunchKernelGGL(assign_row_argmax_apply_kernel, dim3(1), dim3(256), 0, st, overlaps, K, A, gt_max,
                       min_pos_iou, gt_labels, row_arg, gt_inds, labels);
    e = jdet_launch_status();
  }
  return e;
}
