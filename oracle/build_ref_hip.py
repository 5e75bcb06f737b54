#!/usr/bin/env python3
"""Build oracle/_ref/libjdet_ref_hip.so: the reference's OWN GPU kernels, compiled for gfx950.  TEST INFRASTRUCTURE.

The operators SURVEY 8(a) lists as "CUDA only" (the five RoIAligns, DeformConv v1 / v2 sampling, deformable PSRoI
pooling, feature refinement, the RepPoints geometry, convex_sort) have no CPU source in the reference, but their kernel text is plain CUDA C++ --
`__global__` functions, blockIdx / threadIdx, atomicAdd, <<< >>> launches -- and that dialect is what hipcc compiles
natively: no stand-in for a CUDA built-in, header, library or tool is written here.  As oracle/build_ref.py does for
the CPU sources, this recipe reads the kernel text with `ast` from the files WHERE THEY LIE under /root/reference
(nothing is copied into the repo; the generated .hip lives and dies in a TemporaryDirectory), puts every file's text in
its own namespace, and appends `extern "C"` entry points that launch the kernels with the grid / block arithmetic of the
reference's `jt.code` launch snippets (cited per entry point).  The only edit to the kernel text is dropping
`#include <executor.h>` (a Jittor header none of these kernels uses), as in build_ref.py.

Two builds of the same text: `-ffp-contract=off` (libjdet_ref_hip.so: the operation order of the text, no fused
multiply-add -- what the CPU restatement oracle/jdet_oracle.cpp is written to match bit for bit) and hipcc's default
contraction (libjdet_ref_hip_fma.so: what a CUDA toolchain's default `-fmad=true` would do to the same text).

Needs /root/reference and hipcc: runs in the build container (hipcc cross-compiles gfx950 without a GPU); the .so files
travel to the GPU box with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored), where tests/ use them to pin
both the HIP kernels and the CPU restatement against the reference's kernels running on the same device.

dcn_v2.py keeps its kernel text inline in the `jt.code(cuda_header=...)` calls: read from there.  The two pooling headers
compile as they are; from the convolution's backward header (all three sampling kernels + their launch wrappers) the
includes of <cuda_runtime.h> / <cublas_v2.h> and the `extern cublasHandle_t cublas_handle;` declaration are dropped as
well -- the GEMMs that use them live in the launch snippet, not in the kernels.

nms_poly.py: the mask kernel is built; the greedy scan over the mask is host code inside the launch snippet (which also
uses Jittor's allocator) and is restated in the entry point, statement by statement (L207-229).

NOT built: the kernels of models outside SURVEY 8.
"""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from build_ref import OPS, OUT_DIR, module_strings  # noqa: E402

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

PRELUDE = r'''
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <cfloat>
#include <math.h>
#include <stdio.h>
#include <float.h>
#include <algorithm>
#include <vector>
#include <iostream>
using std::min; using std::max;
#define API extern "C" __attribute__((visibility("default")))
static inline int ref_sync() { return (int)hipDeviceSynchronize(); }
'''

UNDEF = ("#undef CUDA_1D_KERNEL_LOOP\n#undef CUDA_KERNEL_LOOP\n#undef THREADS_PER_BLOCK\n#undef PI\n#undef maxn\n"
         "#undef nmax\n#undef CeilDIV\n#undef ROI_ALIGN_VERSION\n")


def ns(name, body):
    body = re.sub(r"#include\s*<executor.h>", "", body).replace("#undef out", "")
    return "namespace %s {\n%s\n}\n%s" % (name, body, UNDEF)


ENTRY = r'''
// ---- roi_align_rotated.py:L265-283 / L286-307 (and the _v1 twin L311-349): one thread per output element
#define ROT_ENTRY(NS, NAME)                                                                                        \
API int NAME##_forward(const float* input, const float* rois, int R, int C, int H, int W, int PH, int PW,          \
                       float spatial_scale, float sampling_ratio, float* out) {                                    \
  const int output_size = R * PH * PW * C;                                                                        \
  if (output_size)                                                                                                 \
    NS::ROIAlignRotatedForward<<<NS::GET_BLOCKS(output_size), 1024>>>(output_size, input, rois, spatial_scale,    \
                                                                      sampling_ratio, C, H, W, PH, PW, out);      \
  return ref_sync();                                                                                               \
}                                                                                                                  \
API int NAME##_backward(const float* grad, const float* rois, int R, int N, int C, int H, int W, int PH, int PW,  \
                        float spatial_scale, float sampling_ratio, float* grad_input) {                            \
  const int output_size = R * PH * PW * C;                                                                        \
  hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)N * C * H * W);                                           \
  if (output_size)                                                                                                 \
    NS::ROIAlignBackward<<<NS::GET_BLOCKS(output_size), 1024>>>(output_size, grad, rois, spatial_scale,           \
                                                                sampling_ratio, C, H, W, PH, PW, grad_input);     \
  return ref_sync();                                                                                               \
}
ROT_ENTRY(ref_rroi, refhip_roi_align_rotated)
ROT_ENTRY(ref_rroi_v1, refhip_roi_align_rotated_v1)

// ---- riroi_align.py:L404-427 / L440-468: the header's own launchers (C = channels per orientation)
API int refhip_riroi_align_forward(const float* input, const float* rois, int R, int C, int H, int W, int PH, int PW,
                                   float spatial_scale, int sample_num, int nO, float* out) {
  hipMemsetAsync(out, 0, sizeof(float) * (size_t)R * C * nO * PH * PW);
  if (R) ref_riroi::RiROIAlignForwardLaucher<float>(input, rois, spatial_scale, sample_num, C, H, W, R, PH, PW, nO, out);
  return ref_sync();
}
API int refhip_riroi_align_backward(const float* grad, const float* rois, int R, int N, int C, int H, int W, int PH,
                                    int PW, float spatial_scale, int sample_num, int nO, float* grad_input) {
  hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)N * C * nO * H * W);
  if (R) ref_riroi::RiROIAlignBackwardLaucher<float>(grad, rois, spatial_scale, sample_num, C, H, W, R, PH, PW, nO,
                                                     grad_input);
  return ref_sync();
}

// ---- roi_align.py:L217-237 / L240-263 (ROI_ALIGN_VERSION 0 and 1; 512 threads per block)
#define HBB_ENTRY(NS, NAME)                                                                                        \
API int NAME##_forward(const float* input, const float* rois, int R, int C, int H, int W, int PH, int PW,          \
                       float spatial_scale, float sampling_ratio, float* out) {                                    \
  const int output_size = R * PH * PW * C;                                                                        \
  if (output_size)                                                                                                 \
    NS::RoIAlignForward<<<(output_size + 511) / 512, 512>>>(output_size, input, C, H, W, PH, PW, rois, out,       \
                                                            spatial_scale, sampling_ratio);                       \
  return ref_sync();                                                                                               \
}                                                                                                                  \
API int NAME##_backward(const float* grad, const float* rois, int R, int N, int C, int H, int W, int PH, int PW,  \
                        float spatial_scale, float sampling_ratio, float* grad_input) {                            \
  const int output_size = R * PH * PW * C;                                                                        \
  hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)N * C * H * W);                                           \
  if (output_size)                                                                                                 \
    NS::RoIAlignBackwardFeature<<<(output_size + 511) / 512, 512>>>(output_size, grad, R, C, H, W, PH, PW,        \
                                                                    grad_input, rois, spatial_scale,              \
                                                                    sampling_ratio);                              \
  return ref_sync();                                                                                               \
}
HBB_ENTRY(ref_hroi0, refhip_roi_align_v0)
HBB_ENTRY(ref_hroi1, refhip_roi_align_v1)

// ---- dcn_v1.py:L309-338 (im2col), L374-410 (col2im), L340-372 (col2im_coord); parallel_imgs = B
API int refhip_deform_im2col(const float* im, const float* offset, int B, int C, int H, int W, int kh, int kw,
                             int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* col) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int num_kernels = C * Ho * Wo * B;
  hipMemsetAsync(col, 0, sizeof(float) * (size_t)C * kh * kw * B * Ho * Wo);
  if (num_kernels)
    ref_dcn::deformable_im2col_gpu_kernel<<<ref_dcn::GET_BLOCKS(num_kernels), ref_dcn::CUDA_NUM_THREADS>>>(
        num_kernels, im, offset, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, C / dg, B, C, dg, Ho, Wo,
        col);
  return ref_sync();
}
API int refhip_deform_col2im(const float* col, const float* offset, int B, int C, int H, int W, int kh, int kw,
                             int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                             float* grad_im) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int num_kernels = C * kh * kw * Ho * Wo * B;
  hipMemsetAsync(grad_im, 0, sizeof(float) * (size_t)B * C * H * W);
  if (num_kernels)
    ref_dcn::deformable_col2im_gpu_kernel<<<ref_dcn::GET_BLOCKS(num_kernels), ref_dcn::CUDA_NUM_THREADS>>>(
        num_kernels, col, offset, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, C / dg, B, dg, Ho, Wo,
        grad_im, B + C + H + W);
  return ref_sync();
}
API int refhip_deform_col2im_coord(const float* col, const float* im, const float* offset, int B, int C, int H, int W,
                                   int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                   int dil_w, int dg, float* grad_offset) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int num_kernels = Ho * Wo * 2 * kh * kw * dg * B;
  hipMemsetAsync(grad_offset, 0, sizeof(float) * (size_t)num_kernels);
  if (num_kernels)
    ref_dcn::deformable_col2im_coord_gpu_kernel<<<ref_dcn::GET_BLOCKS(num_kernels), ref_dcn::CUDA_NUM_THREADS>>>(
        num_kernels, col, im, offset, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
        C * kh * kw / dg, B, 2 * kh * kw * dg, dg, Ho, Wo, grad_offset);
  return ref_sync();
}

// ---- fr.py:L234-254: features (N, C, H, W), best_bboxes (N, H, W, 5)
API int refhip_feature_refine_forward(const float* features, const float* boxes, int N, int C, int H, int W,
                                      float spatial_scale, int points, float* out) {
  const int output_size = N * C * H * W;
  if (output_size)
    ref_fr::feature_refine_forward_kernel<<<ref_fr::GET_BLOCKS(output_size), 1024>>>(output_size, points, features,
                                                                                     boxes, spatial_scale, C, H, W, out);
  return ref_sync();
}
API int refhip_feature_refine_backward(const float* top_grad, const float* boxes, int N, int C, int H, int W,
                                       float spatial_scale, int points, float* bottom_grad) {
  const int output_size = N * C * H * W;
  hipMemsetAsync(bottom_grad, 0, sizeof(float) * (size_t)output_size);          // jt.zeros_like(top_grad), L245
  if (output_size)
    ref_fr::feature_refine_backward_kernel<<<ref_fr::GET_BLOCKS(output_size), 1024>>>(output_size, points, top_grad,
                                                                                      boxes, spatial_scale, C, H, W,
                                                                                      bottom_grad);
  return ref_sync();
}

// ---- reppoints_convex_iou/convex_iou.py:L7-27, reppoints_min_area_bbox/min_area_bbox.py:L7-20 (512 threads)
API int refhip_convex_iou(const float* pointsets, int N, const float* polygons, int M, float* ious) {
  if (N > 0 && M > 0)
    ref_cvx_iou::convex_iou_kernel<<<(N + 511) / 512, 512>>>(N, M, pointsets, polygons, ious);
  return ref_sync();
}
// ---- reppoints_convex_iou/convex_giou.py:L7-27 (aligned pairs, 64 threads per block, (N, 19) output)
API int refhip_convex_giou(const float* pointsets, const float* polygons, int N, float* out) {
  if (N > 0)
    ref_cvx_giou::convex_giou_kernel<<<(N + ref_cvx_giou::threadsPerBlock - 1) / ref_cvx_giou::threadsPerBlock,
                                      ref_cvx_giou::threadsPerBlock>>>(N, N, pointsets, polygons, out);
  return ref_sync();
}
API int refhip_min_area_bbox(const float* pointsets, int N, float* bboxes) {
  if (N > 0) ref_cvx_box::minareabbox_kernel<<<(N + 511) / 512, 512>>>(N, pointsets, bboxes);
  return ref_sync();
}

// ---- convex_sort.py:L186-192 (the scan kernel; start index and order are tensor programs on the caller's side)
API int refhip_convex_sort_scan(const float* x, const float* y, const float* m, const int* start_index,
                                const int* order, int nbs, int npts, int circular, int* convex_index) {
  const int index_size = circular ? npts + 1 : npts;
  if (nbs > 0 && npts > 0)
    ref_cvx_sort::convex_sort_kernel<float><<<(nbs + 511) / 512, 512>>>(nbs, npts, index_size, circular != 0, x, y, m,
                                                                        start_index, order, convex_index);
  return ref_sync();
}

// ---- dcn_v2.py:L711-779: the per-image sampling calls of the backward loop (batch_size = 1; columns (C*kh*kw, Ho*Wo))
API int refhip_dcn2_im2col(const float* im, const float* offset, const float* mask, int C, int H, int W, int kh, int kw,
                           int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* col) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  ref_dcn2::modulated_deformable_im2col_cuda(im, offset, mask, 1, C, H, W, Ho, Wo, kh, kw, pad_h, pad_w, stride_h,
                                             stride_w, dil_h, dil_w, dg, col);
  return ref_sync();
}
API int refhip_dcn2_col2im(const float* col, const float* offset, const float* mask, int C, int H, int W, int kh, int kw,
                           int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                           float* grad_im) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  hipMemsetAsync(grad_im, 0, sizeof(float) * (size_t)C * H * W);
  ref_dcn2::modulated_deformable_col2im_cuda(col, offset, mask, 1, C, H, W, Ho, Wo, kh, kw, pad_h, pad_w, stride_h,
                                             stride_w, dil_h, dil_w, dg, grad_im);
  return ref_sync();
}
API int refhip_dcn2_col2im_coord(const float* col, const float* im, const float* offset, const float* mask, int C, int H,
                                 int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                 int dil_w, int dg, float* grad_offset, float* grad_mask) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  ref_dcn2::modulated_deformable_col2im_coord_cuda(col, im, offset, mask, 1, C, H, W, Ho, Wo, kh, kw, pad_h, pad_w,
                                                   stride_h, stride_w, dil_h, dil_w, dg, grad_offset, grad_mask);
  return ref_sync();
}

// ---- dcn_v2.py:L935-985 / L1118-1175: deformable PSRoI pooling (512 threads, at most 4096 blocks)
API int refhip_psroi_forward(const float* input, const float* bbox, const float* trans, int C, int H, int W, int R,
                             int no_trans, float spatial_scale, int output_dim, int group_size, int pooled_size,
                             int part_size, int sample_per_part, float trans_std, int channels_trans, float* out,
                             float* top_count) {
  const long out_size = (long)R * output_dim * pooled_size * pooled_size;
  const int num_classes = no_trans ? 1 : channels_trans / 2;
  const int channels_each_class = no_trans ? output_dim : output_dim / num_classes;
  const long tmp = out_size % 512L == 0 ? out_size / 512L : out_size / 512L + 1L;
  if (out_size)
    ref_ps_fwd::DeformablePSROIPoolForwardKernel<<<dim3((unsigned)std::min(tmp, 4096L)), dim3(512)>>>(
        out_size, input, spatial_scale, C, H, W, pooled_size, pooled_size, bbox, trans, no_trans, trans_std,
        sample_per_part, output_dim, group_size, part_size, num_classes, channels_each_class, out, top_count);
  return ref_sync();
}
API int refhip_psroi_backward(const float* grad_out, const float* top_count, const float* input, const float* bbox,
                              const float* trans, int N, int C, int H, int W, int R, int no_trans, float spatial_scale,
                              int output_dim, int group_size, int pooled_size, int part_size, int sample_per_part,
                              float trans_std, int channels_trans, float* grad_input, float* grad_trans) {
  const long out_size = (long)R * output_dim * pooled_size * pooled_size;
  const int num_classes = no_trans ? 1 : channels_trans / 2;
  const int channels_each_class = no_trans ? output_dim : output_dim / num_classes;
  const long tmp = out_size % 512L == 0 ? out_size / 512L : out_size / 512L + 1L;
  hipMemsetAsync(grad_input, 0, sizeof(float) * (size_t)N * C * H * W);
  if (!no_trans) hipMemsetAsync(grad_trans, 0, sizeof(float) * (size_t)R * channels_trans * part_size * part_size);
  if (out_size)
    ref_ps_bwd::DeformablePSROIPoolBackwardAccKernel<<<dim3((unsigned)std::min(tmp, 4096L)), dim3(512)>>>(
        out_size, grad_out, top_count, R, spatial_scale, C, H, W, pooled_size, pooled_size, output_dim, grad_input,
        grad_trans, input, bbox, trans, no_trans, trans_std, sample_per_part, group_size, part_size, num_classes,
        channels_each_class);
  return ref_sync();
}

// ---- nms_poly.py:L187-232: polys_sorted (n, 9) on the device, already in descending score order; keep: n host bytes
API int refhip_poly_nms(const float* polys_sorted, int n, float thr, unsigned char* keep) {
  memset(keep, 0, n);
  if (n <= 0) return 0;
  const int tpb = ref_pnms::threadsPerBlock;
  const int col_blocks = (n + tpb - 1) / tpb;
  const size_t bytes = (size_t)n * col_blocks * sizeof(unsigned long long);
  unsigned long long* mask_d = nullptr;
  if (hipMalloc((void**)&mask_d, bytes) != hipSuccess) return -1;
  hipMemset(mask_d, 0, bytes);
  ref_pnms::poly_nms_kernel<<<dim3(col_blocks, col_blocks), dim3(tpb), 0>>>(n, thr, polys_sorted, mask_d);
  int rc = ref_sync();
  std::vector<unsigned long long> mask((size_t)n * col_blocks), remv(col_blocks, 0ull);
  hipMemcpy(mask.data(), mask_d, bytes, hipMemcpyDeviceToHost);
  hipFree(mask_d);
  for (int i = 0; i < n; i++) {
    int nblock = i / tpb;
    int inblock = i % tpb;
    if (!(remv[nblock] & (1ULL << inblock))) {
      keep[i] = 1;
      unsigned long long* p = mask.data() + (size_t)i * col_blocks;
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];
    }
  }
  return rc;
}

// ---- box_iou_rotated.py:L464-485 (and the _v1 twin): 32 x 16 threads per block
#define IOU_CU_ENTRY(NS, NAME)                                                                          \
API int NAME(const float* b1, int n1, const float* b2, int n2, float* ious) {                          \
  if (n1 > 0 && n2 > 0) {                                                                               \
    dim3 blocks((n1 + NS::BLOCK_DIM_X - 1) / NS::BLOCK_DIM_X, (n2 + NS::BLOCK_DIM_Y - 1) / NS::BLOCK_DIM_Y); \
    dim3 threads(NS::BLOCK_DIM_X, NS::BLOCK_DIM_Y);                                                     \
    NS::box_iou_rotated_cuda_kernel<float><<<blocks, threads, 0>>>(n1, n2, b1, b2, ious);              \
  }                                                                                                     \
  return ref_sync();                                                                                    \
}
IOU_CU_ENTRY(ref_iou_cu, refhip_box_iou_rotated)
IOU_CU_ENTRY(ref_iou1_cu, refhip_box_iou_rotated_v1)

// ---- nms_rotated.py:L450-493 (ML_NMS_ROTATED_CUDA_SRC): dets_sorted (n, BOX_LENGTH) on the device in visiting order;
// keep_sorted: n host bytes, position i = the i-th visited box (the snippet writes keep[order[i]])
#define NMS_CU_ENTRY(NS, NAME, BL)                                                                      \
API int NAME(const float* dets_sorted, int n, float iou_threshold, unsigned char* keep_sorted) {       \
  memset(keep_sorted, 0, n);                                                                            \
  if (n <= 0) return 0;                                                                                 \
  const int tpb = NS::threadsPerBlock;                                                                  \
  const int col_blocks = (n + tpb - 1) / tpb;                                                           \
  const size_t bytes = (size_t)n * col_blocks * sizeof(unsigned long long);                             \
  unsigned long long* mask_d = nullptr;                                                                 \
  if (hipMalloc((void**)&mask_d, bytes) != hipSuccess) return -1;                                       \
  hipMemset(mask_d, 0, bytes);                                                                          \
  NS::nms_rotated_cuda_kernel<float><<<dim3(col_blocks, col_blocks), dim3(tpb), 0>>>(n, iou_threshold, \
                                                                                     dets_sorted, mask_d); \
  int rc = ref_sync();                                                                                  \
  std::vector<unsigned long long> mask((size_t)n * col_blocks), remv(col_blocks, 0ull);                 \
  hipMemcpy(mask.data(), mask_d, bytes, hipMemcpyDeviceToHost);                                         \
  hipFree(mask_d);                                                                                      \
  for (int i = 0; i < n; i++) {                                                                         \
    int nblock = i / tpb, inblock = i % tpb;                                                            \
    if (!(remv[nblock] & (1ULL << inblock))) {                                                          \
      keep_sorted[i] = 1;                                                                               \
      unsigned long long* p = mask.data() + (size_t)i * col_blocks;                                     \
      for (int j = nblock; j < col_blocks; j++) remv[j] |= p[j];                                        \
    }                                                                                                   \
  }                                                                                                     \
  return rc;                                                                                            \
}
NMS_CU_ENTRY(ref_nms5_cu, refhip_nms_rotated5, 5)
NMS_CU_ENTRY(ref_nms6_cu, refhip_nms_rotated6, 6)
'''


def inline_headers(path):
    """the `cuda_header=` string constants of a module's jt.code calls, in source order"""
    import ast
    out = []
    for node in ast.walk(ast.parse(open(path).read())):
        if isinstance(node, ast.Call):
            for kw in node.keywords:
                if kw.arg == "cuda_header" and isinstance(kw.value, ast.Constant) and isinstance(kw.value.value, str):
                    out.append((node.lineno, kw.value.value))
    return [h for _, h in sorted(out)]


def source():
    rroi = module_strings(os.path.join(OPS, "roi_align_rotated.py"))["CUDA_HEADER"]
    rroi1 = module_strings(os.path.join(OPS, "roi_align_rotated_v1.py"))["CUDA_HEADER"]
    riroi = module_strings(os.path.join(OPS, "riroi_align.py"))["riroi_cuda_head"]
    hroi = module_strings(os.path.join(OPS, "roi_align.py"))["CUDA_HEADER"]
    dcn = module_strings(os.path.join(OPS, "dcn_v1.py"))["HEADER"]
    fr = module_strings(os.path.join(OPS, "fr.py"))["HEADER"]
    csort = module_strings(os.path.join(OPS, "convex_sort.py"))["CUDA_HEAD"]
    ciou = open(os.path.join(OPS, "reppoints_convex_iou", "convex_iou_kernel.cu")).read()
    cgiou = open(os.path.join(OPS, "reppoints_convex_iou", "convex_giou_kernel.cu")).read()
    cbox = open(os.path.join(OPS, "reppoints_min_area_bbox", "min_area_bbox.cu")).read()
    pnms = module_strings(os.path.join(OPS, "nms_poly.py"))["HEADER"]
    iou_cu = module_strings(os.path.join(OPS, "box_iou_rotated.py"))["IOU_ROTATED_CUDA_HEADER"]
    iou1_cu = module_strings(os.path.join(OPS, "box_iou_rotated_v1.py"))["IOU_ROTATED_CUDA_HEADER"]
    nms_cu = module_strings(os.path.join(OPS, "nms_rotated.py"))["ML_NMS_ROTATED_CUDA_HEADER"]
    und = "#undef HOST_DEVICE\n#undef HOST_DEVICE_INLINE\n#undef CeilDIV\n#undef BOX_LENGTH\n"
    d2 = inline_headers(os.path.join(OPS, "dcn_v2.py"))      # conv forward, conv backward, pooling forward, pooling backward
    assert len(d2) == 4, len(d2)
    d2_conv = d2[1]
    for drop in (r"#include\s*<cuda_runtime.h>", r"#include\s*<cublas_v2.h>",
                 r"namespace jittor \{\s*extern cublasHandle_t cublas_handle;\s*\}\s*// jittor"):
        d2_conv, n = re.subn(drop, "", d2_conv)
        assert n == 1, drop
    parts = [PRELUDE, ns("ref_rroi", rroi), ns("ref_rroi_v1", rroi1), ns("ref_riroi", riroi),
             # roi_align.py:L217-219, L241-243: the version is a #define in front of the header
             "#define ROI_ALIGN_VERSION 0\n", ns("ref_hroi0", hroi),
             "#define ROI_ALIGN_VERSION 1\n", ns("ref_hroi1", hroi),
             ns("ref_dcn", dcn), ns("ref_fr", fr), ns("ref_cvx_sort", csort),
             ns("ref_cvx_iou", ciou.replace("using namespace std;", "")),
             ns("ref_cvx_giou", cgiou),
             ns("ref_cvx_box", cbox),
             ns("ref_dcn2", d2_conv.replace("using namespace std;", "")),
             ns("ref_ps_fwd", d2[2].replace("using namespace std;", "")),
             ns("ref_ps_bwd", d2[3].replace("using namespace std;", "")),
             "#undef THCCeilDiv\n#undef DIVUP\n", ns("ref_pnms", pnms),
             # the CUDA variants of rotated IoU / NMS (exchange-sort hull ordering, `>` suppression rule)
             ns("ref_iou_cu", iou_cu), und, ns("ref_iou1_cu", iou1_cu), und,
             "#define BOX_LENGTH 5\n", ns("ref_nms5_cu", nms_cu), und,
             "#define BOX_LENGTH 6\n", ns("ref_nms6_cu", nms_cu), und, ENTRY]
    return "\n".join(parts)


def build(verbose=True):
    if not os.path.isdir(OPS):
        raise SystemExit("reference not present at %s" % OPS)
    os.makedirs(OUT_DIR, exist_ok=True)
    src = source()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "ref_kernels.hip")
        with open(path, "w") as f:
            f.write(src)
        for out, extra in (("libjdet_ref_hip.so", ["-ffp-contract=off"]), ("libjdet_ref_hip_fma.so", [])):
            # -DNDEBUG: the IoU text asserts inside __host__ __device__ functions; re-including <cassert> inside a
            # namespace would bind them to the host's __assert_fail
            cmd = [HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
                   "-w", "-DNDEBUG"] + extra + [path, "-o", os.path.join(OUT_DIR, out)]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
            if r.returncode != 0:
                print(r.stdout[-6000:])
                raise SystemExit("hipcc failed for %s" % out)
            if verbose:
                print("built", os.path.join(OUT_DIR, out))
    return OUT_DIR


if __name__ == "__main__":
    build()
