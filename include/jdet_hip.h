/* jdet_hip.h -- C ABI of libjdet_hip.so: the MI355X (gfx950) implementation of the
 * rotated-box hot path of Jittor/JDet (reference snapshot 2025-03-10).
 *
 * The reference has no C ABI: every operator below is a `jt.code(...)` call whose C++/CUDA
 * body lives in a Python string (python/jdet/ops/<op>.py).  Each entry point here replaces
 * one such `jt.code` site; the citation on each declaration is the reference call site a
 * maintainer would rebind (see INTEGRATION.md for the ctypes stub).
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no framework types.  `stream` is a hipStream_t passed
 *     as void* (NULL = the legacy default stream).
 *   - return 0 on success, a positive hipError_t value if a launch failed, or a negative
 *     JDET_E_* code for argument errors.  Nothing is launched on error.
 *   - asynchronous: no entry point synchronises the device or allocates memory; scratch
 *     comes from the caller (see the *_workspace queries).
 *   - all floating-point tensors are fp32 (the reference is fp32 everywhere); boxes are
 *     [xc, yc, w, h, theta(rad)]; RoIs are [batch, xc, yc, w, h, theta] (rotated) or
 *     [batch, x1, y1, x2, y2] (horizontal).
 *   - inputs are never written; outputs are fully overwritten (backward kernels zero-fill
 *     their accumulation target themselves, as the reference's cudaMemsetAsync does).
 */
#ifndef JDET_HIP_H_
#define JDET_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* jdet_stream_t; /* hipStream_t */

#define JDET_OK 0
#define JDET_E_BADARG (-1)   /* null pointer, negative size, unknown enum value        */
#define JDET_E_UNSUPPORTED (-2) /* shape outside what the kernels handle (documented below) */
#define JDET_E_WORKSPACE (-3) /* workspace too small                                     */

/* RoIAlign dialects */
#define JDET_ROI_ROTATED 0    /* ROIAlignRotated     ops/roi_align_rotated.py:L61-127,L165-255    */
#define JDET_ROI_ROTATED_V1 1 /* ROIAlignRotated_v1  ops/roi_align_rotated_v1.py:L71-145,L193-298 */
#define JDET_ROI_RIROI 2      /* RiRoIAlign          ops/riroi_align.py:L70-163,L228-358          */
#define JDET_ROI_HBB_V0 3     /* ROIAlign version=0  ops/roi_align.py:L93-204                     */
#define JDET_ROI_HBB_V1 4     /* ROIAlign version=1  (same file, ROI_ALIGN_VERSION 1)             */

/* Feature-map memory layouts.  NHWC ("channels last") is the native layout of the kernels:
 * one bilinear tap is a contiguous C-vector.  NCHW inputs go through jdet_nchw_to_nhwc. */
#define JDET_LAYOUT_NCHW 0
#define JDET_LAYOUT_NHWC 1

int jdet_version(void);

/* Layout conversion helpers (tiled transposes), x: (N,C,H,W) <-> y: (N,H,W,C). */
int jdet_nchw_to_nhwc(const float* x, int N, int C, int H, int W, float* y, jdet_stream_t stream);
int jdet_nhwc_to_nchw(const float* x, int N, int C, int H, int W, float* y, jdet_stream_t stream);

/* RoIAlign forward.  Replaces the jt.code sites roi_align_rotated.py:L265-283,
 * roi_align_rotated_v1.py:L308-326, riroi_align.py:L425-427, roi_align.py:L217-237.
 *   feat    : (N, C, H, W) values stored NHWC, i.e. feat[((n*H+y)*W+x)*C + c]   [layout NHWC]
 *             C is the TOTAL plane count (RiRoIAlign: C = channels * n_orient).
 *   rois    : (R, 6) rotated dialects, (R, 5) horizontal dialects
 *   out     : (R, C, PH, PW) contiguous (the reference's output layout)
 *   sample_num : >0 fixed grid, <=0 adaptive ceil(roi_size / pooled_size)
 *   n_orient   : RiRoIAlign only (C % n_orient == 0); pass 1 otherwise
 *   order      : optional (R) int32 permutation from jdet_roi_spatial_order, or NULL
 * A RoI whose batch index is negative is skipped: its rows of `out` are left untouched and it adds
 * nothing in backward.  (FPN level routing: call once per level on the SAME `out`, with the batch
 * index of off-level RoIs set to -1 -- no mask / gather / scatter-add round trip and no host sync,
 * cf. oriented_single_level.py:L105-112.)
 * Limits: PH*PW <= 256; any C >= 1 (C % 4 == 0 takes the vector path). */
int jdet_roi_align_forward(int variant, const float* feat_nhwc, int N, int C, int H, int W,
                           const float* rois, int R, int PH, int PW, float spatial_scale,
                           int sample_num, int n_orient, const int32_t* order, float* out,
                           jdet_stream_t stream);

/* RoI-stationary forward (the kernels of jdet_roi_align_forward) with the channels-last result layout
 * out_cl (R, PH, PW, C): each wave stores a bin's channel chunk straight from registers (1 KiB contiguous,
 * non-temporal) instead of transposing the RoI's block through LDS.  Rotated v0 / v1, horizontal v0 / v1 and
 * RiRoIAlign with n_orient 4 or 8 (ignored for the other dialects), C % 4 == 0, any sample_num (<= 0 adaptive);
 * otherwise JDET_E_UNSUPPORTED.  `order` as in jdet_roi_align_forward. */
int jdet_roi_align_forward_cl_roi(int variant, const float* feat_nhwc, int N, int C, int H, int W,
                                  const float* rois, int R, int PH, int PW, float spatial_scale, int sample_num,
                                  int n_orient, const int32_t* order, float* out_cl, jdet_stream_t stream);

/* Forward with the channels-last result out_cl (R, PH, PW, C) and the schedule computed inside; same call sites as
 * jdet_roi_align_forward (roi_align_rotated.py:L265-283, roi_align_rotated_v1.py:L308-326, riroi_align.py:L425-427,
 * roi_align.py:L217-237): jdet_roi_spatial_order (R >= 64) + the kernels of jdet_roi_align_forward_cl_roi.
 * workspace: jdet_roi_align_forward_cl_workspace(R, PH, PW) bytes (8 R + 256 for the schedule), any content, 256-byte
 * aligned; too small a buffer returns JDET_E_WORKSPACE.
 * RoIs with a negative batch index are skipped as in jdet_roi_align_forward. */
size_t jdet_roi_align_forward_cl_workspace(int R, int PH, int PW);
int jdet_roi_align_forward_cl(int variant, const float* feat_nhwc, int N, int C, int H, int W, const float* rois,
                              int R, int PH, int PW, float spatial_scale, int sample_num, int n_orient,
                              float* out_cl, void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* The reference's OPERATION ORDER (roi_align_rotated.py:L70-118: w1*lt + w2*rt + w3*lb + w4*rb per sample, samples summed
 * iy-major, then / count) behind its own entry points -- same arguments as jdet_roi_align_forward /
 * jdet_roi_align_forward_cl, results bit-identical to the CPU oracle: the parity twin the tests and smoke() call.  The
 * product entry points above merge duplicate taps inside a bin before loading (fewer vector-memory requests; equal to
 * the reference up to fp32 re-association of the bilinear weights, <= 2e-6 on N(0,1) maps).  The library holds no
 * process-wide arithmetic mode. */
int jdet_roi_align_forward_reference(int variant, const float* feat_nhwc, int N, int C, int H, int W,
                                     const float* rois, int R, int PH, int PW, float spatial_scale, int sample_num,
                                     int n_orient, const int32_t* order, float* out, jdet_stream_t stream);
int jdet_roi_align_forward_cl_reference(int variant, const float* feat_nhwc, int N, int C, int H, int W,
                                        const float* rois, int R, int PH, int PW, float spatial_scale, int sample_num,
                                        int n_orient, float* out_cl, void* workspace, size_t workspace_bytes,
                                        jdet_stream_t stream);

/* XCD-aware spatial schedule for the RoIAlign kernels (no reference counterpart: the reference
 * processes output elements in index order).  Writes a permutation `order` of [0,R): workgroup b
 * processes RoI order[b].  RoIs are bucketed by the Morton code of their centre and contiguous
 * runs are dealt to the 8 XCDs so that each XCD's private L2 sweeps one compact region of the
 * map.  Pure performance hint: any permutation (or NULL = identity) gives identical results.
 * `workspace`: R int32 of scratch.  roi_cols 6 (rotated) or 5 (horizontal). */
int jdet_roi_spatial_order(const float* rois, int R, int roi_cols, float spatial_scale, int N,
                           int H, int W, int32_t* order, int32_t* workspace, jdet_stream_t stream);

/* RoIAlign backward w.r.t. the feature map.  Replaces roi_align_rotated.py:L286-307 (and the
 * _v1 / riroi / hbb twins).  grad_in_nhwc (N,H,W,C) is fully overwritten.
 * With a workspace of jdet_roi_align_backward_workspace(...) bytes the scatter is inverted once on the
 * (roi, sample, tap) index space and executed as a sorted GATHER (integer atomics only, one store per
 * pixel); with workspace = NULL, or when the query returns 0 (RiRoIAlign, sample_num <= 0, C % 4 != 0),
 * it is the reference's scheme: zero-fill + hardware fp32 atomics.  Either way the last bits depend on
 * accumulation order, as in the reference.
 * Workspace size: counters and offsets of the 2x2 pixel patches, two record arrays of R * PH * PW * samples * 4 x 32
 * bytes, the (R, PH*PW, C) transposed gradient, and the patches' direct rows (at most 256 MiB; csr_gather.h) -- 268 MB
 * at 2000 RoIs on a 256 x 256 x 256 map.  The size grows monotonically with R, so a buffer sized for the largest R of a
 * map serves every smaller call on it. */
size_t jdet_roi_align_backward_workspace(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                         int sample_num);
int jdet_roi_align_backward(int variant, const float* grad_out, const float* rois, int R, int N,
                            int C, int H, int W, int PH, int PW, float spatial_scale,
                            int sample_num, int n_orient, const int32_t* order, float* grad_in_nhwc,
                            void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* Same as jdet_roi_align_backward for a channels-last gradient grad_out_cl (R, PH, PW, C): the sorted gather
 * reads it directly (no (R,C,bin) -> (R,bin,C) transpose pass).  Needs the workspace of
 * jdet_roi_align_backward_workspace(); returns JDET_E_UNSUPPORTED where that query returns 0.
 * workspace_clean != 0: the caller keeps this workspace between calls and guarantees that its first
 * jdet_roi_align_backward_clean_bytes() bytes are zero on entry (zero-fill it once); the call hands them back zeroed
 * (stream order), so no memset launch is needed.  workspace_clean == 0: any content, the call zeroes what it needs. */
size_t jdet_roi_align_backward_clean_bytes(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                           int sample_num);
int jdet_roi_align_backward_cl(int variant, const float* grad_out_cl, const float* rois, int R, int N, int C,
                               int H, int W, int PH, int PW, float spatial_scale, int sample_num, int n_orient,
                               float* grad_in_nhwc, void* workspace, size_t workspace_bytes, int workspace_clean,
                               jdet_stream_t stream);

/* The PLAN of a backward, separated from the gather (round 6).  jdet_roi_align_backward[_cl] inverts the scatter on every
 * call (a third of its time at the north-star point) although the inversion depends only on the RoIs, the map size and
 * the bin grid -- all known at the forward (ROIAlignBackward's arguments besides the gradient, roi_align_rotated.py:L284-308).
 *   jdet_roi_align_backward_plan       builds the plan into `plan` (jdet_roi_align_backward_plan_bytes() bytes, any
 *                                      content on entry; 0 bytes = shape served by the atomic path only)
 *   jdet_roi_align_backward_cl_planned the gather alone from a channels-last gradient (R, PH, PW, C); the plan is left
 *                                      intact: any number of gathers (any C) per plan.  RiRoIAlign: JDET_E_UNSUPPORTED
 *                                      (its rows need the orientation mix: use jdet_roi_align_backward_cl).
 * Same values as jdet_roi_align_backward_cl (same rows, same order of accumulation). */
size_t jdet_roi_align_backward_plan_bytes(int variant, int R, int N, int H, int W, int PH, int PW, int sample_num);
int jdet_roi_align_backward_plan(int variant, const float* rois, int R, int N, int H, int W, int PH, int PW,
                                 float spatial_scale, int sample_num, void* plan, size_t plan_bytes,
                                 jdet_stream_t stream);
int jdet_roi_align_backward_cl_planned(int variant, const float* grad_out_cl, int R, int N, int C, int H, int W, int PH,
                                       int PW, int sample_num, float* grad_in_nhwc, const void* plan, size_t plan_bytes,
                                       jdet_stream_t stream);

/* Pairwise rotated IoU, ious (n1, n2) row-major.  Replaces box_iou_rotated.py:L507 and
 * box_iou_rotated_v1.py:L512 (the python-side "too small" zeroing L515-523 stays in the
 * host wrapper).  version 0/1 selects the vertex convention; sort_mode 0 reproduces the
 * reference CPU path (std::sort, L316-325), 1 the reference CUDA exchange sort (L338-351).
 * stride = floats per box row (>= 5). */
int jdet_box_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, int stride,
                         int version, int sort_mode, float* ious, jdet_stream_t stream);

/* Rotated NMS.  Replaces nms_rotated.py:L497-503 (nms_rotated_cpu) / L506-513 (cuda).
 *   dets (n, box_len) box_len 5, or 6 with a label in column 5 (cross-label IoU := 0, L283-286)
 *   order: int32 visiting order = indices by descending score (the caller's argsort, nms_rotated.py:L519,L532).
 *          With labels (box_len 6) any order that is descending in score INSIDE each label gives the same keep
 *          set; visiting class by class lets the kernel skip every 64x64 tile whose label ranges are disjoint.
 *   cmp_ge 1: suppress when iou >= thr (reference CPU rule L444); 0: iou > thr (CUDA rule L403)
 *   keep : n bytes, 1 = kept, indexed by ORIGINAL detection index
 * Entirely on device: zero fill + tile bitmask kernel + one-workgroup greedy scan, no host synchronisation. */
size_t jdet_nms_rotated_workspace(int n);
int jdet_nms_rotated(const float* dets, int n, int box_len, const int32_t* order,
                     float iou_threshold, int cmp_ge, int sort_mode, uint8_t* keep,
                     void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* jdet_nms_rotated with two more switches (same workspace, same keep contract).
 * horizontal != 0: the caller promises every angle is 0; the overlap is the rectangle formula
 *   inter / (a + b - inter) instead of polygon clipping (the horizontal proposal NMS of the two-stage RPNs --
 *   `jt.nms`, nms.py:L4-9 -- and of level-offset tricks such as oriented_rpn_head.py:L214-219).
 * n_labels > 1 (box_len 6): the labels are the integers 0 .. n_labels-1 and `order` visits the boxes label by label
 *   (descending score inside a label): each label is scanned by its own workgroup.  n_labels == 1: any labels. */
int jdet_nms_labeled(const float* dets, int n, int box_len, const int32_t* order, float iou_threshold, int cmp_ge,
                     int sort_mode, int horizontal, int n_labels, uint8_t* keep, void* workspace,
                     size_t workspace_bytes, jdet_stream_t stream);

/* Deformable-conv v1 sampling.  Replace dcn_v1.py:L309-338 (im2col), L374-410 (col2im),
 * L340-372 (col2im_coord).  im (B,C,H,W) NCHW; offset (B, dg*2*kh*kw, Ho, Wo) ordered
 * (dy,dx) per tap; col (C*kh*kw, B, Ho, Wo). */
int jdet_deform_im2col(const float* im, const float* offset, int B, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                       int dil_w, int deform_groups, float* col, jdet_stream_t stream);
int jdet_deform_col2im(const float* col, const float* offset, int B, int C, int H, int W, int kh,
                       int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                       int dil_w, int deform_groups, float* grad_im, jdet_stream_t stream);
int jdet_deform_col2im_coord(const float* col, const float* im, const float* offset, int B, int C,
                             int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                             int stride_w, int dil_h, int dil_w, int deform_groups,
                             float* grad_offset, jdet_stream_t stream);

/* Modulated deformable conv (DCN v2) sampling.  Replace modulated_deformable_im2col / _col2im / _col2im_coord of
 * ops/dcn_v2.py:L86-149, L506-558, L560-627.  Layouts as jdet_deform_* plus mask (B, dg*kh*kw, Ho, Wo): the column
 * element, the column gradient and the offset gradient are multiplied by the tap's mask; grad_mask (same shape as
 * mask) = sum over the group's channels of column gradient x unmasked bilinear sample. */
int jdet_modulated_deform_im2col(const float* im, const float* offset, const float* mask, int B, int C, int H,
                                 int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                 int dil_w, int deform_groups, float* col, jdet_stream_t stream);
int jdet_modulated_deform_col2im(const float* col, const float* offset, const float* mask, int B, int C, int H,
                                 int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                 int dil_w, int deform_groups, float* grad_im, jdet_stream_t stream);
int jdet_modulated_deform_col2im_coord(const float* col, const float* im, const float* offset, const float* mask,
                                       int B, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                       int stride_h, int stride_w, int dil_h, int dil_w, int deform_groups,
                                       float* grad_offset, float* grad_mask, jdet_stream_t stream);

/* Deformable position-sensitive RoI pooling.  Replace DeformablePSROIPoolForwardKernel / ...BackwardAccKernel of
 * ops/dcn_v2.py:L855-932, L1007-1116.  CHANNELS-LAST memory (round 6): input / grad_input (N, H, W, C) with
 * C = output_dim * group_size^2 [the reference: (N,C,H,W)]; rois (R,5) [batch,x1,y1,x2,y2]; trans
 * (R, trans_channels, part, part) or NULL with no_trans; out / top_count / grad_out (R, P, P, output_dim) [the reference's
 * (R, output_dim, P, P) stored channels-last] (top_count = samples counted per bin, consumed by the backward).  One wave
 * per (RoI, bin, class, channel chunk), output channels across the lanes.  The backward zero-fills grad_input and
 * grad_trans (shape of trans); grad_input collects lane-contiguous fp32 atomics, grad_trans two atomics per wave. */
int jdet_deform_psroi_pool_forward(const float* input, const float* rois, const float* trans, int N, int C, int H,
                                   int W, int R, int no_trans, float spatial_scale, int output_dim, int group_size,
                                   int pooled_size, int part_size, int sample_per_part, float trans_std,
                                   int trans_channels, float* out, float* top_count, jdet_stream_t stream);
int jdet_deform_psroi_pool_backward(const float* grad_out, const float* top_count, const float* input,
                                    const float* rois, const float* trans, int N, int C, int H, int W, int R,
                                    int no_trans, float spatial_scale, int output_dim, int group_size,
                                    int pooled_size, int part_size, int sample_per_part, float trans_std,
                                    int trans_channels, float* grad_input, float* grad_trans, jdet_stream_t stream);

/* Channels-last deformable sampling (groups = 1, deform_groups = 1, C % 4 == 0; else JDET_E_UNSUPPORTED).
 * Same arithmetic as jdet_deform_im2col / jdet_deform_col2im (dcn_v1.py:L130-184, L185-241), different
 * layout: x_nhwc (B,H,W,C); offset stays (B, 2*kh*kw, Ho, Wo); cols / grad_cols are
 * (B*Ho*Wo, kh*kw, C) so that  out_nhwc = cols . W^T  and  grad_cols = grad_out_nhwc . W  are plain
 * row-major GEMMs.  col2im is a sorted gather (no fp atomics); workspace size from the _workspace query. */
int jdet_deform_im2col_nhwc(const float* x_nhwc, const float* offset, int B, int C, int H, int W,
                            int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                            int dil_h, int dil_w, float* cols, jdet_stream_t stream);
size_t jdet_deform_col2im_nhwc_workspace(int B, int C, int H, int W, int kh, int kw, int pad_h,
                                         int pad_w, int stride_h, int stride_w, int dil_h, int dil_w);
int jdet_deform_col2im_nhwc(const float* grad_cols, const float* offset, int B, int C, int H, int W,
                            int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                            int dil_h, int dil_w, float* grad_x_nhwc, void* workspace,
                            size_t workspace_bytes, jdet_stream_t stream);

/* 3x3 / stride 1 / pad 1 convolution as an fp32-MFMA implicit GEMM, channels-last, with the epilogue of the dense
 * stack fused: replaces nn.Conv(3x3) [+ bias] [+ ReLU] of ConvModule (models/utils/modules.py:L91-175) as the head
 * towers / FPN output convs / the RPN conv use it, and -- with `offset` non-NULL -- the whole DeformConv forward of
 * ops/dcn_v1.py:L412-454 (deformable im2col L130-184 + matmul) without a column matrix.
 * x (N,H,W,Cin); w (Cout,3,3,Cin) [= torch channels_last memory of a (Cout,Cin,3,3) weight]; bias (Cout) or NULL;
 * rowmask (N*H*W) or NULL multiplies the finished output rows (gap rows of a level pack); offset (N,18,H,W) with
 * (dy,dx) per tap or NULL; y (N,H,W,Cout); tile = 0 (chosen from the problem size), 64 or 128 (edge of the
 * workgroup's output tile; + 1 / + 2 select the 16-deep K step / no intra-workgroup K split, for measurements).  Cin % 16 == 0 and 16-byte aligned x / w, else JDET_E_UNSUPPORTED / JDET_E_BADARG
 * (query: jdet_conv3x3_igemm_supported).  Products are fp32 in, fp32 accumulate (v_mfma_f32_32x32x2). */
int jdet_conv3x3_igemm_supported(int Cin, int Cout);
int jdet_conv3x3_igemm_forward(const float* x_nhwc, int N, int H, int W, int Cin, const float* w_krsc, int Cout,
                               const float* bias, int relu, const float* rowmask, const float* offset,
                               int tile, float* y_nhwc, void* workspace, size_t workspace_bytes,
                               jdet_stream_t stream);
/* optional workspace: maps too small to fill the chip with output tiles are split along K over workgroups (partial
 * tiles summed by a second launch that also applies the epilogue) when a workspace of this size is passed; NULL / 0 =
 * single pass.  0 when the shape is not split. */
size_t jdet_conv3x3_igemm_workspace(int N, int H, int W, int Cin, int Cout);

/* Weight gradient of the same convolution, ACCUMULATED into gw: gw (Cout,3,3,Cin) += sum over positions of
 * gy (N,H,W,Cout) x the (shifted | bilinearly gathered, offset non-NULL) input x (N,H,W,Cin).  Replaces the weight
 * gradient Jittor derives for nn.Conv inside ConvModule (models/utils/modules.py:L91-175) and DeformConvFunction.grad's
 * im2col + matmul into grad_weight (ops/dcn_v1.py:L508-556).  The reduction over positions is split over workgroups
 * and meets in gw by float atomics, so several calls (the pyramid levels of a shared tower) may target one buffer --
 * the caller zero-fills it once (or passes p.grad).  ksplit = 0 chooses the split.  Cin % 4 == 0, Cout % 4 == 0,
 * 16-byte aligned x / gy, else JDET_E_UNSUPPORTED / JDET_E_BADARG. */
int jdet_conv3x3_wgrad_supported(int Cin, int Cout);
int jdet_conv3x3_wgrad(const float* x_nhwc, const float* gy_nhwc, const float* offset, int N, int H, int W, int Cin,
                       int Cout, float* gw_krsc, int ksplit, jdet_stream_t stream);

/* ---- The ResNet bottleneck convolutions with the layer's neighbours in the epilogue (csrc/conv_bn.hip) -------------
 * Replaces, for Bottleneck.execute (python/jdet/models/backbones/resnet.py:L61-93) and ResNet._make_layer's downsample
 * pair (L131-154) under `norm_eval` (L177-185), the nn.Conv -> nn.BatchNorm(eval) [-> + identity] [-> relu] chains and
 * the gradients Jittor derives for them: forward, data gradient (the same kernel on jdet_conv_dgrad_weights' flipped /
 * transposed weights) and, through jdet_conv_wgrad, the weight gradient.
 *
 * jdet_bn_params_t: an eval-mode BatchNorm as the affine map a = weight * rsqrt(var + eps), sh = bias - mean * a
 *   (weight NULL = 1, bias NULL = 0; var NULL: a = weight, sh = bias, i.e. a plain convolution bias).
 * jdet_conv_epilogue_t.mode:
 *   JDET_EPI_FORWARD  y = [relu]( [affine: acc * a + sh] [+ residual (M, Cout)] )
 *   JDET_EPI_ADD      y = acc + grad_out * [act > 0]        grad_out, act: (M, Cout) -- the identity branch's gradient
 *                                                            joins the data gradient of conv1 (the block's grad_x)
 *   JDET_EPI_MASK     g = acc * [act > 0];  y = g * a;  sums (rows, 2, Cout) non-NULL: partial column sums of g and of
 *                     g * (act - bias), rows = jdet_conv_bn_sums_rows(...) -- `bn` / `act` are the BatchNorm and the
 *                     activation act = relu(bn(conv)) of the layer BELOW: y is the gradient w.r.t. that conv's output */
typedef struct jdet_bn_params {
  const float* weight;
  const float* bias;
  const float* mean;
  const float* var;
  float eps;
} jdet_bn_params_t;

#define JDET_EPI_FORWARD 0
#define JDET_EPI_ADD 1
#define JDET_EPI_MASK 2

typedef struct jdet_conv_epilogue {
  int mode;
  int affine; /* FORWARD: apply bn to the accumulator */
  int relu;   /* FORWARD */
  jdet_bn_params_t bn;
  const float* residual; /* FORWARD */
  const float* grad_out; /* ADD */
  const float* act;      /* ADD, MASK */
  float* sums;           /* MASK */
} jdet_conv_epilogue_t;

/* x (N,H,W,Cin), w (Cout,R,R,Cin), y (N,Ho,Wo,Cout), Ho = (H + 2*(R/2) - R) / stride + 1; R in {1, 3}, stride in
 * {1, 2}, Cin % 16 == 0, 16-byte aligned x / w, positions * channels < 2^30 (else JDET_E_UNSUPPORTED / _BADARG).
 * tile: as jdet_conv3x3_igemm_forward.  workspace (jdet_conv_bn_workspace bytes, optional): small maps split their K
 * steps over workgroups.  jdet_conv_bn_sums_rows: rows of `sums` a MASK launch with these arguments writes
 * (with_workspace: whether a sufficient workspace will be passed). */
int jdet_conv_bn_supported(int Cin, int Cout, int R, int stride);
size_t jdet_conv_bn_workspace(int N, int H, int W, int Cin, int Cout, int R, int stride);
size_t jdet_conv_bn_sums_rows(int N, int H, int W, int Cin, int Cout, int R, int stride, int tile,
                              int with_workspace);
int jdet_conv_bn_forward(const float* x_nhwc, int N, int H, int W, int Cin, const float* w_krsc, int Cout, int R,
                         int stride, const jdet_conv_epilogue_t* epilogue, int tile, float* y_nhwc, void* workspace,
                         size_t workspace_bytes, jdet_stream_t stream);
/* weights of the data gradient for any number of layers in ONE launch: dst (Cin,R,R,Cout)[ci][R*R-1-tap][co] =
 * src (Cout,R,R,Cin)[co][tap][ci].  jobs_device: DEVICE array of njobs 32-byte records {const float* src; float* dst;
 * int Cout, Cin, taps, tile_begin} with tile_begin the running sum of ceil(Cout/32) * ceil(Cin/32) * taps over the
 * preceding jobs; total_tiles = that sum over all jobs. */
int jdet_conv_dgrad_weights(const void* jobs_device, int njobs, int total_tiles, jdet_stream_t stream);
/* weight gradient, ACCUMULATED: gw (Cout,R,R,Cin) += sum over positions of gy (N,Ho,Wo,Cout) x the input window of
 * x (N,H,W,Cin); the general (R, stride) form of jdet_conv3x3_wgrad, same tiling / split / atomics. */
int jdet_conv_wgrad(const float* x_nhwc, const float* gy_nhwc, int N, int H, int W, int Cin, int Cout, int R,
                    int stride, float* gw_krsc, int ksplit, jdet_stream_t stream);
/* Backward of a BatchNorm whose (possibly summed) output went through a ReLU, from the ACTIVATION y (the fused forward
 * stores no conv output c): grad_c = grad_y * [y > 0] * a; sums non-NULL: partial sums (rows, 2, C) of
 * g = grad_y * [y > 0] and g * t, rows = jdet_bn_act_backward_from_output_rows(P, C), where t / gamma = xhat:
 *   t = y - bias (y = relu(bn(c)));  identity != NULL: t = y - identity - bias (y = relu(bn(c) + identity));
 *   own_output != NULL: t = own_output - bias (y = relu(other + bn(c)), own_output = bn(c): the downsample branch).
 * Channel counts as jdet_frozen_bn_act_forward. */
size_t jdet_bn_act_backward_from_output_rows(long P, int C);
int jdet_bn_act_backward_from_output(const float* grad_y_nhwc, const float* y_nhwc, const float* identity_nhwc,
                                     const float* own_output_nhwc, long P, int C, const float* weight,
                                     const float* bias, const float* running_mean, const float* running_var,
                                     float eps, float* grad_c_nhwc, float* sums, size_t sums_bytes,
                                     jdet_stream_t stream);
/* Second stage of such partial sums for up to 4 BatchNorm layers in one launch (deterministic: fixed summation order):
 * grad_beta = column sums of the first halves, grad_gamma = column sums of the second halves / gamma (gamma NULL = 1;
 * a zero gamma gives inf / NaN on purpose -- xhat is not recoverable from the activation then).  jobs: HOST array. */
typedef struct jdet_bn_sums_job {
  const float* partial; /* (rows, 2, C) */
  long rows;
  int C;
  const float* gamma;
  float* grad_gamma;
  float* grad_beta;
} jdet_bn_sums_job_t;
int jdet_bn_sums_finish(const jdet_bn_sums_job_t* jobs, int njobs, jdet_stream_t stream);

/* RepPoints geometry on 9-point sets (pointsets (N,18) = 9 (x,y) pairs) and the Graham scan of the polygon-IoU loss.
 * jdet_convex_iou: replaces convex_iou_kernel (ops/reppoints_convex_iou/convex_iou_kernel.cu:L258-305): IoU of the
 *   convex hull of every point set with every quadrilateral polygons (M,8) -> ious (N,M), computed in double.
 * jdet_min_area_bbox: replaces minareabbox_kernel (ops/reppoints_min_area_bbox/min_area_bbox.cu:L49-203, L301-461):
 *   minimum-area rectangle of the hull -> bboxes (N,8), corners (xmax,ymin) (xmin,ymin) (xmin,ymax) (xmax,ymax) of the
 *   rotated frame.
 * jdet_convex_sort: replaces convex_sort_gpu (ops/convex_sort.py:L4-65, L159-194): pts (nbs,npts,2), masks (nbs,npts)
 *   as float 0/1 -> index (nbs, npts + circular) int32 hull indices in scan order, -1 in unused slots (the kernel
 *   writes every slot); npts <= 64, else JDET_E_UNSUPPORTED. */
int jdet_convex_iou(const float* pointsets, int N, const float* polygons, int M, float* ious, jdet_stream_t stream);

/* RepPoints convex GIoU with its gradient (reppoints_convex_iou/convex_giou.py:L29-47, replaces the jt.code site L37-43
 * around convex_giou_kernel.cu:L725-821): ALIGNED pairs -- pointsets (N, 18), polygons (N, 8) -> out (N, 19):
 * out[n, 0:18] = d giou / d pointsets[n, :] (zero for points that are not hull vertices), out[n, 18] = giou =
 * I/U - (C - U)/C of the hull of the 9 points with the quadrilateral (C: hull of both).  Gradient by forward-mode dual
 * numbers (half a wave per pair, one lane per coordinate), double precision inside like the reference. */
int jdet_convex_giou(const float* pointsets, const float* polygons, int N, float* out, jdet_stream_t stream);
int jdet_min_area_bbox(const float* pointsets, int N, float* bboxes, jdet_stream_t stream);
int jdet_convex_sort(const float* pts, const float* masks, int nbs, int npts, int circular, int32_t* index,
                     jdet_stream_t stream);

/* Sigmoid focal loss, replaces the tensor-op chain of models/losses/focal_loss.py:L5-96 (sigmoid_focal_loss with
 * binary_cross_entropy_with_logits): logits (M, C) row-major, labels (M) int32 (0 = background, k = class k),
 * weight (M) or NULL; alpha < 0 disables the alpha term.  *loss_sum = sum over all M*C elements (the caller divides
 * by avg_factor and applies loss_weight); grad_logits (M, C) = d loss_sum / d logits.  Deterministic. */
size_t jdet_sigmoid_focal_loss_workspace(void);
int jdet_sigmoid_focal_loss(const float* logits, const int32_t* labels, const float* weight, long M, int C,
                            float alpha, float gamma, float* loss_sum, float* grad_logits,
                            void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* Weighted smooth-L1 / L1 (models/losses/smooth_l1_loss.py:L5-27, l1_loss.py): element-wise over n values,
 * weight same shape or NULL, beta = 0 -> L1.  *loss_sum = sum; grad_pred = d loss_sum / d pred.  Workspace: the
 * focal-loss query.  Deterministic. */
int jdet_smooth_l1_loss(const float* pred, const float* target, const float* weight, long n, float beta,
                        float* loss_sum, float* grad_pred, void* workspace, size_t workspace_bytes,
                        jdet_stream_t stream);

/* The per-level loss call of a dense head as one node (round 6): what models/roi_heads/s2anet_head.py:L430-508
 * (loss_fam_single / loss_odm_single) composes per pyramid level from reshape + loss + `/ avg_factor` + `* loss_weight`
 * (models/losses/focal_loss.py:L50-53, L86-93; smooth_l1_loss.py:L18-22, L47-53).
 *   - labels / weight / target are read through a BLOCKED row index: logical row r is element
 *     (r / rows_per_block) * block_stride + r % rows_per_block (in rows) of the array -- a level's column window
 *     [:, s:e] of the per-image (N, A[, 5]) target arrays is rows_per_block = e - s, block_stride = A, base + s; a
 *     contiguous array is rows_per_block >= M, any stride.  logits / pred and the gradients are contiguous.
 *   - avg_factor: DEVICE scalar; *loss = (sum / *avg_factor) * loss_weight; grad_* = d sum / d input (the UNIT gradient).
 *   - jdet_loss_grad_scale: out[i] = unit_grad[i] * ((*grad_out * loss_weight) / *avg_factor): the backward of the node.
 * Same operations in the same order as the composition: bit-identical to it.  Workspace: the focal-loss query. */
int jdet_sigmoid_focal_loss_level(const float* logits, const int32_t* labels, long label_rows_per_block,
                                  long label_block_stride, const float* weight, long weight_rows_per_block,
                                  long weight_block_stride, long M, int C, float alpha, float gamma,
                                  const float* avg_factor, float loss_weight, float* loss, float* grad_logits,
                                  void* workspace, size_t workspace_bytes, jdet_stream_t stream);
int jdet_smooth_l1_loss_level(const float* pred, const float* target, long target_rows_per_block,
                              long target_block_stride, const float* weight, long weight_rows_per_block,
                              long weight_block_stride, long rows, int E, float beta, const float* avg_factor,
                              float loss_weight, float* loss, float* grad_pred, void* workspace,
                              size_t workspace_bytes, jdet_stream_t stream);
int jdet_loss_grad_scale(const float* unit_grad, long n, const float* grad_out, const float* avg_factor,
                         float loss_weight, float* out, jdet_stream_t stream);

/* Head glue as a single pass (csrc/level_pack.hip; round 6).
 * jdet_level_pack_nhwc: the small pyramid levels of a weight-shared tower (S2ANetHead.execute runs its towers per level,
 *   models/roi_heads/s2anet_head.py:L207-252; here they run once on a packed canvas) -- levels[l] (N, h_l, w_l, C)
 *   contiguous DEVICE pointers in a HOST array (NULL = a window of zeros), level_hw / level_place HOST arrays of
 *   (h, w) / (row, col) per level; every word of canvas (N, Hp, Wp, C) is written: the level's value inside its window,
 *   0 in the gaps.  num_levels <= 8, C % 4 == 0.  Also the backward of the unpacking (level gradients -> canvas gradient). */
int jdet_level_pack_nhwc(const float* const* levels, const int32_t* level_hw, const int32_t* level_place, int num_levels,
                         int N, int C, int Hp, int Wp, float* canvas, jdet_stream_t stream);

/* AlignConv.get_offset (models/roi_heads/s2anet_head.py:L676-713): anchors (N, H*W, 5) [xc,yc,w,h,theta] in image
 * coordinates -> offset (N, 2*k*k, H, W), (dy, dx) per tap of the k x k kernel (k odd). */
int jdet_align_conv_offset(const float* anchors, int N, int H, int W, float stride, int kernel_size,
                           float* offset, jdet_stream_t stream);

/* Inference-mode ("frozen statistics") BatchNorm fused with ReLU and the residual add, channels-last.
 * Replaces the nn.BatchNorm (eval) -> (+identity) -> relu chains of the backbone (models/backbones/resnet.py:
 * L33-59 BasicBlock, L61-93 Bottleneck, L177-185 norm_eval) -- framework primitives in the reference, not jt.code.
 *   y = act((x - mean) * rsqrt(var + eps) * weight + bias (+ residual)),  tensors [P = N*H*W][C], C % 4 == 0,
 *   C/4 a divisor of 256 or in (256, 1024]; weight / bias may be NULL (1 / 0); relu: 0 | 1.
 * backward: grad_x always; grad_residual (= masked grad_y) if non-NULL; grad_weight / grad_bias if non-NULL (then
 * x_nhwc and a workspace from the _workspace query are required); y_nhwc is required when relu = 1. */
int jdet_frozen_bn_act_forward(const float* x_nhwc, const float* residual_nhwc, long P, int C,
                               const float* weight, const float* bias, const float* running_mean,
                               const float* running_var, float eps, int relu, float* y_nhwc,
                               jdet_stream_t stream);
size_t jdet_frozen_bn_act_backward_workspace(long P, int C);
int jdet_frozen_bn_act_backward(const float* grad_y_nhwc, const float* y_nhwc, const float* x_nhwc, long P,
                                int C, const float* weight, const float* bias, const float* running_mean,
                                const float* running_var, float eps, int relu, float* grad_x_nhwc,
                                float* grad_residual_nhwc, float* grad_weight, float* grad_bias,
                                void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* Bias [+ ReLU] backward of a convolution, channels-last rows (P, C): grad_pre = grad_y * [y > 0] when relu != 0
 * (without ReLU the gradient passes unchanged and grad_pre / y may be NULL) and grad_bias (C) = sum over the P rows of
 * grad_pre.  Replaces the threshold-backward + per-channel-sum pair behind nn.Conv(..., bias=True) [+ relu] of
 * ConvModule (models/utils/modules.py:L91-175): one pass, deterministic two-stage sum.  Workspace:
 * jdet_frozen_bn_act_backward_workspace(P, C); C % 4 == 0 and the same channel counts as the BN kernels. */
int jdet_bias_act_backward(const float* grad_y_nhwc, const float* y_nhwc, long P, int C, int relu,
                           float* grad_pre_nhwc, float* grad_bias, void* workspace, size_t workspace_bytes,
                           jdet_stream_t stream);

/* Per-channel sum over the P rows of a channels-last (P, C) tensor for ANY 1 <= C <= 256: the bias gradient of a
 * convolution WITHOUT an activation whose channel count jdet_bias_act_backward does not take (the 15- and 5-channel
 * output convs of the heads, ConvModule, models/utils/modules.py:L91-175).  Deterministic two-stage sum; workspace:
 * jdet_channel_sum_workspace(P, C) bytes (0 = unsupported). */
size_t jdet_channel_sum_workspace(long P, int C);
int jdet_channel_sum(const float* x_nhwc, long P, int C, float* sums, void* workspace, size_t workspace_bytes,
                     jdet_stream_t stream);

/* Graph-safe forms of two framework operations (csrc/graph_safe.hip): plain kernels where the framework records a memset
 * node into a captured step (memset nodes do not reliably re-execute on replay on this stack).
 *   jdet_zero_fill   : bytes % 4 == 0, p 4-byte aligned (`tensor.zero_()` / `torch.zeros` of a large tensor)
 *   jdet_sum_squares : out[0] = sum x[i]^2 (take_sqrt: its square root = the 2-norm) of a 16-byte aligned fp32 buffer --
 *                      the gradient norm of the clip in optims/optimizer.py:L26-36 (SGD.pre_step); float64 accumulation in
 *                      a fixed order over at most 1024 partial sums (bitwise reproducible); workspace:
 *                      jdet_sum_squares_workspace() bytes, 8-byte aligned, any content. */
int jdet_zero_fill(void* p, size_t bytes, jdet_stream_t stream);
/* Post-capture pass over a hipGraph_t that has NOT been instantiated yet: every memset node (the framework's reduction
 * semaphores, the convolution library's zero fills of atomically adding solvers -- whichever its benchmark picked at
 * capture time) becomes a kernel node with the same destination, pattern, extent, dependencies and dependents.
 * *n_replaced (optional): how many.  Runner's graph mode runs it on every captured step (keep_graph capture). */
int jdet_graph_replace_memset_nodes(void* hip_graph, int* n_replaced);
size_t jdet_sum_squares_workspace(void);
int jdet_sum_squares(const float* x, size_t n, int take_sqrt, float* out, void* workspace, size_t workspace_bytes,
                     jdet_stream_t stream);

/* Active rotating filter.  Replace orn.py:L260-269 (arf_forward) and L271-281 (arf_backward).
 * weight (nOut,nIn,nOri,kH,kW); indices (nOri,kH,kW,nRot) uint8 1-based;
 * out (nOut*nRot, nIn*nOri, kH, kW). */
int jdet_arf_forward(const float* weight, const uint8_t* indices, int nOut, int nIn, int nOri,
                     int kH, int kW, int nRot, float* out, jdet_stream_t stream);
int jdet_arf_backward(const uint8_t* indices, const float* grad_out, int nOut, int nIn, int nOri,
                      int kH, int kW, int nRot, float* grad_weight, jdet_stream_t stream);
/* the same with the expanded bank in channels-last memory (nOut*nRot, kH, kW, nIn*nOri): the layout the channels-last
 * convolution reads (the library otherwise converts the bank inside every forward / gradient call) */
int jdet_arf_forward_cl(const float* weight, const uint8_t* indices, int nOut, int nIn, int nOri, int kH, int kW,
                        int nRot, float* out_cl, jdet_stream_t stream);
int jdet_arf_backward_cl(const uint8_t* indices, const float* grad_out_cl, int nOut, int nIn, int nOri, int kH, int kW,
                         int nRot, float* grad_weight, jdet_stream_t stream);

/* RotationInvariantPooling (orn.py:L595-618: `x.view(N, -1, nOrientation, h, w).max(2)`) on channels-last rows:
 * x (P, C) with C = groups * nO (nO 4 or 8), y (P, groups) = max over each group's nO orientation channels; backward
 * = the gradient autograd derives for that maximum (to the channels equal to it, split evenly among ties). */
int jdet_rip_forward(const float* x_nhwc, long P, int C, int nO, float* y_nhwc, jdet_stream_t stream);
int jdet_rip_backward(const float* x_nhwc, const float* y_nhwc, const float* grad_y_nhwc, long P, int C, int nO,
                      float* grad_x_nhwc, jdet_stream_t stream);

/* Rotated box delta codec (fused elementwise).  Replace the Jittor tensor programs
 * models/boxes/box_ops.py:L229-285 (delta2bbox_rotated: rois (n,5), deltas (n, ncls*5) -> out
 * (n, ncls*5); max_shape / clip_border are accepted by the reference but never applied) and
 * box_ops.py:L180-226 (bbox2delta_rotated: proposals (n,5), gt (n,5) -> (n,5)).
 * means5 / stds5 are HOST pointers to 5 floats.  norm_angle is the floor-mod form L176-178. */
int jdet_delta2bbox_rotated(const float* rois, const float* deltas, int n, int ncls,
                            const float* means5, const float* stds5, float wh_ratio_clip, float* out,
                            jdet_stream_t stream);
int jdet_bbox2delta_rotated(const float* proposals, const float* gt, int n, const float* means5,
                            const float* stds5, float* out, jdet_stream_t stream);

/* Box codecs of the Oriented R-CNN path, one fused launch each.  Replace the Jittor tensor programs
 * models/boxes/coder.py:L332-437 (MidpointOffsetCoder: horizontal anchor (n,4) <-> 6 deltas (dx, dy, dw, dh, da,
 * db) of a rotated box given by its enclosing box and the offsets of its top-most / right-most vertex; decode
 * rebuilds the parallelogram, stretches it to a rectangle and returns the regularised obb (n,5)) and L449-518
 * (OrientedDeltaXYWHTCoder: rotated RoI (n,5) <-> 5 deltas in the RoI's frame, the angle delta taken as the
 * smaller of dtheta / dtheta + pi/2 with a w <-> h swap; decode is class-wise: deltas (n, ncls*5) -> (n, ncls*5)),
 * with obb2hbb / obb2poly / rectpoly2obb / regular_theta / regular_obb of ops/bbox_transforms.py:L499-517,L575-646
 * inlined.  `max_shape` is accepted and ignored by the reference's decoders, so it is not a parameter here.
 * means / stds are HOST pointers (6 resp. 5 floats). */
int jdet_midpoint_offset_decode(const float* anchors_hbb, const float* deltas, long n, const float* means6,
                                const float* stds6, float wh_ratio_clip, float* out_obb, jdet_stream_t stream);
int jdet_midpoint_offset_encode(const float* anchors_hbb, const float* gt_obb, long n, const float* means6,
                                const float* stds6, float* out6, jdet_stream_t stream);
int jdet_oriented_delta_decode(const float* rois, const float* deltas, long n, int ncls, const float* means5,
                               const float* stds5, float wh_ratio_clip, float* out, jdet_stream_t stream);
int jdet_oriented_delta_encode(const float* rois, const float* gt, long n, const float* means5, const float* stds5,
                               float* out, jdet_stream_t stream);

/* Dense anchor targets: replaces the index-list scatter of anchor_target_single
 * (models/boxes/anchor_target.py:L137-168) for the PseudoSampler case.  gt_inds (A) is the assigner's
 * output (0 negative, -1 ignored, i+1 = gt i); every anchor gets label (gt_labels[i] or 1 when gt_labels is
 * NULL; 0 otherwise), label weight (pos_weight / 1 / 0), encoded target (bbox2delta_rotated arithmetic, zeros
 * for non-positives) and box weight (1 / 0).  *num_pos (device, zeroed by the caller) += number of positives.
 * means5 / stds5 are HOST pointers. */
int jdet_anchor_targets_rotated(const float* anchors, const float* gt, const int32_t* gt_labels,
                                const int32_t* gt_inds, int A, int K, const float* means5,
                                const float* stds5, float pos_weight, int32_t* labels,
                                float* label_weights, float* bbox_targets, float* bbox_weights,
                                int32_t* num_pos, jdet_stream_t stream);

/* MaxIoUAssigner.assign_wrt_overlaps (models/boxes/assigner.py:L160-219) in two launches, no host
 * sync (the reference loops over gts in Python with a masked store per gt and jt.sync_all()).
 *   overlaps (K, A) row-major, gts are rows (assigner.py:L143)
 *   negatives: neg_iou_lo <= max < neg_iou_hi (float threshold: lo = 0)
 *   gt_labels (K) int32 or NULL; labels (A) int32 or NULL; labels_filled = assigned_labels_filled
 *   out: gt_inds (A) int32 in {-1, 0, 1..K}, max_overlaps (A), labels (A)
 * Column argmax ties resolve to the first gt (Jittor's tie rule is unpinned). */
/* Top-down step of the FPN (necks/fpn.py:L160-171: `laterals[i-1] += nn.interpolate(laterals[i], mode="nearest")`, then
 * `/= upsample_div_factor`) as one pass over channels-last maps: out = (lateral + nearest_upsample(top)) / div_factor.
 *   lateral / out (N, H, W, C), top (N, Ht, Wt, C) float32, 16-byte aligned, C % 4 == 0 (else JDET_E_UNSUPPORTED)
 *   nearest rule: src = min(floor(dst * in / out), in - 1) in float32 (= dst >> 1 for an exact 2x)
 * backward: grad_top = sum of grad_out over the pre-image of every coarse pixel, / div_factor (the gradient of `lateral` is
 * grad_out / div_factor itself: no launch). */
int jdet_upsample_add_nhwc_forward(const float* lateral, const float* top, int N, int C, int H, int W, int Ht, int Wt,
                                   float div_factor, float* out, jdet_stream_t stream);
int jdet_upsample_add_nhwc_backward(const float* grad_out, int N, int C, int H, int W, int Ht, int Wt, float div_factor,
                                    float* grad_top, jdet_stream_t stream);

/* Input pipeline on the device: uint8 batch (N, Hs, Ws, 3) -> float32 (N, Hs, Ws, 3) = channels-last memory of a
 * (N, 3, Hs, Ws) tensor, (v - mean[c]) / std[c] with the optional channel reversal first (`Normalize`,
 * data/transforms.py:L467-487), zeros outside each image's valid_hw[n] = (height, width) (`collate_batch`,
 * data/custom.py:L90-106).  mean3 / std3 are HOST pointers indexed by output channel.  Bit-identical to the host
 * arithmetic; a quarter of the PCIe bytes. */
int jdet_normalize_u8_nhwc(const uint8_t* src_nhwc, const int32_t* valid_hw, int N, int Hs, int Ws,
                           const float* mean3, const float* std3, int swap_rb, float* dst_nhwc, jdet_stream_t stream);

/* Feature refinement of R3Det, replaces fr.py:L113-159 (forward) and L161-242 (backward, 1 + 4 * points float
 * atomics per scalar there; a sorted gather here).  feat / out / grads: (N, H, W, C) channels-last, C % 4 == 0;
 * boxes (N, H, W, 5) [x_ctr, y_ctr, w, h, angle]: out = feat + sum over the `points` (1: centre; 5: centre + the four
 * corners) of the bilinear sample of feat at that point, with the reference's coordinate convention (box column 0
 * is used as the row coordinate, fr.py:L134-135). */
int jdet_feature_refine_forward(const float* feat_nhwc, const float* boxes, int N, int C, int H, int W,
                                float spatial_scale, int points, float* out_nhwc, jdet_stream_t stream);
size_t jdet_feature_refine_backward_workspace(int N, int C, int H, int W, int points);
int jdet_feature_refine_backward(const float* grad_out_nhwc, const float* boxes, int N, int C, int H, int W,
                                 float spatial_scale, int points, float* grad_in_nhwc, void* workspace,
                                 size_t workspace_bytes, jdet_stream_t stream);

/* Polygon (4-point) IoU matrix ious (n1, n2) and polygon NMS.  Replaces nms_poly.py: `devPolyIoU` L113-133 /
 * `poly_nms` L187-232 / `multiclass_poly_nms` L234-245 and the per-pair CPU `iou_poly` L247-252 of the DOTA
 * evaluation and tile merging.  polys rows: x1 y1 .. x4 y4 (stride >= 8 floats), any orientation, convex or not.
 * mode 0: the kernel's degenerate rule (union == 0 -> 1), mode 1: iou_poly's (inter / max(union, 0.01)).
 * jdet_nms_poly: rows of row_len 8, or 9 with a label in column 8 (different labels never suppress each other;
 * n_labels > 1: labels are 0 .. n_labels-1 and `order` visits them label by label -- one scan workgroup each);
 * suppression at IoU > thr; keep (n) uint8 over original indices; workspace = jdet_nms_rotated_workspace(n). */
int jdet_poly_iou(const float* polys1, int n1, int stride1, const float* polys2, int n2, int stride2, int mode,
                  float* ious, jdet_stream_t stream);
int jdet_nms_poly(const float* polys, int n, int row_len, const int32_t* order, float iou_threshold, int n_labels,
                  uint8_t* keep, void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* Axis-aligned overlaps out (K, A) row-major of K gt boxes (K,4) against A boxes (A, box_stride >= 4; the first four
 * columns are x1,y1,x2,y2): the tensor program of `bbox_overlaps` (models/boxes/iou_calculator.py:L235-350, modes
 * "iou" (iof = 0) and "iof" (iof = 1), not aligned) in the same operation order, one launch.  plus_one: the legacy
 * +1 pixel convention (BboxOverlaps2D_v1).  alive (A) or NULL: columns of boxes with alive == 0 are -1. */
int jdet_bbox_overlaps_hbb(const float* gts, int K, const float* boxes, int A, int box_stride, int iof, int plus_one,
                           float eps, const uint8_t* alive, float* out, jdet_stream_t stream);

size_t jdet_assign_max_iou_workspace(int K);
int jdet_assign_max_iou(const float* overlaps, int K, int A, float pos_iou_thr, float neg_iou_lo,
                        float neg_iou_hi, float min_pos_iou, int match_low_quality,
                        int gt_max_assign_all, const int32_t* gt_labels, int labels_filled,
                        int32_t* gt_inds, float* max_overlaps, int32_t* labels, void* workspace,
                        size_t workspace_bytes, jdet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* JDET_HIP_H_ */
