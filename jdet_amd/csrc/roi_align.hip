// RoIAlign family for gfx950 (MI355X): the exported entry points of the forward (kernels and launchers:
// roi_align_impl.inc; backward: roi_align_bwd.hip).  Two arithmetics, each behind its OWN entry points -- the library
// holds no process-wide mode:
//   jdet_roi_align_forward / _cl_roi / _cl            merged taps (the product path)
//   jdet_roi_align_forward_reference / _cl_reference  the reference's operation order (bit-identical to the CPU
//                                                     oracle: the parity twin of the tests and of smoke())
// The measured alternatives of rounds 3-4 (channel-sliced kernels, line-deduplicated taps) compile from the same
// kernels file into libjdet_experimental.so (experimental/roi_align_modes.hip); nothing of them is in this library.
#include "roi_align_impl.inc"

namespace {

int forward_any(int mode, int variant, const float* feat, int N, int C, int H, int W, const float* rois, int R, int PH,
                int PW, float spatial_scale, int sample_num, int n_orient, const int32_t* order, float* out,
                bool out_cl, jdet_stream_t stream) {
  int e = check_common(variant, feat, rois, out, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (out_cl) {
    if (C % 4 != 0 || (size_t)H * W * C * 4 >= (1ull << 31)) return JDET_E_UNSUPPORTED;
    if (variant == JDET_ROI_RIROI && n_orient != 4 && n_orient != 8) return JDET_E_UNSUPPORTED;
  }
  if (R == 0) return JDET_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_fwd<JDET_ROI_ROTATED>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, out_cl, mode);
    case JDET_ROI_ROTATED_V1:
      return launch_fwd<JDET_ROI_ROTATED_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, out_cl, mode);
    case JDET_ROI_RIROI:
      return launch_fwd<JDET_ROI_RIROI>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, order, st, out_cl, mode);
    case JDET_ROI_HBB_V0:
      return launch_fwd<JDET_ROI_HBB_V0>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, out_cl, mode);
    default:
      return launch_fwd<JDET_ROI_HBB_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, out_cl, mode);
  }
}

// channels-last result under the XCD-aware spatial order; the schedule lives in the caller's workspace
int forward_cl_any(int mode, int variant, const float* feat, int N, int C, int H, int W, const float* rois, int R,
                   int PH, int PW, float spatial_scale, int sample_num, int n_orient, float* out_cl, void* workspace,
                   size_t workspace_bytes, jdet_stream_t stream) {
  int e = check_common(variant, feat, rois, out_cl, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (C % 4 != 0 || (size_t)H * W * C * 4 >= (1ull << 31)) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && n_orient != 4 && n_orient != 8) return JDET_E_UNSUPPORTED;
  if (R == 0) return JDET_OK;
  if (!workspace || workspace_bytes < jdet_roi_align_forward_cl_workspace(R, PH, PW)) return JDET_E_WORKSPACE;
  const int32_t* order = nullptr;
  if (R >= 64) {   // below that the map traffic is too small for the schedule to matter
    int32_t* o = (int32_t*)workspace;
    const int cols = (variant == JDET_ROI_HBB_V0 || variant == JDET_ROI_HBB_V1) ? 5 : 6;
    e = jdet_roi_spatial_order(rois, R, cols, spatial_scale, N, H, W, o, o + R, stream);
    if (e) return e;
    order = o;
  }
  return forward_any(mode, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient, order,
                     out_cl, true, stream);
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

JDET_API int jdet_roi_spatial_order(const float* rois, int R, int roi_cols, float spatial_scale, int N,
                                    int H, int W, int32_t* order, int32_t* workspace,
                                    jdet_stream_t stream) {
  if (R < 0 || (roi_cols != 5 && roi_cols != 6) || N <= 0 || H <= 0 || W <= 0) return JDET_E_BADARG;
  if (R == 0) return JDET_OK;
  if (!rois || !order || !workspace) return JDET_E_BADARG;
  hipLaunchKernelGGL(roi_order_kernel, dim3(1), dim3(kOrderThreads), 0, (hipStream_t)stream, rois, R,
                     roi_cols, spatial_scale, N, H, W, order, workspace);
  return jdet_launch_status();
}

JDET_API int jdet_roi_align_forward(int variant, const float* feat, int N, int C, int H, int W,
                                    const float* rois, int R, int PH, int PW, float spatial_scale,
                                    int sample_num, int n_orient, const int32_t* order, float* out,
                                    jdet_stream_t stream) {
  return forward_any(kFwdMerged, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient, order,
                     out, false, stream);
}

// the same call in the reference's operation order (per-lane accumulation exactly as the reference kernel's:
// bit-identical to the CPU oracle)
JDET_API int jdet_roi_align_forward_reference(int variant, const float* feat, int N, int C, int H, int W,
                                              const float* rois, int R, int PH, int PW, float spatial_scale,
                                              int sample_num, int n_orient, const int32_t* order, float* out,
                                              jdet_stream_t stream) {
  return forward_any(kFwdReference, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient,
                     order, out, false, stream);
}

// RoI-stationary forward with a channels-last result (R, PH, PW, C): same kernels, results stored straight from
// registers (one contiguous 1 KiB row chunk per wave and bin) instead of being transposed through LDS.
JDET_API int jdet_roi_align_forward_cl_roi(int variant, const float* feat, int N, int C, int H, int W,
                                           const float* rois, int R, int PH, int PW, float spatial_scale,
                                           int sample_num, int n_orient, const int32_t* order, float* out_cl,
                                           jdet_stream_t stream) {
  return forward_any(kFwdMerged, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient, order,
                     out_cl, true, stream);
}

// Product forward with a channels-last result: the RoI-stationary kernels under the XCD-aware spatial order.
JDET_API size_t jdet_roi_align_forward_cl_workspace(int R, int PH, int PW) {
  if (R <= 0 || PH <= 0 || PW <= 0) return 256;
  return 256 + 2 * sizeof(int32_t) * (size_t)R;      // the two int32 arrays of the spatial order
}

JDET_API int jdet_roi_align_forward_cl(int variant, const float* feat, int N, int C, int H, int W, const float* rois,
                                       int R, int PH, int PW, float spatial_scale, int sample_num, int n_orient,
                                       float* out_cl, void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  return forward_cl_any(kFwdMerged, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient,
                        out_cl, workspace, workspace_bytes, stream);
}

JDET_API int jdet_roi_align_forward_cl_reference(int variant, const float* feat, int N, int C, int H, int W,
                                                 const float* rois, int R, int PH, int PW, float spatial_scale,
                                                 int sample_num, int n_orient, float* out_cl, void* workspace,
                                                 size_t workspace_bytes, jdet_stream_t stream) {
  return forward_cl_any(kFwdReference, variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient,
                        out_cl, workspace, workspace_bytes, stream);
}

// Atomic-scatter backward (all dialects, any sampling).  The exported jdet_roi_align_backward
// (roi_align_bwd.hip) prefers the sorted-gather path and falls back to this one.
int jdet_roi_align_backward_atomic(int variant, const float* grad_out, const float* rois, int R, int N, int C,
                                   int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                   int n_orient, const int32_t* order, float* grad_in, hipStream_t st) {
  if (!grad_in && (long)N * C * H * W > 0) return JDET_E_BADARG;
  int e = check_common(variant, grad_out, rois, grad_in, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  int he = jdet_zero_async(grad_in, sizeof(float) * (size_t)N * C * H * W, st);
  if (he) return he;
  if (R == 0) return JDET_OK;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_bwd<JDET_ROI_ROTATED>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    case JDET_ROI_ROTATED_V1:
      return launch_bwd<JDET_ROI_ROTATED_V1>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    case JDET_ROI_RIROI:
      return launch_bwd<JDET_ROI_RIROI>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, order, st);
    case JDET_ROI_HBB_V0:
      return launch_bwd<JDET_ROI_HBB_V0>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    default:
      return launch_bwd<JDET_ROI_HBB_V1>(grad_out, rois, grad_in, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
  }
}
