// The measured alternatives of the RoIAlign forward (rounds 3-4), outside the product ABI: the kernels file of the
// product (../roi_align_impl.inc) compiled with the two extra arithmetics --
//   mode 2: merged taps through the channel-sliced kernels (roi_align_sliced.h): XCD x owns channels [32 x, 32 x + 32) of
//           every RoI; reads beyond the L2 1.34 M -> 0.55 M requests, 63.7-80 us against 58 us (profiles/r04_roi_fwd_notes.md)
//   mode 3: taps deduplicated over a line of bins (roi_align_line.h): rows through the L1 -45 %, 71 us against 58 us
//   mode 5: taps merged over PAIRS of neighbouring bins, two accumulators (roi_align_pair.h, round 6): rows -22 %, L1 accesses
//           -20 %, VALU +46 %: 59.4 us against 56.4 us for the rolling-window product kernel (profiles/r06_roi_fwd_ring.md)
// -- behind jdet_roi_align_forward_cl_mode.  Kept with their parity tests as the measured answers to "partition the XCDs
// by channel" and "deduplicate the pixel rows of neighbouring bins"; neither is a product path.
#define JDET_ROI_EXPERIMENTAL_MODES 1
#include "../roi_align_impl.inc"

#include "jdet_experimental.h"

JDET_API size_t jdet_roi_align_forward_cl_mode_workspace(int mode, int R, int PH, int PW) {
  if (R <= 0 || PH <= 0 || PW <= 0) return 256;
  if (mode == kFwdSliced) return jdet_roi_sliced::plan_carve(nullptr, R, (long)PH * PW).bytes;
  return 256 + 2 * sizeof(int32_t) * (size_t)R;
}

JDET_API int jdet_roi_align_forward_cl_mode(int mode, int variant, const float* feat, int N, int C, int H, int W,
                                            const float* rois, int R, int PH, int PW, float spatial_scale,
                                            int sample_num, int n_orient, const int32_t* order, float* out_cl,
                                            void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  if (mode != kFwdSliced && mode != kFwdLine && mode != kFwdStaged && mode != kFwdPair) return JDET_E_BADARG;
  int e = check_common(variant, feat, rois, out_cl, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (C % 4 != 0 || (size_t)H * W * C * 4 >= (1ull << 31)) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && n_orient != 4 && n_orient != 8) return JDET_E_UNSUPPORTED;
  if (R == 0) return JDET_OK;
  hipStream_t st = (hipStream_t)stream;
  if (mode == kFwdSliced) {
    if (!sliced_ok(variant, R, N, C, H, W, PH, PW, sample_num, n_orient)) return JDET_E_UNSUPPORTED;
    if (!workspace || workspace_bytes < jdet_roi_align_forward_cl_mode_workspace(mode, R, PH, PW)) return JDET_E_WORKSPACE;
    switch (variant) {
      case JDET_ROI_ROTATED:
        return launch_sliced<JDET_ROI_ROTATED, 0>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 1, workspace, st);
      case JDET_ROI_ROTATED_V1:
        return launch_sliced<JDET_ROI_ROTATED_V1, 0>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 1, workspace, st);
      case JDET_ROI_RIROI:
        if (n_orient == 8)
          return launch_sliced<JDET_ROI_ROTATED, 8>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 8, workspace, st);
        return launch_sliced<JDET_ROI_ROTATED, 4>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 4, workspace, st);
      case JDET_ROI_HBB_V0:
        return launch_sliced<JDET_ROI_HBB_V0, 0>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 1, workspace, st);
      default:
        return launch_sliced<JDET_ROI_HBB_V1, 0>(feat, rois, out_cl, R, N, C, H, W, PH, PW, spatial_scale, 1, workspace, st);
    }
  }
  // mode 3: the RoI-stationary launch with the line kernel where it applies (`order`: a schedule of
  // jdet_roi_spatial_order, or NULL); mode 4: the footprint-staged kernel (roi_align_stage.h), same launch shape
  const int lm = mode;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_fwd<JDET_ROI_ROTATED>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true, lm);
    case JDET_ROI_ROTATED_V1:
      return launch_fwd<JDET_ROI_ROTATED_V1>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true, lm);
    case JDET_ROI_RIROI:
      return launch_fwd<JDET_ROI_RIROI>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, order, st, true, lm);
    case JDET_ROI_HBB_V0:
      return launch_fwd<JDET_ROI_HBB_V0>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true, lm);
    default:
      return launch_fwd<JDET_ROI_HBB_V1>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true, lm);
  }
}
