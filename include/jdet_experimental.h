/* libjdet_experimental.so -- NOT part of the product ABI (include/jdet_hip.h, libjdet_hip.so).
 *
 * Measured alternatives and calibration probes that scripts/, bench.py (JDET_ROI_FWD_PATH=pool) and
 * tests/test_gpu_experimental_pool.py load explicitly through jdet_amd/_experimental.py.  Nothing in jdet_amd.ops /
 * jdet_amd.models loads this library.  Sources: jdet_amd/csrc/experimental/.
 */
#ifndef JDET_EXPERIMENTAL_H
#define JDET_EXPERIMENTAL_H
#include "jdet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Register-cached RoIAlign forward, channels-last in and out (csrc/experimental/roi_align_pool.hip).  Same jt.code
 * sites and values as jdet_roi_align_forward in its default (merged-tap) mode: roi_align_rotated.py:L265-283,
 * roi_align_rotated_v1.py:L308-326, roi_align.py:L217-237.
 *   out_cl    : (R, PH, PW, C) -- the reference's (R, C, PH, PW) tensor stored channels-last.
 *   workspace : jdet_roi_align_forward_pool_workspace(R) bytes of scratch, no contract on its contents (per RoI: a
 *               stream of merged (pixel slot, weight) taps in blocks of four, per group of bins the byte offsets of
 *               the group's distinct pixels, the group count; plus the XCD schedule of jdet_roi_spatial_order).
 * Three launches: the schedule (R >= 64), a plan kernel (one workgroup per RoI, no map traffic) and a persistent pool
 * kernel in which a wave owns (group of bins, 128 channels), keeps the group's pixel rows in 96 VGPRs (each distinct
 * pixel of the group is loaded once) and accumulates every bin from that register cache through M0-relative operands.
 * Supported (jdet_roi_align_forward_pool_supported() == 1): rotated v0 / v1 and horizontal v0 / v1, C = 128 / 256 /
 * 512, sample_num 1 or 2, PH*PW <= 64, H*W*C*4 < 2 GiB per image; otherwise JDET_E_UNSUPPORTED.  RoIs with a
 * negative batch index are skipped (their rows stay untouched).
 * Measured at the north-star point (profiles/r03_roi_pool_notes.md): plan 56 us + pool 64 us against 60 us for the
 * product kernel -- the register cache halves the rows through the vector L1, but the forward is bound by the traffic
 * beyond the L2, which it does not reduce.  Kept as the measured answer to "dedup the taps of a whole RoI". */
int jdet_roi_align_forward_pool_supported(int variant, int C, int H, int W, int PH, int PW, int sample_num);
size_t jdet_roi_align_forward_pool_workspace(int R);
int jdet_roi_align_forward_pool(int variant, const float* feat_nhwc, int N, int C, int H, int W,
                                const float* rois, int R, int PH, int PW, float spatial_scale, int sample_num,
                                float* out_cl, void* workspace, size_t workspace_bytes, jdet_stream_t stream);

/* The two measured alternatives of the RoIAlign forward that shipped inside libjdet_hip.so behind a process-wide mode in
 * round 4 (csrc/experimental/roi_align_modes.hip compiles the product's kernels file with them), channels-last result:
 *   mode 2: the merged-tap arithmetic through the CHANNEL-SLICED kernels (roi_align_sliced.h): one launch sorts the RoIs
 *           by the Morton code of their centre and writes the PLAN (per (RoI, bin) the merged tap list), the second
 *           gives every XCD one 32-channel slice of every RoI.  sample_num == 2, PH*PW >= 16, C % 32 == 0; bit-equal
 *           to jdet_roi_align_forward_cl.  `order` unused.  workspace: schedule + plan (~136 bytes per (RoI, bin)).
 *   mode 3: every distinct pixel row of a LINE of bins loaded once (roi_align_line.h); sample_num == 2, PH, PW <= 8
 *           (elsewhere the product kernels run); values as the product's up to the order of a bin's sum.  `order`: a
 *           schedule of jdet_roi_spatial_order or NULL; workspace unused.
 * Both measured slower than the product kernel at the north-star point (profiles/r04_roi_fwd_notes.md). */
size_t jdet_roi_align_forward_cl_mode_workspace(int mode, int R, int PH, int PW);
int jdet_roi_align_forward_cl_mode(int mode, int variant, const float* feat_nhwc, int N, int C, int H, int W,
                                   const float* rois, int R, int PH, int PW, float spatial_scale, int sample_num,
                                   int n_orient, const int32_t* order, float* out_cl, void* workspace,
                                   size_t workspace_bytes, jdet_stream_t stream);

/* Calibration probe (scripts/gather_probe.py; csrc/experimental/gather_probe.hip): n_blocks workgroups of 4 waves, every
 * wave loads rows_per_wave pseudo-random 1 KiB rows of buf (total_rows x 256 floats), `unroll` (4 / 8 / 16) in flight,
 * drawn from a window of window_rows rows -- one shared window, or one per workgroup (local_windows != 0). */
int jdet_debug_gather_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave, int local_windows,
                            int n_blocks, int unroll, float* sink, jdet_stream_t stream);
/* ... the same gather with every row added `pairs` times into a 49 x 256 LDS accumulator block (ds_add_f32) that is
 * streamed to out (n_blocks x 49 x 256 floats) at the end: the main loop of a pixel-stationary RoIAlign, emulated. */
/* ... the same rows fetched as dword / dwordx2 / dwordx4 loads (dwords_per_lane 1 / 2 / 4; 4 rows in flight) */
int jdet_debug_gather_width_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                  int dwords_per_lane, int n_blocks, float* sink, jdet_stream_t stream);
int jdet_debug_gather_accumulate_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                       int pairs, int n_blocks, float* out, jdet_stream_t stream);

/* Calibration probe (scripts/dma_probe.py; csrc/experimental/dma_probe.hip), round 6: the gathers of jdet_debug_gather_probe
 * as LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction into a per-wave LDS ring of `unroll` KiB).
 * seg = 1: one 1 KiB map row per instruction; 4 / 8: four 256-byte / eight 128-byte pieces of as many rows (a channel
 * slice of several pixels per instruction, per-lane source addresses).  read != 0: every landed KiB is read back with
 * one ds_read_b128 per lane.  unroll 4 / 8 / 16 with seg 1 / 4, 8 / 16 with seg 8; window_rows * 1024 < 2 GiB. */
int jdet_debug_dma_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave, int local_windows,
                         int n_blocks, int unroll, int seg, int read, float* sink, jdet_stream_t stream);

/* Calibration probe (scripts/mfma_probe.py; csrc/experimental/mfma_probe.hip): shader cycles per wave of `steps` K steps
 * of 32 v_mfma_f32_32x32x2_f32 with the pieces of the conv_wgrad.hip loop added one at a time (variant 0 bare MFMAs, 1 +
 * LDS fragment fetches, 2 + LDS tile writes and the barrier, 3 + buffer loads).  cycles: n_blocks * 4 values. */
int jdet_debug_mfma_probe(int variant, const float* src, int n_blocks, int steps, long long* cycles, float* sink,
                          jdet_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
