// Register-cached RoIAlign forward for gfx950 (MI355X): plan kernel + pool kernel, channels-last in and out.
//
// Same jt.code sites as jdet_roi_align_forward (roi_align_rotated.py:L265-283, roi_align_rotated_v1.py:L308-326,
// roi_align.py:L217-237), same values as the merged-tap kernel of roi_align.hip (reference value up to fp32
// re-association of the bilinear weights, then fma).
//
// Why: the RoI-stationary kernel of roi_align.hip loads every (bin, pixel) pair as its own 1 KiB row
// (0.98 M rows = 1.0 GB through the texture-address path at the north-star point) although a RoI
// touches only 0.45 M distinct pixels: neighbouring bins of a small RoI share most of their pixels.  Here a wave owns
// (RoI, 128 channels), lane = 2 channels, and keeps the pixel rows of a GROUP of bins in 96 VGPRs (48 slots):
//   * plan kernel (one workgroup per RoI, no map traffic): samples -> taps, taps of a bin merged per pixel (as the
//     merged kernel does), then the bins are walked in boustrophedon order and greedily packed into groups whose
//     union of pixels fits the 48 cache slots; every pixel of a group gets ONE slot (a 92x92 u16 table in LDS keyed
//     by the pixel's position inside the RoI's bounding box remembers (group, slot)).  Output per RoI: a stream of
//     8-byte (control, weight) tap entries in blocks of four (a bin = 1..4 blocks; flags on a block's last entry:
//     end of bin + output row, end of RoI, "the next block opens a new group") and per group the byte offsets of its
//     pixels.
//   * pool kernel: the stream arrives 64 entries at a time by ONE coalesced buffer_load_dwordx2 (lane = entry),
//     prefetched a record ahead; per group one dwordx2 load per pixel into a STATIC register pair v[16+2*slot]
//     (soffset = the pixel's byte offset, v_readlane from the group's offset vector), then per tap
//     v_readlane x2 (control, weight) -> s_set_gpr_idx_idx -> two v_fmac_f32 acc, s_weight, v[16 + M0]
//     (VSRC1 relative).  A bin's 512 B leave with one non-temporal dwordx2 store to the channels-last row.
//     (First version: plan through s_load_dwordx16 -- 66 MB of streaming scalar loads ran at the scalar cache's miss
//     bandwidth, 35 us for the loop skeleton alone; profiles/r03_roi_pool_notes.txt.)
//   The hot loop is one asm statement: hipcc has no way to keep a register array addressed through M0 in place
//   (pinned-register constraints make it copy the block around every statement).
#include <stdlib.h>

#include "roi_geom.h"
#include "jdet_experimental.h"

namespace {

using namespace jdet_roi;

typedef unsigned int v4u __attribute__((ext_vector_type(4)));

constexpr int kCap = 48;         // pixel cache slots = VGPR pairs v[16:111] of the pool kernel
constexpr int kMaxBins = 64;     // PH * PW limit of this path
constexpr int kTaps = 16;        // taps per bin (2x2 samples x 4 corners)
constexpr int kMaxGroups = 22;   // a closed group holds > kCap - 16 pixels, i.e. >= 3 bins
constexpr int kTab = 92;         // bounding-box table side (64*sqrt(2) + 2)
// per-RoI plan record: the tap stream (<= 64 bins x 16 entries, + one record of slack for the prefetch), then the
// groups' pixel offsets (64 dwords per group: [0, npix) offsets, padded with the first one; lane 63 = npix)
constexpr int kStreamBytes = (kMaxBins * kTaps + 128) * 8;
constexpr int kOffsOff = kStreamBytes;
constexpr int kMetaOff = kOffsOff + (kMaxGroups + 1) * 256;   // int: groups of the RoI; then kMaxGroups x u16 first entries
constexpr int kPlanBytes = kMetaOff + 64;
static_assert(kOffsOff == 9216 && kPlanBytes == 15168, "the pool kernel's asm uses these literally");

// stream entry = {ctrl, weight}.  ctrl[7:0] = 2 * slot (the M0 index of the pixel's register pair).  On the fourth
// entry of a block: [8] end of bin, [9] end of RoI, [10] the next block opens a new group, [16:11] output bin.
__device__ __forceinline__ int lanes_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

template <int VARIANT>
__global__ __launch_bounds__(256) void roi_plan_kernel(const float* __restrict__ rois, int R, int C, int H, int W,
                                                       int PH, int PW, float spatial_scale, int sample_num,
                                                       uint8_t* __restrict__ plan, const int32_t* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) unsigned short s_tab[kTab * kTab];
  __shared__ int2 s_list[kMaxBins * kTaps];
  __shared__ unsigned short s_tix[kMaxBins * kTaps];
  __shared__ int s_n[kMaxBins];
  __shared__ int s_box[4][4];
  // same schedule as the pool kernel: workgroup b of either kernel runs on XCD b % 8, so a RoI's plan is written
  // and read through the same L2
  const int r = order ? order[blockIdx.x] : (int)blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  const float* roi = rois + (size_t)r * ROI_COLS;
  RoiGeom g = roi_geom<VARIANT>(roi, spatial_scale, sample_num, PH, PW, 1, false);
  uint8_t* rec = plan + (size_t)r * kPlanBytes;
  if (g.batch < 0) {   // masked RoI: no groups, no pool task
    if (threadIdx.x == 0) *reinterpret_cast<int*>(rec + kMetaOff) = 0;
    return;
  }
  const int nbins = PH * PW;
  const int j = lane >> 2, q = lane & 3, qbase = lane & ~3;
  const int iy = q >> 1, ix = q & 1;
  const float inv_count = 1.f / g.count;   // count in {1, 2, 4}: exact

  for (int i = threadIdx.x; i < kTab * kTab / 8; i += 256) reinterpret_cast<uint4*>(s_tab)[i] = make_uint4(0u, 0u, 0u, 0u);

  // ---- wave c: samples of bins 16c .. 16c+15 (boustrophedon order); lane = (bin j, sample q) ----
  const int o = 16 * wave + j;
  int row = o / PW, col = o - row * PW;
  if (row & 1) col = PW - 1 - col;
  const bool sample_ok = o < nbins && iy < g.grid_h && ix < g.grid_w;
  SamplePos p = sample_pos<VARIANT>(g, sample_ok ? row : 0, sample_ok ? col : 0, iy, ix, H, W);
  const int valid = sample_ok && p.valid;
  // bounding box of every tap of the RoI
  int x0 = valid ? p.x_low : (1 << 30), y0 = valid ? p.y_low : (1 << 30);
  int x1 = valid ? p.x_high : -1, y1 = valid ? p.y_high : -1;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off, 64)); y0 = min(y0, __shfl_xor(y0, off, 64));
    x1 = max(x1, __shfl_xor(x1, off, 64)); y1 = max(y1, __shfl_xor(y1, off, 64));
  }
  if (lane == 0) {
    s_box[wave][0] = x0; s_box[wave][1] = y0; s_box[wave][2] = x1; s_box[wave][3] = y1;
  }
  __syncthreads();
#pragma unroll
  for (int w4 = 0; w4 < 4; w4++) {
    x0 = min(x0, s_box[w4][0]); y0 = min(y0, s_box[w4][1]);
    x1 = max(x1, s_box[w4][2]); y1 = max(y1, s_box[w4][3]);
  }
  const bool use_tab = (x1 - x0) < kTab && (y1 - y0) < kTab;   // else: no sharing across bins (always correct)

  const float hy = (float)(1. - (double)p.ly), hx = (float)(1. - (double)p.lx);
  const float w[4] = {hy * hx, hy * p.lx, p.ly * hx, p.ly * p.lx};
  const int ty[4] = {p.y_low, p.y_low, p.y_high, p.y_high};
  const int tx[4] = {p.x_low, p.x_high, p.x_low, p.x_high};
  int pix[4];
#pragma unroll
  for (int k = 0; k < 4; k++) pix[k] = ty[k] * W + tx[k];
  // merge the taps of a bin that hit the same pixel (inside the lane, then across the quad of lanes)
  float tw[4] = {w[0], w[1], w[2], w[3]};
  bool first[4] = {true, true, true, true};
#pragma unroll
  for (int k = 1; k < 4; k++)
#pragma unroll
    for (int m = 0; m < k; m++)
      if (pix[m] == pix[k]) {
        tw[m] += w[k];
        first[k] = false;
      }
#pragma unroll
  for (int d = 1; d < 4; d++) {
    const int src = qbase | ((q + d) & 3);
    const bool earlier = ((q + d) & 3) < q;
    const int ov = __shfl(valid, src, 64);
#pragma unroll
    for (int m = 0; m < 4; m++) {
      const int oo = __shfl(pix[m], src, 64);
      const float ww = __shfl(w[m], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = ov && oo == pix[k];
        tw[k] += same ? ww : 0.f;
        first[k] = first[k] && !(same && earlier);
      }
    }
  }
  int keep[4], mycnt = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    keep[k] = valid && first[k];
    mycnt += keep[k];
  }
  int below = 0, n_bin = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int ci = __shfl(mycnt, qbase | i, 64);
    below += i < q ? ci : 0;
    n_bin += ci;
  }
  if (o < kMaxBins) {
    int pos = o * kTaps + below;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (keep[k]) {
        s_list[pos] = make_int2(pix[k], __float_as_int(tw[k] * inv_count));
        s_tix[pos] = (unsigned short)((ty[k] - y0) * kTab + (tx[k] - x0));
        pos++;
      }
    if (q == 0) s_n[o] = o < nbins ? n_bin : 0;
  }
  __syncthreads();
  if (wave != 0) return;

  // ---- wave 0: greedy packing, one bin per step; lane t < 16 = tap t of the bin ----
  int obin_l = 0;   // lane o: output bin of plan-order index o
  {
    int orow = lane / PW, ocol = lane - orow * PW;
    if (orow & 1) ocol = PW - 1 - ocol;
    obin_l = orow * PW + ocol;
  }
  const int n_l = s_n[lane];
  int gid = 0, npix = 0, first_off = 0, ecur = 0, pend_pos = -1, pend_val = 0;
  int e0_l = 0;   // lane g: first stream entry of group g
  bool opened = false;
  int* offs_out = reinterpret_cast<int*>(rec + kOffsOff);
  int2* stream = reinterpret_cast<int2*>(rec);
  const int t = lane & 15;
  int2 e = s_list[t];
  int tix = s_tix[t];
  for (int ob = 0; ob < nbins; ob++) {
    const int n = __builtin_amdgcn_readlane(n_l, ob);
    const int obin = __builtin_amdgcn_readlane(obin_l, ob);
    const bool tv = lane < n;
    const int2 e_cur = e;
    const int tix_cur = tix;
    if (ob + 1 < nbins) {   // next bin's list entry: independent of the packing state
      e = s_list[(ob + 1) * kTaps + t];
      tix = s_tix[(ob + 1) * kTaps + t];
    }
    const int tag = (tv && use_tab) ? s_tab[tix_cur] : 0;
    bool is_new = tv && (tag >> 6) != gid + 1;
    unsigned long long mask = __ballot(is_new);
    int n_new = __popcll(mask);
    int newgroup = 0;
    if (!opened || npix + n_new > kCap) {
      if (opened) {   // close the running group: pad its offset list with its first pixel, lane 63 = pixel count
        if (lane >= npix) offs_out[gid * 64 + lane] = lane == 63 ? npix : first_off;
        gid++;
      }
      opened = true;
      npix = 0;
      newgroup = 1;
      if (lane == gid) e0_l = ecur;
      is_new = tv;
      mask = __ballot(is_new);
      n_new = n;
      first_off = n ? __builtin_amdgcn_readfirstlane(e_cur.x) * C * 4 : 0;   // n == 0: pixel 0 of the image
    }
    const int slot = is_new ? npix + lanes_below(mask) : (tag & 63);
    if (is_new) {
      if (use_tab) s_tab[tix_cur] = (unsigned short)(((gid + 1) << 6) | slot);
      offs_out[gid * 64 + slot] = e_cur.x * C * 4;
    }
    npix += n_new;
    // the previous bin's last control word, now that it is known whether this bin opens a group
    if (pend_pos >= 0 && lane == 0) stream[pend_pos].x = pend_val | ((newgroup && ob) ? (1 << 10) : 0);
    // this bin's blocks: taps padded to a multiple of 4 with (slot of tap 0, weight 0); an empty bin = one block
    const int slot0 = n ? __builtin_amdgcn_readfirstlane(slot) : 0;
    const int npad = n ? ((n + 3) & ~3) : 4;
    const int ctrl = 2 * (tv ? slot : slot0);
    if (lane < npad) {
      if (lane == npad - 1)
        stream[ecur + lane].y = tv ? e_cur.y : 0;                         // its control word is written one step later
      else
        stream[ecur + lane] = make_int2(ctrl, tv ? e_cur.y : 0);
    }
    pend_pos = ecur + npad - 1;
    pend_val = __builtin_amdgcn_readlane(ctrl, npad - 1) | (1 << 8) | (obin << 11);
    ecur += npad;
  }
  if (lane == 0) stream[pend_pos].x = pend_val | (1 << 9);
  if (lane >= npix) offs_out[gid * 64 + lane] = lane == 63 ? npix : first_off;
  // the pool kernel turns (groups per RoI, first entries) into its task lists
  if (lane < kMaxGroups) reinterpret_cast<unsigned short*>(rec + kMetaOff + 4)[lane] = (unsigned short)e0_l;
  if (lane == 0) *reinterpret_cast<int*>(rec + kMetaOff) = gid + 1;
}

#define JP_KERNEL_NAME roi_pool_kernel
#include "roi_pool_kernel.inc"
#undef JP_KERNEL_NAME
// profiling variants: JDET_POOL_ABL = 1 no taps, 2 no pixel loads, 3 no stores, 8 every pixel in one 512 KiB window
#define JP_KERNEL_NAME roi_pool_kernel_notaps
#define JP_ABL_NOTAPS
#include "roi_pool_kernel.inc"
#undef JP_ABL_NOTAPS
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_noloads
#define JP_ABL_NOLOADS
#include "roi_pool_kernel.inc"
#undef JP_ABL_NOLOADS
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_nostore
#define JP_ABL_NOSTORE
#include "roi_pool_kernel.inc"
#undef JP_ABL_NOSTORE
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_win
#define JP_ABL_WINDOW "0x7fc00"
#include "roi_pool_kernel.inc"
#undef JP_ABL_WINDOW
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_sc1
#define JP_STORE_FLAGS "sc1"
#include "roi_pool_kernel.inc"
#undef JP_STORE_FLAGS
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_sc0sc1
#define JP_STORE_FLAGS "sc0 sc1"
#include "roi_pool_kernel.inc"
#undef JP_STORE_FLAGS
#undef JP_KERNEL_NAME
#define JP_KERNEL_NAME roi_pool_kernel_plain
#define JP_STORE_FLAGS ""
#include "roi_pool_kernel.inc"
#undef JP_STORE_FLAGS
#undef JP_KERNEL_NAME

int env_int_pool(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

JDET_API int jdet_roi_align_forward_pool_supported(int variant, int C, int H, int W, int PH, int PW, int sample_num) {
  if (variant != JDET_ROI_ROTATED && variant != JDET_ROI_ROTATED_V1 && variant != JDET_ROI_HBB_V0 &&
      variant != JDET_ROI_HBB_V1)
    return 0;
  if (C <= 0 || (C != 128 && C != 256 && C != 512) || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return 0;
  if (sample_num != 1 && sample_num != 2) return 0;
  if (PH * PW > kMaxBins) return 0;
  if ((size_t)H * W * C * 4 >= (1ull << 31) || (size_t)PH * PW * C * 4 >= (1ull << 31)) return 0;
  return 1;
}

JDET_API size_t jdet_roi_align_forward_pool_workspace(int R) {
  // the plan records + the two int32 arrays of the XCD schedule
  return (size_t)(R > 0 ? R : 0) * (kPlanBytes + 8) + 256;
}

JDET_API int jdet_roi_align_forward_pool(int variant, const float* feat, int N, int C, int H, int W,
                                         const float* rois, int R, int PH, int PW, float spatial_scale,
                                         int sample_num, float* out_cl, void* workspace, size_t workspace_bytes,
                                         jdet_stream_t stream) {
  if (N < 0 || R < 0) return JDET_E_BADARG;
  if (!jdet_roi_align_forward_pool_supported(variant, C, H, W, PH, PW, sample_num)) return JDET_E_UNSUPPORTED;
  if (R == 0 || N == 0) return JDET_OK;
  if (!feat || !rois || !out_cl || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_roi_align_forward_pool_workspace(R)) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  uint8_t* plan = (uint8_t*)workspace;
  int32_t* order = (int32_t*)(plan + (size_t)R * kPlanBytes);
  const int cols = (variant == JDET_ROI_HBB_V0 || variant == JDET_ROI_HBB_V1) ? 5 : 6;
  const int32_t* use_order = nullptr;
  if (R >= 64) {
    int e = jdet_roi_spatial_order(rois, R, cols, spatial_scale, N, H, W, order, order + R, stream);
    if (e) return e;
    use_order = order;
  }
#define JDET_PLAN(V)                                                                                               \
  hipLaunchKernelGGL(roi_plan_kernel<V>, dim3(R), dim3(256), 0, st, rois, R, C, H, W, PH, PW, spatial_scale,       \
                     sample_num, plan, use_order)
  switch (variant) {
    case JDET_ROI_ROTATED: JDET_PLAN(JDET_ROI_ROTATED); break;
    case JDET_ROI_ROTATED_V1: JDET_PLAN(JDET_ROI_ROTATED_V1); break;
    case JDET_ROI_HBB_V0: JDET_PLAN(JDET_ROI_HBB_V0); break;
    default: JDET_PLAN(JDET_ROI_HBB_V1); break;
  }
#undef JDET_PLAN
  int e = jdet_launch_status();
  if (e) return e;
  static const int abl = env_int_pool("JDET_POOL_ABL", 0);
  static const int lds_pad = env_int_pool("JDET_POOL_LDS", 0);   // profiling: caps the workgroups per CU
  auto kern = abl == 1 ? roi_pool_kernel_notaps : abl == 2 ? roi_pool_kernel_noloads : abl == 3 ? roi_pool_kernel_nostore
              : abl == 8 ? roi_pool_kernel_win : abl == 5 ? roi_pool_kernel_sc1 : abl == 6 ? roi_pool_kernel_sc0sc1
              : abl == 7 ? roi_pool_kernel_plain : roi_pool_kernel;
  // persistent waves: 4 workgroups on each of the 256 CUs (register-limited residency), 8-way interleaved so that
  // workgroup b runs on XCD b % 8.  Dynamic LDS: prefix sums over a run of the schedule + the workgroup's task table.
  static const int wg_per_cu = env_int_pool("JDET_POOL_WGS", 4);
  const int wgs = 256 * wg_per_cu, n_run = jdet_cdiv(R, 8);
  const int tasks_wg = jdet_cdiv((long)n_run * kMaxGroups, wgs / 8) + 1;
  const size_t lds = (size_t)(n_run + 8) * 4 + (size_t)tasks_wg * 16 + lds_pad;
  if (lds > 60 * 1024) return JDET_E_UNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), lds, st, feat, rois, cols, use_order, R, out_cl, plan, C, H, W, PH * PW);
  return jdet_launch_status();
}
