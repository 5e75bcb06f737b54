// RoIAlign family for gfx950 (MI355X): ROIAlignRotated, ROIAlignRotated_v1, RiRoIAlign and the
// horizontal ROIAlign v0/v1, forward and backward.
//
// Reference semantics (per output element, fp32):
//   python/jdet/ops/roi_align_rotated.py:L21-127 (fwd), L128-255 (bwd)
//   python/jdet/ops/roi_align_rotated_v1.py:L71-145, L193-298
//   python/jdet/ops/riroi_align.py:L70-163, L228-358
//   python/jdet/ops/roi_align.py:L13-204
// The reference launches one CUDA thread per output element (n,c,ph,pw): every thread
// recomputes sin/cos + bin geometry and gathers 16 scattered floats from an NCHW map.
//
// MI355X design (not a translation):
//   * feature map is NHWC, so one bilinear tap is ONE contiguous C-vector: a wave64 reads a
//     256-channel tap with a single global_load_dwordx4 (64 lanes x 16 B = 1 KiB).
//   * one workgroup (4 waves) per (RoI, 256-channel chunk).  Sample geometry (position,
//     4 weights, 4 pixel offsets) is computed ONCE per sample, lane-parallel (lane = sample),
//     and broadcast to the wave with v_readlane -> all control flow in the tap loop is
//     wave-uniform and the weights live in SGPRs.
//   * per-lane accumulation order is exactly the reference's (w1*lt + w2*rt + w3*lb + w4*rb,
//     summed iy-major, then / count) with FMA contraction off -> forward is bit-identical to
//     the CPU oracle.
//   * results are staged in LDS as [channel][bin] and written out as one contiguous,
//     float4-coalesced (C_chunk*PH*PW) block in the reference's (R,C,PH,PW) layout.
//   * backward: grad_out chunk staged in LDS, same sample broadcast, hardware
//     global_atomic_add_f32 into the NHWC gradient (lane-contiguous 256 B per instruction).
#include <stdlib.h>

#include <type_traits>

#include "roi_geom.h"

namespace {

using namespace jdet_roi;

// ---------------------------------------------------------------------------------------------
// Forward
// ---------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

// accumulate one sample for the 4 channels of a float4 lane (reference operation order)
__device__ __forceinline__ void acc_sample4(float (&acc)[4], float w1, float w2, float w3, float w4,
                                            const v4f& lt, const v4f& rt, const v4f& lb, const v4f& rb) {
  acc[0] += (w1 * lt.x + w2 * rt.x + w3 * lb.x + w4 * rb.x);
  acc[1] += (w1 * lt.y + w2 * rt.y + w3 * lb.y + w4 * rb.y);
  acc[2] += (w1 * lt.z + w2 * rt.z + w3 * lb.z + w4 * rb.z);
  acc[3] += (w1 * lt.w + w2 * rt.w + w3 * lb.w + w4 * rb.w);
}

// RiRoIAlign (riroi_align.py:L130-152): the channels are C/nO groups of nO orientation planes, and output plane o of
// a group is  r_var * plane (o - ind) + l_var * plane (o - ind + 1)  (indices mod nO) of the sampled value, `ind` and
// the two fractions being per-RoI constants.  Lane owns 4 consecutive channels: a whole group when nO == 4, half of
// one when nO == 8 (the other half sits in the neighbouring lane).  IND is a template argument (the caller switches
// on the wave-uniform `ind`), so every plane lookup is a static register pick -- plus one select on the lane's
// parity when nO == 8, where the two lanes of a pair need planes 4 apart.  Accumulates in the reference's order:
// acc += r_var * val + l_var * val_plus, once per sample.
template <int NO, int IND>
__device__ __forceinline__ void ri_accumulate(float (&acc)[4], const float (&val)[4], int lane, float r_var,
                                              float l_var) {
  if (NO == 4) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      acc[k] += r_var * val[(k - IND + 4) & 3] + l_var * val[(k - IND + 5) & 3];
  } else {
    const bool odd = lane & 1;
    float a[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float other = __shfl_xor(val[k], 1, 64);
      a[k] = odd ? other : val[k];          // planes 0..3 of the group
      a[4 + k] = odd ? val[k] : other;      // planes 4..7
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {           // this lane's output plane o = 4 * odd + k
      const int i0 = (k - IND + 8) & 7, i1 = (k - IND + 9) & 7;
      const float v = odd ? a[i0 ^ 4] : a[i0];
      const float vp = odd ? a[i1 ^ 4] : a[i1];
      acc[k] += r_var * v + l_var * vp;
    }
  }
}

template <int NO>
__device__ __forceinline__ void ri_dispatch(float (&acc)[4], const float (&val)[4], int lane, int ind, float r_var,
                                            float l_var) {
  switch (ind) {   // wave-uniform
    case 0: ri_accumulate<NO, 0>(acc, val, lane, r_var, l_var); break;
    case 1: ri_accumulate<NO, 1>(acc, val, lane, r_var, l_var); break;
    case 2: ri_accumulate<NO, 2>(acc, val, lane, r_var, l_var); break;
    case 3: ri_accumulate<NO, 3>(acc, val, lane, r_var, l_var); break;
    case 4: ri_accumulate<NO, 4 % NO>(acc, val, lane, r_var, l_var); break;
    case 5: ri_accumulate<NO, 5 % NO>(acc, val, lane, r_var, l_var); break;
    case 6: ri_accumulate<NO, 6 % NO>(acc, val, lane, r_var, l_var); break;
    default: ri_accumulate<NO, 7 % NO>(acc, val, lane, r_var, l_var); break;
  }
}

// The same mix with `ind` as a run-time value (wave-uniform or per lane), for the kernels that mix ONCE per finished bin:
// every plane lookup is a 4-way select on (index & 3) plus, for nO == 8, a select between the lane's own registers and
// its pair lane's (the two lanes of a pair hold planes 0-3 / 4-7 of a group).  No register arrays indexed at run time:
// the switch of ri_dispatch above costs 8 template instances whose plane arrays end up in scratch (RiRoIAlign ran
// 92 us against 59 us for the plain dialect at the north-star point).
__device__ __forceinline__ float ri_sel4(const float (&v)[4], int i) {
  const float lo = (i & 1) ? v[1] : v[0], hi = (i & 1) ? v[3] : v[2];
  return (i & 2) ? hi : lo;
}

template <int NO>
__device__ __forceinline__ void ri_mix(float (&out)[4], const float (&val)[4], int lane, int ind, float r_var,
                                       float l_var) {
  if (NO == 4) {
#pragma unroll
    for (int k = 0; k < 4; k++)
      out[k] = 0.f + (r_var * ri_sel4(val, (k - ind) & 3) + l_var * ri_sel4(val, (k - ind + 1) & 3));
  } else {
    const int odd = lane & 1;
    float other[4];
#pragma unroll
    for (int k = 0; k < 4; k++) other[k] = __shfl_xor(val[k], 1, 64);
#pragma unroll
    for (int k = 0; k < 4; k++) {          // this lane's output plane o = 4 * odd + k; source planes (o - ind), (o - ind + 1) mod 8
      const int ta = (4 * odd + k - ind) & 7, tb = (4 * odd + k - ind + 1) & 7;
      const float va = ((ta >> 2) == odd) ? ri_sel4(val, ta & 3) : ri_sel4(other, ta & 3);
      const float vb = ((tb >> 2) == odd) ? ri_sel4(val, tb & 3) : ri_sel4(other, tb & 3);
      out[k] = 0.f + (r_var * va + l_var * vb);
    }
  }
}

// ... and with `ind` as a template argument (the caller switches on the wave-uniform value ONCE, around its whole bin
// loop): every plane lookup is a static register pick.  Lane parity drops out: output plane 4 odd + k reads plane
// (4 odd + t) mod 8 with t = (k - IND) mod 8, i.e. component t & 3 of the lane itself when t < 4, of its pair lane
// otherwise.
template <int NO, int IND>
__device__ __forceinline__ void ri_mix_static(float (&out)[4], const float (&val)[4], float r_var, float l_var) {
  if (NO == 4) {
#pragma unroll
    for (int k = 0; k < 4; k++) out[k] = 0.f + (r_var * val[(k - IND + 4) & 3] + l_var * val[(k - IND + 5) & 3]);
  } else {
    float other[4];
#pragma unroll
    for (int k = 0; k < 4; k++) other[k] = __shfl_xor(val[k], 1, 64);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      constexpr int dummy = 0;
      (void)dummy;
      const int ta = (k - IND + 8) & 7, tb = (k - IND + 9) & 7;
      const float va = ta < 4 ? val[ta & 3] : other[ta & 3];
      const float vb = tb < 4 ? val[tb & 3] : other[tb & 3];
      out[k] = 0.f + (r_var * va + l_var * vb);
    }
  }
}

// one sample into the lane's 4 accumulators; NO == 0: plain RoIAlign, NO == 4 / 8: RiRoIAlign with that many planes
template <int NO>
__device__ __forceinline__ void acc_sample(float (&acc)[4], const RoiGeom& g, int lane, float w1, float w2, float w3,
                                           float w4, const v4f& lt, const v4f& rt, const v4f& lb, const v4f& rb) {
  if constexpr (NO == 0) {
    acc_sample4(acc, w1, w2, w3, w4, lt, rt, lb, rb);
  } else {
    const float val[4] = {(w1 * lt.x + w2 * rt.x + w3 * lb.x + w4 * rb.x), (w1 * lt.y + w2 * rt.y + w3 * lb.y + w4 * rb.y),
                          (w1 * lt.z + w2 * rt.z + w3 * lb.z + w4 * rb.z), (w1 * lt.w + w2 * rt.w + w3 * lb.w + w4 * rb.w)};
    ri_dispatch<NO>(acc, val, lane, g.ind, g.r_var, g.l_var);
  }
}

#include "roi_align_sliced.h"   // channel-sliced forward (product path of the sampling-2 dialects)

// ---- vector fast path: C % 4 == 0, map < 2 GiB per image; RiRoI with 4 or 8 orientation planes -------------
// Lane owns 4 consecutive channels.  Taps are fetched with buffer_load_dwordx4 whose per-tap
// pixel byte offset is an SGPR (soffset) -- no per-load 64-bit VALU address arithmetic -- and the
// per-sample geometry is broadcast from the owning lane with v_readlane.
//   NW  = waves per workgroup;  SG = samples whose 4*SG taps are all in flight before first use
//   ABL = ablation switches for profiling builds only (0 in production):
//         1 taps forced to pixels 0..3 (L1-resident), 2 no interpolation math, 4 no output stream
// Per-RoI prologue shared by the vector kernels: geometry with the control-flow / addressing
// scalars pinned into SGPRs (hipcc otherwise wraps every buffer_load in a waterfall loop, guide
// T20), and a raw buffer descriptor over the RoI's image (out-of-range reads return 0).
template <int VARIANT, bool TRIG = true>
__device__ __forceinline__ RoiGeom vec_prologue(const float* feat, const float* rois, int r, int C, int H,
                                                int W, int PH, int PW, float spatial_scale,
                                                int sample_num, __amdgpu_buffer_rsrc_t& rsrc, int nO = 1) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  RoiGeom g = roi_geom<VARIANT, TRIG>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH, PW, nO, false);
  g.batch = __builtin_amdgcn_readfirstlane(g.batch);
  g.ind = __builtin_amdgcn_readfirstlane(g.ind);
  g.grid_h = __builtin_amdgcn_readfirstlane(g.grid_h);
  g.grid_w = __builtin_amdgcn_readfirstlane(g.grid_w);
  const float* img = feat + (size_t)g.batch * H * W * C;
  const unsigned long long img_bits = (unsigned long long)img;
  const unsigned img_lo = __builtin_amdgcn_readfirstlane((unsigned)img_bits);
  const unsigned img_hi = __builtin_amdgcn_readfirstlane((unsigned)(img_bits >> 32));
  const void* img_u = (const void*)(((unsigned long long)img_hi << 32) | img_lo);
  rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(img_u), 0, __builtin_amdgcn_readfirstlane((int)((size_t)H * W * C * 4)), 0x00020000);
  return g;
}

// Direct path for one (RoI, <=256-channel chunk): every tap is a buffer_load_dwordx4 whose pixel
// byte offset is an SGPR; results go to s_out[channel][bin].  NW waves split the bins.
// OUT_CL: results go straight from registers to the channels-last output row (r, bin, c0 + 4*lane .. +3) -- one
// contiguous 1 KiB non-temporal store per (wave, bin), no LDS staging; otherwise to s_out[channel][bin].
template <int VARIANT, int NW, int SG, int ABL, bool OUT_CL = false, int NO = 0>
__device__ __forceinline__ void direct_chunk(const RoiGeom& g, const __amdgpu_buffer_rsrc_t rsrc, int c0,
                                             int cc, int C, int H, int W, int PW, int nbins, int wave,
                                             int lane, float* __restrict__ s_out, float* __restrict__ out_row = nullptr) {
  const bool lane_ok = lane * 4 < cc;
  const int voff = (c0 + (lane_ok ? lane * 4 : 0)) * 4;  // byte offset of this lane's channels
  const int pix_bytes = C * 4;
  const int spb = g.grid_h * g.grid_w;
  const int nb = (nbins - wave + NW - 1) / NW;
  const int bpc = spb <= 64 ? (spb > 0 ? 64 / spb : 64) : 1;
  const int passes = spb <= 64 ? 1 : (spb + 63) / 64;

  // lane = one sample of this wave's bins; offsets pre-multiplied to bytes
  auto lane_sample = [&](int kg, int pass) -> Sample {
    int my_kb, my_r;
    if (passes == 1) {
      my_kb = spb > 0 ? lane / spb : 0;
      my_r = spb > 0 ? lane % spb : 0;
    } else {
      my_kb = 0;
      my_r = pass * 64 + lane;
    }
    const int my_bin = wave + NW * (kg + my_kb);
    const bool ok = my_kb < bpc && my_bin < nbins && my_r < spb && g.grid_w > 0;
    const int iy = ok ? my_r / g.grid_w : 0;
    const int ix = ok ? my_r % g.grid_w : 0;
    const int bb = ok ? my_bin : 0;
    Sample s = make_sample<VARIANT>(g, bb / PW, bb % PW, iy, ix, H, W);
    if (ABL & 1) { s.o1 = 0; s.o2 = 1; s.o3 = 2; s.o4 = 3; }
    s.o1 *= pix_bytes; s.o2 *= pix_bytes; s.o3 *= pix_bytes; s.o4 *= pix_bytes;
    return s;
  };
  auto tap = [&](int soff) -> v4f {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
  };

  for (int kg = 0; kg < nb; kg += bpc) {
    Sample mine = lane_sample(kg, 0);
    for (int kb = 0; kb < bpc && kg + kb < nb; kb++) {
      const int bin = wave + NW * (kg + kb);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int pass = 0; pass < passes; pass++) {
        if (passes > 1) mine = lane_sample(kg, pass);
        const int lane0 = passes == 1 ? kb * spb : 0;
        const int ns = passes == 1 ? spb : min(64, spb - pass * 64);
        int j = 0;
        for (; j + SG <= ns; j += SG) {
          Sample sv[SG];
          int all_valid = 1;
#pragma unroll
          for (int u = 0; u < SG; u++) {
            sv[u] = bcast(mine, lane0 + j + u);
            all_valid &= sv[u].valid;
          }
          if (all_valid) {
            v4f t[SG][4];
#pragma unroll
            for (int u = 0; u < SG; u++) {
              t[u][0] = tap(sv[u].o1);
              t[u][1] = tap(sv[u].o2);
              t[u][2] = tap(sv[u].o3);
              t[u][3] = tap(sv[u].o4);
            }
            if (ABL & 2) {
#pragma unroll
              for (int u = 0; u < SG; u++) acc[0] += t[u][0].x + t[u][1].y + t[u][2].z + t[u][3].w;
            } else {
#pragma unroll
              for (int u = 0; u < SG; u++)
                acc_sample<NO>(acc, g, lane, sv[u].w1, sv[u].w2, sv[u].w3, sv[u].w4, t[u][0], t[u][1], t[u][2], t[u][3]);
            }
          } else {
#pragma unroll
            for (int u = 0; u < SG; u++)
              if (sv[u].valid)
                acc_sample<NO>(acc, g, lane, sv[u].w1, sv[u].w2, sv[u].w3, sv[u].w4, tap(sv[u].o1), tap(sv[u].o2),
                               tap(sv[u].o3), tap(sv[u].o4));
          }
        }
        for (; j < ns; j++) {
          const Sample s = bcast(mine, lane0 + j);
          if (s.valid)
            acc_sample<NO>(acc, g, lane, s.w1, s.w2, s.w3, s.w4, tap(s.o1), tap(s.o2), tap(s.o3), tap(s.o4));
        }
      }
      if (lane_ok) {
        if (OUT_CL) {
          const v4f o = {acc[0] / g.count, acc[1] / g.count, acc[2] / g.count, acc[3] / g.count};
          __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(out_row + (size_t)bin * C + c0 + lane * 4));
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) s_out[(lane * 4 + k) * nbins + bin] = acc[k] / g.count;
        }
      }
    }
  }
}

template <int VARIANT, int NW, int SG, int ABL, bool OUT_CL = false, int NO = 0>
__global__ __launch_bounds__(NW * 64) void roi_align_fwd_vec_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num,
    const int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) float s_out[];  // [cc][nbins]
  const int r = order ? order[blockIdx.x] : blockIdx.x;  // XCD-aware spatial schedule
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  // wave id is wave-uniform but threadIdx-derived: make that provable (SGPR) for the compiler
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rsrc;
  const RoiGeom g = vec_prologue<VARIANT>(feat, rois, r, C, H, W, PH, PW, spatial_scale, sample_num, rsrc,
                                          NO ? NO : 1);
  if (g.batch < 0) return;  // masked RoI (belongs to another pyramid level): its output rows are not ours
  if (OUT_CL) {
    direct_chunk<VARIANT, NW, SG, ABL, true, NO>(g, rsrc, c0, cc, C, H, W, PW, nbins, wave, lane, s_out,
                                                 out + (size_t)r * nbins * C);
    return;
  }
  direct_chunk<VARIANT, NW, SG, ABL, false, NO>(g, rsrc, c0, cc, C, H, W, PW, nbins, wave, lane, s_out);
  __syncthreads();
  float* __restrict__ dst = out + ((size_t)r * C + c0) * nbins;
  if (ABL & 4) {
    if (threadIdx.x == 0) dst[0] = s_out[0];
    return;
  }
  // coalesced write-out of the contiguous [cc][nbins] block (cc % 4 == 0 -> 16 B aligned)
  const int total = cc * nbins;
  const float4* s4 = reinterpret_cast<const float4*>(s_out);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int i = threadIdx.x; i < (total >> 2); i += NW * 64) d4[i] = s4[i];
}

// ---- merged-tap path (sample_num == 2: the configuration every reference config uses) -----------
// PMC on the reference-order kernel above (profiles/r01_roi_align_fwd_rocprofv3_summary.txt): 2714 VALU
// instructions per wave -- 8 fp32 ops per tap-channel-quad with contraction off, 4 divides per bin,
// double-precision trig in every wave -- keep the SIMDs ~45 % busy, and 1.51 M dwordx4 tap loads keep
// the vector-memory path ~60 % busy: co-limited, neither hides behind the other.  This path attacks both:
//   * the 4 samples of a bin sit bin/2 apart; whenever that is under a pixel their 16 taps revisit
//     the same few pixels (bench RoIs: 58 % of the taps are distinct within their bin).  Lane =
//     sample; inside each quad of lanes (= one bin) every tap looks up the other 15, the first
//     occurrence of a pixel takes the summed weight (pre-divided by the sample count), the rest are
//     dropped; survivors are compacted per bin through a 2 KiB/wave LDS scratch (overlaid on the
//     output staging block, before anything is staged) so that lane 4*bin+i holds entries i, 4+i, ...
//   * the tap loop runs over the dense list in batches of 4 loads: per surviving tap one
//     buffer_load_dwordx4 + two v_pk_fma_f32, no guards, no divides.
//   * sin/cos in double once per workgroup (wave 0) instead of once per wave.
// Result = reference value up to fp32 re-association of the weights (<= a few ulp of sum|w.v|);
// jdet_set_roi_forward_mode(1) selects the reference-order kernel above (bit-identical to the oracle).
// NO = 4 / 8: RiRoIAlign -- VARIANT is the rotated geometry, the orientation planes are mixed once per bin on the
// finished sum (the reference mixes per sample: equal up to fp32 re-association, like the merged weights).
template <int VARIANT, int NW, int ABL = 0, bool OUT_CL = false, int NO = 0>
__global__ __launch_bounds__(NW * 64) void roi_align_fwd_merged_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int PH, int PW, float spatial_scale, const int32_t* __restrict__ order, int abl_mask) {
  extern __shared__ __attribute__((aligned(16))) float s_out[];  // [cc][nbins]; first 2 KiB/wave: tap lists
  __shared__ float s_trig[2];
  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;                       // <= 16 * NW on this path
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr bool kRot = ROI_COLS == 6;
  const float* roi = rois + (size_t)r * ROI_COLS;
  if (kRot && wave == 0 && lane == 0) {
    if (ABL & 16) {   // profiling builds: how much of the workgroup's start-up is the double-precision trig
      s_trig[0] = cosf(roi[5]);
      s_trig[1] = sinf(roi[5]);
    } else {
      s_trig[0] = (float)cos((double)roi[5]);
      s_trig[1] = (float)sin((double)roi[5]);
    }
  }
  __amdgpu_buffer_rsrc_t rsrc;
  RoiGeom g = vec_prologue<VARIANT, false>(feat, rois, r, C, H, W, PH, PW, spatial_scale, 2, rsrc);
  if (g.batch < 0) return;
  if (kRot) {
    __syncthreads();
    g.cosT = s_trig[0];
    g.sinT = s_trig[1];
  }

  const bool lane_ok = lane * 4 < cc;
  const int voff = (c0 + (lane_ok ? lane * 4 : 0)) * 4;
  const int pix_bytes = C * 4;
  const int q = lane & 3, qbase = lane & ~3, kb_mine = lane >> 2;
  // wave w owns bins w, w+NW, ... (measured: better than contiguous runs -- neighbouring bins in flight
  // together share their L1 misses)
  const int nb = (nbins - wave + NW - 1) / NW;
  const int my_bin = wave + NW * kb_mine;
  const bool bin_ok = kb_mine < nb;
  const int bb = bin_ok ? my_bin : 0;
  Sample s = make_sample<VARIANT>(g, bb / PW, bb % PW, q >> 1, q & 1, H, W);
  if (!bin_ok) s.valid = 0;
  if (ABL & 1) {   // profiling builds: same tap structure, every tap inside one (abl_mask+1)-pixel window
    s.o1 &= abl_mask; s.o2 &= abl_mask; s.o3 &= abl_mask; s.o4 &= abl_mask;
  }
  const int o[4] = {s.o1 * pix_bytes, s.o2 * pix_bytes, s.o3 * pix_bytes, s.o4 * pix_bytes};
  const float w[4] = {s.w1, s.w2, s.w3, s.w4};
  float tw[4] = {w[0], w[1], w[2], w[3]};
  bool first[4] = {true, true, true, true};
#pragma unroll
  for (int k = 1; k < 4; k++)
#pragma unroll
    for (int j = 0; j < k; j++)
      if (o[j] == o[k]) {   // x_high == x_low / y_high == y_low at the map border
        tw[j] += w[k];      // (first occurrence collects; later ones are dropped)
        first[k] = false;
      }
#pragma unroll
  for (int d = 1; d < 4; d++) {
    const int src = qbase | ((q + d) & 3);
    const bool earlier = ((q + d) & 3) < q;
    const int ov = __shfl(s.valid, src, 64);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int oo = __shfl(o[j], src, 64);
      const float ww = __shfl(w[j], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = ov && oo == o[k];
        tw[k] += same ? ww : 0.f;
        first[k] = first[k] && !(same && earlier);
      }
    }
  }
  int ri_ind = 0;
  float ri_l = 0.f, ri_r = 1.f;
  if (NO) {
    ri_params(roi[5], NO, ri_ind, ri_l, ri_r);
    ri_ind = __builtin_amdgcn_readfirstlane(ri_ind);
  }
  const float inv_count = 1.f / g.count;   // count == 4 here: exact
  int keep[4], mycnt = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    keep[k] = s.valid && first[k];
    mycnt += keep[k];
  }
  // compaction: position inside the bin = kept taps of lower quad lanes + own lower kept taps
  int below = 0, n_bin = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int ci = __shfl(mycnt, qbase | i, 64);
    below += i < q ? ci : 0;
    n_bin += ci;
  }
  int2* list = reinterpret_cast<int2*>(s_out) + wave * 256 + kb_mine * 16;
  int pos = below;
#pragma unroll
  for (int k = 0; k < 4; k++)
    if (keep[k]) list[pos++] = make_int2(o[k], __float_as_int(tw[k] * inv_count));
  __builtin_amdgcn_wave_barrier();   // list is private to the wave; LDS ops of a wave retire in order
  int e_o[4];
  float e_w[4];
  const int2 e0 = list[0];
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int2 e = list[4 * b + q];
    const bool live = 4 * b + q < n_bin;
    e_o[b] = live ? e.x : e0.x;                       // pad: entry 0's pixel (a tap of this bin) ...
    e_w[b] = live ? __int_as_float(e.y) : 0.f;        // ... with weight 0
  }
  __syncthreads();   // every wave has read its list: s_out may now be used for results

  auto tap = [&](int soff) -> v4f {
    return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
  };
  // (RiRoIAlign: the bin loop is instantiated once per orientation shift and entered through one switch on the
  //  wave-uniform `ind`, so that the mix inside is pure register renaming)
  auto bins = [&](auto ind_c) {
  constexpr int IND = decltype(ind_c)::value;
  for (int kb = 0; kb < nb; kb++) {
    const int bin = wave + NW * kb;
    const int l0 = kb * 4;
    const int n = jdet_readlane_i(n_bin, l0);
    v4f t[4][4];
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (4 * b < n) {
#pragma unroll
        for (int i = 0; i < 4; i++) t[b][i] = tap(jdet_readlane_i(e_o[b], l0 + i));
      }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; b++)
      if (4 * b < n) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float wt = jdet_readlane_f(e_w[b], l0 + i);
          acc.x = __builtin_fmaf(wt, t[b][i].x, acc.x);
          acc.y = __builtin_fmaf(wt, t[b][i].y, acc.y);
          acc.z = __builtin_fmaf(wt, t[b][i].z, acc.z);
          acc.w = __builtin_fmaf(wt, t[b][i].w, acc.w);
        }
      }
    if constexpr (NO != 0) {
      const float val[4] = {acc.x, acc.y, acc.z, acc.w};
      float mixed[4];
      ri_mix_static<NO, IND>(mixed, val, ri_r, ri_l);
      acc = v4f{mixed[0], mixed[1], mixed[2], mixed[3]};
    }
    if (lane_ok) {
      if (OUT_CL) {   // channels-last row (r, bin, :): one contiguous 1 KiB store per wave, no LDS staging
        __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + ((size_t)r * nbins + bin) * C + c0 + lane * 4));
      } else {
        s_out[(lane * 4 + 0) * nbins + bin] = acc.x;
        s_out[(lane * 4 + 1) * nbins + bin] = acc.y;
        s_out[(lane * 4 + 2) * nbins + bin] = acc.z;
        s_out[(lane * 4 + 3) * nbins + bin] = acc.w;
      }
    }
  }
  };
  if constexpr (NO == 0) {
    bins(std::integral_constant<int, 0>{});
  } else {
    switch (ri_ind) {   // wave-uniform
      case 0: bins(std::integral_constant<int, 0>{}); break;
      case 1: bins(std::integral_constant<int, 1>{}); break;
      case 2: bins(std::integral_constant<int, 2>{}); break;
      case 3: bins(std::integral_constant<int, 3>{}); break;
      case 4: bins(std::integral_constant<int, 4 % (NO ? NO : 1)>{}); break;
      case 5: bins(std::integral_constant<int, 5 % (NO ? NO : 1)>{}); break;
      case 6: bins(std::integral_constant<int, 6 % (NO ? NO : 1)>{}); break;
      default: bins(std::integral_constant<int, 7 % (NO ? NO : 1)>{}); break;
    }
  }
  if (OUT_CL) return;
  __syncthreads();
  float* __restrict__ dst = out + ((size_t)r * C + c0) * nbins;
  if (ABL & 4) {
    if (threadIdx.x == 0) dst[0] = s_out[0];
    return;
  }
  const int total = cc * nbins;
  const float4* s4 = reinterpret_cast<const float4*>(s_out);
  float4* d4 = reinterpret_cast<float4*>(dst);
  if (ABL & 8) {
    for (int i = threadIdx.x; i < (total >> 2); i += NW * 64) d4[i] = s4[i];
  } else {
    // write-once output: non-temporal stores keep the 100 MB result stream from evicting map lines out of L2
    // (67.5 -> 64.6 us/step)
    typedef float v4s __attribute__((ext_vector_type(4)));
    const v4s* sv = reinterpret_cast<const v4s*>(s_out);
    v4s* dv = reinterpret_cast<v4s*>(dst);
    for (int i = threadIdx.x; i < (total >> 2); i += NW * 64) __builtin_nontemporal_store(sv[i], &dv[i]);
  }
}

#include "roi_align_line.h"

// (An LDS pixel-cache variant -- per-RoI bitmap + rank dedup, distinct pixels staged once per 64-channel pass, taps
// served by ds_read_b128 -- was built and measured at 155 us against 69 us for the direct path at the time: four
// channel passes with two barriers each and 2 workgroups per CU leave the vector-memory path idle most of the time.
// Removed; DESIGN.md 3.1 / 7 describe what would have to be different.)

template <int VARIANT, int CHMAP>
__global__ __launch_bounds__(kBlock) void roi_align_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out,
    int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num, int nO,
    const int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) float s_out[];  // [cc][nbins]
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr int NW = kBlock / 64;
  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

  const RoiGeom g = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH,
                                      PW, nO, false);
  if (g.batch < 0) return;  // masked RoI (block-uniform)
  const float* __restrict__ img = feat + (size_t)g.batch * H * W * C;

  int src0[4], src1[4];
  bool cval[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cl = chan_of<CHMAP>(lane, k);
    cval[k] = cl < cc;
    const int ch = c0 + (cval[k] ? cl : 0);
    if (VARIANT == JDET_ROI_RIROI) {
      const int c = ch / nO, o = ch % nO;
      const int ind_rot = (o - g.ind + nO) % nO;
      const int ind_rot_plus = (ind_rot + 1 + nO) % nO;
      src0[k] = c * nO + ind_rot;
      src1[k] = c * nO + ind_rot_plus;
    } else {
      src0[k] = ch;
      src1[k] = ch;
    }
  }

  const int spb = g.grid_h * g.grid_w;
  const int nb = (nbins - wave + NW - 1) / NW;
  const int bpc = spb <= 64 ? (spb > 0 ? 64 / spb : 64) : 1;
  const int passes = spb <= 64 ? 1 : (spb + 63) / 64;

  auto lane_sample = [&](int kg, int pass) -> Sample {
    int my_kb, my_r;
    if (passes == 1) {
      my_kb = spb > 0 ? lane / spb : 0;
      my_r = spb > 0 ? lane % spb : 0;
    } else {
      my_kb = 0;
      my_r = pass * 64 + lane;
    }
    const int my_bin = wave + NW * (kg + my_kb);
    const bool ok = my_kb < bpc && my_bin < nbins && my_r < spb && g.grid_w > 0;
    const int iy = ok ? my_r / g.grid_w : 0;
    const int ix = ok ? my_r % g.grid_w : 0;
    const int bb = ok ? my_bin : 0;
    return make_sample<VARIANT>(g, bb / PW, bb % PW, iy, ix, H, W);
  };

  for (int kg = 0; kg < nb; kg += bpc) {
    Sample mine = lane_sample(kg, 0);
    for (int kb = 0; kb < bpc && kg + kb < nb; kb++) {
      const int bin = wave + NW * (kg + kb);
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
      for (int pass = 0; pass < passes; pass++) {
        if (passes > 1) mine = lane_sample(kg, pass);
        const int lane0 = passes == 1 ? kb * spb : 0;
        const int ns = passes == 1 ? spb : min(64, spb - pass * 64);
        for (int j = 0; j < ns; j++) {
          const Sample s = bcast(mine, lane0 + j);
          if (!s.valid) continue;  // reference returns 0 for out-of-map samples
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (!cval[k]) continue;
            const float lt = img[(size_t)s.o1 * C + src0[k]];
            const float rt = img[(size_t)s.o2 * C + src0[k]];
            const float lb = img[(size_t)s.o3 * C + src0[k]];
            const float rb = img[(size_t)s.o4 * C + src0[k]];
            const float val = (s.w1 * lt + s.w2 * rt + s.w3 * lb + s.w4 * rb);
            if (VARIANT == JDET_ROI_RIROI) {
              const float lt1 = img[(size_t)s.o1 * C + src1[k]];
              const float rt1 = img[(size_t)s.o2 * C + src1[k]];
              const float lb1 = img[(size_t)s.o3 * C + src1[k]];
              const float rb1 = img[(size_t)s.o4 * C + src1[k]];
              const float val_plus = (s.w1 * lt1 + s.w2 * rt1 + s.w3 * lb1 + s.w4 * rb1);
              acc[k] += g.r_var * val + g.l_var * val_plus;
            } else {
              acc[k] += val;
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (cval[k]) s_out[chan_of<CHMAP>(lane, k) * nbins + bin] = acc[k] / g.count;
    }
  }
  __syncthreads();
  float* __restrict__ dst = out + ((size_t)r * C + c0) * nbins;
  const int total = cc * nbins;
  if (((total & 3) == 0) && ((((size_t)r * C + c0) * nbins) & 3) == 0) {
    const float4* s4 = reinterpret_cast<const float4*>(s_out);
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = threadIdx.x; i < (total >> 2); i += kBlock) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < total; i += kBlock) dst[i] = s_out[i];
  }
}

// ---------------------------------------------------------------------------------------------
// Backward (feature gradient)
// ---------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(kBlock) void roi_align_bwd_kernel(
    const float* __restrict__ grad_out, const float* __restrict__ rois, float* __restrict__ grad_in,
    int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num, int nO,
    const int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) float s_g[];  // [cc][nbins]
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr int CHMAP = 1;
  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if ((int)rois[(size_t)r * ROI_COLS] < 0) return;  // masked RoI (block-uniform, before any barrier)

  // stage grad_out[r, c0:c0+cc, :, :] (contiguous) into LDS
  {
    const float* __restrict__ src = grad_out + ((size_t)r * C + c0) * nbins;
    const int total = cc * nbins;
    if (((total & 3) == 0) && ((((size_t)r * C + c0) * nbins) & 3) == 0) {
      const float4* s4 = reinterpret_cast<const float4*>(src);
      float4* d4 = reinterpret_cast<float4*>(s_g);
      for (int i = threadIdx.x; i < (total >> 2); i += kBlock) d4[i] = s4[i];
    } else {
      for (int i = threadIdx.x; i < total; i += kBlock) s_g[i] = src[i];
    }
  }
  __syncthreads();

  const RoiGeom g = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH,
                                      PW, nO, true);
  float* __restrict__ img = grad_in + (size_t)g.batch * H * W * C;

  int dst0[4], dst1[4];
  bool cval[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cl = chan_of<CHMAP>(lane, k);
    cval[k] = cl < cc;
    const int ch = c0 + (cval[k] ? cl : 0);
    if (VARIANT == JDET_ROI_RIROI) {
      const int c = ch / nO, o = ch % nO;
      const int ind_rot = (o - g.ind + nO) % nO;
      const int ind_rot_plus = (ind_rot + 1 + nO) % nO;
      dst0[k] = c * nO + ind_rot;
      dst1[k] = c * nO + ind_rot_plus;
    } else {
      dst0[k] = ch;
      dst1[k] = ch;
    }
  }

  const int spb = g.grid_h * g.grid_w;
  if (spb <= 0) return;  // count == 0: the reference divides by zero -> inf*0; nothing sane to add
  const int nb = (nbins - wave + 3) >> 2;
  const int bpc = spb <= 64 ? 64 / spb : 1;
  const int passes = spb <= 64 ? 1 : (spb + 63) / 64;

  auto lane_sample = [&](int kg, int pass) -> Sample {
    int my_kb, my_r;
    if (passes == 1) {
      my_kb = lane / spb;
      my_r = lane % spb;
    } else {
      my_kb = 0;
      my_r = pass * 64 + lane;
    }
    const int my_bin = wave + 4 * (kg + my_kb);
    const bool ok = my_kb < bpc && my_bin < nbins && my_r < spb;
    const int iy = ok ? my_r / g.grid_w : 0;
    const int ix = ok ? my_r % g.grid_w : 0;
    const int bb = ok ? my_bin : 0;
    Sample m = make_sample<VARIANT>(g, bb / PW, bb % PW, iy, ix, H, W);
    // fold 1/count into the weights once per sample (reference: top*w/count per element)
    m.w1 /= g.count;
    m.w2 /= g.count;
    m.w3 /= g.count;
    m.w4 /= g.count;
    return m;
  };

  for (int kg = 0; kg < nb; kg += bpc) {
    Sample mine = lane_sample(kg, 0);
    for (int kb = 0; kb < bpc && kg + kb < nb; kb++) {
      const int bin = wave + 4 * (kg + kb);
      float top[4];
#pragma unroll
      for (int k = 0; k < 4; k++) top[k] = cval[k] ? s_g[chan_of<CHMAP>(lane, k) * nbins + bin] : 0.f;
      for (int pass = 0; pass < passes; pass++) {
        if (passes > 1) mine = lane_sample(kg, pass);
        const int lane0 = passes == 1 ? kb * spb : 0;
        const int ns = passes == 1 ? spb : min(64, spb - pass * 64);
        for (int j = 0; j < ns; j++) {
          const Sample s = bcast(mine, lane0 + j);
          if (!s.valid) continue;
#pragma unroll
          for (int k = 0; k < 4; k++) {
            if (!cval[k]) continue;
            const float g1 = top[k] * s.w1, g2 = top[k] * s.w2, g3 = top[k] * s.w3, g4 = top[k] * s.w4;
            if (VARIANT == JDET_ROI_RIROI) {
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst0[k], g1 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst0[k], g2 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst0[k], g3 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst0[k], g4 * g.r_var);
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst1[k], g1 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst1[k], g2 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst1[k], g3 * g.l_var);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst1[k], g4 * g.l_var);
            } else {
              unsafeAtomicAdd(img + (size_t)s.o1 * C + dst0[k], g1);
              unsafeAtomicAdd(img + (size_t)s.o2 * C + dst0[k], g2);
              unsafeAtomicAdd(img + (size_t)s.o3 * C + dst0[k], g3);
              unsafeAtomicAdd(img + (size_t)s.o4 * C + dst0[k], g4);
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// XCD-aware spatial schedule.
// The feature map (67 MB at 256x256x256 fp32) does not fit a 4 MiB XCD L2, and workgroup b runs
// on XCD b % 8: with RoIs in arbitrary order every XCD streams the whole map through the fabric
// (measured: FETCH 451 MB per launch for 67 MB of map, L2 hit 37 %).  This kernel buckets RoIs by
// the Morton code of their centre cell (counting sort, one workgroup, O(R)), then deals
// contiguous runs of the sorted list to the 8 XCDs: order[b] = sorted[start(b % 8) + b / 8].
// Each XCD then sweeps one compact region of the map and concurrently resident workgroups are
// spatial neighbours (measured: FETCH 139 MB, L2 hit 73 %).
// ---------------------------------------------------------------------------------------------
constexpr int kOrderThreads = 1024;
constexpr int kOrderCellsLog2 = 5;                   // 32 x 32 cells per image
constexpr int kOrderCells = 1 << (2 * kOrderCellsLog2);
constexpr int kOrderMaxImages = 8;                   // bins in LDS: 8 * 1024 * 4 B = 32 KiB

__device__ __forceinline__ unsigned morton2(unsigned x, unsigned y) {
  auto spread = [](unsigned v) {
    v &= 0xffff;
    v = (v | (v << 8)) & 0x00ff00ff;
    v = (v | (v << 4)) & 0x0f0f0f0f;
    v = (v | (v << 2)) & 0x33333333;
    v = (v | (v << 1)) & 0x55555555;
    return v;
  };
  return spread(x) | (spread(y) << 1);
}

__global__ __launch_bounds__(kOrderThreads) void roi_order_kernel(const float* __restrict__ rois, int R,
                                                                 int roi_cols, float spatial_scale, int N,
                                                                 int H, int W, int32_t* __restrict__ order,
                                                                 int32_t* __restrict__ sorted_tmp) {
  constexpr int kKeep = 8;                      // RoIs per thread whose key stays in registers
  constexpr int kLdsSorted = kKeep * kOrderThreads;  // R <= 8192: sorted list lives in LDS
  __shared__ int s_bins[kOrderMaxImages * kOrderCells];
  __shared__ int s_scan[kOrderThreads / 64];
  __shared__ int s_sorted[kLdsSorted];
  const int nimg = min(N, kOrderMaxImages);
  const int nbins = nimg * kOrderCells;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nbins; i += kOrderThreads) s_bins[i] = 0;
  __syncthreads();
  auto key_of = [&](int r) -> int {
    const float* p = rois + (size_t)r * roi_cols;
    float cx, cy;
    if (roi_cols == 5) {
      cx = 0.5f * (p[1] + p[3]) * spatial_scale;
      cy = 0.5f * (p[2] + p[4]) * spatial_scale;
    } else {
      cx = p[1] * spatial_scale;
      cy = p[2] * spatial_scale;
    }
    int b = (int)p[0];
    b = min(max(b, 0), nimg - 1);
    const float fx = fminf(fmaxf(cx / (float)W, 0.f), 0.999999f);
    const float fy = fminf(fmaxf(cy / (float)H, 0.f), 0.999999f);
    const unsigned ix = (unsigned)(fx * (1 << kOrderCellsLog2));
    const unsigned iy = (unsigned)(fy * (1 << kOrderCellsLog2));
    return b * kOrderCells + (int)morton2(ix, iy);
  };
  int mykey[kKeep];
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int r = threadIdx.x + i * kOrderThreads;
    mykey[i] = r < R ? key_of(r) : 0;
    if (r < R) atomicAdd(&s_bins[mykey[i]], 1);
  }
  for (int r = threadIdx.x + kKeep * kOrderThreads; r < R; r += kOrderThreads) atomicAdd(&s_bins[key_of(r)], 1);
  __syncthreads();
  // exclusive scan of the bins: contiguous slice per thread, wave scan by DPP-style shuffles,
  // one LDS hop across the 16 waves
  const int per = (nbins + kOrderThreads - 1) / kOrderThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += s_bins[i];
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int wave_base = 0;
  for (int w = 0; w < wave; w++) wave_base += s_scan[w];
  int run = wave_base + incl - sum;
  for (int i = lo; i < hi; i++) {
    const int c = s_bins[i];
    s_bins[i] = run;
    run += c;
  }
  __syncthreads();
  const bool in_lds = R <= kLdsSorted;
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int r = threadIdx.x + i * kOrderThreads;
    if (r < R) {
      const int pos = atomicAdd(&s_bins[mykey[i]], 1);
      if (in_lds) s_sorted[pos] = r; else sorted_tmp[pos] = r;
    }
  }
  for (int r = threadIdx.x + kKeep * kOrderThreads; r < R; r += kOrderThreads)
    sorted_tmp[atomicAdd(&s_bins[key_of(r)], 1)] = r;
  if (!in_lds) __threadfence();  // global scratch is re-read by other waves: agent-scope release
  __syncthreads();
  // deal contiguous runs to the 8 XCDs (workgroup b -> XCD b % 8 is the observed dispatch rule;
  // a different placement only costs speed): start(x) = sum_{y<x} ceil((R - y) / 8)
  for (int b = threadIdx.x; b < R; b += kOrderThreads) {
    const int x = b & 7, p = b >> 3;
    int start = 0;
#pragma unroll
    for (int y = 0; y < 7; y++) start += y < x ? ((R - y + 7) >> 3) : 0;
    order[b] = in_lds ? s_sorted[start + p]
                      : __hip_atomic_load(sorted_tmp + start + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC tiled transposes: per image a (C, HW) <-> (HW, C) matrix transpose.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x,
                                                        float* __restrict__ y, int rows, int cols) {
  // x: (batch, rows, cols) -> y: (batch, cols, rows); 32x32 tiles, +1 pad (conflict-free)
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int rr = r0 + ty + i, ccol = c0 + tx;
    if (rr < rows && ccol < cols) tile[ty + i][tx] = x[base + (size_t)rr * cols + ccol];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int ccol = c0 + ty + i, rr = r0 + tx;
    if (rr < rows && ccol < cols) y[base + (size_t)ccol * rows + rr] = tile[tx][ty + i];
  }
}

int launch_transpose(const float* x, float* y, int batch, int rows, int cols, hipStream_t st) {
  if (batch == 0 || rows == 0 || cols == 0) return JDET_OK;
  dim3 grid(jdet_cdiv(cols, 32), jdet_cdiv(rows, 32), batch);
  if (grid.y > 65535 || grid.z > 65535) return JDET_E_UNSUPPORTED;
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, st, x, y, rows, cols);
  return jdet_launch_status();
}

// Tuning knobs of the vector forward path (A/B-able from the environment for profiling runs).
int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

// 0 = merged taps (default), 1 = reference operation order (bit-identical to the CPU oracle),
// 2 = merged taps through the channel-sliced kernels where they apply (roi_align_sliced.h; measured, not default)
// 3 = taps deduplicated over a line of bins where that kernel applies (roi_align_line.h; measured, not default)
int g_fwd_reference_order = 0;

template <int VARIANT>
int launch_fwd(const float* feat, const float* rois, float* out, int R, int C, int H, int W, int PH,
               int PW, float scale, int sample_num, int nO, const int32_t* order, hipStream_t st,
               bool out_cl = false) {
  const int chunks = jdet_cdiv(C, kChunkC);
  const size_t lds = (size_t)min(C, kChunkC) * PH * PW * sizeof(float);
  dim3 grid(R, chunks);
  const bool vec = (C % 4 == 0) && VARIANT != JDET_ROI_RIROI && (size_t)H * W * C * 4 < (1ull << 31);
  const int nbins = PH * PW;
  const bool big_ok = (C % 4 == 0) && (size_t)H * W * C * 4 < (1ull << 31);
  if (VARIANT == JDET_ROI_RIROI && big_ok && (nO == 4 || nO == 8)) {
    // orientation planes mixed in registers.  Default: merged-tap kernel + one mix per bin; reference-order mode
    // (or sampling other than 2x2): the per-sample kernel, bit-identical to the scalar one.
    const bool merged = sample_num == 2 && g_fwd_reference_order != 1 && nbins <= 64 && (out_cl || lds >= 8 * 2048);
    const size_t lds_m = out_cl ? 8 * 2048 : lds, lds_v = out_cl ? 16 : lds;
#define JDET_RI(NO_)                                                                                              \
  do {                                                                                                            \
    if (merged && out_cl)                                                                                         \
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<JDET_ROI_ROTATED, 4, 0, true, NO_>), grid, dim3(256), lds_m, \
                         st, feat, rois, out, C, H, W, PH, PW, scale, order, 0);                                  \
    else if (merged)                                                                                              \
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<JDET_ROI_ROTATED, 4, 0, false, NO_>), grid, dim3(256), lds_m, \
                         st, feat, rois, out, C, H, W, PH, PW, scale, order, 0);                                  \
    else if (out_cl)                                                                                              \
      hipLaunchKernelGGL((roi_align_fwd_vec_kernel<JDET_ROI_RIROI, 4, 4, 0, true, NO_>), grid, dim3(256), lds_v, st, \
                         feat, rois, out, C, H, W, PH, PW, scale, sample_num, order);                             \
    else                                                                                                          \
      hipLaunchKernelGGL((roi_align_fwd_vec_kernel<JDET_ROI_RIROI, 4, 4, 0, false, NO_>), grid, dim3(256), lds_v, st, \
                         feat, rois, out, C, H, W, PH, PW, scale, sample_num, order);                             \
  } while (0)
    if (nO == 8) JDET_RI(8);
    else JDET_RI(4);
#undef JDET_RI
    return jdet_launch_status();
  }
  if (out_cl) {   // channels-last output: vector kernels only (the callers check jdet_roi_align_forward_cl_supported)
    if (!vec) return JDET_E_UNSUPPORTED;
    constexpr int V = VARIANT == JDET_ROI_RIROI ? JDET_ROI_ROTATED : VARIANT;
    // (the merged kernel keeps its per-wave tap lists in the first 2 KiB / wave of the dynamic LDS block)
    if (sample_num == 2 && g_fwd_reference_order != 1 && nbins <= 64) {
      // The LDS request caps the workgroups per CU at 4 (36 KiB each; the tap lists need 16): one workgroup fewer
      // in flight per CU leaves the time where it is (59.4 vs 60.9 us at the north-star point) and cuts the reads
      // beyond the L2 by 14 % (1.34 M vs 1.55 M 128-byte requests, profiles/r03_roi_pool_notes.md) -- fewer RoIs
      // in flight, smaller working set.  JDET_ROI_FWD_LDS_KB overrides (profiling).
      static const int lds_kb = env_int("JDET_ROI_FWD_LDS_KB", 36);
      const size_t lds_cl = lds_kb > 16 ? (size_t)lds_kb * 1024 : 8 * 2048;
      static const int line_env = env_int("JDET_ROI_FWD_LINE", 0);     // (A/B runs; 2 = 8 rows per batch)
      const int line = g_fwd_reference_order == 3 ? 2 : (g_fwd_reference_order == 0 ? line_env : 0);
      if (line && PH <= kLineMaxBins && PW <= kLineMaxBins && nbins * 4 <= 256) {
        // taps deduplicated over a line of bins (roi_align_line.h): 36 KiB of tables = 4 workgroups per CU as well
        const size_t lds_ln = (size_t)kLineMaxBins * kLineSlots * (4 + 32);
        if (PH <= 7 && PW <= 7 && line == 2)
          hipLaunchKernelGGL((roi_align_fwd_line_kernel<V, 7, 8>), grid, dim3(256), lds_ln, st, feat, rois, out, C, H,
                             W, PH, PW, scale, order);
        else if (PH <= 7 && PW <= 7)
          hipLaunchKernelGGL((roi_align_fwd_line_kernel<V, 7, 16>), grid, dim3(256), lds_ln, st, feat, rois, out, C, H,
                             W, PH, PW, scale, order);
        else
          hipLaunchKernelGGL((roi_align_fwd_line_kernel<V, 8, 16>), grid, dim3(256), lds_ln, st, feat, rois, out, C, H,
                             W, PH, PW, scale, order);
      } else
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 0, true>), grid, dim3(256), lds_cl, st, feat, rois, out,
                         C, H, W, PH, PW, scale, order, 0);
    }
    else
      hipLaunchKernelGGL((roi_align_fwd_vec_kernel<V, 4, 4, 0, true>), grid, dim3(256), 16, st, feat, rois, out, C, H,
                         W, PH, PW, scale, sample_num, order);
    return jdet_launch_status();
  }
  if (vec && sample_num == 2 && g_fwd_reference_order != 1 && nbins <= 64 && lds >= 8 * 2048) {
    constexpr int V = VARIANT == JDET_ROI_RIROI ? JDET_ROI_ROTATED : VARIANT;
    static const int nw = env_int("JDET_ROI_FWD_WAVES", 4);
    static const int abl = env_int("JDET_ROI_ABLATE", 0);  // profiling builds only
    if (abl == 1)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 1>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (abl == 4)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 4>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (abl == 16)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 16>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (abl == 8)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 8>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (abl == 5)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4, 5>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (nw == 8)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 8>), grid, dim3(512), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else if (nw == 2 && nbins <= 32)
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 2>), grid, dim3(128), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
    else
      hipLaunchKernelGGL((roi_align_fwd_merged_kernel<V, 4>), grid, dim3(256), lds, st, feat, rois, out, C, H, W,
                         PH, PW, scale, order, env_int("JDET_ROI_ABL_MASK", 63));
  } else if (vec) {
    constexpr int V = VARIANT == JDET_ROI_RIROI ? JDET_ROI_ROTATED : VARIANT;  // never RiRoI here
    static const int nw = env_int("JDET_ROI_FWD_WAVES", 4);
    static const int sg = env_int("JDET_ROI_FWD_SG", 4);
    static const int abl = env_int("JDET_ROI_ABLATE", 0);  // profiling builds only
#define JDET_FWD(NW, SG, ABL)                                                                       \
  hipLaunchKernelGGL((roi_align_fwd_vec_kernel<V, NW, SG, ABL>), grid, dim3(NW * 64), lds, st, feat, \
                     rois, out, C, H, W, PH, PW, scale, sample_num, order)
    if (abl == 1 && nw == 8) JDET_FWD(8, 4, 1);
    else if (abl == 2 && nw == 8) JDET_FWD(8, 4, 2);
    else if (abl == 3 && nw == 8) JDET_FWD(8, 4, 3);
    else if (abl == 4 && nw == 8) JDET_FWD(8, 4, 4);
    else if (abl == 7 && nw == 8) JDET_FWD(8, 4, 7);
    else if (abl == 1) JDET_FWD(4, 4, 1);
    else if (abl == 2) JDET_FWD(4, 4, 2);
    else if (abl == 3) JDET_FWD(4, 4, 3);
    else if (abl == 4) JDET_FWD(4, 4, 4);
    else if (abl == 7) JDET_FWD(4, 4, 7);
    else if (nw == 8 && sg == 8) JDET_FWD(8, 8, 0);
    else if (nw == 8) JDET_FWD(8, 4, 0);
    else if (nw == 16) JDET_FWD(16, 4, 0);
    else if (sg == 8) JDET_FWD(4, 8, 0);
    else if (sg == 2) JDET_FWD(4, 2, 0);
    else JDET_FWD(4, 4, 0);
#undef JDET_FWD
  } else if (C % 4 == 0 && VARIANT != JDET_ROI_RIROI) {
    hipLaunchKernelGGL((roi_align_fwd_kernel<VARIANT, 0>), grid, dim3(kBlock), lds, st, feat, rois, out,
                       C, H, W, PH, PW, scale, sample_num, nO, order);
  } else {
    hipLaunchKernelGGL((roi_align_fwd_kernel<VARIANT, 1>), grid, dim3(kBlock), lds, st, feat, rois, out,
                       C, H, W, PH, PW, scale, sample_num, nO, order);
  }
  return jdet_launch_status();
}

template <int VARIANT>
int launch_bwd(const float* gout, const float* rois, float* gin, int R, int C, int H, int W, int PH,
               int PW, float scale, int sample_num, int nO, const int32_t* order, hipStream_t st) {
  const int chunks = jdet_cdiv(C, kChunkC);
  const size_t lds = (size_t)min(C, kChunkC) * PH * PW * sizeof(float);
  hipLaunchKernelGGL((roi_align_bwd_kernel<VARIANT>), dim3(R, chunks), dim3(kBlock), lds, st, gout,
                     rois, gin, C, H, W, PH, PW, scale, sample_num, nO, order);
  return jdet_launch_status();
}


// ---- channel-sliced forward (roi_align_sliced.h) ----
bool sliced_ok(int variant, int R, int N, int C, int H, int W, int PH, int PW, int sample_num, int nO) {
  // jdet_set_roi_forward_mode(2) (or JDET_ROI_FWD_SLICED=1 for profiling runs) selects the channel-sliced kernels.
  // They are NOT the default: measured slower than the RoI-stationary kernels (profiles/r04_roi_fwd_notes.md).
  static const int sliced_env = env_int("JDET_ROI_FWD_SLICED", 0);
  if (!(g_fwd_reference_order == 2 || (sliced_env && g_fwd_reference_order == 0)) || sample_num != 2) return false;
  const long nbins = (long)PH * PW;
  if (nbins < jdet_roi_sliced::kItemsPerWave || C % jdet_roi_sliced::kSliceC != 0) return false;
  if ((size_t)N * H * W * C * 4 >= (1ull << 31) || (long)R * nbins >= (1L << 30)) return false;
  if (variant == JDET_ROI_RIROI && nO != 4 && nO != 8) return false;
  return true;
}

template <int VARIANT, int NO>
int launch_sliced(const float* feat, const float* rois, float* out, int R, int N, int C, int H, int W, int PH, int PW,
                  float scale, int nO, void* ws, hipStream_t st) {
  using namespace jdet_roi_sliced;
  const int nbins = PH * PW;
  const PlanWs w = plan_carve(ws, R, nbins);
  // EXPERIMENT (profiling): JDET_ROI_SLICED_PLANAR=1 reads `feat` as [slice][pixel][32 channels] (every slice one
  // contiguous plane) instead of NHWC -- the caller must pass a map permuted that way
  static const int planar = env_int("JDET_ROI_SLICED_PLANAR", 0);
  const int pix_bytes = planar ? kSliceC * 4 : C * 4;
  const unsigned slice_stride = planar ? (unsigned)((size_t)N * H * W * kSliceC * 4) : (unsigned)(kSliceC * 4);
  hipLaunchKernelGGL((roi_sort_plan_kernel<VARIANT>), dim3(1 + (R + 3) / 4), dim3(1024), 0, st, rois, R, scale, N,
                     pix_bytes, H, W, PH, PW, nO, w.hdr, w.order, w.rrec, w.ent);
  const int nslices = C / kSliceC;
  const long items = (long)R * nbins;
#define JDET_SL(B_, P_, NW_)                                                                                          \
  hipLaunchKernelGGL((roi_pool_sliced_kernel<NO, B_, P_, NW_>),                                                       \
                     dim3((unsigned)(nslices * ((items + NW_ * kItemsPerWave - 1) / (NW_ * kItemsPerWave)))),         \
                     dim3(NW_ * 64), 0, st, feat, w.order, w.rrec, w.ent, out, R, N, C, H * W, nbins, nslices, slice_stride)
  if constexpr (NO == 0) {   // tuning knobs (profiling runs)
    static const int batch = env_int("JDET_ROI_SLICED_BATCH", 8), pred = env_int("JDET_ROI_SLICED_PRED", 0),
                     nw = env_int("JDET_ROI_SLICED_WAVES", 4);
    if (nw == 16 && batch == 4 && pred == 1) JDET_SL(4, 1, 16);
    else if (nw == 16) JDET_SL(8, 0, 16);
    else if (nw == 8 && batch == 4 && pred == 1) JDET_SL(4, 1, 8);
    else if (nw == 1 && batch == 4 && pred == 1) JDET_SL(4, 1, 1);
    else if (batch == 4 && pred == 0) JDET_SL(4, 0, 4);
    else if (batch == 16 && pred == 0) JDET_SL(16, 0, 4);
    else if (batch == 4 && pred == 1) JDET_SL(4, 1, 4);
    else if (batch == 8 && pred == 1) JDET_SL(8, 1, 4);
    else JDET_SL(8, 0, 4);
  } else {
    JDET_SL(8, 0, 4);
  }
#undef JDET_SL
  return jdet_launch_status();
}

int check_common(int variant, const void* a, const void* b, const void* c, int N, int C, int H,
                 int W, int R, int PH, int PW, int n_orient) {
  if (variant < 0 || variant > 4) return JDET_E_BADARG;
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || PH <= 0 || PW <= 0) return JDET_E_BADARG;
  if (R > 0 && (!a || !b || !c)) return JDET_E_BADARG;
  if (PH * PW > 256) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && (n_orient <= 0 || C % n_orient != 0)) return JDET_E_BADARG;
  if ((long)H * W >= (1L << 30)) return JDET_E_UNSUPPORTED;
  return JDET_OK;
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

JDET_API int jdet_set_roi_forward_mode(int mode) {
  const int prev = g_fwd_reference_order;
  if (mode >= 0 && mode <= 3) g_fwd_reference_order = mode;
  return prev;
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
  int e = check_common(variant, feat, rois, out, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (R == 0) return JDET_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_fwd<JDET_ROI_ROTATED>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    case JDET_ROI_ROTATED_V1:
      return launch_fwd<JDET_ROI_ROTATED_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    case JDET_ROI_RIROI:
      return launch_fwd<JDET_ROI_RIROI>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, order, st);
    case JDET_ROI_HBB_V0:
      return launch_fwd<JDET_ROI_HBB_V0>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
    default:
      return launch_fwd<JDET_ROI_HBB_V1>(feat, rois, out, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st);
  }
}

// RoI-stationary forward with a channels-last result (R, PH, PW, C): same kernels, results stored straight from
// registers (one contiguous 1 KiB row chunk per wave and bin) instead of being transposed through LDS.
JDET_API int jdet_roi_align_forward_cl_roi(int variant, const float* feat, int N, int C, int H, int W,
                                           const float* rois, int R, int PH, int PW, float spatial_scale,
                                           int sample_num, int n_orient, const int32_t* order, float* out_cl,
                                           jdet_stream_t stream) {
  int e = check_common(variant, feat, rois, out_cl, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (C % 4 != 0 || (size_t)H * W * C * 4 >= (1ull << 31)) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && n_orient != 4 && n_orient != 8) return JDET_E_UNSUPPORTED;
  if (R == 0) return JDET_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return launch_fwd<JDET_ROI_ROTATED>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true);
    case JDET_ROI_ROTATED_V1:
      return launch_fwd<JDET_ROI_ROTATED_V1>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true);
    case JDET_ROI_RIROI:
      return launch_fwd<JDET_ROI_RIROI>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, n_orient, order, st, true);
    case JDET_ROI_HBB_V0:
      return launch_fwd<JDET_ROI_HBB_V0>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true);
    default:
      return launch_fwd<JDET_ROI_HBB_V1>(feat, rois, out_cl, R, C, H, W, PH, PW, spatial_scale, sample_num, 1, order, st, true);
  }
}

// Product forward with a channels-last result: the channel-sliced kernels (roi_align_sliced.h) where they apply
// (sampling 2, PH*PW >= 16, C % 32 == 0, default arithmetic mode), otherwise the RoI-stationary kernels above under the
// XCD-aware spatial order.  The schedule / per-RoI records live in the caller's workspace.
JDET_API size_t jdet_roi_align_forward_cl_workspace(int R, int PH, int PW) {
  if (R <= 0 || PH <= 0 || PW <= 0) return 256;
  // forward mode 2 (channel-sliced kernels): schedule + plan; otherwise the two int32 arrays of the spatial order
  static const int sliced_env = env_int("JDET_ROI_FWD_SLICED", 0);
  if (g_fwd_reference_order == 2 || sliced_env) return jdet_roi_sliced::plan_carve(nullptr, R, (long)PH * PW).bytes;
  return 256 + 2 * sizeof(int32_t) * (size_t)R;
}

JDET_API int jdet_roi_align_forward_cl(int variant, const float* feat, int N, int C, int H, int W, const float* rois,
                                       int R, int PH, int PW, float spatial_scale, int sample_num, int n_orient,
                                       float* out_cl, void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  int e = check_common(variant, feat, rois, out_cl, N, C, H, W, R, PH, PW, n_orient);
  if (e) return e;
  if (C % 4 != 0 || (size_t)H * W * C * 4 >= (1ull << 31)) return JDET_E_UNSUPPORTED;
  if (variant == JDET_ROI_RIROI && n_orient != 4 && n_orient != 8) return JDET_E_UNSUPPORTED;
  if (R == 0) return JDET_OK;
  if (!workspace || workspace_bytes < jdet_roi_align_forward_cl_workspace(R, PH, PW)) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  if (sliced_ok(variant, R, N, C, H, W, PH, PW, sample_num, n_orient)) {
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
  const int32_t* order = nullptr;
  if (R >= 64) {   // below that the map traffic is too small for the schedule to matter
    int32_t* o = (int32_t*)workspace;
    const int cols = (variant == JDET_ROI_HBB_V0 || variant == JDET_ROI_HBB_V1) ? 5 : 6;
    e = jdet_roi_spatial_order(rois, R, cols, spatial_scale, N, H, W, o, o + R, stream);
    if (e) return e;
    order = o;
  }
  return jdet_roi_align_forward_cl_roi(variant, feat, N, C, H, W, rois, R, PH, PW, spatial_scale, sample_num, n_orient,
                                       order, out_cl, stream);
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
