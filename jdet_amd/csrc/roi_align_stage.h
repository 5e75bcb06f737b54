// Footprint-staged RoIAlign forward (round 6): every distinct pixel of a LINE of bins is fetched ONCE, by LDS-DMA
// (buffer_load_dwordx4 ... lds: per-lane source address, 1 KiB per wave instruction straight into LDS, no VGPR round
// trip), and the taps of the line's bins are served from LDS with ds_read_b128.
//
// Why (profiles/r04_roi_fwd_notes.md, profiles/r06_dma_probe.txt): the merged-tap kernel moves 0.98 M pixel rows of
// 1 KiB through the vector L1 -- the texture path delivers ~25 TB/s of such rows whatever serves them and wherever they
// land, so the row count is the first-order term -- while the bench RoIs hold only 0.49 M distinct (line, pixel) pairs:
// a bin's four samples and the bins of a line revisit the same pixels whenever the sample spacing is under a pixel.
//
// One workgroup (4 waves) per RoI:
//   prologue, lane = sample (8 lines x 32 lanes; a line = the bins of one bin row / bin column, whichever runs along the
//   DENSER sample direction): bilinear geometry once per sample; per line the bounding box of its taps (packed u16
//   min / max butterflies), a bitmap over that box in LDS (atomic OR of the tap bits), a popcount prefix over the
//   bitmap words -- the RANK of a tap's bit is its slot: exact deduplication with no hash and no sort -- then per sample
//   a record (four LDS byte offsets + four weights) and per slot the pixel's byte offset in the map.
//   main loop over (group of lines, channel pass): consecutive lines are packed into groups of at most CAP slots; a pass
//   covers CPP channels (a slot row = CPP * 4 bytes).  Double buffered: the DMA of step n + 1 is issued right after the
//   barrier that opens step n, so a workgroup always has one buffer in flight while it computes from the other.  One DMA
//   instruction fetches 1024 / (4 CPP) slots (per-lane addresses: CPP / 4 lanes per slot); in the compute phase CPP / 4
//   lanes own a bin (4 channels per lane), so a wave works on 256 / CPP bins at once: per sample two broadcast reads of
//   the record, four ds_read_b128 and eight packed FMAs.  Results leave as CPP * 4-byte pieces of the channels-last row
//   (r, bin, :) with non-temporal stores.
// RoIs whose line bitmaps would not fit (a line's tap box over 224 words: sides beyond ~90 map pixels) take the direct
// path of the reference-order kernel inside the same launch.
//
// Arithmetic: weights as in the merged-tap kernel (reference formulas, pre-divided by the exact sample count 4), taps
// accumulated sample by sample in the reference's order with FMAs: equal to the reference-order twin up to fp32
// re-association (tests: <= 2e-6 on N(0,1) maps).  Reference: roi_align_rotated.py:L61-127, roi_align_rotated_v1.py:L71-145,
// roi_align.py:L93-204.
#pragma once

namespace jdet_roi_stage {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned short v2us __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int kMaxLines = 8;       // PH, PW <= 8
constexpr int kLineLanes = 32;     // lanes per line in the prologue (a line has 4 * bins <= 32 samples)
constexpr int kWordsPerLane = 7;
constexpr int kLineWords = kLineLanes * kWordsPerLane;   // bitmap words per line
constexpr int kCap = 128;          // slots per buffer (a line has at most 4 * 32 distinct pixels)
constexpr int kThreads = 256;

template <int CPP>
struct Layout {
  static constexpr int kRowB = CPP * 4;                 // bytes per slot
  static constexpr int kBufB = kCap * kRowB;            // one DMA buffer
  static constexpr int kRecW = 2 * kBufB;               // float4 [256]
  static constexpr int kRecA = kRecW + 256 * 16;        // uint2  [256]
  static constexpr int kSlotPix = kRecA + 256 * 8;      // int    [8 * 128]
  static constexpr int kMisc = kSlotPix + kMaxLines * kCap * 4;   // int P[8], int flag, float trig[2]; int4 gtab[8]
  static constexpr int kTotal = kMisc + 256;
  static_assert(2 * kBufB >= kMaxLines * kLineWords * 8, "prologue bitmaps overlay the DMA buffers");
};

__device__ __forceinline__ unsigned pk16(unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); }

__device__ __forceinline__ unsigned pk_min(unsigned a, unsigned b) {
  const v2us r = __builtin_elementwise_min(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b));
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) {
  const v2us r = __builtin_elementwise_max(__builtin_bit_cast(v2us, a), __builtin_bit_cast(v2us, b));
  return __builtin_bit_cast(unsigned, r);
}

// DPP lane exchanges (no LDS crossbar round trip): quad_perm xor 1 / xor 2, rotations inside a row of 16 lanes
template <int CTRL>
__device__ __forceinline__ unsigned dpp(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
// all-lanes reduction over aligned groups of 32 lanes of a min / max style (idempotent) operator
template <typename OP>
__device__ __forceinline__ unsigned reduce32(unsigned v, OP op) {
  v = op(v, dpp<0xB1>(v));      // quad_perm [1,0,3,2]
  v = op(v, dpp<0x4E>(v));      // quad_perm [2,3,0,1]
  v = op(v, dpp<0x124>(v));     // row_ror:4
  v = op(v, dpp<0x128>(v));     // row_ror:8
  return op(v, (unsigned)__shfl_xor((int)v, 16, 64));
}
// inclusive prefix sum over aligned groups of 32 lanes
__device__ __forceinline__ int scan32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);   // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);   // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);   // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);   // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
  return v;
}

// One LDS-DMA instruction: 64 lanes x 16 bytes from (descriptor, per-lane byte offset + scalar offset) to LDS bytes
// [lds_dst, lds_dst + 1024).  Inline assembly on purpose: with the builtin, hipcc (ROCm 7.2) orders EVERY later ds_read
// of the same LDS object behind the DMA with s_waitcnt vmcnt(0), which serialises the fetch of the next step with the
// compute of the current one; here the kernel counts its own vmcnt.  M0 is saved and restored (compiler-reserved).
__device__ __forceinline__ void dma16(v4i rs, unsigned lds_dst, int voff, int soff) {
  unsigned keep;
  asm volatile(
      "s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff)
      : "memory");
}

// s_waitcnt vmcnt(n), n wave-uniform: all but the n youngest vector-memory operations of this wave have completed
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
    case 0: __builtin_amdgcn_s_waitcnt(0x0f70); break;
    case 1: __builtin_amdgcn_s_waitcnt(0x0f71); break;
    case 2: __builtin_amdgcn_s_waitcnt(0x0f72); break;
    case 3: __builtin_amdgcn_s_waitcnt(0x0f73); break;
    case 4: __builtin_amdgcn_s_waitcnt(0x0f74); break;
    case 5: __builtin_amdgcn_s_waitcnt(0x0f75); break;
    case 6: __builtin_amdgcn_s_waitcnt(0x0f76); break;
    default: __builtin_amdgcn_s_waitcnt(0x0f77); break;   // (more than 7 younger operations: waits for a few of them too)
  }
}

// P7: PH == PW == 7 known at compile time (the configuration of every reference config: divisions by constants)
template <int VARIANT, int CPP, int P7>
__global__ __launch_bounds__(kThreads) void roi_align_fwd_staged_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int C, int H, int W, int PH_,
    int PW_, float spatial_scale, const int32_t* __restrict__ order, int abl) {
  using L = Layout<CPP>;
  constexpr int LPB = CPP / 4;         // lanes per slot (DMA) / per bin (compute)
  constexpr int NBW = 64 / LPB;        // slots per DMA instruction, bins per wave step
  const int PH = P7 ? 7 : PH_, PW = P7 ? 7 : PW_;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // the ONLY LDS object
  v4f* rec_w = reinterpret_cast<v4f*>(smem + L::kRecW);
  uint2* rec_a = reinterpret_cast<uint2*>(smem + L::kRecA);
  int* slotpix = reinterpret_cast<int*>(smem + L::kSlotPix);
  int* s_P = reinterpret_cast<int*>(smem + L::kMisc);            // [8] slots of a line
  int* s_flag = s_P + 8;
  float* s_trig = reinterpret_cast<float*>(s_P + 10);
  int4* s_gtab = reinterpret_cast<int4*>(smem + L::kMisc + 64);   // [8] {first line, lines, first flat slot, slots}
  uint2* bm = reinterpret_cast<uint2*>(smem);                    // [8][kLineWords] {bits, prefix}: overlays the buffers
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem

  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int nbins = PH * PW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr bool kRot = ROI_COLS == 6;
  const float* roi = rois + (size_t)r * ROI_COLS;
  if (kRot && threadIdx.x == 0) {
    if (abl & 32) {   // (profiling: single-precision trig)
      s_trig[0] = cosf(roi[5]);
      s_trig[1] = sinf(roi[5]);
    } else {
      s_trig[0] = (float)cos((double)roi[5]);
      s_trig[1] = (float)sin((double)roi[5]);
    }
  }
  long long t0 = 0, t1 = 0, t2 = 0;
  if (abl & 16) t0 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) *s_flag = 0;
  __amdgpu_buffer_rsrc_t rsrc;
  RoiGeom g = vec_prologue<VARIANT, false>(feat, rois, r, C, H, W, PH, PW, spatial_scale, 2, rsrc);
  if (g.batch < 0) return;
  v4i rs;   // the same descriptor as four SGPRs for the DMA statement
  {
    const unsigned long long img_bits = (unsigned long long)(feat + (size_t)g.batch * H * W * C);
    rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)img_bits);
    rs.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(img_bits >> 32) & 0xffffu));
    rs.z = __builtin_amdgcn_readfirstlane((int)((size_t)H * W * C * 4));
    rs.w = 0x00020000;
  }
  __syncthreads();
  if (kRot) {
    g.cosT = s_trig[0];
    g.sinT = s_trig[1];
  }

  if (abl & 16) t1 = __builtin_readcyclecounter();
  // ---- prologue: lane = sample --------------------------------------------------------------------------------------
  // lines along the denser sample direction: bin rows (d == 0: PH lines of PW bins) when bin_w <= bin_h
  const int d = __builtin_amdgcn_readfirstlane(g.bin_w <= g.bin_h ? 0 : 1);
  const int NL = P7 ? 7 : (d == 0 ? PH : PW), NBL = P7 ? 7 : (d == 0 ? PW : PH);
  const int SW = 2 * PW;                                   // samples per sample row
  const int ln = threadIdx.x >> 5, j = threadIdx.x & 31;   // my line, my sample within it
  const bool s_ok = ln < NL && j < 4 * NBL;
  const int v = s_ok && j >= 2 * NBL ? 1 : 0, u = s_ok ? j - v * 2 * NBL : 0;
  const int iy = d == 0 ? 2 * (s_ok ? ln : 0) + v : u;
  const int ix = d == 0 ? u : 2 * (s_ok ? ln : 0) + v;
  const int sid = iy * SW + ix;                            // natural sample id
  SamplePos p = sample_pos<VARIANT>(g, iy >> 1, ix >> 1, iy & 1, ix & 1, H, W);
  const bool live = s_ok && p.valid;
  // weights (the reference's formulas; pre-divided by the sample count: 4, exact)
  const float hy = (float)(1. - (double)p.ly), hx = (float)(1. - (double)p.lx);
  const float inv_count = 1.f / g.count;
  v4f wq = {hy * hx * inv_count, hy * p.lx * inv_count, p.ly * hx * inv_count, p.ly * p.lx * inv_count};
  if (!live) wq = v4f{0.f, 0.f, 0.f, 0.f};

  // bounding box of the line's taps: (x, y) packed as two u16, min / max over the line's 32 lanes
  const unsigned mn = reduce32(live ? pk16(p.x_low, p.y_low) : 0xffffffffu, [](unsigned a, unsigned b) { return pk_min(a, b); });
  const unsigned mx = reduce32(live ? pk16(p.x_high, p.y_high) : 0u, [](unsigned a, unsigned b) { return pk_max(a, b); });
  const bool line_live = mn != 0xffffffffu;
  const int x0 = mn & 0xffff, y0 = mn >> 16;
  const int wbits = line_live ? (int)(mx & 0xffff) - x0 + 1 : 0;
  const int rows = line_live ? (int)(mx >> 16) - y0 + 1 : 0;
  const int wpr = (wbits + 31) >> 5;
  const int nwords = rows * wpr;
  if (nwords > kLineWords) atomicOr(s_flag, 1);
  // clear the line's bitmap words
  uint2* mybm = bm + ln * kLineWords;
  if (nwords <= kLineWords)
    for (int k = j; k < nwords; k += kLineLanes) mybm[k] = make_uint2(0u, 0u);
  __syncthreads();                                                                   // (A)
  if (__builtin_amdgcn_readfirstlane(*s_flag)) {
    // a line's tap box does not fit its bitmap (RoI sides beyond ~90 map pixels): direct path of the reference-order
    // kernel (roi_align_impl.inc) for this RoI
    for (int c0 = 0; c0 < C; c0 += kChunkC)
      direct_chunk<VARIANT, 4, 4, 0, true, 0>(g, rsrc, c0, min(kChunkC, C - c0), C, H, W, PW, nbins, wave, lane, nullptr,
                                              out + (size_t)r * nbins * C);
    return;
  }
  int i_lo = 0, i_hi = 0, b_x = 0;      // word index of rows y_low / y_high, bit of x_low
  if (live) {
    b_x = p.x_low - x0;
    i_lo = (p.y_low - y0) * wpr;
    i_hi = (p.y_high - y0) * wpr;
    const int bh = p.x_high - x0;
    atomicOr(&mybm[i_lo + (b_x >> 5)].x, 1u << (b_x & 31));
    atomicOr(&mybm[i_lo + (bh >> 5)].x, 1u << (bh & 31));
    atomicOr(&mybm[i_hi + (b_x >> 5)].x, 1u << (b_x & 31));
    atomicOr(&mybm[i_hi + (bh >> 5)].x, 1u << (bh & 31));
  }
  __syncthreads();                                                                   // (B)
  // popcount prefix over the line's words: lane j owns words [7 j, 7 j + 7)
  {
    unsigned bits[kWordsPerLane];
    int tot = 0;
#pragma unroll
    for (int k = 0; k < kWordsPerLane; k++) {
      const int wd = j * kWordsPerLane + k;
      bits[k] = wd < nwords ? mybm[wd].x : 0u;
      tot += __builtin_popcount(bits[k]);
    }
    const int incl = scan32(tot);
    int run = incl - tot;
#pragma unroll
    for (int k = 0; k < kWordsPerLane; k++) {
      const int wd = j * kWordsPerLane + k;
      if (wd < nwords) mybm[wd].y = (unsigned)run;
      run += __builtin_popcount(bits[k]);
    }
    if (j == 31) s_P[ln] = ln < NL ? max(incl, 1) : 0;   // an empty line keeps one dummy slot
  }
  __syncthreads();                                                                   // (C)
  // groups of consecutive lines with at most kCap slots; my line's first flat slot and its base inside its group
  int my_flat = 0, my_gbase = 0, ngroups = 0;
  {
    const int4 pa = *reinterpret_cast<const int4*>(s_P), pb = *reinterpret_cast<const int4*>(s_P + 4);
    const int pl[8] = {pa.x, pa.y, pa.z, pa.w, pb.x, pb.y, pb.z, pb.w};
    int acc = 0, flat = 0, g0 = 0, gflat = 0, gi = 0;
#pragma unroll
    for (int l2 = 0; l2 < kMaxLines; l2++) {
      if (l2 < NL) {
        if (acc + pl[l2] > kCap) {
          if (threadIdx.x == 0) s_gtab[gi] = make_int4(g0, l2 - g0, gflat, acc);
          gi++;
          g0 = l2;
          gflat = flat;
          acc = 0;
        }
        if (l2 == ln) {
          my_flat = flat;
          my_gbase = acc;
        }
        acc += pl[l2];
        flat += pl[l2];
      }
    }
    if (threadIdx.x == 0) s_gtab[gi] = make_int4(g0, NL - g0, gflat, acc);
    ngroups = __builtin_amdgcn_readfirstlane(gi + 1);
  }
  // ranks -> record of my sample, pixel offsets of my taps' slots
  {
    const int pix_bytes = C * 4;
    if (live) {
      const unsigned below = (1u << (b_x & 31)) - 1u;
      const uint2 wl = mybm[i_lo + (b_x >> 5)], wh = mybm[i_hi + (b_x >> 5)];
      const int r_lo = (int)wl.y + __builtin_popcount(wl.x & below);
      const int r_hi = (int)wh.y + __builtin_popcount(wh.x & below);
      const int dx = p.x_high != p.x_low ? 1 : 0;
      slotpix[my_flat + r_lo] = (p.y_low * W + p.x_low) * pix_bytes;
      slotpix[my_flat + r_lo + dx] = (p.y_low * W + p.x_high) * pix_bytes;
      slotpix[my_flat + r_hi] = (p.y_high * W + p.x_low) * pix_bytes;
      slotpix[my_flat + r_hi + dx] = (p.y_high * W + p.x_high) * pix_bytes;
      const unsigned a_lo = (unsigned)(my_gbase + r_lo) * L::kRowB, a_hi = (unsigned)(my_gbase + r_hi) * L::kRowB;
      const unsigned sx = (unsigned)dx * L::kRowB;
      rec_a[sid] = make_uint2(pk16(a_lo, a_lo + sx), pk16(a_hi, a_hi + sx));
    } else if (s_ok) {
      const unsigned a0 = (unsigned)my_gbase * L::kRowB;     // weight 0 on the line's first slot (always fetched)
      rec_a[sid] = make_uint2(pk16(a0, a0), pk16(a0, a0));
    }
    if (s_ok) rec_w[sid] = wq;
    if (ln < NL && j == 0 && !line_live) slotpix[my_flat] = 0;   // the dummy slot of an empty line: pixel 0
  }
  __syncthreads();                                                                   // (D)

  if (abl & 16) t2 = __builtin_readcyclecounter();
  if (abl & 1) return;   // (profiling: prologue only)
  // ---- main loop over (group, channel pass), double buffered -------------------------------------------------------
  const int npass = C / CPP;
  const int nsteps = ngroups * npass;
  const int sub = lane / LPB, chl = lane % LPB;          // my slot / bin inside a wave step, my channel quad
  constexpr int kMaxDma = kCap / NBW / 4;                // DMA instructions of a wave per step
  int poff[kMaxDma];                                     // my lanes' pixel offsets of the group being fetched
  int f_nslots = 0;                                      // slots of the group being fetched
  auto load_group_offsets = [&](int gi) {
    const int4 gt = s_gtab[gi];
    f_nslots = __builtin_amdgcn_readfirstlane(gt.w);
    const int gflat = __builtin_amdgcn_readfirstlane(gt.z);
#pragma unroll
    for (int i = 0; i < kMaxDma; i++) {
      const int sl = (wave + 4 * i) * NBW + sub;
      poff[i] = sl < f_nslots ? slotpix[gflat + sl] + chl * 16 : -1;
    }
  };
  auto issue = [&](int buf, int pass) {      // DMA of the group in poff[], channel pass `pass` -> buffer buf
    if (abl & 2) return;
    const int soff = __builtin_amdgcn_readfirstlane(pass * L::kRowB);
    const unsigned dst = lds0 + buf * L::kBufB;
#pragma unroll
    for (int i = 0; i < kMaxDma; i++) {
      const int first = (wave + 4 * i) * NBW;            // wave-uniform
      if (first < f_nslots) {
        const int vo = poff[i];
        if (vo >= 0) dma16(rs, __builtin_amdgcn_readfirstlane(dst + first * L::kRowB), vo, soff);
      }
    }
  };
  load_group_offsets(0);
  issue(0, 0);
  float* __restrict__ out_r = out + (size_t)r * nbins * C;
  int c_l0 = 0, c_nl = 0;                                 // lines of the group being computed
  int gi = 0, pass = 0;                                   // group / pass of the step being computed
  int younger = 0;                                        // my stores issued after the DMA of the step about to be computed
  for (int step = 0; step < nsteps; step++) {
    wait_vmcnt(younger);                                  // my DMA of this step has landed
    __builtin_amdgcn_s_waitcnt(0xc07f);                   // lgkmcnt(0)
    __builtin_amdgcn_s_barrier();                         // everyone's has; everyone is done with the other buffer
    if (pass == 0) {
      const int4 gt = s_gtab[gi];
      c_l0 = __builtin_amdgcn_readfirstlane(gt.x);
      c_nl = __builtin_amdgcn_readfirstlane(gt.y);
    }
    if (step + 1 < nsteps) {
      if (pass + 1 == npass) {
        load_group_offsets(gi + 1);
        issue((step + 1) & 1, 0);
      } else {
        issue((step + 1) & 1, pass + 1);
      }
    }
    younger = 0;
    const char* buf = smem + (step & 1) * L::kBufB + chl * 16;
    const int nbg = (abl & 4) ? 0 : c_nl * NBL;           // bins of this group
    for (int k = wave; k * NBW < nbg; k += 4) {
      const int bi = k * NBW + sub;
      const bool b_ok = bi < nbg;
      const int bl = b_ok ? bi / NBL : 0, bp = b_ok ? bi - bl * NBL : 0;
      const int ph = d == 0 ? c_l0 + bl : bp, pw = d == 0 ? bp : c_l0 + bl;
      const int s00 = (2 * ph) * SW + 2 * pw;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int s = s00 + (q >> 1) * SW + (q & 1);
        const uint2 a = rec_a[s];
        const v4f w = rec_w[s];
        const v4f t0 = *reinterpret_cast<const v4f*>(buf + (a.x & 0xffffu));
        const v4f t1 = *reinterpret_cast<const v4f*>(buf + (a.x >> 16));
        const v4f t2 = *reinterpret_cast<const v4f*>(buf + (a.y & 0xffffu));
        const v4f t3 = *reinterpret_cast<const v4f*>(buf + (a.y >> 16));
#define JDET_STAGE_FMA(T, WT)                      \
  acc.x = __builtin_fmaf(WT, T.x, acc.x);          \
  acc.y = __builtin_fmaf(WT, T.y, acc.y);          \
  acc.z = __builtin_fmaf(WT, T.z, acc.z);          \
  acc.w = __builtin_fmaf(WT, T.w, acc.w);
        JDET_STAGE_FMA(t0, w.x)
        JDET_STAGE_FMA(t1, w.y)
        JDET_STAGE_FMA(t2, w.z)
        JDET_STAGE_FMA(t3, w.w)
#undef JDET_STAGE_FMA
      }
      if (!(abl & 8)) {
        if (b_ok)
          __builtin_nontemporal_store(
              acc, reinterpret_cast<v4f*>(out_r + (size_t)(ph * PW + pw) * C + pass * CPP + chl * 4));
        younger++;
      }
    }
    if (++pass == npass) {
      pass = 0;
      gi++;
    }
  }
  if ((abl & 16) && threadIdx.x == 0) {   // (profiling: shader-clock stamps of the phases + step count, over the RoI's first row)
    long long* dbg = reinterpret_cast<long long*>(out_r);
    dbg[0] = t0; dbg[1] = t1; dbg[2] = t2; dbg[3] = __builtin_readcyclecounter(); dbg[4] = nsteps;
    dbg[5] = blockIdx.x;
  }
}

}  // namespace jdet_roi_stage
