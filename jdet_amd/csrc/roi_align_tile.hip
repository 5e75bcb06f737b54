// Tile-stationary RoIAlign forward for gfx950 (MI355X): ROIAlignRotated, ROIAlignRotated_v1 and the
// horizontal ROIAlign v0/v1 with a fixed 1x1 / 2x2 sampling grid, channels-last in AND out.
//
// Reference semantics (per output element, fp32): python/jdet/ops/roi_align_rotated.py:L21-127,
// roi_align_rotated_v1.py:L71-145, roi_align.py:L13-204 -- one CUDA thread per output element, 16 scattered
// NCHW reads each.
//
// Why another kernel.  The RoI-stationary kernels of roi_align.hip fetch every bilinear tap from the
// vector L1 / L2: 1.0 GB of tap traffic for a 67 MB map at the north-star point, 15x the map, and they are
// bound by the L2 -> L1 path (profiles/r01_roi_align_fwd_merged_nt_rocprofv3_summary.txt).  Here the MAP is
// stationary instead:
//   * one workgroup owns a TW x TH pixel tile of one image and a group of 32-channel chunks.  It loads the
//     tile plus a halo (the reach of a bin's samples around the bin centre) into LDS with plain coalesced
//     loads -- every map byte leaves HBM once, as a stream, no scheduling pre-pass -- and serves every tap of
//     every bin it owns from LDS (ds_read_b128: 256 B/clk/CU, four times the vector-L1 rate).
//   * a bin (RoI r, ph, pw) is owned by the tile that contains its (clamped) centre pixel.  Every
//     workgroup evaluates that pure function itself: scan the RoIs (bounding box vs tile) -> candidates ->
//     per-candidate 64-bit ownership mask (lane = bin) -> prefix sum -> bin list.  No atomics, no global
//     scratch, deterministic.
//   * per owned bin the 4 samples are reduced ONCE per workgroup to a 16-byte table entry
//     {packed LDS offset + step flags, ly, x-weight of the first-read pixel, x-weight of the second}; the
//     chunk loop re-uses the table for every 32-channel chunk, so the geometry cost is amortised over C.
//   * 8 lanes x float4 = one 128-byte pixel chunk; a wave works on 8 bins at a time.  The lane -> (bin, sub)
//     map follows the four 16-lane service groups of ds_read_b128, and the two bins that share a service
//     cycle read pixels of opposite parity first (the table pre-swaps left/right), so the two 128-byte
//     reads of a cycle hit disjoint bank halves.
//   * output is channels-last (R, PH, PW, C): the 32 channels of a bin are one contiguous 128-byte
//     non-temporal store; no LDS transposition.  The consumer (FC / RoI head) reads the same logical
//     (R, C, PH, PW) tensor through channels-last strides.
//   * the next chunk's window is prefetched into registers while the current one is consumed.
// Arithmetic: EXACT = the reference's operation order (w1*lt + w2*rt + w3*lb + w4*rb, samples iy-major,
// then / count; contraction off) -> bit-identical to the CPU oracle; otherwise the same weights applied with
// fma (<= a few ulp of sum |w v|).  Bins whose samples leave the halo (RoIs larger than the halo was sized
// for) take a per-bin slow path that reads its taps from global memory: any RoI size is handled.
#include <stdlib.h>

#include "roi_geom.h"

namespace {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kTNT = 512;  // threads per workgroup (8 waves), 2 workgroups per CU
constexpr int kTNW = kTNT / 64;
constexpr int kCK = 32;            // channels per chunk
constexpr int kPixB = kCK * 4;     // bytes per pixel in the LDS window
constexpr int kCandCap = 128;      // candidate RoIs per batch (records live in LDS)
constexpr int kBinCap = 224;       // owned bins per pass (table entries live in LDS)
constexpr int kSlowBit = 1 << 30;
static_assert(kCandCap <= 128, "the prefix step scans two candidates per lane of one wave");

struct CandRec {  // 48 B
  float center_w, center_h, start_w, start_h, bin_w, bin_h, cosT, sinT;
  int r;
  int pad[3];
};

template <int TW_, int TH_, int HLO_, int HHI_>
struct TileShape {
  static constexpr int TW = TW_, TH = TH_, HLO = HLO_, HHI = HHI_;
  static constexpr int WW = TW + HLO + HHI;   // window = tile + halo
  static constexpr int WH = TH + HLO + HHI;
  static constexpr int WWP = (WW + 1) & ~1;   // even row stride: a pixel and the one below share a parity
  static constexpr int NPX = WWP * WH;
  static constexpr int ZERO_PX = (NPX + 1) & ~1;  // two all-zero pixels (even, odd) for invalid samples
  static constexpr int WIN_BYTES = (ZERO_PX + 2) * kPixB;
  static constexpr int NSLOT = NPX * 8;       // 16-byte slots
  static constexpr int NPF = (NSLOT + kTNT - 1) / kTNT;
  static constexpr int LDS_BYTES = WIN_BYTES + kBinCap * 64 + kBinCap * 8 + kCandCap * (48 + 8) +
                                   (kCandCap + 4) * 4 + 2 * kTNW * 4 + 16;
  static_assert(WIN_BYTES <= 65536, "table entries hold 16-bit byte offsets");
  static_assert(kCandCap * 4 <= kBinCap * 64, "candidate index list overlays the table");
};

template <int VARIANT, class TS, bool EXACT>
__global__ __launch_bounds__(kTNT, 4) void roi_align_tile_fwd_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int N, int C, int H,
    int W, int R, int PH, int PW, float spatial_scale, int S, int tilesX, int tilesY, int cpg, int ngroups,
    int total) {
  constexpr bool kHbb = VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1;
  constexpr int ROI_COLS = kHbb ? 5 : 6;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f* s_tab = reinterpret_cast<v4f*>(smem + TS::WIN_BYTES);
  int* s_row = reinterpret_cast<int*>(s_tab + kBinCap * 4);
  int* s_bin = s_row + kBinCap;
  CandRec* s_cand = reinterpret_cast<CandRec*>(s_bin + kBinCap);
  u64* s_mask = reinterpret_cast<u64*>(s_cand + kCandCap);
  int* s_off = reinterpret_cast<int*>(s_mask + kCandCap);
  int* s_wcnt = s_off + kCandCap + 4;
  int* s_misc = s_wcnt + 2 * kTNW;
  int* s_cand_r = reinterpret_cast<int*>(s_tab);  // overlay: only live between the scan and the records

  // workgroup b runs on XCD b % 8 (observed; only speed depends on it): give every XCD one contiguous run of
  // (tile, channel group) work items so that neighbouring windows share their halos in that XCD's L2.
  const int per_xcd = (total + 7) >> 3;
  int wk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (wk >= total) return;
  const int grp_id = wk % ngroups;
  wk /= ngroups;
  const int tx = wk % tilesX;
  wk /= tilesX;
  const int ty = wk % tilesY;
  const int n = wk / tilesY;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int tx0 = tx * TS::TW, ty0 = ty * TS::TH;
  const int ox = tx0 - TS::HLO, oy = ty0 - TS::HLO;
  const int nbins = PH * PW;
  const int ns = S * S;  // 1 or 4
  const int nchunks = (C + kCK - 1) / kCK;
  const int chunk0 = grp_id * cpg;
  const int nch = min(cpg, nchunks - chunk0);

  // lane -> (bin slot of the wave, 16-byte sub-slot of the pixel chunk), following the ds_read_b128 service
  // groups {0-3,12-15,20-27} {4-11,16-19,28-31} (+32): each service cycle reads two complete 128-byte pixels,
  // one for the even ("P") and one for the odd ("Q") bin slot of a pair.
  int sgi, role, sub;
  {
    const int m = lane & 31;
    if (m < 4) { sgi = 0; role = 0; sub = m; }
    else if (m < 12) { sgi = 1; role = 0; sub = m - 4; }
    else if (m < 16) { sgi = 0; role = 0; sub = m - 8; }
    else if (m < 20) { sgi = 1; role = 1; sub = m - 16; }
    else if (m < 28) { sgi = 0; role = 1; sub = m - 20; }
    else { sgi = 1; role = 1; sub = m - 24; }
  }
  const int grp = (lane >> 5) * 4 + sgi * 2 + role;

  // raw buffer over image n (out-of-range offsets read 0)
  const float* img = feat + (size_t)n * H * W * C;
  const u64 img_bits = (u64)img;
  const unsigned img_lo = __builtin_amdgcn_readfirstlane((unsigned)img_bits);
  const unsigned img_hi = __builtin_amdgcn_readfirstlane((unsigned)(img_bits >> 32));
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>((const void*)(((u64)img_hi << 32) | img_lo)), 0,
      __builtin_amdgcn_readfirstlane((int)((size_t)H * W * C * 4)), 0x00020000);

  // window slot i = 16 bytes: pixel i >> 3 (row-major over WH x WWP), sub-slot i & 7
  int voff[TS::NPF];
#pragma unroll
  for (int k = 0; k < TS::NPF; k++) {
    const int i = tid + k * kTNT;
    const int px = i >> 3, sb = i & 7;
    const int wy = px / TS::WWP, wx = px - wy * TS::WWP;
    const int gy = oy + wy, gx = ox + wx;
    const bool ok = i < TS::NSLOT && wx < TS::WW && gy >= 0 && gy < H && gx >= 0 && gx < W;
    voff[k] = ok ? ((gy * W + gx) * C + sb * 4) * 4 : 0x7FFFFFF0;
  }
  v4f pf[TS::NPF];
  auto prefetch = [&](int chunk) {
    const int soff = __builtin_amdgcn_readfirstlane(chunk * kCK * 4);
#pragma unroll
    for (int k = 0; k < TS::NPF; k++)
      pf[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], soff, 0));
  };
  auto store_window = [&]() {
#pragma unroll
    for (int k = 0; k < TS::NPF; k++) {
      const int i = tid + k * kTNT;
      if (i < TS::NSLOT) *reinterpret_cast<v4f*>(smem + i * 16) = pf[k];
    }
  };
  auto lds4 = [&](int byte_off) -> v4f { return *reinterpret_cast<const v4f*>(smem + byte_off); };

  if (tid < 16) *reinterpret_cast<v4f*>(smem + TS::ZERO_PX * kPixB + tid * 16) = v4f{0.f, 0.f, 0.f, 0.f};
  prefetch(chunk0);

  auto clampx = [&](float v) -> int { return (int)fminf(fmaxf(floorf(v), 0.f), (float)(W - 1)); };
  auto clampy = [&](float v) -> int { return (int)fminf(fmaxf(floorf(v), 0.f), (float)(H - 1)); };
  auto geom_of = [&](const CandRec& c) -> RoiGeom {
    RoiGeom g;
    g.batch = n;
    g.center_w = c.center_w; g.center_h = c.center_h;
    g.start_w = c.start_w; g.start_h = c.start_h;
    g.bin_w = c.bin_w; g.bin_h = c.bin_h;
    g.cosT = c.cosT; g.sinT = c.sinT;
    g.grid_h = g.grid_w = S;
    g.count = (float)ns;
    g.l_var = 0.f; g.r_var = 1.f; g.ind = 0;
    return g;
  };
  // conservative: may this RoI own a bin whose centre pixel lies in this tile?
  auto cand_test = [&](int r) -> bool {
    const float* p = rois + (size_t)r * ROI_COLS;
    if ((int)p[0] != n) return false;
    float x_lo, x_hi, y_lo, y_hi;
    if (kHbb) {
      const float sw = p[1] * spatial_scale, sh = p[2] * spatial_scale;
      float rw, rh;
      if (VARIANT == JDET_ROI_HBB_V1) {
        rw = fmaxf((p[3] + 1) * spatial_scale - sw, 0.f);
        rh = fmaxf((p[4] + 1) * spatial_scale - sh, 0.f);
      } else {
        rw = fmaxf(p[3] * spatial_scale - sw, 1.f);
        rh = fmaxf(p[4] * spatial_scale - sh, 1.f);
      }
      x_lo = sw; x_hi = sw + rw; y_lo = sh; y_hi = sh + rh;
    } else {
      float cx = p[1] * spatial_scale, cy = p[2] * spatial_scale;
      if (VARIANT == JDET_ROI_ROTATED_V1) { cx -= 0.5f; cy -= 0.5f; }
      const float rw = fmaxf(p[3] * spatial_scale, 1.f), rh = fmaxf(p[4] * spatial_scale, 1.f);
      float sn, cs;
      sincosf(p[5], &sn, &cs);
      sn = fabsf(sn); cs = fabsf(cs);
      const float ex = 0.5f * (rw * cs + rh * sn), ey = 0.5f * (rw * sn + rh * cs);
      x_lo = cx - ex; x_hi = cx + ex; y_lo = cy - ey; y_hi = cy + ey;
    }
    const float slack = 1.5f + 1e-5f * (fabsf(x_lo) + fabsf(x_hi) + fabsf(y_lo) + fabsf(y_hi));
    return clampx(x_hi + slack) >= tx0 && clampx(x_lo - slack) < tx0 + TS::TW &&
           clampy(y_hi + slack) >= ty0 && clampy(y_lo - slack) < ty0 + TS::TH;
  };

  const float inv_count = 1.f / (float)ns;  // 1 or 0.25: exact, and equal to the reference's `/ count`

  // all taps of the owned bins of this pass from the LDS window (table-driven), then the slow bins from global
  auto compute = [&](int nb, int chunk) {
    const int cbase = chunk * kCK;
    const bool ch_ok = cbase + sub * 4 < C;
    for (int base = wave * 8; base < nb; base += kTNW * 8) {
      const int e = base + grp;
      if (e >= nb) continue;
      const int bw = s_bin[e];
      if (bw & kSlowBit) continue;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      // always 4 table entries (a 1x1 grid fills entries 1..3 with weight 0 on the zero pixel).  Two samples
      // at a time: 2 table reads, then 8 tap reads in flight (the register budget is 128 for 2 workgroups / CU)
#pragma nounroll
      for (int h = 0; h < 2; h++) {
        const v4f t0 = s_tab[e * 4 + 2 * h], t1 = s_tab[e * 4 + 2 * h + 1];
        v4f tp[2][4];
        float wq[2][4];
        int swp[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const v4f t = u ? t1 : t0;
          const int pk = __float_as_int(t.x);
          const float ly = t.y, wA = t.z, wB = t.w;
          const int a1 = (pk & 0xFFFF) + sub * 16;
          const int dx = ((pk >> 16) & 1) * kPixB - ((pk >> 17) & 1) * (2 * kPixB);
          const int dy = ((pk >> 18) & 1) * (TS::WWP * kPixB);
          tp[u][0] = lds4(a1);
          tp[u][1] = lds4(a1 + dx);
          tp[u][2] = lds4(a1 + dy);
          tp[u][3] = lds4(a1 + dy + dx);
          const float hy = EXACT ? (float)(1. - (double)ly) : 1.f - ly;   // same value
          wq[u][0] = hy * wA; wq[u][1] = hy * wB; wq[u][2] = ly * wA; wq[u][3] = ly * wB;
          swp[u] = (pk >> 17) & 1;   // first-read pixel is the RIGHT one
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          if (EXACT) {
            // (w1*lt + w2*rt) commutes bit for bit; the bottom row is added left then right
            const v4f top = wq[u][0] * tp[u][0] + wq[u][1] * tp[u][1];
            const v4f pf_ = wq[u][2] * tp[u][2], ps_ = wq[u][3] * tp[u][3];
            acc += (top + (swp[u] ? ps_ : pf_)) + (swp[u] ? pf_ : ps_);
          } else {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
              for (int k = 0; k < 4; k++) acc[k] = __builtin_fmaf(wq[u][q], tp[u][q][k], acc[k]);
          }
        }
      }
      acc *= inv_count;
      if (ch_ok) __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + (size_t)s_row[e] * C + cbase + sub * 4));
    }
    for (int base = wave * 8; base < nb; base += kTNW * 8) {
      const int e = base + grp;
      if (e >= nb) continue;
      const int bw = s_bin[e];
      if (!(bw & kSlowBit)) continue;
      const int ci = (bw & (kSlowBit - 1)) >> 8, bin = bw & 255;
      const RoiGeom g = geom_of(s_cand[ci]);
      const int ph = bin / PW, pw = bin - ph * PW;
      v4f acc = {0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < ns; s++) {
        const Sample sm = make_sample<VARIANT>(g, ph, pw, S == 2 ? s >> 1 : 0, S == 2 ? s & 1 : 0, H, W);
        if (!sm.valid) continue;
        const int cb = (cbase + sub * 4) * 4;
        auto tap = [&](int o) -> v4f {
          return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, o * C * 4 + cb, 0, 0));
        };
        const v4f lt = tap(sm.o1), rt = tap(sm.o2), lb = tap(sm.o3), rb = tap(sm.o4);
        if (EXACT) {
          acc += sm.w1 * lt + sm.w2 * rt + sm.w3 * lb + sm.w4 * rb;
        } else {
#pragma unroll
          for (int k = 0; k < 4; k++) {
            acc[k] = __builtin_fmaf(sm.w1, lt[k], acc[k]);
            acc[k] = __builtin_fmaf(sm.w2, rt[k], acc[k]);
            acc[k] = __builtin_fmaf(sm.w3, lb[k], acc[k]);
            acc[k] = __builtin_fmaf(sm.w4, rb[k], acc[k]);
          }
        }
      }
      acc *= inv_count;
      if (ch_ok) __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + (size_t)s_row[e] * C + cbase + sub * 4));
    }
  };

  int r_begin = 0;
  while (true) {
    // ---- scan: RoIs [r_begin, R) in rounds of kTNT, candidates kept in RoI order, at most kCandCap per batch
    int ncand = 0, next_begin = R, round = 0;
    for (int r0 = r_begin; r0 < R; r0 += kTNT, round++) {
      const int r = r0 + tid;
      const bool cand = r < R && cand_test(r);
      const u64 bal = __ballot(cand);
      if (lane == 0) s_wcnt[(round & 1) * kTNW + wave] = __popcll(bal);
      __syncthreads();
      int before = 0, tot = 0;
#pragma unroll
      for (int w2 = 0; w2 < kTNW; w2++) {
        const int c = s_wcnt[(round & 1) * kTNW + w2];
        tot += c;
        before += w2 < wave ? c : 0;
      }
      const int rank = ncand + before + __popcll(bal & ((1ull << lane) - 1ull));
      if (cand) {
        if (rank < kCandCap) s_cand_r[rank] = r;
        else if (rank == kCandCap) s_misc[0] = r;   // first RoI that did not fit: the next batch starts here
      }
      if (ncand + tot > kCandCap) {
        ncand = kCandCap;
        next_begin = -1;
        break;
      }
      ncand += tot;
    }
    __syncthreads();
    if (next_begin < 0) next_begin = s_misc[0];

    // ---- per-candidate geometry (double-precision trig once per candidate, as the reference-order kernels)
    if (tid < ncand) {
      const int r = s_cand_r[tid];
      const RoiGeom g = roi_geom<VARIANT, true>(rois + (size_t)r * ROI_COLS, spatial_scale, S, PH, PW, 1, false);
      CandRec c;
      c.center_w = g.center_w; c.center_h = g.center_h;
      c.start_w = g.start_w; c.start_h = g.start_h;
      c.bin_w = g.bin_w; c.bin_h = g.bin_h;
      c.cosT = g.cosT; c.sinT = g.sinT;
      c.r = r;
      c.pad[0] = c.pad[1] = c.pad[2] = 0;
      s_cand[tid] = c;
    }
    __syncthreads();

    // ---- ownership: lane = bin; a bin belongs to the tile holding its clamped centre pixel
    for (int ci = wave; ci < ncand; ci += kTNW) {
      const RoiGeom g = geom_of(s_cand[ci]);
      bool own = false;
      if (lane < nbins) {
        const int ph = lane / PW, pw = lane - ph * PW;
        const float yy = g.start_h + ((float)ph + 0.5f) * g.bin_h;
        const float xx = g.start_w + ((float)pw + 0.5f) * g.bin_w;
        float x, y;
        roi_xform<VARIANT>(g, xx, yy, x, y);
        const int px = clampx(x), py = clampy(y);
        own = px >= tx0 && px < tx0 + TS::TW && py >= ty0 && py < ty0 + TS::TH;
      }
      const u64 mk = __ballot(own);
      if (lane == 0) s_mask[ci] = mk;
    }
    __syncthreads();
    if (wave == 0) {
      const int c0 = 2 * lane < ncand ? __popcll(s_mask[2 * lane]) : 0;
      const int c1 = 2 * lane + 1 < ncand ? __popcll(s_mask[2 * lane + 1]) : 0;
      int incl = c0 + c1;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      const int excl = incl - (c0 + c1);
      s_off[2 * lane] = excl;
      s_off[2 * lane + 1] = excl + c0;
      if (lane == 63) s_misc[1] = incl;
    }
    __syncthreads();
    const int total_bins = s_misc[1];

    for (int pass0 = 0; pass0 < total_bins; pass0 += kBinCap) {
      const int nb = min(kBinCap, total_bins - pass0);
      // ---- bin list of this pass (position = rank of the bin in (candidate, bin) order)
      for (int ci = wave; ci < ncand; ci += kTNW) {
        const u64 mk = s_mask[ci];
        if ((mk >> lane) & 1ull) {
          const int idx = s_off[ci] + __popcll(mk & ((1ull << lane) - 1ull)) - pass0;
          if (idx >= 0 && idx < kBinCap) {
            s_bin[idx] = (ci << 8) | lane;
            s_row[idx] = s_cand[ci].r * nbins + lane;
          }
        }
      }
      __syncthreads();
      // ---- sample tables: thread = (bin entry, sample)
      for (int i = tid; i < nb * 4; i += kTNT) {
        const int e = i >> 2, s = i & 3;
        const int bw = s_bin[e];
        const int ci = (bw & (kSlowBit - 1)) >> 8, bin = bw & 255;
        const int rl = e & 1;
        int packed = (TS::ZERO_PX + rl) * kPixB;
        float ly = 0.f, wA = 0.f, wB = 0.f;
        if (s < ns) {
          const RoiGeom g = geom_of(s_cand[ci]);
          const int ph = bin / PW, pw = bin - ph * PW;
          const SamplePos p = sample_pos<VARIANT>(g, ph, pw, S == 2 ? s >> 1 : 0, S == 2 ? s & 1 : 0, H, W);
          if (p.valid) {
            const int wx = p.x_low - ox, wy = p.y_low - oy;
            const int xs = p.x_high - p.x_low, ys = p.y_high - p.y_low;
            if (wx >= 0 && wx + xs < TS::WW && wy >= 0 && wy + ys < TS::WH) {
              const float hx = (float)(1. - (double)p.lx);  // reference: `1. - lx` in double
              const int first = wy * TS::WWP + wx;
              // the even slot of a pair reads the even pixel of (left, right) first, the odd slot the odd one
              const int swp = (xs && ((first & 1) != rl)) ? 1 : 0;
              ly = p.ly;
              wA = swp ? p.lx : hx;
              wB = swp ? hx : p.lx;
              packed = ((first + swp) * kPixB) | (xs << 16) | (swp << 17) | (ys << 18);
            } else {
              atomicOr(&s_bin[e], kSlowBit);   // a tap outside the window: the whole bin reads from global
            }
          }
        }
        s_tab[i] = v4f{__int_as_float(packed), ly, wA, wB};
      }
      __syncthreads();
      // ---- chunk loop: window of chunk cc in LDS, chunk cc+1 in flight in registers
      for (int cc = 0; cc < nch; cc++) {
        store_window();
        __syncthreads();
        const bool last = cc + 1 == nch && pass0 + kBinCap >= total_bins && next_begin >= R;
        if (!last) prefetch(chunk0 + (cc + 1 == nch ? 0 : cc + 1));
        compute(nb, chunk0 + cc);
        __syncthreads();
      }
    }
    if (next_begin >= R) break;
    r_begin = next_begin;
  }
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

template <int VARIANT, class TS, bool EXACT>
int launch_tile(const float* feat, const float* rois, float* out, int N, int C, int H, int W, int R, int PH, int PW,
                float scale, int S, hipStream_t st) {
  auto kern = roi_align_tile_fwd_kernel<VARIANT, TS, EXACT>;
  static bool attr_set[64] = {};   // per device: > 64 KiB of dynamic LDS has to be opted into once
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return JDET_E_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TS::LDS_BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const int tilesX = jdet_cdiv(W, TS::TW), tilesY = jdet_cdiv(H, TS::TH);
  const int nchunks = jdet_cdiv(C, kCK);
  static const int cpg_env = env_int("JDET_ROI_TILE_CPG", 0);
  // channel chunks per workgroup: the sample tables are built once per workgroup, so more chunks amortise the
  // geometry; fewer chunks give more, smaller work items (balance).  Aim for >= ~2048 workgroups.
  int cpg = cpg_env > 0 ? cpg_env : 4;
  const long tiles = (long)N * tilesX * tilesY;
  while (cpg > 1 && tiles * jdet_cdiv(nchunks, cpg) < 2048) cpg >>= 1;
  if (cpg_env > 0) cpg = cpg_env;
  cpg = cpg < 1 ? 1 : (cpg > nchunks ? nchunks : cpg);
  const int ngroups = jdet_cdiv(nchunks, cpg);
  const long total = tiles * ngroups;
  if (total > (1L << 30)) return JDET_E_UNSUPPORTED;
  const int per_xcd = (int)((total + 7) / 8);
  hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(kTNT), TS::LDS_BYTES, st, feat, rois, out, N, C, H, W, R, PH, PW,
                     scale, S, tilesX, tilesY, cpg, ngroups, (int)total);
  return jdet_launch_status();
}

typedef TileShape<16, 8, 4, 5> Shape0;   // halo sized for bins up to 9.1 px (RoIs up to 64 px at 7x7)
typedef TileShape<16, 8, 3, 4> Shape1;
typedef TileShape<16, 8, 2, 3> Shape2;
typedef TileShape<12, 8, 4, 5> Shape3;

template <int VARIANT>
int dispatch_tile(int exact, const float* feat, const float* rois, float* out, int N, int C, int H, int W, int R,
                  int PH, int PW, float scale, int S, hipStream_t st) {
  if (exact) return launch_tile<VARIANT, Shape0, true>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, st);
  if (VARIANT == JDET_ROI_ROTATED) {   // tuning shapes are instantiated for one dialect only
    static const int shape = env_int("JDET_ROI_TILE_SHAPE", 0);
    if (shape == 1) return launch_tile<VARIANT, Shape1, false>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, st);
    if (shape == 2) return launch_tile<VARIANT, Shape2, false>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, st);
    if (shape == 3) return launch_tile<VARIANT, Shape3, false>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, st);
  }
  return launch_tile<VARIANT, Shape0, false>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, st);
}

}  // namespace

JDET_API int jdet_roi_align_forward_cl_supported(int variant, int C, int H, int W, int PH, int PW, int sample_num) {
  if (variant != JDET_ROI_ROTATED && variant != JDET_ROI_ROTATED_V1 && variant != JDET_ROI_HBB_V0 &&
      variant != JDET_ROI_HBB_V1)
    return 0;
  if (C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return 0;
  if (C % 4 != 0 || (sample_num != 1 && sample_num != 2) || PH * PW > 64) return 0;
  if ((size_t)H * W * C * 4 >= (1ull << 31)) return 0;
  return 1;
}

JDET_API int jdet_roi_align_forward_cl(int variant, const float* feat, int N, int C, int H, int W, const float* rois,
                                       int R, int PH, int PW, float spatial_scale, int sample_num, int exact_order,
                                       float* out, jdet_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || PH <= 0 || PW <= 0) return JDET_E_BADARG;
  if (variant < 0 || variant > 4) return JDET_E_BADARG;
  if (!jdet_roi_align_forward_cl_supported(variant, C, H, W, PH, PW, sample_num)) return JDET_E_UNSUPPORTED;
  if ((long)R * PH * PW >= (1L << 31)) return JDET_E_UNSUPPORTED;
  if (R == 0 || N == 0) return JDET_OK;
  if (!feat || !rois || !out) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return dispatch_tile<JDET_ROI_ROTATED>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, st);
    case JDET_ROI_ROTATED_V1:
      return dispatch_tile<JDET_ROI_ROTATED_V1>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, st);
    case JDET_ROI_HBB_V0:
      return dispatch_tile<JDET_ROI_HBB_V0>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, st);
    default:
      return dispatch_tile<JDET_ROI_HBB_V1>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, st);
  }
}
