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
// stationary instead: a workgroup stages a TW x TH pixel tile plus a halo (the reach of a bin's samples
// around the bin centre) of one 32-channel chunk in LDS with plain coalesced loads -- every map byte leaves
// HBM once, as a stream -- and serves every tap of the bins it owns from LDS (ds_read_b128: 256 B/clk/CU,
// four times the vector-L1 rate).  A bin (RoI r, ph, pw) is owned by the tile that contains its (clamped)
// centre pixel.
//
// Two launches:
//   1. roi_tile_plan_kernel  -- one workgroup per tile, no feature-map traffic.  Scan the RoI rows (staged
//      through LDS with coalesced loads; bounding circle vs tile) -> candidates -> per-candidate geometry
//      (double-precision trig, as the reference-order kernels) -> 64-bit ownership mask per candidate
//      (lane = bin) -> prefix sum -> the tile's bin list.  Per owned bin the 4 samples are reduced to four
//      16-byte table entries {packed LDS offset + step flags, ly, x-weight of the first-read pixel, x-weight of
//      the second}, written to a workspace together with the bin's output row.  Ranges are handed out by one
//      atomic cursor per launch; everything else is deterministic.
//   2. roi_align_tile_pool_kernel -- work item = (tile, channel group, part of <= kPart bins): heavy tiles are
//      split over several workgroups (the measured imbalance of one-workgroup-per-tile was 1.8x).  Copy the
//      part's table into LDS, then per 32-channel chunk: window of chunk c in LDS while chunk c+1 is in flight
//      in registers; 8 lanes x float4 = one 128-byte pixel chunk, a wave works on 8 bins at a time.  The
//      lane -> (bin, sub) map follows the four 16-lane service groups of ds_read_b128, and the two bins that
//      share a service cycle read pixels of opposite parity first (the table pre-swaps left / right), so the
//      two 128-byte reads of a cycle hit disjoint bank halves (measured: 0.6 % conflict cycles).
//      Output is channels-last (R, PH, PW, C): the 32 channels of a bin are one 128-byte non-temporal store.
// The plan depends only on (rois, geometry), not on the map: the backward pass can reuse it.
// Arithmetic: EXACT = the reference's operation order (w1*lt + w2*rt + w3*lb + w4*rb, samples iy-major,
// then / count; contraction off) -> bit-identical to the CPU oracle; otherwise the same weights applied with
// fma (<= a few ulp of sum |w v|).  Bins whose samples leave the halo (RoIs larger than the halo was sized
// for) take a per-bin slow path that reads its taps from global memory: any RoI size is handled.
#include <stdio.h>
#include <stdlib.h>

#include "roi_geom.h"

namespace {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kCK = 32;            // channels per chunk
constexpr int kPixB = kCK * 4;     // bytes per pixel in the LDS window
constexpr int kSlowBit = 1 << 30;  // in the row word: some tap of the bin lies outside the window
constexpr int kTabDw = 20;         // table dwords per bin: 4 samples x {packed, 4 weights}

// ------------------------------------------------------------------------------------------------------
// tile shape (shared by both kernels: the table entries hold window-relative byte offsets)
// ------------------------------------------------------------------------------------------------------
template <int TW_, int TH_, int HLO_, int HHI_>
struct TileShape {
  static constexpr int TW = TW_, TH = TH_, HLO = HLO_, HHI = HHI_;
  static constexpr int WW = TW + HLO + HHI;   // window = tile + halo
  static constexpr int WH = TH + HLO + HHI;
  static constexpr int WWP = (WW + 1) & ~1;   // even row stride: a pixel and the one below share a parity
  static constexpr int NPX = WWP * WH;
  // all-zero pixels for invalid samples: (even, odd) start pixels whose right and lower neighbours (every sample
  // reads base, base + 1 px, base + 1 row, base + 1 row + 1 px) are zero too
  static constexpr int ZERO_PX = (NPX + 1) & ~1;
  static constexpr int ZERO_N = WWP + 4;
  static constexpr int WIN_BYTES = (ZERO_PX + ZERO_N) * kPixB;
  static_assert(WIN_BYTES <= 65536, "table entries hold 16-bit byte offsets");
};

struct PlanHdr {   // per tile
  int offset;      // first bin of the tile in the plan arrays
  int count;       // bins owned by the tile
};

// workspace layout of one plan (all 256-byte aligned)
struct PlanWs {
  int* cursor;      // [0] bins handed out so far (zeroed before the plan kernel)
  PlanHdr* hdr;     // [ntiles]
  int* rows;        // [R * nbins]   output row (r * nbins + bin) | kSlowBit
  float* tab;       // [R * nbins * kTabDw]  five dwords per sample: {packed, w first-top, second-top, first-bottom, second-bottom}
  size_t bytes;
};

inline size_t a256(size_t x) { return (x + 255) & ~(size_t)255; }

inline PlanWs plan_carve(void* ws, long ntiles, long nbins_total) {
  PlanWs p;
  char* b = (char*)ws;
  size_t off = 0;
  p.cursor = (int*)(b + off);   off += 256;
  p.hdr = (PlanHdr*)(b + off);  off += a256(sizeof(PlanHdr) * ntiles);
  p.rows = (int*)(b + off);     off += a256(sizeof(int) * nbins_total);
  p.tab = (float*)(b + off);    off += a256(sizeof(float) * kTabDw * nbins_total);
  p.bytes = off;
  return p;
}

// upper bound of the plan positions: every bin once + one pad per tile and per candidate batch of a tile
inline long plan_bins_bound(long ntiles, long R, long nbins) { return R * nbins + ntiles * (R / 256 + 2); }

// ------------------------------------------------------------------------------------------------------
// 1. plan
// ------------------------------------------------------------------------------------------------------
constexpr int kPNT = 512;          // threads per plan workgroup
constexpr int kPNW = kPNT / 64;
constexpr int kCandCap = 256;      // candidate RoIs per batch
constexpr int kRoiBlock = 2048;    // RoI rows staged per scan block (48 KiB of dynamic LDS)
constexpr int kBinBlock = 512;     // owned bins whose tables are written per step
static_assert(kCandCap <= 256, "the prefix step scans four candidates per lane of one wave");

struct CandRec {  // 48 B
  float center_w, center_h, start_w, start_h, bin_w, bin_h, cosT, sinT;
  int r;
  int pad[3];
};

template <int VARIANT, class TS>
__global__ __launch_bounds__(kPNT) void roi_tile_plan_kernel(const float* __restrict__ rois, int H, int W, int R,
                                                            int PH, int PW, float spatial_scale, int S, int tilesX,
                                                            int tilesY, PlanHdr* __restrict__ hdr,
                                                            int* __restrict__ cursor, int* __restrict__ rows,
                                                            float* __restrict__ tab, int exact,
                                                            unsigned long long* __restrict__ dbg) {
  constexpr bool kHbb = VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1;
  constexpr int ROI_COLS = kHbb ? 5 : 6;
  extern __shared__ __attribute__((aligned(16))) float s_roi[];   // min(R, kRoiBlock) rows
  __shared__ CandRec s_cand[kCandCap];
  __shared__ u64 s_mask[kCandCap];
  __shared__ int s_off[kCandCap + 4];
  __shared__ int s_cand_r[kCandCap];
  __shared__ int s_bin[kBinBlock];
  __shared__ int s_wcnt[(kRoiBlock / kPNT) * kPNW];
  __shared__ int s_misc[4];

  int wk = blockIdx.x;
  const int tile_id = wk;
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
  const int ns = S * S;
  auto stamp = [&](int slot) {   // profiling hook (jdet_debug_roi_tile_timeline)
    if (dbg != nullptr && tid == 0) dbg[(size_t)blockIdx.x * 32 + slot] = __builtin_amdgcn_s_memtime();
  };
  stamp(0);

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
  // conservative: may this RoI (row rl of the staged block) own a bin whose centre pixel lies in this tile?
  auto cand_test = [&](int rl) -> bool {
    const float* p = s_roi + rl * ROI_COLS;
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
      // bounding circle of the rectangle: no trigonometry in the scan (the extra candidates are filtered
      // exactly by the ownership test)
      const float rad = 0.5f * sqrtf(rw * rw + rh * rh);
      x_lo = cx - rad; x_hi = cx + rad; y_lo = cy - rad; y_hi = cy + rad;
    }
    const float slack = 1.5f + 1e-5f * (fabsf(x_lo) + fabsf(x_hi) + fabsf(y_lo) + fabsf(y_hi));
    return clampx(x_hi + slack) >= tx0 && clampx(x_lo - slack) < tx0 + TS::TW &&
           clampy(y_hi + slack) >= ty0 && clampy(y_lo - slack) < ty0 + TS::TH;
  };

  const bool rois_vec = (((uintptr_t)rois) & 15) == 0;

  // One batch = up to kCandCap candidate RoIs starting at RoI r_begin: fills s_cand / s_mask / s_off and sets
  // (ncand, total_bins, next_begin = first RoI of the next batch or R).
  int ncand = 0, total_bins = 0, next_begin = R;
  auto build_batch = [&](int r_begin) {
    ncand = 0;
    next_begin = R;
    for (int r0 = r_begin; r0 < R && next_begin >= 0; r0 += kRoiBlock) {
      // stage the rows of RoIs [r0, r0 + kRoiBlock): coalesced 16-byte loads, all in flight together (one row
      // per lane straight from global is 24-byte strided: 12 cache lines per wave-load, and a scalar copy loop
      // pays one memory latency per element: measured 15-19 k cycles per workgroup for 2000 RoIs)
      {
        const int nfl = min(kRoiBlock, R - r0) * ROI_COLS;
        const float* src = rois + (size_t)r0 * ROI_COLS;
        constexpr int NV = (kRoiBlock * 6 / 4 + kPNT - 1) / kPNT;
        if (rois_vec && ((r0 * ROI_COLS) & 3) == 0) {
          const int nv = nfl >> 2;
          v4f tmp[NV];
#pragma unroll
          for (int k = 0; k < NV; k++) {
            const int i = tid + k * kPNT;
            tmp[k] = i < nv ? reinterpret_cast<const v4f*>(src)[i] : v4f{0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int k = 0; k < NV; k++) {
            const int i = tid + k * kPNT;
            if (i < nv) reinterpret_cast<v4f*>(s_roi)[i] = tmp[k];
          }
          if (tid < (nfl & 3)) s_roi[nv * 4 + tid] = src[nv * 4 + tid];
        } else {
          for (int i = tid; i < nfl; i += kPNT) s_roi[i] = src[i];
        }
      }
      __syncthreads();
      // all rounds of the block tested first, ONE barrier, then the ranks (candidates stay in RoI order)
      constexpr int KR = kRoiBlock / kPNT;
      u64 bal[KR];
#pragma unroll
      for (int k = 0; k < KR; k++) {
        const int rl = k * kPNT + tid;
        const bool cand = r0 + rl < R && cand_test(rl);
        bal[k] = __ballot(cand);
        if (lane == 0) s_wcnt[k * kPNW + wave] = __popcll(bal[k]);
      }
      __syncthreads();
      int runc = ncand, mybase[KR];
#pragma unroll
      for (int k = 0; k < KR; k++) {
#pragma unroll
        for (int w2 = 0; w2 < kPNW; w2++) {
          if (w2 == wave) mybase[k] = runc;
          runc += s_wcnt[k * kPNW + w2];
        }
      }
#pragma unroll
      for (int k = 0; k < KR; k++) {
        if ((bal[k] >> lane) & 1ull) {
          const int rank = mybase[k] + __popcll(bal[k] & ((1ull << lane) - 1ull));
          const int r = r0 + k * kPNT + tid;
          if (rank < kCandCap) s_cand_r[rank] = r;
          else if (rank == kCandCap) s_misc[0] = r;   // first RoI that did not fit: the next batch starts here
        }
      }
      if (runc > kCandCap) {
        ncand = kCandCap;
        next_begin = -1;
      } else {
        ncand = runc;
      }
      __syncthreads();   // the next block's staging overwrites the rows read above
    }
    __syncthreads();
    if (next_begin < 0) next_begin = s_misc[0];
    stamp(1);
    // per-candidate geometry (double-precision trig once per candidate)
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
    stamp(2);
    // ownership: lane = bin; a bin belongs to the tile holding its clamped centre pixel
    const int lane_ph = lane / PW, lane_pw = lane - lane_ph * PW;
    for (int ci = wave; ci < ncand; ci += kPNW) {
      const RoiGeom g = geom_of(s_cand[ci]);
      bool own = false;
      if (lane < nbins) {
        const float yy = g.start_h + ((float)lane_ph + 0.5f) * g.bin_h;
        const float xx = g.start_w + ((float)lane_pw + 0.5f) * g.bin_w;
        float x, y;
        roi_xform<VARIANT>(g, xx, yy, x, y);
        const int px = clampx(x), py = clampy(y);
        own = px >= tx0 && px < tx0 + TS::TW && py >= ty0 && py < ty0 + TS::TH;
      }
      const u64 mk = __ballot(own);
      if (lane == 0) s_mask[ci] = mk;
    }
    __syncthreads();
    if (wave == 0) {   // exclusive prefix of the popcounts: four candidates per lane
      int c[4], sum = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        c[j] = 4 * lane + j < ncand ? __popcll(s_mask[4 * lane + j]) : 0;
        sum += c[j];
      }
      int incl = sum;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
      }
      int run = incl - sum;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        s_off[4 * lane + j] = run;
        run += c[j];
      }
      if (lane == 63) s_misc[1] = incl;
    }
    __syncthreads();
    total_bins = s_misc[1];
    stamp(3);
  };

  // rows + table entries of the current batch at plan position `base`: the owned bins are compacted into a list
  // (kBinBlock at a time) and one thread computes one (bin, sample)
  // base < 0: the position was requested by alloc_begin() and is read from LDS behind the first barrier
  auto write_batch = [&](int base) {
    const float wscale = exact ? 1.f : 1.f / (float)ns;   // 1 or 0.25: folding `/ count` into the weights is exact
    for (int b0 = 0; b0 < total_bins; b0 += kBinBlock) {
      const int nb = min(kBinBlock, total_bins - b0);
      for (int ci = wave; ci < ncand; ci += kPNW) {
        const u64 mk = s_mask[ci];
        if ((mk >> lane) & 1ull) {
          const int e = s_off[ci] + __popcll(mk & ((1ull << lane) - 1ull)) - b0;
          if (e >= 0 && e < kBinBlock) s_bin[e] = (ci << 8) | lane;
        }
      }
      __syncthreads();
      if (base < 0) base = s_misc[2];
      for (int i = tid; i < nb * 4; i += kPNT) {   // nb * 4 <= 4 * kPNT: every wave runs the same trip count
        const int e = i >> 2, s = i & 3;
        const int bw = s_bin[e];
        const int ci = bw >> 8, bin = bw & 255;
        const int rl = (b0 + e) & 1;   // parity of the bin's position in the TILE list (batches and parts start even)
        int packed = (TS::ZERO_PX + rl) * kPixB;   // zero pixel, second-read delta 0
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        int swp = 0, slow = 0;
        if (s < ns) {
          const RoiGeom g = geom_of(s_cand[ci]);
          const int ph = bin / PW, pw = bin - ph * PW;
          const SamplePos p = sample_pos<VARIANT>(g, ph, pw, S == 2 ? s >> 1 : 0, S == 2 ? s & 1 : 0, H, W);
          if (p.valid) {
            // Every sample reads (x_low, y_low) and its right / lower / diagonal neighbours.  At the map border the
            // reference clamps x_high = x_low (y_high = y_low) AND sets lx = 0 (ly = 0): the clamped taps carry
            // weight exactly 0, and the window slot read instead lies outside the map = zero-filled.
            const int wx = p.x_low - ox, wy = p.y_low - oy;
            if (wx >= 0 && wx + 1 < TS::WW && wy >= 0 && wy + 1 < TS::WH) {
              const float hy = (float)(1. - (double)p.ly);  // reference: `1. - ly` in double
              const float hx = (float)(1. - (double)p.lx);
              w1 = hy * hx * wscale; w2 = hy * p.lx * wscale; w3 = p.ly * hx * wscale; w4 = p.ly * p.lx * wscale;
              const int first = wy * TS::WWP + wx;
              // the even slot of a pair reads the even pixel of (left, right) first, the odd slot the odd one
              // (not in reference-order mode: there the taps are combined in the reference's order)
              swp = (!exact && ((first & 1) != rl)) ? 1 : 0;
              packed = ((first + swp) * kPixB) | ((swp ? -kPixB : kPixB) << 16);
            } else {
              slow = 1;   // a tap outside the window: the whole bin reads from global
            }
          }
        }
        float* dst = tab + (size_t)(base + b0 + e) * kTabDw + 5 * s;
        dst[0] = __int_as_float(packed);
        dst[1] = swp ? w2 : w1;
        dst[2] = swp ? w1 : w2;
        dst[3] = swp ? w4 : w3;
        dst[4] = swp ? w3 : w4;
        // the 4 samples of a bin are 4 consecutive lanes
        slow |= __shfl_xor(slow, 1, 64);
        slow |= __shfl_xor(slow, 2, 64);
        if (s == 0) rows[base + b0 + e] = (s_cand[ci].r * nbins + bin) | (slow ? kSlowBit : 0);
      }
      __syncthreads();
    }
  };
  auto alloc_begin = [&](int count) {   // the atomic's round trip hides behind the bin-list step of write_batch
    if (tid == 0) {
      const int b = count > 0 ? atomicAdd(cursor, (count + 1) & ~1) : 0;
      s_misc[2] = b;
      PlanHdr h;
      h.offset = b;
      h.count = count;
      hdr[tile_id] = h;
    }
  };
  auto alloc = [&](int count) -> int {
    if (tid == 0) {
      // even start: position parity inside the tile list == global position parity
      const int b = count > 0 ? atomicAdd(cursor, (count + 1) & ~1) : 0;
      s_misc[2] = b;
      PlanHdr h;
      h.offset = b;
      h.count = count;
      hdr[tile_id] = h;
    }
    __syncthreads();
    return s_misc[2];
  };

  build_batch(0);
  if (next_begin >= R) {   // the common case: one batch
    alloc_begin(total_bins);
    stamp(4);
    write_batch(-1);
    stamp(5);
    return;
  }
  // more candidates than one batch holds: count over all batches, allocate once, then build them again and write.
  // (every batch but the last is padded to an even bin count so that a bin's parity is its list-position parity)
  int grand = (total_bins + 1) & ~1;
  for (int rb = next_begin; rb < R;) {
    __syncthreads();
    build_batch(rb);
    grand += next_begin >= R ? total_bins : (total_bins + 1) & ~1;
    rb = next_begin;
  }
  const int base = alloc(grand);
  int run = 0;
  for (int rb = 0; rb < R;) {
    __syncthreads();
    build_batch(rb);
    write_batch(base + run);
    if ((total_bins & 1) && next_begin < R && tid == 0) {   // pad entry: a masked row, never stored
      rows[base + run + total_bins] = -1;
    }
    run += (total_bins + 1) & ~1;
    rb = next_begin;
  }
}

// ------------------------------------------------------------------------------------------------------
// 2. pool
// ------------------------------------------------------------------------------------------------------
constexpr int kTNT = 512;    // threads per pool workgroup (8 waves), 2 workgroups per CU
constexpr int kTNW = kTNT / 64;
constexpr int kPart = 192;   // bins per pass (even: the table's pair parity is position parity)
constexpr int kMaxTiles = 1 << 20;

template <class TS>
struct PoolLds {
  static constexpr int NSLOT = TS::NPX * 8;   // 16-byte slots of the window
  static constexpr int NPF = (NSLOT + kTNT - 1) / kTNT;
  static constexpr int BYTES = TS::WIN_BYTES + kPart * kTabDw * 4 + kPart * 4;
};

// Work item = (part p of tile t, channel group): bins [p * kPart, (p + 1) * kPart) of the tile's list.  Items are
// dispatched part-major -- first the first parts of all tiles (the long items: up to kPart bins), then the second
// parts (remainders), ... -- so that the short items fill the slots freed by the early finishers (longest-first
// list scheduling; with tile-major order the measured slot utilisation was 54 %).  Parts beyond a tile's count exit
// at once; tiles with more than nparts * kPart bins loop.
template <int VARIANT, class TS, bool EXACT>
__global__ __launch_bounds__(kTNT, 4) void roi_align_tile_pool_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int C, int H, int W,
    int PH, int PW, float spatial_scale, int S, int tilesX, int tilesY, int ntiles, int cpg, int ngroups, int nparts,
    const PlanHdr* __restrict__ hdr, const int* __restrict__ rows, const float* __restrict__ tab,
    int* __restrict__ cursor, unsigned long long* __restrict__ dbg) {
  constexpr bool kHbb = VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1;
  constexpr int ROI_COLS = kHbb ? 5 : 6;
  constexpr int NPF = PoolLds<TS>::NPF, NSLOT = PoolLds<TS>::NSLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f* s_tab = reinterpret_cast<v4f*>(smem + TS::WIN_BYTES);
  int* s_row = reinterpret_cast<int*>(s_tab + kPart * (kTabDw / 4));

  // the plan's cursor is handed back zeroed (workspace contract: first 256 bytes zero on entry and on return)
  if (blockIdx.x == 0 && threadIdx.x == 0) *cursor = 0;

  // blockIdx -> (part, channel group, tile): part-major; inside one part, workgroup b runs on XCD b % 8 (observed;
  // only speed depends on it) and every XCD gets one contiguous band of tiles (halos of neighbours meet in its L2)
  const int per_part = ((ntiles * ngroups + 7) >> 3) << 3;
  const int part = blockIdx.x / per_part;
  const int b_in = blockIdx.x - part * per_part;
  const int per_xcd = per_part >> 3;
  int wk = (b_in & 7) * per_xcd + (b_in >> 3);
  if (wk >= ntiles * ngroups) return;
  const int grp_id = wk % ngroups;
  const int tile_id = wk / ngroups;
  const PlanHdr th = hdr[tile_id];
  if (part * kPart >= th.count) return;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  auto stamp = [&](int slot) {   // profiling hook (jdet_debug_roi_tile_timeline)
    if (dbg != nullptr && tid == 0 && slot < 32) dbg[(size_t)blockIdx.x * 32 + slot] = __builtin_amdgcn_s_memtime();
  };
  if (dbg != nullptr && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[(size_t)blockIdx.x * 32 + 31] = ((unsigned long long)xcc << 32) | hw;
  }
  stamp(0);
  stamp(1);

  const int nbins = PH * PW;
  const int ns = S * S;  // 1 or 4
  const int nchunks = (C + kCK - 1) / kCK;
  const int chunk0 = grp_id * cpg;
  const int nch = min(cpg, nchunks - chunk0);
  if (nch <= 0) return;

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

  for (int i = tid; i < TS::ZERO_N * 8; i += kTNT)
    *reinterpret_cast<v4f*>(smem + TS::ZERO_PX * kPixB + i * 16) = v4f{0.f, 0.f, 0.f, 0.f};

  // Chunk order is rotated by XCD.  A pixel is 1 KiB (C = 256) and a chunk is one 128-byte line of it: if every
  // workgroup walked the chunks in the same order, the whole chip would at any moment read the SAME 128 bytes of
  // every 1 KiB.  Inside one XCD the workgroups stay roughly in step; across XCDs all lines of a pixel are in flight.
  const int rot = (int)(blockIdx.x & 7) % nch;
  auto chunk_at = [&](int cc) -> int {
    const int c = cc + rot;
    return chunk0 + (c >= nch ? c - nch : c);
  };
  auto lds4 = [&](int byte_off) -> v4f { return *reinterpret_cast<const v4f*>(smem + byte_off); };
  const float inv_count = 1.f / (float)ns;  // 1 or 0.25: exact, and equal to the reference's `/ count`

  // ---- one tile segment: bins [a, b) of tile t's list
  auto segment = [&](int t, int a, int b) {
    const PlanHdr th = hdr[t];
    b = min(b, th.count);   // (the even padding of the prefix is not a bin)
    if (a >= b) return;
    int w2 = t;
    const int tx = w2 % tilesX;
    w2 /= tilesX;
    const int ty = w2 % tilesY;
    const int n = w2 / tilesY;
    const int ox = tx * TS::TW - TS::HLO, oy = ty * TS::TH - TS::HLO;
    // raw buffer over image n (out-of-range offsets read 0)
    const float* img = feat + (size_t)n * H * W * C;
    const u64 img_bits = (u64)img;
    const unsigned img_lo = __builtin_amdgcn_readfirstlane((unsigned)img_bits);
    const unsigned img_hi = __builtin_amdgcn_readfirstlane((unsigned)(img_bits >> 32));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>((const void*)(((u64)img_hi << 32) | img_lo)), 0,
        __builtin_amdgcn_readfirstlane((int)((size_t)H * W * C * 4)), 0x00020000);
    // window slot i = 16 bytes: pixel i >> 3 (row-major over WH x WWP), sub-slot i & 7
    int voff[NPF];
#pragma unroll
    for (int k = 0; k < NPF; k++) {
      const int i = tid + k * kTNT;
      const int px = i >> 3, sb = i & 7;
      const int wy = px / TS::WWP, wx = px - wy * TS::WWP;
      const int gy = oy + wy, gx = ox + wx;
      const bool ok = i < NSLOT && wx < TS::WW && gy >= 0 && gy < H && gx >= 0 && gx < W;
      voff[k] = ok ? ((gy * W + gx) * C + sb * 4) * 4 : 0x7FFFFFF0;
    }
    v4f pf[NPF];
    auto prefetch = [&](int chunk) {
      const int soff = __builtin_amdgcn_readfirstlane(chunk * kCK * 4);
#pragma unroll
      for (int k = 0; k < NPF; k++)
        pf[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[k], soff, 0));
    };
    auto store_window = [&]() {
#pragma unroll
      for (int k = 0; k < NPF; k++) {
        const int i = tid + k * kTNT;
        if (i < NSLOT) *reinterpret_cast<v4f*>(smem + i * 16) = pf[k];
      }
    };
    // all taps of the bins of this pass from the LDS window (table-driven), then the slow bins from global
    auto compute = [&](int nb, int chunk) {
      const int cbase = chunk * kCK;
      const bool ch_ok = cbase + sub * 4 < C;
      bool any_slow = false;
      for (int base = wave * 8; base < nb; base += kTNW * 8) {
        const int e = base + grp;
        if (e >= nb) continue;
        const int rw = s_row[e];
        if (rw < 0) continue;           // pad entry between two candidate batches of the plan
        if (rw & kSlowBit) {
          any_slow = true;
          continue;
        }
        v4f acc = {0.f, 0.f, 0.f, 0.f};
        // two samples at a time (10 dwords = 3 table reads, the middle one shared by both halves): 8 tap reads in
        // flight; the register budget is 128 for 2 workgroups / CU
#pragma nounroll
        for (int h = 0; h < 2; h++) {
          float tq[12];
#pragma unroll
          for (int j = 0; j < 3; j++) {
            const v4f q = s_tab[e * (kTabDw / 4) + 2 * h + j];
            tq[4 * j] = q.x; tq[4 * j + 1] = q.y; tq[4 * j + 2] = q.z; tq[4 * j + 3] = q.w;
          }
          // half 0: dwords 0..9 of the bin = tq[0..9]; half 1: dwords 10..19 = tq[2..11]
          float s0[5], s1[5];
#pragma unroll
          for (int j = 0; j < 5; j++) {
            s0[j] = h ? tq[2 + j] : tq[j];
            s1[j] = h ? tq[7 + j] : tq[5 + j];
          }
          v4f tp[2][4];
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int pk = __float_as_int(u ? s1[0] : s0[0]);
            const int ra = (pk & 0xFFFF) + sub * 16;
            const int rb = ra + (pk >> 16);
            tp[u][0] = lds4(ra);
            tp[u][1] = lds4(rb);
            tp[u][2] = lds4(ra + TS::WWP * kPixB);
            tp[u][3] = lds4(rb + TS::WWP * kPixB);
          }
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const float* w = u ? &s1[1] : &s0[1];
            if (EXACT) {
              // reference order (no left / right swap in the plan): ((w1*lt + w2*rt) + w3*lb) + w4*rb, then +=
              acc += ((w[0] * tp[u][0] + w[1] * tp[u][1]) + w[2] * tp[u][2]) + w[3] * tp[u][3];
            } else {
#pragma unroll
              for (int q = 0; q < 4; q++)
#pragma unroll
                for (int k = 0; k < 4; k++) acc[k] = __builtin_fmaf(w[q], tp[u][q][k], acc[k]);
            }
          }
        }
        if (EXACT) acc *= inv_count;   // (the fma path has 1 / count folded into the weights)
        if (ch_ok) __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + (size_t)rw * C + cbase + sub * 4));
      }
      if (!__any(any_slow)) return;
      for (int base = wave * 8; base < nb; base += kTNW * 8) {
        const int e = base + grp;
        if (e >= nb) continue;
        const int rw = s_row[e];
        if (rw < 0 || !(rw & kSlowBit)) continue;
        const int row = rw & (kSlowBit - 1);
        const int r = row / nbins, bin = row - r * nbins;
        const RoiGeom g = roi_geom<VARIANT, true>(rois + (size_t)r * ROI_COLS, spatial_scale, S, PH, PW, 1, false);
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
        if (ch_ok) __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + (size_t)row * C + cbase + sub * 4));
      }
    };

    prefetch(chunk_at(0));
    for (int p0 = a; p0 < b; p0 += kPart) {   // a is even
      const int nb = min(kPart, b - p0);
      const int g0 = th.offset + p0;
      {
        const v4f* src = reinterpret_cast<const v4f*>(tab + (size_t)g0 * kTabDw);   // g0 even -> 16-byte aligned
        for (int i = tid; i < nb * (kTabDw / 4); i += kTNT) s_tab[i] = src[i];
        for (int i = tid; i < nb; i += kTNT) s_row[i] = rows[g0 + i];
      }
      for (int cc = 0; cc < nch; cc++) {
        store_window();
        __syncthreads();
        stamp(5 + 2 * cc);
        const bool last = cc + 1 == nch && p0 + kPart >= b;
        if (!last) prefetch(chunk_at(cc + 1 == nch ? 0 : cc + 1));
        compute(nb, chunk_at(cc));
        stamp(6 + 2 * cc);
        __syncthreads();
      }
    }
  };

  int nseg = 0, nbin_done = 0;
  for (int p0 = part * kPart; p0 < th.count; p0 += nparts * kPart) {   // one pass for all but extremely dense tiles
    segment(tile_id, p0, min(th.count, p0 + kPart));
    nseg++;
    nbin_done += min(th.count, p0 + kPart) - p0;
  }
  if (dbg != nullptr && tid == 0) dbg[(size_t)blockIdx.x * 32 + 30] = ((unsigned long long)nseg << 32) | (unsigned)nbin_done;
  stamp(29);
}

unsigned long long* g_tile_dbg = nullptr;

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

typedef TileShape<16, 8, 4, 5> Shape0;   // halo sized for bins up to 9.1 px (RoIs up to 64 px at 7x7)

template <int VARIANT, class TS, bool EXACT>
int launch_tile(const float* feat, const float* rois, float* out, int N, int C, int H, int W, int R, int PH, int PW,
                float scale, int S, void* ws, hipStream_t st) {
  auto pool = roi_align_tile_pool_kernel<VARIANT, TS, EXACT>;
  auto plan = roi_tile_plan_kernel<VARIANT, TS>;
  static bool attr_set[64] = {};   // per device: > 64 KiB of dynamic LDS has to be opted into once
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return JDET_E_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)pool, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       PoolLds<TS>::BYTES);
    if (e != hipSuccess) return (int)e;
    e = hipFuncSetAttribute((const void*)plan, hipFuncAttributeMaxDynamicSharedMemorySize, kRoiBlock * 24);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const int tilesX = jdet_cdiv(W, TS::TW), tilesY = jdet_cdiv(H, TS::TH);
  const long ntiles = (long)N * tilesX * tilesY;
  const int nbins = PH * PW;
  if (ntiles > kMaxTiles) return JDET_E_UNSUPPORTED;
  PlanWs p = plan_carve(ws, ntiles, plan_bins_bound(ntiles, R, nbins));
  const size_t plan_lds = (size_t)(R < kRoiBlock ? R : kRoiBlock) * 24;
  hipLaunchKernelGGL(plan, dim3((unsigned)ntiles), dim3(kPNT), plan_lds, st, rois, H, W, R, PH, PW, scale, S, tilesX,
                     tilesY, p.hdr, p.cursor, p.rows, p.tab, EXACT ? 1 : 0,
                     g_tile_dbg ? g_tile_dbg + (size_t)4096 * 32 : nullptr);
  const int nchunks = jdet_cdiv(C, kCK);
  static const int cpg_env = env_int("JDET_ROI_TILE_CPG", 0);
  static const int parts_env = env_int("JDET_ROI_TILE_PARTS", 0);
  // channel chunks per work item (default: all -- the table copy is paid once per item)
  int cpg = cpg_env > 0 ? cpg_env : nchunks;
  cpg = cpg < 1 ? 1 : (cpg > nchunks ? nchunks : cpg);
  const int ngroups = jdet_cdiv(nchunks, cpg);
  // parts per tile in the grid (a part that its tile does not need exits at once; denser tiles loop)
  const int nparts = parts_env > 0 ? parts_env : 4;
  const long per_part = ((ntiles * ngroups + 7) / 8) * 8;
  if (per_part * nparts > (1L << 30)) return JDET_E_UNSUPPORTED;
  static const int dbg_print = env_int("JDET_ROI_TILE_DEBUG", 0);
  if (dbg_print) {
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pool, kTNT, PoolLds<TS>::BYTES) != hipSuccess) nb = -1;
    fprintf(stderr, "[jdet tile] lds=%d B grid=%ld cpg=%d ngroups=%d nparts=%d occupancy=%d blocks/CU\n",
            PoolLds<TS>::BYTES, per_part * nparts, cpg, ngroups, nparts, nb);
  }
  hipLaunchKernelGGL(pool, dim3((unsigned)(per_part * nparts)), dim3(kTNT), PoolLds<TS>::BYTES, st, feat, rois, out, C,
                     H, W, PH, PW, scale, S, tilesX, tilesY, (int)ntiles, cpg, ngroups, nparts, p.hdr, p.rows, p.tab,
                     p.cursor, g_tile_dbg);
  return jdet_launch_status();
}

template <int VARIANT>
int dispatch_tile(int exact, const float* feat, const float* rois, float* out, int N, int C, int H, int W, int R,
                  int PH, int PW, float scale, int S, void* ws, hipStream_t st) {
  if (exact) return launch_tile<VARIANT, Shape0, true>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, ws, st);
  return launch_tile<VARIANT, Shape0, false>(feat, rois, out, N, C, H, W, R, PH, PW, scale, S, ws, st);
}

}  // namespace

// Profiling hook (not part of the operator surface): when set, every workgroup of the pool kernel writes 32
// s_memtime stamps (phase boundaries) + its hardware id into buf[blockIdx * 32 ...].  NULL switches it off.
JDET_API int jdet_debug_roi_tile_timeline(void* buf) {
  g_tile_dbg = (unsigned long long*)buf;
  return JDET_OK;
}

JDET_API int jdet_roi_align_forward_cl_supported(int variant, int C, int H, int W, int PH, int PW, int sample_num) {
  if (variant != JDET_ROI_ROTATED && variant != JDET_ROI_ROTATED_V1 && variant != JDET_ROI_HBB_V0 &&
      variant != JDET_ROI_HBB_V1)
    return 0;
  if (C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0) return 0;
  if (C % 4 != 0 || (sample_num != 1 && sample_num != 2) || PH * PW > 64) return 0;
  if ((size_t)H * W * C * 4 >= (1ull << 31)) return 0;
  return 1;   // (plus: N * tiles <= 8192, reported by a zero workspace size)
}

JDET_API size_t jdet_roi_align_forward_cl_workspace(int N, int H, int W, int R, int PH, int PW) {
  if (N <= 0 || H <= 0 || W <= 0 || R <= 0 || PH <= 0 || PW <= 0) return 0;
  const long ntiles = (long)N * jdet_cdiv(W, Shape0::TW) * jdet_cdiv(H, Shape0::TH);
  if (ntiles > kMaxTiles) return 0;
  return plan_carve(nullptr, ntiles, plan_bins_bound(ntiles, R, PH * PW)).bytes;
}

JDET_API int jdet_roi_align_forward_cl(int variant, const float* feat, int N, int C, int H, int W, const float* rois,
                                       int R, int PH, int PW, float spatial_scale, int sample_num, int exact_order,
                                       float* out, void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  if (N < 0 || C <= 0 || H <= 0 || W <= 0 || R < 0 || PH <= 0 || PW <= 0) return JDET_E_BADARG;
  if (variant < 0 || variant > 4) return JDET_E_BADARG;
  if (!jdet_roi_align_forward_cl_supported(variant, C, H, W, PH, PW, sample_num)) return JDET_E_UNSUPPORTED;
  if ((long)R * PH * PW >= (1L << 29)) return JDET_E_UNSUPPORTED;
  if (R == 0 || N == 0) return JDET_OK;
  if (!feat || !rois || !out) return JDET_E_BADARG;
  if (!workspace || workspace_bytes < jdet_roi_align_forward_cl_workspace(N, H, W, R, PH, PW)) return JDET_E_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return dispatch_tile<JDET_ROI_ROTATED>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, workspace, st);
    case JDET_ROI_ROTATED_V1:
      return dispatch_tile<JDET_ROI_ROTATED_V1>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, workspace, st);
    case JDET_ROI_HBB_V0:
      return dispatch_tile<JDET_ROI_HBB_V0>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, workspace, st);
    default:
      return dispatch_tile<JDET_ROI_HBB_V1>(exact_order, feat, rois, out, N, C, H, W, R, PH, PW, spatial_scale, sample_num, workspace, st);
  }
}
