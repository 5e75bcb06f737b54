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
  static constexpr int ZERO_PX = (NPX + 1) & ~1;  // two all-zero pixels (even, odd) for invalid samples
  static constexpr int WIN_BYTES = (ZERO_PX + 2) * kPixB;
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
  v4f* tab;         // [R * nbins * 4]
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
  p.tab = (v4f*)(b + off);      off += a256(sizeof(v4f) * 4 * nbins_total);
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
constexpr int kRoiBlock = 1024;    // RoI rows staged per scan block (24 KiB)
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
                                                            v4f* __restrict__ tab) {
  constexpr bool kHbb = VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1;
  constexpr int ROI_COLS = kHbb ? 5 : 6;
  __shared__ __attribute__((aligned(16))) float s_roi[kRoiBlock * 6];
  __shared__ CandRec s_cand[kCandCap];
  __shared__ u64 s_mask[kCandCap];
  __shared__ int s_off[kCandCap + 4];
  __shared__ int s_cand_r[kCandCap];
  __shared__ int s_wcnt[2 * kPNW];
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
    int round = 0;
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
#pragma nounroll
      for (int k = 0; k < kRoiBlock / kPNT; k++) {
        if (r0 + k * kPNT >= R || next_begin < 0) break;
        const int rl = k * kPNT + tid;
        const int r = r0 + rl;
        const bool cand = r < R && cand_test(rl);
        const u64 bal = __ballot(cand);
        if (lane == 0) s_wcnt[(round & 1) * kPNW + wave] = __popcll(bal);
        __syncthreads();
        int before = 0, tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < kPNW; w2++) {
          const int c = s_wcnt[(round & 1) * kPNW + w2];
          tot += c;
          before += w2 < wave ? c : 0;
        }
        const int rank = ncand + before + __popcll(bal & ((1ull << lane) - 1ull));
        if (cand) {
          if (rank < kCandCap) s_cand_r[rank] = r;
          else if (rank == kCandCap) s_misc[0] = r;   // first RoI that did not fit: the next batch starts here
        }
        round++;
        if (ncand + tot > kCandCap) {
          ncand = kCandCap;
          next_begin = -1;
        } else {
          ncand += tot;
        }
      }
      __syncthreads();   // the next block's staging overwrites the rows read above
    }
    __syncthreads();
    if (next_begin < 0) next_begin = s_misc[0];
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
  };

  // rows + table entries of the current batch at plan position `base`
  auto write_batch = [&](int base) {
    for (int ci = wave; ci < ncand; ci += kPNW) {
      const u64 mk = s_mask[ci];
      if ((mk >> lane) & 1ull) {
        const int e = s_off[ci] + __popcll(mk & ((1ull << lane) - 1ull));
        const int rl = e & 1;   // parity of the bin's position in the TILE list (batches and parts start even)
        const RoiGeom g = geom_of(s_cand[ci]);
        const int ph = lane / PW, pw = lane - ph * PW;
        int slow = 0;
        v4f ent[4];
#pragma unroll
        for (int s = 0; s < 4; s++) {
          int packed = (TS::ZERO_PX + rl) * kPixB;
          float ly = 0.f, wA = 0.f, wB = 0.f;
          if (s < ns) {
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
                slow = 1;   // a tap outside the window: the whole bin reads from global
              }
            }
          }
          ent[s] = v4f{__int_as_float(packed), ly, wA, wB};
        }
        v4f* dst = tab + (size_t)(base + e) * 4;
#pragma unroll
        for (int s = 0; s < 4; s++) dst[s] = ent[s];
        rows[base + e] = (s_cand[ci].r * nbins + lane) | (slow ? kSlowBit : 0);
      }
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
    const int base = alloc(total_bins);
    write_batch(base);
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
constexpr int kTNT = 512;   // threads per pool workgroup (8 waves), 2 workgroups per CU
constexpr int kTNW = kTNT / 64;
constexpr int kPart = 192;  // bins per work item (even: the table's pair parity is position parity)

template <class TS>
struct PoolLds {
  static constexpr int NSLOT = TS::NPX * 8;   // 16-byte slots of the window
  static constexpr int NPF = (NSLOT + kTNT - 1) / kTNT;
  static constexpr int BYTES = TS::WIN_BYTES + kPart * 64 + kPart * 4;
};

template <int VARIANT, class TS, bool EXACT>
__global__ __launch_bounds__(kTNT, 4) void roi_align_tile_pool_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int C, int H, int W,
    int PH, int PW, float spatial_scale, int S, int tilesX, int tilesY, int cpg, int ngroups, int nparts, int total,
    const PlanHdr* __restrict__ hdr, const int* __restrict__ rows, const v4f* __restrict__ tab,
    unsigned long long* __restrict__ dbg) {
  constexpr bool kHbb = VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1;
  constexpr int ROI_COLS = kHbb ? 5 : 6;
  constexpr int NPF = PoolLds<TS>::NPF, NSLOT = PoolLds<TS>::NSLOT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  v4f* s_tab = reinterpret_cast<v4f*>(smem + TS::WIN_BYTES);
  int* s_row = reinterpret_cast<int*>(s_tab + kPart * 4);

  // workgroup b runs on XCD b % 8 (observed; only speed depends on it): give every XCD one contiguous run of work
  // items so that neighbouring windows (and the parts / channel groups of one tile) share that XCD's L2.
  const int per_xcd = (total + 7) >> 3;
  int wk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (wk >= total) return;
  const int part = wk % nparts;
  wk /= nparts;
  const int grp_id = wk % ngroups;
  wk /= ngroups;
  const int tile_id = wk;
  const PlanHdr th = hdr[tile_id];
  if (part * kPart >= th.count) return;   // nothing for this part (most tiles need one or two parts)
  const int tx = wk % tilesX;
  wk /= tilesX;
  const int ty = wk % tilesY;
  const int n = wk / tilesY;

  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int ox = tx * TS::TW - TS::HLO, oy = ty * TS::TH - TS::HLO;
  auto stamp = [&](int slot) {   // profiling hook (jdet_debug_roi_tile_timeline)
    if (dbg != nullptr && tid == 0 && slot < 32) dbg[(size_t)blockIdx.x * 32 + slot] = __builtin_amdgcn_s_memtime();
  };
  if (dbg != nullptr && tid == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    dbg[(size_t)blockIdx.x * 32 + 31] = ((unsigned long long)xcc << 32) | hw;
    dbg[(size_t)blockIdx.x * 32 + 30] = (unsigned)th.count;
  }
  stamp(0);
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
  auto lds4 = [&](int byte_off) -> v4f { return *reinterpret_cast<const v4f*>(smem + byte_off); };

  // Chunk order is rotated by XCD.  A pixel is 1 KiB (C = 256) and a chunk is one 128-byte line of it: if every
  // workgroup walked the chunks in the same order, the whole chip would at any moment read the SAME 128 bytes of
  // every 1 KiB.  Inside one XCD the workgroups stay roughly in step (their halos overlap in that XCD's L2);
  // across XCDs all lines of a pixel are in flight.
  const int rot = (int)(blockIdx.x & 7) % nch;
  auto chunk_at = [&](int cc) -> int {
    const int c = cc + rot;
    return chunk0 + (c >= nch ? c - nch : c);
  };
  if (tid < 16) *reinterpret_cast<v4f*>(smem + TS::ZERO_PX * kPixB + tid * 16) = v4f{0.f, 0.f, 0.f, 0.f};
  prefetch(chunk_at(0));

  const float inv_count = 1.f / (float)ns;  // 1 or 0.25: exact, and equal to the reference's `/ count`

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

  // passes: parts [part, part + nparts, ...) of the tile's bin list (one pass for all but extremely dense tiles)
  for (int p0 = part * kPart; p0 < th.count; p0 += nparts * kPart) {
    const int nb = min(kPart, th.count - p0);
    const int g0 = th.offset + p0;
    for (int i = tid; i < nb * 4; i += kTNT) s_tab[i] = tab[(size_t)g0 * 4 + i];
    for (int i = tid; i < nb; i += kTNT) s_row[i] = rows[g0 + i];
    stamp(4);
    for (int cc = 0; cc < nch; cc++) {
      store_window();
      __syncthreads();
      stamp(5 + 2 * cc);
      const bool last = cc + 1 == nch && p0 + nparts * kPart >= th.count;
      if (!last) prefetch(chunk_at(cc + 1 == nch ? 0 : cc + 1));
      compute(nb, chunk_at(cc));
      stamp(6 + 2 * cc);
      __syncthreads();
    }
  }
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
  static bool attr_set[64] = {};   // per device: > 64 KiB of dynamic LDS has to be opted into once
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return JDET_E_UNSUPPORTED;
  if (!attr_set[dev]) {
    hipError_t e = hipFuncSetAttribute((const void*)pool, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       PoolLds<TS>::BYTES);
    if (e != hipSuccess) return (int)e;
    attr_set[dev] = true;
  }
  const int tilesX = jdet_cdiv(W, TS::TW), tilesY = jdet_cdiv(H, TS::TH);
  const long ntiles = (long)N * tilesX * tilesY;
  const int nbins = PH * PW;
  if (ntiles > (1L << 24)) return JDET_E_UNSUPPORTED;
  PlanWs p = plan_carve(ws, ntiles, plan_bins_bound(ntiles, R, nbins));
  int he = jdet_zero_async(p.cursor, 256, st);
  if (he) return he;
  hipLaunchKernelGGL((roi_tile_plan_kernel<VARIANT, TS>), dim3((unsigned)ntiles), dim3(kPNT), 0, st, rois, H, W, R,
                     PH, PW, scale, S, tilesX, tilesY, p.hdr, p.cursor, p.rows, p.tab);
  const int nchunks = jdet_cdiv(C, kCK);
  static const int cpg_env = env_int("JDET_ROI_TILE_CPG", 0);
  static const int parts_env = env_int("JDET_ROI_TILE_PARTS", 0);
  // channel chunks per work item: more chunks amortise the table copy, fewer give more, smaller work items
  int cpg = cpg_env > 0 ? cpg_env : 8;
  cpg = cpg < 1 ? 1 : (cpg > nchunks ? nchunks : cpg);
  const int ngroups = jdet_cdiv(nchunks, cpg);
  // parts per tile: a work item pools <= kPart bins; tiles with more bins than nparts * kPart loop
  const int nparts = parts_env > 0 ? parts_env : 3;
  const long total = ntiles * ngroups * nparts;
  if (total > (1L << 30)) return JDET_E_UNSUPPORTED;
  const int per_xcd = (int)((total + 7) / 8);
  static const int dbg_print = env_int("JDET_ROI_TILE_DEBUG", 0);
  if (dbg_print) {
    int nb = -1;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)pool, kTNT, PoolLds<TS>::BYTES) != hipSuccess) nb = -1;
    fprintf(stderr, "[jdet tile] lds=%d B grid=%d cpg=%d ngroups=%d nparts=%d occupancy=%d blocks/CU\n",
            PoolLds<TS>::BYTES, per_xcd * 8, cpg, ngroups, nparts, nb);
  }
  hipLaunchKernelGGL(pool, dim3(per_xcd * 8), dim3(kTNT), PoolLds<TS>::BYTES, st, feat, rois, out, C, H, W, PH, PW,
                     scale, S, tilesX, tilesY, cpg, ngroups, nparts, (int)total, p.hdr, p.rows, p.tab, g_tile_dbg);
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
  return 1;
}

JDET_API size_t jdet_roi_align_forward_cl_workspace(int N, int H, int W, int R, int PH, int PW) {
  if (N <= 0 || H <= 0 || W <= 0 || R <= 0 || PH <= 0 || PW <= 0) return 0;
  const long ntiles = (long)N * jdet_cdiv(W, Shape0::TW) * jdet_cdiv(H, Shape0::TH);
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
