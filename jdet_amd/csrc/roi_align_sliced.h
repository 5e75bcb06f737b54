// Channel-sliced RoIAlign forward for gfx950 (round 4): the product forward of the sample_num == 2 dialects.
//
// Reference semantics: python/jdet/ops/roi_align_rotated.py:L61-127 (and the _v1 / riroi / horizontal twins);
// arithmetic = the merged-tap mode of roi_align.hip (same geometry functions, same tap merge, same fma chain).
//
// Why sliced.  MI355X has 8 XCDs with a private 4 MiB L2 each, and workgroup b runs on XCD b % 8.  The
// RoI-stationary kernel of rounds 1-3 gave every workgroup all 256 channels of one RoI and cut the map eight ways by
// RoI centre: big RoIs reach across the cuts (1.44 x the map is compulsory), and a pixel row is 1 KiB, so the rows a
// few hundred RoIs in flight touch exceed an L2 (measured: 171 MB read beyond the L2 for a 67 MB map).  Here the cut
// is by CHANNEL: XCD x owns channels [32 x, 32 x + 32) of EVERY RoI, and all XCDs walk the RoIs in the same Morton
// order.  A pixel is then one 128-byte line in exactly one L2, every map byte is compulsory in one L2 only, and half
// of a 256 x 256 map slice (8.4 MB) fits the 4 MiB.
//
// Work mapping.  Items = (RoI in processing order, bin), flat.  A wave owns 16 consecutive items:
//   geometry   lane quad = one item (lane = one of its 4 samples): position, weights, tap merge inside the quad,
//              merged list compacted into the wave's private LDS block (16 lists x <= 16 (offset, weight) pairs);
//   taps       two rounds of 8 items: a group of 8 lanes = one item x 32 channels (dwordx4 per lane = one 128-byte
//              line per tap per group), every group walks ITS item's list (per-group voffset), BATCH loads in flight;
//   store      one 128-byte line per group into the channels-last row (r, bin, 32 slice .. +31).
// Per-RoI constants (double-precision trig, bin sizes, RiRoI orientation constants) come from 64-byte records written
// once per launch by roi_prep_kernel -- the Morton counting sort of the schedule, now emitting records in processing
// order with masked RoIs (batch < 0) dropped -- and read through the scalar cache (a wave spans <= 2 RoIs).
// (Included by roi_align.hip inside its unnamed namespace, after ri_dispatch<>.)
#pragma once
#include "roi_geom.h"

namespace jdet_roi_sliced {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));

struct RoiRec {      // one per RoI, in processing order; 64 bytes
  int r;             // index into rois / out
  int batch;
  float center_w, center_h, start_w, start_h, bin_h, bin_w, cosT, sinT;
  int ind;           // RiRoIAlign orientation constants (riroi_align.py:L105-113)
  float l_var, r_var;
  int pad[3];
};
static_assert(sizeof(RoiRec) == 64, "one scalar-cache line per record");

constexpr int kHdrBytes = 256;          // workspace: [0] = number of unmasked RoIs, then the records
constexpr int kSliceC = 32;             // channels per slice = 8 lanes x dwordx4 = one 128-byte line per pixel
constexpr int kItemsPerWave = 16;
constexpr int kListStride = 34;         // dwords per item list: [n, pad, 16 x (offset, weight)]; 34 b mod 64 distinct
                                        // even banks for the 8 groups of a round: conflict-free ds_read_b64

// ---------------------------------------------------------------------------------------------------------------
// Schedule + per-RoI records: one workgroup.  Counting sort on the Morton code of the RoI centre (32 x 32 cells per
// image), masked RoIs last; record p of the result describes the p-th RoI to process.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kPrepThreads = 1024;
constexpr int kCellsLog2 = 5;
constexpr int kCells = 1 << (2 * kCellsLog2);
constexpr int kMaxImages = 8;

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

template <int VARIANT>
__global__ __launch_bounds__(kPrepThreads) void roi_prep_kernel(const float* __restrict__ rois, int R,
                                                               float spatial_scale, int N, int H, int W, int PH,
                                                               int PW, int nO, int* __restrict__ hdr,
                                                               RoiRec* __restrict__ recs) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  __shared__ int s_bins[kMaxImages * kCells + 1];
  __shared__ int s_scan[kPrepThreads / 64];
  const int nimg = min(max(N, 1), kMaxImages);
  const int nbins = nimg * kCells + 1;           // last bin: masked RoIs
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nbins; i += kPrepThreads) s_bins[i] = 0;
  __syncthreads();
  auto key_of = [&](int r) -> int {
    const float* p = rois + (size_t)r * ROI_COLS;
    const int b = (int)p[0];
    if (b < 0) return nbins - 1;
    float cx, cy;
    if (ROI_COLS == 5) {
      cx = 0.5f * (p[1] + p[3]) * spatial_scale;
      cy = 0.5f * (p[2] + p[4]) * spatial_scale;
    } else {
      cx = p[1] * spatial_scale;
      cy = p[2] * spatial_scale;
    }
    const float fx = fminf(fmaxf(cx / (float)W, 0.f), 0.999999f);
    const float fy = fminf(fmaxf(cy / (float)H, 0.f), 0.999999f);
    const unsigned ix = (unsigned)(fx * (1 << kCellsLog2));
    const unsigned iy = (unsigned)(fy * (1 << kCellsLog2));
    return min(b, nimg - 1) * kCells + (int)morton2(ix, iy);
  };
  constexpr int kKeep = 4;                       // keys of the first 4096 RoIs stay in registers
  int mykey[kKeep];
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int r = threadIdx.x + i * kPrepThreads;
    mykey[i] = r < R ? key_of(r) : 0;
    if (r < R) atomicAdd(&s_bins[mykey[i]], 1);
  }
  for (int r = threadIdx.x + kKeep * kPrepThreads; r < R; r += kPrepThreads) atomicAdd(&s_bins[key_of(r)], 1);
  __syncthreads();
  const int n_masked = s_bins[nbins - 1];
  // exclusive scan of the bins: contiguous slice per thread, wave scan, one LDS hop across the 16 waves
  const int per = (nbins + kPrepThreads - 1) / kPrepThreads;
  const int lo = threadIdx.x * per, hi = min(lo + per, nbins);
  int sum = 0;
  for (int i = lo; i < hi; i++) sum += s_bins[i];
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off, 64);
    if (lane >= off) incl += v;
  }
  __syncthreads();                               // n_masked has been read by everybody
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; w++) run += s_scan[w];
  for (int i = lo; i < hi; i++) {
    const int c = s_bins[i];
    s_bins[i] = run;
    run += c;
  }
  __syncthreads();
  if (threadIdx.x == 0) hdr[0] = R - n_masked;
  auto emit = [&](int r, int key) {
    const int pos = atomicAdd(&s_bins[key], 1);
    const float* roi = rois + (size_t)r * ROI_COLS;
    RoiGeom g = roi_geom<VARIANT>(roi, spatial_scale, 2, PH, PW, max(nO, 1), false);
    RoiRec rec;
    rec.r = r;
    rec.batch = g.batch;
    if (g.batch >= N) {                          // no such image: every sample out of range -> zeros (not a fault)
      rec.batch = 0;
      g.center_w = g.center_h = g.start_w = g.start_h = -1e30f;
    }
    rec.center_w = g.center_w; rec.center_h = g.center_h;
    rec.start_w = g.start_w;   rec.start_h = g.start_h;
    rec.bin_h = g.bin_h;       rec.bin_w = g.bin_w;
    rec.cosT = g.cosT;         rec.sinT = g.sinT;
    rec.ind = 0; rec.l_var = 0.f; rec.r_var = 1.f;
    if (nO > 1 && ROI_COLS == 6) ri_params(roi[5], nO, rec.ind, rec.l_var, rec.r_var);
    rec.pad[0] = rec.pad[1] = rec.pad[2] = 0;
    int4* dst = reinterpret_cast<int4*>(recs + pos);
    const int4* src = reinterpret_cast<const int4*>(&rec);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  };
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int r = threadIdx.x + i * kPrepThreads;
    if (r < R) emit(r, mykey[i]);
  }
  for (int r = threadIdx.x + kKeep * kPrepThreads; r < R; r += kPrepThreads) emit(r, key_of(r));
}

// ---------------------------------------------------------------------------------------------------------------
// Forward.  grid.x = nslices * ceil(n_items / 64) workgroups of 4 waves; slice = blockIdx.x % nslices.
//   NO     0: plain RoIAlign; 4 / 8: RiRoIAlign orientation mix on the finished bin (VARIANT = rotated geometry)
//   BATCH  tap loads issued back to back per group before the first use (4 / 8 / 16)
//   STORE  0 non-temporal, 1 plain, 2 write-through (sc1)
// ---------------------------------------------------------------------------------------------------------------
template <int VARIANT, int NO, int BATCH, int STORE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96))) void roi_align_fwd_sliced_kernel(
    const float* __restrict__ feat, const int* __restrict__ hdr, const RoiRec* __restrict__ recs,
    float* __restrict__ out, int R, int N, int C, int H, int W, int PH, int PW, int nslices) {
  __shared__ __attribute__((aligned(16))) int s_lists[4 * kItemsPerWave * kListStride];
  const int nbins = PH * PW;                                  // >= 16 on this path: a wave spans <= 2 RoIs
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane(blockIdx.x % nslices), blk = __builtin_amdgcn_readfirstlane(blockIdx.x / nslices);
  const int n_valid = __builtin_amdgcn_readfirstlane(hdr[0]);
  const int n_items = n_valid * nbins;
  const int item0 = (blk * 4 + wave) * kItemsPerWave;
  if (item0 >= n_items) return;                               // (no workgroup-wide barrier below)
  const int sA = item0 / nbins;                               // wave-uniform
  const int sB = min(sA + 1, n_valid - 1);
  const int splitB = (sA + 1) * nbins;                        // items >= splitB belong to record sB
  const RoiRec A = recs[sA], B = recs[sB];                    // uniform addresses -> scalar loads

  // ---- geometry: quad of lanes = one item ----
  const int q = lane & 3, qbase = lane & ~3, kb = lane >> 2;
  const int item = item0 + kb;
  const bool item_ok = item < n_items;
  const bool isB = item >= splitB;
  RoiGeom g;
  g.batch = isB ? B.batch : A.batch;
  g.center_w = isB ? B.center_w : A.center_w;  g.center_h = isB ? B.center_h : A.center_h;
  g.start_w = isB ? B.start_w : A.start_w;     g.start_h = isB ? B.start_h : A.start_h;
  g.bin_h = isB ? B.bin_h : A.bin_h;           g.bin_w = isB ? B.bin_w : A.bin_w;
  g.cosT = isB ? B.cosT : A.cosT;              g.sinT = isB ? B.sinT : A.sinT;
  g.grid_h = g.grid_w = 2;
  g.count = 4.f;
  g.l_var = 0.f; g.r_var = 1.f; g.ind = 0;
  const int bin = item_ok ? item - (isB ? splitB : splitB - nbins) : 0;
  Sample s = make_sample<VARIANT>(g, bin / PW, bin % PW, q >> 1, q & 1, H, W);
  if (!item_ok) s.valid = 0;
  const unsigned pix_bytes = (unsigned)C * 4u;
  const unsigned img_off = (unsigned)g.batch * (unsigned)(H * W);       // pixels before this image
  // byte offsets of the 4 taps; an invalid sample carries a sentinel no tap can equal (offsets are multiples of 4)
  const unsigned kNone = 0xffffffffu;
  const unsigned o[4] = {s.valid ? (img_off + s.o1) * pix_bytes : kNone, s.valid ? (img_off + s.o2) * pix_bytes : kNone,
                         s.valid ? (img_off + s.o3) * pix_bytes : kNone, s.valid ? (img_off + s.o4) * pix_bytes : kNone};
  const float w[4] = {s.w1, s.w2, s.w3, s.w4};
  // Tap merge inside the quad (= the 16 taps of the item): the first occurrence of a pixel collects the weights of the
  // others, the rest are dropped.  Kept in VGPR integers on purpose: as bool arrays the 48 compare results become 48
  // live SGPR-pair masks and spill (measured: 100 v_writelane, waterfall loops around the buffer loads).
  float tw[4] = {w[0], w[1], w[2], w[3]};
  unsigned drop = 0;           // bit k: tap k is not the first occurrence of its pixel
#pragma unroll
  for (int k = 1; k < 4; k++)
#pragma unroll
    for (int j = 0; j < k; j++) {   // x_high == x_low / y_high == y_low at the map border
      const bool dup = o[j] == o[k];
      tw[j] += dup ? w[k] : 0.f;
      drop |= dup ? (1u << k) : 0u;
    }
#pragma unroll
  for (int d = 1; d < 4; d++) {
    const int src = qbase | ((q + d) & 3);
    const unsigned em = ((q + d) & 3) < q ? 0xfu : 0u;    // source lane is an earlier sample of the item
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const unsigned oo = (unsigned)__shfl((int)o[j], src, 64);
      const float ww = __shfl(w[j], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = oo == o[k];
        tw[k] += same ? ww : 0.f;
        drop |= same ? (em & (1u << k)) : 0u;
      }
    }
  }
  int keep[4], mycnt = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    keep[k] = (s.valid && !((drop >> k) & 1u)) ? 1 : 0;
    mycnt += keep[k];
  }
  int below = 0, n_bin = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int ci = __shfl(mycnt, qbase | i, 64);
    below += i < q ? ci : 0;
    n_bin += ci;
  }
  int* my_list = s_lists + (wave * kItemsPerWave + kb) * kListStride;
  if (q == 0) my_list[0] = n_bin;
  {
    int2* ent = reinterpret_cast<int2*>(my_list + 2);
    int pos = below;
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (keep[k]) ent[pos++] = make_int2((int)o[k], __float_as_int(tw[k] * 0.25f));   // / count (= 4): exact
  }
  __builtin_amdgcn_wave_barrier();   // the lists are private to the wave; a wave's LDS operations retire in order

  // ---- taps: group of 8 lanes = one item x 32 channels ----
  const int grp = lane >> 3, l8 = lane & 7;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(feat), 0, __builtin_amdgcn_readfirstlane((int)((size_t)N * H * W * C * 4)), 0x00020000);
  const int soff = __builtin_amdgcn_readfirstlane(slice * (kSliceC * 4));
  const unsigned lane_off = (unsigned)l8 * 16u;
#pragma unroll 1
  for (int round = 0; round < 2; round++) {
    const int lb = round * 8 + grp;                            // local item of this group
    const int it = item0 + lb;
    const int* list = s_lists + (wave * kItemsPerWave + lb) * kListStride;
    const int n = it < n_items ? list[0] : 0;
    int n_max = 0;
#pragma unroll
    for (int gg = 0; gg < 8; gg++) n_max = max(n_max, __builtin_amdgcn_readlane(n, gg * 8));
    const int2* ent = reinterpret_cast<const int2*>(list + 2);
    const int2 e0 = ent[0];
    const unsigned pad_off = n > 0 ? (unsigned)e0.x : 0u;      // padding taps re-read the item's first pixel (cached)
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < n_max; i0 += BATCH) {
      unsigned e_o[BATCH];
      float e_w[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int2 e = ent[(i0 + u) & 15];
        const bool live = i0 + u < n;
        e_o[u] = live ? (unsigned)e.x : pad_off;
        e_w[u] = live ? __int_as_float(e.y) : 0.f;
      }
      v4f t[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++)
        t[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(e_o[u] + lane_off), soff, 0));
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        acc.x = __builtin_fmaf(e_w[u], t[u].x, acc.x);
        acc.y = __builtin_fmaf(e_w[u], t[u].y, acc.y);
        acc.z = __builtin_fmaf(e_w[u], t[u].z, acc.z);
        acc.w = __builtin_fmaf(e_w[u], t[u].w, acc.w);
      }
    }
    const bool gB = it >= splitB;
    if constexpr (NO != 0) {
      const float val[4] = {acc.x, acc.y, acc.z, acc.w};
      float mixed[4] = {0.f, 0.f, 0.f, 0.f};
      // per-group orientation constants: the switch on `ind` diverges between the (<= 2) RoIs of a wave at most
      ri_dispatch<NO>(mixed, val, lane, gB ? B.ind : A.ind, gB ? B.r_var : A.r_var, gB ? B.l_var : A.l_var);
      acc = v4f{mixed[0], mixed[1], mixed[2], mixed[3]};
    }
    if (it < n_items) {
      const int r = gB ? B.r : A.r;
      const int gbin = it - (gB ? splitB : splitB - nbins);
      float* dst = out + ((size_t)r * nbins + gbin) * C + slice * kSliceC + l8 * 4;
      if (STORE == 0) {
        __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(dst));
      } else if (STORE == 1) {
        *reinterpret_cast<v4f*>(dst) = acc;
      } else {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(acc) : "memory");
      }
    }
  }
}

}  // namespace jdet_roi_sliced
