// Channel-sliced RoIAlign forward for gfx950 (round 4): the product forward of the sample_num == 2 dialects.
//
// Reference semantics: python/jdet/ops/roi_align_rotated.py:L61-127 (and the _v1 / riroi / horizontal twins);
// arithmetic = the merged-tap mode of roi_align.hip (same geometry functions, same tap merge, same fma chain:
// bit-equal results, tests/test_gpu_roi_align.py).
//
// Why sliced.  MI355X has 8 XCDs with a private 4 MiB L2 each, and workgroup b runs on XCD b % 8.  The
// RoI-stationary kernel of rounds 1-3 gave every workgroup all 256 channels of one RoI and cut the map eight ways by
// RoI centre: big RoIs reach across the cuts (1.44 x the map is compulsory), and a pixel row is 1 KiB, so the rows a
// few hundred RoIs in flight touch exceed an L2 (measured: 171 MB read beyond the L2 for a 67 MB map).  Here the cut
// is by CHANNEL: XCD x owns channels [32 x, 32 x + 32) of EVERY RoI, and all XCDs walk the RoIs in the same Morton
// order.  A pixel is then one 128-byte line in exactly one L2, every map byte is compulsory in one L2 only, and half
// of a 256 x 256 map slice (8.4 MB) fits the 4 MiB.  Measured (profiles/r04_roi_fwd_notes.md): 70.5 MB beyond the L2.
//
// Two launches:
//   roi_sort_plan_kernel   block 0: counting sort of the RoIs on the Morton code of their centre (masked RoIs dropped);
//                          blocks 1..: the PLAN -- per item (RoI, bin) the merged tap list [(byte offset, weight / 4)]
//                          of its 4 samples (lane quad = item; geometry with double-precision trig once per RoI).
//                          Done ONCE per item: the first sliced version recomputed it in every (item, slice) wave
//                          and was VALU bound at 80 us (44.6 M VALU instructions, 5 x the RoI-stationary kernel).
//   roi_pool_sliced_kernel items = (RoI in sorted order, bin), flat; a wave owns 16 consecutive items, in two rounds
//                          of 8: a group of 8 lanes = one item x 32 channels (dwordx4 per lane = one 128-byte line
//                          per tap per group), every group walks ITS item's list, BATCH loads in flight; one 128-byte
//                          store per group into the channels-last row.
// The plan costs 8 B per merged tap (7 MB at the north-star point), read once per XCD.
// (Included by roi_align_impl.inc (JDET_ROI_EXPERIMENTAL_MODES) inside its unnamed namespace, after ri_mix<>.)
#pragma once
#include "roi_geom.h"

namespace jdet_roi_sliced {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kSliceC = 32;             // channels per slice = 8 lanes x dwordx4 = one 128-byte line per pixel
constexpr int kItemsPerWave = 16;
constexpr int kMaxTaps = 16;            // 4 samples x 4 taps
constexpr int kListStride = 36;         // dwords per item list in LDS (16 (offset, weight) pairs + 4 pad): 16-byte aligned
                                        // rows, and 36 b mod 64 puts the 8 groups of a round on distinct bank pairs

// workspace: [hdr 256 B][order][rrec][ent]
struct PlanWs {
  int* hdr;            // [0] number of unmasked RoIs
  int* order;          // processing order: sorted position -> RoI
  float4* rrec;        // per RoI: (ind as int bits, l_var, r_var, 0) -- RiRoIAlign orientation constants
  int2* ent;           // per item: kMaxTaps slots of (byte offset into the map tensor, weight); entry 0's offset
                       // carries the list length in its low 5 bits
  size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline PlanWs plan_carve(void* ws, long R, long nbins) {
  PlanWs w;
  char* p = (char*)ws;
  size_t off = 0;
  w.hdr = (int*)(p + off);            off += 256;
  w.order = (int*)(p + off);          off += align256(sizeof(int) * (size_t)R);
  w.rrec = (float4*)(p + off);        off += align256(sizeof(float4) * (size_t)R);
  w.ent = (int2*)(p + off);           off += align256(sizeof(int2) * kMaxTaps * (size_t)R * nbins);
  w.bytes = off;
  return w;
}

// ---------------------------------------------------------------------------------------------------------------
// launch 1: block 0 = schedule, block 1 + j = plans of RoIs 4 j .. 4 j + 3.  1024 threads.
// ---------------------------------------------------------------------------------------------------------------
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

// Counting sort on the Morton code of the RoI centre (32 x 32 cells per image), masked RoIs (batch < 0) dropped.
template <int ROI_COLS>
__device__ __forceinline__ void roi_sort_block(const float* __restrict__ rois, int R, float spatial_scale, int N, int H,
                                               int W, int* __restrict__ hdr, int* __restrict__ order, int* s_bins,
                                               int* s_scan) {
  constexpr int T = 1024;
  const int nimg = min(max(N, 1), kMaxImages);
  const int nkeys = nimg * kCells + 1;           // last key: masked RoIs
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nkeys; i += T) s_bins[i] = 0;
  __syncthreads();
  auto key_of = [&](int r) -> int {
    const float* p = rois + (size_t)r * ROI_COLS;
    const int b = (int)p[0];
    if (b < 0) return nkeys - 1;
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
    const int r = threadIdx.x + i * T;
    mykey[i] = r < R ? key_of(r) : 0;
    if (r < R) atomicAdd(&s_bins[mykey[i]], 1);
  }
  for (int r = threadIdx.x + kKeep * T; r < R; r += T) atomicAdd(&s_bins[key_of(r)], 1);
  __syncthreads();
  const int n_masked = s_bins[nkeys - 1];
  const int per = (nkeys + T - 1) / T;
  const int lo = threadIdx.x * per, hi = min(lo + per, nkeys);
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
  // masked RoIs sort last and are written as -1: the pool kernel stops at the first negative entry
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int r = threadIdx.x + i * T;
    if (r < R) order[atomicAdd(&s_bins[mykey[i]], 1)] = mykey[i] == nkeys - 1 ? -1 : r;
  }
  for (int r = threadIdx.x + kKeep * T; r < R; r += T) {
    const int k = key_of(r);
    order[atomicAdd(&s_bins[k], 1)] = k == nkeys - 1 ? -1 : r;
  }
}

template <int VARIANT>
__global__ __launch_bounds__(1024, 8) void roi_sort_plan_kernel(const float* __restrict__ rois, int R, float spatial_scale,
                                                            int N, int pix_bytes_, int H, int W, int PH, int PW,
                                                            int nO, int* __restrict__ hdr, int* __restrict__ order,
                                                            float4* __restrict__ rrec, int2* __restrict__ ent) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  __shared__ int s_bins[kMaxImages * kCells + 1];
  __shared__ int s_scan[16];
  __shared__ RoiGeom s_geom[4];
  if (blockIdx.x == 0) {
    roi_sort_block<ROI_COLS>(rois, R, spatial_scale, N, H, W, hdr, order, s_bins, s_scan);
    return;
  }
  // plan: quarter `sub` of the workgroup (256 threads = 64 lane quads) serves one RoI; lane quad = item (RoI, bin),
  // lane = one of its 4 samples
  const int sub = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int r = (blockIdx.x - 1) * 4 + sub;
  if (t == 0 && r < R) {
    const float* roi = rois + (size_t)r * ROI_COLS;
    RoiGeom g = roi_geom<VARIANT>(roi, spatial_scale, 2, PH, PW, max(nO, 1), false);
    if (g.batch >= N) {      // no such image: every sample out of range -> zeros (never an out-of-bounds read)
      g.batch = 0;
      g.center_w = g.center_h = g.start_w = g.start_h = -1e30f;
    }
    s_geom[sub] = g;
    int ind = 0;
    float l_var = 0.f, r_var = 1.f;
    if (nO > 1 && ROI_COLS == 6) ri_params(roi[5], nO, ind, l_var, r_var);
    rrec[r] = make_float4(__int_as_float(ind), l_var, r_var, 0.f);
  }
  __syncthreads();
  if (r >= R) return;
  const RoiGeom g = s_geom[sub];
  if (g.batch < 0) return;                       // masked: the pool kernel never visits it
  const int nbins = PH * PW;
  const int lane = t & 63, q = lane & 3, qbase = lane & ~3;
  const unsigned pix_bytes = (unsigned)pix_bytes_;   // bytes from one pixel to the next (4 C in the NHWC map)
  const unsigned img_off = (unsigned)g.batch * (unsigned)(H * W);       // pixels before this image
  for (int b0 = 0; b0 < nbins; b0 += 64) {
    const int bin = b0 + (t >> 2);
    const bool bin_ok = bin < nbins;
    const int bb = bin_ok ? bin : 0;
    Sample s = make_sample<VARIANT>(g, bb / PW, bb % PW, q >> 1, q & 1, H, W);
    if (!bin_ok) s.valid = 0;
    // byte offsets of the 4 taps; an invalid sample carries a sentinel no tap can equal (offsets are multiples of 4)
    const unsigned kNone = 0xffffffffu;
    const unsigned o[4] = {s.valid ? (img_off + s.o1) * pix_bytes : kNone, s.valid ? (img_off + s.o2) * pix_bytes : kNone,
                           s.valid ? (img_off + s.o3) * pix_bytes : kNone, s.valid ? (img_off + s.o4) * pix_bytes : kNone};
    const float w[4] = {s.w1, s.w2, s.w3, s.w4};
    // Tap merge inside the quad (= the 16 taps of the item): the first occurrence of a pixel collects the weights of
    // the others, the rest are dropped (same order of additions as roi_align_fwd_merged_kernel).  Integer masks in
    // VGPRs on purpose: as bool arrays the 48 compare results become 48 live SGPR-pair masks and spill.
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
    if (bin_ok) {
      // entry 0 carries the list length in the low bits of its offset (offsets are multiples of 4 C >= 128 bytes)
      int2* dst = ent + ((size_t)r * nbins + bin) * kMaxTaps;
      int pos = below;
#pragma unroll
      for (int k = 0; k < 4; k++)
        if (keep[k]) {
          dst[pos] = make_int2((int)(o[k] | (pos == 0 ? (unsigned)n_bin : 0u)), __float_as_int(tw[k] * 0.25f));   // / count (= 4): exact
          pos++;
        }
      if (n_bin == 0 && q == 0) dst[0] = make_int2(0, 0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// launch 2.  grid.x = nslices * ceil(R * nbins / 64) workgroups of 4 waves; slice = blockIdx.x % nslices.
//   NO     0: plain RoIAlign; 4 / 8: RiRoIAlign orientation mix on the finished bin
//   BATCH  tap loads issued back to back per group before the first use (4 / 8 / 16)
//   PRED   0: a group with fewer taps than the wave's longest list pads with (first pixel, weight 0) -- a cached
//             address; 1: its lanes are switched off for the surplus loads
// ---------------------------------------------------------------------------------------------------------------
template <int NO, int BATCH, int PRED, int NW = 4>
__global__ __launch_bounds__(NW * 64) void roi_pool_sliced_kernel(
    const float* __restrict__ feat, const int* __restrict__ order, const float4* __restrict__ rrec,
    const int2* __restrict__ ent, float* __restrict__ out, int R, int N, int C, int HW, int nbins, int nslices,
    unsigned slice_stride) {
  __shared__ __attribute__((aligned(16))) int s_lists[NW * kItemsPerWave * kListStride];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int slice = __builtin_amdgcn_readfirstlane(blockIdx.x % nslices);
  const int blk = __builtin_amdgcn_readfirstlane(blockIdx.x / nslices);
  const int n_items = R * nbins;
  const int item0 = (blk * NW + wave) * kItemsPerWave;
  if (item0 >= n_items) return;                               // (no workgroup-wide barrier below)
  // nbins >= 16 on this path: the wave's items belong to at most two RoIs.  Hop 1: which RoIs (-1: past the last one)
  const int sA = item0 / nbins;
  const int rA = __builtin_amdgcn_readfirstlane(order[sA]);
  const int rB = __builtin_amdgcn_readfirstlane(order[min(sA + 1, R - 1)]);
  if (rA < 0) return;
  const int splitB = (sA + 1) * nbins;                        // items >= splitB belong to rB
  const int grp = lane >> 3, l8 = lane & 7;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(feat), 0, __builtin_amdgcn_readfirstlane((int)((size_t)N * HW * C * 4)), 0x00020000);
  const int soff = __builtin_amdgcn_readfirstlane((int)((unsigned)slice * slice_stride));
  const unsigned lane_off = (unsigned)l8 * 16u;
  // Hop 2: both rounds' lists -- lane l8 fetches entries 2 l8, 2 l8 + 1 of its group's item (the whole 128-byte slot
  // in one instruction per round: a partial read would pull the line anyway), staged in the wave's LDS block.
  bool it_ok2[2];
  size_t slot2[2];
  int r2[2], n2[2];
  v4i c[2];
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const int it = item0 + round * 8 + grp;
    const bool gB = it >= splitB;
    r2[round] = gB ? rB : rA;
    it_ok2[round] = it < n_items && r2[round] >= 0;
    const int bin = it - (gB ? splitB : splitB - nbins);
    slot2[round] = it_ok2[round] ? (size_t)r2[round] * nbins + bin : (size_t)rA * nbins;
    c[round] = *reinterpret_cast<const v4i*>(ent + slot2[round] * kMaxTaps + 2 * l8);
  }
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const int nn = __shfl(c[round].x & 31, lane & ~7, 64);    // list length: low bits of entry 0's offset
    n2[round] = it_ok2[round] ? nn : 0;
    if (l8 == 0) c[round].x &= ~127;
    *reinterpret_cast<v4i*>(s_lists + (wave * kItemsPerWave + round * 8 + grp) * kListStride + 4 * l8) = c[round];
  }
  __builtin_amdgcn_wave_barrier();     // the lists are private to the wave; a wave's LDS operations retire in order
#pragma unroll
  for (int round = 0; round < 2; round++) {
    const int lb = round * 8 + grp;                            // local item of this group
    const bool it_ok = it_ok2[round];
    const int r = r2[round], n = n2[round];
    const size_t slot = slot2[round];
    const int* list = s_lists + (wave * kItemsPerWave + lb) * kListStride;
    int n_max = 0;
#pragma unroll
    for (int gg = 0; gg < 8; gg++) n_max = max(n_max, __builtin_amdgcn_readlane(n, gg * 8));
    const int2* le = reinterpret_cast<const int2*>(list);
    unsigned pad_off = 0;
    if (!PRED) {
      const int2 e0 = le[0];
      pad_off = n > 0 ? (unsigned)e0.x : 0u;   // padding taps re-read the item's first pixel (cached)
    }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    for (int i0 = 0; i0 < n_max; i0 += BATCH) {
      unsigned e_o[BATCH];
      float e_w[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int2 e = le[(i0 + u) & (kMaxTaps - 1)];
        const bool live = i0 + u < n;
        e_o[u] = live ? (unsigned)e.x : pad_off;
        e_w[u] = live ? __int_as_float(e.y) : 0.f;
      }
      v4f t[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        if (PRED) {
          t[u] = v4f{0.f, 0.f, 0.f, 0.f};
          if (i0 + u < n)
            t[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(e_o[u] + lane_off), soff, 0));
        } else {
          t[u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(e_o[u] + lane_off), soff, 0));
        }
      }
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        acc.x = __builtin_fmaf(e_w[u], t[u].x, acc.x);
        acc.y = __builtin_fmaf(e_w[u], t[u].y, acc.y);
        acc.z = __builtin_fmaf(e_w[u], t[u].z, acc.z);
        acc.w = __builtin_fmaf(e_w[u], t[u].w, acc.w);
      }
    }
    if constexpr (NO != 0) {
      const float4 rr = rrec[it_ok ? r : rA];
      const float val[4] = {acc.x, acc.y, acc.z, acc.w};
      float mixed[4];
      ri_mix<NO>(mixed, val, lane, __float_as_int(rr.x), rr.z, rr.y);     // per-group orientation constants
      acc = v4f{mixed[0], mixed[1], mixed[2], mixed[3]};
    }
    if (it_ok)
      __builtin_nontemporal_store(acc, reinterpret_cast<v4f*>(out + slot * C + slice * kSliceC + l8 * 4));
  }
}

}  // namespace jdet_roi_sliced
