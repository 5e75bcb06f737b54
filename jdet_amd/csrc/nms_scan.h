// Greedy NMS scan shared by the rotated-box and polygon NMS (box_iou_rotated.hip, poly_iou.hip): the suppression
// bit matrix (n x ceil(n/64) u64, upper triangle, rows / columns in visiting order) is produced by the caller's tile
// kernel, the scan below turns it into the keep flags on the device.
#pragma once
#include "common.h"

namespace jdet_nms {

constexpr int kScanBlock = 1024;
constexpr int kScanMaxWords = 8192;  // n <= 524288

// Greedy scan.  Boxes of different labels never interact and are visited label by label, so every label is its own
// greedy problem: workgroup g scans the positions [seg_begin, seg_end) whose label is g (n_labels == 1: everything).
// Per 64-row block: wave 0 resolves the diagonal tile (readlane chain), then all 16 waves OR the rows of the kept
// boxes into the running `removed` words of the column blocks that can be affected (<= tile_jmax): the (kept row,
// column) pairs are flattened over the 1024 threads, 8 independent loads in flight per thread.  Row blocks that
// straddle two labels are visited by both workgroups, each touching only its own rows.
static __global__ __launch_bounds__(kScanBlock) void nms_scan_kernel(const unsigned long long* __restrict__ mask,
                                                              int n, const int32_t* __restrict__ order,
                                                              const int* __restrict__ tile_jmax,
                                                              const float* __restrict__ labels, int label_stride,
                                                              int n_labels, uint8_t* __restrict__ keep) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long s_remv[];  // col_blocks words
  __shared__ int s_rows[64];
  __shared__ int s_nkept;
  __shared__ int s_seg[2];
  const int col_blocks = (n + 63) >> 6;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) {
    s_seg[0] = n_labels > 1 ? n : 0;
    s_seg[1] = n;
  }
  __syncthreads();
  if (n_labels > 1) {
    // positions are sorted by label: the segment of label g starts at the first position whose label is >= g and
    // ends where the first label >= g + 1 sits; every boundary is found by exactly one thread
    const float g = (float)blockIdx.x;
    for (int pos = threadIdx.x; pos < n; pos += kScanBlock) {
      const float l = labels[(size_t)order[pos] * label_stride];
      const float lp = pos > 0 ? labels[(size_t)order[pos - 1] * label_stride] : -INFINITY;
      if (lp < g && l >= g) s_seg[0] = pos;
      if (lp < g + 1.f && l >= g + 1.f) s_seg[1] = pos;
    }
    __syncthreads();
  }
  const int seg_lo = s_seg[0], seg_hi = min(s_seg[1], n);
  if (seg_lo >= seg_hi) return;
  const int c_lo = seg_lo >> 6, c_hi = (seg_hi + 63) >> 6;
  for (int j = c_lo + threadIdx.x; j < col_blocks; j += kScanBlock) s_remv[j] = 0ull;
  __syncthreads();
  // The rows a block contributes depend on which of its boxes survive, but WHICH words can be needed does not: all
  // 64 rows x (columns c+1 .. jmax) of block c+1 are fetched into registers while block c is being resolved, so the
  // scan's critical path holds no global-memory round trip (it had two per block: 142 us for 2000 boxes).  Blocks too
  // wide for 8 words per thread keep the fetch-after-resolve path.
  __shared__ unsigned long long s_keepbits;
  constexpr int kPre = 8;
  auto block_cols = [&](int c) { return max(min(tile_jmax[c], c_hi - 1) - c, 0); };
  auto fetch = [&](int c, unsigned long long (&w)[kPre], unsigned long long& diag) {
    const int ncols = block_cols(c), items = 64 * ncols, rows = min(64, n - c * 64);
    diag = 0ull;
    if (wave == 0) {
      const int row = c * 64 + lane;
      if (row >= seg_lo && row < seg_hi) diag = mask[(size_t)row * col_blocks + c];
    }
#pragma unroll
    for (int u = 0; u < kPre; u++) {
      const int it = threadIdx.x + u * kScanBlock;
      w[u] = 0ull;
      if (items <= kPre * kScanBlock && it < items) {
        const int ri = it / ncols;
        if (ri < rows) w[u] = mask[(size_t)(c * 64 + ri) * col_blocks + c + 1 + (it - ri * ncols)];
      }
    }
  };
  unsigned long long cur[kPre], nxt[kPre], dcur, dnxt = 0ull;
  fetch(c_lo, cur, dcur);
  for (int c = c_lo; c < c_hi; c++) {
    if (c + 1 < c_hi) fetch(c + 1, nxt, dnxt);
    if (wave == 0) {
      // diagonal tile: lane = row; resolve the within-tile greedy dependency with readlanes
      const int row = c * 64 + lane;
      const bool own = row >= seg_lo && row < seg_hi;
      const unsigned long long d = dcur;
      unsigned long long removed = s_remv[c];
      unsigned long long keepbits = 0ull;
      const unsigned long long ownbits = __ballot(own);
      const unsigned int dlo = (unsigned int)d, dhi = (unsigned int)(d >> 32);
      // fully unrolled: row index and readlane lane are immediates, a suppressed row costs a bit test and a branch
      // (the rolled loop spent ~20 scalar instructions per row on 64-bit shifts: 3 us of a lone wave per block);
      // rows past the end / of other labels have their `own` bit clear
#pragma unroll
      for (int i = 0; i < 64; i++) {
        if (((ownbits & ~removed) >> i) & 1ull) {
          keepbits |= 1ull << i;
          const unsigned long long di =
              ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dhi, i) << 32) |
              (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)dlo, i);
          removed |= di;
        }
      }
      const bool mine = (keepbits >> lane) & 1ull;
      if (own) keep[order[row]] = (uint8_t)mine;
      if (mine) s_rows[__popcll(keepbits & ((1ull << lane) - 1ull))] = lane;   // k-th kept row of the block
      if (lane == 0) {
        s_nkept = __popcll(keepbits);
        s_keepbits = keepbits;
      }
    }
    __syncthreads();
    const int ncols = block_cols(c);                // column blocks c+1 .. jmax (later labels' columns hold no bit)
    if (64 * ncols <= kPre * kScanBlock) {
      const unsigned long long kb = s_keepbits;
#pragma unroll
      for (int u = 0; u < kPre; u++) {
        const int it = threadIdx.x + u * kScanBlock;
        if (cur[u]) {   // (non-zero only for it < 64 * ncols)
          const int ri = it / ncols;
          if ((kb >> ri) & 1ull) atomicOr(&s_remv[c + 1 + (it - ri * ncols)], cur[u]);
        }
      }
    } else {
      const int items = s_nkept * ncols;
      for (int it0 = threadIdx.x; it0 < items; it0 += kScanBlock * 8) {
        unsigned long long w[8];
        int jj[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int it = it0 + u * kScanBlock;
          w[u] = 0ull;
          jj[u] = 0;
          if (it < items) {
            const int ri = it / ncols;
            jj[u] = c + 1 + (it - ri * ncols);
            w[u] = mask[(size_t)(c * 64 + s_rows[ri]) * col_blocks + jj[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (w[u]) atomicOr(&s_remv[jj[u]], w[u]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kPre; u++) cur[u] = nxt[u];
    dcur = dnxt;
  }
}


inline size_t mask_bytes(int n) {
  const size_t col_blocks = ((size_t)n + 63) >> 6;
  return (((size_t)n * col_blocks * sizeof(unsigned long long)) + 255) & ~(size_t)255;
}

// mask (n x col_blocks u64) + tile_jmax (col_blocks int): both zeroed by the caller before the tile kernel runs
inline size_t workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t col_blocks = ((size_t)n + 63) >> 6;
  return mask_bytes(n) + ((col_blocks * sizeof(int) + 255) & ~(size_t)255);
}

// labels: pointer to the label of box 0, label_stride floats between boxes (ignored when n_labels == 1)
inline int launch_scan(const unsigned long long* mask, int n, const int32_t* order, const int* tile_jmax,
                       const float* labels, int label_stride, int n_labels, uint8_t* keep, hipStream_t st) {
  const int col_blocks = (n + 63) >> 6;
  const size_t lds = (size_t)col_blocks * sizeof(unsigned long long);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(n_labels), dim3(kScanBlock), lds, st, mask, n, order, tile_jmax, labels,
                     label_stride, n_labels, keep);
  return jdet_launch_status();
}

}  // namespace jdet_nms
