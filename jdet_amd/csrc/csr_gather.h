// Sorted-gather machinery shared by the scatter-shaped backward passes (RoIAlign backward, deformable
// col2im): a scatter  dst[key, :] += w * src[row, :]  over many (key, row, w) taps is inverted once into a
// CSR over `key` (integer atomics only) and executed as a gather -- one wave per destination row, lanes =
// channels, one coalesced store per row.  See roi_align_bwd.hip for the derivation and measurements.
#pragma once
#include "common.h"

namespace jdet_csr {

typedef float v4f __attribute__((ext_vector_type(4)));

struct Entry {
  int src;    // roi * nbins + bin
  float w;    // bilinear weight / count
};

// Exclusive scan of the n = N*H*W row counters in ONE launch: every workgroup scans its own 2048-element tile
// (offsets[] then holds tile-local offsets) and publishes the tile total; the workgroup that finishes last (ticket
// counter) scans the <= few hundred tile totals into tile_base[] (tile_base[ntiles] = grand total) and resets the
// ticket.  A row's place = offsets[key] + tile_base[key / kScanTile] (row_begin below).
constexpr int kScanTile = 2048;  // 256 threads x 8
constexpr int kScanTileLog2 = 11;

// `cap` / `skip_if_zero` (patch-keyed variant below): a row's first `cap` entries live outside the CSR, so the scanned
// length is max(count - cap, 0); a launch whose *skip_if_zero is 0 (no row went past its cap) has nothing to do.
static __global__ __launch_bounds__(256) void csr_scan_kernel(const int* __restrict__ counts, int n, int ntiles,
                                                              int* __restrict__ offsets, int* __restrict__ tile_sum,
                                                              int* __restrict__ tile_base, int* __restrict__ ticket,
                                                              int cap = 0, const int* __restrict__ skip_if_zero = nullptr) {
  if (skip_if_zero != nullptr && *skip_if_zero == 0) return;
  __shared__ int s_wave[4];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = blockIdx.x * kScanTile + threadIdx.x * 8;
  int v[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    v[i] = lo + i < n ? max(counts[lo + i] - cap, 0) : 0;
    sum += v[i];
  }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int u = __shfl_up(incl, off, 64);
    if (lane >= off) incl += u;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; w++) run += s_wave[w];
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (lo + i < n) {
      offsets[lo + i] = run;
      run += v[i];
    }
  if (threadIdx.x == 255) {
    __hip_atomic_store(&tile_sum[blockIdx.x], run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();   // the total is visible device-wide before the ticket is taken
    s_last = atomicAdd(ticket, 1) == ntiles - 1;
  }
  __syncthreads();
  if (!s_last) return;
  __threadfence();
  // last workgroup: exclusive scan of the tile totals, 256 at a time
  __shared__ int s_carry;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int t0 = 0; t0 < ntiles; t0 += 256) {
    const int t = t0 + threadIdx.x;
    const int x = t < ntiles ? __hip_atomic_load(&tile_sum[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    int inc = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int u = __shfl_up(inc, off, 64);
      if (lane >= off) inc += u;
    }
    __syncthreads();               // s_wave of the previous round / of the tile scan has been read
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int base = s_carry;
    for (int w = 0; w < wave; w++) base += s_wave[w];
    if (t < ntiles) tile_base[t] = base + inc - x;
    __syncthreads();
    if (threadIdx.x == 255) s_carry = base + inc;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    tile_base[ntiles] = s_carry;
    *ticket = 0;                   // ready for the next launch on this workspace
  }
}

__device__ __forceinline__ int row_begin(const int* __restrict__ offsets, const int* __restrict__ tile_base, int key) {
  return offsets[key] + tile_base[key >> kScanTileLog2];
}

// tap_pos[e] = the value the tap's atomicAdd on counts[key] returned when the row lengths were counted: its place
// inside the row.  No second round of atomics here.
static __global__ __launch_bounds__(256) void csr_fill_kernel(const int* __restrict__ tap_key,
                                                      const int* __restrict__ tap_pos,
                                                      const float* __restrict__ tap_w, long ntaps, int spb4,
                                                      const int* __restrict__ offsets,
                                                      const int* __restrict__ tile_base, Entry* __restrict__ entries) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= ntaps) return;
  const int key = tap_key[e];
  if (key < 0) return;
  const int pos = row_begin(offsets, tile_base, key) + tap_pos[e];
  Entry en;
  en.src = (int)(e / spb4);  // (roi * nbins + bin): taps are ordered roi, bin, sample, tap
  en.w = tap_w[e];
  entries[pos] = en;
}

// One wave per destination row (pixel); lane owns 4 consecutive channels of a 256-channel chunk.
// Work mapping (row_w > 0: the keys are the pixels of row_w-wide images stacked into n_rows rows): a workgroup's 4
// waves take a 2x2 pixel patch -- the four taps of a bilinear sample are a 2x2 patch and read the same source row,
// so the patch shares its L1 lines -- and stripes of 8 pixel rows go round-robin to the 8 XCDs (workgroup b runs
// on XCD b % 8), so a source row is fetched into one XCD's L2 (two at a stripe border) instead of into all of them,
// while clustered RoIs still spread over the XCDs.  row_w == 0: 4 consecutive keys per workgroup.
// The row counters are handed back zeroed (the next CSR build on this workspace needs no memset).
template <int UNROLL>
static __global__ __launch_bounds__(256) void csr_gather_kernel(const float* __restrict__ gT,
                                                        const int* __restrict__ offsets,
                                                        const int* __restrict__ tile_base, int ntiles,
                                                        const Entry* __restrict__ entries, int npix, int C,
                                                        int row_w, int n_rows, int* __restrict__ counts,
                                                        float* __restrict__ grad_in) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  int p;
  if (row_w > 0) {
    const int pw = (row_w + 1) >> 1;                        // patches per patch row
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int stripe = xcd + 8 * (j / (4 * pw));            // 4 patch rows (8 pixel rows) per stripe
    const int local = j % (4 * pw);
    const int y = (stripe * 4 + local / pw) * 2 + (wave >> 1);
    const int x = (local % pw) * 2 + (wave & 1);
    if (y >= n_rows || x >= row_w) return;
    p = y * row_w + x;
  } else {
    p = blockIdx.x * 4 + wave;
    if (p >= npix) return;
  }
  const int beg = __builtin_amdgcn_readfirstlane(row_begin(offsets, tile_base, p));
  const int end = __builtin_amdgcn_readfirstlane(p + 1 < npix ? row_begin(offsets, tile_base, p + 1)
                                                              : tile_base[ntiles]);
  if (lane == 0) counts[p] = 0;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool ok = c < C;                       // C % 4 == 0 on this path
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    int i = beg;
    for (; i + UNROLL <= end; i += UNROLL) {
      Entry en[UNROLL];
      v4f v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) en[u] = entries[i + u];   // wave-uniform -> scalar loads
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
        v[u] = ok ? *reinterpret_cast<const v4f*>(gT + (size_t)en[u].src * C + c) : v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int u = 0; u < UNROLL; u++) acc += en[u].w * v[u];
    }
    for (; i < end; i++) {
      const Entry en = entries[i];
      if (ok) acc += en.w * *reinterpret_cast<const v4f*>(gT + (size_t)en.src * C + c);
    }
    if (ok) *reinterpret_cast<v4f*>(grad_in + (size_t)p * C + c) = acc;
  }
}


inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// workspace layout of one CSR build over `nkeys` destination rows and `ntaps` taps
struct CsrWs {
  int* counts;     // nkeys row counters + the scan's ticket: csr_zero_bytes() bytes, zero before the taps are counted
  int* offsets;
  int* tile_sum;
  int* tile_base;
  int* tap_key;
  int* tap_pos;
  float* tap_w;
  Entry* entries;
  size_t bytes;
};

inline CsrWs csr_carve(void* ws, long nkeys, long ntaps) {
  CsrWs w;
  char* p = (char*)ws;
  size_t off = 0;
  const long ntiles = (nkeys + kScanTile - 1) / kScanTile;
  w.counts = (int*)(p + off);   off += align256(sizeof(int) * (nkeys + 1));
  w.offsets = (int*)(p + off);  off += align256(sizeof(int) * (nkeys + 1));
  w.tile_sum = (int*)(p + off); off += align256(sizeof(int) * (ntiles + 1));
  w.tile_base = (int*)(p + off); off += align256(sizeof(int) * (ntiles + 1));
  w.tap_key = (int*)(p + off);  off += align256(sizeof(int) * ntaps);
  w.tap_pos = (int*)(p + off);  off += align256(sizeof(int) * ntaps);
  w.tap_w = (float*)(p + off);  off += align256(sizeof(float) * ntaps);
  w.entries = (Entry*)(p + off); off += align256(sizeof(Entry) * ntaps);
  w.bytes = off;
  return w;
}

inline size_t csr_zero_bytes(long nkeys) { return sizeof(int) * (size_t)(nkeys + 1); }

// counts[] must already hold the row lengths and tap_key / tap_pos / tap_w the taps (key < 0 = dropped tap;
// tap_pos = return value of the atomicAdd that counted it); counts[nkeys] (the ticket) must be zero.
// taps_per_src consecutive taps share one source row (entry.src = tap index / taps_per_src).
// row_w / n_rows: see csr_gather_kernel (0, 0: no spatial work mapping).
// On return (stream order) the first csr_zero_bytes(nkeys) bytes of the workspace are zero again.
inline int csr_finish_and_gather(const CsrWs& w, long nkeys, long ntaps, int taps_per_src, const float* src, int C,
                                 float* dst, int row_w, int n_rows, hipStream_t st) {
  const int ntiles = (int)((nkeys + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(csr_scan_kernel, dim3(ntiles), dim3(256), 0, st, w.counts, (int)nkeys, ntiles, w.offsets,
                     w.tile_sum, w.tile_base, w.counts + nkeys);
  hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((ntaps + 255) / 256)), dim3(256), 0, st, w.tap_key, w.tap_pos,
                     w.tap_w, ntaps, taps_per_src, w.offsets, w.tile_base, w.entries);
  unsigned blocks = (unsigned)((nkeys + 3) / 4);
  if (row_w > 0) {
    const int pw = (row_w + 1) / 2, patch_rows = (n_rows + 1) / 2, stripes = (patch_rows + 3) / 4;
    blocks = 8u * (unsigned)((stripes + 7) / 8) * 4u * (unsigned)pw;
  }
  hipLaunchKernelGGL((csr_gather_kernel<4>), dim3(blocks), dim3(256), 0, st, src, w.offsets, w.tile_base, ntiles,
                     w.entries, (int)nkeys, C, row_w, n_rows, w.counts, dst);
  return jdet_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// Patch-keyed variant.  The four taps of a bilinear sample are a 2x2 pixel block, and neighbouring samples of a bin
// overlap: keyed by PIXEL, every source row is fetched once per pixel it touches (~10x on the RoIAlign bench, 1 GB
// through the vector L1 for a 100 MB gradient -- the gather ran at 56 % of the L1's bandwidth, i.e. it was bound by
// that).  Keyed by the aligned 2x2 PATCH (key = image * ceil(H/2) * ceil(W/2) + (y >> 1) * ceil(W/2) + (x >> 1)), an
// entry carries one source row and the four weights of the patch's pixels: the taps of one source row that fall into
// one patch merge into ONE entry, one wave accumulates the four pixels of a patch from one row load per entry.
// Fewer entries (0.46 x), fewer counter atomics, and the L1 traffic of the gather drops with them.
struct TapRec {      // written by the producer kernel (compacted: kept taps only), consumed by csr_fill_patch_kernel
  int key;           // patch
  int pos;           // place in the patch's row = return value of the atomicAdd that counted it
  int src;           // source row
  float w[4];        // weights of pixels (y0,x0) (y0,x0+1) (y0+1,x0) (y0+1,x0+1)
  int pad;
};
struct PatchEntry {
  int src;
  float w[4];
  int pad[3];
};
static_assert(sizeof(TapRec) == 32 && sizeof(PatchEntry) == 32, "32-byte records: aligned vector / scalar accesses");

// Direct rows.  A patch's first `cap` entries (cap = a power of two chosen from the shape, patch_direct_cap) have a
// fixed home, direct[key * cap + place]: the producer writes them where the gather reads them, and the count -> scan
// -> fill round trip (5.9 + 9.9 us of the 76 us backward at the north-star point) only runs for what goes past the
// cap -- rows under clustered RoIs -- with the CSR machinery unchanged (cap = 0 IS the path of rounds 2-5, and runs at
// its speed: 76.2 us).  A producer workgroup that wrote overflow records raises over_flag (a plain store of 1: a
// SUM there cost 13 us -- 2000 atomics on one address); the scan and fill launches find 0 and return (~1.6 us each).
// 76.3 -> 66.5-67.5 us for the whole backward (profiles/r05_roi_bwd_notes.md).
struct PatchWs {
  int* counts;      // nkeys row counters, [nkeys] = scan ticket, [nkeys + 1] = over_flag: patch_zero_bytes()
  int* offsets;
  int* tile_sum;
  int* tile_base;
  int* seg_n;       // overflow records per producer segment (one per producer workgroup, seg_cap records each)
  TapRec* recs;
  PatchEntry* entries;   // CSR of the overflow
  PatchEntry* direct;    // nkeys x cap
  int cap;
  size_t bytes;
};

inline size_t patch_zero_bytes(long nkeys) { return sizeof(int) * (size_t)(nkeys + 2); }

// cap = the power of two >= 2.5 x the expected row length (about half of the taps survive the merge), 32..256, halved
// while the direct rows would take more than 256 MiB; 0 = no direct rows
inline int patch_direct_cap(long nkeys, long max_recs) {
  const double expect = 0.5 * (double)max_recs / (double)(nkeys > 0 ? nkeys : 1);
  int cap = 32;
  while (cap < 256 && (double)cap < 2.5 * expect) cap <<= 1;
  while (cap >= 16 && (size_t)nkeys * cap * sizeof(PatchEntry) > ((size_t)256 << 20)) cap >>= 1;
  return cap >= 16 ? cap : 0;
}

inline PatchWs patch_carve(void* ws, long nkeys, long nsegs, long seg_cap, int cap) {
  const long max_recs = nsegs * seg_cap;
  PatchWs w;
  char* p = (char*)ws;
  size_t off = 0;
  const long ntiles = (nkeys + kScanTile - 1) / kScanTile;
  w.cap = cap;
  w.counts = (int*)(p + off);    off += align256(sizeof(int) * (nkeys + 2));
  w.offsets = (int*)(p + off);   off += align256(sizeof(int) * (nkeys + 1));
  w.tile_sum = (int*)(p + off);  off += align256(sizeof(int) * (ntiles + 1));
  w.tile_base = (int*)(p + off); off += align256(sizeof(int) * (ntiles + 1));
  w.seg_n = (int*)(p + off);     off += align256(sizeof(int) * nsegs);
  w.recs = (TapRec*)(p + off);   off += align256(sizeof(TapRec) * max_recs);
  w.entries = (PatchEntry*)(p + off); off += align256(sizeof(PatchEntry) * max_recs);
  w.direct = (PatchEntry*)(p + off);  off += align256(sizeof(PatchEntry) * (size_t)nkeys * cap);
  w.bytes = off;
  return w;
}

static __global__ __launch_bounds__(256) void csr_fill_patch_kernel(const TapRec* __restrict__ recs,
                                                                    const int* __restrict__ seg_n, int seg_cap,
                                                                    const int* __restrict__ offsets,
                                                                    const int* __restrict__ tile_base,
                                                                    PatchEntry* __restrict__ entries) {
  const int n = seg_n[blockIdx.x];
  const TapRec* __restrict__ seg = recs + (size_t)blockIdx.x * seg_cap;
  for (int e = threadIdx.x; e < n; e += 256) {
    const int4 a = reinterpret_cast<const int4*>(seg + e)[0];   // key, pos, src, w0
    const int4 b = reinterpret_cast<const int4*>(seg + e)[1];   // w1, w2, w3, pad
    const int dst = row_begin(offsets, tile_base, a.x) + a.y;
    reinterpret_cast<int4*>(entries + dst)[0] = make_int4(a.z, a.w, b.x, b.y);
    reinterpret_cast<int4*>(entries + dst)[1] = make_int4(b.z, 0, 0, 0);
  }
}

// acc[q] += w[q] * row(src) over `count` entries at `ent`: fetched 64 at a time, one per lane (one coalesced 2 KiB
// read), and handed round by readlane -- no scalar-load latency inside the loop, UNROLL row loads issued back to back
// (the row loads are L2 latency bound: with 4 in flight behind a scalar entry load per iteration the gather ran 46 us,
// patch-keyed or not).
template <int UNROLL>
__device__ __forceinline__ void patch_accumulate(const PatchEntry* __restrict__ ent, int count,
                                                 const float* __restrict__ col, int C, int lane, v4f (&acc)[4]) {
  for (int base = 0; base < count; base += 64) {
    const int n = min(64, count - base);
    // lane l: entry base + l; lanes past the end: the batch's first source row with zero weights (a cached address)
    const int4 a = reinterpret_cast<const int4*>(ent + base + (lane < n ? lane : 0))[0];
    const float w3 = reinterpret_cast<const float*>(ent + base + (lane < n ? lane : 0))[4];
    const int e_src = a.x;
    const float e_w[4] = {lane < n ? __int_as_float(a.y) : 0.f, lane < n ? __int_as_float(a.z) : 0.f,
                          lane < n ? __int_as_float(a.w) : 0.f, lane < n ? w3 : 0.f};
    for (int i = 0; i < n; i += UNROLL) {
      v4f v[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; u++) {
        const int src = __builtin_amdgcn_readlane(e_src, (i + u) & 63);
        v[u] = *reinterpret_cast<const v4f*>(col + (size_t)src * C);
      }
#pragma unroll
      for (int u = 0; u < UNROLL; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e_w[q]), (i + u) & 63));
          acc[q] += w * v[u];
        }
    }
  }
}

// One wave per patch; lane owns 4 consecutive channels of a 256-channel chunk; 4 accumulators (the patch's pixels).
// Workgroup = 2x2 patches; every XCD owns one 2-D block of the workgroup grid (below).
// img_h / img_w: pixel size of one image; keys = images x ceil(img_h/2) x ceil(img_w/2) patches.
// A patch's entries: min(count, cap) in its direct row, the rest (count > cap only) in the CSR of the overflow.
// The row counters and over_flag are handed back zeroed (a patch's counter after its entries were read).
// KEEP: the counters / flag stay as they are -- the caller gathers again from the same rows (a plan kept from the forward)
template <int UNROLL, bool KEEP = false>
static __global__ __launch_bounds__(256) void csr_gather_patch_kernel(const float* __restrict__ gT,
                                                              const int* __restrict__ offsets,
                                                              const int* __restrict__ tile_base,
                                                              const PatchEntry* __restrict__ entries,
                                                              const PatchEntry* __restrict__ direct, int cap,
                                                              int nkeys, int C, int n_img, int img_h, int img_w,
                                                              int* __restrict__ counts, float* __restrict__ grad_in) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (!KEEP && blockIdx.x == 0 && threadIdx.x == 0) counts[nkeys + 1] = 0;   // over_flag: scan and fill have read it
  const int php = (img_h + 1) >> 1, pwp = (img_w + 1) >> 1;
  const int bw = (pwp + 1) >> 1;                               // workgroups per row of patches
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  // XCD x owns ONE 2-D block of the workgroup grid (4 x 2 blocks; 8 x 1 for a single column) and walks it in compact
  // 8 x 8 sub-tiles (32 x 32 pixels): the gradient rows its ~256 workgroups in flight touch belong to the RoIs over a few
  // such squares instead of over a 16-pixel band across the whole map (rounds 2-4: stripes of 8 patch rows round-robin)
  const int gr = (n_img * php + 1) >> 1;
  const int nbc = bw >= 2 ? 2 : 1, nbr = 8 / nbc;
  const int BR = (gr + nbr - 1) / nbr, BC = (bw + nbc - 1) / nbc;
  const int tiles_c = (BC + 7) >> 3;
  const int t = j >> 6, within = j & 63;
  const int lr = (t / tiles_c) * 8 + (within >> 3), lc = (t % tiles_c) * 8 + (within & 7);
  if (lr >= BR || lc >= BC) return;
  const int wg_row = (xcd / nbc) * BR + lr, wg_col = (xcd % nbc) * BC + lc;
  const int prow = wg_row * 2 + (wave >> 1);                   // over all images stacked
  const int pcol = wg_col * 2 + (wave & 1);
  if (prow >= n_img * php || pcol >= pwp) return;
  const int p = prow * pwp + pcol;
  const int img = prow / php, py = prow - img * php;
  const int cnt = __builtin_amdgcn_readfirstlane(counts[p]);
  const int n_direct = min(cnt, cap), n_over = cnt - n_direct;
  const PatchEntry* __restrict__ row_d = direct + (size_t)p * cap;
  const PatchEntry* __restrict__ row_o =
      entries + (n_over > 0 ? __builtin_amdgcn_readfirstlane(row_begin(offsets, tile_base, p)) : 0);
  const int y0 = py * 2, x0 = pcol * 2;
  const bool has_y1 = y0 + 1 < img_h, has_x1 = x0 + 1 < img_w;
  float* __restrict__ out0 = grad_in + ((size_t)(img * img_h + y0) * img_w + x0) * C;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int c = c0 + lane * 4;
    const bool ok = c < C;                       // C % 4 == 0 on this path
    const float* __restrict__ col = gT + (ok ? c : 0);
    v4f acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    patch_accumulate<UNROLL>(row_d, n_direct, col, C, lane, acc);
    if (n_over > 0) patch_accumulate<UNROLL>(row_o, n_over, col, C, lane, acc);
    if (ok) {
      *reinterpret_cast<v4f*>(out0 + c) = acc[0];
      if (has_x1) *reinterpret_cast<v4f*>(out0 + C + c) = acc[1];
      if (has_y1) *reinterpret_cast<v4f*>(out0 + (size_t)img_w * C + c) = acc[2];
      if (has_y1 && has_x1) *reinterpret_cast<v4f*>(out0 + (size_t)(img_w + 1) * C + c) = acc[3];
    }
  }
  if (!KEEP && lane == 0) counts[p] = 0;
}

// counts[] hold the row lengths, direct[] the first `cap` entries of every row, segment g of recs its seg_n[g] overflow
// records, counts[nkeys + 1] their total; counts[nkeys] (ticket) is zero.
// On return (stream order) the first patch_zero_bytes(nkeys) bytes of the workspace are zero again.
// overflow records -> CSR (both launches return at once when no row went past the cap)
inline int patch_finish(const PatchWs& w, long nkeys, long nsegs, long seg_cap, hipStream_t st) {
  const int ntiles = (int)((nkeys + kScanTile - 1) / kScanTile);
  int* over_flag = w.counts + nkeys + 1;
  hipLaunchKernelGGL(csr_scan_kernel, dim3(ntiles), dim3(256), 0, st, w.counts, (int)nkeys, ntiles, w.offsets,
                     w.tile_sum, w.tile_base, w.counts + nkeys, w.cap, (const int*)over_flag);
  hipLaunchKernelGGL(csr_fill_patch_kernel, dim3((unsigned)nsegs), dim3(256), 0, st, w.recs, w.seg_n, (int)seg_cap,
                     w.offsets, w.tile_base, w.entries);
  return jdet_launch_status();
}

// keep_plan: counters, direct rows and overflow CSR are left intact (the next gather reads them again); otherwise the
// first patch_zero_bytes(nkeys) bytes of the workspace are handed back zeroed
inline int patch_gather(const PatchWs& w, long nkeys, const float* src, int C, float* dst, int n_img, int img_h,
                        int img_w, bool keep_plan, hipStream_t st) {
  const int php = (img_h + 1) / 2, pwp = (img_w + 1) / 2, bw = (pwp + 1) / 2;
  const int wg_rows = (n_img * php + 1) / 2;
  // XCD x owns one 2-D block of the workgroup grid, walked in 8 x 8 sub-tiles (see the kernel): 77.8 -> 76.3 us for the
  // whole backward against the stripes of 8 patch rows of rounds 2-4 (A/B pairs on one box, profiles/r05_roi_bwd_notes.md)
  const int nbc = bw >= 2 ? 2 : 1, nbr = 8 / nbc;
  const int BR = (wg_rows + nbr - 1) / nbr, BC = (bw + nbc - 1) / nbc;
  const unsigned blocks = 8u * 64u * (unsigned)(((BR + 7) / 8) * ((BC + 7) / 8));
  if (keep_plan)
    hipLaunchKernelGGL((csr_gather_patch_kernel<8, true>), dim3(blocks), dim3(256), 0, st, src, w.offsets, w.tile_base,
                       w.entries, w.direct, w.cap, (int)nkeys, C, n_img, img_h, img_w, w.counts, dst);
  else
    hipLaunchKernelGGL((csr_gather_patch_kernel<8, false>), dim3(blocks), dim3(256), 0, st, src, w.offsets, w.tile_base,
                       w.entries, w.direct, w.cap, (int)nkeys, C, n_img, img_h, img_w, w.counts, dst);
  return jdet_launch_status();
}

// counts[] hold the row lengths, direct[] the first `cap` entries of every row, segment g of recs its seg_n[g] overflow
// records, counts[nkeys + 1] their flag; counts[nkeys] (ticket) is zero.
// On return (stream order) the first patch_zero_bytes(nkeys) bytes of the workspace are zero again.
inline int patch_finish_and_gather(const PatchWs& w, long nkeys, long nsegs, long seg_cap, const float* src, int C,
                                   float* dst, int n_img, int img_h, int img_w, hipStream_t st) {
  const int e = patch_finish(w, nkeys, nsegs, seg_cap, st);
  if (e) return e;
  return patch_gather(w, nkeys, src, C, dst, n_img, img_h, img_w, false, st);
}

}  // namespace jdet_csr
