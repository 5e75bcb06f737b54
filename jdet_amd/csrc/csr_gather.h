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

static __global__ __launch_bounds__(256) void csr_scan_kernel(const int* __restrict__ counts, int n, int ntiles,
                                                              int* __restrict__ offsets, int* __restrict__ tile_sum,
                                                              int* __restrict__ tile_base, int* __restrict__ ticket) {
  __shared__ int s_wave[4];
  __shared__ int s_last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = blockIdx.x * kScanTile + threadIdx.x * 8;
  int v[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    v[i] = lo + i < n ? counts[lo + i] : 0;
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

}  // namespace jdet_csr
