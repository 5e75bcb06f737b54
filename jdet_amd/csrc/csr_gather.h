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

// exclusive scan of the n = N*H*W pixel counters in two launches: (a) every workgroup scans its own
// 2048-element tile and publishes the tile total, (b) every workgroup adds the totals of the tiles
// before it (<= a few hundred values, summed redundantly per workgroup) and zeroes the cursors.
constexpr int kScanTile = 2048;  // 256 threads x 8

static __global__ __launch_bounds__(256) void csr_scan_local_kernel(const int* __restrict__ counts, int n,
                                                            int* __restrict__ offsets, int* __restrict__ tile_sum) {
  __shared__ int s_wave[4];
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
  if (threadIdx.x == 255) tile_sum[blockIdx.x] = run;
}

static __global__ __launch_bounds__(256) void csr_scan_add_kernel(int n, int ntiles, const int* __restrict__ tile_sum,
                                                          int* __restrict__ offsets) {
  __shared__ int s_part[4];
  int part = 0;
  for (int t = threadIdx.x; t < (int)blockIdx.x; t += 256) part += tile_sum[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
  if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
  __syncthreads();
  const int base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
  const int lo = blockIdx.x * kScanTile + threadIdx.x * 8;
#pragma unroll
  for (int i = 0; i < 8; i++)
    if (lo + i < n) offsets[lo + i] += base;
  if (blockIdx.x == ntiles - 1 && threadIdx.x == 0) {
    // total = base + this tile's total -> offsets[n]
    offsets[n] = base + tile_sum[ntiles - 1];
  }
}

// tap_pos[e] = the value the tap's atomicAdd on counts[key] returned when the row lengths were counted: its place
// inside the row.  No second round of atomics here.
static __global__ __launch_bounds__(256) void csr_fill_kernel(const int* __restrict__ tap_key,
                                                      const int* __restrict__ tap_pos,
                                                      const float* __restrict__ tap_w, long ntaps, int spb4,
                                                      const int* __restrict__ offsets, Entry* __restrict__ entries) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= ntaps) return;
  const int key = tap_key[e];
  if (key < 0) return;
  const int pos = offsets[key] + tap_pos[e];
  Entry en;
  en.src = (int)(e / spb4);  // (roi * nbins + bin): taps are ordered roi, bin, sample, tap
  en.w = tap_w[e];
  entries[pos] = en;
}

// one wave per pixel; lane owns 4 consecutive channels of a 256-channel chunk
template <int UNROLL>
static __global__ __launch_bounds__(256) void csr_gather_kernel(const float* __restrict__ gT,
                                                        const int* __restrict__ offsets,
                                                        const Entry* __restrict__ entries, int npix, int C,
                                                        float* __restrict__ grad_in) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int p = blockIdx.x * 4 + wave;
  if (p >= npix) return;
  const int beg = __builtin_amdgcn_readfirstlane(offsets[p]);
  const int end = __builtin_amdgcn_readfirstlane(offsets[p + 1]);
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
  int* counts;
  int* offsets;
  int* tile_sum;
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
  w.counts = (int*)(p + off);   off += align256(sizeof(int) * nkeys);
  w.offsets = (int*)(p + off);  off += align256(sizeof(int) * (nkeys + 1));
  w.tile_sum = (int*)(p + off); off += align256(sizeof(int) * ((nkeys + kScanTile - 1) / kScanTile + 1));
  w.tap_key = (int*)(p + off);  off += align256(sizeof(int) * ntaps);
  w.tap_pos = (int*)(p + off);  off += align256(sizeof(int) * ntaps);
  w.tap_w = (float*)(p + off);  off += align256(sizeof(float) * ntaps);
  w.entries = (Entry*)(p + off); off += align256(sizeof(Entry) * ntaps);
  w.bytes = off;
  return w;
}

// counts[] must already hold the row lengths and tap_key / tap_pos / tap_w the taps (key < 0 = dropped tap;
// tap_pos = return value of the atomicAdd that counted it).
// taps_per_src consecutive taps share one source row (entry.src = tap index / taps_per_src).
inline int csr_finish_and_gather(const CsrWs& w, long nkeys, long ntaps, int taps_per_src, const float* src, int C,
                                 float* dst, hipStream_t st) {
  const int ntiles = (int)((nkeys + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(csr_scan_local_kernel, dim3(ntiles), dim3(256), 0, st, w.counts, (int)nkeys, w.offsets, w.tile_sum);
  hipLaunchKernelGGL(csr_scan_add_kernel, dim3(ntiles), dim3(256), 0, st, (int)nkeys, ntiles, w.tile_sum, w.offsets);
  hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)((ntaps + 255) / 256)), dim3(256), 0, st, w.tap_key, w.tap_pos,
                     w.tap_w, ntaps, taps_per_src, w.offsets, w.entries);
  hipLaunchKernelGGL((csr_gather_kernel<4>), dim3((unsigned)((nkeys + 3) / 4)), dim3(256), 0, st, src, w.offsets,
                     w.entries, (int)nkeys, C, dst);
  return jdet_launch_status();
}

}  // namespace jdet_csr
