// Calibration probe for the gather-shaped kernels (RoIAlign forward taps, the backward gather, deformable sampling):
// what does this chip deliver when every wave fetches whole 1 KiB rows (64 lanes x 16 B, one buffer_load_dwordx4)
// from pseudo-random places of a buffer?  The window the rows are drawn from selects the level that serves them:
// a few rows -> vector L1, ~1 MiB -> L2, the whole 64 MiB map -> L2 misses (Infinity Cache / HBM).  The RoIAlign
// kernels are compared with these rates in DESIGN.md 3.1 -- the HBM stream roofline (8 TB/s) is not the one a
// row gather can reach.  Not on any product path (scripts/gather_probe.py).
#include "common.h"
#include "jdet_experimental.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {   // integer hash: the row sequence of a wave
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// rows_per_wave loads of 1 KiB each, UNROLL in flight; row index = hash(wave, i) % window_rows + base(wave)
template <int UNROLL>
__global__ __launch_bounds__(256) void gather_probe_kernel(const float* __restrict__ buf, long total_rows,
                                                           int window_rows, int rows_per_wave, int local_windows,
                                                           float* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  // local_windows: every workgroup draws from its own window (placed by a hash of the workgroup id) -- the RoIAlign
  // situation, a RoI's pixels; otherwise all waves share window 0
  const long base = local_windows ? (long)(mix(blockIdx.x * 2654435761u) % (unsigned)(total_rows - window_rows + 1)) : 0;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < rows_per_wave; i += UNROLL) {
    v4f v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const long row = base + (long)(mix(wave * 7919u + (unsigned)(i + u)) % (unsigned)window_rows);
      v[u] = *reinterpret_cast<const v4f*>(buf + row * 256 + lane * 4);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[wave * 64 + lane] = acc.x;   // keeps the loads alive
}

// The same 1 KiB row fetched as 4 / 2 / 1 load instructions of 4 / 8 / 16 bytes per lane (dword / dwordx2 / dwordx4):
// does the texture path deliver more bytes per clock with narrower loads?
template <int W>
__global__ __launch_bounds__(256) void gather_width_probe_kernel(const float* __restrict__ buf, long total_rows,
                                                                 int window_rows, int rows_per_wave,
                                                                 float* __restrict__ sink) {
  typedef float vw __attribute__((ext_vector_type(W)));
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  float acc = 0.f;
  for (int i = 0; i < rows_per_wave; i += 4) {
    vw v[4][4 / W];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long row = (long)(mix(wave * 7919u + (unsigned)(i + u)) % (unsigned)window_rows);
#pragma unroll
      for (int j = 0; j < 4 / W; j++)
        v[u][j] = *reinterpret_cast<const vw*>(buf + row * 256 + j * 64 * W + lane * W);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int j = 0; j < 4 / W; j++)
        acc += v[u][j][0];
  }
  if (acc == 12345.678f) sink[wave * 64 + lane] = acc;
}

// The main loop of a pixel-stationary RoIAlign, emulated: every loaded row is added `pairs` times into an LDS
// accumulator block [49 bins][256 channels] with ds_add_f32 (channel 4*lane+k of a bin lives at k*64+lane: conflict
// free), and at the end the block (50 KB) is streamed to `out` with non-temporal stores -- loads through the TA, LDS
// atomics and the output stream together, without any of the per-RoI set-up.
__global__ __launch_bounds__(256) void gather_accumulate_probe_kernel(const float* __restrict__ buf, long total_rows,
                                                                      int window_rows, int rows_per_wave, int pairs,
                                                                      float* __restrict__ out) {
  extern __shared__ float s_acc[];   // [49][256]
  const int lane = threadIdx.x & 63;
  const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const long base = (long)(mix(blockIdx.x * 2654435761u) % (unsigned)(total_rows - window_rows + 1));
  for (int i = threadIdx.x; i < 49 * 256; i += 256) s_acc[i] = 0.f;
  __syncthreads();
  for (int i = 0; i < rows_per_wave; i += 16) {
    v4f v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const long row = base + (long)(mix(wave * 7919u + (unsigned)(i + u)) % (unsigned)window_rows);
      v[u] = *reinterpret_cast<const v4f*>(buf + row * 256 + lane * 4);
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      if (i + u >= rows_per_wave) break;
      for (int p = 0; p < pairs; p++) {
        const int bin = (int)(mix(wave * 31u + (unsigned)((i + u) * 4 + p)) % 49u);   // wave-uniform
        float* a = s_acc + bin * 256 + lane;
        const float w = 0.25f + 0.125f * p;
        unsafeAtomicAdd(a, w * v[u].x);
        unsafeAtomicAdd(a + 64, w * v[u].y);
        unsafeAtomicAdd(a + 128, w * v[u].z);
        unsafeAtomicAdd(a + 192, w * v[u].w);
      }
    }
  }
  __syncthreads();
  float* __restrict__ dst = out + (size_t)blockIdx.x * 49 * 256;
  for (int b = threadIdx.x >> 6; b < 49; b += 4) {
    const float* a = s_acc + b * 256 + lane;
    const v4f o = {a[0], a[64], a[128], a[192]};
    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(dst + b * 256 + lane * 4));
  }
}

}  // namespace

JDET_API int jdet_debug_gather_accumulate_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                                int pairs, int n_blocks, float* out, jdet_stream_t stream) {
  if (!buf || !out || total_rows <= 0 || window_rows <= 0 || window_rows > total_rows || rows_per_wave <= 0 ||
      n_blocks <= 0 || pairs < 0)
    return JDET_E_BADARG;
  hipLaunchKernelGGL(gather_accumulate_probe_kernel, dim3(n_blocks), dim3(256), 49 * 256 * sizeof(float),
                     (hipStream_t)stream, buf, total_rows, window_rows, rows_per_wave, pairs, out);
  return jdet_launch_status();
}

JDET_API int jdet_debug_gather_width_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                           int dwords_per_lane, int n_blocks, float* sink, jdet_stream_t stream) {
  if (!buf || !sink || total_rows <= 0 || window_rows <= 0 || window_rows > total_rows || rows_per_wave <= 0 ||
      n_blocks <= 0)
    return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (dwords_per_lane == 1)
    hipLaunchKernelGGL(gather_width_probe_kernel<1>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, sink);
  else if (dwords_per_lane == 2)
    hipLaunchKernelGGL(gather_width_probe_kernel<2>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, sink);
  else
    hipLaunchKernelGGL(gather_width_probe_kernel<4>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, sink);
  return jdet_launch_status();
}

// buf: total_rows x 256 floats.  n_blocks workgroups of 4 waves, rows_per_wave row loads per wave.
JDET_API int jdet_debug_gather_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                     int local_windows, int n_blocks, int unroll, float* sink, jdet_stream_t stream) {
  if (!buf || !sink || total_rows <= 0 || window_rows <= 0 || window_rows > total_rows || rows_per_wave <= 0 ||
      n_blocks <= 0)
    return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  if (unroll == 16)
    hipLaunchKernelGGL(gather_probe_kernel<16>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, local_windows, sink);
  else if (unroll == 8)
    hipLaunchKernelGGL(gather_probe_kernel<8>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, local_windows, sink);
  else
    hipLaunchKernelGGL(gather_probe_kernel<4>, dim3(n_blocks), dim3(256), 0, st, buf, total_rows, window_rows,
                       rows_per_wave, local_windows, sink);
  return jdet_launch_status();
}
