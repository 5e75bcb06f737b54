// Calibration probe, round 6: the row gathers of gather_probe.hip as LDS-DMA (buffer_load_dwordx4 ... lds: 64 lanes x
// 16 B = 1 KiB per wave instruction, per-lane source address, destination M0 + 16 * lane, no VGPR round trip).
// What does the texture path deliver when the rows land in LDS instead of registers, by the level that serves them,
// and does it matter whether the 1 KiB of an instruction is ONE map row (SEG 1), four 256-byte pieces of four rows
// (SEG 4: a 64-channel slice of four pixels) or eight 128-byte pieces (SEG 8)?  scripts/dma_probe.py.
#include "common.h"
#include "jdet_experimental.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ unsigned mix(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// every wave: rows_per_wave KiB in instructions of 1 KiB, UNROLL instructions in flight into its own UNROLL KiB of LDS;
// after each batch one ds_read_b128 per instruction consumes the data (READ) or nothing does.
template <int UNROLL, int SEG, int READ>
__global__ __launch_bounds__(256) void dma_probe_kernel(const float* __restrict__ buf, long total_rows, int window_rows,
                                                        int rows_per_wave, int local_windows,
                                                        float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char s_ring[];   // 4 waves x UNROLL KiB
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned wave = blockIdx.x * 4 + wv;
  const long base = local_windows ? (long)(mix(blockIdx.x * 2654435761u) % (unsigned)(total_rows - window_rows + 1)) : 0;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(buf + base * 256), 0, (int)((long)window_rows * 1024), 0x00020000);
  char* mine = s_ring + wv * UNROLL * 1024;
  constexpr int LPS = 64 / SEG;          // lanes per segment
  const int seg = lane / LPS, sl = lane % LPS;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < rows_per_wave; i += UNROLL) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      // SEG == 1: wave-uniform row; else a row per lane group, piece (i + u) % SEG of it
      const unsigned row = mix(wave * 7919u + (unsigned)((i + u) * SEG + seg)) % (unsigned)window_rows;
      const int voff = (int)row * 1024 + ((i + u) % SEG) * (1024 / SEG) + sl * 16;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(mine + u * 1024), 16, voff, 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
    if (READ) {
#pragma unroll
      for (int u = 0; u < UNROLL; u++) acc += *reinterpret_cast<const v4f*>(mine + u * 1024 + lane * 16);
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[wave * 64 + lane] = acc.x;
}

template <int UNROLL, int SEG>
int launch(int read, const float* buf, long total_rows, int window_rows, int rows_per_wave, int local_windows,
           int n_blocks, float* sink, hipStream_t st) {
  const size_t lds = 4 * UNROLL * 1024;
  if (read)
    hipLaunchKernelGGL((dma_probe_kernel<UNROLL, SEG, 1>), dim3(n_blocks), dim3(256), lds, st, buf, total_rows,
                       window_rows, rows_per_wave, local_windows, sink);
  else
    hipLaunchKernelGGL((dma_probe_kernel<UNROLL, SEG, 0>), dim3(n_blocks), dim3(256), lds, st, buf, total_rows,
                       window_rows, rows_per_wave, local_windows, sink);
  return jdet_launch_status();
}

}  // namespace

JDET_API int jdet_debug_dma_probe(const float* buf, long total_rows, int window_rows, int rows_per_wave,
                                  int local_windows, int n_blocks, int unroll, int seg, int read, float* sink,
                                  jdet_stream_t stream) {
  if (!buf || !sink || total_rows <= 0 || window_rows <= 0 || window_rows > total_rows || rows_per_wave <= 0 ||
      n_blocks <= 0 || (long)window_rows * 1024 >= (1l << 31))
    return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
#define JDET_DMA_CASE(U, S) \
  if (unroll == U && seg == S) return launch<U, S>(read, buf, total_rows, window_rows, rows_per_wave, local_windows, n_blocks, sink, st);
  JDET_DMA_CASE(4, 1) JDET_DMA_CASE(8, 1) JDET_DMA_CASE(16, 1)
  JDET_DMA_CASE(4, 4) JDET_DMA_CASE(8, 4) JDET_DMA_CASE(16, 4)
  JDET_DMA_CASE(8, 8) JDET_DMA_CASE(16, 8)
#undef JDET_DMA_CASE
  return JDET_E_UNSUPPORTED;
}
