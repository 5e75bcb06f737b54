// Calibration probe (scripts/mfma_probe.py): what one wave's fp32 MFMA stream costs per instruction once the pieces of
// the conv_wgrad.hip K loop are added one at a time.  `variant`:
//   0  bare: groups of 4 independent v_mfma_f32_32x32x2_f32
//   1  + per group two ds_read2_b32 fragment fetches, one group ahead (the wgrad loop's LDS pattern)
//   2  + per 8 groups: 4 ds_write_b128 into the other buffer and a workgroup barrier
//   3  + per 8 groups: 4 buffer loads (a 64 KiB window: L2 / L1 hits) feeding those writes
// Output: per wave, shader cycles (s_memtime) of the whole loop; the host divides by the MFMA count.
#include "../common.h"
#include "jdet_experimental.h"

namespace {
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void mfma_probe_kernel(const float* __restrict__ src, int steps, long long* __restrict__ cycles, float* __restrict__ sink) {
  constexpr int SA = 160, TILE = 16 * SA * 4;
  __shared__ __attribute__((aligned(16))) char s_raw[4 * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * TILE / 4; i += 256) reinterpret_cast<float*>(s_raw)[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 65536u, 0x00020000);
  const int fa_off = ((lane >> 5) * SA + (wave >> 1) * 64 + (lane & 31)) * 4;
  const int fb_off = TILE + ((lane >> 5) * SA + (wave & 1) * 64 + (lane & 31)) * 4;
  const int st0 = ((tid / 32) * SA + (tid % 32) * 4) * 4, st1 = st0 + 8 * SA * 4;
  v16f acc[2][2];
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
  unsigned voff = (unsigned)(tid * 16);
  v4f r[4] = {v4f{1, 2, 3, 4}, v4f{1, 2, 3, 4}, v4f{1, 2, 3, 4}, v4f{1, 2, 3, 4}};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int step = 0; step < steps; step++) {
    const int buf = step & 1;
    const char* sb = s_raw + buf * 2 * TILE;
    char* so = s_raw + (buf ^ 1) * 2 * TILE;
    if (VARIANT >= 3) {
#pragma unroll
      for (int k = 0; k < 4; k++)
        r[k] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, (voff + k * 4096u + step * 64u) & 65535u, 0, 0));
    }
    float fa[2][2], fb[2][2];
    auto frags = [&](int q) {
#pragma unroll
      for (int i = 0; i < 2; i++) fa[q & 1][i] = *reinterpret_cast<const float*>(sb + fa_off + (2 * q * SA + i * 32) * 4);
#pragma unroll
      for (int j = 0; j < 2; j++) fb[q & 1][j] = *reinterpret_cast<const float*>(sb + fb_off + (2 * q * SA + j * 32) * 4);
    };
    if (VARIANT >= 1) frags(0);
    else {
      fa[0][0] = fa[0][1] = fa[1][0] = fa[1][1] = 1.f + lane;
      fb[0][0] = fb[0][1] = fb[1][0] = fb[1][1] = 2.f;
    }
#pragma unroll
    for (int q = 0; q < 8; q++) {
      if (VARIANT >= 1 && q + 1 < 8) frags(q + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[q & 1][i], fb[q & 1][j], acc[i][j], 0, 0, 0);
      if (VARIANT >= 2 && q == 6) {
        *reinterpret_cast<v4f*>(so + st0) = r[0];
        *reinterpret_cast<v4f*>(so + st1) = r[1];
        *reinterpret_cast<v4f*>(so + TILE + st0) = r[2];
        *reinterpret_cast<v4f*>(so + TILE + st1) = r[3];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (VARIANT >= 2) __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 2; j++)
      for (int e = 0; e < 16; e++) s += acc[i][j][e];
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  if (s == 12345.f) sink[0] = s;
}
// Round 6: the 64 x 64-tile kernels' inner structure (one 32 x 32 tile per wave, conv_bn.hip) against the 128-tile one, MFMAs and
// LDS reads only (16 MFMAs per step either way):
//   4  one accumulator, 8 ds_read_b128 per step (4 A + 4 B, each feeding 4 MFMAs)      -- the 64-tile step
//   5  four accumulators (2 x 2 tiles), 4 ds_read_b128 per step (2 A + 2 B)              -- the 128-tile step
//   6  one accumulator, operands in registers (the bare dependent chain)
//   7  two accumulators (even / odd slices), 8 ds_read_b128 per step
template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4)))
void mfma_probe2_kernel(const float* __restrict__ src, int steps, long long* __restrict__ cycles, float* __restrict__ sink) {
  __shared__ __attribute__((aligned(16))) char s_raw[32768];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // operands from `src` (random data: the matrix pipe's power, hence the clock under the power cap, depends on the bits it toggles)
  for (int i = tid; i < 32768 / 4; i += 256) reinterpret_cast<float*>(s_raw)[i] = src[(i + blockIdx.x * 64) & 16383];
  __syncthreads();
  // conv_bn's XOR-swizzled 16-byte chunks: row = lane & 31 (+ 32 * wave half), chunk = 2 * q + (lane >> 5)
  auto off = [&](int row, int chunk) { return (row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2)) * 4; };
  int fa_off[4], fb_off[4];
  for (int q = 0; q < 4; q++) {
    fa_off[q] = off((wave >> 1) * 32 + (lane & 31), 2 * q + (lane >> 5));
    fb_off[q] = 8192 + off((wave & 1) * 32 + (lane & 31), 2 * q + (lane >> 5));
  }
  v16f acc[4];
  for (int i = 0; i < 4; i++)
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int step = 0; step < steps; step++) {
    const char* sb = s_raw + (step & 1) * 16384;
    if constexpr (VARIANT == 4 || VARIANT == 7) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const v4f fa = *reinterpret_cast<const v4f*>(sb + fa_off[q]);
        const v4f fb = *reinterpret_cast<const v4f*>(sb + fb_off[q]);
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const int a = VARIANT == 7 ? (kk & 1) : 0;
          acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc[a], 0, 0, 0);
        }
      }
    } else if constexpr (VARIANT == 5) {
      v4f fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        fa[i] = *reinterpret_cast<const v4f*>(sb + fa_off[i]);
        fb[i] = *reinterpret_cast<const v4f*>(sb + fb_off[i]);
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
            acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i * 2 + j], 0, 0, 0);
    } else {
      const float fa = src[lane], fb = src[64 + lane] + step;
#pragma unroll
      for (int kk = 0; kk < 16; kk++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[0], 0, 0, 0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; i++)
    for (int e = 0; e < 16; e++) s += acc[i][e];
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
  if (s == 12345.f) sink[0] = s;
}
}  // namespace

// cycles: n_blocks * 4 entries (one per wave); src: at least 64 KiB of floats
JDET_API int jdet_debug_mfma_probe(int variant, const float* src, int n_blocks, int steps, long long* cycles, float* sink,
                                   jdet_stream_t stream) {
  if (n_blocks <= 0 || steps <= 0 || !src || !cycles || !sink) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case 0: hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 1: hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 2: hipLaunchKernelGGL(mfma_probe_kernel<2>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 3: hipLaunchKernelGGL(mfma_probe_kernel<3>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 4: hipLaunchKernelGGL(mfma_probe2_kernel<4>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 5: hipLaunchKernelGGL(mfma_probe2_kernel<5>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 6: hipLaunchKernelGGL(mfma_probe2_kernel<6>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    case 7: hipLaunchKernelGGL(mfma_probe2_kernel<7>, dim3(n_blocks), dim3(256), 0, st, src, steps, cycles, sink); break;
    default: return JDET_E_BADARG;
  }
  return jdet_launch_status();
}
