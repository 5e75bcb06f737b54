// Shared helpers for the gfx950 kernels of libjdet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "jdet_hip.h"

#define JDET_API extern "C" __attribute__((visibility("default")))

// wave64 everywhere on CDNA4
#define JDET_WAVE 64

static inline int jdet_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? JDET_OK : (int)e;
}

static inline int jdet_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float jdet_readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int jdet_readlane_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}

// Zero fill as an ordinary kernel.  hipMemsetAsync is avoided on purpose: these libraries are replayed from
// HIP graphs (Runner graph mode) and a plain kernel node is the one thing every capture handles identically.
// `p` must be 4-byte aligned and `bytes` a multiple of 4 (all callers: fp32 / int32 arrays).
static __global__ __launch_bounds__(256) void jdet_zero_kernel(uint32_t* __restrict__ p, size_t nwords) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nwords; i += stride) {
    if (i + 4 <= nwords && (((uintptr_t)(p + i)) & 15) == 0) {
      *reinterpret_cast<uint4*>(p + i) = make_uint4(0u, 0u, 0u, 0u);
    } else {
      for (size_t k = i; k < nwords && k < i + 4; k++) p[k] = 0u;
    }
  }
}

static inline int jdet_zero_async(void* p, size_t bytes, hipStream_t st) {
  if (bytes == 0) return JDET_OK;
  const size_t nwords = bytes / 4;
  size_t blocks = (nwords + 1023) / 1024;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(jdet_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (uint32_t*)p, nwords);
  return jdet_launch_status();
}
