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
