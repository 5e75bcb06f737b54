// Head glue as a single pass (round 6, profiles/r06_glue.md).
//
// LevelPack (jdet_amd/models/utils/level_pack.py; the weight-shared towers of S2ANetHead.execute,
// python/jdet/models/roi_heads/s2anet_head.py:L207-252, run once for the small pyramid levels): the levels' maps placed
// in one canvas with zero gaps between them.  Composed from framework ops that was a zero fill of the canvas and one
// strided window copy per level (28 us a copy at the 2 x 1024^2 step: 12 launches per step forward, 20 in backward);
// here ONE kernel writes every word of the canvas once -- a level's value inside a window, zero in the gaps -- at copy
// rate.  The backward of the unpacking (level gradients -> canvas gradient) is the same operation.
// (Measured with it and removed: an in-place bias row add for the library-forward prediction layers -- neutral in the step,
//  and not bit-equal to the library's own bias handling on 1x1 layers.)
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int kPackMaxLevels = 8;

struct PackLevels {
  const float* src[kPackMaxLevels];   // (N, h, w, C) contiguous, or NULL: a window of zeros
  int h[kPackMaxLevels], w[kPackMaxLevels], r0[kPackMaxLevels], c0[kPackMaxLevels];
  int n;
};

// canvas (N, Hp, Wp, C), C % 4 == 0: one thread per (position, 4 channels)
__global__ __launch_bounds__(256) void level_pack_kernel(PackLevels lv, int N, int C4, int Hp, int Wp,
                                                         float* __restrict__ canvas) {
  const long total = (long)N * Hp * Wp * C4;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % C4);
    long p = t / C4;
    const int x = (int)(p % Wp);
    p /= Wp;
    const int y = (int)(p % Hp);
    const int n = (int)(p / Hp);
    v4f v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int l = 0; l < kPackMaxLevels; l++) {
      if (l >= lv.n) break;
      const int yy = y - lv.r0[l], xx = x - lv.c0[l];
      if (yy >= 0 && yy < lv.h[l] && xx >= 0 && xx < lv.w[l]) {
        if (lv.src[l]) v = reinterpret_cast<const v4f*>(lv.src[l])[(((long)n * lv.h[l] + yy) * lv.w[l] + xx) * C4 + c];
        break;
      }
    }
    reinterpret_cast<v4f*>(canvas)[t] = v;
  }
}

}  // namespace

JDET_API int jdet_level_pack_nhwc(const float* const* levels, const int32_t* level_hw, const int32_t* level_place,
                                  int num_levels, int N, int C, int Hp, int Wp, float* canvas, jdet_stream_t stream) {
  if (num_levels < 0 || num_levels > kPackMaxLevels || N < 0 || C <= 0 || Hp <= 0 || Wp <= 0) return JDET_E_BADARG;
  if (C % 4 != 0) return JDET_E_UNSUPPORTED;
  if (N == 0) return JDET_OK;
  if (!canvas || (num_levels > 0 && (!levels || !level_hw || !level_place))) return JDET_E_BADARG;
  PackLevels lv;
  lv.n = num_levels;
  for (int l = 0; l < kPackMaxLevels; l++) {
    const bool on = l < num_levels;
    lv.src[l] = on ? levels[l] : nullptr;
    lv.h[l] = on ? level_hw[2 * l] : 0;
    lv.w[l] = on ? level_hw[2 * l + 1] : 0;
    lv.r0[l] = on ? level_place[2 * l] : 0;
    lv.c0[l] = on ? level_place[2 * l + 1] : 0;
    if (on && (lv.h[l] <= 0 || lv.w[l] <= 0 || lv.r0[l] < 0 || lv.c0[l] < 0 || lv.r0[l] + lv.h[l] > Hp ||
               lv.c0[l] + lv.w[l] > Wp))
      return JDET_E_BADARG;
  }
  const long total = (long)N * Hp * Wp * (C / 4);
  long grid = (total + 255) / 256;
  if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(level_pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, lv, N, C / 4, Hp, Wp,
                     canvas);
  return jdet_launch_status();
}
