// Device side of the input pipeline: the batch crosses PCIe as the uint8 pixels the decoder produced
// (3 bytes per pixel instead of the 12 of a normalised float image) and is normalised, channel-swapped, zero-padded
// to the batch size and laid out channels-last on the device in one pass.
// Reference arithmetic (python/jdet/data/transforms.py: `Normalize` L467-487 -- image[::-1] when to_bgr, then
// (image - mean) / std in float32 -- and the zero padding of `collate_batch`, data/custom.py:L90-106): the same two
// float32 operations per element, so the result is bit-identical to the host path.
#include "common.h"

namespace {

struct Norm3 {
  float mean[3], std[3];
};

// src (N, Hs, Ws, 3) uint8, the image of sample n occupies rows < valid_hw[2n], columns < valid_hw[2n+1];
// dst (N, Hs, Ws, 3) float32 = the channels-last memory of a logical (N, 3, Hs, Ws) tensor.
__global__ __launch_bounds__(256) void normalize_u8_kernel(const uint8_t* __restrict__ src,
                                                           const int32_t* __restrict__ valid_hw, int N, int Hs, int Ws,
                                                           Norm3 p, int swap_rb, float* __restrict__ dst) {
  const long total = (long)N * Hs * Ws;
  for (long px = (long)blockIdx.x * 256 + threadIdx.x; px < total; px += (long)gridDim.x * 256) {
    const int x = (int)(px % Ws);
    const int y = (int)((px / Ws) % Hs);
    const int n = (int)(px / ((long)Ws * Hs));
    const bool in = y < valid_hw[2 * n] && x < valid_hw[2 * n + 1];
    const uint8_t* s = src + px * 3;
    float* d = dst + px * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float v = (float)s[swap_rb ? 2 - c : c];
      d[c] = in ? (v - p.mean[c]) / p.std[c] : 0.f;
    }
  }
}

}  // namespace

// mean3 / std3: HOST float[3], indexed by OUTPUT channel (i.e. after the optional channel reversal, as the reference
// applies them).  valid_hw: device int32 (N, 2) [height, width] of every image inside the (Hs, Ws) canvas.
JDET_API int jdet_normalize_u8_nhwc(const uint8_t* src_nhwc, const int32_t* valid_hw, int N, int Hs, int Ws,
                                    const float* mean3, const float* std3, int swap_rb, float* dst_nhwc,
                                    jdet_stream_t stream) {
  if (N < 0 || Hs <= 0 || Ws <= 0 || !mean3 || !std3) return JDET_E_BADARG;
  if (N == 0) return JDET_OK;
  if (!src_nhwc || !valid_hw || !dst_nhwc) return JDET_E_BADARG;
  Norm3 p;
  for (int c = 0; c < 3; c++) {
    p.mean[c] = mean3[c];
    p.std[c] = std3[c];
  }
  const long total = (long)N * Hs * Ws;
  long blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(normalize_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src_nhwc,
                     valid_hw, N, Hs, Ws, p, swap_rb ? 1 : 0, dst_nhwc);
  return jdet_launch_status();
}
