// RoIAlign backward as a sorted gather (no floating-point atomics).
//
// Reference: ROIAlignBackward (roi_align_rotated.py:L165-255 and the _v1 / horizontal twins) does
// 4 global atomicAdd per (output element, sample): 401 M fp32 atomics at the north-star point
// (2000 RoIs x 256 ch x 49 bins x 4 samples x 4 taps); the first HIP version of that scatter ran
// 1.26 ms, bound by L2 atomic throughput.
//
// Every contribution is  grad_in[pixel, c] += w * grad_out[roi, c, bin]  with (pixel, w) independent
// of c.  So the scatter is inverted once per launch on the (roi, sample, tap) index space -- 1.57 M
// entries instead of 401 M atomics -- and then GATHERED:
//   K1  tap list: one lane per (roi, sample): pixel key + weight/count of its 4 taps; integer
//       atomicAdd into a per-pixel counter (CSR row lengths)
//   K2  exclusive scan of the N*H*W counters (one workgroup)
//   K3  fill: every tap takes a slot in its pixel's row: entry = (roi*nbins + bin, w)
//   K4  grad_out (R,C,PH,PW) -> gT (R, PH*PW, C): makes a contribution's channel vector contiguous
//   K5  gather: one wave per pixel, lanes = channels (dwordx4), loop over the pixel's entries,
//       acc += w * gT[entry]; ONE coalesced store per pixel (zeros for untouched pixels, so no
//       memset pass).  fp32 adds happen in registers.
// Summation order inside a pixel follows slot order (integer-atomic order), i.e. it is as
// order-nondeterministic in the last bits as the reference's atomics; values agree to fp32 tolerance.
// RiRoIAlign and adaptive sampling (sample_num <= 0, unbounded samples per bin) keep the atomic path.
#include "roi_geom.h"

namespace {

using namespace jdet_roi;

typedef float v4f __attribute__((ext_vector_type(4)));

struct Entry {
  int src;    // roi * nbins + bin
  float w;    // bilinear weight / count
};

template <int VARIANT>
__global__ __launch_bounds__(256) void bwd_taps_kernel(const float* __restrict__ rois, int R, int H, int W,
                                                      int PH, int PW, float spatial_scale, int sample_num,
                                                      int* __restrict__ tap_key, float* __restrict__ tap_w,
                                                      int* __restrict__ counts) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  const int nbins = PH * PW, spb = sample_num * sample_num, S = nbins * spb;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)R * S) return;
  const int r = (int)(t / S), s = (int)(t % S);
  const int bin = s / spb, rr = s % spb;
  const RoiGeom g = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH, PW, 1, true);
  const Sample sm = make_sample<VARIANT>(g, bin / PW, bin % PW, rr / sample_num, rr % sample_num, H, W);
  const int o[4] = {sm.o1, sm.o2, sm.o3, sm.o4};
  const float w[4] = {sm.w1 / g.count, sm.w2 / g.count, sm.w3 / g.count, sm.w4 / g.count};
  const int base = g.batch * H * W;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int key = -1;
    if (sm.valid && w[k] != 0.f && g.batch >= 0) {  // batch < 0: masked RoI
      key = base + o[k];
      atomicAdd(&counts[key], 1);
    }
    tap_key[t * 4 + k] = key;
    tap_w[t * 4 + k] = w[k];
  }
}

// exclusive scan of the n = N*H*W pixel counters in two launches: (a) every workgroup scans its own
// 2048-element tile and publishes the tile total, (b) every workgroup adds the totals of the tiles
// before it (<= a few hundred values, summed redundantly per workgroup) and zeroes the cursors.
constexpr int kScanTile = 2048;  // 256 threads x 8

__global__ __launch_bounds__(256) void bwd_scan_local_kernel(const int* __restrict__ counts, int n,
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

__global__ __launch_bounds__(256) void bwd_scan_add_kernel(int n, int ntiles, const int* __restrict__ tile_sum,
                                                          int* __restrict__ offsets, int* __restrict__ cursor) {
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
    if (lo + i < n) {
      offsets[lo + i] += base;
      cursor[lo + i] = 0;
    }
  if (blockIdx.x == ntiles - 1 && threadIdx.x == 0) {
    // total = base + this tile's total -> offsets[n]
    offsets[n] = base + tile_sum[ntiles - 1];
  }
}

__global__ __launch_bounds__(256) void bwd_fill_kernel(const int* __restrict__ tap_key,
                                                      const float* __restrict__ tap_w, long ntaps, int spb4,
                                                      const int* __restrict__ offsets, int* __restrict__ cursor,
                                                      Entry* __restrict__ entries) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= ntaps) return;
  const int key = tap_key[e];
  if (key < 0) return;
  const int pos = offsets[key] + atomicAdd(&cursor[key], 1);
  Entry en;
  en.src = (int)(e / spb4);  // (roi * nbins + bin): taps are ordered roi, bin, sample, tap
  en.w = tap_w[e];
  entries[pos] = en;
}

// (R, C, nbins) -> (R, nbins, C), 32x32 LDS tiles
__global__ __launch_bounds__(256) void bwd_transpose_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           int rows, int cols) {
  __shared__ float tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int rr = r0 + ty + i, cc = c0 + tx;
    if (rr < rows && cc < cols) tile[ty + i][tx] = x[base + (size_t)rr * cols + cc];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 32; i += 8) {
    const int cc = c0 + ty + i, rr = r0 + tx;
    if (rr < rows && cc < cols) y[base + (size_t)cc * rows + rr] = tile[tx][ty + i];
  }
}

// one wave per pixel; lane owns 4 consecutive channels of a 256-channel chunk
template <int UNROLL>
__global__ __launch_bounds__(256) void bwd_gather_kernel(const float* __restrict__ gT,
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

struct BwdWs {
  int* counts;
  int* offsets;
  int* cursor;
  int* tile_sum;
  int* tap_key;
  float* tap_w;
  Entry* entries;
  float* gT;
  size_t bytes;
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

BwdWs carve(void* ws, long npix, long ntaps, long gT_floats) {
  BwdWs w;
  char* p = (char*)ws;
  size_t off = 0;
  w.counts = (int*)(p + off);  off += align256(sizeof(int) * npix);
  w.offsets = (int*)(p + off); off += align256(sizeof(int) * (npix + 1));
  w.cursor = (int*)(p + off);  off += align256(sizeof(int) * npix);
  w.tile_sum = (int*)(p + off); off += align256(sizeof(int) * ((npix + kScanTile - 1) / kScanTile + 1));
  w.tap_key = (int*)(p + off); off += align256(sizeof(int) * ntaps);
  w.tap_w = (float*)(p + off); off += align256(sizeof(float) * ntaps);
  w.entries = (Entry*)(p + off); off += align256(sizeof(Entry) * ntaps);
  w.gT = (float*)(p + off);    off += align256(sizeof(float) * gT_floats);
  w.bytes = off;
  return w;
}

template <int VARIANT>
int run_gather(const float* grad_out, const float* rois, int R, int N, int C, int H, int W, int PH, int PW,
               float scale, int sample_num, float* grad_in, void* ws, hipStream_t st) {
  const int nbins = PH * PW, spb = sample_num * sample_num;
  const long npix = (long)N * H * W, ntaps = (long)R * nbins * spb * 4;
  BwdWs w = carve(ws, npix, ntaps, (long)R * nbins * C);
  hipError_t he = hipMemsetAsync(w.counts, 0, sizeof(int) * npix, st);
  if (he != hipSuccess) return (int)he;
  const long nsamp = (long)R * nbins * spb;
  hipLaunchKernelGGL((bwd_taps_kernel<VARIANT>), dim3((unsigned)((nsamp + 255) / 256)), dim3(256), 0, st, rois, R,
                     H, W, PH, PW, scale, sample_num, w.tap_key, w.tap_w, w.counts);
  const int ntiles = (int)((npix + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(bwd_scan_local_kernel, dim3(ntiles), dim3(256), 0, st, w.counts, (int)npix, w.offsets, w.tile_sum);
  hipLaunchKernelGGL(bwd_scan_add_kernel, dim3(ntiles), dim3(256), 0, st, (int)npix, ntiles, w.tile_sum, w.offsets,
                     w.cursor);
  hipLaunchKernelGGL(bwd_fill_kernel, dim3((unsigned)((ntaps + 255) / 256)), dim3(256), 0, st, w.tap_key, w.tap_w,
                     ntaps, spb * 4, w.offsets, w.cursor, w.entries);
  dim3 tg(jdet_cdiv(nbins, 32), jdet_cdiv(C, 32), R);
  hipLaunchKernelGGL(bwd_transpose_kernel, tg, dim3(256), 0, st, grad_out, w.gT, C, nbins);
  hipLaunchKernelGGL((bwd_gather_kernel<4>), dim3((unsigned)((npix + 3) / 4)), dim3(256), 0, st, w.gT, w.offsets,
                     w.entries, (int)npix, C, grad_in);
  return jdet_launch_status();
}

}  // namespace

// Defined in roi_align.hip: the atomic scatter path (RiRoI, adaptive sampling, odd channel counts).
int jdet_roi_align_backward_atomic(int variant, const float* grad_out, const float* rois, int R, int N, int C,
                                   int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                   int n_orient, const int32_t* order, float* grad_in, hipStream_t st);

static bool gather_ok(int variant, int R, int N, int C, int H, int W, int PH, int PW, int sample_num) {
  if (variant == JDET_ROI_RIROI || sample_num <= 0 || C % 4 != 0 || R <= 0) return false;
  const long npix = (long)N * H * W, ntaps = (long)R * PH * PW * sample_num * sample_num * 4;
  if (npix >= (1L << 30) || ntaps >= (1L << 31) || (long)R * PH * PW >= (1L << 31) || R > 65535) return false;
  return true;
}

JDET_API size_t jdet_roi_align_backward_workspace(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                                 int sample_num) {
  if (!gather_ok(variant, R, N, C, H, W, PH, PW, sample_num)) return 0;
  const long npix = (long)N * H * W, ntaps = (long)R * PH * PW * sample_num * sample_num * 4;
  return carve(nullptr, npix, ntaps, (long)R * PH * PW * C).bytes;
}

JDET_API int jdet_roi_align_backward(int variant, const float* grad_out, const float* rois, int R, int N,
                                     int C, int H, int W, int PH, int PW, float spatial_scale,
                                     int sample_num, int n_orient, const int32_t* order, float* grad_in,
                                     void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  hipStream_t st = (hipStream_t)stream;
  const size_t need = jdet_roi_align_backward_workspace(variant, R, N, C, H, W, PH, PW, sample_num);
  if (need == 0 || workspace == nullptr)
    return jdet_roi_align_backward_atomic(variant, grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale,
                                          sample_num, n_orient, order, grad_in, st);
  if (workspace_bytes < need) return JDET_E_WORKSPACE;
  if (variant < 0 || variant > 4 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || !grad_out ||
      !rois || !grad_in)
    return JDET_E_BADARG;
  switch (variant) {
    case JDET_ROI_ROTATED:
      return run_gather<JDET_ROI_ROTATED>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, st);
    case JDET_ROI_ROTATED_V1:
      return run_gather<JDET_ROI_ROTATED_V1>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, st);
    case JDET_ROI_HBB_V0:
      return run_gather<JDET_ROI_HBB_V0>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, st);
    default:
      return run_gather<JDET_ROI_HBB_V1>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, st);
  }
}
