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
//   K1  tap list, one workgroup per RoI: geometry (double-precision trig) once, then one lane per (bin, sample):
//       pixel key + weight/count of its 4 taps; integer atomicAdd into a per-pixel counter (CSR row lengths)
//       whose return value is the tap's place in its row
//   K2  exclusive scan of the N*H*W counters (one launch; the last workgroup scans the tile totals)
//   K3  fill: entry = (roi*nbins + bin, w) written at row offset + place (no second round of atomics)
//   K4  (only for an (R,C,PH,PW) gradient) grad_out -> gT (R, PH*PW, C): a contribution's channel vector becomes
//       contiguous; a channels-last gradient (jdet_roi_align_backward_cl) is that matrix already
//   K5  gather: one wave per pixel, lanes = channels (dwordx4), loop over the pixel's entries,
//       acc += w * gT[entry]; ONE coalesced store per pixel (zeros for untouched pixels, so no
//       memset pass).  fp32 adds happen in registers.  2x2 pixel patches per workgroup, 8-row stripes round-robin
//       over the XCDs (csr_gather.h); the counters are zeroed again on the way out, so a caller that keeps the
//       workspace (workspace_clean) pays no memset launch.
// Summation order inside a pixel follows slot order (integer-atomic order), i.e. it is as
// order-nondeterministic in the last bits as the reference's atomics; values agree to fp32 tolerance.
// RiRoIAlign and adaptive sampling (sample_num <= 0, unbounded samples per bin) keep the atomic path.
#include "csr_gather.h"
#include "roi_geom.h"

namespace {

using namespace jdet_roi;
using namespace jdet_csr;

// One workgroup per RoI: thread 0 does the geometry (double-precision trig included) once, then one lane per
// (bin, sample) lists its 4 taps.
template <int VARIANT>
__global__ __launch_bounds__(256) void bwd_taps_kernel(const float* __restrict__ rois, int H, int W, int PH, int PW,
                                                      float spatial_scale, int sample_num,
                                                      int* __restrict__ tap_key, int* __restrict__ tap_pos,
                                                      float* __restrict__ tap_w, int* __restrict__ counts) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  __shared__ RoiGeom s_geom;
  const int r = blockIdx.x;
  if (threadIdx.x == 0)
    s_geom = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH, PW, 1, true);
  __syncthreads();
  const RoiGeom g = s_geom;
  const int nbins = PH * PW, spb = sample_num * sample_num, S = nbins * spb;
  const int base = g.batch * H * W;
  for (int s = threadIdx.x; s < S; s += 256) {   // S % 4 == 0 when spb == 4: a quad of lanes enters or leaves together
    const int bin = s / spb, rr = s % spb;
    const long t = (long)r * S + s;
    const Sample sm = make_sample<VARIANT>(g, bin / PW, bin % PW, rr / sample_num, rr % sample_num, H, W);
    const int o[4] = {sm.o1, sm.o2, sm.o3, sm.o4};
    float w[4] = {sm.w1 / g.count, sm.w2 / g.count, sm.w3 / g.count, sm.w4 / g.count};
    bool first[4] = {true, true, true, true};
    if (spb == 4) {
      // the 4 samples of a bin are 4 consecutive lanes: taps of the bin that hit the same pixel read the same
      // grad_out row -> one entry with the summed weight (as in the forward kernel; 58 % of the taps survive on the
      // bench RoIs), fewer rows for the gather to fetch
      const int lane = threadIdx.x & 63, q = lane & 3, qbase = lane & ~3;
      const float w0[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
      for (int k = 1; k < 4; k++)
#pragma unroll
        for (int j = 0; j < k; j++)
          if (o[j] == o[k]) {
            w[j] += w0[k];
            first[k] = false;
          }
#pragma unroll
      for (int d = 1; d < 4; d++) {
        const int src = qbase | ((q + d) & 3);
        const bool earlier = ((q + d) & 3) < q;
        const int ov = __shfl(sm.valid, src, 64);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int oo = __shfl(o[j], src, 64);
          const float ww = __shfl(w0[j], src, 64);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool same = ov && oo == o[k];
            w[k] += same ? ww : 0.f;
            first[k] = first[k] && !(same && earlier);
          }
        }
      }
    }
    int4 key4, pos4;
    int* key = &key4.x;
    int* pos = &pos4.x;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      key[k] = -1;
      pos[k] = 0;
      if (sm.valid && first[k] && w[k] != 0.f && g.batch >= 0) {  // batch < 0: masked RoI
        key[k] = base + o[k];
        pos[k] = atomicAdd(&counts[key[k]], 1);
      }
    }
    reinterpret_cast<int4*>(tap_key)[t] = key4;
    reinterpret_cast<int4*>(tap_pos)[t] = pos4;
    reinterpret_cast<float4*>(tap_w)[t] = make_float4(w[0], w[1], w[2], w[3]);
  }
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

template <int VARIANT>
int run_gather(const float* grad_out, const float* rois, int R, int N, int C, int H, int W, int PH, int PW,
               float scale, int sample_num, float* grad_in, void* ws, bool grad_out_cl, bool ws_clean,
               hipStream_t st) {
  const int nbins = PH * PW, spb = sample_num * sample_num;
  const long npix = (long)N * H * W, ntaps = (long)R * nbins * spb * 4;
  CsrWs w = csr_carve(ws, npix, ntaps);
  float* gT = (float*)((char*)ws + w.bytes);
  if (!ws_clean) {   // workspace of unknown content: zero the row counters + ticket (a clean one is handed back clean)
    int he = jdet_zero_async(w.counts, csr_zero_bytes(npix), st);
    if (he) return he;
  }
  hipLaunchKernelGGL((bwd_taps_kernel<VARIANT>), dim3(R), dim3(256), 0, st, rois, H, W, PH, PW, scale, sample_num,
                     w.tap_key, w.tap_pos, w.tap_w, w.counts);
  if (grad_out_cl)   // channels-last (R, PH, PW, C) IS the (R, nbins, C) row matrix the gather wants
    return csr_finish_and_gather(w, npix, ntaps, spb * 4, grad_out, C, grad_in, W, N * H, st);
  dim3 tg(jdet_cdiv(nbins, 32), jdet_cdiv(C, 32), R);
  hipLaunchKernelGGL(bwd_transpose_kernel, tg, dim3(256), 0, st, grad_out, gT, C, nbins);
  return csr_finish_and_gather(w, npix, ntaps, spb * 4, gT, C, grad_in, W, N * H, st);
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
  return csr_carve(nullptr, npix, ntaps).bytes + align256(sizeof(float) * (size_t)R * PH * PW * C);
}

// Bytes at the start of the workspace that the gather path needs zero on entry and leaves zero on return (the row
// counters + the scan's ticket); 0 when the shape is served by the atomic path.
JDET_API size_t jdet_roi_align_backward_clean_bytes(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                                   int sample_num) {
  if (!gather_ok(variant, R, N, C, H, W, PH, PW, sample_num)) return 0;
  return csr_zero_bytes((long)N * H * W);
}

static int backward_gather(int variant, const float* grad_out, const float* rois, int R, int N, int C, int H, int W,
                           int PH, int PW, float spatial_scale, int sample_num, float* grad_in, void* workspace,
                           bool grad_out_cl, bool ws_clean, hipStream_t st) {
  switch (variant) {
    case JDET_ROI_ROTATED:
      return run_gather<JDET_ROI_ROTATED>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, grad_out_cl, ws_clean, st);
    case JDET_ROI_ROTATED_V1:
      return run_gather<JDET_ROI_ROTATED_V1>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, grad_out_cl, ws_clean, st);
    case JDET_ROI_HBB_V0:
      return run_gather<JDET_ROI_HBB_V0>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, grad_out_cl, ws_clean, st);
    default:
      return run_gather<JDET_ROI_HBB_V1>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, grad_out_cl, ws_clean, st);
  }
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
  return backward_gather(variant, grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in,
                         workspace, false, false, st);
}

JDET_API int jdet_roi_align_backward_cl(int variant, const float* grad_out_cl, const float* rois, int R, int N,
                                        int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                        float* grad_in, void* workspace, size_t workspace_bytes,
                                        int workspace_clean, jdet_stream_t stream) {
  if (variant < 0 || variant > 4 || N < 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || R < 0)
    return JDET_E_BADARG;
  const size_t need = jdet_roi_align_backward_workspace(variant, R, N, C, H, W, PH, PW, sample_num);
  if (need == 0) return JDET_E_UNSUPPORTED;   // RiRoIAlign, adaptive sampling, C % 4, R == 0: use the (R,C,PH,PW) entry
  if (!workspace || workspace_bytes < need) return JDET_E_WORKSPACE;
  if (!grad_out_cl || !rois || !grad_in) return JDET_E_BADARG;
  return backward_gather(variant, grad_out_cl, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in,
                         workspace, true, workspace_clean != 0, (hipStream_t)stream);
}
