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
//   K1  entries, one workgroup per RoI: geometry (double-precision trig) once, then one lane per (bin, sample): the
//       taps of a bin that fall into one aligned 2x2 pixel PATCH merge into one entry (source row, 4 weights;
//       csr_gather.h); an integer atomicAdd on the patch's counter returns the entry's place in the patch's row; the
//       first `cap` entries of a row are written straight to their home (direct rows), the rest become records
//   K2  (rows past the cap only; returns at once otherwise) exclusive scan of the overflow lengths
//   K3  (ditto) fill: overflow record -> CSR entry at row offset + place (no second round of atomics)
//   K4  (only for an (R,C,PH,PW) gradient) grad_out -> gT (R, PH*PW, C): a contribution's channel vector becomes
//       contiguous; a channels-last gradient (jdet_roi_align_backward_cl) is that matrix already
//   K5  gather: one wave per patch, lanes = channels (dwordx4), loop over the patch's entries (direct row, then CSR),
//       acc[pixel] += w[pixel] * gT[entry]; ONE coalesced store per pixel (zeros for untouched pixels, so no memset
//       pass).  fp32 adds happen in registers.  2x2 patches per workgroup, every XCD owns one 2-D block of the map
//       (csr_gather.h); the counters are zeroed again on the way out, so a caller that keeps the workspace
//       (workspace_clean) pays no memset launch.
// Summation order inside a pixel follows slot order (integer-atomic order), i.e. it is as
// order-nondeterministic in the last bits as the reference's atomics; values agree to fp32 tolerance.
// RiRoIAlign runs the same gather on orientation-mixed gradient rows (riroi_mix_rows_kernel); adaptive sampling
// (sample_num <= 0, unbounded samples per bin) keeps the atomic path.
#include "csr_gather.h"
#include "roi_geom.h"

namespace {

using namespace jdet_roi;
using namespace jdet_csr;

long patch_keys(int N, int H, int W) { return (long)N * ((H + 1) / 2) * ((W + 1) / 2); }


// One merged entry (source row, the four weights of a patch's pixels): counted in its patch's row (integer atomic, the
// return value is its place) and written to its home in the patch's direct row, or -- past the cap -- as a record of the
// producer's own segment (compacted through the LDS counter) for the scan + fill of the overflow CSR.
__device__ __forceinline__ void emit_entry(int key, int src_row, float w0, float w1, float w2, float w3,
                                           int* __restrict__ counts, PatchEntry* __restrict__ direct, int cap,
                                           TapRec* __restrict__ seg, int* s_n) {
  const int pos = atomicAdd(&counts[key], 1);
  if (pos < cap) {
    int4* dst = reinterpret_cast<int4*>(direct + (size_t)key * cap + pos);
    dst[0] = make_int4(src_row, __float_as_int(w0), __float_as_int(w1), __float_as_int(w2));
    dst[1] = make_int4(__float_as_int(w3), 0, 0, 0);
  } else {
    const int local = atomicAdd(s_n, 1);
    int4* dst = reinterpret_cast<int4*>(seg + local);
    dst[0] = make_int4(key, pos - cap, src_row, __float_as_int(w0));
    dst[1] = make_int4(__float_as_int(w1), __float_as_int(w2), __float_as_int(w3), 0);
  }
}

// One workgroup per RoI: thread 0 does the geometry (double-precision trig included) once, then one lane per
// (bin, sample) turns its 4 taps into patch entries (csr_gather.h: key = aligned 2x2 pixel patch, 4 weights), merges
// the entries of its bin that share a patch (the 4 samples of a bin are 4 consecutive lanes) and emits the survivors
// (emit_entry).  Measured and not kept (profiles/r05_roi_bwd_notes.md): the four counter atomics of a lane issued back
// to back (+2 us), the merge through LDS hash tables (equal), through fma selects (equal), 8 workgroups per CU (spills).
template <int VARIANT>
__global__ __launch_bounds__(256) void bwd_patch_taps_kernel(const float* __restrict__ rois, int H, int W, int PH,
                                                            int PW, float spatial_scale, int sample_num,
                                                            TapRec* __restrict__ recs, int* __restrict__ seg_n,
                                                            int* __restrict__ counts, PatchEntry* __restrict__ direct,
                                                            int cap, int* __restrict__ over_flag) {
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  __shared__ RoiGeom s_geom;
  __shared__ int s_n;
  const int r = blockIdx.x;
  if (threadIdx.x == 0) {
    s_geom = roi_geom<VARIANT>(rois + (size_t)r * ROI_COLS, spatial_scale, sample_num, PH, PW, 1, true);
    s_n = 0;
  }
  __syncthreads();
  const RoiGeom g = s_geom;
  const int nbins = PH * PW, spb = sample_num * sample_num, S = nbins * spb;
  const int php = (H + 1) >> 1, pwp = (W + 1) >> 1;
  const int kbase = g.batch * php * pwp;
  const int lane = threadIdx.x & 63, q = lane & 3, qbase = lane & ~3;
  for (int s0 = 0; s0 < S; s0 += 256) {       // a quad of lanes is active or idle together (S % 4 == 0 when spb == 4)
    const bool active = s0 + (int)threadIdx.x < S;
    const int s = active ? s0 + threadIdx.x : S - 1;
    const int bin = s / spb, rr = s % spb;
    const SamplePos sp = sample_pos<VARIANT>(g, bin / PW, bin % PW, rr / sample_num, rr % sample_num, H, W);
    const int valid = active && sp.valid && g.batch >= 0;     // batch < 0: masked RoI
    const float hy = (float)(1. - (double)sp.ly), hx = (float)(1. - (double)sp.lx);   // as make_sample
    const float w0[4] = {(hy * hx) / g.count, (hy * sp.lx) / g.count, (sp.ly * hx) / g.count,
                         (sp.ly * sp.lx) / g.count};
    const int ys[4] = {sp.y_low, sp.y_low, sp.y_high, sp.y_high};
    const int xs[4] = {sp.x_low, sp.x_high, sp.x_low, sp.x_high};
    int key[4], slot[4];
    float wv[4][4];
    bool first[4] = {true, true, true, true};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      key[k] = kbase + (ys[k] >> 1) * pwp + (xs[k] >> 1);
      slot[k] = (ys[k] & 1) * 2 + (xs[k] & 1);
#pragma unroll
      for (int i = 0; i < 4; i++) wv[k][i] = slot[k] == i ? w0[k] : 0.f;
    }
    // own taps that share a patch: the first one collects
#pragma unroll
    for (int k = 1; k < 4; k++)
#pragma unroll
      for (int j = 0; j < k; j++)
        if (key[j] == key[k]) {
#pragma unroll
          for (int i = 0; i < 4; i++) wv[j][i] += slot[k] == i ? w0[k] : 0.f;
          first[k] = false;
        }
    if (spb == 4) {
      // the other three samples of the bin (same quad of lanes; S % 4 == 0, so a quad is active or idle together):
      // every tap of theirs in one of my patches adds its weight; a tap of mine survives only if no lower lane of
      // the quad has one in the same patch
#pragma unroll
      for (int d = 1; d < 4; d++) {
        const int src = qbase | ((q + d) & 3);
        const bool earlier = ((q + d) & 3) < q;
        const int ov = __shfl(valid, src, 64);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int ok = __shfl(key[j], src, 64);
          const int os = __shfl(slot[j], src, 64);
          const float ow = __shfl(w0[j], src, 64);
#pragma unroll
          for (int k = 0; k < 4; k++) {
            const bool same = ov && ok == key[k];
#pragma unroll
            for (int i = 0; i < 4; i++) wv[k][i] += (same && os == i) ? ow : 0.f;
            first[k] = first[k] && !(same && earlier);
          }
        }
      }
    }
    const int src_row = r * nbins + bin;
    TapRec* __restrict__ seg = recs + (size_t)r * S * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool keep = valid && first[k] &&
                        (wv[k][0] != 0.f || wv[k][1] != 0.f || wv[k][2] != 0.f || wv[k][3] != 0.f);
      if (keep) emit_entry(key[k], src_row, wv[k][0], wv[k][1], wv[k][2], wv[k][3], counts, direct, cap, seg, &s_n);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    seg_n[r] = s_n;
    if (s_n) *over_flag = 1;     // a flag, not a sum: every writer stores the same value (no serialised atomics)
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

// RiRoIAlign backward = rotated RoIAlign backward of the orientation-mixed gradient: with
// out(c, o) = r * val(c, o - ind) + l * val(c, o - ind + 1)   (plane indices mod nO; riroi_align.py:L130-152),
// d val(c, j) = r * G(c, j + ind) + l * G(c, j + ind - 1).  One thread per (row, channel group) of the (R, nbins, C)
// row matrix; src == dst is fine: the nO planes of a group are read into registers before any is written.
__global__ __launch_bounds__(256) void riroi_mix_rows_kernel(const float* src, float* dst,
                                                            const float* __restrict__ rois, long n_groups, int nbins,
                                                            int groups_per_row, int nO) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n_groups) return;
  const long row = t / groups_per_row;
  const int r = (int)(row / nbins);
  int ind;
  float l_var, r_var;
  ri_params(rois[(size_t)r * 6 + 5], nO, ind, l_var, r_var);
  float v[16];
  for (int j = 0; j < nO; j++) v[j] = src[t * nO + j];
  for (int j = 0; j < nO; j++) dst[t * nO + j] = r_var * v[(j + ind) % nO] + l_var * v[(j + ind - 1 + nO) % nO];
}

template <int VARIANT>
int run_gather(const float* grad_out, const float* rois, int R, int N, int C, int H, int W, int PH, int PW,
               float scale, int sample_num, float* grad_in, void* ws, bool grad_out_cl, bool ws_clean,
               hipStream_t st, int n_orient = 0) {
  const int nbins = PH * PW, spb = sample_num * sample_num;
  const long nkeys = patch_keys(N, H, W), seg_cap = (long)nbins * spb * 4;
  PatchWs w = patch_carve(ws, nkeys, R, seg_cap, patch_direct_cap(nkeys, R * seg_cap));
  float* gT = (float*)((char*)ws + w.bytes);
  if (!ws_clean) {   // workspace of unknown content: zero the row counters and the ticket (a clean one is handed back clean)
    int he = jdet_zero_async(w.counts, patch_zero_bytes(nkeys), st);
    if (he) return he;
  }
  hipLaunchKernelGGL((bwd_patch_taps_kernel<VARIANT>), dim3(R), dim3(256), 0, st, rois, H, W, PH, PW, scale,
                     sample_num, w.recs, w.seg_n, w.counts, w.direct, w.cap, w.counts + nkeys + 1);
  const float* rows = grad_out;   // channels-last (R, PH, PW, C) IS the (R, nbins, C) row matrix the gather wants
  if (!grad_out_cl) {
    dim3 tg(jdet_cdiv(nbins, 32), jdet_cdiv(C, 32), R);
    hipLaunchKernelGGL(bwd_transpose_kernel, tg, dim3(256), 0, st, grad_out, gT, C, nbins);
    rows = gT;
  }
  if (n_orient > 1) {   // RiRoIAlign: the gather reads orientation-mixed rows (written to the workspace copy)
    const long n_groups = (long)R * nbins * (C / n_orient);
    hipLaunchKernelGGL(riroi_mix_rows_kernel, dim3((unsigned)((n_groups + 255) / 256)), dim3(256), 0, st, rows, gT,
                       rois, n_groups, nbins, C / n_orient, n_orient);
    rows = gT;
  }
  return patch_finish_and_gather(w, nkeys, R, seg_cap, rows, C, grad_in, N, H, W, st);
}

// The plan of a backward: counters + direct rows + overflow CSR of the patch rows (everything run_gather builds before the
// gather).  It depends on the RoIs, the map size and the bin grid -- not on the gradient or on C.
template <int VARIANT>
int build_plan(const float* rois, int R, int N, int H, int W, int PH, int PW, float scale, int sample_num, void* plan,
               hipStream_t st) {
  const int nbins = PH * PW, spb = sample_num * sample_num;
  const long nkeys = patch_keys(N, H, W), seg_cap = (long)nbins * spb * 4;
  PatchWs w = patch_carve(plan, nkeys, R, seg_cap, patch_direct_cap(nkeys, R * seg_cap));
  int he = jdet_zero_async(w.counts, patch_zero_bytes(nkeys), st);
  if (he) return he;
  hipLaunchKernelGGL((bwd_patch_taps_kernel<VARIANT>), dim3(R), dim3(256), 0, st, rois, H, W, PH, PW, scale,
                     sample_num, w.recs, w.seg_n, w.counts, w.direct, w.cap, w.counts + nkeys + 1);
  return patch_finish(w, nkeys, R, seg_cap, st);
}

}  // namespace

// Defined in roi_align.hip: the atomic scatter path (RiRoI, adaptive sampling, odd channel counts).
int jdet_roi_align_backward_atomic(int variant, const float* grad_out, const float* rois, int R, int N, int C,
                                   int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                   int n_orient, const int32_t* order, float* grad_in, hipStream_t st);

static bool gather_ok(int variant, int R, int N, int C, int H, int W, int PH, int PW, int sample_num) {
  if (sample_num <= 0 || C % 4 != 0 || R <= 0) return false;
  const long npix = (long)N * H * W, ntaps = (long)R * PH * PW * sample_num * sample_num * 4;
  if (npix >= (1L << 30) || ntaps >= (1L << 31) || (long)R * PH * PW >= (1L << 31) || R > 65535) return false;
  return true;
}

JDET_API size_t jdet_roi_align_backward_workspace(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                                 int sample_num) {
  if (!gather_ok(variant, R, N, C, H, W, PH, PW, sample_num)) return 0;
  const long nkeys = patch_keys(N, H, W), seg_cap = (long)PH * PW * sample_num * sample_num * 4;
  return patch_carve(nullptr, nkeys, R, seg_cap, patch_direct_cap(nkeys, R * seg_cap)).bytes +
         align256(sizeof(float) * (size_t)R * PH * PW * C);
}

// Bytes at the start of the workspace that the gather path needs zero on entry and leaves zero on return (the row
// counters + the scan's ticket); 0 when the shape is served by the atomic path.
JDET_API size_t jdet_roi_align_backward_clean_bytes(int variant, int R, int N, int C, int H, int W, int PH, int PW,
                                                   int sample_num) {
  if (!gather_ok(variant, R, N, C, H, W, PH, PW, sample_num)) return 0;
  return patch_zero_bytes(patch_keys(N, H, W));
}

static int backward_gather(int variant, const float* grad_out, const float* rois, int R, int N, int C, int H, int W,
                           int PH, int PW, float spatial_scale, int sample_num, float* grad_in, void* workspace,
                           bool grad_out_cl, bool ws_clean, hipStream_t st, int n_orient = 1) {
  switch (variant) {
    case JDET_ROI_RIROI:   // rotated geometry; the orientation mix is applied to the gradient rows first
      if (n_orient < 1 || n_orient > 16 || C % n_orient != 0) return JDET_E_UNSUPPORTED;
      return run_gather<JDET_ROI_ROTATED>(grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in, workspace, grad_out_cl, ws_clean, st, n_orient);
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
  if (variant == JDET_ROI_RIROI && (n_orient < 1 || n_orient > 16 || C % n_orient != 0))
    return jdet_roi_align_backward_atomic(variant, grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale,
                                          sample_num, n_orient, order, grad_in, st);
  return backward_gather(variant, grad_out, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in,
                         workspace, false, false, st, n_orient);
}

JDET_API int jdet_roi_align_backward_cl(int variant, const float* grad_out_cl, const float* rois, int R, int N,
                                        int C, int H, int W, int PH, int PW, float spatial_scale, int sample_num,
                                        int n_orient, float* grad_in, void* workspace, size_t workspace_bytes,
                                        int workspace_clean, jdet_stream_t stream) {
  if (variant < 0 || variant > 4 || N < 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || R < 0)
    return JDET_E_BADARG;
  const size_t need = jdet_roi_align_backward_workspace(variant, R, N, C, H, W, PH, PW, sample_num);
  if (need == 0) return JDET_E_UNSUPPORTED;   // adaptive sampling, C % 4, R == 0: use the (R,C,PH,PW) entry
  if (!workspace || workspace_bytes < need) return JDET_E_WORKSPACE;
  if (!grad_out_cl || !rois || !grad_in) return JDET_E_BADARG;
  return backward_gather(variant, grad_out_cl, rois, R, N, C, H, W, PH, PW, spatial_scale, sample_num, grad_in,
                         workspace, true, workspace_clean != 0, (hipStream_t)stream, n_orient);
}

// ---- the plan of a backward, built once per RoI set and gathered from any number of times (round 6) ----------------
// In training the RoIs of a step are known at the forward: the plan (the inversion of the scatter: 21 us of the 67 us
// backward at the north-star point) can be emitted then -- beside the forward kernel, on another stream -- and the
// backward is the gather alone.
JDET_API size_t jdet_roi_align_backward_plan_bytes(int variant, int R, int N, int H, int W, int PH, int PW,
                                                  int sample_num) {
  if (variant < 0 || variant > 4 || !gather_ok(variant, R, N, 4, H, W, PH, PW, sample_num)) return 0;
  const long nkeys = patch_keys(N, H, W), seg_cap = (long)PH * PW * sample_num * sample_num * 4;
  return patch_carve(nullptr, nkeys, R, seg_cap, patch_direct_cap(nkeys, R * seg_cap)).bytes;
}

JDET_API int jdet_roi_align_backward_plan(int variant, const float* rois, int R, int N, int H, int W, int PH, int PW,
                                          float spatial_scale, int sample_num, void* plan, size_t plan_bytes,
                                          jdet_stream_t stream) {
  if (variant < 0 || variant > 4 || N <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || R < 0) return JDET_E_BADARG;
  const size_t need = jdet_roi_align_backward_plan_bytes(variant, R, N, H, W, PH, PW, sample_num);
  if (need == 0) return JDET_E_UNSUPPORTED;
  if (!plan || plan_bytes < need) return JDET_E_WORKSPACE;
  if (!rois) return JDET_E_BADARG;
  hipStream_t st = (hipStream_t)stream;
  switch (variant) {
    case JDET_ROI_RIROI:   // rotated geometry (the orientation mix acts on the gradient rows, not on the plan)
    case JDET_ROI_ROTATED:
      return build_plan<JDET_ROI_ROTATED>(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, plan, st);
    case JDET_ROI_ROTATED_V1:
      return build_plan<JDET_ROI_ROTATED_V1>(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, plan, st);
    case JDET_ROI_HBB_V0:
      return build_plan<JDET_ROI_HBB_V0>(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, plan, st);
    default:
      return build_plan<JDET_ROI_HBB_V1>(rois, R, N, H, W, PH, PW, spatial_scale, sample_num, plan, st);
  }
}

JDET_API int jdet_roi_align_backward_cl_planned(int variant, const float* grad_out_cl, int R, int N, int C, int H, int W,
                                                int PH, int PW, int sample_num, float* grad_in, const void* plan,
                                                size_t plan_bytes, jdet_stream_t stream) {
  if (variant < 0 || variant > 4 || N <= 0 || C <= 0 || H <= 0 || W <= 0 || PH <= 0 || PW <= 0 || R < 0)
    return JDET_E_BADARG;
  if (variant == JDET_ROI_RIROI) return JDET_E_UNSUPPORTED;   // needs the mixed copy of the rows: the unplanned entry
  if (!gather_ok(variant, R, N, C, H, W, PH, PW, sample_num)) return JDET_E_UNSUPPORTED;
  const size_t need = jdet_roi_align_backward_plan_bytes(variant, R, N, H, W, PH, PW, sample_num);
  if (!plan || plan_bytes < need) return JDET_E_WORKSPACE;
  if (!grad_out_cl || !grad_in) return JDET_E_BADARG;
  const long nkeys = patch_keys(N, H, W), seg_cap = (long)PH * PW * sample_num * sample_num * 4;
  const PatchWs w = patch_carve(const_cast<void*>(plan), nkeys, R, seg_cap, patch_direct_cap(nkeys, R * seg_cap));
  return patch_gather(w, nkeys, grad_out_cl, C, grad_in, N, H, W, true, (hipStream_t)stream);
}
