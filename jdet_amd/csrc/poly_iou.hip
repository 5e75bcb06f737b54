// Polygon (quadrilateral) IoU and polygon NMS.
//
// Reference: python/jdet/ops/nms_poly.py -- `devPolyIoU` (L113-133: intersection of two 4-point polygons as a signed
// sum over origin-fan triangle pairs, each pair clipped as convex polygons, L79-111), `poly_nms_kernel` + host scan
// (L135-232), `multiclass_poly_nms` (L234-245), and the CPU-side `iou_poly` (L247-252, shapely) that the DOTA
// evaluation and the tile merging call per pair (data/devkits/voc_eval.py:L196, result_merge.py:L50,L108).
//
// Same definition, own formulation: area(P n Q) = sum_i sum_j s_i s_j area(T_i n U_j) with T_i = (c, p_i, p_i+1),
// U_j = (c, q_j, q_j+1) the fan triangles about a common point c and s = their orientations -- exact for any simple
// polygons, convex or not.  c is the centroid of the 8 vertices instead of the coordinate origin the reference uses:
// in fp32, fans about (0,0) of boxes near (1000,1000) cancel areas of order 1e6 to get an overlap of order 1e2; about
// the centroid every term is of the overlap's own size.  Triangle pairs are clipped with Sutherland-Hodgman against
// the three half planes of the second triangle (<= 6 vertices) and measured with the shoelace formula.
// mode 0: the reference kernel's degenerate rule (union == 0 -> (inter + 1) / (union + 1));
// mode 1: `iou_poly`'s rule (inter / max(union, 0.01)).
// Parity: unpinned by reference execution (CUDA-only source, no fixtures; shapely is not installed here) -- the tests
// hold this kernel to a float64 restatement of the same definition and to closed-form areas (tests/test_gpu_poly.py).
#include "common.h"
#include "nms_scan.h"

namespace {

struct P2 {
  float x, y;
};

__device__ __forceinline__ float cross2(const P2& a, const P2& b) { return a.x * b.y - a.y * b.x; }
__device__ __forceinline__ float cross3(const P2& o, const P2& a, const P2& b) {
  return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x);
}

// keep the part of the convex polygon v[0..n) on the left of (or on) the directed line e0 -> e1
__device__ __forceinline__ int clip_left(P2* v, int n, const P2& e0, const P2& e1) {
  P2 out[8];
  int m = 0;
  for (int i = 0; i < n; i++) {
    const P2 cur = v[i], nxt = v[i + 1 == n ? 0 : i + 1];
    const float dc = cross3(e0, e1, cur), dn = cross3(e0, e1, nxt);
    if (dc >= 0.f) out[m++] = cur;
    if ((dc > 0.f && dn < 0.f) || (dc < 0.f && dn > 0.f)) {
      const float t = dc / (dc - dn);
      out[m].x = cur.x + t * (nxt.x - cur.x);
      out[m].y = cur.y + t * (nxt.y - cur.y);
      m++;
    }
  }
  for (int i = 0; i < m; i++) v[i] = out[i];
  return m;
}

// signed overlap of the fan triangles (0, a, b) and (0, c, d): + when they wind the same way
__device__ __forceinline__ float fan_pair(P2 a, P2 b, P2 c, P2 d) {
  const float wa = cross2(a, b), wc = cross2(c, d);
  if (wa == 0.f || wc == 0.f) return 0.f;
  if (wa < 0.f) { const P2 t = a; a = b; b = t; }
  if (wc < 0.f) { const P2 t = c; c = d; d = t; }
  const P2 o = {0.f, 0.f};
  P2 v[8];
  v[0] = o; v[1] = a; v[2] = b;
  int n = 3;
  n = clip_left(v, n, o, c);
  if (n >= 3) n = clip_left(v, n, c, d);
  if (n >= 3) n = clip_left(v, n, d, o);
  if (n < 3) return 0.f;
  float twice = 0.f;
  for (int i = 0; i < n; i++) twice += cross2(v[i], v[i + 1 == n ? 0 : i + 1]);
  const float area = 0.5f * fabsf(twice);
  return (wa < 0.f) != (wc < 0.f) ? -area : area;
}

__device__ __forceinline__ float quad_area2(const P2* p) {   // twice the signed area
  float s = 0.f;
  for (int i = 0; i < 4; i++) s += cross2(p[i], p[(i + 1) & 3]);
  return s;
}

__device__ float poly_iou(const float* __restrict__ pa, const float* __restrict__ pb, int mode) {
  P2 p[4], q[4];
  float cx = 0.f, cy = 0.f;
  for (int i = 0; i < 4; i++) {
    p[i].x = pa[2 * i]; p[i].y = pa[2 * i + 1];
    q[i].x = pb[2 * i]; q[i].y = pb[2 * i + 1];
    cx += p[i].x + q[i].x;
    cy += p[i].y + q[i].y;
  }
  cx *= 0.125f; cy *= 0.125f;
  // quick reject: disjoint bounding boxes
  float pminx = p[0].x, pmaxx = p[0].x, pminy = p[0].y, pmaxy = p[0].y;
  float qminx = q[0].x, qmaxx = q[0].x, qminy = q[0].y, qmaxy = q[0].y;
  for (int i = 1; i < 4; i++) {
    pminx = fminf(pminx, p[i].x); pmaxx = fmaxf(pmaxx, p[i].x); pminy = fminf(pminy, p[i].y); pmaxy = fmaxf(pmaxy, p[i].y);
    qminx = fminf(qminx, q[i].x); qmaxx = fmaxf(qmaxx, q[i].x); qminy = fminf(qminy, q[i].y); qmaxy = fmaxf(qmaxy, q[i].y);
  }
  for (int i = 0; i < 4; i++) {
    p[i].x -= cx; p[i].y -= cy;
    q[i].x -= cx; q[i].y -= cy;
  }
  const float a1 = 0.5f * fabsf(quad_area2(p)), a2 = 0.5f * fabsf(quad_area2(q));
  float inter = 0.f;
  if (!(pmaxx < qminx || qmaxx < pminx || pmaxy < qminy || qmaxy < pminy)) {
    // orientation-normalised polygons: the double sum is then the (non-negative) intersection area
    const bool rp = quad_area2(p) < 0.f, rq = quad_area2(q) < 0.f;
    for (int i = 0; i < 4; i++) {
      const P2 a = p[rp ? (4 - i) & 3 : i], b = p[rp ? (3 - i) & 3 : (i + 1) & 3];
      for (int j = 0; j < 4; j++) {
        const P2 c = q[rq ? (4 - j) & 3 : j], d = q[rq ? (3 - j) & 3 : (j + 1) & 3];
        inter += fan_pair(a, b, c, d);
      }
    }
    inter = fmaxf(inter, 0.f);
  }
  const float uni = a1 + a2 - inter;
  if (mode == 1) return inter / fmaxf(uni, 0.01f);
  return uni == 0.f ? (inter + 1.f) / (uni + 1.f) : inter / uni;
}

__global__ __launch_bounds__(256) void poly_iou_kernel(const float* __restrict__ p1, int n1, int stride1,
                                                       const float* __restrict__ p2, int n2, int stride2, int mode,
                                                       float* __restrict__ ious) {
  const long total = (long)n1 * n2;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int i = (int)(t / n2), j = (int)(t % n2);
    ious[t] = poly_iou(p1 + (size_t)i * stride1, p2 + (size_t)j * stride2, mode);
  }
}

// one wave per 64 x 64 tile of the (visiting order) pair matrix, as nms_mask_kernel of box_iou_rotated.hip;
// row_len 9: column 8 is the label (different labels never suppress each other)
__global__ __launch_bounds__(64) void poly_nms_mask_kernel(const float* __restrict__ polys, int n, int row_len,
                                                           const int32_t* __restrict__ order, float thr,
                                                           unsigned long long* __restrict__ mask,
                                                           int* __restrict__ tile_jmax) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;
  const int lane = threadIdx.x;
  const int col_blocks = (n + 63) >> 6;
  const int col = col_blk * 64 + lane;
  const bool col_ok = col < n;
  float cb[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (col_ok) {
    const float* p = polys + (size_t)order[col] * row_len;
    for (int k = 0; k < row_len; k++) cb[k] = p[k];
  }
  if (row_len == 9 && col_blk != row_blk) {   // no pair of equal labels in this tile: nothing to do
    const int rpos = row_blk * 64 + lane;
    const float rl = rpos < n ? polys[(size_t)order[rpos] * 9 + 8] : 0.f;
    float rlo = rpos < n ? rl : INFINITY, rhi = rpos < n ? rl : -INFINITY;
    float clo = col_ok ? cb[8] : INFINITY, chi = col_ok ? cb[8] : -INFINITY;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      rlo = fminf(rlo, __shfl_xor(rlo, off, 64)); rhi = fmaxf(rhi, __shfl_xor(rhi, off, 64));
      clo = fminf(clo, __shfl_xor(clo, off, 64)); chi = fmaxf(chi, __shfl_xor(chi, off, 64));
    }
    if (rhi < clo || chi < rlo) return;
  }
  if (lane == 0) atomicMax(&tile_jmax[row_blk], col_blk);
  const int rows = min(64, n - row_blk * 64);
  for (int i = 0; i < rows; i++) {
    const int row = row_blk * 64 + i;
    const float* rp = polys + (size_t)order[row] * row_len;   // wave-uniform
    bool hit = false;
    if (col_ok && col > row && !(row_len == 9 && rp[8] != cb[8])) hit = poly_iou(rp, cb, 0) > thr;
    const unsigned long long word = __ballot(hit);
    if (lane == 0 && word) mask[(size_t)row * col_blocks + col_blk] = word;
  }
}

}  // namespace

JDET_API int jdet_poly_iou(const float* polys1, int n1, int stride1, const float* polys2, int n2, int stride2,
                           int mode, float* ious, jdet_stream_t stream) {
  if (n1 < 0 || n2 < 0 || stride1 < 8 || stride2 < 8 || (mode != 0 && mode != 1)) return JDET_E_BADARG;
  if (n1 == 0 || n2 == 0) return JDET_OK;
  if (!polys1 || !polys2 || !ious) return JDET_E_BADARG;
  const long total = (long)n1 * n2;
  const long blocks = (total + 255) / 256;
  hipLaunchKernelGGL(poly_iou_kernel, dim3((unsigned)(blocks > 1048576 ? 1048576 : blocks)), dim3(256), 0,
                     (hipStream_t)stream, polys1, n1, stride1, polys2, n2, stride2, mode, ious);
  return jdet_launch_status();
}

// workspace: jdet_nms_rotated_workspace(n).  polys (n, row_len): 8 coordinates (+ label when row_len == 9);
// order: visiting order (descending score; label by label when n_labels > 1, labels 0 .. n_labels-1).
JDET_API int jdet_nms_poly(const float* polys, int n, int row_len, const int32_t* order, float iou_threshold,
                           int n_labels, uint8_t* keep, void* workspace, size_t workspace_bytes,
                           jdet_stream_t stream) {
  if (n < 0 || (row_len != 8 && row_len != 9) || n_labels < 1 || n_labels > 65535 || (n_labels > 1 && row_len != 9))
    return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!polys || !order || !keep || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_nms::workspace_bytes(n)) return JDET_E_WORKSPACE;
  const int col_blocks = (n + 63) >> 6;
  if (col_blocks > jdet_nms::kScanMaxWords) return JDET_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mask = (unsigned long long*)workspace;
  int* tile_jmax = (int*)((char*)workspace + jdet_nms::mask_bytes(n));
  int e = jdet_zero_async(workspace, jdet_nms::workspace_bytes(n), st);
  if (e) return e;
  hipLaunchKernelGGL(poly_nms_mask_kernel, dim3(col_blocks, col_blocks), dim3(64), 0, st, polys, n, row_len, order,
                     iou_threshold, mask, tile_jmax);
  e = jdet_launch_status();
  if (e) return e;
  return jdet_nms::launch_scan(mask, n, order, tile_jmax, polys + 8, 9, n_labels, keep, st);
}
