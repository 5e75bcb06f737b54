// RepPoints convex GIoU with the gradient w.r.t. the 9 points, for gfx950.
//
// Reference (CUDA only): ops/reppoints_convex_iou/convex_giou.py:L29-47 (`reppoints_convex_giou`: aligned pairs,
// output (N, 19) = 18 point gradients + the GIoU) and convex_giou_kernel.cu:L725-821 (`devrIoU`): hull of the 9 points
// (Jarvis, L613-723), intersection area with the quadrilateral and its gradient through the clipping (L117-447),
// hull area and its gradient (L68-115), area of the hull of both polygons' vertices and its gradient (L539-610), then
//     giou = I/U - (C - U)/C,   U = |A| + |B| - I,
//     d giou = (U + I)/U^2 dI - I/U^2 dA - (dI - dA)/C - U/C^2 dC                            (L782-790)
// with the gradient of a point that is not a hull vertex zero.  The reference derives the ~40 partial derivatives by
// hand, one CUDA thread per pair with 100-point local arrays.
//
// MI355X design (not a translation): the value is a composition of differentiable pieces selected by discrete decisions
// (which points are hull vertices, which edges cross), so the gradient is taken by FORWARD-MODE DUAL NUMBERS: half a
// wave (32 lanes) per pair, lane j < 18 carries the dual seed d/d(coordinate j), every lane runs the SAME control flow
// (decisions look at the value parts only, which are identical across the lanes of a pair), and the dual part of the
// final GIoU in lane j IS the j-th partial derivative -- 18 derivatives in the time of one evaluation, no hand-written
// derivative code to get wrong.  Geometry: Andrew monotone-chain hulls (the 9 points; the union of both polygons'
// vertices), Sutherland-Hodgman clipping of the convex hull by the convex quadrilateral, shoelace areas; all in double
// like the reference.  Collinear / duplicate hull candidates are dropped (they change neither the areas nor, almost
// everywhere, the derivatives).
#include "common.h"

namespace {

struct Dual {
  double v, d;
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return {a.v + b.v, a.d + b.d}; }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return {a.v - b.v, a.d - b.d}; }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return {a.v * b.v, a.d * b.v + a.v * b.d}; }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return {q, (a.d - q * b.d) / b.v};
}
__device__ __forceinline__ Dual dconst(double v) { return {v, 0.0}; }

struct DP {
  Dual x, y;
};

__device__ __forceinline__ Dual cross(const DP& o, const DP& a, const DP& b) {
  return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y);
}

constexpr int kMaxPts = 16;   // 9 + 4 hull inputs; a clipped polygon has at most 9 + 4 vertices

// Andrew's monotone chain; p[0..n) is sorted in place; returns the hull size, h counter-clockwise
__device__ int hull(DP* p, int n, DP* h) {
  for (int i = 1; i < n; i++) {        // insertion sort by (x, y) values
    const DP t = p[i];
    int j = i - 1;
    while (j >= 0 && (p[j].x.v > t.x.v || (p[j].x.v == t.x.v && p[j].y.v > t.y.v))) {
      p[j + 1] = p[j];
      j--;
    }
    p[j + 1] = t;
  }
  int m = 0;
  for (int i = 0; i < n; i++) {
    while (m >= 2 && cross(h[m - 2], h[m - 1], p[i]).v <= 0.0) m--;
    h[m++] = p[i];
  }
  const int lower = m + 1;
  for (int i = n - 2; i >= 0; i--) {
    while (m >= lower && cross(h[m - 2], h[m - 1], p[i]).v <= 0.0) m--;
    h[m++] = p[i];
  }
  return m > 1 ? m - 1 : m;            // the last point repeats the first
}

__device__ Dual area(const DP* p, int n) {     // shoelace, positive for counter-clockwise
  Dual s = dconst(0.0);
  for (int i = 0; i < n; i++) {
    const DP& a = p[i];
    const DP& b = p[i + 1 == n ? 0 : i + 1];
    s = s + (a.x * b.y - b.x * a.y);
  }
  return s * dconst(0.5);
}

// convex polygon p (n vertices, any orientation) clipped to the left of every edge of the counter-clockwise convex q
__device__ int clip(DP* p, int n, const DP* q, int nq, DP* tmp) {
  for (int e = 0; e < nq && n > 0; e++) {
    const DP& a = q[e];
    const DP& b = q[e + 1 == nq ? 0 : e + 1];
    int m = 0;
    for (int i = 0; i < n; i++) {
      const DP& cur = p[i];
      const DP& nxt = p[i + 1 == n ? 0 : i + 1];
      const Dual sc = cross(a, b, cur), sn = cross(a, b, nxt);
      const bool in_c = sc.v >= 0.0, in_n = sn.v >= 0.0;
      if (in_c) tmp[m++] = cur;
      if (in_c != in_n) {                       // the edge crosses the line: the crossing point moves with both ends
        const Dual t = sc / (sc - sn);
        tmp[m++] = DP{cur.x + t * (nxt.x - cur.x), cur.y + t * (nxt.y - cur.y)};
      }
    }
    n = m;
    for (int i = 0; i < n; i++) p[i] = tmp[i];
  }
  return n;
}

// 32 lanes per pair: lane j < 18 differentiates w.r.t. coordinate j of the point set; lane 18 stores the value
__global__ __launch_bounds__(64) void convex_giou_kernel(const float* __restrict__ pointsets,
                                                        const float* __restrict__ polygons, int N,
                                                        float* __restrict__ out) {
  const int pair = blockIdx.x * 2 + (threadIdx.x >> 5);
  const int j = threadIdx.x & 31;
  if (pair >= N || j > 18) return;
  DP pts[kMaxPts], P[kMaxPts], Q[4], work[kMaxPts], tmp[kMaxPts];
  for (int i = 0; i < 9; i++) {
    pts[i].x = Dual{(double)pointsets[(size_t)pair * 18 + 2 * i], j == 2 * i ? 1.0 : 0.0};
    pts[i].y = Dual{(double)pointsets[(size_t)pair * 18 + 2 * i + 1], j == 2 * i + 1 ? 1.0 : 0.0};
  }
  for (int i = 0; i < 4; i++) {
    Q[i].x = dconst((double)polygons[(size_t)pair * 8 + 2 * i]);
    Q[i].y = dconst((double)polygons[(size_t)pair * 8 + 2 * i + 1]);
  }
  for (int i = 0; i < 9; i++) work[i] = pts[i];
  const int nP = hull(work, 9, P);
  Dual aQ = area(Q, 4);
  if (aQ.v < 0.0) {                              // make the quadrilateral counter-clockwise
    const DP t = Q[1];
    Q[1] = Q[3];
    Q[3] = t;
    aQ = dconst(0.0) - aQ;
  }
  const Dual aP = area(P, nP);                   // the hull is counter-clockwise: >= 0
  for (int i = 0; i < nP; i++) work[i] = P[i];
  const int nI = clip(work, nP, Q, 4, tmp);
  Dual inter = nI >= 3 ? area(work, nI) : dconst(0.0);
  if (inter.v < 0.0) inter = dconst(0.0) - inter;
  const Dual uni = aP + aQ - inter;
  for (int i = 0; i < nP; i++) work[i] = P[i];
  for (int i = 0; i < 4; i++) work[nP + i] = Q[i];
  const int nC = hull(work, nP + 4, tmp);
  const Dual encl = area(tmp, nC);
  const Dual giou = inter / uni - (encl - uni) / encl;
  float* o = out + (size_t)pair * 19;
  if (j < 18) o[j] = (float)giou.d;
  else o[18] = (float)giou.v;
}

}  // namespace

JDET_API int jdet_convex_giou(const float* pointsets, const float* polygons, int N, float* out, jdet_stream_t stream) {
  if (N < 0) return JDET_E_BADARG;
  if (N == 0) return JDET_OK;
  if (!pointsets || !polygons || !out) return JDET_E_BADARG;
  hipLaunchKernelGGL(convex_giou_kernel, dim3((unsigned)((N + 1) / 2)), dim3(64), 0, (hipStream_t)stream, pointsets,
                     polygons, N, out);
  return jdet_launch_status();
}
