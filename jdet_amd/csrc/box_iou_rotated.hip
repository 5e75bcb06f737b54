// Rotated-box IoU and rotated NMS for gfx950 (MI355X).
//
// Reference semantics:
//   python/jdet/ops/box_iou_rotated.py:L13-310   (vertices, 16 edge intersections + 8 containment
//                                                  tests, Graham hull, shoelace area, IoU)
//   python/jdet/ops/box_iou_rotated_v1.py:L69-76 (opposite vertex convention)
//   python/jdet/ops/nms_rotated.py:L281-310      (label column -> IoU 0 across labels)
//   python/jdet/ops/nms_rotated.py:L414-449      (CPU greedy rule, `>=`), L352-411 + L450-493
//                                                (CUDA 64x64 bitmask kernel, `>`, host scan)
//
// MI355X design (not a translation):
//   * IoU is ALU/latency bound (20 B per box, ~400-3000 flop per overlapping pair).  The
//     intersection-point array (<= 24 points) is the only dynamically indexed state; it lives
//     in LDS as [slot][lane] (bank = lane -> conflict-free for any per-lane slot), everything
//     else is fully unrolled into VGPRs -> no scratch memory.
//   * a conservative circumscribed-circle test returns the exact 0 the reference would compute
//     for clearly disjoint pairs (the overwhelming majority) before any heavy work.
//   * float/double mix, comparison constants and operation order follow the reference's CPU
//     path literally and the file is compiled with -ffp-contract=off, so IoU values -- and
//     therefore NMS keep masks and assigner labels -- are bit-identical to the CPU oracle.
//     The hull sort replays libstdc++'s insertion sort (what std::sort runs for <= 16 elements).
//   * NMS: wave64 == 64-bit suppression word.  One wave per 64x64 tile of score-sorted boxes:
//     lanes are the 64 column boxes, the row box is wave-uniform, __ballot() yields the mask
//     word directly; only upper-triangle tiles run.  The greedy scan stays on the device (one
//     workgroup: diagonal tile resolved by v_readlane chain, remaining words OR-ed in parallel
//     into LDS) -- no cudaDeviceSynchronize + host loop as in the reference.
#include "common.h"
#include "nms_scan.h"

namespace {

struct P2 {
  float x, y;
};
__device__ __forceinline__ P2 mk(float x, float y) { P2 p; p.x = x; p.y = y; return p; }
__device__ __forceinline__ P2 sub(P2 a, P2 b) { return mk(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float dot_2d(P2 a, P2 b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ float cross_2d(P2 a, P2 b) { return a.x * b.y - b.x * a.y; }

struct RBox {
  float x_ctr, y_ctr, w, h, a;
};

template <int V1>
__device__ __forceinline__ void rotated_vertices(const RBox& box, P2 (&pts)[4]) {
  const double theta = box.a;
  const float cosTheta2 = (float)cos(theta) * 0.5f;
  const float sinTheta2 = (float)sin(theta) * 0.5f;
  if (!V1) {
    pts[0].x = box.x_ctr - sinTheta2 * box.h - cosTheta2 * box.w;
    pts[0].y = box.y_ctr + cosTheta2 * box.h - sinTheta2 * box.w;
    pts[1].x = box.x_ctr + sinTheta2 * box.h - cosTheta2 * box.w;
    pts[1].y = box.y_ctr - cosTheta2 * box.h - sinTheta2 * box.w;
  } else {
    pts[0].x = box.x_ctr + sinTheta2 * box.h + cosTheta2 * box.w;
    pts[0].y = box.y_ctr + cosTheta2 * box.h - sinTheta2 * box.w;
    pts[1].x = box.x_ctr - sinTheta2 * box.h + cosTheta2 * box.w;
    pts[1].y = box.y_ctr - cosTheta2 * box.h - sinTheta2 * box.w;
  }
  pts[2].x = 2 * box.x_ctr - pts[0].x;
  pts[2].y = 2 * box.y_ctr - pts[0].y;
  pts[3].x = 2 * box.x_ctr - pts[1].x;
  pts[3].y = 2 * box.y_ctr - pts[1].y;
}

// hull-sort predicate of the reference CPU path (box_iou_rotated.py:L318-325)
__device__ __forceinline__ bool cpu_less(P2 A, P2 B) {
  const float temp = cross_2d(A, B);
  if ((double)fabsf(temp) < 1e-6) return dot_2d(A, A) < dot_2d(B, B);
  return temp > 0;
}

// Per-lane point stack in LDS: element i of this lane is at base[i * NT].
template <int NT>
struct LanePts {
  float* x;
  float* y;
  __device__ __forceinline__ P2 get(int i) const { return mk(x[i * NT], y[i * NT]); }
  __device__ __forceinline__ void set(int i, P2 p) const {
    x[i * NT] = p.x;
    y[i * NT] = p.y;
  }
};

// IoU of two boxes given as 5 floats each (raw, un-shifted), reference single_box_iou_rotated
// (box_iou_rotated.py:L281-310).  SORT 0 = CPU std::sort replay, 1 = CUDA exchange sort.
template <int V1, int NT>
__device__ float single_box_iou(const float* b1, const float* b2, int sort_mode, LanePts<NT> q) {
  RBox box1, box2;
  const double center_shift_x = (double)(b1[0] + b2[0]) / 2.0;
  const double center_shift_y = (double)(b1[1] + b2[1]) / 2.0;
  box1.x_ctr = (float)((double)b1[0] - center_shift_x);
  box1.y_ctr = (float)((double)b1[1] - center_shift_y);
  box1.w = b1[2]; box1.h = b1[3]; box1.a = b1[4];
  box2.x_ctr = (float)((double)b2[0] - center_shift_x);
  box2.y_ctr = (float)((double)b2[1] - center_shift_y);
  box2.w = b2[2]; box2.h = b2[3]; box2.a = b2[4];
  const float area1 = box1.w * box1.h;
  const float area2 = box2.w * box2.h;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;

  P2 pts1[4], pts2[4], vec1[4], vec2[4];
  rotated_vertices<V1>(box1, pts1);
  rotated_vertices<V1>(box2, pts2);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    vec1[i] = sub(pts1[(i + 1) & 3], pts1[i]);
    vec2[i] = sub(pts2[(i + 1) & 3], pts2[i]);
  }
  // --- get_intersection_points (L74-153)
  int num = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float det = cross_2d(vec2[j], vec1[i]);
      if ((double)fabsf(det) <= 1e-14) continue;
      const P2 vec12 = sub(pts2[j], pts1[i]);
      const float t1 = cross_2d(vec2[j], vec12) / det;
      const float t2 = cross_2d(vec1[i], vec12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f) {
        q.set(num++, mk(pts1[i].x + vec1[i].x * t1, pts1[i].y + vec1[i].y * t1));
      }
    }
  }
  {
    const P2 AB = vec2[0], DA = vec2[3];
    const float ABdotAB = dot_2d(AB, AB), ADdotAD = dot_2d(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub(pts1[i], pts2[0]);
      const float APdotAB = dot_2d(AP, AB);
      const float APdotAD = -dot_2d(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        q.set(num++, pts1[i]);
    }
  }
  {
    const P2 AB = vec1[0], DA = vec1[3];
    const float ABdotAB = dot_2d(AB, AB), ADdotAD = dot_2d(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const P2 AP = sub(pts2[i], pts1[0]);
      const float APdotAB = dot_2d(AP, AB);
      const float APdotAD = -dot_2d(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        q.set(num++, pts2[i]);
    }
  }
  if (num <= 2) return 0.f;

  // --- convex_hull_graham (L155-238), in place: q[i] <- p[i] - start
  int t = 0;
  {
    P2 best = q.get(0);
    for (int i = 1; i < num; i++) {
      const P2 p = q.get(i);
      if (p.y < best.y || (p.y == best.y && p.x < best.x)) {
        t = i;
        best = p;
      }
    }
    for (int i = 0; i < num; i++) q.set(i, sub(q.get(i), best));
    const P2 tmp = q.get(0);
    q.set(0, q.get(t));
    q.set(t, tmp);
  }
  int k;
  if (sort_mode == 0) {
    // The reference fills dist[] BEFORE std::sort and never permutes it (L190-193, L316-325),
    // so Step 4's scan sees pre-sort distances: find k on the unsorted array.
    for (k = 1; k < num; k++) {
      const P2 p = q.get(k);
      if ((double)dot_2d(p, p) > 1e-8) break;
    }
    // libstdc++ std::__insertion_sort over q[1..num)
    for (int i = 2; i < num; i++) {
      const P2 val = q.get(i);
      if (cpu_less(val, q.get(1))) {
        for (int m = i; m > 1; m--) q.set(m, q.get(m - 1));
        q.set(1, val);
      } else {
        int nx = i - 1;
        P2 pv = q.get(nx);
        while (cpu_less(val, pv)) {
          q.set(nx + 1, pv);
          nx--;
          pv = q.get(nx);
        }
        q.set(nx + 1, val);
      }
    }
  } else {
    // reference CUDA exchange sort (L338-351); dist[] is swapped with q there, i.e. always
    // equals dot(q[i],q[i])
    for (int i = 1; i < num - 1; i++) {
      for (int j = i + 1; j < num; j++) {
        const P2 qi = q.get(i), qj = q.get(j);
        const float crossProduct = cross_2d(qi, qj);
        if (((double)crossProduct < -1e-6) ||
            ((double)fabsf(crossProduct) < 1e-6 && dot_2d(qi, qi) > dot_2d(qj, qj))) {
          q.set(i, qj);
          q.set(j, qi);
        }
      }
    }
    for (k = 1; k < num; k++) {
      const P2 p = q.get(k);
      if ((double)dot_2d(p, p) > 1e-8) break;
    }
  }
  if (k == num) return 0.f / (area1 + area2 - 0.f);  // hull is one point: area 0
  q.set(1, q.get(k));
  int m = 2;
  for (int i = k + 1; i < num; i++) {
    const P2 qi = q.get(i);
    while (m > 1) {
      const P2 a = q.get(m - 2);
      if (cross_2d(sub(qi, a), sub(q.get(m - 1), a)) >= 0) m--; else break;
    }
    q.set(m++, qi);
  }
  // --- polygon_area (L240-252)
  float area = 0;
  if (m > 2) {
    const P2 q0 = q.get(0);
    for (int i = 1; i < m - 1; i++)
      area += fabsf(cross_2d(sub(q.get(i), q0), sub(q.get(i + 1), q0)));
    area = area / 2.0f;
  }
  const float intersection = area;
  return intersection / (area1 + area2 - intersection);
}

// Conservative disjointness test on the circumscribed circles.  True only when the boxes are
// separated by a margin far above fp32 rounding, in which case the reference finds no
// intersection point and returns exactly 0.  NaN/inf fall through to the full path.
__device__ __forceinline__ bool surely_disjoint(const float* b1, const float* b2) {
  const float dx = b1[0] - b2[0], dy = b1[1] - b2[1];
  const float d2 = dx * dx + dy * dy;
  const float r1 = 0.5f * sqrtf(b1[2] * b1[2] + b1[3] * b1[3]);
  const float r2 = 0.5f * sqrtf(b2[2] * b2[2] + b2[3] * b2[3]);
  const float rr = r1 + r2;
  return d2 > rr * rr * 1.001f + 1e-3f;
}

template <int NT>
__device__ __forceinline__ float iou_dispatch(const float* b1, const float* b2, int version,
                                              int sort_mode, LanePts<NT> q) {
  if (surely_disjoint(b1, b2)) return 0.f;
  return version ? single_box_iou<1, NT>(b1, b2, sort_mode, q)
                 : single_box_iou<0, NT>(b1, b2, sort_mode, q);
}

// ---------------------------------------------------------------------------------------------
// pairwise IoU: one lane per (i, j), j fastest (coalesced stores)
// ---------------------------------------------------------------------------------------------
constexpr int kIouBlock = 128;

__global__ __launch_bounds__(kIouBlock) void box_iou_kernel(const float* __restrict__ boxes1, int n1,
                                                            const float* __restrict__ boxes2, int n2,
                                                            int stride, int version, int sort_mode,
                                                            float* __restrict__ ious) {
  __shared__ float s_x[24 * kIouBlock];
  __shared__ float s_y[24 * kIouBlock];
  LanePts<kIouBlock> q;
  q.x = s_x + threadIdx.x;
  q.y = s_y + threadIdx.x;
  const long total = (long)n1 * n2;
  for (long p = (long)blockIdx.x * kIouBlock + threadIdx.x; p < total; p += (long)gridDim.x * kIouBlock) {
    const int i = (int)(p / n2), j = (int)(p % n2);
    float a[5], b[5];
#pragma unroll
    for (int k = 0; k < 5; k++) {
      a[k] = boxes1[(size_t)i * stride + k];
      b[k] = boxes2[(size_t)j * stride + k];
    }
    ious[p] = iou_dispatch<kIouBlock>(a, b, version, sort_mode, q);
  }
}

// ---------------------------------------------------------------------------------------------
// NMS: tile mask kernel (one wave per 64x64 tile) + on-device greedy scan
// ---------------------------------------------------------------------------------------------
// min / max label of the 64 boxes of tile `blk` (box_len == 6)
__device__ __forceinline__ void tile_label_range(const float* __restrict__ dets, const int32_t* __restrict__ order,
                                                 int n, int blk, int lane, float& lo, float& hi) {
  const int pos = blk * 64 + lane;
  const float l = pos < n ? dets[(size_t)order[pos] * 6 + 5] : 0.f;
  lo = pos < n ? l : INFINITY;
  hi = pos < n ? l : -INFINITY;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, off, 64));
    hi = fmaxf(hi, __shfl_xor(hi, off, 64));
  }
}

// `mask` and `tile_jmax` arrive zeroed.  Tiles whose label ranges are disjoint (ml_nms with the boxes visited
// class by class: almost all of them) return at once; tile_jmax[r] = last column block of row block r that can hold
// a set bit, so that the scan reads only those.
// HBB: the boxes are (xc, yc, w, h, 0 [, label]) -- axis-aligned; the overlap is the rectangle formula
// inter / (a + b - inter) instead of the polygon clipping (same value up to rounding, ~30x fewer instructions: the
// RPN proposal NMS of the two-stage detectors, 8.7 k boxes per image, spent 0.55-0.74 ms here).
template <bool HBB>
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ dets, int n, int box_len,
                                                      const int32_t* __restrict__ order, float thr,
                                                      int cmp_ge, int sort_mode,
                                                      unsigned long long* __restrict__ mask,
                                                      int* __restrict__ tile_jmax) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x;
  if (col_blk < row_blk) return;  // only the upper triangle is ever read by the scan
  const int lane = threadIdx.x;
  if (box_len == 6 && col_blk != row_blk) {
    float rlo, rhi, clo, chi;
    tile_label_range(dets, order, n, row_blk, lane, rlo, rhi);
    tile_label_range(dets, order, n, col_blk, lane, clo, chi);
    if (rhi < clo || chi < rlo) return;   // no pair of equal labels in this tile
  }
  if (lane == 0) atomicMax(&tile_jmax[row_blk], col_blk);
  __shared__ float s_x[24 * 64];
  __shared__ float s_y[24 * 64];
  __shared__ float s_row[64 * 6];                  // the tile's row boxes (lane i staged row i)
  __shared__ float s_col[64 * 6];
  __shared__ unsigned short s_pair[64 * 64];       // candidate pairs (row << 6 | column), rotated boxes only
  __shared__ unsigned long long s_word[64];
  LanePts<64> q;
  q.x = s_x + lane;
  q.y = s_y + lane;
  const int col_blocks = (n + 63) >> 6;
  const int col = col_blk * 64 + lane;  // position in visiting order
  const bool col_ok = col < n;
  float cb[6] = {0, 0, 0, 0, 0, 0};
  if (col_ok) {
    const float* p = dets + (size_t)order[col] * box_len;
#pragma unroll
    for (int k = 0; k < 5; k++) cb[k] = p[k];
    if (box_len == 6) cb[5] = p[5];
  }
  const int rows = min(64, n - row_blk * 64);
  {
    float rb[6] = {0, 0, 0, 0, 0, 0};
    if (lane < rows) {
      const float* p = dets + (size_t)order[row_blk * 64 + lane] * box_len;
#pragma unroll
      for (int k = 0; k < 5; k++) rb[k] = p[k];
      if (box_len == 6) rb[5] = p[5];
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
      s_row[lane * 6 + k] = rb[k];
      s_col[lane * 6 + k] = cb[k];
    }
    s_word[lane] = 0ull;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (HBB) {
    for (int i = 0; i < rows; i++) {
      const int row = row_blk * 64 + i;
      float rb[6];
#pragma unroll
      for (int k = 0; k < 6; k++) rb[k] = s_row[i * 6 + k];   // same address in every lane: broadcast
      bool hit = false;
      if (col_ok && col > row && !(box_len == 6 && rb[5] != cb[5])) {
        const float iw = fminf(rb[0] + 0.5f * rb[2], cb[0] + 0.5f * cb[2]) - fmaxf(rb[0] - 0.5f * rb[2], cb[0] - 0.5f * cb[2]);
        const float ih = fminf(rb[1] + 0.5f * rb[3], cb[1] + 0.5f * cb[3]) - fmaxf(rb[1] - 0.5f * rb[3], cb[1] - 0.5f * cb[3]);
        const float inter = fmaxf(iw, 0.f) * fmaxf(ih, 0.f);
        const float uni = rb[2] * rb[3] + cb[2] * cb[3] - inter;
        const float ovr = uni > 0.f ? inter / uni : 0.f;
        hit = cmp_ge ? (ovr >= thr) : (ovr > thr);
      }
      const unsigned long long word = __ballot(hit);
      if (lane == 0 && word) mask[(size_t)row * col_blocks + col_blk] = word;
    }
    return;
  }
  // Rotated boxes, two phases.  The polygon clipping costs thousands of instructions and only pairs whose
  // circumscribed circles meet need it: running it inside the row loop left most lanes of most iterations idle while
  // one pair was being clipped (355 us for 2000 boxes, 4x the per-pair time of box_iou_kernel).
  //   1. every (row, column) pair of the tile that is in order, of one label (ml_nms: different labels -> IoU 0,
  //      nms_rotated.py:L283-286) and not surely disjoint is appended to a list (one cheap pass, lane = column);
  //   2. the list is clipped 64 pairs at a time, one pair per lane, and the hits OR-ed into the rows' words.
  // Same pairs, same function, same bits as before.  A threshold that an IoU of exactly 0 passes makes EVERY pair a
  // hit: then the disjoint ones have to stay in the list.
  const bool zero_hits = cmp_ge ? !(thr > 0.f) : (thr < 0.f);
  int total = 0;
  for (int i = 0; i < rows; i++) {
    const int row = row_blk * 64 + i;
    float rb[6];
#pragma unroll
    for (int k = 0; k < 6; k++) rb[k] = s_row[i * 6 + k];
    const bool cand = col_ok && col > row && !(box_len == 6 && rb[5] != cb[5]) &&
                      (zero_hits || !surely_disjoint(rb, cb));
    const unsigned long long word = __ballot(cand);
    if (cand) s_pair[total + __builtin_amdgcn_mbcnt_hi((unsigned)(word >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)word, 0))] =
        (unsigned short)((i << 6) | lane);
    total += __popcll(word);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    if (t < total) {
      const int pr = s_pair[t];
      const int i = pr >> 6, j = pr & 63;
      float rb[5], b2[5];
#pragma unroll
      for (int k = 0; k < 5; k++) {
        rb[k] = s_row[i * 6 + k];
        b2[k] = s_col[j * 6 + k];
      }
      // argument order (earlier, later) as in the CPU loop L443
      const float ovr = iou_dispatch<64>(rb, b2, 0, sort_mode, q);
      if (cmp_ge ? (ovr >= thr) : (ovr > thr)) atomicOr(&s_word[i], 1ull << j);
    }
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  if (lane < rows) {
    const unsigned long long word = s_word[lane];
    if (word) mask[(size_t)(row_blk * 64 + lane) * col_blocks + col_blk] = word;
  }
}

}  // namespace

JDET_API int jdet_box_iou_rotated(const float* boxes1, int n1, const float* boxes2, int n2, int stride,
                                  int version, int sort_mode, float* ious, jdet_stream_t stream) {
  if (n1 < 0 || n2 < 0 || stride < 5 || (version != 0 && version != 1) ||
      (sort_mode != 0 && sort_mode != 1))
    return JDET_E_BADARG;
  if (n1 == 0 || n2 == 0) return JDET_OK;
  if (!boxes1 || !boxes2 || !ious) return JDET_E_BADARG;
  const long total = (long)n1 * n2;
  const int grid = (int)((total + kIouBlock - 1) / kIouBlock > 1048576 ? 1048576
                                                                       : (total + kIouBlock - 1) / kIouBlock);
  hipLaunchKernelGGL(box_iou_kernel, dim3(grid), dim3(kIouBlock), 0, (hipStream_t)stream, boxes1, n1,
                     boxes2, n2, stride, version, sort_mode, ious);
  return jdet_launch_status();
}

JDET_API size_t jdet_nms_rotated_workspace(int n) { return jdet_nms::workspace_bytes(n); }

// horizontal != 0: every angle is 0 (the caller's promise): rectangle overlap instead of polygon clipping.
// n_labels > 1 (box_len 6 only): the labels are the integers 0 .. n_labels-1 and `order` visits the boxes label by
// label -- one scan workgroup per label instead of one for all.
JDET_API int jdet_nms_labeled(const float* dets, int n, int box_len, const int32_t* order, float iou_threshold,
                              int cmp_ge, int sort_mode, int horizontal, int n_labels, uint8_t* keep,
                              void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  if (n < 0 || (box_len != 5 && box_len != 6) || (sort_mode != 0 && sort_mode != 1) || n_labels < 1 ||
      n_labels > 65535 || (n_labels > 1 && box_len != 6))
    return JDET_E_BADARG;
  if (n == 0) return JDET_OK;
  if (!dets || !order || !keep || !workspace) return JDET_E_BADARG;
  if (workspace_bytes < jdet_nms_rotated_workspace(n)) return JDET_E_WORKSPACE;
  const int col_blocks = (n + 63) >> 6;
  if (col_blocks > jdet_nms::kScanMaxWords) return JDET_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  unsigned long long* mask = (unsigned long long*)workspace;
  int* tile_jmax = (int*)((char*)workspace + jdet_nms::mask_bytes(n));
  int e = jdet_zero_async(workspace, jdet_nms_rotated_workspace(n), st);
  if (e) return e;
  if (horizontal)
    hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(col_blocks, col_blocks), dim3(64), 0, st, dets, n, box_len,
                       order, iou_threshold, cmp_ge ? 1 : 0, sort_mode, mask, tile_jmax);
  else
    hipLaunchKernelGGL(nms_mask_kernel<false>, dim3(col_blocks, col_blocks), dim3(64), 0, st, dets, n, box_len,
                       order, iou_threshold, cmp_ge ? 1 : 0, sort_mode, mask, tile_jmax);
  e = jdet_launch_status();
  if (e) return e;
  return jdet_nms::launch_scan(mask, n, order, tile_jmax, dets + 5, 6, n_labels, keep, st);
}

JDET_API int jdet_nms_rotated(const float* dets, int n, int box_len, const int32_t* order,
                              float iou_threshold, int cmp_ge, int sort_mode, uint8_t* keep,
                              void* workspace, size_t workspace_bytes, jdet_stream_t stream) {
  return jdet_nms_labeled(dets, n, box_len, order, iou_threshold, cmp_ge, sort_mode, 0, 1, keep, workspace,
                          workspace_bytes, stream);
}
