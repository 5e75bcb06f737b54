// jdet_oracle.cpp -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's rotated-box hot path (Jittor/JDet,
// snapshot 2025-03-10).  It is the *checker* for the HIP kernels in
// jdet_amd/csrc/: only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load it.  The product path (jdet_amd) never links,
// imports or falls back to anything in this file.
//
// Language: C++ (g++), not plain C, because the reference's CPU IoU path
// orders hull points with std::sort (python/jdet/ops/box_iou_rotated.py:L316-325)
// and the libstdc++ ordering is part of what "the Jittor CPU reference" means.
//
// Every function follows the float/double mix of the cited reference lines
// literally (e.g. `1. - ly` is a double subtraction rounded to float), so that
// it agrees bit-for-bit with the reference kernel text when that text is
// host-compiled (oracle/build_ref.py -> oracle/_ref/, used by
// tests/golden/gen_golden.py).  Compile with -ffp-contract=off.
//
// Pinning status: (a) the reference's known-answer literals (box_iou_rotated.py:L513-514, nms_rotated.py:L599-603);
// (b) golden vectors produced by the reference's true CPU sources compiled where they lie (oracle/build_ref.py ->
// tests/golden/*.npz: rotated IoU, rotated NMS, ARF), bit for bit; (c) for the operators the reference has as GPU
// kernel text only (RoIAlign x5, DeformConv v1 / v2 sampling, PSRoI pooling, feature refinement, RepPoints geometry,
// convex_sort, polygon NMS): the reference's own kernels compiled for gfx950 (oracle/build_ref_hip.py) and run on the
// device next to this restatement (tests/test_gpu_reference_kernels.py: bit-equal where no trigonometry is involved,
// <= 4e-6 where the kernel text's cos(float) resolves to the device's cosf); (d) closed forms that involve neither this
// file nor any reference build (tests/closed_form.py, tests/test_dcn_v2_oracle.py, tests/test_convex_oracle.py).

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#define JO_API extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------
// RoIAlign family
//   variant 0: ROIAlignRotated      ops/roi_align_rotated.py:L61-127 / L165-255
//   variant 1: ROIAlignRotated_v1   ops/roi_align_rotated_v1.py:L71-145 / L193-298
//   variant 2: RiRoIAlign           ops/riroi_align.py:L70-163 / L228-358
//   variant 3: ROIAlign (hbb) v0    ops/roi_align.py:L93-204, ROI_ALIGN_VERSION 0
//   variant 4: ROIAlign (hbb) v1    same, ROI_ALIGN_VERSION 1
// Feature map NCHW fp32, rois (R,6)=[b,xc,yc,w,h,theta] or (R,5)=[b,x1,y1,x2,y2].
// ---------------------------------------------------------------------------
namespace {

enum { V_ROT = 0, V_ROT_V1 = 1, V_RI = 2, V_HBB0 = 3, V_HBB1 = 4 };

// bilinear_interpolate_gradient: roi_align_rotated.py:L128-163 (v1: `<` clamps,
// roi_align_rotated_v1.py:L161-164).  Returns false when the sample is out of
// the map (weights 0, indices -1 in the reference).
inline bool bilinear_setup(int variant, int height, int width, float y, float x,
                           float& w1, float& w2, float& w3, float& w4,
                           int& x_low, int& x_high, int& y_low, int& y_high) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    w1 = w2 = w3 = w4 = 0.f;
    x_low = x_high = y_low = y_high = -1;
    return false;
  }
  if (variant == V_ROT_V1) {
    if (y < 0) y = 0;
    if (x < 0) x = 0;
  } else {
    if (y <= 0) y = 0;
    if (x <= 0) x = 0;
  }
  y_low = (int)y;
  x_low = (int)x;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = 1. - ly;  // double subtraction, rounded to float (L49)
  float hx = 1. - lx;
  w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
  return true;
}

// bilinear_interpolate: roi_align_rotated.py:L21-59
inline float bilinear_interp(int variant, const float* data, int height, int width,
                             float y, float x) {
  float w1, w2, w3, w4;
  int xl, xh, yl, yh;
  if (!bilinear_setup(variant, height, width, y, x, w1, w2, w3, w4, xl, xh, yl, yh)) return 0.f;
  float lt = data[yl * width + xl];
  float rt = data[yl * width + xh];
  float lb = data[yh * width + xl];
  float rb = data[yh * width + xh];
  float val = (w1 * lt + w2 * rt + w3 * lb + w4 * rb);
  return val;
}

struct RoiGeom {
  int batch;
  float center_w, center_h;   // rotated variants
  float start_w, start_h;     // offset of bin (0,0): -w/2 (rotated) or x1*s (hbb)
  float bin_h, bin_w;
  int grid_h, grid_w;
  float cosT, sinT;
  float count;
  // RiRoIAlign
  float l_var, r_var;
  int ind;
};

// Per-RoI scalar prologue shared by forward and backward.  `sample_num` is the
// int the kernel receives (the python wrapper passes a float literal that C++
// truncates: roi_align_rotated.py:L271,L280).
inline RoiGeom roi_geom(int variant, const float* roi, float spatial_scale, int sample_num,
                        int pooled_h, int pooled_w, int nOrientation, bool backward) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.l_var = 0.f; g.r_var = 1.f; g.ind = 0;
  float roi_width, roi_height;
  if (variant == V_HBB0 || variant == V_HBB1) {
    // roi_align.py:L105-117
    float roi_start_w = roi[1] * spatial_scale;
    float roi_start_h = roi[2] * spatial_scale;
    if (variant == V_HBB1) {
      float roi_end_w = (roi[3] + 1) * spatial_scale;
      float roi_end_h = (roi[4] + 1) * spatial_scale;
      roi_width = fmaxf(roi_end_w - roi_start_w, 0.);
      roi_height = fmaxf(roi_end_h - roi_start_h, 0.);
    } else {
      float roi_end_w = roi[3] * spatial_scale;
      float roi_end_h = roi[4] * spatial_scale;
      roi_width = fmaxf(roi_end_w - roi_start_w, 1.);
      roi_height = fmaxf(roi_end_h - roi_start_h, 1.);
    }
    g.start_w = roi_start_w;
    g.start_h = roi_start_h;
    g.center_w = g.center_h = 0.f;
    g.cosT = 1.f; g.sinT = 0.f;
  } else {
    // roi_align_rotated.py:L77-85 ; v1 L89-90 subtracts 0.5
    g.center_w = roi[1] * spatial_scale;
    g.center_h = roi[2] * spatial_scale;
    if (variant == V_ROT_V1) {
      g.center_w = roi[1] * spatial_scale - (float)0.5;
      g.center_h = roi[2] * spatial_scale - (float)0.5;
    }
    roi_width = roi[3] * spatial_scale;
    roi_height = roi[4] * spatial_scale;
    float theta = roi[5];
    roi_width = std::max(roi_width, (float)1.);
    roi_height = std::max(roi_height, (float)1.);
    g.start_h = -roi_height / 2.0;  // L99-100
    g.start_w = -roi_width / 2.0;
    // host-compiled `cos(theta)` on a float resolves to the double overload
    // (global ::cos), the CUDA build to cosf; both round to float here.
    g.cosT = (float)cos((double)theta);
    g.sinT = (float)sin((double)theta);
    if (variant == V_RI) {
      // riroi_align.py:L105-113 ; PI literal L8
      float ind_float = theta * nOrientation / (2 * 3.141592653);
      int ind = (int)floor(ind_float);
      g.l_var = ind_float - (float)ind;
      g.r_var = 1.0 - g.l_var;
      g.ind = (ind + nOrientation) % nOrientation;
    }
  }
  g.bin_h = (float)roi_height / (float)pooled_h;
  g.bin_w = (float)roi_width / (float)pooled_w;
  g.grid_h = (sample_num > 0) ? sample_num : (int)ceil(roi_height / pooled_h);
  g.grid_w = (sample_num > 0) ? sample_num : (int)ceil(roi_width / pooled_w);
  int cnt = g.grid_h * g.grid_w;
  // v1 forward clamps count to >= 1 (v1 L120); v1 backward does not (v1 L246)
  if (variant == V_ROT_V1 && !backward) cnt = std::max(cnt, 1);
  g.count = (float)cnt;
  return g;
}

// sample position for (ph,pw,iy,ix): roi_align_rotated.py:L106-118 (v1 L133-134;
// hbb roi_align.py:L129-132)
inline void sample_xy(int variant, const RoiGeom& g, int ph, int pw, int iy, int ix,
                      float& x, float& y) {
  const float yy = g.start_h + ph * g.bin_h +
                   static_cast<float>(iy + .5f) * g.bin_h / static_cast<float>(g.grid_h);
  const float xx = g.start_w + pw * g.bin_w +
                   static_cast<float>(ix + .5f) * g.bin_w / static_cast<float>(g.grid_w);
  if (variant == V_HBB0 || variant == V_HBB1) {
    x = xx; y = yy;
  } else if (variant == V_ROT_V1) {
    x = xx * g.cosT + yy * g.sinT + g.center_w;
    y = yy * g.cosT - xx * g.sinT + g.center_h;
  } else {
    x = xx * g.cosT - yy * g.sinT + g.center_w;
    y = xx * g.sinT + yy * g.cosT + g.center_h;
  }
}

}  // namespace

// out: (R, C*nO, PH, PW).  For variants != RI pass nOrientation = 1.
// `channels` is the per-orientation channel count (C), so the map has C*nO planes.
JO_API void jo_roi_align_forward(int variant, const float* feat, int N, int channels, int H, int W,
                                 const float* rois, int R, int PH, int PW, float spatial_scale,
                                 int sample_num, int nOrientation, float* out) {
  (void)N;
  const int roi_cols = (variant == V_HBB0 || variant == V_HBB1) ? 5 : 6;
  const int nO = (variant == V_RI) ? nOrientation : 1;
#pragma omp parallel for schedule(dynamic, 1)
  for (int n = 0; n < R; n++) {
    const RoiGeom g = roi_geom(variant, rois + (size_t)n * roi_cols, spatial_scale, sample_num,
                               PH, PW, nO, false);
    for (int c = 0; c < channels; c++)
      for (int o = 0; o < nO; o++) {
        int ind_rot = (o - g.ind + nO) % nO;
        int ind_rot_plus = (ind_rot + 1 + nO) % nO;
        const float* d0 = feat + ((size_t)(g.batch * channels * nO + c * nO + ind_rot)) * H * W;
        const float* d1 = feat + ((size_t)(g.batch * channels * nO + c * nO + ind_rot_plus)) * H * W;
        for (int ph = 0; ph < PH; ph++)
          for (int pw = 0; pw < PW; pw++) {
            float output_val = 0.;
            for (int iy = 0; iy < g.grid_h; iy++)
              for (int ix = 0; ix < g.grid_w; ix++) {
                float x, y;
                sample_xy(variant, g, ph, pw, iy, ix, x, y);
                float val = bilinear_interp(variant, d0, H, W, y, x);
                if (variant == V_RI) {
                  float val_plus = bilinear_interp(variant, d1, H, W, y, x);
                  output_val += g.r_var * val + g.l_var * val_plus;  // riroi L158
                } else {
                  output_val += val;
                }
              }
            output_val /= g.count;
            out[(((size_t)n * channels * nO + c * nO + o) * PH + ph) * PW + pw] = output_val;
          }
      }
  }
}

// grad_in: (N, C*nO, H, W), zero-filled here (reference: cudaMemsetAsync L302).
// Serial accumulation in index order n,c,o,ph,pw,iy,ix (the reference's atomics
// have no defined order; comparisons against this use an fp32 tolerance).
JO_API void jo_roi_align_backward(int variant, const float* grad_out, int N, int channels, int H,
                                  int W, const float* rois, int R, int PH, int PW,
                                  float spatial_scale, int sample_num, int nOrientation,
                                  float* grad_in) {
  const int roi_cols = (variant == V_HBB0 || variant == V_HBB1) ? 5 : 6;
  const int nO = (variant == V_RI) ? nOrientation : 1;
  memset(grad_in, 0, sizeof(float) * (size_t)N * channels * nO * H * W);
  for (int n = 0; n < R; n++) {
    const RoiGeom g = roi_geom(variant, rois + (size_t)n * roi_cols, spatial_scale, sample_num,
                               PH, PW, nO, true);
    for (int c = 0; c < channels; c++)
      for (int o = 0; o < nO; o++) {
        int ind_rot = (o - g.ind + nO) % nO;
        int ind_rot_plus = (ind_rot + 1 + nO) % nO;
        float* d0 = grad_in + ((size_t)(g.batch * channels * nO + c * nO + ind_rot)) * H * W;
        float* d1 = grad_in + ((size_t)(g.batch * channels * nO + c * nO + ind_rot_plus)) * H * W;
        for (int ph = 0; ph < PH; ph++)
          for (int pw = 0; pw < PW; pw++) {
            const float top =
                grad_out[(((size_t)n * channels * nO + c * nO + o) * PH + ph) * PW + pw];
            for (int iy = 0; iy < g.grid_h; iy++)
              for (int ix = 0; ix < g.grid_w; ix++) {
                float x, y;
                sample_xy(variant, g, ph, pw, iy, ix, x, y);
                float w1, w2, w3, w4;
                int xl, xh, yl, yh;
                bilinear_setup(variant, H, W, y, x, w1, w2, w3, w4, xl, xh, yl, yh);
                float g1 = top * w1 / g.count;
                float g2 = top * w2 / g.count;
                float g3 = top * w3 / g.count;
                float g4 = top * w4 / g.count;
                if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
                  if (variant == V_RI) {  // riroi L337-353
                    d0[yl * W + xl] += g1 * g.r_var;
                    d0[yl * W + xh] += g2 * g.r_var;
                    d0[yh * W + xl] += g3 * g.r_var;
                    d0[yh * W + xh] += g4 * g.r_var;
                    d1[yl * W + xl] += g1 * g.l_var;
                    d1[yl * W + xh] += g2 * g.l_var;
                    d1[yh * W + xl] += g3 * g.l_var;
                    d1[yh * W + xh] += g4 * g.l_var;
                  } else {
                    d0[yl * W + xl] += g1;
                    d0[yl * W + xh] += g2;
                    d0[yh * W + xl] += g3;
                    d0[yh * W + xh] += g4;
                  }
                }
              }
          }
      }
  }
}

// ---------------------------------------------------------------------------
// Rotated IoU  (ops/box_iou_rotated.py:L13-310, CPU header L312-326;
//               _v1 vertex convention ops/box_iou_rotated_v1.py:L69-76)
// ---------------------------------------------------------------------------
namespace {

struct Pt {
  float x, y;
  Pt(float px = 0, float py = 0) : x(px), y(py) {}
  Pt operator+(const Pt& p) const { return Pt(x + p.x, y + p.y); }
  Pt& operator+=(const Pt& p) { x += p.x; y += p.y; return *this; }
  Pt operator-(const Pt& p) const { return Pt(x - p.x, y - p.y); }
  Pt operator*(const float c) const { return Pt(x * c, y * c); }
};
inline float dot_2d(const Pt& A, const Pt& B) { return A.x * B.x + A.y * B.y; }
inline float cross_2d(const Pt& A, const Pt& B) { return A.x * B.y - B.x * A.y; }

struct RBox { float x_ctr, y_ctr, w, h, a; };

// L52-72 (v0) ; v1 L69-76
inline void rotated_vertices(int v1, const RBox& box, Pt (&pts)[4]) {
  double theta = box.a;
  float cosTheta2 = (float)cos(theta) * 0.5f;
  float sinTheta2 = (float)sin(theta) * 0.5f;
  if (!v1) {
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

// L74-153
inline int intersection_points(const Pt (&pts1)[4], const Pt (&pts2)[4], Pt (&inter)[24]) {
  Pt vec1[4], vec2[4];
  for (int i = 0; i < 4; i++) {
    vec1[i] = pts1[(i + 1) % 4] - pts1[i];
    vec2[i] = pts2[(i + 1) % 4] - pts2[i];
  }
  int num = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      float det = cross_2d(vec2[j], vec1[i]);
      if (fabs(det) <= 1e-14) continue;
      Pt vec12 = pts2[j] - pts1[i];
      float t1 = cross_2d(vec2[j], vec12) / det;
      float t2 = cross_2d(vec1[i], vec12) / det;
      if (t1 >= 0.0f && t1 <= 1.0f && t2 >= 0.0f && t2 <= 1.0f)
        inter[num++] = pts1[i] + vec1[i] * t1;
    }
  {
    const Pt& AB = vec2[0];
    const Pt& DA = vec2[3];
    float ABdotAB = dot_2d(AB, AB), ADdotAD = dot_2d(DA, DA);
    for (int i = 0; i < 4; i++) {
      Pt AP = pts1[i] - pts2[0];
      float APdotAB = dot_2d(AP, AB);
      float APdotAD = -dot_2d(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        inter[num++] = pts1[i];
    }
  }
  {
    const Pt& AB = vec1[0];
    const Pt& DA = vec1[3];
    float ABdotAB = dot_2d(AB, AB), ADdotAD = dot_2d(DA, DA);
    for (int i = 0; i < 4; i++) {
      Pt AP = pts2[i] - pts1[0];
      float APdotAB = dot_2d(AP, AB);
      float APdotAD = -dot_2d(AP, DA);
      if ((APdotAB >= 0) && (APdotAD >= 0) && (APdotAB <= ABdotAB) && (APdotAD <= ADdotAD))
        inter[num++] = pts2[i];
    }
  }
  return num;
}

// L155-238 with the CPU sort L316-325 (sort_mode 0) or the CUDA exchange sort
// L338-351 (sort_mode 1).  shift_to_zero is always true at the single call site
// (L275).
inline int convex_hull_graham(const Pt (&p)[24], int num_in, Pt (&q)[24], int sort_mode) {
  int t = 0;
  for (int i = 1; i < num_in; i++)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  const Pt start = p[t];
  for (int i = 0; i < num_in; i++) q[i] = p[i] - start;
  Pt tmp = q[0];
  q[0] = q[t];
  q[t] = tmp;
  float dist[24];
  for (int i = 0; i < num_in; i++) dist[i] = dot_2d(q[i], q[i]);
  if (sort_mode == 0) {
    std::sort(q + 1, q + num_in, [](const Pt& A, const Pt& B) -> bool {
      float temp = cross_2d(A, B);
      if (fabs(temp) < 1e-6) {
        return dot_2d(A, A) < dot_2d(B, B);
      } else {
        return temp > 0;
      }
    });
    // NOTE (reference quirk, L190-203): `dist` is filled BEFORE the sort and is
    // not permuted by std::sort, so Step 4 below reads pre-sort distances.
  } else {
    for (int i = 1; i < num_in - 1; i++)
      for (int j = i + 1; j < num_in; j++) {
        float crossProduct = cross_2d(q[i], q[j]);
        if ((crossProduct < -1e-6) || (fabs(crossProduct) < 1e-6 && dist[i] > dist[j])) {
          Pt q_tmp = q[i]; q[i] = q[j]; q[j] = q_tmp;
          float d_tmp = dist[i]; dist[i] = dist[j]; dist[j] = d_tmp;
        }
      }
  }
  int k;
  for (k = 1; k < num_in; k++)
    if (dist[k] > 1e-8) break;
  if (k == num_in) {
    q[0] = p[t];
    return 1;
  }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num_in; i++) {
    while (m > 1 && cross_2d(q[i] - q[m - 2], q[m - 1] - q[m - 2]) >= 0) m--;
    q[m++] = q[i];
  }
  return m;
}

// L240-252
inline float polygon_area(const Pt (&q)[24], int m) {
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; i++) area += fabs(cross_2d(q[i] - q[0], q[i + 1] - q[0]));
  return area / 2.0;
}

// L254-310 ; nms_rotated.py:L281-310 adds the label test for BOX_LENGTH 6
inline float single_box_iou_rotated(const float* b1, const float* b2, int v1, int sort_mode,
                                    int box_len) {
  if (box_len == 6 && b1[5] != b2[5]) return 0.0;
  RBox box1, box2;
  auto center_shift_x = (b1[0] + b2[0]) / 2.0;  // float add, double divide
  auto center_shift_y = (b1[1] + b2[1]) / 2.0;
  box1.x_ctr = b1[0] - center_shift_x;
  box1.y_ctr = b1[1] - center_shift_y;
  box1.w = b1[2]; box1.h = b1[3]; box1.a = b1[4];
  box2.x_ctr = b2[0] - center_shift_x;
  box2.y_ctr = b2[1] - center_shift_y;
  box2.w = b2[2]; box2.h = b2[3]; box2.a = b2[4];
  const float area1 = box1.w * box1.h;
  const float area2 = box2.w * box2.h;
  if (area1 < 1e-14 || area2 < 1e-14) return 0.f;
  Pt inter[24], ordered[24];
  Pt pts1[4], pts2[4];
  rotated_vertices(v1, box1, pts1);
  rotated_vertices(v1, box2, pts2);
  int num = intersection_points(pts1, pts2, inter);
  if (num <= 2) return 0.0;
  int num_convex = convex_hull_graham(inter, num, ordered, sort_mode);
  const float intersection = polygon_area(ordered, num_convex);
  const float iou = intersection / (area1 + area2 - intersection);
  return iou;
}

}  // namespace

// ious (N,M) row-major.  version 0 = box_iou_rotated, 1 = box_iou_rotated_v1
// (without the python-side too-small post-processing, v1 L515-523, which the
// host wrapper applies).  sort_mode 0 = CPU std::sort, 1 = CUDA exchange sort.
JO_API void jo_box_iou_rotated(const float* b1, int n1, const float* b2, int n2, int stride,
                               int version, int sort_mode, float* ious) {
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++)
      ious[(size_t)i * n2 + j] =
          single_box_iou_rotated(b1 + (size_t)i * stride, b2 + (size_t)j * stride, version,
                                 sort_mode, 5);
}

// Greedy NMS, ops/nms_rotated.py:L414-449 (CPU source).  dets (n, box_len),
// order = indices by descending score.  cmp_ge 1 = CPU rule `ovr >= thr` (L444),
// 0 = CUDA rule `> thr` (L403).  keep: n bytes (bool over ORIGINAL indices).
JO_API void jo_nms_rotated(const float* dets, int n, int box_len, const int32_t* order,
                           float iou_threshold, int cmp_ge, int sort_mode, uint8_t* keep) {
  std::vector<uint8_t> suppressed(n, 0);
  memset(keep, 0, n);
  for (int _i = 0; _i < n; _i++) {
    int i = order[_i];
    if (suppressed[i] == 1) continue;
    keep[i] = 1;
    for (int _j = _i + 1; _j < n; _j++) {
      int j = order[_j];
      if (suppressed[j] == 1) continue;
      float ovr = single_box_iou_rotated(dets + (size_t)i * box_len, dets + (size_t)j * box_len, 0,
                                         sort_mode, box_len);
      if (cmp_ge ? (ovr >= iou_threshold) : (ovr > iou_threshold)) suppressed[j] = 1;
    }
  }
}

// ---------------------------------------------------------------------------
// Deformable conv v1 sampling (ops/dcn_v1.py:L25-306)
// ---------------------------------------------------------------------------
namespace {

// L25-56
inline float dcn_bilinear(const float* data, int data_width, int height, int width, float h,
                          float w) {
  int h_low = (int)floor(h);
  int w_low = (int)floor(w);
  int h_high = h_low + 1;
  int w_high = w_low + 1;
  float lh = h - h_low;
  float lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = data[h_low * data_width + w_low];
  float v2 = 0;
  if (h_low >= 0 && w_high <= width - 1) v2 = data[h_low * data_width + w_high];
  float v3 = 0;
  if (h_high <= height - 1 && w_low >= 0) v3 = data[h_high * data_width + w_low];
  float v4 = 0;
  if (h_high <= height - 1 && w_high <= width - 1) v4 = data[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
  return val;
}

// L58-85
inline float dcn_gradient_weight(float argmax_h, float argmax_w, int h, int w, int height,
                                 int width) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int argmax_h_low = (int)floor(argmax_h);
  int argmax_w_low = (int)floor(argmax_w);
  int argmax_h_high = argmax_h_low + 1;
  int argmax_w_high = argmax_w_low + 1;
  float weight = 0;
  if (h == argmax_h_low && w == argmax_w_low) weight = (h + 1 - argmax_h) * (w + 1 - argmax_w);
  if (h == argmax_h_low && w == argmax_w_high) weight = (h + 1 - argmax_h) * (argmax_w + 1 - w);
  if (h == argmax_h_high && w == argmax_w_low) weight = (argmax_h + 1 - h) * (w + 1 - argmax_w);
  if (h == argmax_h_high && w == argmax_w_high) weight = (argmax_h + 1 - h) * (argmax_w + 1 - w);
  return weight;
}

// L87-128
inline float dcn_coordinate_weight(float argmax_h, float argmax_w, int height, int width,
                                   const float* im, int data_width, int bp_dir) {
  if (argmax_h <= -1 || argmax_h >= height || argmax_w <= -1 || argmax_w >= width) return 0;
  int hl = (int)floor(argmax_h);
  int wl = (int)floor(argmax_w);
  int hh = hl + 1;
  int wh = wl + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - argmax_w) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += -1 * (argmax_w - wl) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - argmax_w) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_w - wl) * im[hh * data_width + wh];
  } else if (bp_dir == 1) {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - argmax_h) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - argmax_h) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += -1 * (argmax_h - hl) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (argmax_h - hl) * im[hh * data_width + wh];
  }
  return weight;
}

}  // namespace

// deformable_im2col_gpu_kernel L130-184.  im (B,C,H,W); offset (B, dg*2*kh*kw, Ho, Wo);
// col (C*kh*kw, B, Ho, Wo).
JO_API void jo_deform_im2col(const float* im, const float* offset, int B, int C, int H, int W,
                             int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                             int dil_h, int dil_w, int dg, float* col) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int cpg = C / dg;
#pragma omp parallel for schedule(static)
  for (int c_im = 0; c_im < C; c_im++)
    for (int b = 0; b < B; b++)
      for (int h_col = 0; h_col < Ho; h_col++)
        for (int w_col = 0; w_col < Wo; w_col++) {
          const int g = c_im / cpg;
          const int h_in = h_col * stride_h - pad_h;
          const int w_in = w_col * stride_w - pad_w;
          const float* im_ptr = im + ((size_t)b * C + c_im) * H * W;
          const float* off_ptr = offset + ((size_t)b * dg + g) * 2 * kh * kw * Ho * Wo;
          for (int i = 0; i < kh; ++i)
            for (int j = 0; j < kw; ++j) {
              const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_col) * Wo + w_col];
              const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_col) * Wo + w_col];
              float val = 0.f;
              const float h_im = h_in + i * dil_h + offset_h;
              const float w_im = w_in + j * dil_w + offset_w;
              if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
                val = dcn_bilinear(im_ptr, W, H, W, h_im, w_im);
              col[((((size_t)c_im * kh * kw + i * kw + j) * B + b) * Ho + h_col) * Wo + w_col] = val;
            }
        }
}

// deformable_col2im_gpu_kernel L185-241: grad_im (B,C,H,W) zero-filled here.
JO_API void jo_deform_col2im(const float* col, const float* offset, int B, int C, int H, int W,
                             int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                             int dil_h, int dil_w, int dg, float* grad_im) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int cpg = C / dg;
  memset(grad_im, 0, sizeof(float) * (size_t)B * C * H * W);
  const long n = (long)C * kh * kw * Ho * Wo * B;
  for (long index = 0; index < n; index++) {
    const int j = (index / Wo / Ho / B) % kw;
    const int i = (index / Wo / Ho / B / kw) % kh;
    const int c = index / Wo / Ho / B / kw / kh;
    const int g = c / cpg;
    int w_out = index % Wo;
    int h_out = (index / Wo) % Ho;
    int b = (index / Wo / Ho) % B;
    int w_in = w_out * stride_w - pad_w;
    int h_in = h_out * stride_h - pad_h;
    const float* off_ptr = offset + ((size_t)b * dg + g) * 2 * kh * kw * Ho * Wo;
    const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
    const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
    const float cur_inv_h = h_in + i * dil_h + offset_h;
    const float cur_inv_w = w_in + j * dil_w + offset_w;
    const float cur_top_grad = col[index];
    const int cur_h = (int)cur_inv_h;
    const int cur_w = (int)cur_inv_w;
    for (int dy = -2; dy <= 2; dy++)
      for (int dx = -2; dx <= 2; dx++)
        if (cur_h + dy >= 0 && cur_h + dy < H && cur_w + dx >= 0 && cur_w + dx < W &&
            fabsf(cur_inv_h - (cur_h + dy)) < 1 && fabsf(cur_inv_w - (cur_w + dx)) < 1) {
          size_t pos = (((size_t)b * C + c) * H + cur_h + dy) * W + cur_w + dx;
          float weight = dcn_gradient_weight(cur_inv_h, cur_inv_w, cur_h + dy, cur_w + dx, H, W);
          grad_im[pos] += weight * cur_top_grad;
        }
  }
}

// deformable_col2im_coord_gpu_kernel L243-306: grad_offset (B, dg*2*kh*kw, Ho, Wo).
JO_API void jo_deform_col2im_coord(const float* col, const float* im, const float* offset, int B,
                                   int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                   int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                                   float* grad_offset) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int cpg = C * kh * kw / dg;  // channel_per_deformable_group (col channels)
  const int offset_channels = 2 * kh * kw * dg;
  const long n = (long)Ho * Wo * offset_channels * B;
#pragma omp parallel for schedule(static)
  for (long index = 0; index < n; index++) {
    float val = 0;
    int w = index % Wo;
    int h = (index / Wo) % Ho;
    int c = (index / Wo / Ho) % offset_channels;
    int b = (index / Wo / Ho) / offset_channels;
    const int g = c / (2 * kh * kw);
    const int col_step = kh * kw;
    int cnt = 0;
    const float* col_ptr = col + (size_t)g * cpg * B * Wo * Ho;
    const float* im_ptr = im + ((size_t)b * dg + g) * cpg / kh / kw * H * W;
    const float* off_ptr = offset + ((size_t)b * dg + g) * 2 * kh * kw * Ho * Wo;
    const int offset_c = c - g * 2 * kh * kw;
    for (int col_c = (offset_c / 2); col_c < cpg; col_c += col_step) {
      const long col_pos = ((((long)col_c * B + b) * Ho) + h) * Wo + w;
      const int bp_dir = offset_c % 2;
      int j = (col_pos / Wo / Ho / B) % kw;
      int i = (col_pos / Wo / Ho / B / kw) % kh;
      int w_out = col_pos % Wo;
      int h_out = (col_pos / Wo) % Ho;
      int w_in = w_out * stride_w - pad_w;
      int h_in = h_out * stride_h - pad_h;
      const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
      const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
      float inv_h = h_in + i * dil_h + offset_h;
      float inv_w = w_in + j * dil_w + offset_w;
      if (inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W) inv_h = inv_w = -2;
      const float weight =
          dcn_coordinate_weight(inv_h, inv_w, H, W, im_ptr + (size_t)cnt * H * W, W, bp_dir);
      val += weight * col_ptr[col_pos];
      cnt += 1;
    }
    grad_offset[index] = val;
  }
}

// ---------------------------------------------------------------------------
// Modulated deformable conv (DCN v2), operator level (ops/dcn_v2.py:L11-306 forward, L308-781 backward)
//   input (B,C,H,W); offset (B, dg*2*kh*kw, Ho, Wo); mask (B, dg*kh*kw, Ho, Wo); weight (Cout, C, kh, kw); bias (Cout)
// Parity status: UNPINNED by reference execution (CUDA-only text, cuBLAS calls: not buildable here); held to closed
// forms in tests/test_dcn_v2_oracle.py (mask = 1 equals the pinned v1 sampling, integer offsets = shifted conv x mask,
// linearity in the mask, adjoint identities, central differences).
// ---------------------------------------------------------------------------
namespace {

// modulated_deformable_im2col_gpu_kernel L86-149 for ONE image: columns (C*kh*kw, Ho*Wo)
void dcn2_im2col_image(const float* im, const float* offset, const float* mask, int C, int H, int W, int kh, int kw,
                       int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, int Ho, int Wo,
                       float* col) {
  const int cpg = C / dg;
  for (int c_im = 0; c_im < C; c_im++)
    for (int h_col = 0; h_col < Ho; h_col++)
      for (int w_col = 0; w_col < Wo; w_col++) {
        const int g = c_im / cpg;
        const int h_in = h_col * stride_h - pad_h;
        const int w_in = w_col * stride_w - pad_w;
        const float* im_ptr = im + (size_t)c_im * H * W;
        const float* off_ptr = offset + (size_t)g * 2 * kh * kw * Ho * Wo;
        const float* mask_ptr = mask + (size_t)g * kh * kw * Ho * Wo;
        for (int i = 0; i < kh; ++i)
          for (int j = 0; j < kw; ++j) {
            const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_col) * Wo + w_col];
            const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_col) * Wo + w_col];
            const float m = mask_ptr[((i * kw + j) * Ho + h_col) * Wo + w_col];
            float val = 0.f;
            const float h_im = h_in + i * dil_h + offset_h;
            const float w_im = w_in + j * dil_w + offset_w;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W) val = dcn_bilinear(im_ptr, W, H, W, h_im, w_im);
            col[(((size_t)c_im * kh * kw + i * kw + j) * Ho + h_col) * Wo + w_col] = val * m;
          }
      }
}

}  // namespace

// dcn_v2_conv_forward L11-306: output = bias (ones GEMM, L238-250) + weight . columns (L260-272), per image
JO_API void jo_dcn_v2_forward(const float* input, const float* offset, const float* mask, const float* weight,
                              const float* bias, int B, int C, int H, int W, int Cout, int kh, int kw, int pad_h,
                              int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg, float* output) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int K = C * kh * kw, P = Ho * Wo;
  std::vector<float> col((size_t)K * P);
  for (int b = 0; b < B; b++) {
    dcn2_im2col_image(input + (size_t)b * C * H * W, offset + (size_t)b * dg * 2 * kh * kw * P,
                      mask + (size_t)b * dg * kh * kw * P, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h,
                      dil_w, dg, Ho, Wo, col.data());
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; co++)
      for (int q = 0; q < P; q++) {
        double acc = bias[co];
        for (int k = 0; k < K; k++) acc += (double)weight[(size_t)co * K + k] * col[(size_t)k * P + q];
        output[((size_t)b * Cout + co) * P + q] = (float)acc;
      }
  }
}

// dcn_v2_conv_backward L308-781, the per-image loop of L711-779.  All five gradients; grad_weight / grad_bias
// accumulate over the batch (beta = 1, L757-776).  NOTE L651-653: modulated_deformable_col2im_cuda hands the kernel
// (pad_h, pad_h) -- the input gradient uses pad_h for BOTH axes; reproduced (a symmetric padding hides it).
JO_API void jo_dcn_v2_backward(const float* input, const float* offset, const float* mask, const float* weight,
                               const float* grad_output, int B, int C, int H, int W, int Cout, int kh, int kw,
                               int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dg,
                               float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight,
                               float* grad_bias) {
  const int Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  const int Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  const int K = C * kh * kw, P = Ho * Wo, kk = kh * kw;
  std::vector<float> col((size_t)K * P);
  std::vector<double> gw((size_t)Cout * K, 0.0), gb(Cout, 0.0);
  memset(grad_input, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int b = 0; b < B; b++) {
    const float* in_n = input + (size_t)b * C * H * W;
    const float* off_n = offset + (size_t)b * dg * 2 * kk * P;
    const float* mask_n = mask + (size_t)b * dg * kk * P;
    const float* go_n = grad_output + (size_t)b * Cout * P;
    float* gi_n = grad_input + (size_t)b * C * H * W;
    float* goff_n = grad_offset + (size_t)b * dg * 2 * kk * P;
    float* gmask_n = grad_mask + (size_t)b * dg * kk * P;
    // columns = weight^T . grad_output   (L722-730)
#pragma omp parallel for schedule(static)
    for (int k = 0; k < K; k++)
      for (int q = 0; q < P; q++) {
        double acc = 0;
        for (int co = 0; co < Cout; co++) acc += (double)weight[(size_t)co * K + k] * go_n[(size_t)co * P + q];
        col[(size_t)k * P + q] = (float)acc;
      }
    // modulated_deformable_col2im_coord_gpu_kernel L560-627 with batch_size = 1
    {
      const int cpg = C * kk / dg;
      const int offset_channels = 2 * kk * dg;
      const long n = (long)Ho * Wo * offset_channels;
      for (long index = 0; index < n; index++) {
        float val = 0, mval = 0;
        int w = index % Wo;
        int h = (index / Wo) % Ho;
        int c = (index / Wo / Ho) % offset_channels;
        const int g = c / (2 * kk);
        const int col_step = kk;
        int cnt = 0;
        const float* col_ptr = col.data() + (size_t)g * cpg * Wo * Ho;
        const float* im_ptr = in_n + (size_t)g * cpg / kh / kw * H * W;
        const float* off_ptr = off_n + (size_t)g * 2 * kk * P;
        const float* mask_ptr = mask_n + (size_t)g * kk * P;
        const int offset_c = c - g * 2 * kk;
        for (int col_c = (offset_c / 2); col_c < cpg; col_c += col_step) {
          const long col_pos = (((long)col_c * Ho) + h) * Wo + w;
          const int bp_dir = offset_c % 2;
          int j = (col_pos / Wo / Ho) % kw;
          int i = (col_pos / Wo / Ho / kw) % kh;
          int w_out = col_pos % Wo;
          int h_out = (col_pos / Wo) % Ho;
          int w_in = w_out * stride_w - pad_w;
          int h_in = h_out * stride_h - pad_h;
          const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
          const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
          const float m = mask_ptr[((i * kw + j) * Ho + h_out) * Wo + w_out];
          float inv_h = h_in + i * dil_h + offset_h;
          float inv_w = w_in + j * dil_w + offset_w;
          if (inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W) {
            inv_h = inv_w = -2;
          } else {
            mval += col_ptr[col_pos] * dcn_bilinear(im_ptr + (size_t)cnt * H * W, W, H, W, inv_h, inv_w);
          }
          const float weight_c = dcn_coordinate_weight(inv_h, inv_w, H, W, im_ptr + (size_t)cnt * H * W, W, bp_dir);
          val += weight_c * col_ptr[col_pos] * m;
          cnt += 1;
        }
        goff_n[index] = val;
        if (offset_c % 2 == 0) gmask_n[(((size_t)g * kk + offset_c / 2) * Ho + h) * Wo + w] = mval;
      }
    }
    // modulated_deformable_col2im_gpu_kernel L506-558, called with (pad_h, pad_h) (L651-653)
    {
      const int cpg = C / dg;
      const int pad_w_used = pad_h;
      const long n = (long)C * kk * Ho * Wo;
      for (long index = 0; index < n; index++) {
        const int j = (index / Wo / Ho) % kw;
        const int i = (index / Wo / Ho / kw) % kh;
        const int c = index / Wo / Ho / kw / kh;
        const int g = c / cpg;
        int w_out = index % Wo;
        int h_out = (index / Wo) % Ho;
        int w_in = w_out * stride_w - pad_w_used;
        int h_in = h_out * stride_h - pad_h;
        const float* off_ptr = off_n + (size_t)g * 2 * kk * P;
        const float* mask_ptr = mask_n + (size_t)g * kk * P;
        const float offset_h = off_ptr[((2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
        const float offset_w = off_ptr[((2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
        const float m = mask_ptr[((i * kw + j) * Ho + h_out) * Wo + w_out];
        const float cur_inv_h = h_in + i * dil_h + offset_h;
        const float cur_inv_w = w_in + j * dil_w + offset_w;
        const float cur_top_grad = col[index] * m;
        const int cur_h = (int)cur_inv_h;
        const int cur_w = (int)cur_inv_w;
        for (int dy = -2; dy <= 2; dy++)
          for (int dx = -2; dx <= 2; dx++)
            if (cur_h + dy >= 0 && cur_h + dy < H && cur_w + dx >= 0 && cur_w + dx < W &&
                fabsf(cur_inv_h - (cur_h + dy)) < 1 && fabsf(cur_inv_w - (cur_w + dx)) < 1) {
              size_t pos = ((size_t)c * H + cur_h + dy) * W + cur_w + dx;
              float wgt = dcn_gradient_weight(cur_inv_h, cur_inv_w, cur_h + dy, cur_w + dx, H, W);
              gi_n[pos] += wgt * cur_top_grad;
            }
      }
    }
    // grad_weight += grad_output . columns(input)^T, grad_bias += grad_output . ones   (L746-776)
    dcn2_im2col_image(in_n, off_n, mask_n, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dg, Ho,
                      Wo, col.data());
#pragma omp parallel for schedule(static)
    for (int co = 0; co < Cout; co++) {
      for (int k = 0; k < K; k++) {
        double acc = 0;
        for (int q = 0; q < P; q++) acc += (double)go_n[(size_t)co * P + q] * col[(size_t)k * P + q];
        gw[(size_t)co * K + k] += acc;
      }
      double sb = 0;
      for (int q = 0; q < P; q++) sb += go_n[(size_t)co * P + q];
      gb[co] += sb;
    }
  }
  for (size_t i = 0; i < gw.size(); i++) grad_weight[i] = (float)gw[i];
  for (int co = 0; co < Cout; co++) grad_bias[co] = (float)gb[co];
}

// ---------------------------------------------------------------------------
// Deformable PSRoI pooling (ops/dcn_v2.py:L832-932 forward, L1007-1116 backward).  Parity status: UNPINNED by
// reference execution (CUDA only); closed forms in tests/test_dcn_v2_oracle.py (affine map, adjoint, differences).
// ---------------------------------------------------------------------------
namespace {

struct PsBin {
  int n, ctop, ph, pw, batch, class_id, part_h, part_w, gh, gw;
  float roi_w, roi_h, wstart, hstart, sub_w, sub_h;
};

// the per-element preamble shared by L866-902 and L1031-1062
PsBin ps_bin(long index, const float* rois, const float* trans, int no_trans, float spatial_scale, int output_dim,
             int group_size, int P, int part_size, int spp, float trans_std, int num_classes, int ch_each_class) {
  PsBin b;
  b.pw = index % P;
  b.ph = (index / P) % P;
  b.ctop = (index / P / P) % output_dim;
  b.n = index / P / P / output_dim;
  const float* r = rois + (size_t)b.n * 5;
  b.batch = (int)r[0];
  float roi_start_w = static_cast<float>(roundf(r[1])) * spatial_scale - 0.5;
  float roi_start_h = static_cast<float>(roundf(r[2])) * spatial_scale - 0.5;
  float roi_end_w = static_cast<float>(roundf(r[3]) + 1.) * spatial_scale - 0.5;
  float roi_end_h = static_cast<float>(roundf(r[4]) + 1.) * spatial_scale - 0.5;
  b.roi_w = std::max((double)(roi_end_w - roi_start_w), 0.1);
  b.roi_h = std::max((double)(roi_end_h - roi_start_h), 0.1);
  float bin_size_h = b.roi_h / static_cast<float>(P);
  float bin_size_w = b.roi_w / static_cast<float>(P);
  b.sub_h = bin_size_h / static_cast<float>(spp);
  b.sub_w = bin_size_w / static_cast<float>(spp);
  b.part_h = (int)floorf(static_cast<float>(b.ph) / P * part_size);
  b.part_w = (int)floorf(static_cast<float>(b.pw) / P * part_size);
  b.class_id = b.ctop / ch_each_class;
  float trans_x = no_trans ? 0.f
                           : trans[((((size_t)b.n * num_classes + b.class_id) * 2) * part_size + b.part_h) * part_size +
                                   b.part_w] * trans_std;
  float trans_y = no_trans ? 0.f
                           : trans[((((size_t)b.n * num_classes + b.class_id) * 2 + 1) * part_size + b.part_h) *
                                       part_size + b.part_w] * trans_std;
  b.wstart = static_cast<float>(b.pw) * bin_size_w + roi_start_w;
  b.wstart += trans_x * b.roi_w;
  b.hstart = static_cast<float>(b.ph) * bin_size_h + roi_start_h;
  b.hstart += trans_y * b.roi_h;
  int gw = (int)floorf(static_cast<float>(b.pw) * group_size / P);
  int gh = (int)floorf(static_cast<float>(b.ph) * group_size / P);
  b.gw = std::min(std::max(gw, 0), group_size - 1);
  b.gh = std::min(std::max(gh, 0), group_size - 1);
  return b;
}

}  // namespace

JO_API void jo_deform_psroi_forward(const float* input, const float* rois, const float* trans, int N, int C, int H,
                                    int W, int R, int no_trans, float spatial_scale, int output_dim, int group_size, int P,
                                    int part_size, int spp, float trans_std, int trans_channels, float* out,
                                    float* top_count) {
  const int num_classes = no_trans ? 1 : trans_channels / 2;
  const int cec = no_trans ? output_dim : output_dim / num_classes;
  const long count = (long)R * output_dim * P * P;
  for (long index = 0; index < count; index++) {
    const PsBin b = ps_bin(index, rois, trans, no_trans, spatial_scale, output_dim, group_size, P, part_size, spp,
                           trans_std, num_classes, cec);
    float sum = 0;
    int cnt = 0;
    if (b.batch < 0 || b.batch >= N) {   // (out-of-bounds read in the reference; here: nothing pooled)
      out[index] = 0.f;
      top_count[index] = 0;
      continue;
    }
    const float* data = input + (size_t)b.batch * C * H * W;
    for (int ih = 0; ih < spp; ih++)
      for (int iw = 0; iw < spp; iw++) {
        float w = b.wstart + iw * b.sub_w;
        float h = b.hstart + ih * b.sub_h;
        if (w < -0.5 || w > W - 0.5 || h < -0.5 || h > H - 0.5) continue;
        w = std::min(std::max((double)w, 0.), W - 1.);
        h = std::min(std::max((double)h, 0.), H - 1.);
        int c = (b.ctop * group_size + b.gh) * group_size + b.gw;
        const float* plane = data + (size_t)c * H * W;
        // bilinear_interp L832-854
        int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        float dist_x = static_cast<float>(w - x1), dist_y = static_cast<float>(h - y1);
        float value11 = plane[y1 * W + x1], value12 = plane[y2 * W + x1];
        float value21 = plane[y1 * W + x2], value22 = plane[y2 * W + x2];
        float value = (1 - dist_x) * (1 - dist_y) * value11 + (1 - dist_x) * dist_y * value12 +
                      dist_x * (1 - dist_y) * value21 + dist_x * dist_y * value22;
        sum += value;
        cnt++;
      }
    out[index] = cnt == 0 ? 0.f : sum / cnt;
    top_count[index] = cnt;
  }
}

JO_API void jo_deform_psroi_backward(const float* top_diff, const float* top_count, const float* input,
                                     const float* rois, const float* trans, int N, int C, int H, int W, int R,
                                     int no_trans, float spatial_scale, int output_dim, int group_size, int P,
                                     int part_size, int spp, float trans_std, int trans_channels, float* grad_input,
                                     float* grad_trans) {
  const int num_classes = no_trans ? 1 : trans_channels / 2;
  const int cec = no_trans ? output_dim : output_dim / num_classes;
  memset(grad_input, 0, sizeof(float) * (size_t)N * C * H * W);
  if (!no_trans) memset(grad_trans, 0, sizeof(float) * (size_t)R * trans_channels * part_size * part_size);
  const long count = (long)R * output_dim * P * P;
  for (long index = 0; index < count; index++) {
    if (top_count[index] <= 0) continue;
    const PsBin b = ps_bin(index, rois, trans, no_trans, spatial_scale, output_dim, group_size, P, part_size, spp,
                           trans_std, num_classes, cec);
    float diff_val = top_diff[index] / top_count[index];
    const float* data = input + (size_t)b.batch * C * H * W;
    float* gdata = grad_input + (size_t)b.batch * C * H * W;
    for (int ih = 0; ih < spp; ih++)
      for (int iw = 0; iw < spp; iw++) {
        float w = b.wstart + iw * b.sub_w;
        float h = b.hstart + ih * b.sub_h;
        if (w < -0.5 || w > W - 0.5 || h < -0.5 || h > H - 0.5) continue;
        w = std::min(std::max((double)w, 0.), W - 1.);
        h = std::min(std::max((double)h, 0.), H - 1.);
        int c = (b.ctop * group_size + b.gh) * group_size + b.gw;
        int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        float dist_x = w - x0, dist_y = h - y0;
        float q00 = (1 - dist_x) * (1 - dist_y), q01 = (1 - dist_x) * dist_y;
        float q10 = dist_x * (1 - dist_y), q11 = dist_x * dist_y;
        size_t base = (size_t)c * H * W;
        gdata[base + y0 * W + x0] += q00 * diff_val;
        gdata[base + y1 * W + x0] += q01 * diff_val;
        gdata[base + y0 * W + x1] += q10 * diff_val;
        gdata[base + y1 * W + x1] += q11 * diff_val;
        if (no_trans) continue;
        float U00 = data[base + y0 * W + x0], U01 = data[base + y1 * W + x0];
        float U10 = data[base + y0 * W + x1], U11 = data[base + y1 * W + x1];
        float diff_x = (U11 * dist_y + U10 * (1 - dist_y) - U01 * dist_y - U00 * (1 - dist_y)) * trans_std * diff_val;
        diff_x *= b.roi_w;
        float diff_y = (U11 * dist_x + U01 * (1 - dist_x) - U10 * dist_x - U00 * (1 - dist_x)) * trans_std * diff_val;
        diff_y *= b.roi_h;
        grad_trans[((((size_t)b.n * num_classes + b.class_id) * 2) * part_size + b.part_h) * part_size + b.part_w] +=
            diff_x;
        grad_trans[((((size_t)b.n * num_classes + b.class_id) * 2 + 1) * part_size + b.part_h) * part_size +
                   b.part_w] += diff_y;
      }
  }
}

// ---------------------------------------------------------------------------
// RepPoints geometry: convex IoU, minimum-area rectangle (CUDA-only reference text restated statement by statement),
// and the Graham scan of convex_sort.  Parity status: UNPINNED by reference execution; pinned by exact geometry in
// tests/test_convex_oracle.py (hulls of known point sets, analytic overlaps, rectangles of rotated boxes,
// scipy.spatial.ConvexHull).
// ---------------------------------------------------------------------------
namespace cvx {

const double eps = 1E-8;
const int maxn = 100;

inline int sig(double d) { return int(d > eps) - int(d < -eps); }

struct Point {
  double x, y;
  Point() {}
  Point(double x, double y) : x(x), y(y) {}
};

inline bool point_same(Point& a, Point& b) { return sig(a.x - b.x) == 0 && sig(a.y - b.y) == 0; }
inline void swap1(Point* a, Point* b) {
  Point temp = *a;
  *a = *b;
  *b = temp;
}
inline void reverse1(Point* a, const int n) {
  Point temp[maxn];
  for (int i = 0; i < n; i++) temp[i] = a[i];
  for (int i = 0; i < n; i++) a[i] = temp[n - 1 - i];
}
inline double cross(Point o, Point a, Point b) { return (a.x - o.x) * (b.y - o.y) - (b.x - o.x) * (a.y - o.y); }
inline double dis(Point a, Point b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y); }
inline double area(Point* ps, int n) {
  ps[n] = ps[0];
  double res = 0;
  for (int i = 0; i < n; i++) res += ps[i].x * ps[i + 1].y - ps[i].y * ps[i + 1].x;
  return res / 2.0;
}
inline int lineCross(Point a, Point b, Point c, Point d, Point& p) {
  double s1 = cross(a, b, c), s2 = cross(a, b, d);
  if (sig(s1) == 0 && sig(s2) == 0) return 2;
  if (sig(s2 - s1) == 0) return 0;
  p.x = (c.x * s2 - d.x * s1) / (s2 - s1);
  p.y = (c.y * s2 - d.y * s1) / (s2 - s1);
  return 1;
}
// convex_iou_kernel.cu:L82-110
inline void polygon_cut(Point* p, int& n, Point a, Point b) {
  Point pp[maxn];
  for (int i = 0; i < maxn; i++) pp[i] = Point(0, 0);   // (device stack garbage in the reference)
  int m = 0;
  p[n] = p[0];
  for (int i = 0; i < n; i++) {
    if (sig(cross(a, b, p[i])) > 0) {
      pp[m] = p[i];
      m++;
    }
    if (sig(cross(a, b, p[i])) != sig(cross(a, b, p[i + 1]))) {
      lineCross(a, b, p[i], p[i + 1], pp[m]);
      m++;
    }
  }
  n = 0;
  for (int i = 0; i < m; i++)
    if (!i || !(point_same(pp[i], pp[i - 1]))) {
      p[n] = pp[i];
      n++;
    }
  while (n > 1 && point_same(p[n - 1], p[0])) n--;
}
// L113-140
inline double intersectArea(Point a, Point b, Point c, Point d) {
  Point o(0, 0);
  int s1 = sig(cross(o, a, b));
  int s2 = sig(cross(o, c, d));
  if (s1 == 0 || s2 == 0) return 0.0;
  if (s1 == -1) swap1(&a, &b);
  if (s2 == -1) swap1(&c, &d);
  Point p[10] = {o, a, b};
  int n = 3;
  polygon_cut(p, n, o, c);
  polygon_cut(p, n, c, d);
  polygon_cut(p, n, d, o);
  double res = area(p, n);
  if (s1 * s2 == -1) res = -res;
  return res;
}
// L141-155
inline double intersectAreaO(Point* ps1, int n1, Point* ps2, int n2) {
  if (area(ps1, n1) < 0) reverse1(ps1, n1);
  if (area(ps2, n2) < 0) reverse1(ps2, n2);
  ps1[n1] = ps1[0];
  ps2[n2] = ps2[0];
  double res = 0;
  for (int i = 0; i < n1; i++)
    for (int j = 0; j < n2; j++) res += intersectArea(ps1[i], ps1[i + 1], ps2[j], ps2[j + 1]);
  return res;
}

// Jarvis_and_index, convex_iou_kernel.cu:L157-256 (T = double) and min_area_bbox.cu:L205-299 (T = float: points and
// distances in float, the turn test in double).  The index output is not needed by either caller restated here.
template <typename T>
struct PT {
  T x, y;
};
template <typename T>
inline T disT(PT<T> a, PT<T> b) { return (a.x - b.x) * (a.x - b.x) + (a.y - b.y) * (a.y - b.y); }

template <typename T>
void Jarvis(PT<T>* in_poly, int& n_poly) {
  PT<T> p_max = in_poly[0], p_k;
  int max_index = 0, k_index;
  int Stack[40], top1, top2;
  double sign;
  PT<T> right_point[40], left_point[40];
  for (int i = 0; i < n_poly; i++) {
    if (in_poly[i].y < in_poly[0].y || (in_poly[i].y == in_poly[0].y && in_poly[i].x < in_poly[0].x))
      std::swap(in_poly[0], in_poly[i]);
    if (i == 0) {
      p_max = in_poly[0];
      max_index = 0;
    }
    if (in_poly[i].y > p_max.y || (in_poly[i].y == p_max.y && in_poly[i].x > p_max.x)) {
      p_max = in_poly[i];
      max_index = i;
    }
  }
  if (max_index == 0) {
    max_index = 1;
    p_max = in_poly[max_index];
  }
  k_index = 0, Stack[0] = 0, top1 = 0;
  while (k_index != max_index && top1 < 18) {      // (cap: the reference loops unbounded)
    p_k = p_max;
    k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      const PT<T> s = in_poly[Stack[top1]];
      sign = ((double)in_poly[i].x - (double)s.x) * ((double)p_k.y - (double)s.y) -
             ((double)p_k.x - (double)s.x) * ((double)in_poly[i].y - (double)s.y);
      if ((sign > 0) || ((sign == 0) && (disT(s, in_poly[i]) > disT(s, p_k)))) {
        p_k = in_poly[i];
        k_index = i;
      }
    }
    top1++;
    Stack[top1] = k_index;
  }
  for (int i = 0; i <= top1; i++) right_point[i] = in_poly[Stack[i]];
  k_index = 0, Stack[0] = 0, top2 = 0;
  while (k_index != max_index && top2 < 18) {
    p_k = p_max;
    k_index = max_index;
    for (int i = 1; i < n_poly; i++) {
      const PT<T> s = in_poly[Stack[top2]];
      sign = ((double)in_poly[i].x - (double)s.x) * ((double)p_k.y - (double)s.y) -
             ((double)p_k.x - (double)s.x) * ((double)in_poly[i].y - (double)s.y);
      if ((sign < 0) || ((sign == 0) && (disT(s, in_poly[i]) > disT(s, p_k)))) {
        p_k = in_poly[i];
        k_index = i;
      }
    }
    top2++;
    Stack[top2] = k_index;
  }
  for (int i = top2 - 1; i >= 0; i--) left_point[i] = in_poly[Stack[i]];
  int total = top1 + top2;
  if (total > 19) total = 19;
  for (int i = 0; i < total; i++) {
    if (i <= top1)
      in_poly[i] = right_point[i];
    else
      in_poly[i] = left_point[top2 - (i - top1)];
  }
  n_poly = total;
}

}  // namespace cvx

// convex_iou_kernel.cu:L258-305: ious (N, M); pointsets (N, 18), polygons (M, 8)
JO_API void jo_convex_iou(const float* pointsets, int N, const float* polygons, int M, float* ious) {
  using namespace cvx;
  for (int a = 0; a < N; a++)
    for (int b = 0; b < M; b++) {
      const float* p = pointsets + (size_t)a * 18;
      const float* q = polygons + (size_t)b * 8;
      Point ps1[maxn], ps2[maxn];
      PT<double> convex[maxn];
      for (int i = 0; i < 9; i++) {
        convex[i].x = (double)p[i * 2];
        convex[i].y = (double)p[i * 2 + 1];
      }
      int n_convex = 9;
      Jarvis<double>(convex, n_convex);
      int n1 = n_convex;
      for (int i = 0; i < n1; i++) ps1[i] = Point(convex[i].x, convex[i].y);
      int n2 = 4;
      for (int i = 0; i < n2; i++) ps2[i] = Point((double)q[i * 2], (double)q[i * 2 + 1]);
      double inter_area = intersectAreaO(ps1, n1, ps2, n2);
      double S_pred = area(ps1, n1);
      double union_area = fabs(S_pred) + fabs(area(ps2, n2)) - inter_area;
      ious[(size_t)a * M + b] = (float)(inter_area / union_area);
    }
}

// the hull alone (test aid): hull (N, 9, 2) padded with NaN, counts (N)
JO_API void jo_convex_hull9(const float* pointsets, int N, float* hull, int* counts) {
  for (int a = 0; a < N; a++) {
    cvx::PT<double> c[40];
    for (int i = 0; i < 9; i++) c[i] = {(double)pointsets[(size_t)a * 18 + 2 * i], (double)pointsets[(size_t)a * 18 + 2 * i + 1]};
    int n = 9;
    cvx::Jarvis<double>(c, n);
    counts[a] = n;
    for (int i = 0; i < 9; i++) {
      hull[((size_t)a * 9 + i) * 2] = i < n ? (float)c[i].x : NAN;
      hull[((size_t)a * 9 + i) * 2 + 1] = i < n ? (float)c[i].y : NAN;
    }
  }
}

// min_area_bbox.cu:L49-203 (minBoundingRect), L301-399 (Findminbox): bboxes (N, 8)
JO_API void jo_min_area_bbox(const float* pointsets, int N, float* bboxes) {
  const int maxn = 20;
  for (int a = 0; a < N; a++) {
    const float* p = pointsets + (size_t)a * 18;
    float* minpoints = bboxes + (size_t)a * 8;
    cvx::PT<float> convex[40], ps[40];
    float pi = 3.1415926;
    for (int i = 0; i < 9; i++) {
      convex[i].x = p[i * 2];
      convex[i].y = p[i * 2 + 1];
    }
    int n_convex = 9;
    cvx::Jarvis<float>(convex, n_convex);
    int n1 = n_convex;
    for (int i = 0; i < n1; i++) ps[i] = convex[i];
    ps[n1] = convex[0];
    // ---- minBoundingRect(ps, n1 + 1, minbbox)
    const int n_points = n1 + 1;
    float minbox[5] = {0};
    {
      float convex_points[2][maxn];
      for (int j = 0; j < n_points; j++) convex_points[0][j] = ps[j].x;
      for (int j = 0; j < n_points; j++) convex_points[1][j] = ps[j].y;
      cvx::PT<float> edges[maxn];
      float edges_angles[maxn];
      float unique_angles[maxn];
      int n_edges = n_points - 1;
      int n_unique = 0;
      int unique_flag = 0;
      for (int i = 0; i < n_edges; i++) {
        edges[i].x = ps[i + 1].x - ps[i].x;
        edges[i].y = ps[i + 1].y - ps[i].y;
      }
      for (int i = 0; i < n_edges; i++) {
        edges_angles[i] = atan2((double)edges[i].y, (double)edges[i].x);
        if (edges_angles[i] >= 0)
          edges_angles[i] = fmod((double)edges_angles[i], (double)pi / 2);
        else
          edges_angles[i] = edges_angles[i] - (int)(edges_angles[i] / (pi / 2) - 1) * (pi / 2);
      }
      unique_angles[0] = edges_angles[0];
      n_unique += 1;
      for (int i = 1; i < n_edges; i++) {
        for (int j = 0; j < n_unique; j++)
          if (edges_angles[i] == unique_angles[j]) unique_flag += 1;
        if (unique_flag == 0) {
          unique_angles[n_unique] = edges_angles[i];
          n_unique += 1;
          unique_flag = 0;
        } else {
          unique_flag = 0;
        }
      }
      float minarea = 1e12;
      for (int i = 0; i < n_unique; i++) {
        float R[2][2];
        float rot_points[2][maxn];
        R[0][0] = cosf(unique_angles[i]);
        R[0][1] = cosf(unique_angles[i] - pi / 2);
        R[1][0] = cosf(unique_angles[i] + pi / 2);
        R[1][1] = cosf(unique_angles[i]);
        for (int m = 0; m < 2; m++)
          for (int n = 0; n < n_points; n++) {
            float sum = 0.0;
            for (int k = 0; k < 2; k++) sum = sum + R[m][k] * convex_points[k][n];
            rot_points[m][n] = sum;
          }
        float xmin = 1e12, ymin = 1e12, xmax = -1e12, ymax = -1e12;
        for (int j = 0; j < n_points; j++) {
          if (!(std::isinf(rot_points[0][j]) || std::isnan(rot_points[0][j]))) {
            if (rot_points[0][j] < xmin) xmin = rot_points[0][j];
            if (rot_points[0][j] > xmax) xmax = rot_points[0][j];
          }
          if (!(std::isinf(rot_points[1][j]) || std::isnan(rot_points[1][j]))) {
            if (rot_points[1][j] < ymin) ymin = rot_points[1][j];
            if (rot_points[1][j] > ymax) ymax = rot_points[1][j];
          }
        }
        float area = (xmax - xmin) * (ymax - ymin);
        if (area < minarea) {
          minarea = area;
          minbox[0] = unique_angles[i];
          minbox[1] = xmin;
          minbox[2] = ymin;
          minbox[3] = xmax;
          minbox[4] = ymax;
        }
      }
    }
    float angle = minbox[0], xmin = minbox[1], ymin = minbox[2], xmax = minbox[3], ymax = minbox[4];
    float R[2][2];
    R[0][0] = cosf(angle);
    R[0][1] = cosf(angle - pi / 2);
    R[1][0] = cosf(angle + pi / 2);
    R[1][1] = cosf(angle);
    const float corner[4][2] = {{xmax, ymin}, {xmin, ymin}, {xmin, ymax}, {xmax, ymax}};
    for (int c = 0; c < 4; c++)
      for (int n = 0; n < 2; n++) {
        float sum = 0.0;
        for (int k = 0; k < 2; k++) sum = sum + corner[c][k] * R[k][n];
        minpoints[c * 2 + n] = sum;
      }
  }
}

// convex_sort.py:L159-194 (start index, order) + L4-65 (scan): index (nbs, npts + circular), -1 filled
JO_API void jo_convex_sort(const float* pts, const float* masks, int nbs, int npts, int circular, int* index) {
  const int index_size = circular ? npts + 1 : npts;
  for (int i = 0; i < nbs * index_size; i++) index[i] = -1;
  if (npts == 0) return;
  std::vector<float> cosv(npts);
  std::vector<int> order(npts);
  for (int b = 0; b < nbs; b++) {
    const float* px = pts + (size_t)b * npts * 2;
    const float* m = masks + (size_t)b * npts;
    int start = 0;
    float best = 0;
    for (int i = 0; i < npts; i++) {
      float masked_y = m[i] * px[2 * i + 1] + (1 - m[i]) * 10000000.f;
      if (i == 0 || masked_y < best) {
        best = masked_y;
        start = i;
      }
    }
    const float sx = px[2 * start], sy = px[2 * start + 1];
    for (int i = 0; i < npts; i++) {
      float dx = px[2 * i] - sx, dy = px[2 * i + 1] - sy;
      cosv[i] = dx / sqrtf(dx * dx + dy * dy + 0.000001f);
      order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return cosv[a] > cosv[c]; });
    int* sub = index + (size_t)b * index_size;
    sub[0] = start;
    int c_i = 0;
    for (int _j = 0; _j < npts; _j++) {
      const int j = order[_j];
      if (j == start) continue;
      if (m[j] < 0.5) continue;
      const float x0 = px[2 * j], y0 = px[2 * j + 1];
      float x1 = px[2 * sub[c_i]], y1 = px[2 * sub[c_i] + 1];
      float d = (x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0);
      if (d < 0.000001) continue;
      if (c_i < 2) {
        sub[++c_i] = j;
      } else {
        float x2 = px[2 * sub[c_i - 1]], y2 = px[2 * sub[c_i - 1] + 1];
        while (1) {
          float t = (x1 - x2) * (y0 - y2) - (y1 - y2) * (x0 - x2);
          if (t >= 0) {
            sub[++c_i] = j;
            break;
          } else {
            if (c_i <= 1) {
              sub[c_i] = j;
              break;
            } else {
              c_i--;
              x1 = px[2 * sub[c_i]];
              y1 = px[2 * sub[c_i] + 1];
              x2 = px[2 * sub[c_i - 1]];
              y2 = px[2 * sub[c_i - 1] + 1];
            }
          }
        }
      }
    }
    if (circular) sub[++c_i] = sub[0];
  }
}

// ---------------------------------------------------------------------------
// Active rotating filter (ops/orn.py:L138-211 CPU kernels)
//   weight (nOut, nIn, nOri, kH, kW) ; indices (nOri, kH, kW, nRot) uint8, 1-based
//   output (nOut*nRot, nIn*nOri, kH, kW)
// ---------------------------------------------------------------------------
JO_API void jo_arf_forward(const float* weight, const uint8_t* indices, int nOut, int nIn,
                           int nOri, int kH, int kW, int nRot, float* out) {
  const int nEntry = nOri * kH * kW;
  for (int i = 0; i < nOut; i++)
    for (int j = 0; j < nIn; j++)
      for (int l = 0; l < nEntry; l++) {
        float val = weight[((size_t)i * nIn + j) * nEntry + l];
        for (int k = 0; k < nRot; k++) {
          int index = (int)indices[l * nRot + k] - 1;
          out[(size_t)i * (nRot * nIn * nEntry) + (size_t)k * (nIn * nEntry) + j * nEntry + index] =
              val;
        }
      }
}

JO_API void jo_arf_backward(const uint8_t* indices, const float* grad_out, int nOut, int nIn,
                            int nOri, int kH, int kW, int nRot, float* grad_w) {
  const int nEntry = nOri * kH * kW;
  for (int i = 0; i < nOut; i++)
    for (int j = 0; j < nIn; j++)
      for (int l = 0; l < nEntry; l++) {
        float* val = grad_w + ((size_t)i * nIn + j) * nEntry + l;
        *val = 0;
        for (int k = 0; k < nRot; k++) {
          int index = (int)indices[l * nRot + k] - 1;
          *val = *val + grad_out[(size_t)i * (nRot * nIn * nEntry) + (size_t)k * (nIn * nEntry) +
                                 j * nEntry + index];
        }
      }
}

JO_API int jo_version(void) { return 1; }
