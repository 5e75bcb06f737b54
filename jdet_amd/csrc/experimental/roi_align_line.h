// RoIAlign forward, channels-last out, sample_num == 2: taps deduplicated over a whole LINE of bins (round 4).
//
// Reference semantics: python/jdet/ops/roi_align_rotated.py:L61-127 (and the _v1 / horizontal twins); values equal the
// merged-tap kernel's up to the order of a bin's weight sums (<= 2e-6 on N(0,1) maps, same tests).
//
// The merged-tap kernel loads, per bin, every distinct pixel row its 4 samples touch: 876 k row loads of 1 KiB at the
// north-star point.  But neighbouring bins of a thin RoI sit on the same pixels: over a whole line of bins -- a bin row
// when the bins are narrower than tall, a bin column otherwise -- the distinct rows are 511 k (0.58 x; the whole RoI:
// 454 k), and the time of the forward follows the rows through the vector L1 first (profiles/r04_roi_fwd_notes.md).
// So here a wave owns a LINE: 7 accumulators (one per bin of the line) live in registers, every distinct pixel row of
// the line is loaded once and multiplied into all 7 with its per-bin weights (mostly zeros: dense FMAs are cheaper
// than a branch per weight).
//   1  lane = sample: geometry, then the 16 taps of a bin merged inside its quad of lanes (as the merged-tap kernel)
//   2  the surviving (pixel, weight) entries go into the line's open-addressing table in LDS: atomicCAS on the pixel
//      key finds / claims the slot, the weight is a plain store to (slot, bin of the line) -- one writer per cell,
//      so the result is deterministic
//   3  per line, one wave compacts the table in place (ballot + rank) and walks it: key and the 8 weights of an entry
//      come back as LDS broadcasts (no readlane traffic), 4 row loads in flight, 7 x 2 packed FMAs per row
//   4  7 non-temporal 1 KiB stores per line
// Measured (profiles/r04_roi_fwd_notes.md, step 2): rows through the L1 -45 %, L2 requests -37 %, but 71 us against 58 us
// for the merged-tap kernel: the set-up (table init + hash inserts behind the double-precision trig: 11.7 us per
// workgroup, every wave waiting) and 14 packed FMAs + 3 LDS reads per row cost more than the saved rows.  Forward
// jdet_roi_align_forward_cl_mode(3, ...) of libjdet_experimental.so runs it; not a product path.
// (Included by roi_align_impl.inc (JDET_ROI_EXPERIMENTAL_MODES) inside its unnamed namespace, after the merged-tap kernel.)
#pragma once

constexpr int kLineSlots = 128;          // >= 8 bins x 16 taps; a power of two (hash: 7 bits)
constexpr int kLineMaxBins = 8;          // bins per line, lines per RoI

template <int VARIANT, int NPER, int BATCH>
__global__ __launch_bounds__(256) void roi_align_fwd_line_kernel(
    const float* __restrict__ feat, const float* __restrict__ rois, float* __restrict__ out, int C, int H, int W,
    int PH, int PW, float spatial_scale, const int32_t* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char s_line[];     // [lines][slots] keys | [lines][slots][8] weights
  __shared__ float s_trig[2];
  int* s_key = reinterpret_cast<int*>(s_line);
  float* s_w = reinterpret_cast<float*>(s_line + kLineMaxBins * kLineSlots * 4);
  const int r = order ? order[blockIdx.x] : blockIdx.x;
  const int c0 = blockIdx.y * kChunkC;
  const int cc = min(kChunkC, C - c0);
  const int nbins = PH * PW;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  constexpr int ROI_COLS = (VARIANT == JDET_ROI_HBB_V0 || VARIANT == JDET_ROI_HBB_V1) ? 5 : 6;
  constexpr bool kRot = ROI_COLS == 6;
  const float* roi = rois + (size_t)r * ROI_COLS;
  if (kRot && threadIdx.x == 0) {
    s_trig[0] = (float)cos((double)roi[5]);
    s_trig[1] = (float)sin((double)roi[5]);
  }
  __amdgpu_buffer_rsrc_t rsrc;
  RoiGeom g = vec_prologue<VARIANT, false>(feat, rois, r, C, H, W, PH, PW, spatial_scale, 2, rsrc);
  if (g.batch < 0) return;
  // tables: keys -1, weights 0
  {
    v4f* z = reinterpret_cast<v4f*>(s_w);
    for (int i = threadIdx.x; i < kLineMaxBins * kLineSlots * 2; i += 256) z[i] = v4f{0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < kLineMaxBins * kLineSlots; i += 256) s_key[i] = -1;
  }
  __syncthreads();
  if (kRot) {
    g.cosT = s_trig[0];
    g.sinT = s_trig[1];
  }
  // lines run along the direction in which neighbouring bins are closer together
  const bool rows = g.bin_w <= g.bin_h;                      // line = bin row (ph), position in the line = pw
  const int nlines = rows ? PH : PW, nper = rows ? PW : PH;

  // ---- 1: lane = sample (thread t: bin t / 4, sample t % 4) ----
  const int q = lane & 3, qbase = lane & ~3;
  const int bin = threadIdx.x >> 2;
  const bool bin_ok = bin < nbins;
  const int bb = bin_ok ? bin : 0;
  const int ph = bb / PW, pw = bb - ph * PW;
  Sample s = make_sample<VARIANT>(g, ph, pw, q >> 1, q & 1, H, W);
  if (!bin_ok) s.valid = 0;
  const int o[4] = {s.o1, s.o2, s.o3, s.o4};
  const float w[4] = {s.w1, s.w2, s.w3, s.w4};
  float tw[4] = {w[0], w[1], w[2], w[3]};
  bool first[4] = {true, true, true, true};
#pragma unroll
  for (int k = 1; k < 4; k++)
#pragma unroll
    for (int j = 0; j < k; j++)
      if (o[j] == o[k]) {   // x_high == x_low / y_high == y_low at the map border
        tw[j] += w[k];
        first[k] = false;
      }
#pragma unroll
  for (int d = 1; d < 4; d++) {
    const int src = qbase | ((q + d) & 3);
    const bool earlier = ((q + d) & 3) < q;
    const int ov = __shfl(s.valid, src, 64);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int oo = __shfl(o[j], src, 64);
      const float ww = __shfl(w[j], src, 64);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool same = ov && oo == o[k];
        tw[k] += same ? ww : 0.f;
        first[k] = first[k] && !(same && earlier);
      }
    }
  }
  // ---- 2: entries into the line's table ----
  const float inv_count = 1.f / g.count;   // count == 4 here: exact
  const int line = rows ? ph : pw, pos = rows ? pw : ph;
  int* keys = s_key + line * kLineSlots;
  float* wts = s_w + (size_t)line * kLineSlots * 8;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (s.valid && first[k]) {
      const int key = o[k];
      int h = (int)(((unsigned)key * 0x9E3779B1u) >> 25);      // multiplicative hash, top 7 bits
      for (;;) {
        const int prev = atomicCAS(&keys[h], -1, key);
        if (prev == -1 || prev == key) break;
        h = (h + 1) & (kLineSlots - 1);
      }
      wts[h * 8 + pos] = tw[k] * inv_count;
    }
  }
  __syncthreads();

  // ---- 3: one wave per line ----
  const bool lane_ok = lane * 4 < cc;
  const unsigned voff = (unsigned)((c0 + (lane_ok ? lane * 4 : 0)) * 4);
  const unsigned pix_bytes = (unsigned)C * 4u;
  for (int ln = wave; ln < nlines; ln += 4) {
    int* lk = s_key + ln * kLineSlots;
    v4f* lw = reinterpret_cast<v4f*>(s_w + (size_t)ln * kLineSlots * 8);
    // in-place compaction: both halves are read into registers before anything is written
    const int k0 = lk[lane], k1 = lk[64 + lane];
    const v4f a0 = lw[lane * 2], b0 = lw[lane * 2 + 1], a1 = lw[(64 + lane) * 2], b1 = lw[(64 + lane) * 2 + 1];
    const unsigned long long m0 = __ballot(k0 >= 0), m1 = __ballot(k1 >= 0);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int n0 = __popcll(m0), n = n0 + __popcll(m1);
    __builtin_amdgcn_wave_barrier();
    if (k0 >= 0) {
      const int p = __popcll(m0 & below);
      lk[p] = k0;
      lw[p * 2] = a0;
      lw[p * 2 + 1] = b0;
    }
    if (k1 >= 0) {
      const int p = n0 + __popcll(m1 & below);
      lk[p] = k1;
      lw[p * 2] = a1;
      lw[p * 2 + 1] = b1;
    }
    __builtin_amdgcn_wave_barrier();     // LDS operations of one wave retire in order
    v4f acc[NPER];
#pragma unroll
    for (int b = 0; b < NPER; b++) acc[b] = v4f{0.f, 0.f, 0.f, 0.f};
    // two register sets: the rows of batch k + 1 are requested before the FMAs of batch k
    v4f t[2][BATCH];
    auto request = [&](int i, int set) {
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        const int e = i + u < n ? i + u : n - 1;              // tail: the last row again (a cached line), never used
        const unsigned off = (unsigned)lk[e] * pix_bytes + voff;     // LDS broadcast
        t[set][u] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, off, 0, 0));
      }
    };
    auto consume = [&](int i, int set) {
#pragma unroll
      for (int u = 0; u < BATCH; u++) {
        if (i + u < n) {                                       // wave-uniform
          const v4f wa = lw[(i + u) * 2], wb = lw[(i + u) * 2 + 1];   // LDS broadcast
#pragma unroll
          for (int b = 0; b < NPER; b++) {
            const float wt = b < 4 ? wa[b] : wb[b - 4];
            acc[b].x = __builtin_fmaf(wt, t[set][u].x, acc[b].x);
            acc[b].y = __builtin_fmaf(wt, t[set][u].y, acc[b].y);
            acc[b].z = __builtin_fmaf(wt, t[set][u].z, acc[b].z);
            acc[b].w = __builtin_fmaf(wt, t[set][u].w, acc[b].w);
          }
        }
      }
    };
    if (n > 0) request(0, 0);
    for (int i = 0; i < n; i += 2 * BATCH) {
      if (i + BATCH < n) request(i + BATCH, 1);
      consume(i, 0);
      if (i + 2 * BATCH < n) request(i + 2 * BATCH, 0);
      if (i + BATCH < n) consume(i + BATCH, 1);
    }
    // ---- 4 ----
    if (lane_ok) {
#pragma unroll
      for (int b = 0; b < NPER; b++)
        if (b < nper) {
          const int ob = rows ? ln * PW + b : b * PW + ln;
          __builtin_nontemporal_store(acc[b], reinterpret_cast<v4f*>(out + ((size_t)r * nbins + ob) * C + c0 + lane * 4));
        }
    }
  }
}
