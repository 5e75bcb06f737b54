// fp32 MFMA implicit GEMM for the 3x3 / stride 1 / pad 1 convolutions of the dense stack (FPN output convs, the head
// towers, the RPN conv) and -- with a gathered A operand -- the deformable convolution of AlignConv, channels-last.
//
// Reference modules: nn.Conv (+ ReLU) inside ConvModule (python/jdet/models/utils/modules.py:L91-175), the towers of
// models/roi_heads/s2anet_head.py:L127-205 / rotated_retina_head.py, necks/fpn.py:L150-201; DeformConv =
// im2col -> matmul, ops/dcn_v1.py:L412-454.
//
//   Y[m, n] = sum_{tap, c} A[m, (tap, c)] * Wt[n, (tap, c)]            m = (image, y, x), n = output channel
//   plain conv : A[m, (tap, c)] = X[pixel(m) + tap, c]   (0 outside the image)
//   deformable : A[m, (tap, c)] = bilinear sample of X[., c] at pixel(m) + tap + offset[m, tap]  (dcn_v1.py:L132-166)
//
// Tiling (one workgroup = 4 * KG waves, BT x BT outputs with BT = 128 or 64, K step = BK = 32 or 16 channels of one tap):
//   * v_mfma_f32_32x32x2_f32: the waves form KG groups of 2 x 2; a wave owns (BT/2)^2 outputs = T x T tiles of 32 x 32
//     (T = 2: 64 accumulator registers) and, with KG = 2, every other 8-deep slice of each K step (intra-workgroup
//     split-K: twice the waves per SIMD for the same LDS bytes per MFMA; the two partial tiles meet in LDS at the end).
//     The instruction takes ONE f32 of A and of B per lane (lane l: row / column l & 31, k = l >> 5); which two k of
//     the tile a given instruction consumes is free as long as A and B agree, so lanes 0-31 take k = 8q .. 8q+3 and
//     lanes 32-63 take k = 8q+4 .. 8q+7 of an 8-wide slice: ONE ds_read_b128 per operand tile feeds four MFMAs.
//   * both operand tiles sit in LDS as [row][BK k] = rows of 16-byte chunks, chunk c of row r stored at position
//     c ^ f(r) (BK = 16: f = (r >> 2) & 3, BK = 32: f = (r >> 1) & 7): each 16-lane group of a ds_read_b128 (lanes
//     {0-3, 12-15, 20-27}, ...) and each 8-lane group of a ds_write_b128 covers every bank once -- measured
//     SQ_LDS_BANK_CONFLICT = 0 -- without padding.  The weight tile [n][k] is the weight tensor's own
//     (Cout, 3, 3, Cin) memory order: no transposition anywhere.  Fragment addresses are per-lane constants
//     (the XOR commutes with the slice index), tile / buffer selection goes into the instruction's immediate offset.
//   * global loads are raw buffer loads: per-thread byte offsets fixed per tap (A) / per kernel (B), the K step's
//     channel offset in the scalar offset operand, rows outside the image / past Cout get an out-of-range offset and
//     read as zero -- no address arithmetic, no selects in the K loop.
//   * double-buffered LDS, register-staged: the loads of K step t+1 are issued before the MFMAs of step t and stored
//     to the other buffer after them: one barrier per K step.
//   * workgroup -> tile mapping is XCD-aware: consecutive workgroup ids go round-robin over the 8 XCDs (each with its
//     own L2), so XCD x takes the x-th contiguous band of M tiles with the N tiles of one M tile adjacent: the input
//     rows (incl. the halo shared by neighbouring tiles) of a band stay in one L2.
//   * epilogue in registers: + bias, ReLU, x per-position mask (the gap rows of a LevelPack), 128-byte row segments.
// The matrix pipe is the bound: 2 * M * N * K flop at 157 TFLOP/s (fp32-input MFMA = the fp32 vector rate).
#include <cstdlib>

#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x;        // (N, H, W, Cin)
  const float* w;        // (Cout, 3, 3, Cin)
  const float* bias;     // (Cout) or null
  const float* rowmask;  // (N*H*W) or null: multiplies the finished row (after bias / ReLU)
  const float* offset;   // deformable: (N, 18, H, W) [per tap (dy, dx), dcn_v1.py:L132-140]; null = plain conv
  float* y;              // (N, H, W, Cout)
  float* partial;        // cross-workgroup K split: (ksplit, N*H*W, Cout) partial sums (no epilogue), else null
  int N, H, W, Cin, Cout, relu, ksplit;
};

constexpr unsigned kOob = 0xFFFFFFF0u;   // a byte offset past every buffer: the load returns zeros

__device__ __forceinline__ v4f buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// byte position of chunk (4 floats) `chunk` of LDS row `row` (see header)
template <int BK>
__device__ __forceinline__ int swz_bytes(int row, int chunk) {
  return (row * BK + ((chunk ^ (BK == 16 ? (row >> 2) & 3 : (row >> 1) & 7)) << 2)) * 4;
}

// DEPTH = 2 (round 6; 64 x 64 tiles, one wave group, plain taps): operand tiles requested TWO K steps ahead into two
// register sets -- see conv_bn.hip: a 64 x 64 tile's K step is 0.43 us of MFMAs per wave, less than a global round trip.
template <int BT, int BK, int KG, bool DEFORM, int DEPTH = 1>
__global__ __launch_bounds__(256 * KG)
__attribute__((amdgpu_waves_per_eu(BT == 128 ? (KG == 2 ? 4 : (BK == 32 ? 2 : (DEFORM ? 3 : 4))) : 4)))
void conv3x3_igemm_kernel(ConvArgs a) {
  constexpr int NTHR = 256 * KG;
  constexpr int T = BT / 64;             // 32 x 32 tiles per wave and direction
  constexpr int CH = BK / 4;             // 16-byte chunks per LDS row
  constexpr int RPP = NTHR / CH;         // loader: RPP rows x CH chunks per pass
  constexpr int PASSES = BT / RPP;
  constexpr int NC = DEFORM ? 4 : 1;
  constexpr int TILE = BT * BK * 4;      // bytes of one operand tile
  constexpr int QN = BK / 8 / KG;        // 8-deep slices per wave and K step
  static_assert(PASSES >= 1 && QN >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) char s_raw[4 * TILE];     // [buffer][A | B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long M = (long)a.N * a.H * a.W;
  // ---- XCD-aware tile id ----
  const int NT = (a.Cout + BT - 1) / BT;
  const int total = gridDim.x;
  int logical = blockIdx.x;
  if ((total & 7) == 0) logical = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  // (32-bit unsigned index arithmetic in the prologue -- the host refuses positions * channels >= 2^30; see conv_bn.hip)
  const int mtile = (int)((unsigned)logical / (unsigned)NT);
  const long m0 = (long)mtile * BT;
  const int n0 = (logical - mtile * NT) * BT;
  const __amdgpu_buffer_rsrc_t rx =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (unsigned)(M * a.Cin * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rw =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.w, 0, (unsigned)((long)a.Cout * 9 * a.Cin * 4), 0x00020000);
  // ---- loader role: pass p covers row p * RPP + tid / CH, chunk tid % CH (4 channels of the BK of a K step) ----
  const int lchunk = tid % CH, lrow = tid / CH;
  int img[PASSES], py[PASSES], px[PASSES];
  bool m_ok[PASSES];
  unsigned wv[PASSES];                   // byte offset of this thread's weight chunk at (tap 0, channel 0)
  int st_off[PASSES];                    // byte position of this thread's chunk inside an operand tile
#pragma unroll
  for (int p = 0; p < PASSES; p++) {
    const int row = p * RPP + lrow;
    const long lm = m0 + row;
    m_ok[p] = lm < M;
    img[p] = py[p] = px[p] = 0;
    if (m_ok[p]) {
      const unsigned hw = (unsigned)(a.H * a.W), ulm = (unsigned)lm;
      const unsigned im = ulm / hw, rem = ulm - im * hw, yy = rem / (unsigned)a.W;
      img[p] = (int)im;
      py[p] = (int)yy;
      px[p] = (int)(rem - yy * (unsigned)a.W);
    }
    wv[p] = n0 + row < a.Cout ? ((unsigned)((n0 + row) * 9 * a.Cin + lchunk * 4)) * 4u : kOob;
    st_off[p] = swz_bytes<BK>(row, lchunk);
  }
  const int all_steps = 9 * (a.Cin / BK);   // (host: BK = 32 only when Cin % 32 == 0)
  // cross-workgroup K split: blockIdx.y takes the K steps [step0, step0 + nsteps) and leaves a partial tile
  const int step0 = (int)((unsigned)all_steps * blockIdx.y / (unsigned)a.ksplit);
  const int nsteps = (int)((unsigned)all_steps * (blockIdx.y + 1u) / (unsigned)a.ksplit) - step0;
  const int spt = a.Cin / BK;

  unsigned av[PASSES][NC];               // byte offsets of the tap's source pixel(s), or kOob
  float aw[PASSES][NC];                  // deformable: bilinear weights
  auto set_tap = [&](int tap) {
    const int r = (tap * 11) >> 5, s = tap - r * 3;      // tap / 3 for tap < 9
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
#pragma unroll
      for (int k = 0; k < NC; k++) {
        av[p][k] = kOob;
        aw[p][k] = 0.f;
      }
      if (!DEFORM) {      // branch-free (round 6, as conv_bn.hip's)
        const int yy = py[p] + r - 1, xx = px[p] + s - 1;
        const bool in = m_ok[p] && (unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W;
        const unsigned off = ((unsigned)(((img[p] * a.H + yy) * a.W + xx) * a.Cin + lchunk * 4)) * 4u;   // element index < 2^30: the byte offset is formed unsigned
        av[p][0] = in ? off : kOob;
        continue;
      }
      if (!m_ok[p]) continue;
      {
        // dcn_v1.py:L132-166 (deformable_im2col): h_im = h_in + i*dil + offset_h, zero outside (-1, H) x (-1, W),
        // corners outside the image contribute 0 (dmcn_im2col_bilinear L25-56)
        const size_t ob = (((size_t)img[p] * 18 + 2 * tap) * a.H + py[p]) * a.W + px[p];
        const float oh = a.offset[ob], ow = a.offset[ob + (size_t)a.H * a.W];
        const float h_im = (float)(py[p] + r - 1) + oh, w_im = (float)(px[p] + s - 1) + ow;
        if (h_im > -1.f && w_im > -1.f && h_im < (float)a.H && w_im < (float)a.W) {
          const int hl = (int)floorf(h_im), wl = (int)floorf(w_im);
          const float lh = h_im - hl, lw = w_im - wl, hh = 1.f - lh, hw = 1.f - lw;
          const float wt[4] = {hh * hw, hh * lw, lh * hw, lh * lw};
          const int cy[4] = {hl, hl, hl + 1, hl + 1}, cx[4] = {wl, wl + 1, wl, wl + 1};
#pragma unroll
          for (int k = 0; k < 4; k++) {
            aw[p][k] = wt[k];
            if (cy[k] >= 0 && cy[k] < a.H && cx[k] >= 0 && cx[k] < a.W)
              av[p][k] = ((unsigned)(((img[p] * a.H + cy[k]) * a.W + cx[k]) * a.Cin + lchunk * 4)) * 4u;
          }
        }
      }
    }
  };

  static_assert(DEPTH == 1 || (DEPTH == 2 && !DEFORM), "two register sets: plain taps only");
  v4f ra[DEPTH][PASSES], rb[DEPTH][PASSES];
  auto load_set = [&](auto setc, int tap, int c) {     // c: first channel of the K step
    constexpr int S = decltype(setc)::value;
    const unsigned sa = (unsigned)(c * 4), sb = (unsigned)((tap * a.Cin + c) * 4);
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      if constexpr (!DEFORM) {
        ra[S][p] = buf_load(rx, av[p][0], sa);
      } else {
        v4f v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = buf_load(rx, av[p][k], sa);
        ra[S][p] = aw[p][0] * v[0] + aw[p][1] * v[1] + aw[p][2] * v[2] + aw[p][3] * v[3];
      }
      rb[S][p] = buf_load(rw, wv[p], sb);
    }
  };
  auto store_set = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int p = 0; p < PASSES; p++) {
      *reinterpret_cast<v4f*>(s_raw + buf * 2 * TILE + st_off[p]) = ra[S][p];
      *reinterpret_cast<v4f*>(s_raw + buf * 2 * TILE + TILE + st_off[p]) = rb[S][p];
    }
  };
  using Set0 = std::integral_constant<int, 0>;
  auto load_step = [&](int tap, int c) { load_set(Set0{}, tap, c); };
  auto store_step = [&](int buf) { store_set(Set0{}, buf); };

  // ---- compute role: wave (kg, wm, wn) owns outputs [wm*BT/2, +BT/2) x [wn*BT/2, +BT/2) and the 8-deep slices
  // q = qq * KG + kg of every K step ----
  const int kg = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
  const int frow = lane & 31, fhalf = lane >> 5;
  int fa_off[QN], fb_off[QN];            // byte positions of this lane's fragments (tile 0, buffer 0)
#pragma unroll
  for (int qq = 0; qq < QN; qq++) {
    const int chunk = (qq * KG + kg) * 2 + fhalf;
    fa_off[qq] = swz_bytes<BK>(wm * (BT / 2) + frow, chunk);
    fb_off[qq] = TILE + swz_bytes<BK>(wn * (BT / 2) + frow, chunk);
  }
  v16f acc[T][T];
#pragma unroll
  for (int i = 0; i < T; i++)
#pragma unroll
    for (int j = 0; j < T; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  int tap = step0 / spt, c = (step0 - tap * spt) * BK;
  set_tap(tap);
  auto advance = [&]() {          // the load cursor: next K step (next BK channels, then the next tap)
    c += BK;
    if (c == a.Cin) {
      c = 0;
      tap++;
      set_tap(tap);
    }
  };
  auto mfma_step = [&](int buf) {
    const char* sb = s_raw + buf * 2 * TILE;
#pragma unroll
    for (int qq = 0; qq < QN; qq++) {
      v4f fa[T], fb[T];
#pragma unroll
      for (int i = 0; i < T; i++) {       // tile i: 32 rows further = the same swizzle (f repeats every 16 / 32 rows)
        fa[i] = *reinterpret_cast<const v4f*>(sb + fa_off[qq] + i * 32 * BK * 4);
        fb[i] = *reinterpret_cast<const v4f*>(sb + fb_off[qq] + i * 32 * BK * 4);
      }
#pragma unroll
      for (int kk = 0; kk < 4; kk++)
#pragma unroll
        for (int i = 0; i < T; i++)
#pragma unroll
          for (int j = 0; j < T; j++)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk], fb[j][kk], acc[i][j], 0, 0, 0);
    }
  };
  if constexpr (DEPTH == 1) {
    load_step(tap, c);
    store_step(0);
    __syncthreads();
    for (int step = 0; step < nsteps; step++) {
      const int buf = step & 1;
      const bool more = step + 1 < nsteps;
      if (more) {
        advance();
        load_step(tap, c);            // in flight during the MFMAs below
      }
      mfma_step(buf);
      if (more) store_step(buf ^ 1);   // the other buffer: its last readers passed the barrier of the previous step
      __syncthreads();
    }
  } else {
    // At the top of turn s: LDS buffer (s & 1) holds step s, register set (s + 1) & 1 holds step s + 1 (in flight), set
    // (s & 1) is free and takes step s + 2; steady-state turns request unconditionally (counted waits: conv_bn.hip).
    using Set1 = std::integral_constant<int, DEPTH - 1>;
    load_set(Set0{}, tap, c);
    if (nsteps > 1) {
      advance();
      load_set(Set1{}, tap, c);
    }
    store_set(Set0{}, 0);
    __syncthreads();
    int step = 0;
    for (; step + 3 < nsteps; step += 2) {
      advance();
      load_set(Set0{}, tap, c);
      mfma_step(0);
      store_set(Set1{}, 1);
      __syncthreads();
      advance();
      load_set(Set1{}, tap, c);
      mfma_step(1);
      store_set(Set0{}, 0);
      __syncthreads();
    }
    while (step < nsteps) {
      if (step + 2 < nsteps) {
        advance();
        load_set(Set0{}, tap, c);
      }
      mfma_step(0);
      if (step + 1 < nsteps) store_set(Set1{}, 1);
      __syncthreads();
      if (++step >= nsteps) break;
      if (step + 2 < nsteps) {
        advance();
        load_set(Set1{}, tap, c);
      }
      mfma_step(1);
      if (step + 1 < nsteps) store_set(Set0{}, 0);
      __syncthreads();
      ++step;
    }
  }

  // ---- KG = 2: the second group hands its partial tile over through LDS (the operand buffers are free now) ----
  if (KG == 2) {
    float* red = reinterpret_cast<float*>(s_raw);           // [wave & 3][T*T*16][64 lanes]
    static_assert(KG == 1 || 4 * T * T * 16 * 64 * 4 <= 4 * TILE, "reduction buffer");
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < T; i++)
#pragma unroll
        for (int j = 0; j < T; j++)
#pragma unroll
          for (int e = 0; e < 16; e++) red[(((wave & 3) * T * T + i * T + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (kg == 1) return;
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // A tile inside the map (round 6, conv_bn.hip's finding): raw buffer loads / stores with one 32-bit lane offset and the 16
  // row offsets in SGPRs instead of 16 predicated accesses with 64-bit addresses per 32 x 32 tile; same values, same order.
  // (the 64 x 64 tile only: on the 128 x 128 tile the extra live values cost the 127-VGPR kernel two spills, and its
  //  epilogue is 2 % of a 275 us launch)
  if (T == 1 && m0 + BT <= M) {
    const unsigned ybytes = (unsigned)(M * a.Cout * 4);
    float* const dst = a.partial ? a.partial + (size_t)blockIdx.y * M * a.Cout : a.y;
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rm =
        __builtin_amdgcn_make_buffer_rsrc((void*)(a.rowmask ? a.rowmask : a.x), 0, (unsigned)(M * 4), 0x00020000);
#pragma unroll
    for (int i = 0; i < T; i++) {
      const unsigned mrow = (unsigned)(m0 + wm * (BT / 2) + i * 32 + 4 * (lane >> 5));
      float mk[16];
#pragma unroll
      for (int e = 0; e < 16; e++)
        mk[e] = (a.rowmask && !a.partial)
                    ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rm, mrow * 4u, (unsigned)(((e & 3) + 8 * (e >> 2)) * 4), 0))
                    : 1.f;
#pragma unroll
      for (int j = 0; j < T; j++) {
        const int n = n0 + wn * (BT / 2) + j * 32 + (lane & 31);
        if (n < a.Cout) {
          const float b = a.bias ? a.bias[n] : 0.f;
          const unsigned base = (mrow * (unsigned)a.Cout + (unsigned)n) * 4u;
#pragma unroll
          for (int e = 0; e < 16; e++) {
            float v = acc[i][j][e];
            if (KG == 2)
              v += reinterpret_cast<const float*>(s_raw)[(((wave & 3) * T * T + i * T + j) * 16 + e) * 64 + lane];
            if (!a.partial) {
              v += b;
              if (a.relu) v = fmaxf(v, 0.f);
              if (a.rowmask) v *= mk[e];
            }
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, base,
                                                  (unsigned)(((e & 3) + 8 * (e >> 2)) * a.Cout * 4), 0);
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < T; i++) {
    float mk[16];
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const long m = m0 + wm * (BT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
      mk[e] = (a.rowmask && m < M) ? a.rowmask[m] : 1.f;       // all 16 loads in flight together
    }
#pragma unroll
    for (int j = 0; j < T; j++) {
      const int n = n0 + wn * (BT / 2) + j * 32 + (lane & 31);
      const float b = (a.bias && n < a.Cout) ? a.bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const long m = m0 + wm * (BT / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (m < M && n < a.Cout) {
          float v = acc[i][j][e];
          if (KG == 2)      // the other wave group's partial sum (read tile by tile: no second accumulator set live)
            v += reinterpret_cast<const float*>(s_raw)[(((wave & 3) * T * T + i * T + j) * 16 + e) * 64 + lane];
          if (a.partial) {
            a.partial[((size_t)blockIdx.y * M + m) * a.Cout + n] = v;
          } else {
            v += b;
            if (a.relu) v = fmaxf(v, 0.f);
            if (a.rowmask) v *= mk[e];
            a.y[(size_t)m * a.Cout + n] = v;
          }
        }
      }
    }
  }
}

// second stage of the cross-workgroup K split: y = [mask *] [relu] (sum of the partial tiles + bias); one float4 of
// channels per thread, the partial planes read in a fixed order (deterministic)
__global__ __launch_bounds__(256) void conv_ksplit_finish_kernel(const float* __restrict__ partial, int ksplit, long M,
                                                                int Cout, const float* __restrict__ bias, int relu,
                                                                const float* __restrict__ rowmask,
                                                                float* __restrict__ y) {
  const long n4 = M * Cout / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    v4f v = reinterpret_cast<const v4f*>(partial)[i];
    for (int k = 1; k < ksplit; k++) v += reinterpret_cast<const v4f*>(partial)[(size_t)k * n4 + i];
    const long m = i * 4 / Cout;
    const int n = (int)(i * 4 - m * Cout);
    if (bias) v += *reinterpret_cast<const v4f*>(bias + n);
    if (relu)
#pragma unroll
      for (int k = 0; k < 4; k++) v[k] = fmaxf(v[k], 0.f);
    if (rowmask) v *= rowmask[m];
    reinterpret_cast<v4f*>(y)[i] = v;
  }
}

template <int BT, int BK, int KG>
int launch(const ConvArgs& a, hipStream_t st) {
  const long M = (long)a.N * a.H * a.W;
  const long tiles = ((M + BT - 1) / BT) * ((a.Cout + BT - 1) / BT);
  if (a.offset) {
    hipLaunchKernelGGL((conv3x3_igemm_kernel<BT, BK, KG, true>), dim3((unsigned)tiles, a.ksplit), dim3(256 * KG), 0, st, a);
    return jdet_launch_status();
  }
  if constexpr (BT == 64 && KG == 1) {
    static const char* e = getenv("JDET_CONV_IGEMM_DEEP");      // operand tiles two K steps ahead (A/B switch; default on)
    if (!e || atoi(e) != 0) {
      hipLaunchKernelGGL((conv3x3_igemm_kernel<BT, BK, KG, false, 2>), dim3((unsigned)tiles, a.ksplit), dim3(256), 0, st, a);
      return jdet_launch_status();
    }
  }
  hipLaunchKernelGGL((conv3x3_igemm_kernel<BT, BK, KG, false>), dim3((unsigned)tiles, a.ksplit), dim3(256 * KG), 0, st, a);
  return jdet_launch_status();
}

}  // namespace

// Supported: Cin % 16 == 0, 16-byte aligned tensors, N*H*W*max(Cin, Cout) < 2^30; any Cout, any N, H, W.
JDET_API int jdet_conv3x3_igemm_supported(int Cin, int Cout) { return Cin > 0 && Cin % 16 == 0 && Cout > 0; }

// Cross-workgroup K split for small maps (a tile's 2304-deep reduction is otherwise the floor of the launch): how many
// ways a problem is split when a workspace is offered, and the bytes that takes.  1 = no split.
static int ksplit_for(long M, int Cin, int Cout, bool deform) {
  const long tiles64 = ((M + 63) / 64) * ((Cout + 63) / 64);
  if (deform || Cout % 4 != 0 || tiles64 >= 384) return 1;
  const int steps = 9 * (Cin / (Cin % 32 == 0 ? 32 : 16));
  int k = (int)(768 / tiles64);                 // aim at ~3 workgroups per CU
  if (k > 8) k = 8;
  if (k > steps / 4) k = steps / 4;             // at least 4 K steps per part
  return k < 2 ? 1 : k;
}

JDET_API size_t jdet_conv3x3_igemm_workspace(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || !jdet_conv3x3_igemm_supported(Cin, Cout)) return 0;
  const long M = (long)N * H * W;
  const int k = ksplit_for(M, Cin, Cout, false);
  return k > 1 ? sizeof(float) * (size_t)k * M * Cout : 0;
}

JDET_API int jdet_conv3x3_igemm_forward(const float* x_nhwc, int N, int H, int W, int Cin, const float* w_krsc, int Cout,
                                        const float* bias, int relu, const float* rowmask, const float* offset,
                                        int tile, float* y_nhwc, void* workspace, size_t workspace_bytes,
                                        jdet_stream_t stream) {
  if (N < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return JDET_E_BADARG;
  if (!jdet_conv3x3_igemm_supported(Cin, Cout)) return JDET_E_UNSUPPORTED;
  if (N == 0) return JDET_OK;
  if (!x_nhwc || !w_krsc || !y_nhwc) return JDET_E_BADARG;
  if ((((uintptr_t)x_nhwc) | ((uintptr_t)w_krsc)) & 15) return JDET_E_BADARG;
  const long M = (long)N * H * W;
  if (M * (Cin > Cout ? Cin : Cout) >= (1L << 30)) return JDET_E_UNSUPPORTED;     // 32-bit byte offsets
  ConvArgs a{x_nhwc, w_krsc, bias, rowmask, offset, y_nhwc, nullptr, N, H, W, Cin, Cout, relu ? 1 : 0, 1};
  // tile: 0 = automatic; 64 / 128 = edge of the output tile; + 1 selects the 16-deep K step, + 2 a single wave group
  // (no intra-workgroup K split) -- measurement aids.  A forced tile never uses the cross-workgroup split.
  const int edge = tile & ~3;
  if (tile != 0 && edge != 64 && edge != 128) return JDET_E_BADARG;
  const long tiles128 = ((M + 127) / 128) * ((Cout + 127) / 128);
  const bool big = tile ? edge == 128 : tiles128 >= 512;
  const bool k32 = Cin % 32 == 0 && !(tile & 1);
  // intra-workgroup K split (8 waves): always for the 128 tile (4 waves per SIMD at 2 workgroups per CU); for the 64
  // tile only while the grid is small (the per-tile latency chain is the floor there: 56 vs 59 us), not once 64-tiles
  // alone fill the chip (87 vs 94 us at 2 x 64^2 positions).  Never for the gather (its registers do not fit).
  const long tiles64 = ((M + 63) / 64) * ((Cout + 63) / 64);
  const bool split = k32 && !(tile & 2) && !offset && (big || tile || tiles64 < 512);
  hipStream_t st = (hipStream_t)stream;
  if (tile == 0 && !big) {
    const int ks = ksplit_for(M, Cin, Cout, offset != nullptr);
    if (ks > 1 && workspace && workspace_bytes >= sizeof(float) * (size_t)ks * M * Cout) {
      // the finish kernel moves float4s: y, bias and the scratch must be 16-byte aligned as well (views at odd offsets)
      if ((((uintptr_t)y_nhwc) | ((uintptr_t)bias) | ((uintptr_t)workspace)) & 15 || Cout % 4 != 0) goto unsplit;
      a.partial = (float*)workspace;
      a.ksplit = ks;
      int e = k32 ? launch<64, 32, 1>(a, st) : launch<64, 16, 1>(a, st);
      if (e) return e;
      const long n4 = M * Cout / 4;
      long g = (n4 + 255) / 256;
      if (g > 4096) g = 4096;
      hipLaunchKernelGGL(conv_ksplit_finish_kernel, dim3((unsigned)g), dim3(256), 0, st, a.partial, ks, M, Cout, bias,
                         relu ? 1 : 0, rowmask, y_nhwc);
      return jdet_launch_status();
    }
  }
unsplit:
  if (big) return k32 ? (split ? launch<128, 32, 2>(a, st) : launch<128, 32, 1>(a, st)) : launch<128, 16, 1>(a, st);
  return k32 ? (split ? launch<64, 32, 2>(a, st) : launch<64, 32, 1>(a, st)) : launch<64, 16, 1>(a, st);
}
