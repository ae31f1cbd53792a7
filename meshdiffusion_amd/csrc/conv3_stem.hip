// md_conv3_stem: the input convolution of the U-Net for inference -- conv3x3(4, nf) on the noisy grid plus the input-independent
// pos_layer(coords) + mask_layer(mask) terms (lib/diffusion/models/ddpm_res64.py:87-92, applied :138-146) -- in the dx-folded
// form of DDPMUNet3D._stem_forward: the three x-shifted copies of the 4 input channels are the K = 16 (12 used) channels of a
// 3 x 3 x 1-tap conv (operand: md_ncdhw_to_s16b_xfold), the constant terms come in as a residual shared by the batch.
//
// Replaces the generic tile MD_CFG_C3X_128_K16 (0.73 ms at 64^3, B = 8, for 58 GFLOP and 1.07 GB of output: ~2600 cycles of
// address / barrier work per tap and a residual load waited for in front of every 16-byte store) and the md_gn_stats pass
// over its output (0.19 ms): the GroupNorm sums of the first ResnetBlock come from this kernel's epilogue.
// Workgroup = 128 output channels x (4 x 8 x 8) positions, 8 waves x (2 x 2) accumulator tiles (the geometry of
// md_conv3_main); ONE K chunk: the halo (6 x 10 x 8 positions, no x halo, 30 KB) is loaded once; the weights go through LDS
// one (kd) row of three tap tiles at a time (2 stages x 24 KB), one barrier per row; 78 KB of LDS and 128 registers per wave: two
// workgroups per CU (the kernel is bound by its 256 KB of epilogue traffic per workgroup: the other workgroup multiplies meanwhile).
// Same arithmetic as the generic tile: bf16x3 products, taps in the same order.
#include "md_common.h"

namespace {
constexpr int ST_NT = 128, ST_TZ = 4, ST_TY = 8, ST_TX = 8;
constexpr int ST_HS = 6 * 10 * 8;                          // 480 halo slots per (channel group, plane)
constexpr int ST_W_ITEMS = 2 * 2 * ST_NT;                  // 512 uint4 per tap tile (8 KB): [kg 2][plane 2][row 128]
constexpr int ST_WROW = 3 * ST_W_ITEMS;                    // the three kh tiles of a kd row
constexpr int ST_W_LDS_BYTES = 2 * ST_WROW * 16;           // 49152: two stages
constexpr int ST_LDS_BYTES = ST_W_LDS_BYTES + 4 * ST_HS * 16;   // + 30720 = 79872
constexpr int ST_THREADS = 512;
}  // namespace

struct StArgs {
  const uint4* x;          // S16B [B][2][2][P] items (16 channels: the x-folded operand)
  const uint4* wpk;        // md_pack_weights(rows = cout, kdim = 16, taps = 9, nt = 128, kc = 16)
  float* out;              // F32B [B][cout/8][P][8]
  const float* bias;       // [cout] or null
  const float* residual;   // F32B [cout/8][P][8] shared by the batch, or null
  double* stats;           // [B][cout][2] += (sum, sum of squares) of the output, or null
  int batch, cout, D, H, W;
};

// (4 waves per SIMD: two workgroups per CU)
__global__ __launch_bounds__(ST_THREADS, 4) void md_conv3_stem_kernel(const StArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[ST_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int j = lane & 31, h = lane >> 5;
  const int D = A.D, H = A.H, W = A.W;
  const int64_t P = (int64_t)D * H * W;
  const int ntx = W / ST_TX, nty = H / ST_TY;
  const int tiles = ntx * nty * (D / ST_TZ);
  int bid = blockIdx.x;      // XCD-aware order: one contiguous run of tiles per XCD
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles, t = bid % tiles;
  const int x0 = (t % ntx) * ST_TX, y0 = ((t / ntx) % nty) * ST_TY, z0 = (t / (ntx * nty)) * ST_TZ;
  const int rt = blockIdx.y;

  // ---- halo: (plane-major item gp = tid >> 7 in [kg][plane] order, position (tid & 127) + 128 i) -> LDS, zero outside ------
  {
    const int gp = tid >> 7;                                  // (kg * 2 + plane), wave-uniform
    const uint4* src = A.x + ((int64_t)b * 4 + gp) * P;
    uint4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = (tid & 127) + i * 128;
      const int hx = p & 7, hy = (p >> 3) % 10, hz = p / 80;
      const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + hx;
      const bool ok = p < ST_HS && iz >= 0 && iz < D && iy >= 0 && iy < H;
      v[i] = make_uint4(0, 0, 0, 0);
      if (ok) v[i] = src[((int64_t)iz * H + iy) * W + ix];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = (tid & 127) + i * 128;
      if (p < ST_HS) *(uint4*)(lds + ST_W_LDS_BYTES + (gp * ST_HS + p) * 16) = v[i];
    }
  }
  // ---- weights: row R = kd = three consecutive tap tiles (kh = 0..2) ---------------------------------------------------------
  const uint4* wbase = A.wpk + (int64_t)rt * 9 * ST_W_ITEMS + tid;
  uint4 wreg0, wreg1, wreg2;
  auto w_issue = [&](int R) {
    const uint4* wp = wbase + (int64_t)R * ST_WROW;
    wreg0 = wp[0]; wreg1 = wp[ST_THREADS]; wreg2 = wp[2 * ST_THREADS];
  };
  auto w_commit = [&](int stage) {
    unsigned char* dst = lds + stage * (ST_WROW * 16) + tid * 16;
    *(uint4*)dst = wreg0;
    *(uint4*)(dst + ST_THREADS * 16) = wreg1;
    *(uint4*)(dst + 2 * ST_THREADS * 16) = wreg2;
  };
  w_issue(0);
  w_commit(0);
  w_issue(1);

  // ---- fragment addresses ------------------------------------------------------------------------------------------------------
  const unsigned char* pA = lds + (h * 2 * ST_NT + wr * 64 + j) * 16;          // + stage * 24576 + kh * 8192, + plane * 2048, + rm * 512
  // halo [kg 2][plane 2][HS]: slot = ((wc + kd) * 10 + cm * 4 + (j >> 3) + kh) * 8 + (j & 7)
  const unsigned char* pB = lds + ST_W_LDS_BYTES + (h * 2 * ST_HS + (wc * 10 + (j >> 3)) * 8 + (j & 7)) * 16;

  f32x16 acc[2][2];
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int cm = 0; cm < 2; ++cm) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rm][cm][r] = 0.f;
      asm volatile("" : "+a"(acc[rm][cm]));
    }
  struct Frags { bf16x8 ahi[2], alo[2], bhi[2], blo[2]; };
  Frags F;                   // one set: 128 registers per wave is the budget of two workgroups per CU (4 waves per SIMD cover the reads)
#define ST_LOAD_FRAGS(Fx, STAGE, KD, KH)                                                                             \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) {                                                                 \
    Fx.ahi[rm] = *(const bf16x8*)(pA + (STAGE) * (ST_WROW * 16) + (KH) * (ST_W_ITEMS * 16) + (rm * 32) * 16);        \
    Fx.alo[rm] = *(const bf16x8*)(pA + (STAGE) * (ST_WROW * 16) + (KH) * (ST_W_ITEMS * 16) + (ST_NT + rm * 32) * 16); \
  }                                                                                                                  \
  _Pragma("unroll") for (int cm = 0; cm < 2; ++cm) {                                                                 \
    Fx.bhi[cm] = *(const bf16x8*)(pB + (((KD) * 10 + (KH) + cm * 4) * 8) * 16);                                       \
    Fx.blo[cm] = *(const bf16x8*)(pB + (((KD) * 10 + (KH) + cm * 4) * 8 + ST_HS) * 16);                               \
  }
#define ST_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#define ST_MMA(Fx)                                                                                          \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    ST_MFMA(Fx.alo[rm], Fx.bhi[cm], acc[rm][cm]);                                                           \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    ST_MFMA(Fx.ahi[rm], Fx.blo[cm], acc[rm][cm]);                                                           \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    ST_MFMA(Fx.ahi[rm], Fx.bhi[cm], acc[rm][cm]);

  __syncthreads();                                          // halo and weight row 0 in LDS
  // ---- 9 taps: row kd reads weight stage kd & 1; W(kd + 1) is committed at its first tap, W(kd + 2) requested; one barrier per
  // row, in front of the next row's first fragment read
#pragma unroll
  for (int u = 0; u < 9; ++u) {
    const int kd = u / 3, kh = u % 3, stage = kd & 1;
    if (kh == 0 && kd > 0) __syncthreads();                 // W(kd) visible; everyone is done with the stage W(kd + 1) goes to
    if (kh == 0 && kd < 2) {
      w_commit(stage ^ 1);
      if (kd == 0) w_issue(2);
    }
    ST_LOAD_FRAGS(F, stage, kd, kh)
    ST_MMA(F)
  }
#undef ST_LOAD_FRAGS
#undef ST_MMA
#undef ST_MFMA

  // ---- epilogue: bias + shared residual, 16-byte stores into F32B, GroupNorm sums.  One (row tile, 4-row group) at a time so
  // that few values are live: its two column tiles' residuals are requested, combined, stored; the sums of its 4 channels over
  // the lane's 2 positions are reduced over the 16-lane DPP rows at once and parked in LDS (the weight stages are free).
  const int rows = A.cout;
  float* outp = A.out + (int64_t)b * (rows / 8) * P * 8;
  const bool want_stats = A.stats != nullptr;
  auto row_sum = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
  };
  float* red = (float*)lds;   // [8 waves][2 rows of a half-wave][64 channels][2]
  __syncthreads();            // every wave has read its last weight fragments
  int64_t gp[2];
#pragma unroll
  for (int cm = 0; cm < 2; ++cm) gp[cm] = ((int64_t)(z0 + wc) * H + (y0 + cm * 4 + (j >> 3))) * W + (x0 + (j & 7));
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rt * ST_NT + wr * 64 + rm * 32 + 8 * q + 4 * h;
      if (row >= rows) continue;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f}, rv[2];
      if (A.bias != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (row + e < rows) bv[e] = A.bias[row + e];      // (a parameter view need not be 16-byte aligned)
      }
#pragma unroll
      for (int cm = 0; cm < 2; ++cm) {
        rv[cm] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (A.residual != nullptr) rv[cm] = *(const f32x4*)(A.residual + ((int64_t)(row >> 3) * P + gp[cm]) * 8 + (row & 7));
      }
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int cm = 0; cm < 2; ++cm) {
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = acc[rm][cm][q * 4 + e];                 // (acc + bias) + residual: the generic kernel's association
          v += bv[e];
          v += rv[cm][e];
          o4[e] = v;
          s1[e] += v;
          s2[e] += v * v;
        }
        *(f32x4*)(outp + ((int64_t)(row >> 3) * P + gp[cm]) * 8 + (row & 7)) = o4;
      }
      if (want_stats) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a1 = row_sum(s1[e]), a2 = row_sum(s2[e]);
          if ((lane & 15) == 0) {
            const int jr = (lane >> 4) & 1, ch = rm * 32 + 8 * q + 4 * h + e;
            red[((wid * 2 + jr) * 64 + ch) * 2] = a1;
            red[((wid * 2 + jr) * 64 + ch) * 2 + 1] = a2;
          }
        }
      }
    }
  if (want_stats) {      // workgroup-uniform
    __syncthreads();
    if (tid < 256) {
      const int ch = tid >> 1, which = tid & 1;          // channel within the 128-row tile
      const int w0 = (ch >> 6) * 4, c64 = ch & 63;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) sum += red[(((w0 + (k >> 1)) * 2 + (k & 1)) * 64 + c64) * 2 + which];
      const int row = rt * ST_NT + ch;
      if (row < rows) atomicAdd(A.stats + ((int64_t)b * rows + row) * 2 + which, (double)sum);
    }
  }
}

extern "C" int md_conv3_stem(const void* x16, const void* wpk, float* out, const float* bias, const float* residual, double* stats,
                             int32_t batch, int32_t cout, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!x16 || !wpk || !out || batch <= 0) return MD_ERR_BAD_ARG;
  if (cout <= 0 || (cout & 7)) return MD_ERR_UNSUPPORTED;
  if (D <= 0 || H <= 0 || W <= 0 || (D % ST_TZ) || (H % ST_TY) || (W % ST_TX)) return MD_ERR_UNSUPPORTED;
  StArgs a;
  a.x = (const uint4*)x16; a.wpk = (const uint4*)wpk; a.out = out; a.bias = bias; a.residual = residual; a.stats = stats;
  a.batch = batch; a.cout = cout; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / ST_TZ) * (H / ST_TY) * (W / ST_TX);
  const dim3 grid((unsigned)(tiles * batch), (unsigned)((cout + ST_NT - 1) / ST_NT));
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_conv3_stem_kernel, grid, dim3(ST_THREADS), 0, (hipStream_t)stream, a);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
