// md_conv3_head: the output head of the U-Net for inference -- GroupNorm + SiLU + 3x3x3 conv to 4 channels
// (lib/diffusion/models/ddpm_res64.py:120-121, applied :186-189: nn.GroupNorm(32, nf), act, conv3x3(nf, channels)) -- as the
// dx-folded conv of DDPMUNet3D._head_forward: rows = (co, kw) pairs (12 of the 32 rows of an MFMA tile), 3 x 3 x 1 taps,
// md_fold_dx adds the three x-shifted columns and the bias afterwards.
//
// Replaces md_gn_apply (fp32 -> normalised, activated, split S16B: 1.07 GB read + 1.07 GB written at 64^3, B = 8) followed by
// the generic tile MD_CFG_C3X_32 (~250 address / branch instructions per tap and wave next to 6 MFMAs): 0.44 + 0.67 ms per
// step.  Here the fp32 tensor is read once; a workgroup = 32 rows x (4 x 8 x 8) positions, 8 waves, ONE accumulator tile per
// wave (a plane half: 4 rows of 8 positions), K in 16-channel chunks:
//   * halo of a chunk: 6 x 10 x 8 positions (no x halo: the taps along x are folded into the rows) x 16 channels, read as
//     fp32 (32 contiguous bytes per position and 8-channel group), y = x a + c (folded GroupNorm affine of md_gn_finalize),
//     SiLU with v_exp_f32 / v_rcp_f32, zero outside the grid, bf16 hi / lo split -- in registers, then LDS;
//   * weights of a chunk: 9 tap tiles x 2 KB (WPK, nt = 32, kc = 16) through LDS;
//   * per tap a wave issues 3 dependent MFMAs (bf16x3, one accumulator) on 2 + 2 fragment reads: a wave is latency-bound by
//     itself, and the transform is as much VALU time (16 quarter-rate exp / rcp per item) as the chunk's MFMAs.  So the halo
//     and the weights are SINGLE-buffered (48 KB of LDS, < 128 registers): TWO workgroups per CU, four waves per SIMD -- one
//     workgroup's transform phase runs under the other's MFMA phase.  Per chunk: request chunk c + 1 | 9 taps | barrier |
//     transform + commit | barrier.  (First form: double-buffered, one workgroup per CU, 0.77-0.81 ms at 64^3, B = 8; staggering
//     the transform between the two waves of a SIMD changed nothing -- each wave's serial chain is the bound.)
#include "md_common.h"

namespace {
constexpr int HD_KC = 16, HD_TZ = 4, HD_TY = 8, HD_TX = 8;
constexpr int HD_HS = 6 * 10 * 8;                        // 480 halo slots per (channel group, plane)
constexpr int HD_HBUF = 4 * HD_HS * 16;                  // 30720 B: [kg 2][plane 2][480] items
constexpr int HD_WTAP = 2 * 2 * 32;                      // 128 items per tap tile: [kg 2][plane 2][row 32]
constexpr int HD_WSTAGE = 9 * HD_WTAP * 16;              // 18432 B per chunk
constexpr int HD_LDS = HD_WSTAGE + HD_HBUF;              // 49152 B: two workgroups per CU
constexpr int HD_THREADS = 512;
}  // namespace

struct HdArgs {
  const float* x;          // F32B [B][cin/8][P][8]
  const float* ac;         // [B][cin][2] folded GroupNorm affine (a, c)
  const uint4* wpk;        // md_pack_weights(rows, kdim = cin, taps = 9, nt = 32, kc = 16)
  float* y;                // F32B [B][rows_alloc/8][P][8]
  int batch, cin, rows_alloc, D, H, W;
};

// (512 threads, 6 waves per SIMD: 80 registers and 48 KB of LDS put three workgroups on a CU)
__global__ __launch_bounds__(HD_THREADS, 6) void md_conv3_head_kernel(const HdArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[HD_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wid & 3, yh = wid >> 2;
  const int j = lane & 31, h = lane >> 5;
  const int D = A.D, H = A.H, W = A.W;
  const int64_t P = (int64_t)D * H * W;
  const int ntx = W / HD_TX, nty = H / HD_TY;
  const int tiles = ntx * nty * (D / HD_TZ);
  int bid = blockIdx.x;      // XCD-aware order: one contiguous run of tiles per XCD
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles, t = bid % tiles;
  const int x0 = (t % ntx) * HD_TX, y0 = ((t / ntx) % nty) * HD_TY, z0 = (t / (ntx * nty)) * HD_TZ;
  const int ncc = A.cin / HD_KC;

  // ---- halo items of this thread: channel group kg = tid >> 8, halo positions (tid & 255) + 256 i (i < 2) -----------------
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 8);
  int hsrc[2], hdst[2];
  unsigned live = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (tid & 255) + i * 256;
    const bool in_halo = p < HD_HS;
    const int hx = p & 7, hy = (p >> 3) % 10, hz = in_halo ? p / 80 : 0;
    const int iz = z0 + hz - 1, iy = y0 + hy - 1, ix = x0 + hx;
    const bool ok = in_halo && iz >= 0 && iz < D && iy >= 0 && iy < H;
    hsrc[i] = ok ? (int)((((int64_t)iz * H + iy) * W + ix) * 2) : 0;
    hdst[i] = ((kg * 2) * HD_HS + (in_halo ? p : 0)) * 16;          // items beyond the halo (tid & 255 >= 224, i = 1) rewrite slot 0 with
    if (ok) live |= 1u << i;                                         // ... see act_commit: they are skipped there
  }
  const bool has2 = (tid & 255) + 256 < HD_HS;
  const uint4* xb = (const uint4*)A.x + (int64_t)b * (A.cin >> 3) * P * 2;
  const float* acb = A.ac + (int64_t)b * A.cin * 2;
  const uint4* wbase = A.wpk + tid;

  struct Stage { uint4 h[4]; uint4 w[3]; f32x4 ac[4]; };
  Stage S;
  auto issue = [&](int cc) {                              // fp32 halo items, weights and affine of chunk cc -> registers
    const int g8 = cc * 2 + kg;
    const uint4* cb = xb + (int64_t)g8 * P * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i) { S.h[2 * i] = cb[hsrc[i]]; S.h[2 * i + 1] = cb[hsrc[i] + 1]; }
    const uint4* wp = wbase + (int64_t)cc * (9 * HD_WTAP);
    S.w[0] = wp[0]; S.w[1] = wp[HD_THREADS];
    S.w[2] = wp[tid < 9 * HD_WTAP - 2 * HD_THREADS ? 2 * HD_THREADS : 0];
    const f32x4* ap = (const f32x4*)(acb + g8 * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) S.ac[q] = ap[q];
  };
  auto commit = [&]() {                                   // affine + SiLU + zero pad + split -> halo buffer; weights -> stage
    unsigned char* hb = lds + HD_WSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint4 r0 = S.h[2 * i], r1 = S.h[2 * i + 1];
      const float v[8] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w),
                          __uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w)};
      const bool lv = (live >> i) & 1u;
      float yv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = v[e] * S.ac[e >> 1][(e & 1) * 2] + S.ac[e >> 1][(e & 1) * 2 + 1];
        y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
        yv[e] = lv ? y : 0.f;
      }
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) md_split2(yv[2 * e], yv[2 * e + 1], hi[e], lo[e]);
      if (i == 0 || has2) {
        *(uint4*)(hb + hdst[i]) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *(uint4*)(hb + hdst[i] + HD_HS * 16) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
    unsigned char* wb = lds + tid * 16;
    *(uint4*)wb = S.w[0];
    *(uint4*)(wb + HD_THREADS * 16) = S.w[1];
    if (tid < 9 * HD_WTAP - 2 * HD_THREADS) *(uint4*)(wb + 2 * HD_THREADS * 16) = S.w[2];
  };

  // ---- fragment addresses --------------------------------------------------------------------------------------------------
  const unsigned char* pA = lds + (h * 2 * 32 + j) * 16;                                   // + tap * 2048, + plane * 512
  const unsigned char* pB = lds + HD_WSTAGE + (h * 2 * HD_HS + (wc * 10 + yh * 4 + (j >> 3)) * 8 + (j & 7)) * 16;   // + (kd * 80 + kh * 8) * 16, + plane * HS * 16

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  struct Frags { bf16x8 ahi, alo, bhi, blo; };
  Frags F[2];
#define HD_LOAD(Fx, TAP)                                                                          \
  Fx.ahi = *(const bf16x8*)(pA + (TAP) * (HD_WTAP * 16));                                          \
  Fx.alo = *(const bf16x8*)(pA + (TAP) * (HD_WTAP * 16) + 32 * 16);                                \
  Fx.bhi = *(const bf16x8*)(pB + ((((TAP) / 3) * 80 + ((TAP) % 3) * 8)) * 16);                     \
  Fx.blo = *(const bf16x8*)(pB + ((((TAP) / 3) * 80 + ((TAP) % 3) * 8) + HD_HS) * 16);
#define HD_MMA(Fx)                                                               \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fx.alo, Fx.bhi, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fx.ahi, Fx.blo, acc, 0, 0, 0);  \
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Fx.ahi, Fx.bhi, acc, 0, 0, 0);

  issue(0);
  commit();
  __syncthreads();
  for (int c = 0; c < ncc; ++c) {
    if (c + 1 < ncc) issue(c + 1);             // in flight during the 9 taps (and the other workgroup's phases)
    HD_LOAD(F[0], 0)
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      if (tap < 8) { HD_LOAD(F[(tap + 1) & 1], tap + 1) }
      HD_MMA(F[tap & 1])
    }
    if (c + 1 < ncc) {
      __syncthreads();                         // everyone is done reading this chunk's halo and weights
      commit();
      __syncthreads();
    }
  }
#undef HD_LOAD
#undef HD_MMA

  // ---- epilogue: rows 0 .. rows_alloc - 1 (<= 16 for the 12 (co, kw) rows of a 4-channel head) -------------------------------
  float* yp = A.y + (int64_t)b * (A.rows_alloc / 8) * P * 8;
  const int64_t gp = ((int64_t)(z0 + wc) * H + (y0 + yh * 4 + (j >> 3))) * W + x0 + (j & 7);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 8 * q + 4 * h;
    if (row < A.rows_alloc) {
      const f32x4 o = {acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]};
      *(f32x4*)(yp + ((int64_t)(row >> 3) * P + gp) * 8 + (row & 7)) = o;
    }
  }
}

extern "C" int md_conv3_head(const float* x, const float* ac, const void* wpk, float* y, int32_t batch, int32_t cin,
                             int32_t rows_alloc, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!x || !ac || !wpk || !y || batch <= 0) return MD_ERR_BAD_ARG;
  if (cin <= 0 || (cin % (2 * HD_KC)) || rows_alloc <= 0 || rows_alloc > 32 || (rows_alloc & 7)) return MD_ERR_UNSUPPORTED;
  if (D <= 0 || H <= 0 || W <= 0 || (D % HD_TZ) || (H % HD_TY) || (W % HD_TX)) return MD_ERR_UNSUPPORTED;
  if ((int64_t)D * H * W * 2 >= (int64_t)1 << 31) return MD_ERR_UNSUPPORTED;
  HdArgs a;
  a.x = x; a.ac = ac; a.wpk = (const uint4*)wpk; a.y = y;
  a.batch = batch; a.cin = cin; a.rows_alloc = rows_alloc; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / HD_TZ) * (H / HD_TY) * (W / HD_TX);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_conv3_head_kernel, dim3((unsigned)(tiles * batch)), dim3(HD_THREADS), 0, (hipStream_t)stream, a);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
