// md_conv3_s2: the stride-2 3x3x3 convolution of Downsample (inference), reading the raw fp32 tensor.
//
// Reference op: Downsample.forward, lib/diffusion/models/layers.py:626-643: F.pad(x, (0, 1, 0, 1, 0, 1)) followed by
// nn.Conv3d(C, C, 3, stride=2, padding=0): out[o] = sum_t w[t] * x[2 o + t], t in {0, 1, 2} per axis, zero beyond the far face.
//
// Replaces the generic tile MD_CFG_C3_S2 of gemm_conv.hip (4 x 4 x 4 outputs, one 32 x 32 accumulator per wave: 6 MFMAs next
// to ~250 address / branch instructions per tap and wave, stride-2 fragment reads with 2-way bank conflicts: 122 TF/s on
// 128 -> 128 @ 64^3 -> 32^3) AND the split pass that fed it (md_gn_apply norm = 0: fp32 -> S16B, 8 B per element).
// Same arithmetic: bf16x3 products (a_lo*b_hi + a_hi*b_lo + a_hi*b_hi, fp32 accumulation in v_mfma_f32_32x32x16_bf16).
//
// Workgroup = 128 output channels x (4 x 8 x 8) output positions, 8 waves (2 row halves x 4 output planes), each wave
// 2 x 2 accumulator tiles -- the geometry of md_conv3_main_kernel.  A 4 x 8 x 8 output tile needs 9 x 17 x 17 input
// positions: too many for LDS with any K chunk, so the K loop runs over (16-channel chunk, kd) SLABS: the 4 input planes
// 2 (z0 + zl) + kd, 17 x 17 positions each, 16 channels, hi / lo planes = 74 KB, reused by the 9 (kh, kw) taps.
//   * the slab is read as fp32 F32B (32 contiguous bytes per position and 8-channel group), split to bf16 hi / lo in
//     registers while the previous slab is being multiplied, and committed at the slab switch (single buffer, 2 barriers);
//   * LDS image of a slab row (plane zl, input row hy): the 9 even-x positions, then the 8 odd-x positions, padded to 20
//     slots of 16 B: a fragment read (8 outputs along x, stride 2 in the input) touches 8 CONSECUTIVE slots, the next
//     output row is 40 slots = 8 mod 16 further => the 16 lanes of a ds_read_b128 group hit 16 different bank groups;
//   * weights: WPK tiles of md_pack_weights (nt = 128, kc = 16, 27 taps), the three kw tiles of a (kd, kh) row (24 KB) per
//     stage, double buffered: one barrier per 36 MFMAs of a wave; every LDS address is one VGPR + immediate (two slabs =
//     18 taps unrolled, so stage and fragment-set parities are compile-time).
#include "md_common.h"

namespace {
constexpr int S2_NT = 128, S2_KC = 16, S2_TZ = 4, S2_TY = 8, S2_TX = 8;
constexpr int S2_YH = 17, S2_XH = 17, S2_ROW = 20;
constexpr int S2_HS = S2_TZ * S2_YH * S2_ROW;              // 1360 slots per (channel group, plane)
constexpr int S2_HPOS = S2_TZ * S2_YH * S2_XH;             // 1156 input positions per slab
constexpr int S2_W_ITEMS = 2 * 2 * S2_NT;                  // 512 uint4 per tap tile (8 KB): [kg 2][plane 2][row 128]
constexpr int S2_WROW = 3 * S2_W_ITEMS;                    // the three kw tiles of a (kd, kh) row
constexpr int S2_W_LDS_BYTES = 2 * S2_WROW * 16;           // 49152: two stages
constexpr int S2_LDS_BYTES = S2_W_LDS_BYTES + 4 * S2_HS * 16;   // + 87040 = 136192
constexpr int S2_THREADS = 512;
constexpr int S2_IT = (S2_HPOS + 255) / 256;               // 5 (position, channel group) items per thread and slab
}  // namespace

struct S2Args {
  const float* x;          // F32B [B][cin/8][(2D)(2H)(2W)][8]
  const uint4* wpk;        // md_pack_weights(rows = cout, kdim = cin, taps = 27, nt = 128, kc = 16)
  float* out;              // F32B [B][rows_alloc/8][D H W][8]
  const float* bias;       // may be null; per sample with stride bias_bstride (0 = shared)
  double* stats;           // may be null: [B][rows_alloc][2] += (sum, sum of squares) of the output
  int64_t bias_bstride;
  int batch, cin, rows, rows_alloc, D, H, W;
};

__global__ __launch_bounds__(S2_THREADS) void md_conv3_s2_kernel(const S2Args A) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[S2_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 2, wc = wid & 3;
  const int j = lane & 31, h = lane >> 5;

  const int D = A.D, H = A.H, W = A.W;
  const int Di = 2 * D, Hi = 2 * H, Wi = 2 * W;
  const int64_t P = (int64_t)D * H * W, Pin = 8 * P;
  const int ntx = W / S2_TX, nty = H / S2_TY, ntz = D / S2_TZ;
  const int tiles = ntx * nty * ntz;
  int bid = blockIdx.x;      // XCD-aware order: one contiguous run of tiles per XCD (block b runs on XCD b % 8)
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles, t = bid % tiles;
  const int x0 = (t % ntx) * S2_TX, y0 = ((t / ntx) % nty) * S2_TY, z0 = (t / (ntx * nty)) * S2_TZ;
  const int rt = blockIdx.y;
  const int ncc = A.cin / S2_KC;
  const int nslabs = ncc * 3, nrows = nslabs * 3;

  // ---- slab items of this thread: channel group kg = tid >> 8 (wave-uniform), positions (tid & 255) + 256 i ----------
  // Branch-free: an item outside the grid (far faces: the reference pads (0, 1)) or beyond the slab loads from offset 0 and
  // is zeroed by a mask after the split; an item beyond the slab is stored to a pad slot of the row layout (never read).
  const int kg = __builtin_amdgcn_readfirstlane(tid >> 8);
  int hsrc[S2_IT], hdst[S2_IT];      // source offset (uint4 units, relative to the (chunk, kd) base); LDS byte offset
  unsigned vmask = 0, zedge = 0;     // bit i: the item is inside the grid in y / x; its input plane falls off the grid for kd = 2
#pragma unroll
  for (int i = 0; i < S2_IT; ++i) {
    const int p = (tid & 255) + i * 256;
    const bool in_slab = p < S2_HPOS;
    const int hx = p % S2_XH, hy = (p / S2_XH) % S2_YH, zl = in_slab ? p / (S2_XH * S2_YH) : 0;
    const int iz = 2 * (z0 + zl), iy = 2 * y0 + hy, ix = 2 * x0 + hx;
    const bool live = in_slab && iy < Hi && ix < Wi;
    const int slot = in_slab ? (zl * S2_YH + hy) * S2_ROW + (hx & 1) * 9 + (hx >> 1) : S2_XH + (tid & 1);
    hdst[i] = S2_W_LDS_BYTES + ((kg * 2) * S2_HS + slot) * 16;
    hsrc[i] = live ? (int)((((int64_t)iz * Hi + iy) * Wi + ix) * 2) : 0;
    if (live) vmask |= 1u << i;
    if (iz + 2 >= Di) zedge |= 1u << i;
  }
  const uint4* xb = (const uint4*)A.x + (int64_t)b * (A.cin >> 3) * Pin * 2;
  const int64_t plane2 = (int64_t)Hi * Wi * 2;        // uint4 units per input z plane
  uint4 hreg[2 * S2_IT];
  unsigned lmask = 0;                                 // live bits of the slab in hreg
  auto act_issue = [&](int slab) {                    // slab = chunk * 3 + kd
    const int cc = slab / 3, kd = slab - cc * 3;
    const uint4* cb = xb + (int64_t)(cc * 2 + kg) * Pin * 2 + kd * plane2;
    lmask = kd == 2 ? (vmask & ~zedge) : vmask;
#pragma unroll
    for (int i = 0; i < S2_IT; ++i) {
      const uint4* sp = cb + ((kd == 2 && ((zedge >> i) & 1u)) ? 0 : hsrc[i]);
      hreg[2 * i] = sp[0]; hreg[2 * i + 1] = sp[1];
    }
  };
  auto act_transform = [&](int i) {                   // 8 fp32 channels -> (hi plane item, lo plane item) in place
    const uint4 r0 = hreg[2 * i], r1 = hreg[2 * i + 1];
    const uint32_t keep = ((lmask >> i) & 1u) ? 0xFFFFFFFFu : 0u;
    uint32_t hw[4], lw[4];
    md_split2(__uint_as_float(r0.x), __uint_as_float(r0.y), hw[0], lw[0]);
    md_split2(__uint_as_float(r0.z), __uint_as_float(r0.w), hw[1], lw[1]);
    md_split2(__uint_as_float(r1.x), __uint_as_float(r1.y), hw[2], lw[2]);
    md_split2(__uint_as_float(r1.z), __uint_as_float(r1.w), hw[3], lw[3]);
    hreg[2 * i] = make_uint4(hw[0] & keep, hw[1] & keep, hw[2] & keep, hw[3] & keep);
    hreg[2 * i + 1] = make_uint4(lw[0] & keep, lw[1] & keep, lw[2] & keep, lw[3] & keep);
  };
  auto act_commit = [&]() {
#pragma unroll
    for (int i = 0; i < S2_IT; ++i) {
      *(uint4*)(lds + hdst[i]) = hreg[2 * i];
      *(uint4*)(lds + hdst[i] + S2_HS * 16) = hreg[2 * i + 1];
    }
  };

  // ---- weights: row R = (chunk * 3 + kd) * 3 + kh = three consecutive tap tiles -------------------------------------------
  const uint4* wbase = A.wpk + (int64_t)rt * nrows * S2_WROW + tid;
  uint4 wreg0, wreg1, wreg2;
  auto w_issue = [&](int R) {
    const uint4* wp = wbase + (int64_t)R * S2_WROW;
    wreg0 = wp[0]; wreg1 = wp[S2_THREADS]; wreg2 = wp[2 * S2_THREADS];
  };
  auto w_commit = [&](int stage) {
    unsigned char* dst = lds + stage * (S2_WROW * 16) + tid * 16;
    *(uint4*)dst = wreg0;
    *(uint4*)(dst + S2_THREADS * 16) = wreg1;
    *(uint4*)(dst + 2 * S2_THREADS * 16) = wreg2;
  };

  // ---- fragment addresses: one VGPR each + immediates ------------------------------------------------------------------------
  // weights [kw 3][kg 2][plane 2][row 128]: byte = stage * 24576 + kw * 8192 + ((h * 2 + plane) * 128 + wr * 64 + rm * 32 + j) * 16
  const unsigned char* pA = lds + (h * 2 * S2_NT + wr * 64 + j) * 16;
  // slab [kg 2][plane 2][HS]: slot = (wc * 17 + 2 (cm * 4 + (j >> 3)) + kh) * 20 + (kw & 1) * 9 + (j & 7) + (kw >> 1)
  const unsigned char* pB = lds + S2_W_LDS_BYTES + (h * 2 * S2_HS + (wc * S2_YH + 2 * (j >> 3)) * S2_ROW + (j & 7)) * 16;

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
  Frags F[2];
#define S2_LOAD_FRAGS(Fx, STAGE, KH, KW)                                                                            \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) {                                                                \
    Fx.ahi[rm] = *(const bf16x8*)(pA + (STAGE) * (S2_WROW * 16) + (KW) * (S2_W_ITEMS * 16) + (rm * 32) * 16);       \
    Fx.alo[rm] = *(const bf16x8*)(pA + (STAGE) * (S2_WROW * 16) + (KW) * (S2_W_ITEMS * 16) + (S2_NT + rm * 32) * 16); \
  }                                                                                                                 \
  _Pragma("unroll") for (int cm = 0; cm < 2; ++cm) {                                                                \
    Fx.bhi[cm] = *(const bf16x8*)(pB + ((KH) * S2_ROW + ((KW) & 1) * 9 + ((KW) >> 1) + cm * 8 * S2_ROW) * 16);      \
    Fx.blo[cm] = *(const bf16x8*)(pB + (S2_HS + (KH) * S2_ROW + ((KW) & 1) * 9 + ((KW) >> 1) + cm * 8 * S2_ROW) * 16); \
  }
#define S2_MFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#define S2_MMA(Fx)                                                                                          \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    S2_MFMA(Fx.alo[rm], Fx.bhi[cm], acc[rm][cm]);                                                           \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    S2_MFMA(Fx.ahi[rm], Fx.blo[cm], acc[rm][cm]);                                                           \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)         \
    S2_MFMA(Fx.ahi[rm], Fx.bhi[cm], acc[rm][cm]);
  // 8 fragment reads of the next tap in the shadow of this tap's 12 MFMAs
#define S2_INTERLEAVE()                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {          \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);        \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        \
  }

  // ---- prologue: slab 0, weight rows 0 (committed) and 1 (in flight) -------------------------------------------------------
  w_issue(0);
  act_issue(0);
#pragma unroll
  for (int i = 0; i < S2_IT; ++i) act_transform(i);
  act_commit();
  w_commit(0);
  w_issue(nrows > 1 ? 1 : 0);
  __syncthreads();
  S2_LOAD_FRAGS(F[0], 0, 0, 0)

  // ---- main loop: two slabs (18 taps) per iteration.  Row r6 = u / 3 of the iteration reads weight stage r6 & 1; tap u
  // multiplies fragment set u & 1 while set (u + 1) & 1 is read for tap u + 1.  Per row:
  //   tap 0: W(R + 1) -> the other stage (its last readers passed the barrier of the row before), W(R + 2) requested;
  //          at kh = 0 the next slab's fp32 items are requested, at tap 1 of kh = 2 they are split (VALU beside the MFMAs)
  //   tap 2: barrier (at kh = 2: barrier, slab commit, barrier) BEFORE its MFMAs, whose operands are already in registers,
  //          then the fragments of the next row's first tap are read behind them.
  for (int sl = 0; sl < nslabs; sl += 2) {
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int kw = u % 3, kh = (u / 3) % 3, r6 = u / 3, stage = r6 & 1;
      const int R = sl * 3 + r6;
      const int slab = sl + u / 9;
      const bool more = slab + 1 < nslabs;
      Frags& Fc = F[u & 1];
      Frags& Fn = F[(u + 1) & 1];
      if (kw == 0) {
        w_commit(stage ^ 1);
        w_issue(R + 2 < nrows ? R + 2 : nrows - 1);
        if (kh == 0) act_issue(more ? slab + 1 : slab);
      }
      if (kh == 2 && kw == 1) {        // two rows (~2.5 us) after the request
#pragma unroll
        for (int i = 0; i < S2_IT; ++i) act_transform(i);
      }
      if (kw < 2) {
        S2_LOAD_FRAGS(Fn, stage, kh, kw + 1)
        S2_MMA(Fc)
        S2_INTERLEAVE()
      } else {
        if (kh == 2) {
          __syncthreads();             // every wave holds its last fragments of this slab in registers
          act_commit();
        }
        __syncthreads();               // W(R + 1) and (at kh = 2) the new slab are visible
        S2_LOAD_FRAGS(Fn, stage ^ 1, (kh + 1) % 3, 0)
        S2_MMA(Fc)
        S2_INTERLEAVE()
      }
    }
  }
#undef S2_LOAD_FRAGS
#undef S2_MMA
#undef S2_MFMA
#undef S2_INTERLEAVE

  // ---- epilogue: bias, 16-byte stores into F32B, optional GroupNorm sums (as md_conv3_main_kernel) ------------------------
  const int rows = A.rows, rows_alloc = A.rows_alloc;
  float* outp = A.out + (int64_t)b * (rows_alloc / 8) * P * 8;
  const float* biasp = A.bias ? A.bias + (int64_t)b * A.bias_bstride : nullptr;
  const bool want_stats = A.stats != nullptr;
  float st1[2][4][4], st2[2][4][4];
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) { st1[rm][q][e] = 0.f; st2[rm][q][e] = 0.f; }
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rt * S2_NT + wr * 64 + rm * 32 + 8 * q + 4 * h;
      if (row >= rows_alloc) continue;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (biasp != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (row + e < rows) bv[e] = biasp[row + e];
      }
#pragma unroll
      for (int cm = 0; cm < 2; ++cm) {
        const int y = cm * 4 + (j >> 3), x = j & 7;
        const int64_t gp = ((int64_t)(z0 + wc) * H + (y0 + y)) * W + (x0 + x);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[rm][cm][q * 4 + e] + bv[e];
          o4[e] = v;
          st1[rm][q][e] += v;
          st2[rm][q][e] += v * v;
        }
        *(f32x4*)(outp + ((int64_t)(row >> 3) * P + gp) * 8 + (row & 7)) = o4;
      }
    }
  if (want_stats) {      // workgroup-uniform
    auto row_sum = [](float v) {
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
      return v;
    };
#pragma unroll
    for (int rm = 0; rm < 2; ++rm)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          st1[rm][q][e] = row_sum(st1[rm][q][e]);
          st2[rm][q][e] = row_sum(st2[rm][q][e]);
        }
    __syncthreads();
    float* red = (float*)lds;   // [8 waves][2 rows of a half-wave][64 channels][2]
    if ((lane & 15) == 0) {
      const int jr = (lane >> 4) & 1;
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ch = rm * 32 + 8 * q + 4 * h + e;
            red[((wid * 2 + jr) * 64 + ch) * 2] = st1[rm][q][e];
            red[((wid * 2 + jr) * 64 + ch) * 2 + 1] = st2[rm][q][e];
          }
    }
    __syncthreads();
    if (tid < 256) {
      const int ch = tid >> 1, which = tid & 1;          // channel within the 128-row tile
      const int w0 = (ch >> 6) * 4, c64 = ch & 63;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) sum += red[(((w0 + (k >> 1)) * 2 + (k & 1)) * 64 + c64) * 2 + which];
      const int row = rt * S2_NT + ch;
      if (row < rows_alloc) atomicAdd(A.stats + ((int64_t)b * rows_alloc + row) * 2 + which, (double)sum);
    }
  }
}

extern "C" int md_conv3_s2(const float* x, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                           double* stats, int32_t batch, int32_t cin, int32_t rows, int32_t rows_alloc, int32_t D,
                           int32_t H, int32_t W, void* stream) {
  if (!x || !wpk || !out || batch <= 0 || rows <= 0 || rows_alloc < rows || (rows_alloc & 7)) return MD_ERR_BAD_ARG;
  if (cin <= 0 || (cin % (2 * S2_KC))) return MD_ERR_UNSUPPORTED;     // whole iterations of two slabs
  if (D <= 0 || H <= 0 || W <= 0 || (D % S2_TZ) || (H % S2_TY) || (W % S2_TX)) return MD_ERR_UNSUPPORTED;
  if ((int64_t)D * H * W * 8 * 2 >= (int64_t)1 << 31) return MD_ERR_UNSUPPORTED;       // 32-bit slab offsets
  S2Args a;
  a.x = x; a.wpk = (const uint4*)wpk; a.out = out; a.bias = bias; a.stats = stats; a.bias_bstride = bias_bstride;
  a.batch = batch; a.cin = cin; a.rows = rows; a.rows_alloc = rows_alloc; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / S2_TZ) * (H / S2_TY) * (W / S2_TX);
  const dim3 grid((unsigned)(tiles * batch), (unsigned)((rows + S2_NT - 1) / S2_NT));
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_conv3_s2_kernel, grid, dim3(S2_THREADS), 0, (hipStream_t)stream, a);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
