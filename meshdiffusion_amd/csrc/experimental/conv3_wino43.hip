// EXPERIMENTAL (off by default, MD_WINO43=1): the 3x3x3 stride-1 convolution through Winograd F(4,3) along w -- 6 products
// per 4 outputs, i.e. 1/2 of the direct form's matrix-core work (F(2,3) in conv3_wino.hip: 2/3), and an operand T of 1.5x
// the input (F(2,3): 2x).  Same reference ops as conv3_wino.hip (lib/diffusion/models/layers.py:118-124, :676-681, :618-623;
// ddpm_res64.py:174-176).  Error of one 128 -> 128 conv vs fp64 in bf16x3: 1.3e-5 (F(2,3) 5.5e-6, direct 4.5e-6;
// tools/wino_numerics.py).
//
// For an output quad y0..y3 (x = 4i .. 4i+3) with inputs d0..d5 = in[4i-1 .. 4i+4] and taps g0 g1 g2 along w:
//   t0 = 4d0 - 5d2 + d4        t1 = -4d1 - 4d2 + d3 + d4    t2 = 4d1 - 4d2 - d3 + d4
//   t3 = -2d1 - d2 + 2d3 + d4  t4 = 2d1 - d2 - 2d3 + d4     t5 = 4d1 - 5d3 + d5
//   G0 = g0/4   G1 = -(g0+g1+g2)/6   G2 = -(g0-g1+g2)/6   G3 = g0/24 + g1/12 + g2/6   G4 = g0/24 - g1/12 + g2/6   G5 = g2
//   m_f = t_f G_f (summed over channels and the 9 (kd, kh) taps)
//   y0 = m0+m1+m2+m3+m4   y1 = m1-m2+2m3-2m4   y2 = m1+m2+4m3+4m4   y3 = m1-m2+8m3-8m4+m5
//
// Six frequencies on the four frequency-private waves of conv3_wino.hip: wave w owns frequency w (128 rows x 64 quad
// columns = 8 accumulator tiles) plus one half of the rows of frequency 4 + (w >> 1) (4 tiles): 12 tiles = 192 AccVGPRs,
// 36 MFMAs per step.  Its halo image holds two frequency slices (2 x 480 entries = the same 15 pieces of 1 KB per chunk);
// the slices of frequencies 4 and 5 are fetched by two waves each.  Workgroup tile = 128 Cout x (4 x 8 x 8) positions.
#include "md_common.h"
#include "meshdiffusion_hip_experimental.h"

namespace {
constexpr int W4_THREADS = 256;
constexpr int W4_TZ = 4, W4_TY = 8, W4_TX = 8;          // output tile; 2 quads along w
constexpr int W4_TPOS = 6 * 10 * 2;                      // 120 halo entries (dz, hy, quad) per (frequency, k-group, plane)
constexpr int W4_HBUF = 2 * 4 * W4_TPOS * 16;            // 15360 B: [fsel 2][h 2][plane 2][120][16 B] = one chunk, two slices
constexpr int W4_NPC = W4_HBUF / 1024;                   // 15 pieces
constexpr int W4_XSTRIDE = 36;
constexpr int W4_XREGION = 6 * 64 * W4_XSTRIDE;          // floats: [f 6][col 64][36]
constexpr int W4_RED = 4 * 2 * 32 * 2;
constexpr int W4_EPI_BYTES = 2 * W4_XREGION * 4 + 2 * W4_RED * 4;                                   // 114688 B
constexpr int W4_LDS_BYTES = 4 * 2 * W4_HBUF > W4_EPI_BYTES ? 4 * 2 * W4_HBUF : W4_EPI_BYTES;       // 122880 B

__device__ const uint4 w4_zero16 = {0u, 0u, 0u, 0u};
typedef uint32_t w4_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 w4_gload16(const void* p) {
  return __builtin_bit_cast(uint4, *(__attribute__((address_space(1))) const w4_u32x4*)(uintptr_t)p);
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// md_wino43_prep: one thread = one (sample, 8-channel group, z, y, quad)
//   T[B][C/8][6][2][D][H][W/4][8 bf16]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void md_wino43_prep_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                             int c1, int c2, const float* __restrict__ ac, int silu, int ups,
                                                             uint4* __restrict__ T, int batch, int D, int H, int W) {
  const int Wq = W >> 2;
  const int64_t Nq = (int64_t)D * H * Wq;
  const int CG = (c1 + c2) >> 3;
  const int64_t n = (int64_t)batch * CG * Nq;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  const int64_t pos4 = id % Nq;
  const int cg = (int)((id / Nq) % CG);
  const int b = (int)(id / (Nq * CG));
  const int qd = (int)(pos4 % Wq), y = (int)((pos4 / Wq) % H), z = (int)(pos4 / ((int64_t)Wq * H));
  int Di = D, Hi = H, Wi = W, zs = z, ys = y;
  if (ups) { Di >>= 1; Hi >>= 1; Wi >>= 1; zs >>= 1; ys >>= 1; }
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  const float* src = (cg * 8 < c1) ? x1 + ((int64_t)b * (c1 >> 3) + cg) * Pin * 8
                                   : x2 + ((int64_t)b * (c2 >> 3) + (cg - (c1 >> 3))) * Pin * 8;
  float a[8], c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = 1.f; c[e] = 0.f; }
  if (ac != nullptr) {
    const f32x4* ap = (const f32x4*)(ac + ((int64_t)b * (c1 + c2) + cg * 8) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ap[q];
      a[2 * q] = v[0]; c[2 * q] = v[1]; a[2 * q + 1] = v[2]; c[2 * q + 1] = v[3];
    }
  }
  float d[6][8];
#pragma unroll
  for (int q = 0; q < 6; ++q) {
    const int xw = 4 * qd - 1 + q;
    const bool live = xw >= 0 && xw < W;             // the conv pads the ACTIVATED tensor with zeros
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    if (live) {
      const int xs = ups ? (xw >> 1) : xw;
      const f32x4* p = (const f32x4*)(src + (((int64_t)zs * Hi + ys) * Wi + xs) * 8);
      v0 = p[0]; v1 = p[1];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float yv = e < 4 ? v0[e] : v1[e - 4];
      if (ac != nullptr) {
        yv = yv * a[e] + c[e];
        if (silu) yv = yv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(yv * -1.4426950408889634f));
      }
      d[q][e] = live ? yv : 0.f;
    }
  }
  uint4* out = T + ((int64_t)b * CG + cg) * 12 * Nq + pos4;       // [f 6][plane 2][Nq] items of 16 B
#pragma unroll
  for (int f = 0; f < 6; ++f) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d0 = d[0][e], d1 = d[1][e], d2 = d[2][e], d3 = d[3][e], d4 = d[4][e], d5 = d[5][e];
      t[e] = f == 0 ? (4.f * d0 - 5.f * d2) + d4
           : f == 1 ? (d3 + d4) - 4.f * (d1 + d2)
           : f == 2 ? (d4 - d3) + 4.f * (d1 - d2)
           : f == 3 ? (d4 - d2) + 2.f * (d3 - d1)
           : f == 4 ? (d4 - d2) - 2.f * (d3 - d1)
                    : (4.f * d1 - 5.f * d3) + d5;
    }
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) md_split2(t[2 * q], t[2 * q + 1], hw[q], lw[q]);
    out[(int64_t)(f * 2) * Nq] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    out[(int64_t)(f * 2 + 1) * Nq] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// md_wino43_pack_weights: one thread = one 16-byte item
//   layout [cout/128][cin/16][tap (kd,kh) 9][f 6][row tile 4][plane 2][h 2][row 32][8 bf16]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void md_wino43_pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ wpk,
                                                                     int cout, int cin, int64_t s_row, int64_t s_k, int flip) {
  const int64_t n = (int64_t)cout * cin * 54 / 4;        // cout * cin * 9 taps * 6 values * 2 planes / 8
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  int64_t r = id;
  const int row = (int)(r % 32); r /= 32;
  const int h = (int)(r % 2); r /= 2;
  const int plane = (int)(r % 2); r /= 2;
  const int rtile = (int)(r % 4); r /= 4;
  const int f = (int)(r % 6); r /= 6;
  const int tap = (int)(r % 9); r /= 9;
  const int nchunk = cin / 16;
  const int chunk = (int)(r % nchunk); r /= nchunk;
  const int ct = (int)r;
  const int co = (ct * 4 + rtile) * 32 + row;
  uint32_t word[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t half[2];
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
      const int ci = chunk * 16 + h * 8 + 2 * q + e2;
      const float* g = w + (int64_t)co * s_row + (int64_t)ci * s_k;
      const int t0 = tap * 3;
      const float g0 = g[flip ? 26 - t0 : t0], g1 = g[flip ? 25 - t0 : t0 + 1], g2 = g[flip ? 24 - t0 : t0 + 2];
      const float G = f == 0 ? g0 * 0.25f
                    : f == 1 ? ((g0 + g1) + g2) * (-1.0f / 6.0f)
                    : f == 2 ? ((g0 - g1) + g2) * (-1.0f / 6.0f)
                    : f == 3 ? (g0 * (1.0f / 24.0f) + g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f)
                    : f == 4 ? (g0 * (1.0f / 24.0f) - g1 * (1.0f / 12.0f)) + g2 * (1.0f / 6.0f)
                             : g2;
      uint32_t hi, lo;
      md_split(G, hi, lo);
      half[e2] = plane ? lo : hi;
    }
    word[q] = half[0] | (half[1] << 16);
  }
  wpk[id] = make_uint4(word[0], word[1], word[2], word[3]);
}

// ------------------------------------------------------------------------------------------------------------------
// md_conv3_wino43
// ------------------------------------------------------------------------------------------------------------------
struct W4Args {
  const uint4* T;          // [B][cin/8][6][2][D][H][W/4] items of 16 B
  const uint4* wpk;
  float* out;              // F32B [B][cout/8][P][8]
  const float* bias;       // may be null; per sample with stride bias_bstride (0 = shared)
  const float* residual;   // F32B like out, may be null
  double* stats;           // may be null
  int64_t bias_bstride, res_bstride;
  int batch, cin, cout, D, H, W;
};

__global__ __launch_bounds__(W4_THREADS) void md_conv3_wino43_kernel(const W4Args A) {
  __shared__ __attribute__((aligned(16))) unsigned char w4_smem[W4_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  const int fo = wid;                       // the frequency this wave owns entirely
  const int fs = 4 + (wid >> 1);            // the frequency it shares: row tiles 2 hs, 2 hs + 1
  const int hs = wid & 1;

  const int D = A.D, H = A.H, W = A.W, Wq = W >> 2;
  const int64_t P = (int64_t)D * H * W, Nq = P >> 2;
  const int ntx = W / W4_TX, nty = H / W4_TY, ntz = D / W4_TZ;
  const int tiles = ntx * nty * ntz;
  int bid = blockIdx.x;     // XCD-aware order: one contiguous run of tiles per XCD (block b runs on XCD b % 8)
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles, t = bid % tiles;
  const int x0 = (t % ntx) * W4_TX, y0 = ((t / ntx) % nty) * W4_TY, z0 = (t / (ntx * nty)) * W4_TZ;
  const int rtb = blockIdx.y;
  const int CG = A.cin >> 3, nchunk = A.cin >> 4;
  const int nsteps = nchunk * 9;

  unsigned char* my_smem = w4_smem + wid * 2 * W4_HBUF;      // this wave's two halo buffers

  // ---- halo pieces: piece k = entries [64 k, 64 k + 64) of the linear image [fsel][h][plane][dz][hy][quad] --------------
  const uint4* zsrc = &w4_zero16;
  auto halo_load = [&](int chunk, int k) -> uint4 {
    const int e = k * 64 + lane;
    const int fsel = e / (4 * W4_TPOS), r = e % (4 * W4_TPOS);
    const int hp = r / W4_TPOS, tp = r % W4_TPOS;
    const int dz = tp / 20, hy = (tp >> 1) % 10, qd = tp & 1;
    const int z = z0 + dz - 1, y = y0 + hy - 1;
    const bool live = (z >= 0) & (z < D) & (y >= 0) & (y < H);
    const int f = fsel ? fs : fo;
    // [B][cg][f][plane][Nq]: cg = 2 chunk + (hp >> 1), plane = hp & 1
    const int64_t off = ((((int64_t)b * CG + 2 * chunk + (hp >> 1)) * 6 + f) * 2 + (hp & 1)) * Nq + ((int64_t)z * H + y) * Wq + (x0 >> 2) + qd;
    const uint4* src = live ? A.T + off : zsrc;
    return w4_gload16(src);
  };
  auto halo_store = [&](int buf, int k, const uint4& v) { *(uint4*)(my_smem + buf * W4_HBUF + k * 1024 + lane * 16) = v; };
  // weights of step s: 48 fragments of 1 KB ([f 6][rt 4][plane 2]); own frequency: 8, shared: the 4 of its two row tiles
  const uint4* wbase = A.wpk + ((int64_t)rtb * nsteps) * 6 * 512 + lane;                       // + s * 3072 + (f * 8 + frag) * 64
  auto load_A = [&](int s, bf16x8 (&dst)[12]) {
    const uint4* wp = wbase + (int64_t)s * 3072;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = __builtin_bit_cast(bf16x8, wp[(fo * 8 + i) * 64]);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[8 + i] = __builtin_bit_cast(bf16x8, wp[(fs * 8 + hs * 4 + i) * 64]);
  };
  // halo fragment of column tile ct (= output planes z0 + 2 ct, z0 + 2 ct + 1), tap (kd, kh):
  //   entry (2 ct + zz + kd) * 20 + (yy + kh) * 2 + qd   with lane j = zz * 16 + yy * 2 + qd
  const unsigned char* vB = my_smem + h * 2 * W4_TPOS * 16 + ((j >> 4) * 20 + (j & 15)) * 16;
  auto read_B = [&](int tap, int buf, bf16x8 (&dst)[8]) {
    const int kd = tap / 3, kh = tap % 3;
    const unsigned char* p = vB + buf * W4_HBUF + (kd * 20 + kh * 2) * 16;
#pragma unroll
    for (int fsel = 0; fsel < 2; ++fsel)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        dst[(fsel * 2 + ct) * 2] = *(const bf16x8*)(p + (fsel * 4 * W4_TPOS + ct * 40) * 16);
        dst[(fsel * 2 + ct) * 2 + 1] = *(const bf16x8*)(p + (fsel * 4 * W4_TPOS + W4_TPOS + ct * 40) * 16);
      }
  };

  f32x16 acc[12];          // tiles 0..7: own frequency [rt 4][ct 2]; 8..11: shared frequency [rt2 2][ct 2]
#pragma unroll
  for (int i = 0; i < 12; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    asm volatile("" : "+a"(acc[i]));
  }
  bf16x8 Ar[2][12];     // weights [step parity][own: rt * 2 + plane | shared: 8 + rt2 * 2 + plane]
  bf16x8 Bf[2][8];      // halo    [step parity][(fsel * 2 + ct) * 2 + plane]
  uint4 hst[2][3];

  // ---- prologue ---------------------------------------------------------------------------------------------------
  {
    uint4 h0[W4_NPC];
#pragma unroll
    for (int k = 0; k < W4_NPC; ++k) h0[k] = halo_load(0, k);
    load_A(0, Ar[0]);
#pragma unroll
    for (int k = 0; k < W4_NPC; ++k) halo_store(0, k, h0[k]);
  }
  read_B(0, 0, Bf[0]);

  // ---- main loop (the structure of md_conv3_wino_kernel; weights requested one step ahead: two register sets) ------------
  // tile i: A fragments (hi, lo) = a[ai], a[ai + 1], B fragments (hi, lo) = bq[bi], bq[bi + 1]
#define W4_AI(i) ((i) < 8 ? ((i) >> 1) * 2 : 8 + (((i) - 8) >> 1) * 2)
#define W4_BI(i) ((i) < 8 ? ((i) & 1) * 2 : (2 + ((i) & 1)) * 2)
#define W4_MFMA(PASS, i, a, bq)                                                                                   \
  acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[W4_AI(i) + ((PASS) == 0 ? 1 : 0)], bq[W4_BI(i) + ((PASS) == 1 ? 1 : 0)], \
                                                   acc[i], 0, 0, 0)
  for (int c0 = 0; c0 < nchunk; c0 += 2) {
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int tap = u % 9, cpar = u / 9;
      const int s = c0 * 9 + u;
      const int sw = s + 1 < nsteps ? s + 1 : nsteps - 1;
      const int cn = c0 + cpar + 1 < nchunk ? c0 + cpar + 1 : nchunk - 1;
      bf16x8 (&Aw)[12] = Ar[u & 1], (&Bc)[8] = Bf[u & 1], (&Bn)[8] = Bf[(u + 1) & 1];
      load_A(sw, Ar[(u + 1) & 1]);
      if (tap < 8) read_B(tap + 1, cpar, Bn);
      else read_B(0, cpar ^ 1, Bn);
#pragma unroll
      for (int m = 0; m < 12; ++m) W4_MFMA(0, m, Aw, Bc);
#pragma unroll
      for (int i_ = 0; i_ < 4; ++i_) {                 // 8 LDS reads and 12 global loads under the 12 MFMAs of the first pass
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (tap >= 2 && tap <= 6) {
#pragma unroll
        for (int q = 0; q < 3; ++q) halo_store(cpar ^ 1, (tap - 2) * 3 + q, hst[tap & 1][q]);
      }
      if (tap <= 4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) hst[tap & 1][q] = halo_load(cn, tap * 3 + q);
      }
#pragma unroll
      for (int m = 0; m < 24; ++m) {
        if (m < 12) W4_MFMA(1, m, Aw, Bc); else W4_MFMA(2, m - 12, Aw, Bc);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef W4_MFMA
#undef W4_AI
#undef W4_BI

  // ---- epilogue: six frequencies of an output quad meet through LDS, one 32-row tile per round ------------------------------
  float* xreg = (float*)w4_smem;
  float* red = xreg + 2 * W4_XREGION;
  const int rows_total = A.cout;
  float* outp = A.out + (int64_t)b * rows_total * P;
  const float* resp = A.residual ? A.residual + (int64_t)b * A.res_bstride : nullptr;
  const float* biasp = A.bias ? A.bias + (int64_t)b * A.bias_bstride : nullptr;
  const bool want_stats = A.stats != nullptr;
  // this wave finishes column tile ec = wid >> 1 for the row quarters q = 2 (wid & 1), 2 (wid & 1) + 1 of every round:
  // lane j -> output plane z0 + 2 ec + (j >> 4), row y0 + ((j >> 1) & 7), quad j & 1: x = x0 + 4 (j & 1) + {0, 1, 2, 3}
  const int ec = wid >> 1, qh = wid & 1;
  const int64_t gp0 = ((int64_t)(z0 + 2 * ec + (j >> 4)) * H + (y0 + ((j >> 1) & 7))) * W + x0 + 4 * (j & 1);
  auto row_sum = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
  };
  auto flush_stats = [&](int r) {     // channel ch = 8 q + 4 h + e was summed by the waves with (wid & 1) == q >> 1
    if (tid < 64) {
      const int ch = tid >> 1, which = tid & 1;
      const float* rb = red + (r & 1) * W4_RED;
      const int wq = ch >> 4;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += rb[(((wq + 2 * (k >> 1)) * 2 + (k & 1)) * 32 + ch) * 2 + which];
      const int row = rtb * 128 + r * 32 + ch;
      atomicAdd(A.stats + ((int64_t)b * rows_total + row) * 2 + which, (double)sum);
    }
  };
  f32x4 pbias[2][2], pres[2][2][4];
  auto prefetch = [&](int r) {
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int row = rtb * 128 + r * 32 + 8 * (2 * qh + q2) + 4 * h;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      pbias[r & 1][q2] = biasp != nullptr ? *(const f32x4*)(biasp + row) : z;
#pragma unroll
      for (int xx = 0; xx < 4; ++xx)
        pres[r & 1][q2][xx] = resp != nullptr ? *(const f32x4*)(resp + ((int64_t)(row >> 3) * P + gp0 + xx) * 8 + (row & 7)) : z;
    }
  };
  prefetch(0);
  __syncthreads();                                       // every wave is done with its halo buffers (the exchange area aliases them)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float* xr = xreg + (r & 1) * W4_XREGION;
    // m_fo of rows 32 r ..: accumulator tiles [r][ct]; m_fs of the same rows from the wave that holds that half
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[r * 2 + ct][q * 4 + e];
        *(f32x4*)(xr + ((fo * 64 + ct * 32 + j) * W4_XSTRIDE + 8 * q + 4 * h)) = v;
      }
    if (hs == (r >> 1)) {
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[8 + (r & 1) * 2 + ct][q * 4 + e];
          *(f32x4*)(xr + ((fs * 64 + ct * 32 + j) * W4_XSTRIDE + 8 * q + 4 * h)) = v;
        }
    }
    if (r < 3) prefetch(r + 1);
    __syncthreads();
    if (want_stats && r > 0) flush_stats(r - 1);
    float s1[2][4], s2[2][4];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
      const int q = 2 * qh + q2;
      f32x4 m[6];
#pragma unroll
      for (int ff = 0; ff < 6; ++ff) m[ff] = *(const f32x4*)(xr + ((ff * 64 + ec * 32 + j) * W4_XSTRIDE + 8 * q + 4 * h));
      const int row = rtb * 128 + r * 32 + 8 * q + 4 * h;
      const f32x4 bv = pbias[r & 1][q2];
      float* op = outp + ((int64_t)(row >> 3) * P + gp0) * 8 + (row & 7);
      f32x4 o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ps = m[1][e] + m[2][e], pd = m[1][e] - m[2][e], qs = m[3][e] + m[4][e], qd2 = m[3][e] - m[4][e];
        float v0 = (m[0][e] + ps) + qs;
        float v1 = pd + 2.f * qd2;
        float v2 = ps + 4.f * qs;
        float v3 = (pd + 8.f * qd2) + m[5][e];
        v0 += bv[e]; v1 += bv[e]; v2 += bv[e]; v3 += bv[e];
        v0 += pres[r & 1][q2][0][e]; v1 += pres[r & 1][q2][1][e]; v2 += pres[r & 1][q2][2][e]; v3 += pres[r & 1][q2][3][e];
        o[0][e] = v0; o[1][e] = v1; o[2][e] = v2; o[3][e] = v3;
        s1[q2][e] = (v0 + v1) + (v2 + v3);
        s2[q2][e] = (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
#pragma unroll
      for (int xx = 0; xx < 4; ++xx) *(f32x4*)(op + xx * 8) = o[xx];
    }
    if (want_stats) {
      float* rb = red + (r & 1) * W4_RED;
#pragma unroll
      for (int q2 = 0; q2 < 2; ++q2)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float a1 = row_sum(s1[q2][e]), a2 = row_sum(s2[q2][e]);
          if ((lane & 15) == 0) {
            const int jr = (lane >> 4) & 1, ch = 8 * (2 * qh + q2) + 4 * h + e;
            rb[((wid * 2 + jr) * 32 + ch) * 2] = a1;
            rb[((wid * 2 + jr) * 32 + ch) * 2 + 1] = a2;
          }
        }
    }
  }
  if (want_stats) {
    __syncthreads();
    flush_stats(3);
  }
}

// ------------------------------------------------------------------------------------------------------------------
extern "C" int64_t md_wino43_operand_bytes(int32_t batch, int32_t cin, int32_t D, int32_t H, int32_t W) {
  if (batch <= 0 || cin <= 0 || (cin & 7) || D <= 0 || H <= 0 || W <= 0 || (W & 3)) return MD_ERR_BAD_ARG;
  return (int64_t)batch * (cin / 8) * 12 * ((int64_t)D * H * (W / 4)) * 16;
}

extern "C" int md_wino43_prep(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                              int32_t ups, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!x1 || !t_out || batch <= 0 || c1 <= 0 || c2 < 0 || (c1 & 7) || (c2 & 7) || (c2 > 0 && !x2)) return MD_ERR_BAD_ARG;
  if (silu && !ac) return MD_ERR_BAD_ARG;
  if (D <= 0 || H <= 0 || W <= 0 || (W & 3) || (ups && ((D | H | W) & 1))) return MD_ERR_BAD_ARG;
  const int64_t n = (int64_t)batch * ((c1 + c2) / 8) * D * H * (W / 4);
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino43_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac, silu,
                     ups, (uint4*)t_out, batch, D, H, W);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int64_t md_wino43_weight_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  return (int64_t)cout * cin * 54 * 4;
}

extern "C" int md_wino43_pack_weights(const float* w, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                                      int32_t flip, void* stream) {
  if (!w || !wpk || cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  const int64_t n = (int64_t)cout * cin * 54 / 4;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino43_pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (uint4*)wpk, cout, cin, s_row, s_k, flip);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_conv3_wino43(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                               const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                               int32_t cout, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!t_in || !wpk || !out || batch <= 0) return MD_ERR_BAD_ARG;
  if (cin <= 0 || cout <= 0 || (cin % 32) || (cout % 128)) return MD_ERR_UNSUPPORTED;
  if (D <= 0 || H <= 0 || W <= 0 || (D % W4_TZ) || (H % W4_TY) || (W % W4_TX)) return MD_ERR_UNSUPPORTED;
  W4Args a;
  a.T = (const uint4*)t_in; a.wpk = (const uint4*)wpk; a.out = out; a.bias = bias; a.residual = residual; a.stats = stats;
  a.bias_bstride = bias_bstride; a.res_bstride = res_bstride;
  a.batch = batch; a.cin = cin; a.cout = cout; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / W4_TZ) * (H / W4_TY) * (W / W4_TX);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_conv3_wino43_kernel, dim3((unsigned)(tiles * batch), (unsigned)(cout / 128)), dim3(W4_THREADS), 0,
                     (hipStream_t)stream, a);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
