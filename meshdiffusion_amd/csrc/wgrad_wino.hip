// Weight gradient of the 3x3x3 stride-1 convolutions in the Winograd F(2,3)-along-w domain (autograd of
// lib/diffusion/models/layers.py:118-124 for the layers whose forward / data gradient run through csrc/conv3_wino.hip).
//
// For an output pair (x = 2i, 2i+1) with output gradients dy0 dy1 and the activated inputs d0..d3 = a[2i-1 .. 2i+2]:
//     dg_k = sum_pairs (dy0 d_k + dy1 d_{k+1}),  k = 0, 1, 2                       (6 products per pair)
// is the transpose of the forward's bilinear form, so it has the transposed minimal algorithm (4 products per pair):
//     t = B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)        -- exactly the operand T the forward conv consumed
//     u = (dy0, dy0 + dy1, dy0 - dy1, dy1)                    -- md_wino_prep_dual's second output
//     n_f = sum_pairs u_f t_f ;   dg0 = n0 + (n1 + n2)/2,  dg1 = (n1 - n2)/2,  dg2 = (n1 + n2)/2 - n3.
// The 27-tap contraction over positions becomes 9 (kd, kh) row shifts x 4 frequencies over HALF the columns = 2/3 of the
// matrix-core work of csrc/wgrad.hip, and the w shift of the taps is gone: a (kd, kh) tap is a plain ROW shift of T.
//
// Operands stay in the channel-innermost layout the conv kernels use, [B][C/8][f 4][plane 2][D][H][W/2][8 bf16] (hi / lo
// planes): no PB16 re-layout pass.  The contraction index of an MFMA (8 consecutive k per lane) is the PAIR index here,
// which is strided in that layout; the LDS image of a row is [plane][8-pair segment][32-channel block][pair 8][32 ch] and
// fragments come out of it with ds_read_b64_tr_b16 (gfx950 transpose read: a 16-lane group reads a [4 pair][16 channel]
// block and each lane receives one channel's 4 pairs; layout and semantics pinned by tools/probes/tr_probe.hip).
//
// One workgroup (4 waves, one per SIMD, 192 accumulator registers each) = one frequency f, one kd, a 128 co x 128 ci tile
// and the THREE kh taps, over a range of (sample, z) planes: the dY row (b, z, y) is paired with the T rows
// (b, z + kd - 1, y - 1 .. y + 1).  Rows stream through LDS: U double-buffered, T in a 4-slot ring (each T row serves three
// dY rows); the row -1 / H of a plane is an all-zero ring entry shared by consecutive planes, so the whole range is ONE
// uniform stream of steps: { request T[n+2], U[n+1] ; 72 MFMAs on U[n] x T[n-1..n+1] ; store the requested rows ; barrier }.
// Partial sums go to a workspace; md_wgrad_wino_reduce sums the K ranges in a fixed order, applies the output transform and
// accumulates into dW (deterministic).
//
// Arithmetic: bf16x3 (lo*hi + hi*lo + hi*hi, fp32 accumulate), like the forward.
#include "md_common.h"

namespace {

constexpr int WW_THREADS = 256;
typedef short ww_v4i16 __attribute__((ext_vector_type(4)));

struct WwArgs {
  const uint4* U;      // transformed dY    [B][co/8][4][2][Ph] items of 16 B
  const uint4* T;      // transformed input [B][ci/8][4][2][Ph]
  float* partial;      // [ksplit][f 4][kd 3][kh 3][co][ci]
  int batch, co, ci, D, H, Wp;
  int co_tiles, ci_tiles, ksplit;
};

__device__ __forceinline__ uint4 ww_gload16(const uint4* p) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(uint4, *(__attribute__((address_space(1))) const u32x4*)(uintptr_t)p);
}

// 8 consecutive pairs of one channel: two transpose reads (pairs 0-3 and 4-7 of the lane's segment)
__device__ __forceinline__ bf16x8 ww_frag(const unsigned char* p) {
  typedef __attribute__((address_space(3))) ww_v4i16 lds_v4;
  const ww_v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
  const ww_v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + 256));
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// Stream element cursor: element n of a workgroup's stream = (valid plane v, j): j = 0 the shared zero row / no dY row,
// j >= 1 row j - 1 of the plane.
struct WwCursor {
  int v, j;
  __device__ void next(int H) { if (++j > H) { j = 0; ++v; } }
};

// NSEG = 8-pair segments per row: 4 (W = 64) or 2 (W = 32)
template <int NSEG>
__global__ __launch_bounds__(WW_THREADS) void md_wgrad_wino_kernel(const WwArgs g) {
  constexpr int SEGB = 2048;                       // [32-ch block 4][pair 8][32 ch] bf16
  constexpr int PLANEB = NSEG * SEGB;
  constexpr int ROWB = 2 * PLANEB;                 // one operand row (128 channels, both planes): 16 KB / 8 KB
  constexpr int NQ = NSEG;                         // 1 KB load instructions per wave, row and operand
  constexpr int KS = NSEG / 2;                     // MFMA k-steps (16 pairs) per row
  __shared__ __attribute__((aligned(16))) unsigned char smem[6 * ROWB];
  unsigned char* const ubuf = smem;                // 2 rows of U
  unsigned char* const tring = smem + 2 * ROWB;    // 4 rows of T

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;         // 64-row (co) and 64-column (ci) half of the 128 x 128 tile

  // ---- work item: runs of 32 consecutive items per XCD (block b lands on XCD b % 8), so the kd / f / tile units of one
  //      K range -- which read the same rows -- share an L2
  const int units = g.co_tiles * g.ci_tiles * 12;
  const int bx = blockIdx.x, slot = bx >> 3;
  const int w = (slot >> 5) * 256 + (bx & 7) * 32 + (slot & 31);
  if (w >= g.ksplit * units) return;
  const int r = w / units, u = w - r * units;
  const int kd = u % 3, f = (u / 3) & 3;
  const int tci = (u / 12) % g.ci_tiles, tco = u / (12 * g.ci_tiles);
  const int D = g.D, H = g.H, Wp = g.Wp;
  const int64_t Ph = (int64_t)D * H * Wp;
  // valid planes of this kd: (b, z) with 0 <= z + kd - 1 < D; index v -> b = v / nz, z = v % nz + z_first
  const int nz = kd == 1 ? D : D - 1, z_first = kd == 0 ? 1 : 0;
  const int NV = g.batch * nz;
  const int v0 = (int)((int64_t)NV * r / g.ksplit), v1 = (int)((int64_t)NV * (r + 1) / g.ksplit);

  // ---- per-lane constants of the staging loads: instruction qq of a row = (plane, segment, 64-channel half); lane =
  //      (channel group of the half, pair of the segment): 8 consecutive lanes read 128 contiguous bytes
  int64_t goff[NQ];
  int loff[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int qq = wave * NQ + q;
    const int plane = qq & 1, seg = (qq >> 1) % NSEG, half = qq / (2 * NSEG);
    const int cg = half * 8 + (lane >> 3), k = lane & 7;
    goff[q] = ((int64_t)cg * 8 + f * 2 + plane) * Ph + seg * 8 + k;
    loff[q] = plane * PLANEB + seg * SEGB + (cg >> 2) * 512 + k * 64 + (cg & 3) * 16;
  }
  const uint4* const ubase = g.U + ((int64_t)tco * 16) * 8 * Ph;      // + b * (co/8) * 8 * Ph + row * Wp + goff
  const uint4* const tbase = g.T + ((int64_t)tci * 16) * 8 * Ph;
  const int64_t u_bstride = (int64_t)(g.co >> 3) * 8 * Ph, t_bstride = (int64_t)(g.ci >> 3) * 8 * Ph;
  auto u_row = [&](const WwCursor& c) -> const uint4* {               // c.j >= 1
    const int b = c.v / nz, z = c.v % nz + z_first;
    return ubase + (int64_t)b * u_bstride + ((int64_t)z * H + (c.j - 1)) * Wp;
  };
  auto t_row = [&](const WwCursor& c) -> const uint4* {
    const int b = c.v / nz, z = c.v % nz + z_first + kd - 1;
    return tbase + (int64_t)b * t_bstride + ((int64_t)z * H + (c.j - 1)) * Wp;
  };

  // ---- fragment addresses: lane = (segment of the k-step s, 16-channel half sub, i); see the file header
  const int s = lane >> 5, sub = (lane >> 4) & 1, i = lane & 15;
  const int frag_lo = s * SEGB + (i >> 2) * 64 + sub * 32 + (i & 3) * 8;
  const int a_off = frag_lo + (wr * 2) * 512;      // + rt * 512 + plane * PLANEB + ks * 2 * SEGB
  const int b_off = frag_lo + (wc * 2) * 512;

  f32x16 acc[3][2][2];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kh][rt][ct][e] = 0.f;

  const int nsteps = (v1 - v0) * (H + 1) + 1;      // stream elements; the last one is the closing zero row
  if (v1 > v0) {
    uint4 st_u[NQ], st_t[NQ];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    // ---- prologue: T[0] (zero row), T[1] (row 0 of the first plane); U[0] does not exist, U[1] is requested in step 0
    WwCursor ct2 = {v0, 1};                        // the T element about to be requested
    {
      const uint4* tp = t_row(ct2);
#pragma unroll
      for (int q = 0; q < NQ; ++q) st_t[q] = ww_gload16(tp + goff[q]);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        *(uint4*)(tring + 0 * ROWB + loff[q]) = zero4;
        *(uint4*)(tring + 1 * ROWB + loff[q]) = st_t[q];
      }
      ct2.next(H);
    }
    __syncthreads();
    WwCursor cu1 = {v0, 1};                        // the U element about to be requested (n + 1)
    WwCursor cn = {v0, 0};                         // the element being computed (n)
    for (int n = 0; n < nsteps; ++n) {
      // ---- requests: T[n+2], U[n+1]
      const bool t_req = n + 2 < nsteps, t_zero = t_req && (ct2.j == 0 || ct2.v >= v1);
      const bool u_req = n + 1 < nsteps && cu1.j != 0 && cu1.v < v1;
      if (t_req && !t_zero) {
        const uint4* tp = t_row(ct2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) st_t[q] = ww_gload16(tp + goff[q]);
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) st_t[q] = zero4;
      }
      if (u_req) {
        const uint4* up = u_row(cu1);
#pragma unroll
        for (int q = 0; q < NQ; ++q) st_u[q] = ww_gload16(up + goff[q]);
      }
      // keep the requests up here: the scheduler otherwise sinks them next to the LDS stores below
      __builtin_amdgcn_sched_barrier(0);
      // ---- compute: U[n] x T[n-1], T[n], T[n+1]
      if (cn.j != 0) {
        const unsigned char* up = ubuf + (n & 1) * ROWB + a_off;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          bf16x8 ah[2], al[2];
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) {
            ah[rt] = ww_frag(up + ks * 2 * SEGB + rt * 512);
            al[rt] = ww_frag(up + ks * 2 * SEGB + rt * 512 + PLANEB);
          }
#pragma unroll
          for (int kh = 0; kh < 3; ++kh) {
            const unsigned char* tp = tring + ((n + 3 + kh) & 3) * ROWB + b_off + ks * 2 * SEGB;     // T[n - 1 + kh]
            bf16x8 bh[2], bl[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              bh[ct] = ww_frag(tp + ct * 512);
              bl[ct] = ww_frag(tp + ct * 512 + PLANEB);
            }
            // three passes over the four independent accumulators: no back-to-back dependent MFMAs
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
                acc[kh][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[rt], bh[ct], acc[kh][rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
                acc[kh][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bl[ct], acc[kh][rt][ct], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
              for (int ct = 0; ct < 2; ++ct)
                acc[kh][rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[rt], bh[ct], acc[kh][rt][ct], 0, 0, 0);
          }
        }
      }
      // ---- stores: T[n+2] -> ring slot (n+2) % 4 (last read in step n-1), U[n+1] -> the other U buffer
      if (t_req) {
        unsigned char* dst = tring + ((n + 2) & 3) * ROWB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) *(uint4*)(dst + loff[q]) = st_t[q];
        ct2.next(H);
      }
      if (u_req) {
        unsigned char* dst = ubuf + ((n + 1) & 1) * ROWB;
#pragma unroll
        for (int q = 0; q < NQ; ++q) *(uint4*)(dst + loff[q]) = st_u[q];
      }
      cu1.next(H);
      cn.next(H);
      __syncthreads();
    }
  }

  // ---- partial sums: [r][f][kd][kh][co][ci], lane -> ci (128-byte rows)
  const int RT = g.co_tiles * 128, CT = g.ci_tiles * 128;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float* o = g.partial + (((((int64_t)r * 4 + f) * 3 + kd) * 3 + kh) * RT + tco * 128 + wr * 64 + rt * 32) * CT +
                   tci * 128 + wc * 64 + ct * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
          o[(int64_t)row * CT] = acc[kh][rt][ct][e];
        }
      }
}

// dw[co*s_row + ci*s_k + ((kd*3 + kh)*3 + kw)*s_tap] += output transform of sum_r partial[r][f][kd][kh][co][ci]
__global__ void md_wgrad_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int co, int ci, int ksplit,
                                            int64_t s_row, int64_t s_k, int64_t s_tap) {
  const int64_t total = (int64_t)9 * co * ci;
  const int64_t tile = (int64_t)co * ci;           // one (f, kd, kh) slab
  const int64_t slab = 36 * tile;                  // one K range
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(id % ci);
    const int row = (int)((id / ci) % co);
    const int tap = (int)(id / tile);              // kd * 3 + kh
    float n[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float* p = partial + ((int64_t)f * 9 + tap) * tile + (int64_t)row * ci + col;
      float sum = 0.f;
      for (int r = 0; r < ksplit; ++r) sum += p[r * slab];
      n[f] = sum;
    }
    const float h12 = 0.5f * (n[1] + n[2]);
    float* d = dw + row * s_row + col * s_k + (int64_t)(tap * 3) * s_tap;
    d[0] += n[0] + h12;
    d[s_tap] += 0.5f * (n[1] - n[2]);
    d[2 * s_tap] += h12 - n[3];
  }
}

}  // namespace

extern "C" int64_t md_wgrad_wino_workspace_bytes(int32_t co, int32_t ci, int32_t ksplit) {
  if (co <= 0 || ci <= 0 || (co % 128) || (ci % 128) || ksplit <= 0) return MD_ERR_BAD_ARG;
  return (int64_t)ksplit * 36 * co * ci * 4;
}

extern "C" int md_wgrad_wino(const void* u_dy, const void* t_act, float* dw, void* workspace, int64_t workspace_bytes,
                             int32_t batch, int32_t co, int32_t ci, int32_t D, int32_t H, int32_t W, int32_t ksplit,
                             int64_t s_row, int64_t s_k, int64_t s_tap, void* stream) {
  if (!u_dy || !t_act || !dw || !workspace || batch <= 0 || ksplit <= 0) return MD_ERR_BAD_ARG;
  if (co <= 0 || ci <= 0 || (co % 128) || (ci % 128)) return MD_ERR_UNSUPPORTED;
  if (D < 2 || H < 2 || (W != 64 && W != 32)) return MD_ERR_UNSUPPORTED;      // rows of 32 or 16 pairs
  if (workspace_bytes < md_wgrad_wino_workspace_bytes(co, ci, ksplit)) return MD_ERR_BAD_ARG;
  if (ksplit > batch * (D - 1)) return MD_ERR_BAD_ARG;                         // every K range owns at least one plane
  WwArgs g;
  g.U = (const uint4*)u_dy; g.T = (const uint4*)t_act; g.partial = (float*)workspace;
  g.batch = batch; g.co = co; g.ci = ci; g.D = D; g.H = H; g.Wp = W / 2;
  g.co_tiles = co / 128; g.ci_tiles = ci / 128; g.ksplit = ksplit;
  const int64_t items = (int64_t)ksplit * g.co_tiles * g.ci_tiles * 12;
  const int64_t blocks = (items + 255) / 256 * 256;
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  const dim3 grid((unsigned)blocks), blk(WW_THREADS);
  if (W == 64) hipLaunchKernelGGL((md_wgrad_wino_kernel<4>), grid, blk, 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((md_wgrad_wino_kernel<2>), grid, blk, 0, (hipStream_t)stream, g);
  MD_HIP_CHECK_LAUNCH();
  const int64_t total = (int64_t)9 * co * ci;
  int rb = (int)((total + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(md_wgrad_wino_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     dw, co, ci, ksplit, s_row, s_k, s_tap);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
