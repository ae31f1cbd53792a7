// md_wino_prep in two phases through LDS (the host package's default where W divides 256; MD_WINO_PREP_V2=0: one thread per pair).  Same result, bit for bit, as
// md_wino_prep_kernel (conv3_wino.hip: GroupNorm affine + SiLU (+ dropout) of lib/diffusion/models/layers.py:676-682, zero
// pad, F(2,3) input transform along w, bf16 hi / lo split -> T[B][C/8][4][2][D][H][W/2][8]).
//
// md_wino_prep_kernel lets one thread own one output pair: it loads 4 positions (2 of them its neighbours'), so every
// activation is computed twice and a 16-byte load touches 32 different cache lines per wave.  Here a workgroup takes 256
// consecutive positions (whole rows: W divides 256) of one (sample, 8-channel group):
//   phase 1  one POSITION per thread: 32 contiguous bytes (a wave reads 2 KB in a row), activation once, -> LDS (fp32)
//   phase 2  one (pair, frequency half) per thread: waves 0-1 write d0-d2 and d1+d2, waves 2-3 d2-d1 and d1-d3 of the 128
//            pairs, 3 positions each out of LDS; every store instruction writes 1 KB contiguous
#include "md_common.h"

namespace {
constexpr int P2_POS = 256;               // positions per workgroup
constexpr int P2_STRIDE = 12;             // floats per position in LDS (8 + pad: 48 B keeps 16-byte alignment)
// LDS image of a channel group's 256 activated positions, round 5: EVEN and ODD positions in two regions, [parity][128][12 floats],
// the odd region 16 floats further.  Phase 2 reads positions 2 pi - 1 + k (pi = the lane's pair): with the positions in one run a
// lane group of a ds_read_b128 stepped by 24 dwords -- period 8 over the 64 banks, every read 2-way conflicted (37 % of the f6
// kernel's LDS cycles were conflict cycles, profiles/r04_final_pmc_sq.summary.txt).  De-interleaved, the lanes of one read all want
// the same parity at consecutive indices: 12 dwords apart, and 12 l mod 64 puts the 16 lanes of every ds_read_b128 group on 16
// different 16-byte slots.  The phase-1 stores (8-lane groups, 32 banks) alternate between the regions: the 16-float offset keeps
// the even lanes' slots {0, 12, 24, 4} and the odd lanes' {16, 28, 8, 20} apart.  (tests/test_cpu_kernel_layouts.py replays both.)
constexpr int P2_REGION = (P2_POS / 2) * P2_STRIDE + 16;      // floats between the even and the odd region
constexpr int P2_GROUP = 2 * P2_REGION;                        // floats per channel group
#ifdef P2_OLD_IMAGE      // A/B build only (tools/bench_prep.py): the round-4 image, positions in one run
__device__ __forceinline__ int p2_slot(int p) { return p * P2_STRIDE; }
#else
__device__ __forceinline__ int p2_slot(int p) { return (p & 1) * P2_REGION + (p >> 1) * P2_STRIDE; }
#endif
}  // namespace

// DUAL (md_wino_prep_dual): also writes U, the transposed algorithm's transform of the same activated tensor,
//   u = (d1, d1 + d2, d1 - d2, d2) per pair (d1, d2 = the pair's own two positions), in the layout of T -- the dY operand of
//   the Winograd weight gradient (csrc/wgrad_wino.hip) when the tensor is an output gradient.
// F8 (md_wino_prep_f8): T in the "f16f8" operand format of md_conv3_wino_f8 -- same geometry, plane 0 = 8 fp16 (hi), plane 1 =
//   [e4m3(t) x 8 | e4m3((t - hi) 2^11) x 8] (md_split_f16f8) instead of the bf16 hi / lo planes.
// eq (F8 / f6 only, may be null): the static per-input-channel power-of-two equaliser s_c of md_wino_equaliser, [c1 + c2] floats; the
//   activated value is multiplied by it (exact) and the packed weights carry 1 / s_c, so that the channels of a 16-channel K block
//   reach the 4-bit-significand cross-term images with comparable magnitudes whatever the GroupNorm gammas in front are.
template <bool DUAL, bool F8 = false>
__global__ __launch_bounds__(256) void md_wino_prep2_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                            int c1, int c2, const float* __restrict__ ac, int silu, int ups,
                                                            uint4* __restrict__ T, uint4* __restrict__ U, float* __restrict__ sums,
                                                            int batch, int D, int H, int W, uint32_t thr16, float drop_scale,
                                                            uint64_t seed, const float* __restrict__ eq) {
  __shared__ __attribute__((aligned(16))) float act[P2_GROUP];
  __shared__ float wsum[32];                       // DUAL with sums: [wave][channel]
  float psum[8];                                   // this thread's (activated) values, for the channel sums
  const int tid = threadIdx.x;
  const int Wp = W >> 1;
  const int64_t P = (int64_t)D * H * W, Ph = P >> 1;
  const int CG = (c1 + c2) >> 3;
  const int nblk = (int)(P / P2_POS);
  const int blk = blockIdx.x % nblk;
  const int cg = (blockIdx.x / nblk) % CG;
  const int b = blockIdx.x / (nblk * CG);
  int Di = D, Hi = H, Wi = W;
  if (ups) { Di >>= 1; Hi >>= 1; Wi >>= 1; }
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  const float* src = (cg * 8 < c1) ? x1 + ((int64_t)b * (c1 >> 3) + cg) * Pin * 8
                                   : x2 + ((int64_t)b * (c2 >> 3) + (cg - (c1 >> 3))) * Pin * 8;
  // ---- phase 1 ----------------------------------------------------------------------------------------------------
  {
    const int64_t p = (int64_t)blk * P2_POS + tid;                  // position of the OUTPUT grid
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / ((int64_t)W * H));
    const int64_t spos = ups ? ((int64_t)(z >> 1) * Hi + (y >> 1)) * Wi + (x >> 1) : p;
    const f32x4* sp = (const f32x4*)(src + spos * 8);
    const f32x4 v0 = sp[0], v1 = sp[1];
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
    uint64_t bits[2] = {0, 0};
    if (thr16) {
      const uint64_t quad0 = (uint64_t)(((int64_t)b * (c1 + c2) + cg * 8) >> 2) * (uint64_t)Pin;
      bits[0] = md_drop_bits(seed, quad0 + (uint64_t)spos);
      bits[1] = md_drop_bits(seed, quad0 + (uint64_t)Pin + (uint64_t)spos);
    }
    float yv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = e < 4 ? v0[e] : v1[e - 4];
      if (ac != nullptr) {
        t = t * a[e] + c[e];
        if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
      }
      if (thr16) t = md_drop_keep(bits[e >> 2], e & 3, thr16) ? t * drop_scale : 0.f;
      yv[e] = t;
    }
    if constexpr (F8) {
      if (eq != nullptr) {
        const f32x4* ep = (const f32x4*)(eq + cg * 8);
        const f32x4 e0 = ep[0], e1 = ep[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) yv[e] *= e < 4 ? e0[e] : e1[e - 4];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) psum[e] = yv[e];
    float* dst = act + p2_slot(tid);
    *(f32x4*)dst = f32x4{yv[0], yv[1], yv[2], yv[3]};
    *(f32x4*)(dst + 4) = f32x4{yv[4], yv[5], yv[6], yv[7]};
  }
  if constexpr (DUAL) {
    // per-(sample, channel) sums of the tensor (the bias gradient when it is an output gradient: replaces an md_channel_sums
    // pass over the same 4 bytes per element).  Every wave reduces the values its threads hold (DPP inside the 16-lane rows,
    // two cross-row steps), the four partial sums meet after the barrier the two phases need anyway, one float atomic per
    // channel and block.  (First version: wave 0 re-read the staged block and reduced it alone -- the operand pass went from
    // 0.60 to 0.96 ms per 128-channel launch.)
    if (sums != nullptr) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = psum[e];
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if ((tid & 63) == 0) wsum[(tid >> 6) * 8 + e] = v;
      }
    }
  }
  __syncthreads();
  if constexpr (DUAL) {
    if (sums != nullptr && tid < 8)
      atomicAdd(sums + (int64_t)b * (c1 + c2) + cg * 8 + tid, (wsum[tid] + wsum[8 + tid]) + (wsum[16 + tid] + wsum[24 + tid]));
  }
  // ---- phase 2 ----------------------------------------------------------------------------------------------------
  {
    const int pi = tid & 127, fh = tid >> 7;                         // pair of the workgroup, frequency half (wave-uniform)
    const int lp = 2 * pi;                                           // local position of the pair's first output
    const int x = lp % W;                                            // x of that position (rows are whole)
    // d_k = activated input at x - 1 + k; outside the row: zero (the conv pads the ACTIVATED tensor)
    float d[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool need = fh == 0 ? k < 3 : k > 0;                     // half 0 uses d0 d1 d2, half 1 uses d1 d2 d3
      const int xx = x - 1 + k;
      const bool live = need && xx >= 0 && xx < W;
      f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0;
      if (live) {
        const float* s = act + p2_slot(lp - 1 + k);
        u0 = *(const f32x4*)s; u1 = *(const f32x4*)(s + 4);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) d[k][e] = e < 4 ? u0[e] : u1[e - 4];
    }
    const int64_t pos2 = ((int64_t)blk * P2_POS >> 1) + pi;
    uint4* out = T + ((int64_t)b * CG + cg) * 8 * Ph + pos2;          // [f][plane][Ph] items of 16 B
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int f = 2 * fh + g;
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        t[e] = f == 0 ? d[0][e] - d[2][e] : f == 1 ? d[1][e] + d[2][e] : f == 2 ? d[2][e] - d[1][e] : d[1][e] - d[3][e];
      if constexpr (F8) {
        uint4 hv;
        uint32_t q8[2], ql8[2];
        md_split_f16f8(t, hv, q8, ql8);
        out[(int64_t)(f * 2) * Ph] = hv;
        out[(int64_t)(f * 2 + 1) * Ph] = make_uint4(q8[0], q8[1], ql8[0], ql8[1]);
      } else {
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) md_split2(t[2 * q], t[2 * q + 1], hw[q], lw[q]);
        out[(int64_t)(f * 2) * Ph] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        out[(int64_t)(f * 2 + 1) * Ph] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
      }
    }
    if constexpr (DUAL) {
      uint4* uo = U + ((int64_t)b * CG + cg) * 8 * Ph + pos2;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int f = 2 * fh + g;
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e)
          t[e] = f == 0 ? d[1][e] : f == 1 ? d[1][e] + d[2][e] : f == 2 ? d[1][e] - d[2][e] : d[2][e];
        uint32_t hw[4], lw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) md_split2(t[2 * q], t[2 * q + 1], hw[q], lw[q]);
        uo[(int64_t)(f * 2) * Ph] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
        uo[(int64_t)(f * 2 + 1) * Ph] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
      }
    }
  }
}

// md_wino_prep_f6: T in the "f16f6" operand format of md_conv3_wino_f6 -- the geometry of the f16f8 operand, plane 0 = 8 fp16 (hi) per
// channel group, plane 1 = the 32-byte MX record of a 16-channel K block ([e2m3 codes of (t, (t - hi) 2^11) x 16 | E8M0 scale | 0],
// md_split_f16f6) split over the block's two channel groups.  The block scale couples two channel groups, so a workgroup takes
// BOTH of them (16 channels x 256 positions): phase 1 activates every position once per group (2 x 32 bytes per thread), phase 2
// owns a (pair, frequency half) with all 16 channels.
// DUAL (md_wino_prep_dual_f6, round 6: the data-gradient convs of a training step in f16f6): the tensor is an output gradient;
//   T = the f16f6 operand of md_conv3_wino_f6 of `tscale` x the tensor (a power of two that lifts gradient magnitudes into the
//   fp16 plane's normal range; the conv's launch divides it out again), U = the bf16 hi / lo operand of md_wgrad_wino of the
//   UNSCALED tensor (bit-identical to md_wino_prep_dual's), sums = its per-(sample, channel) sums (bias gradients).
template <bool DUAL>
__global__ __launch_bounds__(256) void md_wino_prep2_f6_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1, int c2,
                                                               const float* __restrict__ ac, int silu, int ups, uint4* __restrict__ T, int batch,
                                                               int D, int H, int W, const float* __restrict__ eq, uint4* __restrict__ U,
                                                               float* __restrict__ sums, float tscale, const uint32_t* __restrict__ amax_bits) {
  __shared__ __attribute__((aligned(16))) float act[2 * P2_GROUP];
  __shared__ float wsum[4][16];                    // DUAL with sums: [wave][channel]
  if constexpr (DUAL) {
    if (amax_bits != nullptr) tscale = ldexpf(1.f, md_dgrad_lift_log2(amax_bits[0]));      // the dynamic lift (md_absmax of this tensor)
  }
  const int tid = threadIdx.x;
  const int Wp = W >> 1;
  const int64_t P = (int64_t)D * H * W, Ph = P >> 1;
  const int CG = (c1 + c2) >> 3, CGP = CG >> 1;
  const int nblk = (int)(P / P2_POS);
  const int blk = blockIdx.x % nblk;
  const int cgp = (blockIdx.x / nblk) % CGP;
  const int b = blockIdx.x / (nblk * CGP);
  int Di = D, Hi = H, Wi = W;
  if (ups) { Di >>= 1; Hi >>= 1; Wi >>= 1; }
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  // ---- phase 1 ----
  {
    const int64_t p = (int64_t)blk * P2_POS + tid;
    const int x = (int)(p % W), y = (int)((p / W) % H), z = (int)(p / ((int64_t)W * H));
    const int64_t spos = ups ? ((int64_t)(z >> 1) * Hi + (y >> 1)) * Wi + (x >> 1) : p;
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2) {
      const int cg = 2 * cgp + g2;
      const float* src = (cg * 8 < c1) ? x1 + ((int64_t)b * (c1 >> 3) + cg) * Pin * 8
                                       : x2 + ((int64_t)b * (c2 >> 3) + (cg - (c1 >> 3))) * Pin * 8;
      const f32x4* sp = (const f32x4*)(src + spos * 8);
      const f32x4 v0 = sp[0], v1 = sp[1];
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
      float yv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = e < 4 ? v0[e] : v1[e - 4];
        if (ac != nullptr) {
          t = t * a[e] + c[e];
          if (silu) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t * -1.4426950408889634f));
        }
        yv[e] = t;
      }
      if (eq != nullptr) {
        const f32x4* ep = (const f32x4*)(eq + cg * 8);
        const f32x4 e0 = ep[0], e1 = ep[1];
#pragma unroll
        for (int e = 0; e < 8; ++e) yv[e] *= e < 4 ? e0[e] : e1[e - 4];
      }
      float* dst = act + g2 * P2_GROUP + p2_slot(tid);
      *(f32x4*)dst = f32x4{yv[0], yv[1], yv[2], yv[3]};
      *(f32x4*)(dst + 4) = f32x4{yv[4], yv[5], yv[6], yv[7]};
      if constexpr (DUAL) {
        if (sums != nullptr) {      // per-wave channel sums as in md_wino_prep2_kernel<true>
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = md_wave_sum(yv[e]);
            if ((tid & 63) == 0) wsum[tid >> 6][g2 * 8 + e] = v;
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (DUAL) {
    if (sums != nullptr && tid < 16)
      atomicAdd(sums + (int64_t)b * (c1 + c2) + cgp * 16 + tid, (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]));
  }
  // ---- phase 2 ----
  {
    const int pi = tid & 127, fh = tid >> 7;
    const int lp = 2 * pi;
    const int x = lp % W;
    float d[4][16];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool need = fh == 0 ? k < 3 : k > 0;
      const int xx = x - 1 + k;
      const bool live = need && xx >= 0 && xx < W;
#pragma unroll
      for (int g2 = 0; g2 < 2; ++g2) {
        f32x4 u0 = {0.f, 0.f, 0.f, 0.f}, u1 = u0;
        if (live) {
          const float* sv = act + g2 * P2_GROUP + p2_slot(lp - 1 + k);
          u0 = *(const f32x4*)sv; u1 = *(const f32x4*)(sv + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) d[k][g2 * 8 + e] = e < 4 ? u0[e] : u1[e - 4];
      }
    }
    const int64_t pos2 = ((int64_t)blk * P2_POS >> 1) + pi;
    uint4* out0 = T + ((int64_t)b * CG + 2 * cgp) * 8 * Ph + pos2;        // channel group 2 cgp: [f][plane][Ph] items of 16 B
    uint4* out1 = out0 + 8 * Ph;                                          // channel group 2 cgp + 1
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const int f = 2 * fh + g;
      float t[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        t[e] = f == 0 ? d[0][e] - d[2][e] : f == 1 ? d[1][e] + d[2][e] : f == 2 ? d[2][e] - d[1][e] : d[1][e] - d[3][e];
        // the lift (dynamic: max |dy| lands in [16, 32); or the caller's constant), saturated at the fp16 plane's range so that a
        // constant lift can clip an element but never turn the whole data gradient into inf / NaN
        if constexpr (DUAL) t[e] = fminf(fmaxf(t[e] * tscale, -60000.f), 60000.f);
      }
      uint4 h0, h1, r0, r1;
      md_split_f16f6(t, false, 0, h0, h1, r0, r1);
      out0[(int64_t)(f * 2) * Ph] = h0;
      out1[(int64_t)(f * 2) * Ph] = h1;
      out0[(int64_t)(f * 2 + 1) * Ph] = r0;
      out1[(int64_t)(f * 2 + 1) * Ph] = r1;
    }
    if constexpr (DUAL) {
      uint4* uo0 = U + ((int64_t)b * CG + 2 * cgp) * 8 * Ph + pos2;
      uint4* uo1 = uo0 + 8 * Ph;
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int f = 2 * fh + g;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          uint32_t hw[4], lw[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int e0 = g2 * 8 + 2 * q;
            const float ta = f == 0 ? d[1][e0] : f == 1 ? d[1][e0] + d[2][e0] : f == 2 ? d[1][e0] - d[2][e0] : d[2][e0];
            const float tb = f == 0 ? d[1][e0 + 1] : f == 1 ? d[1][e0 + 1] + d[2][e0 + 1] : f == 2 ? d[1][e0 + 1] - d[2][e0 + 1] : d[2][e0 + 1];
            md_split2(ta, tb, hw[q], lw[q]);
          }
          uint4* uo = g2 ? uo1 : uo0;
          uo[(int64_t)(f * 2) * Ph] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
          uo[(int64_t)(f * 2 + 1) * Ph] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
        }
      }
    }
  }
}

static int md_wino_prep2_launch(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                                int32_t ups, void* t_out, void* u_out, float* sums, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p,
                                uint64_t drop_seed, void* stream, bool f8 = false, const float* eq = nullptr) {
  if (!x1 || !t_out || batch <= 0 || c1 <= 0 || c2 < 0 || (c1 & 7) || (c2 & 7) || (c2 > 0 && !x2)) return MD_ERR_BAD_ARG;
  if (silu && !ac) return MD_ERR_BAD_ARG;      // SiLU is applied together with the folded GroupNorm affine only
  if (D <= 0 || H <= 0 || W <= 0 || (W & 1) || (ups && ((D | H | W) & 1))) return MD_ERR_BAD_ARG;
  if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && (ups || c2 > 0))) return MD_ERR_BAD_ARG;
  const int64_t P = (int64_t)D * H * W;
  if ((P2_POS % W) || (P % P2_POS)) return MD_ERR_UNSUPPORTED;      // whole rows per workgroup
  const int64_t blocks = (int64_t)batch * ((c1 + c2) / 8) * (P / P2_POS);
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  if (f8)
    hipLaunchKernelGGL((md_wino_prep2_kernel<false, true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac,
                       silu, ups, (uint4*)t_out, (uint4*)nullptr, (float*)nullptr, batch, D, H, W, md_drop_thr16(drop_p),
                       1.0f / (1.0f - drop_p), drop_seed, eq);
  else if (u_out)
    hipLaunchKernelGGL((md_wino_prep2_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac,
                       silu, ups, (uint4*)t_out, (uint4*)u_out, sums, batch, D, H, W, md_drop_thr16(drop_p), 1.0f / (1.0f - drop_p),
                       drop_seed, (const float*)nullptr);
  else
    hipLaunchKernelGGL((md_wino_prep2_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac,
                       silu, ups, (uint4*)t_out, (uint4*)nullptr, (float*)nullptr, batch, D, H, W, md_drop_thr16(drop_p),
                       1.0f / (1.0f - drop_p), drop_seed, (const float*)nullptr);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_wino_prep_v2(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                               int32_t ups, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p,
                               uint64_t drop_seed, void* stream) {
  return md_wino_prep2_launch(x1, x2, c1, c2, ac, silu, ups, t_out, nullptr, nullptr, batch, D, H, W, drop_p, drop_seed, stream);
}

extern "C" int md_wino_prep_dual(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                                 int32_t ups, void* t_out, void* u_out, float* sums, int32_t batch, int32_t D, int32_t H,
                                 int32_t W, float drop_p, uint64_t drop_seed, void* stream) {
  if (!u_out || (sums && ups)) return MD_ERR_BAD_ARG;
  return md_wino_prep2_launch(x1, x2, c1, c2, ac, silu, ups, t_out, u_out, sums, batch, D, H, W, drop_p, drop_seed, stream);
}


extern "C" int md_wino_prep_f8(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                               int32_t ups, const float* eq, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream) {
  return md_wino_prep2_launch(x1, x2, c1, c2, ac, silu, ups, t_out, nullptr, nullptr, batch, D, H, W, 0.f, 0, stream, true, eq);
}

extern "C" int md_wino_prep_f6(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                               int32_t ups, const float* eq, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!x1 || !t_out || batch <= 0 || c1 <= 0 || c2 < 0 || (c1 & 15) || (c2 & 15) || (c2 > 0 && !x2)) return MD_ERR_BAD_ARG;   // whole 16-channel blocks per part
  if (silu && !ac) return MD_ERR_BAD_ARG;
  if (D <= 0 || H <= 0 || W <= 0 || (W & 1) || (ups && ((D | H | W) & 1))) return MD_ERR_BAD_ARG;
  const int64_t P = (int64_t)D * H * W;
  if ((P2_POS % W) || (P % P2_POS)) return MD_ERR_UNSUPPORTED;
  const int64_t blocks = (int64_t)batch * ((c1 + c2) / 16) * (P / P2_POS);
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_prep2_f6_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac, silu, ups,
                     (uint4*)t_out, batch, D, H, W, eq, (uint4*)nullptr, (float*)nullptr, 1.0f, (const uint32_t*)nullptr);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_wino_prep_dual_f6(const float* x, int32_t c, void* t_out, void* u_out, float* sums, float tscale, const uint32_t* amax_bits,
                                    int32_t batch, int32_t D, int32_t H, int32_t W, void* stream) {
  if (!x || !t_out || !u_out || batch <= 0 || c <= 0 || (c & 15)) return MD_ERR_BAD_ARG;
  if (D <= 0 || H <= 0 || W <= 0 || (W & 1) || (!amax_bits && !(tscale > 0.f))) return MD_ERR_BAD_ARG;
  const int64_t P = (int64_t)D * H * W;
  if ((P2_POS % W) || (P % P2_POS)) return MD_ERR_UNSUPPORTED;
  const int64_t blocks = (int64_t)batch * (c / 16) * (P / P2_POS);
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_prep2_f6_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, (const float*)nullptr, c, 0,
                     (const float*)nullptr, 0, 0, (uint4*)t_out, batch, D, H, W, (const float*)nullptr, (uint4*)u_out, sums, tscale, amax_bits);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
