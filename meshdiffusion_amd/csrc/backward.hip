// Backward-pass support kernels ("first correct version"): everything that is a contraction
// re-uses md_gemm_conv (dgrad = conv with flipped/transposed WPK tiles; wgrad = split-K GEMM over positions), so
// this file only holds the streaming pieces around it.
//
// PB16 layout (wgrad operands): bf16 [G + Pp + G][ceil(B/8)][2][C][8 samples], positions on the zero-padded grid
// (D+2p)(H+2p)(W+2p) (p = 1 for 3x3x3 / 1x1x1 layers, 2 for 5x5x5) plus G guard positions of zeros on both sides.  The contraction index of a weight gradient is
// (position, sample); blocking it by 8 SAMPLES (not 8 positions) means any spatial tap shift keeps the 8-blocks
// intact, so dW[tap] = sum_k dY[k] * A[k + off(tap)] is the plain GEMM with the B pointer moved by off(tap).
//
// Reference semantics: torch autograd of lib/diffusion/models/layers.py:646-689 (ResnetBlockDDPM), :573-582 (NIN),
// :611-643 (Up/Downsample), nn.GroupNorm + nn.SiLU.
#include "md_common.h"

// ---- F32B / S16B -> PB16 -----------------------------------------------------------------------------------
// mode 0: src = F32B fp32 [B][C/8][P][8] (split here); mode 1: src = S16B [B][C/8][2][P][8] (planes copied)
// up: source grid is (D/2,H/2,W/2) and is nearest-upsampled; stuff: source grid is (D/2..) placed at odd fine
// positions (2o+1), zeros elsewhere (dgrad/wgrad of the stride-2 Downsample conv).
// One thread = one (padded position, sample block, group of 8 channels): consecutive lanes take consecutive positions,
// so each of the 8 per-sample loads of a wave covers 64 x 32 contiguous bytes of the source, and every lane writes the
// two full 128-byte lines [8 channels][8 samples] (hi / lo plane) of its position.
//
// zsplit > 1 (small batches): every sample is cut into `zsplit` z-slabs of D planes and each slab takes one of the 8
// slots of a sample block ("virtual samples" v = b * zsplit + slab), so a batch of 1 fills the MFMA k-group with 8
// slabs instead of 1 sample + 7 zeros.  The weight gradient is a sum over positions, so summing over slabs is exact
// provided the ACTIVATION operand's z-halo planes hold the neighbouring slab's data (zhalo = 1) while the dY operand's
// stay zero (zhalo = 0: each output position belongs to exactly one slab).  D is the slab depth, Dfull = D * zsplit.
__global__ __launch_bounds__(256) void md_to_pb16_kernel(const void* __restrict__ src, uint16_t* __restrict__ out, int B, int C,
                                                         int Cs, int D, int H, int W, int guard, int mode, int up, int stuff,
                                                         int pad, int zsplit, int zhalo) {
  const int Dp = D + 2 * pad, Hp = H + 2 * pad, Wp = W + 2 * pad;
  const int64_t Pp = (int64_t)Dp * Hp * Wp;
  const int VB = B * zsplit;
  const int bg_n = (VB + 7) / 8;   // a partial last block of 8 (virtual) samples is zero filled
  const int cg_n = C / 8;
  const int64_t total = Pp * bg_n * cg_n;
  const int Dfull = D * zsplit;
  const int Ds = (up || stuff) ? Dfull / 2 : Dfull, Hs = (up || stuff) ? H / 2 : H, Ws = (up || stuff) ? W / 2 : W;
  const int64_t Ps = (int64_t)Ds * Hs * Ws;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pp = i % Pp;
    const int cg = (int)((i / Pp) % cg_n);
    const int bg = (int)(i / (Pp * cg_n));
    const int px = (int)(pp % Wp) - pad, py = (int)((pp / Wp) % Hp) - pad, pz = (int)(pp / ((int64_t)Wp * Hp)) - pad;
    uint32_t hi[8][8], lo[8][8];   // [sample][channel]
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) { hi[k][e] = 0; lo[k][e] = 0; }
    const bool inxy = (px >= 0) & (px < W) & (py >= 0) & (py < H) & (cg * 8 < Cs) & (zhalo || ((pz >= 0) & (pz < D)));
    if (inxy) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int v = bg * 8 + k;
        if (v >= VB) break;
        const int b = v / zsplit, gz = (v - b * zsplit) * D + pz;   // plane of the whole sample
        bool inb = (gz >= 0) & (gz < Dfull);
        int sx = px, sy = py, sz = gz;
        if (up) { sx >>= 1; sy >>= 1; sz >>= 1; }
        if (stuff) { inb = inb & (px & 1) & (py & 1) & (gz & 1); sx >>= 1; sy >>= 1; sz >>= 1; }
        if (!inb) continue;
        const int64_t sp = ((int64_t)sz * Hs + sy) * Ws + sx;
        if (mode == 0) {
          const f32x4* s4 = (const f32x4*)((const float*)src + (((int64_t)b * (Cs / 8) + cg) * Ps + sp) * 8);
          const f32x4 v0 = s4[0], v1 = s4[1];
#pragma unroll
          for (int e = 0; e < 4; ++e) { md_split(v0[e], hi[k][e], lo[k][e]); md_split(v1[e], hi[k][4 + e], lo[k][4 + e]); }
        } else {
          const uint16_t* s16 = (const uint16_t*)src + ((((int64_t)b * (Cs / 8) + cg) * 2) * Ps + sp) * 8;
          const uint4 h4 = *(const uint4*)s16, l4 = *(const uint4*)(s16 + Ps * 8);
          const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w}, lw[4] = {l4.x, l4.y, l4.z, l4.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            hi[k][e] = (hw[e >> 1] >> (16 * (e & 1))) & 0xffffu;
            lo[k][e] = (lw[e >> 1] >> (16 * (e & 1))) & 0xffffu;
          }
        }
      }
    }
    uint4* o = (uint4*)(out + ((((int64_t)(guard + pp) * bg_n + bg) * 2) * C + cg * 8) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      o[e] = make_uint4(hi[0][e] | (hi[1][e] << 16), hi[2][e] | (hi[3][e] << 16), hi[4][e] | (hi[5][e] << 16), hi[6][e] | (hi[7][e] << 16));
      o[C + e] = make_uint4(lo[0][e] | (lo[1][e] << 16), lo[2][e] | (lo[3][e] << 16), lo[4][e] | (lo[5][e] << 16), lo[6][e] | (lo[7][e] << 16));
    }
  }
}

extern "C" int64_t md_pb16_bytes(int32_t batch, int32_t C, int32_t D, int32_t H, int32_t W, int32_t guard, int32_t pad) {
  if (batch <= 0 || C <= 0 || D <= 0 || H <= 0 || W <= 0 || guard < 0 || pad < 1 || pad > 2) return MD_ERR_BAD_ARG;
  return ((int64_t)(D + 2 * pad) * (H + 2 * pad) * (W + 2 * pad) + 2 * (int64_t)guard) * ((batch + 7) / 8) * 2 * C * 8 * 2;
}

extern "C" int md_to_pb16(const void* src, void* out, int32_t batch, int32_t C, int32_t c_src, int32_t D, int32_t H,
                          int32_t W, int32_t guard, int32_t pad, int32_t mode, int32_t up, int32_t stuff, int32_t zsplit,
                          int32_t zhalo, void* stream) {
  if (zsplit < 1 || zsplit > 8 || (zsplit & (zsplit - 1))) return MD_ERR_BAD_ARG;
  if (!src || !out || md_pb16_bytes(batch * zsplit, C, D, H, W, guard, pad) < 0 || (C % 8) || (c_src % 8) || c_src <= 0 || c_src > C ||
      mode < 0 || mode > 1)
    return MD_ERR_BAD_ARG;
  if ((up || stuff) && (((D * zsplit) | H | W) & 1)) return MD_ERR_BAD_ARG;
  // the kernel writes every position of the padded grid (zeros on the halo); only the two guards need clearing
  const int64_t pos_bytes = (int64_t)((batch * zsplit + 7) / 8) * 2 * C * 8 * 2;
  const int64_t Pp = (int64_t)(D + 2 * pad) * (H + 2 * pad) * (W + 2 * pad);
  if (guard > 0) {
    hipError_t e = hipMemsetAsync(out, 0, (size_t)(guard * pos_bytes), (hipStream_t)stream);
    if (e == hipSuccess)
      e = hipMemsetAsync((char*)out + (guard + Pp) * pos_bytes, 0, (size_t)(guard * pos_bytes), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  const int64_t total = Pp * ((batch * zsplit + 7) / 8) * (C / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 65536) blocks = 65536;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_to_pb16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (uint16_t*)out,
                     batch, C, c_src, D, H, W, guard, mode, up, stuff, pad, zsplit, zhalo);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- wgrad finish: GEMM result [ntap][rows/8][cols][8] (F32B per tap) -> weight gradient, accumulated ---------
//   dw[(row*s_row + col*s_k + tap*s_tap)] += g   (same stride convention as md_pack_weights)
__global__ void md_wgrad_finish_kernel(const float* __restrict__ g, float* __restrict__ dw, int rows, int cols,
                                       int cols_alloc, int ntap, int tap0, int64_t s_row, int64_t s_k, int64_t s_tap) {
  const int64_t total = (int64_t)ntap * rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % cols);
    const int row = (int)((i / cols) % rows);
    const int t = (int)(i / ((int64_t)cols * rows));
    const int rows8 = (rows + 7) / 8;
    const float v = g[(((int64_t)t * rows8 + (row >> 3)) * cols_alloc + col) * 8 + (row & 7)];
    dw[row * s_row + col * s_k + (tap0 + t) * s_tap] += v;
  }
}

extern "C" int md_wgrad_finish(const float* g, float* dw, int32_t rows, int32_t cols, int32_t cols_alloc, int32_t ntap,
                               int32_t tap0, int64_t s_row, int64_t s_k, int64_t s_tap, void* stream) {
  if (!g || !dw || rows <= 0 || cols <= 0 || cols_alloc < cols || ntap <= 0) return MD_ERR_BAD_ARG;
  const int64_t total = (int64_t)ntap * rows * cols;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wgrad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, dw, rows, cols,
                     cols_alloc, ntap, tap0, s_row, s_k, s_tap);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- GroupNorm (+SiLU) backward -----------------------------------------------------------------------------------
// forward: z = (x - mu) * a + beta, a = rstd*gamma; y = silu ? z*sigmoid(z) : z.   params float4 = (mu, a, beta, rstd).
// pass 1 (stats):    S1[b,c] = sum_p dz, S2[b,c] = sum_p dz * xhat      (dz = dy * silu'(z), xhat = (x-mu)*rstd)
// pass 2 (finalize): per (b, group): G1 = sum_c gamma_c S1, G2 = sum_c gamma_c S2;
//                    coef[b,c] = (rstd*gamma_c, rstd*G1/n, rstd*G2/n, 0); dgamma[c] += S2[b,c]; dbeta[c] += S1[b,c]
// pass 3 (apply):    dx = k1*dz - k2 - xhat*k3   (+= into dx when accumulate)
static constexpr int GB_BLOCK = 256, GB_ITEMS = 16, GB_CHUNK = GB_BLOCK * GB_ITEMS / 2;

__device__ __forceinline__ float md_silu_grad(float z) {
  const float s = 1.0f / (1.0f + expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

__global__ __launch_bounds__(GB_BLOCK) void md_gn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ params, double* __restrict__ sums,
                                                                   int C, int64_t P, int c_total, int c_off, int dy_ctotal,
                                                                   int silu, uint32_t thr16, float drop_scale, uint64_t seed) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, half = tid & 1;
  const int64_t p0 = (int64_t)blockIdx.x * GB_CHUNK;
  const f32x4* xp = (const f32x4*)(x + (((int64_t)b * (C / 8) + cg) * P) * 8);
  const f32x4* dp = (const f32x4*)(dy + (((int64_t)b * (dy_ctotal / 8) + (c_off / 8) + cg) * P) * 8);
  float mu[4], a[4], bt[4], r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const f32x4 pr = *(const f32x4*)(params + ((int64_t)b * c_total + c_off + cg * 8 + half * 4 + e) * 4);
    mu[e] = pr[0]; a[e] = pr[1]; bt[e] = pr[2]; r[e] = pr[3];
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i) {
    const int64_t pos = p0 + ((tid + i * GB_BLOCK) >> 1);
    if (pos < P) {
      const f32x4 xv = xp[pos * 2 + half];
      f32x4 dv = dp[pos * 2 + half];
      if (thr16) {  // the forward dropped / rescaled y after SiLU: the same mask gates dy
        const uint64_t bits = md_drop_bits(seed, (uint64_t)(((int64_t)b * c_total + c_off + cg * 8 + half * 4) >> 2) * (uint64_t)P + (uint64_t)pos);
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[e] = md_drop_keep(bits, e, thr16) ? dv[e] * drop_scale : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xc = xv[e] - mu[e];
        const float z = xc * a[e] + bt[e];
        const float dz = silu ? dv[e] * md_silu_grad(z) : dv[e];
        s1[e] += dz;
        s2[e] += dz * (xc * r[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
  }
  __shared__ float red[GB_BLOCK / 64][2][8];
  const int lane = tid & 63, wid = tid >> 6;
  if (lane < 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wid][lane][e] = s1[e]; red[wid][lane][4 + e] = s2[e]; }
  }
  __syncthreads();
  if (tid < 16) {
    const int hh = (tid >> 2) & 1, e = tid & 3, which = tid >> 3;
    double acc = 0.0;
    for (int w = 0; w < GB_BLOCK / 64; ++w) acc += (double)red[w][hh][which * 4 + e];
    atomicAdd(&sums[((int64_t)b * c_total + c_off + cg * 8 + hh * 4 + e) * 2 + which], acc);
  }
}

__global__ void md_gn_bwd_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ params,
                                          const float* __restrict__ gamma, float* __restrict__ coef,
                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int c_total, int groups,
                                          int64_t P) {
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = c_total / groups;
  const int lane = threadIdx.x;
  double g1 = 0.0, g2 = 0.0;
  for (int c = lane; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    const int64_t o = ((int64_t)b * c_total + ch) * 2;
    g1 += (double)gamma[ch] * sums[o];
    g2 += (double)gamma[ch] * sums[o + 1];
    if (dbeta) atomicAdd(&dbeta[ch], (float)sums[o]);
    if (dgamma) atomicAdd(&dgamma[ch], (float)sums[o + 1]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { g1 += __shfl_xor(g1, o, 64); g2 += __shfl_xor(g2, o, 64); }
  const double n = (double)cpg * (double)P;
  for (int c = lane; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    const float r = params[((int64_t)b * c_total + ch) * 4 + 3];
    f32x4 o4 = {r * gamma[ch], (float)((double)r * g1 / n), (float)((double)r * g2 / n), 0.f};
    *(f32x4*)(coef + ((int64_t)b * c_total + ch) * 4) = o4;
  }
}

__global__ __launch_bounds__(GB_BLOCK) void md_gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   const float* __restrict__ params, const float* __restrict__ coef,
                                                                   float* __restrict__ dx, int C, int64_t P, int c_total,
                                                                   int c_off, int dy_ctotal, int silu, int accumulate,
                                                                   uint32_t thr16, float drop_scale, uint64_t seed,
                                                                   const float* __restrict__ residual, float* __restrict__ ch_sums,
                                                                   uint32_t* __restrict__ amax_bits) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, half = tid & 1;
  const int64_t p0 = (int64_t)blockIdx.x * GB_CHUNK;
  float amax = 0.f;                                   // max |dx| of what this thread writes (amax_bits: the f16f6 data-gradient conv's lift)
  const int64_t xo = (((int64_t)b * (C / 8) + cg) * P) * 8;
  const f32x4* xp = (const f32x4*)(x + xo);
  f32x4* op = (f32x4*)(dx + xo);
  const f32x4* rp = residual ? (const f32x4*)(residual + xo) : nullptr;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const f32x4* dp = (const f32x4*)(dy + (((int64_t)b * (dy_ctotal / 8) + (c_off / 8) + cg) * P) * 8);
  float mu[4], a[4], bt[4], r[4], k1[4], k2[4], k3[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t ci = ((int64_t)b * c_total + c_off + cg * 8 + half * 4 + e) * 4;
    const f32x4 pr = *(const f32x4*)(params + ci), cf = *(const f32x4*)(coef + ci);
    mu[e] = pr[0]; a[e] = pr[1]; bt[e] = pr[2]; r[e] = pr[3]; k1[e] = cf[0]; k2[e] = cf[1]; k3[e] = cf[2];
  }
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i) {
    const int64_t pos = p0 + ((tid + i * GB_BLOCK) >> 1);
    if (pos < P) {
      const f32x4 xv = xp[pos * 2 + half];
      f32x4 dv = dp[pos * 2 + half];
      if (thr16) {
        const uint64_t bits = md_drop_bits(seed, (uint64_t)(((int64_t)b * c_total + c_off + cg * 8 + half * 4) >> 2) * (uint64_t)P + (uint64_t)pos);
#pragma unroll
        for (int e = 0; e < 4; ++e) dv[e] = md_drop_keep(bits, e, thr16) ? dv[e] * drop_scale : 0.f;
      }
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if (accumulate) o = op[pos * 2 + half];
      else if (rp) o = rp[pos * 2 + half];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xc = xv[e] - mu[e];
        const float z = xc * a[e] + bt[e];
        const float dz = silu ? dv[e] * md_silu_grad(z) : dv[e];
        const float g = k1[e] * dz - k2[e] - (xc * r[e]) * k3[e];
        o[e] += g;
        cs[e] += g;
      }
      op[pos * 2 + half] = o;
      amax = fmaxf(amax, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
  }
  if (amax_bits) {   // as md_absmax would find it in a separate pass over dx
    // one atomic per wave at most, and none once the word already holds a larger value (a plain, possibly stale read: a stale
    // smaller value only costs an atomic) -- 65 k same-address atomics per launch otherwise serialise in the L2
    amax = md_wave_max(amax);
    if ((tid & 63) == 0 && __float_as_uint(amax) > __builtin_nontemporal_load(amax_bits)) atomicMax(amax_bits, __float_as_uint(amax));
  }
  if (ch_sums) {   // per-(sample, channel) sums of the GroupNorm input gradient (bias / FiLM gradients of the producer)
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int o = 32; o > 1; o >>= 1) cs[e] += __shfl_xor(cs[e], o, 64);
    if ((tid & 63) < 2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) atomicAdd(&ch_sums[(int64_t)b * c_total + c_off + cg * 8 + half * 4 + e], cs[e]);
    }
  }
}

extern "C" int md_gn_bwd_stats(const float* x, const float* dy, const float* params, double* sums, int32_t batch, int32_t C,
                               int64_t P, int32_t c_total, int32_t c_off, int32_t dy_ctotal, int32_t silu, float drop_p,
                               uint64_t drop_seed, void* stream) {
  if (!(drop_p >= 0.f && drop_p < 1.f)) return MD_ERR_BAD_ARG;
  if (!x || !dy || !params || !sums || batch <= 0 || C <= 0 || (C % 8) || (c_off % 8) || c_off + C > c_total) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + GB_CHUNK - 1) / GB_CHUNK), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_bwd_stats_kernel, grid, dim3(GB_BLOCK), 0, (hipStream_t)stream, x, dy, params, sums, C, P, c_total,
                     c_off, dy_ctotal, silu, md_drop_thr16(drop_p), 1.0f / (1.0f - drop_p), (uint64_t)drop_seed);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_gn_bwd_finalize(const double* sums, const float* params, const float* gamma, float* coef, float* dgamma,
                                  float* dbeta, int32_t batch, int32_t c_total, int32_t groups, int64_t P, void* stream) {
  if (!sums || !params || !gamma || !coef || batch <= 0 || groups <= 0 || (c_total % groups)) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_bwd_finalize_kernel, dim3((unsigned)(batch * groups)), dim3(64), 0, (hipStream_t)stream, sums, params,
                     gamma, coef, dgamma, dbeta, c_total, groups, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_gn_bwd_apply(const float* x, const float* dy, const float* params, const float* coef, float* dx, int32_t batch,
                               int32_t C, int64_t P, int32_t c_total, int32_t c_off, int32_t dy_ctotal, int32_t silu,
                               int32_t accumulate, float drop_p, uint64_t drop_seed, const float* residual, float* ch_sums,
                               uint32_t* amax_bits, void* stream) {
  if (!(drop_p >= 0.f && drop_p < 1.f)) return MD_ERR_BAD_ARG;
  if (!x || !dy || !params || !coef || !dx || batch <= 0 || C <= 0 || (C % 8) || (c_off % 8) || c_off + C > c_total) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + GB_CHUNK - 1) / GB_CHUNK), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_bwd_apply_kernel, grid, dim3(GB_BLOCK), 0, (hipStream_t)stream, x, dy, params, coef, dx, C, P, c_total,
                     c_off, dy_ctotal, silu, accumulate, md_drop_thr16(drop_p), 1.0f / (1.0f - drop_p), (uint64_t)drop_seed, residual,
                     ch_sums, amax_bits);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- per-(b, channel) sums of an F32B tensor (bias / FiLM gradients): out[b][c] (+)= sum_p x ----------------------
__global__ __launch_bounds__(256) void md_channel_sums_kernel(const float* __restrict__ x, float* __restrict__ out, int C, int64_t P) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, half = tid & 1;
  const int64_t p0 = (int64_t)blockIdx.x * GB_CHUNK;
  const f32x4* xp = (const f32x4*)(x + (((int64_t)b * (C / 8) + cg) * P) * 8);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GB_ITEMS; ++i) {
    const int64_t pos = p0 + ((tid + i * 256) >> 1);
    if (pos < P) { const f32x4 v = xp[pos * 2 + half]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) s[e] += __shfl_xor(s[e], o, 64);
  if ((tid & 63) < 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&out[(int64_t)b * C + cg * 8 + (tid & 1) * 4 + e], s[e]);
  }
}

extern "C" int md_channel_sums(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream) {
  if (!x || !out || batch <= 0 || C <= 0 || (C % 8) || P <= 0) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + GB_CHUNK - 1) / GB_CHUNK), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_channel_sums_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- F32B resampling of gradients: mode 0: out[coarse] (+)= sum of the 8 fine children (Upsample backward);
//      mode 1: out[fine 2o+1] = in[coarse o], zeros elsewhere (zero-stuffing for the stride-2 conv's dgrad) -------------
__global__ void md_grad_resample_kernel(const float* __restrict__ in, float* __restrict__ out, int CG, int Dc, int Hc, int Wc,
                                        int mode, int accumulate) {
  const int64_t Pc = (int64_t)Dc * Hc * Wc, Pf = Pc * 8;
  const int Hf = 2 * Hc, Wf = 2 * Wc;
  const int64_t total = (mode == 0 ? Pc : Pf) * CG * 2;  // float4 items per batch
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int half = (int)(i & 1);
    const int64_t pq = (i >> 1) % (mode == 0 ? Pc : Pf);
    const int cg = (int)((i >> 1) / (mode == 0 ? Pc : Pf));
    if (mode == 0) {
      const int x = (int)(pq % Wc), y = (int)((pq / Wc) % Hc), z = (int)(pq / ((int64_t)Wc * Hc));
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t pf = ((int64_t)(2 * z + (k >> 2)) * Hf + (2 * y + ((k >> 1) & 1))) * Wf + (2 * x + (k & 1));
        const f32x4 v = *(const f32x4*)(in + ((((int64_t)b * CG + cg) * Pf + pf) * 8 + half * 4));
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
      }
      f32x4* o = (f32x4*)(out + ((((int64_t)b * CG + cg) * Pc + pq) * 8 + half * 4));
      if (accumulate) { const f32x4 p = *o; s[0] += p[0]; s[1] += p[1]; s[2] += p[2]; s[3] += p[3]; }
      *o = s;
    } else {
      const int x = (int)(pq % Wf), y = (int)((pq / Wf) % Hf), z = (int)(pq / ((int64_t)Wf * Hf));
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if ((x & 1) && (y & 1) && (z & 1)) {
        const int64_t pc = ((int64_t)(z >> 1) * Hc + (y >> 1)) * Wc + (x >> 1);
        v = *(const f32x4*)(in + ((((int64_t)b * CG + cg) * Pc + pc) * 8 + half * 4));
      }
      *(f32x4*)(out + ((((int64_t)b * CG + cg) * Pf + pq) * 8 + half * 4)) = v;
    }
  }
}

extern "C" int md_grad_resample(const float* in, float* out, int32_t batch, int32_t C, int32_t Dc, int32_t Hc, int32_t Wc,
                                int32_t mode, int32_t accumulate, void* stream) {
  if (!in || !out || batch <= 0 || C <= 0 || (C % 8) || Dc <= 0 || Hc <= 0 || Wc <= 0 || mode < 0 || mode > 1) return MD_ERR_BAD_ARG;
  const int64_t total = (int64_t)Dc * Hc * Wc * (mode == 0 ? 1 : 8) * (C / 8) * 2;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_grad_resample_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, in, out,
                     C / 8, Dc, Hc, Wc, mode, accumulate);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- S16B blocked transpose: in [B][R/8][2][Cn][8r] -> out [B][Cn/8][2][R][8c]  (swap which index is 8-blocked) ------
__global__ void md_s16b_transpose_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int R, int Cn) {
  const int b = blockIdx.y;
  const int64_t total = (int64_t)(R / 8) * (Cn / 8) * 2;  // one thread = one 8x8 block of one plane
  const uint16_t* ib = in + (int64_t)b * R * Cn * 2;
  uint16_t* ob = out + (int64_t)b * R * Cn * 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int part = (int)(i & 1);
    const int cb = (int)((i >> 1) % (Cn / 8));
    const int rb = (int)((i >> 1) / (Cn / 8));
    uint16_t blk[8][8];  // [c][r]
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const uint4 v = *(const uint4*)(ib + ((((int64_t)rb * 2 + part) * Cn) + cb * 8 + c) * 8);
      const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { blk[c][2 * k] = (uint16_t)(w[k] & 0xffff); blk[c][2 * k + 1] = (uint16_t)(w[k] >> 16); }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      uint32_t w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (uint32_t)blk[2 * k][r] | ((uint32_t)blk[2 * k + 1][r] << 16);
      *(uint4*)(ob + ((((int64_t)cb * 2 + part) * R) + rb * 8 + r) * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

extern "C" int md_s16b_transpose(const void* in, void* out, int32_t batch, int32_t R, int32_t Cn, void* stream) {
  if (!in || !out || batch <= 0 || R <= 0 || Cn <= 0 || (R % 8) || (Cn % 8)) return MD_ERR_BAD_ARG;
  const int64_t total = (int64_t)(R / 8) * (Cn / 8) * 2;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_s16b_transpose_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)in, (uint16_t*)out, R, Cn);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- softmax-over-keys backward: dS[key][q] = alpha * P[key][q] * (dP[key][q] - sum_key' P[key'][q] dP[key'][q]) ------
// p: S16B [B][NK/8][2][NQ][8] (hi+lo = P), dp: fp32 [B][NK/8][NQ][8]; ds: S16B like p (split of dS).
__global__ __launch_bounds__(256) void md_softmax_keys_bwd_kernel(const uint16_t* __restrict__ p, const float* __restrict__ dp,
                                                                  uint16_t* __restrict__ ds, int nk, int nq, float alpha) {
  const int b = blockIdx.y;
  const int tid = threadIdx.x, half = tid & 1, ql = (tid >> 1) & 31, ks = tid >> 6;
  const int q = blockIdx.x * 32 + ql;
  const int nkb = nk / 8;
  const int64_t plane = (int64_t)nq * 8;
  const uint16_t* pb = p + (int64_t)b * nkb * 2 * plane;
  uint16_t* ob = ds + (int64_t)b * nkb * 2 * plane;
  const f32x4* dpp = (const f32x4*)(dp + (int64_t)b * nk * nq);
  float dot = 0.f;
  for (int kb = ks; kb < nkb; kb += 4) {
    const uint16_t* ph = pb + ((int64_t)kb * 2) * plane + (int64_t)q * 8 + half * 4;
    const uint2 h2 = *(const uint2*)ph, l2 = *(const uint2*)(ph + plane);
    const f32x4 dv = dpp[((int64_t)kb * nq + q) * 2 + half];
    const float pv[4] = {md_bf2f(h2.x & 0xffff) + md_bf2f(l2.x & 0xffff), md_bf2f(h2.x >> 16) + md_bf2f(l2.x >> 16),
                         md_bf2f(h2.y & 0xffff) + md_bf2f(l2.y & 0xffff), md_bf2f(h2.y >> 16) + md_bf2f(l2.y >> 16)};
    dot += pv[0] * dv[0] + pv[1] * dv[1] + pv[2] * dv[2] + pv[3] * dv[3];
  }
  dot += __shfl_xor(dot, 1, 64);
  __shared__ float sd[4][32];
  if (half == 0) sd[ks][ql] = dot;
  __syncthreads();
  const float tot = sd[0][ql] + sd[1][ql] + sd[2][ql] + sd[3][ql];
  for (int kb = ks; kb < nkb; kb += 4) {
    const int64_t o = ((int64_t)kb * 2) * plane + (int64_t)q * 8 + half * 4;
    const uint2 h2 = *(const uint2*)(pb + o), l2 = *(const uint2*)(pb + o + plane);
    const f32x4 dv = dpp[((int64_t)kb * nq + q) * 2 + half];
    const float pv[4] = {md_bf2f(h2.x & 0xffff) + md_bf2f(l2.x & 0xffff), md_bf2f(h2.x >> 16) + md_bf2f(l2.x >> 16),
                         md_bf2f(h2.y & 0xffff) + md_bf2f(l2.y & 0xffff), md_bf2f(h2.y >> 16) + md_bf2f(l2.y >> 16)};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) md_split(alpha * pv[e] * (dv[e] - tot), hi[e], lo[e]);
    *(uint2*)(ob + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(ob + o + plane) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
  }
}

extern "C" int md_softmax_keys_bwd(const void* p, const float* dp, void* ds, int32_t batch, int32_t n_keys, int32_t n_q,
                                   float alpha, void* stream) {
  if (!p || !dp || !ds || batch <= 0 || n_keys <= 0 || (n_keys % 8) || n_q <= 0 || (n_q % 32)) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_softmax_keys_bwd_kernel, dim3((unsigned)(n_q / 32), (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)p, dp, (uint16_t*)ds, n_keys, n_q, alpha);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
