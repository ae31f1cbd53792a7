// md_wino_equaliser: the static per-input-channel power-of-two equaliser of the f16f8 / f16f6 Winograd convs.
//
// Reference op: nn.GroupNorm(32, C) -> nn.SiLU -> nn.Conv3d 3x3x3 (lib/diffusion/models/layers.py:652-662, 676-682; the same pair in
// front of the head, ddpm_res64.py:120-121,186-189), computed by the reference in fp32.  The f16f8 / f16f6 arithmetic (md_common.h)
// takes the two cross terms of a product from 4-bit-significand images; in the f16f6 form the 16 input channels of a K block share
// ONE power-of-two scale, so a channel whose activation is 2^-3 of its block's largest keeps one or two significant bits and a
// channel at 2^-6 none (measured on the CPU model, tools/f16f8_numerics.py: 1.8e-5 -> 1.1e-4 -> 3.1e-4 per conv).  A trained
// GroupNorm affine produces exactly that spread -- and it is known BEFORE any data is seen: gamma_c, beta_c and the weight rows are
// parameters.  So, once per weight version:
//     a_c = rms of silu(gamma_c z + beta_c), z ~ N(0, 1)      (what GroupNorm hands the activation, per channel)
//     g_c = rms of w[:, c, :, :, :]
//     s_c = 2^round(log2(g_c / a_c) / 2)                       (SmoothQuant's alpha = 1/2 split, rounded to a power of two)
// the operand pass multiplies the activated value by s_c (md_wino_prep_f8 / _f6 `eq`), the packed weights carry w / s_c
// (md_wino_pack_weights_f8 / _f6 `eq`): both exact, the product unchanged, and activations AND weights of a block now have the
// magnitude sqrt(a_c g_c) -- flat whenever every channel matters equally, and where it is not flat the small channels are the ones
// that matter less.  The expectation is a 64-point midpoint rule on z in [-6, 6] (tests/ and tools/f16f8_numerics.py restate it).
// Round 6: (1) `a2m` (may be null): MEASURED per-channel mean squares of the operand (md_wino_operand_ms over one calibration
// evaluation, DDPMUNet3D.calibrate) replace the static estimate a_c^2 -- GroupNorm normalises GROUPS of 4-16 channels, so a channel's own
// scale inside its group survives it and the static estimate (unit variance per channel) misses it; the raw residual stream in front
// of an Upsample conv has no GroupNorm at all (gamma = beta = null: measured only).  (2) the exponent is bounded by the fp16 headroom
// of the operand's hi plane: the Winograd input transform doubles a value, so s_c (|gamma_c| z_max + |beta_c|) 2 < 2^15 with
// z_max = 8 -- a near-dead channel (a_c ~ 0) no longer gets 2^14 and turns an outlier voxel into inf.
#include "md_common.h"

__global__ __launch_bounds__(256) void md_wino_equaliser_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k,
                                                                const float* __restrict__ a2m, float* __restrict__ eq) {
  __shared__ float red[8];
  const int c = blockIdx.x, tid = threadIdx.x;
  float g2 = 0.f;
  const int n = cout * 27;
  for (int i = tid; i < n; i += 256) {
    const float v = w[(int64_t)(i / 27) * s_row + (int64_t)c * s_k + (i % 27)];
    g2 += v * v;
  }
  g2 = md_wave_sum(g2);
  float a2 = 0.f, pz = 0.f;
  if (tid < 64 && gamma != nullptr) {
    const float z = -6.0f + 12.0f * ((float)tid + 0.5f) / 64.0f;
    const float pdf = expf(-0.5f * z * z);
    const float y = gamma[c] * z + beta[c];
    const float sy = y / (1.0f + expf(-y));
    a2 = md_wave_sum(sy * sy * pdf);
    pz = md_wave_sum(pdf);
  }
  if ((tid & 63) == 0) red[tid >> 6] = g2;
  if (tid == 0) { red[4] = a2; red[5] = pz; }
  __syncthreads();
  if (tid == 0) {
    const float gg = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
    const float aa = a2m != nullptr ? a2m[c] : (gamma != nullptr ? red[4] / red[5] : 0.f);
    float e = 0.f;
    if (gg > 0.f && aa > 0.f && gg < 1e30f && aa < 1e30f) {
      e = rintf(0.25f * (log2f(gg) - log2f(aa)));
      float hi = 14.f;
      if (gamma != nullptr) {
        const float top = 8.0f * fabsf(gamma[c]) + fabsf(beta[c]);      // the largest activation GroupNorm + SiLU can hand over at |z| <= 8
        if (top > 0.f && top < 1e30f) hi = fminf(hi, floorf(14.f - log2f(top)));
      }
      e = fminf(fmaxf(e, -14.f), fmaxf(hi, -14.f));
    }
    eq[c] = ldexpf(1.0f, (int)e);
  }
}

// md_wino_operand_ms: per-channel mean square of the operand a GroupNorm -> SiLU -> conv (or a raw conv: ac = null) reads, over all
// samples and positions: ms[c] += sum act^2 / (batch * P) (ms zeroed by the caller).  One thread = one position x 8 channels.
__global__ __launch_bounds__(256) void md_wino_operand_ms_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int c1, int c2,
                                                                 const float* __restrict__ ac, int silu, int batch, int64_t P, float inv_n,
                                                                 float* __restrict__ ms) {
  __shared__ float wsum[4][8];
  const int CG = (c1 + c2) >> 3;
  const int64_t nblk = (P + 255) / 256;
  const int64_t blk = blockIdx.x % nblk;
  const int cg = (int)((blockIdx.x / nblk) % CG);
  const int b = (int)(blockIdx.x / (nblk * CG));
  const int64_t p = blk * 256 + threadIdx.x;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  if (p < P) {
    const float* src = (cg * 8 < c1) ? x1 + ((int64_t)b * (c1 >> 3) + cg) * P * 8 : x2 + ((int64_t)b * (c2 >> 3) + (cg - (c1 >> 3))) * P * 8;
    const f32x4 v0 = *(const f32x4*)(src + p * 8), v1 = *(const f32x4*)(src + p * 8 + 4);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = e < 4 ? v0[e] : v1[e - 4];
      if (ac != nullptr) {
        const float* ap = ac + ((int64_t)b * (c1 + c2) + cg * 8 + e) * 2;
        t = t * ap[0] + ap[1];
        if (silu) t = t / (1.0f + expf(-t));
      }
      v[e] = t * t;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float r = md_wave_sum(v[e]);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][e] = r;
  }
  __syncthreads();
  if (threadIdx.x < 8)
    atomicAdd(ms + cg * 8 + threadIdx.x, ((wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x])) * inv_n);
}

// measured form only: shift every exponent by the same u so that the largest equalised channel rms, max_c eq_c sqrt(a2m_c), lands at
// 2^0 -- the operand's fp16 hi plane then sits mid-range whatever the tensor's absolute magnitude is (a 1e-5-magnitude residual
// stream would otherwise be subnormal there); the packed weights absorb 2^-u in their own power-of-two pre-scale.  One workgroup.
__global__ __launch_bounds__(256) void md_wino_equaliser_level_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                      const float* __restrict__ a2m, int cin, float* __restrict__ eq) {
  __shared__ float red[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < cin; c += 256) {
    const float v = eq[c] * sqrtf(fmaxf(a2m[c], 0.f));
    m = (v > m && v < 1e30f) ? v : m;
  }
  m = md_wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if (!(m > 0.f)) return;
  const int u = -(int)rintf(log2f(m));
  for (int c = threadIdx.x; c < cin; c += 256) {
    int e = ilogbf(eq[c]) + u;
    if (gamma != nullptr) {
      const float top = 8.0f * fabsf(gamma[c]) + fabsf(beta[c]);
      if (top > 0.f && top < 1e30f) e = min(e, (int)floorf(14.f - log2f(top)));
    }
    eq[c] = ldexpf(1.0f, max(-60, min(60, e)));
  }
}

static int md_wino_equaliser_launch(const float* gamma, const float* beta, const float* w, const float* a2m, int32_t cout, int32_t cin, int64_t s_row,
                                    int64_t s_k, float* eq, void* stream) {
  if (!w || !eq || cout <= 0 || cin <= 0 || ((gamma == nullptr) != (beta == nullptr)) || (!gamma && !a2m)) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_equaliser_kernel, dim3((unsigned)cin), dim3(256), 0, (hipStream_t)stream, gamma, beta, w, cout, cin, s_row, s_k, a2m, eq);
  MD_HIP_CHECK_LAUNCH();
  if (a2m != nullptr) {
    hipLaunchKernelGGL(md_wino_equaliser_level_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, gamma, beta, a2m, cin, eq);
    MD_HIP_CHECK_LAUNCH();
  }
  return MD_OK;
}

extern "C" int md_wino_equaliser(const float* gamma, const float* beta, const float* w, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                                 float* eq, void* stream) {
  if (!gamma || !beta) return MD_ERR_BAD_ARG;
  return md_wino_equaliser_launch(gamma, beta, w, nullptr, cout, cin, s_row, s_k, eq, stream);
}

extern "C" int md_wino_equaliser_measured(const float* gamma, const float* beta, const float* w, const float* a2m, int32_t cout, int32_t cin,
                                          int64_t s_row, int64_t s_k, float* eq, void* stream) {
  if (!a2m) return MD_ERR_BAD_ARG;
  return md_wino_equaliser_launch(gamma, beta, w, a2m, cout, cin, s_row, s_k, eq, stream);
}

extern "C" int md_wino_operand_ms(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t batch, int64_t P,
                                  float* ms, void* stream) {
  if (!x1 || !ms || batch <= 0 || P <= 0 || c1 <= 0 || c2 < 0 || (c1 & 7) || (c2 & 7) || (c2 > 0 && !x2) || (silu && !ac)) return MD_ERR_BAD_ARG;
  const int64_t blocks = (int64_t)batch * ((c1 + c2) / 8) * ((P + 255) / 256);
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_operand_ms_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac, silu, batch, P,
                     1.0f / ((float)batch * (float)P), ms);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
