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
#include "md_common.h"

__global__ __launch_bounds__(256) void md_wino_equaliser_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k,
                                                                float* __restrict__ eq) {
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
  if (tid < 64) {
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
    const float aa = red[4] / red[5];
    float e = 0.f;
    if (gg > 0.f && aa > 0.f && gg < 1e30f && aa < 1e30f) {
      e = rintf(0.25f * (log2f(gg) - log2f(aa)));
      e = fminf(fmaxf(e, -14.f), 14.f);
    }
    eq[c] = ldexpf(1.0f, (int)e);
  }
}

extern "C" int md_wino_equaliser(const float* gamma, const float* beta, const float* w, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                                 float* eq, void* stream) {
  if (!gamma || !beta || !w || !eq || cout <= 0 || cin <= 0) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_equaliser_kernel, dim3((unsigned)cin), dim3(256), 0, (hipStream_t)stream, gamma, beta, w, cout, cin, s_row, s_k, eq);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
