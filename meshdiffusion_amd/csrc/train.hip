// Training-step kernels that do not depend on the U-Net backward:
// forward noising, masked DDPM loss (+ its gradient w.r.t. eps_hat), global grad norm, and the fused
// clip + Adam + EMA parameter update.  All HBM-streaming, 16 B per lane.
//
// Reference: lib/diffusion/losses.py:59-78 (perturb + loss), :38-52 (warm-up, clip, Adam step),
// lib/diffusion/models/ema.py:32-51 (EMA update).
#include "md_common.h"

#pragma clang fp contract(off)

// x_t = (sqrt_ac[b] * x0 + sqrt_1mac[b] * noise) * mask      (losses.py:63-65; same op order)
__global__ void md_ddpm_perturb_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                       const float* __restrict__ mask, const float* __restrict__ coef,
                                       float* __restrict__ out, int64_t CP, int64_t P) {
  const int b = blockIdx.y;
  const float a = coef[b * 2 + 0], s = coef[b * 2 + 1];
  const int64_t base = (int64_t)b * CP;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < CP; i += (int64_t)gridDim.x * blockDim.x * 4) {
    const f32x4 xv = *(const f32x4*)(x0 + base + i), nv = *(const f32x4*)(noise + base + i);
    f32x4 mv = {1.f, 1.f, 1.f, 1.f};
    if (mask) mv = *(const f32x4*)(mask + (i % P));
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float t0 = a * xv[e];
      const float t1 = s * nv[e];
      o[e] = (t0 + t1) * mv[e];
    }
    *(f32x4*)(out + base + i) = o;
  }
}

// sums[b] += sum_i (eps_hat - noise)^2 * mask   (fp64);  optional grad[b,i] = gscale * 2 (eps_hat-noise) * mask
__global__ __launch_bounds__(256) void md_masked_sq_err_kernel(const float* __restrict__ e, const float* __restrict__ n,
                                                               const float* __restrict__ mask, double* __restrict__ sums,
                                                               float* __restrict__ grad, float gscale, int64_t CP, int64_t P) {
  const int b = blockIdx.y;
  const int64_t base = (int64_t)b * CP;
  double acc = 0.0;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < CP; i += (int64_t)gridDim.x * blockDim.x * 4) {
    const f32x4 ev = *(const f32x4*)(e + base + i), nv = *(const f32x4*)(n + base + i);
    f32x4 mv = {1.f, 1.f, 1.f, 1.f};
    if (mask) mv = *(const f32x4*)(mask + (i % P));
    f32x4 g;
    float part = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = ev[k] - nv[k];
      part += (d * d) * mv[k];
      g[k] = gscale * 2.f * d * mv[k];
    }
    acc += (double)part;
    if (grad) *(f32x4*)(grad + base + i) = g;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[b], red[0] + red[1] + red[2] + red[3]);
}

// out[0] += sum g^2 (fp64)
__global__ __launch_bounds__(256) void md_grad_sqnorm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ out) {
  double acc = 0.0;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const f32x4 v = *(const f32x4*)(g + i * 4);
    acc += (double)(v[0] * v[0] + v[1] * v[1]) + (double)(v[2] * v[2] + v[3] * v[3]);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int64_t i = n4 * 4; i < n; ++i) acc += (double)g[i] * (double)g[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  __shared__ double red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// One pass over (p, g, m, v, ema): clip, Adam (torch.optim.Adam semantics), EMA.
//   g' = g * clip ; m = m + (g' - m)*(1-b1) ; v = v*b2 + (1-b2)*g'*g'
//   p = p - step_size * m / (sqrt(v)/bc2_sqrt + eps) ; ema = ema - (1-d)*(ema - p)
// `clip_from_sqnorm` (device double, may be NULL): clip = min(1, max_norm / (sqrt(*sq) + 1e-6)).
__global__ __launch_bounds__(256) void md_adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v,
                                                          float* __restrict__ ema, int64_t n, float b1, float b2,
                                                          float eps, float wd, float step_size, float bc2_sqrt,
                                                          float one_minus_decay, const double* __restrict__ sqnorm,
                                                          float max_norm) {
  float clip = 1.f;
  if (sqnorm != nullptr && max_norm >= 0.f) {
    const float tn = (float)sqrt(*sqnorm);
    const float c = max_norm / (tn + 1e-6f);
    clip = c < 1.f ? c : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    float gi = g[i] * clip;
    float pi = p[i];
    if (wd != 0.f) gi = gi + wd * pi;
    float mi = m[i], vi = v[i];
    mi = mi + (gi - mi) * (1.f - b1);
    vi = vi * b2 + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi = pi - step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (ema != nullptr) {
      const float s = ema[i];
      ema[i] = s - one_minus_decay * (s - pi);
    }
  }
}
#pragma clang fp contract(fast)

static int grid_for(int64_t items) {
  int64_t b = (items + 255) / 256;
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

extern "C" int md_ddpm_perturb(const float* x0, const float* noise, const float* mask, const float* coef,
                               float* out, int32_t batch, int32_t C, int64_t P, void* stream) {
  if (!x0 || !noise || !coef || !out || batch <= 0 || C <= 0 || P <= 0 || (P % 4)) return MD_ERR_BAD_ARG;
  const int64_t CP = (int64_t)C * P;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_ddpm_perturb_kernel, dim3((unsigned)grid_for(CP / 4), (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, x0, noise, mask, coef, out, CP, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_masked_sq_err(const float* eps_hat, const float* noise, const float* mask, double* sums,
                                float* grad, float gscale, int32_t batch, int32_t C, int64_t P, void* stream) {
  if (!eps_hat || !noise || !sums || batch <= 0 || C <= 0 || P <= 0 || (P % 4)) return MD_ERR_BAD_ARG;
  const int64_t CP = (int64_t)C * P;
  int blocks = grid_for(CP / 4);
  if (blocks > 256) blocks = 256;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_masked_sq_err_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, eps_hat, noise, mask, sums, grad, gscale, CP, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// md_absmax: amax_bits[0] = max(amax_bits[0], max |x|) as the float's bit pattern (non-negative floats order like their bits); NaN never
// wins, +inf does.  The caller zeroes the word.  One streaming read; used for the dynamic lift of the f16f6 data-gradient convs.
__global__ __launch_bounds__(256) void md_absmax_kernel(const float* __restrict__ x, int64_t n4, int64_t n, uint32_t* __restrict__ amax_bits) {
  float m = 0.f;
  const f32x4* x4 = (const f32x4*)x;
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  // eight independent 16-byte loads in flight per thread (one per iteration left the pass latency-bound at ~2.4 TB/s)
  for (; i + 7 * stride < n4; i += 8 * stride) {
    f32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(x4 + i + k * stride);
#pragma unroll
    for (int k = 0; k < 8; ++k) m = fmaxf(m, fmaxf(fmaxf(fabsf(v[k][0]), fabsf(v[k][1])), fmaxf(fabsf(v[k][2]), fabsf(v[k][3]))));
  }
  for (; i < n4; i += stride) {
    const f32x4 v = x4[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  if (blockIdx.x == 0)
    for (int64_t k = n4 * 4 + threadIdx.x; k < n; k += 256) m = fmaxf(m, fabsf(x[k]));
  m = md_wave_max(m);
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __builtin_nontemporal_load(amax_bits)) atomicMax(amax_bits, __float_as_uint(m));
}

extern "C" int md_absmax(const float* x, int64_t n, uint32_t* amax_bits, void* stream) {
  if (!x || !amax_bits || n <= 0 || ((uintptr_t)x & 15)) return MD_ERR_BAD_ARG;
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n4, n, amax_bits);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_grad_sqnorm(const float* g, int64_t n, double* out, void* stream) {
  if (!g || !out || n <= 0 || ((uintptr_t)g & 15)) return MD_ERR_BAD_ARG;
  int blocks = grid_for(n / 4);
  if (blocks > 1024) blocks = 1024;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_grad_sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, out);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_adam_ema_step(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                float ema_decay, const double* grad_sqnorm, float max_norm, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step <= 0) return MD_ERR_BAD_ARG;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_adam_ema_kernel, dim3((unsigned)grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                     ema, n, beta1, beta2, eps, weight_decay, step_size, bc2_sqrt, 1.f - ema_decay, grad_sqnorm, max_norm);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
