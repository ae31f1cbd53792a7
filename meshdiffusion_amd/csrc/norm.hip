// GroupNorm(32, C, eps) statistics / apply (+SiLU) / bf16x3 split, 8-channel blocked layouts.
// Replaces nn.GroupNorm + nn.SiLU (reference lib/diffusion/models/layers.py:652,660,676,681,589;
// ddpm_res64.py:120,186) and the torch.cat before it (ddpm_res64.py:174-176).
// All three kernels are pure HBM streaming: 16 B per lane, consecutive lanes contiguous.
#include "md_common.h"
#include <hip/hip_fp16.h>

static constexpr int GN_BLOCK = 256;
static constexpr int GN_ITEMS = 16;                              // float4 items per thread
static constexpr int GN_CHUNK = GN_BLOCK * GN_ITEMS / 2;         // positions per block

// x: F32B [B][C/8][P][8]; sums: double [B][c_total][2] (sum, sumsq), pre-zeroed.
__global__ __launch_bounds__(GN_BLOCK) void md_gn_stats_kernel(const float* __restrict__ x,
                                                               double* __restrict__ sums, int C,
                                                               int64_t P, int c_total, int c_off) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x, half = tid & 1;
  const int64_t p0 = (int64_t)blockIdx.x * GN_CHUNK;
  const f32x4* xp = (const f32x4*)(x + (((int64_t)b * (C / 8) + cg) * P) * 8);
  float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < GN_ITEMS; ++i) {
    const int64_t pos = p0 + ((tid + i * GN_BLOCK) >> 1);
    if (pos < P) {
      const f32x4 v = xp[pos * 2 + half];
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
    }
  }
  // reduce over lanes with equal parity (keeps `half`), then over the 4 waves
#pragma unroll
  for (int e = 0; e < 4; ++e) {
#pragma unroll
    for (int o = 32; o > 1; o >>= 1) { s[e] += __shfl_xor(s[e], o, 64); q[e] += __shfl_xor(q[e], o, 64); }
  }
  __shared__ float red[GN_BLOCK / 64][2][8];
  const int lane = tid & 63, wid = tid >> 6;
  if (lane < 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wid][lane][e] = s[e]; red[wid][lane][4 + e] = q[e]; }
  }
  __syncthreads();
  if (tid < 16) {
    const int hh = (tid >> 2) & 1, e = tid & 3, isq = tid >> 3;
    double acc = 0.0;
    for (int w = 0; w < GN_BLOCK / 64; ++w) acc += (double)red[w][hh][isq * 4 + e];
    const int c = c_off + cg * 8 + hh * 4 + e;
    atomicAdd(&sums[((int64_t)b * c_total + c) * 2 + isq], acc);
  }
}

// params: float4 [B][c_total] = (mean, rstd*gamma, beta, rstd)
__global__ void md_gn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, float* __restrict__ params,
                                      int c_total, int groups, int64_t P, float eps, float* __restrict__ ac) {
  const int b = blockIdx.x / groups, g = blockIdx.x % groups;
  const int cpg = c_total / groups;
  const int lane = threadIdx.x;
  double s = 0.0, q = 0.0;
  for (int c = lane; c < cpg; c += 64) {
    const int64_t o = ((int64_t)b * c_total + g * cpg + c) * 2;
    s += sums[o];
    q += sums[o + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  const double n = (double)cpg * (double)P;
  const double mean = s / n;
  double var = q / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int c = lane; c < cpg; c += 64) {
    const int ch = g * cpg + c;
    f32x4 o4 = {(float)mean, rstd * gamma[ch], beta[ch], rstd};  // [3] = rstd: used by the backward pass
    *(f32x4*)(params + ((int64_t)b * c_total + ch) * 4) = o4;
    if (ac != nullptr) {   // y = x*a + c, the form the conv's fused halo loader applies
      const float a_ = rstd * gamma[ch];
      ac[((int64_t)b * c_total + ch) * 2] = a_;
      ac[((int64_t)b * c_total + ch) * 2 + 1] = (float)((double)beta[ch] - mean * (double)a_);
    }
  }
}

// out: S16B [B][c_total/8][2][P][8].  One thread = one position x 8 channels: two 16-byte loads (32 contiguous bytes),
// one 16-byte store per output plane, so every store instruction of a wave covers 1 KiB of consecutive positions.
__global__ __launch_bounds__(GN_BLOCK) void md_gn_apply_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ params,
                                                               uint16_t* __restrict__ out,
                                                               uint16_t* __restrict__ out_raw, int C,
                                                               int64_t P, int c_total, int c_off,
                                                               int norm, int silu, uint32_t thr16,
                                                               float drop_scale, uint64_t seed) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * GN_CHUNK;
  const f32x4* xp = (const f32x4*)(x + (((int64_t)b * (C / 8) + cg) * P) * 8);
  float mean[8], a[8], bt[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (norm) {
      const f32x4 pr = *(const f32x4*)(params + ((int64_t)b * c_total + c_off + cg * 8 + e) * 4);
      mean[e] = pr[0]; a[e] = pr[1]; bt[e] = pr[2];
    } else {
      mean[e] = 0.f; a[e] = 1.f; bt[e] = 0.f;
    }
  }
  const int64_t plane = P * 8;
  uint16_t* ohi = out + (((int64_t)b * (c_total / 8) + (c_off / 8) + cg) * 2) * plane;
  uint16_t* rhi = out_raw ? out_raw + (((int64_t)b * (c_total / 8) + (c_off / 8) + cg) * 2) * plane : nullptr;
  const uint64_t quad0 = (uint64_t)(((int64_t)b * c_total + c_off + cg * 8) >> 2) * (uint64_t)P;   // dropout quad index base
#pragma unroll
  for (int i = 0; i < GN_ITEMS / 2; ++i) {
    const int64_t pos = p0 + tid + i * GN_BLOCK;
    if (pos < P) {
      const f32x4 v0 = xp[pos * 2], v1 = xp[pos * 2 + 1];
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      uint32_t hi[8], lo[8];
      if (rhi) {  // second output: bf16 split of the raw input (operand of the NIN shortcut), same read
#pragma unroll
        for (int e = 0; e < 8; ++e) md_split(v[e], hi[e], lo[e]);
        *(uint4*)(rhi + pos * 8) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
        *(uint4*)(rhi + plane + pos * 8) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
      }
      uint64_t bits[2] = {0, 0};
      if (thr16) {
        bits[0] = md_drop_bits(seed, quad0 + (uint64_t)pos);
        bits[1] = md_drop_bits(seed, quad0 + (uint64_t)P + (uint64_t)pos);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float y = v[e];
        if (norm) y = (y - mean[e]) * a[e] + bt[e];
        if (silu & 1) y = md_silu(y);
        if (thr16) y = md_drop_keep(bits[e >> 2], e & 3, thr16) ? y * drop_scale : 0.f;   // nn.Dropout (layers.py:682)
        // experiment hook (tests/longrun_parity.py, MD_DEBUG_ACT_FP16=1): round the operand to fp16 first, which is
        // what a weights-split-only fp16 scheme (2 MFMAs per product) would feed the matrix cores
        if (silu & 2) y = __half2float(__float2half_rn(y));
        if (silu & 4) { hi[e] = md_f2h(y); lo[e] = 0; } else md_split(y, hi[e], lo[e]);
      }
      *(uint4*)(ohi + pos * 8) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
      if (!(silu & 4))
        *(uint4*)(ohi + plane + pos * 8) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
    }
  }
}

extern "C" int md_gn_stats(const float* x, double* sums, int32_t batch, int32_t C, int64_t P,
                           int32_t c_total, int32_t c_off, void* stream) {
  if (!x || !sums || batch <= 0 || C <= 0 || (C % 8) || P <= 0 || c_off < 0 || c_off + C > c_total)
    return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + GN_CHUNK - 1) / GN_CHUNK), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_stats_kernel, grid, dim3(GN_BLOCK), 0, (hipStream_t)stream, x, sums, C, P,
                     c_total, c_off);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_gn_finalize(const double* sums, const float* gamma, const float* beta,
                              float* params, int32_t batch, int32_t c_total, int32_t groups,
                              int64_t P, float eps, float* ac, void* stream) {
  if (!sums || !gamma || !beta || !params || batch <= 0 || groups <= 0 || (c_total % groups))
    return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_finalize_kernel, dim3((unsigned)(batch * groups)), dim3(64), 0,
                     (hipStream_t)stream, sums, gamma, beta, params, c_total, groups, P, eps, ac);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_gn_apply(const float* x, const float* params, void* out, void* out_raw, int32_t batch,
                           int32_t C, int64_t P, int32_t c_total, int32_t c_off, int32_t norm, int32_t silu,
                           float drop_p, uint64_t drop_seed, void* stream) {
  if (!(drop_p >= 0.f && drop_p < 1.f)) return MD_ERR_BAD_ARG;
  if (!x || !out || (norm && !params) || batch <= 0 || C <= 0 || (C % 8) || (c_off % 8) ||
      (c_total % 8) || c_off + C > c_total || P <= 0)
    return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + GN_CHUNK - 1) / GN_CHUNK), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gn_apply_kernel, grid, dim3(GN_BLOCK), 0, (hipStream_t)stream, x, params,
                     (uint16_t*)out, (uint16_t*)out_raw, C, P, c_total, c_off, norm, silu, md_drop_thr16(drop_p),
                     1.0f / (1.0f - drop_p), (uint64_t)drop_seed);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_zero(void* p, int64_t bytes, void* stream) {
  if (!p || bytes < 0) return MD_ERR_BAD_ARG;
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  return e == hipSuccess ? MD_OK : (int)e;
}

// keep/(1-p) factors of the dropout mask as an F32B tensor [B][C/8][P][8] (tests build the reference's explicit mask
// from it; the training path itself never materialises the mask).
__global__ __launch_bounds__(256) void md_dropout_scale_kernel(float* __restrict__ out, int C, int64_t P, int c_total,
                                                               int c_off, uint32_t thr16, float scale, uint64_t seed) {
  const int cg = blockIdx.y, b = blockIdx.z;
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t pos = t >> 1;
  const int half = (int)(t & 1);
  if (pos >= P) return;
  const uint64_t bits = md_drop_bits(seed, (uint64_t)(((int64_t)b * c_total + c_off + cg * 8 + half * 4) >> 2) * (uint64_t)P + (uint64_t)pos);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (!thr16 || md_drop_keep(bits, e, thr16)) ? scale : 0.f;
  *(f32x4*)(out + ((((int64_t)b * (C / 8) + cg) * P + pos) * 8 + half * 4)) = o;
}

extern "C" int md_dropout_scale(float* out, int32_t batch, int32_t C, int64_t P, int32_t c_total, int32_t c_off,
                                float drop_p, uint64_t drop_seed, void* stream) {
  if (!out || batch <= 0 || C <= 0 || (C % 8) || (c_off % 8) || c_off + C > c_total || P <= 0 || !(drop_p >= 0.f && drop_p < 1.f))
    return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P * 2 + 255) / 256), (unsigned)(C / 8), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_dropout_scale_kernel, grid, dim3(256), 0, (hipStream_t)stream, out, C, P, c_total, c_off,
                     md_drop_thr16(drop_p), 1.0f / (1.0f - drop_p), (uint64_t)drop_seed);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
