// Small streaming kernels: timestep embedding, dense layers, layout conversion,
// the DDPM ancestral update and the inpainting blend.
#include "md_common.h"

// ---- y[b][o] = sum_i act(x[b][i]) * w[o][i] + bias[o]  (nn.Linear; one wave per output) ----
template <int BB>
__global__ __launch_bounds__(256) void md_linear_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w,
                                                        const float* __restrict__ bias,
                                                        float* __restrict__ y, int batch, int in_dim,
                                                        int out_dim, int silu_in) {
  const int lane = threadIdx.x & 63;
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= out_dim) return;
  const float* wr = w + (int64_t)o * in_dim;
  for (int b0 = 0; b0 < batch; b0 += BB) {
    float acc[BB];
#pragma unroll
    for (int k = 0; k < BB; ++k) acc[k] = 0.f;
    for (int i = lane; i < in_dim; i += 64) {
      const float wv = wr[i];
#pragma unroll
      for (int k = 0; k < BB; ++k) {
        if (b0 + k < batch) {
          float xv = x[(int64_t)(b0 + k) * in_dim + i];
          if (silu_in) xv = md_silu(xv);
          acc[k] += xv * wv;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < BB; ++k) {
      const float s = md_wave_sum(acc[k]);
      if (lane == 0 && b0 + k < batch) y[(int64_t)(b0 + k) * out_dim + o] = s + (bias ? bias[o] : 0.f);
    }
  }
}

extern "C" int md_linear(const float* x, const float* w, const float* bias, float* y, int32_t batch,
                         int32_t in_dim, int32_t out_dim, int32_t silu_in, void* stream) {
  if (!x || !w || !y || batch <= 0 || in_dim <= 0 || out_dim <= 0) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_linear_kernel<8>, dim3((unsigned)((out_dim + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, w, bias, y, batch, in_dim, out_dim, silu_in);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- layout conversions ----------------------------------------------------------------
// NCDHW fp32 [B][C][P] -> S16B [B][c_pad/8][2][P][8]
__global__ void md_ncdhw_to_s16b_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C,
                                        int c_pad, int64_t P) {
  const int b = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= P) return;
  for (int cg = 0; cg < c_pad / 8; ++cg) {
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e;
      const float v = (c < C) ? x[((int64_t)b * C + c) * P + pos] : 0.f;
      md_split(v, hi[e], lo[e]);
    }
    uint4* o = out + (((int64_t)b * (c_pad / 8) + cg) * 2) * P + pos;
    o[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    o[P] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
}

extern "C" int md_ncdhw_to_s16b(const float* x, void* out, int32_t batch, int32_t C, int32_t c_pad,
                                int64_t P, void* stream) {
  if (!x || !out || batch <= 0 || C <= 0 || c_pad < C || (c_pad % 8) || P <= 0) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_ncdhw_to_s16b_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (uint4*)out,
                     C, c_pad, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// NCDHW fp32 [B][C][D][H][W] -> S16B [B][c_pad/8][2][P][8] with the kx x-shifted copies of every channel side by side:
// channel ci*kx + dx holds x[ci] shifted by dx - kx/2 along x (zero outside the grid).  Operand of the dx-folded stem
// conv: a k^3 conv over C channels = a k x k x 1 conv over C*kx channels (K = 12 or 20 instead of 4 per tap, k times
// fewer taps).
__global__ void md_ncdhw_to_s16b_xfold_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C, int kx, int c_pad,
                                              int64_t P, int W) {
  const int b = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= P) return;
  const int xx = (int)(pos % W);
  for (int cg = 0; cg < c_pad / 8; ++cg) {
    uint32_t hi[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = cg * 8 + e, ci = c / kx, sh = c % kx - kx / 2;
      float v = 0.f;
      if (ci < C && xx + sh >= 0 && xx + sh < W) v = x[((int64_t)b * C + ci) * P + pos + sh];
      md_split(v, hi[e], lo[e]);
    }
    uint4* o = out + (((int64_t)b * (c_pad / 8) + cg) * 2) * P + pos;
    o[0] = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    o[P] = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
  }
}

extern "C" int md_ncdhw_to_s16b_xfold(const float* x, void* out, int32_t batch, int32_t C, int32_t kx, int32_t c_pad, int32_t D,
                                      int32_t H, int32_t W, void* stream) {
  if (!x || !out || batch <= 0 || C <= 0 || (kx != 3 && kx != 5) || c_pad < C * kx || (c_pad % 8) || D <= 0 || H <= 0 || W <= 0)
    return MD_ERR_BAD_ARG;
  const int64_t P = (int64_t)D * H * W;
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_ncdhw_to_s16b_xfold_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, (uint4*)out, C, kx, c_pad, P, W);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// mode 0: F32B -> NCDHW ; mode 1: NCDHW -> F32B ; mode 2: S16B (hi+lo) -> NCDHW
__global__ void md_relayout_kernel(const void* __restrict__ xin, float* __restrict__ out, int C,
                                   int64_t P, int mode) {
  const int b = blockIdx.z, c = blockIdx.y;
  const int64_t pos = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= P) return;
  const int cg = c >> 3, e = c & 7;
  const int64_t nc = ((int64_t)b * C + c) * P + pos;
  if (mode == 0) {
    out[nc] = ((const float*)xin)[(((int64_t)b * (C / 8) + cg) * P + pos) * 8 + e];
  } else if (mode == 1) {
    out[(((int64_t)b * (C / 8) + cg) * P + pos) * 8 + e] = ((const float*)xin)[nc];
  } else {
    const uint16_t* s = (const uint16_t*)xin;
    const int64_t o = ((((int64_t)b * (C / 8) + cg) * 2) * P + pos) * 8 + e;
    out[nc] = md_bf2f(s[o]) + md_bf2f(s[o + P * 8]);
  }
}

static int relayout(const void* x, float* out, int32_t batch, int32_t C, int64_t P, int mode, void* stream) {
  if (!x || !out || batch <= 0 || C <= 0 || (C % 8) || P <= 0) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)((P + 255) / 256), (unsigned)C, (unsigned)batch);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_relayout_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, out, C, P, mode);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
extern "C" int md_f32b_to_ncdhw(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream) {
  return relayout(x, out, batch, C, P, 0, stream);
}
extern "C" int md_ncdhw_to_f32b(const float* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream) {
  return relayout(x, out, batch, C, P, 1, stream);
}
extern "C" int md_s16b_to_ncdhw(const void* x, float* out, int32_t batch, int32_t C, int64_t P, void* stream) {
  return relayout(x, out, batch, C, P, 2, stream);
}

// ---- DDPM ancestral update (reference models/utils.py:191-198, sampling.py:222-230,476-478) ----
// Same operation order as the reference, no FMA contraction, so that with identical eps/z the
// update is bit-identical to the PyTorch elementwise chain.
#pragma clang fp contract(off)
__global__ void md_ancestral_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                         const float* __restrict__ z, const float* __restrict__ mask,
                                         const float* __restrict__ coef, float* __restrict__ x_out,
                                         float* __restrict__ xm_out, int64_t CP, int64_t P) {
  const int b = blockIdx.y;
  const float beta = coef[b * 4 + 0], sigma = coef[b * 4 + 1], sq1mb = coef[b * 4 + 2], sqb = coef[b * 4 + 3];
  const int64_t base = (int64_t)b * CP;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < CP;
       i += (int64_t)gridDim.x * blockDim.x * 4) {
    const f32x4 xv = *(const f32x4*)(x + base + i);
    const f32x4 ev = *(const f32x4*)(eps + base + i);
    const f32x4 zv = *(const f32x4*)(z + base + i);
    f32x4 mv = {1.f, 1.f, 1.f, 1.f};
    if (mask) mv = *(const f32x4*)(mask + (i % P));
    f32x4 xo, xmo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float score = (-ev[e]) / sigma;
      const float xm = (xv[e] + beta * score) / sq1mb;
      const float xn = xm + sqb * zv[e];
      xo[e] = xn * mv[e];
      xmo[e] = xm * mv[e];
    }
    *(f32x4*)(x_out + base + i) = xo;
    *(f32x4*)(xm_out + base + i) = xmo;
  }
}

extern "C" int md_ancestral_step(const float* x, const float* eps, const float* z, const float* mask,
                                 const float* coef, float* x_out, float* x_mean_out, int32_t batch,
                                 int32_t C, int64_t P, void* stream) {
  if (!x || !eps || !z || !coef || !x_out || !x_mean_out || batch <= 0 || C <= 0 || P <= 0 || (P % 4))
    return MD_ERR_BAD_ARG;
  const int64_t CP = (int64_t)C * P;
  int blocks = (int)((CP / 4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_ancestral_step_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, x, eps, z, mask, coef, x_out, x_mean_out, CP, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- deterministic DDIM update (reference sde_lib.py:113-140 `discretize_ddim`, sampling.py:556-565) ----
// The reference keeps the state in float64 from the first update on (x.double() arithmetic, float32 only for the U-Net
// input); so does this kernel.  Same operation order, no FMA contraction:
//   x0s = x - a2*eps ; sst = x - x0s ; x0_pred = x0s / a1 ; x_new = r1*x + (r2 - r1)*sst   (r1 = a1p/a1, r2 = a2p/a2)
// then both results times the grid mask, and the optional channel-`ch` inpainting blend v*(1-pm) + partial*pm
// (sampling.py:562-564).  coef[b] = {a1, a2, r1, r2} as doubles (a1, a2 are float32 table entries widened).
__global__ void md_ddim_step_kernel(const double* __restrict__ x, const float* __restrict__ eps,
                                    const float* __restrict__ mask, const double* __restrict__ coef,
                                    const float* __restrict__ partial, const float* __restrict__ pmask, int ch,
                                    double* __restrict__ x_out, double* __restrict__ x0_out, float* __restrict__ x_f32,
                                    int64_t CP, int64_t P) {
  const int b = blockIdx.y;
  const double a1 = coef[b * 4 + 0], a2 = coef[b * 4 + 1], r1 = coef[b * 4 + 2], r2 = coef[b * 4 + 3];
  const double r21 = (-r1) + r2;
  const int64_t base = (int64_t)b * CP;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < CP; i += (int64_t)gridDim.x * blockDim.x) {
    const double xv = x[base + i];
    const double x0s = xv - a2 * (double)eps[base + i];
    const double sst = xv - x0s;
    double x0p = x0s / a1;
    double xn = r1 * xv + r21 * sst;
    const int64_t pos = i % P;
    if (mask) { const double m = (double)mask[pos]; xn = xn * m; x0p = x0p * m; }
    if (partial != nullptr && i / P == ch) {
      const double pm = (double)pmask[pos], pv = (double)partial[pos];
      xn = xn * (1.0 - pm) + pv * pm;
      x0p = x0p * (1.0 - pm) + pv * pm;
    }
    x_out[base + i] = xn;
    x0_out[base + i] = x0p;
    x_f32[base + i] = (float)xn;
  }
}

extern "C" int md_ddim_step(const double* x, const float* eps, const float* mask, const double* coef, const float* partial,
                            const float* pmask, int32_t ch, double* x_out, double* x0_out, float* x_f32, int32_t batch,
                            int32_t C, int64_t P, void* stream) {
  if (!x || !eps || !coef || !x_out || !x0_out || !x_f32 || batch <= 0 || C <= 0 || P <= 0) return MD_ERR_BAD_ARG;
  if ((partial != nullptr) != (pmask != nullptr) || (partial != nullptr && (ch < 0 || ch >= C))) return MD_ERR_BAD_ARG;
  const int64_t CP = (int64_t)C * P;
  int blocks = (int)((CP + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_ddim_step_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, x, eps,
                     mask, coef, partial, pmask, ch, x_out, x0_out, x_f32, CP, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- inpainting blend of one channel (reference sampling.py:455-466) -----------------------
__global__ void md_inpaint_blend_kernel(float* __restrict__ x, const float* __restrict__ src,
                                        const float* __restrict__ pmask, const float* __restrict__ gmask,
                                        int C, int ch, int64_t P, int64_t src_bstride) {
  const int b = blockIdx.y;
  float* xc = x + ((int64_t)b * C + ch) * P;
  const float* sc = src + (int64_t)b * src_bstride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = pmask[i];
    float v = xc[i] * (1.f - m) + sc[i] * m;
    if (gmask) v = v * gmask[i];
    xc[i] = v;
  }
}

// re-noise the conditioned channel to level t (reference sampling.py:460-466):
//   upd = mean_coef[b]*x + std[b]*z ;  x = (x*(1-m) + upd*m)*gm ;  x_mean[ch] = x[ch]
__global__ void md_inpaint_renoise_kernel(float* __restrict__ x, float* __restrict__ x_mean,
                                          const float* __restrict__ z, const float* __restrict__ pmask,
                                          const float* __restrict__ gmask, const float* __restrict__ coef,
                                          int C, int ch, int64_t P) {
  const int b = blockIdx.y;
  const float mc = coef[b * 2 + 0], sd = coef[b * 2 + 1];
  float* xc = x + ((int64_t)b * C + ch) * P;
  float* xm = x_mean ? x_mean + ((int64_t)b * C + ch) * P : nullptr;
  const float* zc = z + (int64_t)b * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (int64_t)gridDim.x * blockDim.x) {
    const float m = pmask[i];
    const float xv = xc[i];
    const float mean = mc * xv;
    const float upd = mean + sd * zc[i];
    float v = xv * (1.f - m) + upd * m;
    if (gmask) v = v * gmask[i];
    xc[i] = v;
    if (xm) xm[i] = v;
  }
}
#pragma clang fp contract(fast)

extern "C" int md_inpaint_blend(float* x, const float* src, const float* pmask, const float* gmask,
                                int32_t batch, int32_t C, int32_t ch, int64_t P, int64_t src_bstride,
                                void* stream) {
  if (!x || !src || !pmask || batch <= 0 || C <= 0 || ch < 0 || ch >= C || P <= 0) return MD_ERR_BAD_ARG;
  int blocks = (int)((P + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_inpaint_blend_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, x, src, pmask, gmask, C, ch, P, src_bstride);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_inpaint_renoise(float* x, float* x_mean, const float* z, const float* pmask,
                                  const float* gmask, const float* coef, int32_t batch, int32_t C,
                                  int32_t ch, int64_t P, void* stream) {
  if (!x || !z || !pmask || !coef || batch <= 0 || C <= 0 || ch < 0 || ch >= C || P <= 0) return MD_ERR_BAD_ARG;
  int blocks = (int)((P + 255) / 256);
  if (blocks > 1024) blocks = 1024;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_inpaint_renoise_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, x, x_mean, z, pmask, gmask, coef, C, ch, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- attention softmax over keys (reference layers.py:603-605) -------------------------------
// s: [B][NK/8][NQ][8] fp32 (S^T, keys blocked by 8); p: S16B [B][NK/8][2][NQ][8].
// A block owns 32 queries; thread = (key slice, query, half); two passes (online max/sum, then
// normalise + split) so S is read twice and P written once.
__global__ __launch_bounds__(256) void md_softmax_keys_kernel(const float* __restrict__ s,
                                                              uint16_t* __restrict__ p, int nk, int nq) {
  const int b = blockIdx.y;
  const int tid = threadIdx.x, half = tid & 1, ql = (tid >> 1) & 31, ks = tid >> 6;
  const int q = blockIdx.x * 32 + ql;
  const int nkb = nk / 8;
  const f32x4* sp = (const f32x4*)(s + (int64_t)b * nk * nq);
  float m = -INFINITY, l = 0.f;
  for (int kb = ks; kb < nkb; kb += 4) {
    const f32x4 v = sp[((int64_t)kb * nq + q) * 2 + half];
    const float vm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
    const float mn = fmaxf(m, vm);
    l = l * expf(m - mn) + expf(v[0] - mn) + expf(v[1] - mn) + expf(v[2] - mn) + expf(v[3] - mn);
    m = mn;
  }
  {  // merge the two halves of a query (adjacent lanes)
    const float mo = __shfl_xor(m, 1, 64), lo = __shfl_xor(l, 1, 64);
    const float mn = fmaxf(m, mo);
    l = l * expf(m - mn) + lo * expf(mo - mn);
    m = mn;
  }
  __shared__ float sm[4][32], sl[4][32];
  if (half == 0) { sm[ks][ql] = m; sl[ks][ql] = l; }
  __syncthreads();
  float M = sm[0][ql];
#pragma unroll
  for (int k = 1; k < 4; ++k) M = fmaxf(M, sm[k][ql]);
  float L = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) L += sl[k][ql] * expf(sm[k][ql] - M);
  const int64_t plane = (int64_t)nq * 8;
  uint16_t* pb = p + (int64_t)b * nkb * 2 * plane;
  for (int kb = ks; kb < nkb; kb += 4) {
    const f32x4 v = sp[((int64_t)kb * nq + q) * 2 + half];
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) md_split(expf(v[e] - M) / L, hi[e], lo[e]);
    uint16_t* o = pb + ((int64_t)kb * 2) * plane + (int64_t)q * 8 + half * 4;
    *(uint2*)o = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    *(uint2*)(o + plane) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
  }
}

extern "C" int md_softmax_keys(const float* s, void* p, int32_t batch, int32_t n_keys, int32_t n_q,
                               void* stream) {
  if (!s || !p || batch <= 0 || n_keys <= 0 || (n_keys % 8) || n_q <= 0 || (n_q % 32)) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_softmax_keys_kernel, dim3((unsigned)(n_q / 32), (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, s, (uint16_t*)p, n_keys, n_q);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- timestep embedding entry (freq table on device, passed via `emb` scratch protocol) ------
// To keep the ABI at plain pointers the table is recomputed on the device with the same float32
// operations the reference uses: (float)j * (float)(-ln(1e4)/(half-1)) followed by expf.
__global__ void md_temb_freq_kernel(const float* __restrict__ t, float* __restrict__ emb, int batch,
                                    int dim, float neg_scale) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= batch * half) return;
  const int b = i / half, j = i % half;
  const float f = expf((float)j * neg_scale);
  const float arg = t[b] * f;
  emb[(int64_t)b * dim + j] = sinf(arg);
  emb[(int64_t)b * dim + half + j] = cosf(arg);
  if ((dim & 1) && j == 0) emb[(int64_t)b * dim + dim - 1] = 0.f;
}

extern "C" int md_timestep_embedding(const float* t, float* emb, int32_t batch, int32_t dim, void* stream) {
  if (!t || !emb || batch <= 0 || dim < 4) return MD_ERR_BAD_ARG;
  const int half = dim / 2;
  const float neg_scale = (float)(-(log(10000.0) / (double)(half - 1)));
  const int n = batch * half;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_temb_freq_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, t, emb, batch, dim, neg_scale);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

// ---- dx-folded head: out[b][co][p] = bias[co] + sum_dx y[b][co*kx + dx][p shifted by dx - kx/2 along x] -----------------
__global__ void md_fold_dx_kernel(const float* __restrict__ y, const float* __restrict__ bias, float* __restrict__ out, int B,
                                  int co, int kx, int rows_alloc, int64_t P, int W) {
  const int64_t total = (int64_t)B * co * P;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i % P;
    const int c = (int)((i / P) % co);
    const int b = (int)(i / (P * co));
    const int x = (int)(p % W);
    float acc = bias ? bias[c] : 0.f;
    for (int dx = 0; dx < kx; ++dx) {
      const int xs = x + dx - kx / 2;
      if (xs < 0 || xs >= W) continue;
      const int row = c * kx + dx;
      acc += y[(((int64_t)b * (rows_alloc / 8) + (row >> 3)) * P + (p + dx - kx / 2)) * 8 + (row & 7)];
    }
    out[i] = acc;
  }
}

extern "C" int md_fold_dx(const float* y, const float* bias, float* out, int32_t batch, int32_t co, int32_t kx, int32_t rows_alloc,
                          int32_t D, int32_t H, int32_t W, void* stream) {
  if (!y || !out || batch <= 0 || co <= 0 || (kx != 3 && kx != 5) || rows_alloc < co * kx || (rows_alloc % 8) || D <= 0 || H <= 0 ||
      W <= 0)
    return MD_ERR_BAD_ARG;
  const int64_t P = (int64_t)D * H * W, total = (int64_t)batch * co * P;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 65536) blocks = 65536;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_fold_dx_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, bias, out, batch, co, kx,
                     rows_alloc, P, W);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
