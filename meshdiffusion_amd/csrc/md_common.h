// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "meshdiffusion_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8;  // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(16))) float f32x16; // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

// hipGetLastError() is sticky per host thread: an error left behind by an unrelated earlier runtime
// call (e.g. a device probe before the context existed) must not be attributed to our launch.
#define MD_HIP_CLEAR_ERROR() (void)hipGetLastError()
#define MD_HIP_CHECK_LAUNCH()                      \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// fp32 -> bf16 bits, round to nearest even (finite inputs).
__device__ __forceinline__ uint32_t md_f2bf(float f) {
  uint32_t u = __float_as_uint(f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ float md_bf2f(uint32_t h) { return __uint_as_float(h << 16); }

// bf16x3 split: x ~= hi + lo with hi = bf16(x), lo = bf16(x - hi).
__device__ __forceinline__ void md_split(float x, uint32_t& hi, uint32_t& lo) {
  hi = md_f2bf(x);
  lo = md_f2bf(x - md_bf2f(hi));
}

// Two values at once: one v_cvt_pk_bf16_f32 (RNE) per plane, results already packed as [lo half = x0 | hi half = x1].
typedef __bf16 md_bf16x2 __attribute__((ext_vector_type(2)));
typedef float md_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void md_split2(float x0, float x1, uint32_t& hi, uint32_t& lo) {
  const md_f32x2 y = {x0, x1};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(y, md_bf16x2));
  const md_f32x2 hf = {__uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(y - hf, md_bf16x2));
}

// fp16x2 mode ("weights split, activations single"): w ~= hi + lo with hi = fp16(w), lo = fp16(w - hi);
// activations are one fp16 (saturated to the finite range).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t md_f2h(float f) {
  f = fminf(fmaxf(f, -65504.f), 65504.f);
  return (uint32_t)__half_as_ushort(__float2half_rn(f));
}
__device__ __forceinline__ float md_h2f(uint32_t h) { return __half2float(__ushort_as_half((unsigned short)h)); }
__device__ __forceinline__ void md_split_f16(float x, uint32_t& hi, uint32_t& lo) {
  hi = md_f2h(x);
  lo = md_f2h(x - md_h2f(hi));
}

// "f16f8" split (the Winograd conv's second arithmetic, conv3_wino.hip F8 path): t ~= hi + lo with hi = fp16(t) (RNE) and
// lo = t - hi (exact in fp32).  A product a*b is then  fp16(a)*fp16(b)  [one v_mfma_f32_32x32x16_f16]  +  the two cross terms
// a*b_lo + a_lo*b  computed from 4-bit-significand images  q8(a) q8(b_lo 2^11) + q8(a_lo 2^11) q8(b)  [OCP e4m3, RNE, saturated
// at +-448] as HALF of one K-concatenated v_mfma_scale_f32_32x32x64_f8f6f4 with scale 2^-11: two 32-cycle matrix-core units per
// product instead of bf16x3's three (tools/probes/f8_probe.hip pins the instruction's lane / byte / scale semantics and the
// rounding of v_cvt_pk_fp8_f32; tools/f16f8_numerics.py the error: 1.3e-5 per conv against 5.5e-6).
// 8 values -> `hi`: 8 fp16; `q`: [q8(t0..7) | q8(lo0..7 * 2^11)] (an activation item; weights swap the halves, see
// md_pack_wino_f8_item).
__device__ __forceinline__ float md_clamp448(float x) { return __builtin_amdgcn_fmed3f(x, -448.f, 448.f); }
__device__ __forceinline__ uint32_t md_f16f8_pair(float a, float b, float& la, float& lb) {      // fp16 pair (RNE) + the scaled remainders
  const _Float16 ha = (_Float16)__builtin_amdgcn_fmed3f(a, -65504.f, 65504.f), hb = (_Float16)__builtin_amdgcn_fmed3f(b, -65504.f, 65504.f);
  la = (a - (float)ha) * 2048.f;
  lb = (b - (float)hb) * 2048.f;
  return (uint32_t)__builtin_bit_cast(unsigned short, ha) | ((uint32_t)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ uint32_t md_e4m3x4(float a, float b, float c, float d) {                 // bytes a | b | c | d, low first
  int v = __builtin_amdgcn_cvt_pk_fp8_f32(md_clamp448(a), md_clamp448(b), 0, false);
  return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(md_clamp448(c), md_clamp448(d), v, true);
}
__device__ __forceinline__ void md_split_f16f8(const float (&t)[8], uint4& hi, uint32_t (&q)[2], uint32_t (&ql)[2]) {
  float l[8];
  hi.x = md_f16f8_pair(t[0], t[1], l[0], l[1]);
  hi.y = md_f16f8_pair(t[2], t[3], l[2], l[3]);
  hi.z = md_f16f8_pair(t[4], t[5], l[4], l[5]);
  hi.w = md_f16f8_pair(t[6], t[7], l[6], l[7]);
  q[0] = md_e4m3x4(t[0], t[1], t[2], t[3]);
  q[1] = md_e4m3x4(t[4], t[5], t[6], t[7]);
  ql[0] = md_e4m3x4(l[0], l[1], l[2], l[3]);
  ql[1] = md_e4m3x4(l[4], l[5], l[6], l[7]);
}

// "f16f6" split (the Winograd conv's third arithmetic, conv3_wino.hip F6 path): as f16f8, but the two cross terms are computed from MX
// block-scaled e2m3 images (OCP fp6: 2 exponent + 3 mantissa bits, max 7.5), which v_mfma_scale_f32_32x32x64_f8f6f4 multiplies at
// twice its e4m3 rate.  A K block = the 32 values a lane feeds to that MFMA = 16 channels x (value, scaled remainder); the block
// shares ONE power-of-two scale (its E8M0 byte travels in the 7th dword of the 32-byte record and is handed to the MFMA as the
// lane's scale operand).  tools/probes/f6_probe.hip pins: v_cvt_scalef32_2xpk16_fp6_f32 = RNE onto e2m3, saturating, the two
// sources INTERLEAVED (position 2i = src0[i], 2i + 1 = src1[i]); format code 2; every lane's own byte 0 scales its own block.
// hipcc (ROCm 7.2) lets that instruction's destination overlap its sources although it reads them while writing (the probe's
// first run lost two values): the wrapper marks the destination early-clobber.
typedef float md_f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t md_u32x6 __attribute__((ext_vector_type(6)));
__device__ __forceinline__ md_u32x6 md_cvt_2xpk16_fp6(md_f32x16 a, md_f32x16 b) {
  md_u32x6 r;
  asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, 1.0" : "=&v"(r) : "v"(a), "v"(b));
  return r;
}
// 16 values -> hi0 / hi1: their fp16 images (channels 0-7 / 8-15); r0 / r1: the 32-byte record [6 dwords of e2m3 codes | E8M0 byte
// + scale_bias | 0] split in two 16-byte items.  Source order of the codes: activations (t, remainder), weights -- lo_first --
// (remainder, t): position 2i of a weight block meets position 2i of an activation block, so the MFMA sums t_w rem_a + rem_w t_a.
__device__ __forceinline__ void md_split_f16f6(const float (&t)[16], bool lo_first, int scale_bias, uint4& hi0, uint4& hi1, uint4& r0,
                                               uint4& r1) {
  float l[16];
  uint32_t hw[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) hw[i] = md_f16f8_pair(t[2 * i], t[2 * i + 1], l[2 * i], l[2 * i + 1]);
  hi0 = make_uint4(hw[0], hw[1], hw[2], hw[3]);
  hi1 = make_uint4(hw[4], hw[5], hw[6], hw[7]);
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) m = fmaxf(m, fmaxf(fabsf(t[i]), fabsf(l[i])));
  // block exponent e = floor(log2 m) - 2: the scaled values lie in (-8, 8), the top sixteenth of the binade saturates at 7.5
  int eb = (int)(__float_as_uint(m) >> 23);                 // biased exponent of m >= 0
  eb = eb < 20 ? 20 : eb;                                   // an all-zero / denormal block: any scale, the codes are zero
  const float inv = __uint_as_float((uint32_t)(256 - eb) << 23);      // 2^-(eb - 127 - 2)
  md_f32x16 a, b;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = (lo_first ? l[i] : t[i]) * inv;
    b[i] = (lo_first ? t[i] : l[i]) * inv;
  }
  const md_u32x6 c = md_cvt_2xpk16_fp6(a, b);
  r0 = make_uint4(c[0], c[1], c[2], c[3]);
  r1 = make_uint4(c[4], c[5], (uint32_t)(eb - 2 + scale_bias), 0u);
}

// Counter-based dropout mask (training).  One 64-bit hash per 4 consecutive channels of one position gives four
// 16-bit uniforms; element e is kept when its field >= thr16 = round(p * 65536).  `q` = ((b * c_total + first channel
// of the quad) / 4) * P + pos identifies the quad, so the forward, the backward and md_dropout_scale regenerate the
// same mask from (seed, p) alone -- no mask tensor is stored.
__device__ __forceinline__ uint64_t md_drop_bits(uint64_t seed, uint64_t q) {
  uint64_t z = seed + (q + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool md_drop_keep(uint64_t bits, int e, uint32_t thr16) {
  return (uint32_t)((bits >> (16 * e)) & 0xFFFFu) >= thr16;
}
static inline uint32_t md_drop_thr16(float p) {
  const float t = p * 65536.0f + 0.5f;
  return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}

__device__ __forceinline__ float md_silu(float x) { return x / (1.0f + expf(-x)); }

__device__ __forceinline__ float md_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// Dynamic power-of-two lift of a training data gradient (md_absmax -> md_wino_prep_dual_f6 -> md_conv3_wino_f6_scaled): the exponent k with
// max |dy| * 2^k in [16, 32) -- after the Winograd input transform (x2) still 10 binades under the fp16 maximum, and the fp16 plane stays
// normal for elements down to 2^-19 of the largest.  Operand pass and conv derive k from the same word, so they agree bit for bit.
__device__ __forceinline__ int md_dgrad_lift_log2(uint32_t amax_bits) {
  const float amax = __uint_as_float(amax_bits);
  if (!(amax > 0.f) || !(amax < 3e38f)) return 0;
  const int k = 4 - ilogbf(amax);
  return k < -60 ? -60 : (k > 60 ? 60 : k);
}
__device__ __forceinline__ float md_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
