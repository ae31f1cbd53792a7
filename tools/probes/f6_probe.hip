// Probe of the MX 6-bit path of gfx950 (run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/f6_probe.hip -o /tmp/f6_probe && /tmp/f6_probe)
// What the "f16f6" Winograd conv (csrc/conv3_wino.hip, FMT = 2) relies on, each checked against a host computation in double:
//  1. v_cvt_scalef32_2xpk16_fp6_f32 (32 floats -> 32 e2m3 codes in 6 registers): rounding, saturation, what the scale operand does, and
//     -- through v_cvt_scalef32_pk32_f32_fp6, the decoder -- which of the 32 positions each of the two 16-float sources lands on;
//  2. v_mfma_scale_f32_32x32x64_f8f6f4 with both operands in e2m3 (cbsz = blgp = 2): a lane's 6 registers are the 32 k of its half
//     (lanes 0-31: k 0..31, lanes 32-63: k 32..63), and EVERY LANE'S OWN scale byte (E8M0, byte 0 of the scale register) scales its
//     own 32-value block: A block (row lane & 31, k half lane >> 5), B block (column lane & 31, k half lane >> 5);
//  3. throughput under the power limit, one wave per SIMD, 16 accumulator tiles, random operands:
//        mix B: 32 x v_mfma_f32_32x32x16_f16 + 16 x scaled fp8 K = 64      (the f16f8 pair-step: 4 units of 32 cycles)
//        mix C: 32 x v_mfma_f32_32x32x16_f16 + 16 x scaled e2m3 K = 64     (f16f6: 3 units if the 6-bit form runs at twice the fp8 rate)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// hipcc (ROCm 7.2) lets the destination of v_cvt_scalef32_2xpk16_fp6_f32 overlap its sources (first run of this probe: dst v[32:37] on
// src1 v[18:33] -- the last two values of source 1 came out as +-7.5); the instruction reads its sources while it writes: early clobber.
__device__ __forceinline__ u32x6 cvt_2xpk16_fp6(f32x16 a, f32x16 b, float scale) {
  u32x6 r;
  asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(scale));
  return r;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static double e2m3_round(double x) {      // nearest (ties to even) representable e2m3 value, saturating at +-7.5
  const double ax = std::fabs(x);
  double step = 0.125;
  if (ax >= 4) step = 0.5; else if (ax >= 2) step = 0.25;
  double q = std::nearbyint(ax / step) * step;      // default rounding mode: ties to even
  if (q > 7.5) q = 7.5;
  return x < 0 ? -q : q;
}

// ---- 1. encode / decode ----------------------------------------------------------------------------------------------
__global__ void enc_dec_kernel(const float* x, float* y, unsigned* codes, float enc_scale, float dec_scale) {
  const int l = threadIdx.x;
  f32x16 s0, s1;
  for (int i = 0; i < 16; ++i) { s0[i] = x[l * 32 + i]; s1[i] = x[l * 32 + 16 + i]; }
  const u32x6 c = cvt_2xpk16_fp6(s0, s1, enc_scale);
  for (int q = 0; q < 6; ++q) codes[l * 6 + q] = c[q];
  const f32x32 d = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(c, dec_scale);
  for (int i = 0; i < 32; ++i) y[l * 32 + i] = d[i];
}

// ---- 2. MFMA semantics ------------------------------------------------------------------------------------------------
// A[32][64], Bt[32][64] floats on the e2m3 grid; lane (kb, i) encodes k = 32 kb + m: m < 16 -> source 0, m >= 16 -> source 1.
__global__ void sem_kernel(const float* A, const float* Bt, const int* sa, const int* sb, float* C, int fmt) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  f32x16 a0, a1, b0, b1;
  for (int m = 0; m < 16; ++m) {
    a0[m] = A[i * 64 + kb * 32 + m]; a1[m] = A[i * 64 + kb * 32 + 16 + m];
    b0[m] = Bt[i * 64 + kb * 32 + m]; b1[m] = Bt[i * 64 + kb * 32 + 16 + m];
  }
  const u32x6 ca = cvt_2xpk16_fp6(a0, a1, 1.0f);
  const u32x6 cb = cvt_2xpk16_fp6(b0, b1, 1.0f);
  i32x8 a, b;
  for (int q = 0; q < 6; ++q) { a[q] = (int)ca[q]; b[q] = (int)cb[q]; }
  a[6] = sa[l]; a[7] = 0x55555555; b[6] = sb[l]; b[7] = 0x2a2a2a2a;      // the record layout of the conv kernel: scale in element 6, junk in 7
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  if (fmt == 2) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, a[6], 0, b[6]);
  else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 3, 3, 0, a[6], 0, b[6]);
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = c[r];
}

// ---- 3. throughput ----------------------------------------------------------------------------------------------------
template <int MIX>
__global__ __launch_bounds__(256) void thr_kernel(const uint4* src, float* sink, int iters) {
  const int tid = threadIdx.x;
  f32x16 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    asm volatile("" : "+a"(acc[t]));
  }
  uint4 ra[8], rb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    ra[q] = src[(blockIdx.x & 63) * 4096 + q * 256 + tid];
    rb[q] = src[(blockIdx.x & 63) * 4096 + 2048 + q * 256 + tid];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int t = 0; t < 16; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ra[(t >> 2) * 2 + p]),
                                                        __builtin_bit_cast(f16x8, rb[(t & 3) * 2 + p]), acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      i32x8 a8, b8;
      const uint4 a0 = ra[(t >> 2) * 2], a1 = ra[(t >> 2) * 2 + 1], b0 = rb[(t & 3) * 2], b1 = rb[(t & 3) * 2 + 1];
      a8[0] = a0.x; a8[1] = a0.y; a8[2] = a0.z; a8[3] = a0.w; a8[4] = a1.x; a8[5] = a1.y; a8[6] = a1.z; a8[7] = a1.w;
      b8[0] = b0.x; b8[1] = b0.y; b8[2] = b0.z; b8[3] = b0.w; b8[4] = b1.x; b8[5] = b1.y; b8[6] = b1.z; b8[7] = b1.w;
      if constexpr (MIX == 1) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 127 - 11, 0, 127);
      else acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 2, 2, 0, 127 - 11, 0, 127);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][7];
  if (s == 123.456f) sink[tid] = s;
}

int main() {
  srand(3);
  // ---------------- 1. encode / decode ----------------
  {
    std::vector<float> x(64 * 32), y(64 * 32);
    // lane 0: source 0 = 0.125 (i + 1) (exact, distinct), source 1 = -(0.125 (i + 1)): the decoder's positions tell the order
    for (int i = 0; i < 16; ++i) { x[i] = 0.125f * (i + 1); x[16 + i] = -0.125f * (i + 1); }
    // lane 1: rounding / saturation cases
    const float cases[32] = {0.0624f, 0.0626f, 0.1875f, 0.3125f, 0.9375f, 1.0625f, 1.1875f, 1.9375f, 2.125f, 2.375f, 3.875f, 4.25f, 4.75f, 7.25f,
                             7.5f, 7.74f, 7.76f, 8.0f, 12.f, 100.f, -7.76f, -100.f, 1e-9f, -1e-9f, 0.f, -0.f, 3.0f, 5.0f, 6.0f, 0.5f, 0.75f, 1.5f};
    for (int i = 0; i < 32; ++i) x[32 + i] = cases[i];
    for (int l = 2; l < 64; ++l)
      for (int i = 0; i < 32; ++i) x[l * 32 + i] = (float)((rand() / (double)RAND_MAX * 2 - 1) * 7.4);
    float *dx, *dy; unsigned* dc;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dy, y.size() * 4)); CK(hipMalloc(&dc, 64 * 6 * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(enc_dec_kernel, dim3(1), dim3(64), 0, 0, dx, dy, dc, 1.0f, 1.0f);
    CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned> codes(64 * 6);
    CK(hipMemcpy(codes.data(), dc, codes.size() * 4, hipMemcpyDeviceToHost));
    printf("encode(2xpk16, scale 1) -> decode(pk32, scale 1), lane 0 (sources 0.125 (i+1) | -0.125 (i+1)):\n  ");
    for (int i = 0; i < 32; ++i) printf("%g ", y[i]);
    printf("\n  codes: %08x %08x %08x %08x %08x %08x\n", codes[0], codes[1], codes[2], codes[3], codes[4], codes[5]);
    bool concat = true, inter = true;
    for (int i = 0; i < 16; ++i) {
      concat &= (y[i] == x[i] && y[16 + i] == x[16 + i]);
      inter &= (y[2 * i] == x[i] && y[2 * i + 1] == x[16 + i]);
    }
    printf("  order: %s\n", concat ? "CONCATENATED (decoded[i] = src0[i], decoded[16 + i] = src1[i])" : (inter ? "INTERLEAVED (decoded[2i] = src0[i], decoded[2i+1] = src1[i])" : "OTHER"));
    printf("  rounding / saturation (lane 1): ");
    for (int i = 0; i < 32; ++i) printf("%g->%g ", x[32 + i], y[32 + (concat ? i : (inter ? (i < 16 ? 2 * i : 2 * (i - 16) + 1) : i))]);
    printf("\n");
    int bad = 0;
    for (int l = 2; l < 64; ++l)
      for (int i = 0; i < 32; ++i) {
        const int pos = concat ? i : (inter ? (i < 16 ? 2 * i : 2 * (i - 16) + 1) : i);
        if (y[l * 32 + pos] != (float)e2m3_round(x[l * 32 + i])) {
          if (bad < 6) printf("  mismatch x = %.6f: device %.4f host %.4f\n", x[l * 32 + i], y[l * 32 + pos], e2m3_round(x[l * 32 + i]));
          ++bad;
        }
      }
    printf("  random values in (-7.4, 7.4): %d of %d differ from host RNE e2m3\n", bad, 62 * 32);
    for (int t = 0; t < 2; ++t) {      // what the scale operands do
      const float es = t == 0 ? 4.0f : 1.0f, ds = t == 0 ? 1.0f : 4.0f;
      hipLaunchKernelGGL(enc_dec_kernel, dim3(1), dim3(64), 0, 0, dx, dy, dc, es, ds);
      CK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
      printf("  encode scale %g, decode scale %g: lane 0 decodes to %g %g %g %g ... (inputs 0.125 0.25 0.375 0.5)\n", es, ds, y[0], y[concat ? 1 : 2], y[concat ? 2 : 4],
             y[concat ? 3 : 6]);
    }
  }
  // ---------------- 2. MFMA semantics ----------------
  {
    std::vector<float> A(32 * 64), Bt(32 * 64);
    const double grid[8] = {0.125, 0.5, 0.875, 1.0, 1.75, 2.5, 3.0, 6.0};
    for (auto& v : A) v = (float)(grid[rand() & 7] * ((rand() & 1) ? 1 : -1));
    for (auto& v : Bt) v = (float)(grid[rand() & 7] * ((rand() & 1) ? 1 : -1));
    std::vector<int> sa(64), sb(64);
    for (int l = 0; l < 64; ++l) { sa[l] = 127 - 8 + (rand() % 16) + ((rand() & 0xffff) << 8); sb[l] = 127 - 8 + (rand() % 16) + ((rand() & 0xffff) << 8); }   // junk above byte 0
    float *dA, *dB, *dC; int *dsa, *dsb;
    CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, Bt.size() * 4)); CK(hipMalloc(&dC, 32 * 32 * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    for (int fmt = 2; fmt <= 3; ++fmt) {
      hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC, fmt);
      std::vector<float> C(32 * 32);
      CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
      // hypotheses for the scale of block (row / column x, half kb): H0 own lane (kb * 32 + x); H1 lane x (half 0) for both halves; H2 none
      for (int hyp = 0; hyp < 3; ++hyp) {
        double worst = 0, ref = 0;
        for (int i = 0; i < 32; ++i)
          for (int j = 0; j < 32; ++j) {
            double s = 0;
            for (int kb = 0; kb < 2; ++kb) {
              double blk = 0;
              for (int m = 0; m < 32; ++m) blk += (double)A[i * 64 + kb * 32 + m] * (double)Bt[j * 64 + kb * 32 + m];
              const int ea = hyp == 0 ? (sa[kb * 32 + i] & 0xff) - 127 : (hyp == 1 ? (sa[i] & 0xff) - 127 : 0);
              const int eb = hyp == 0 ? (sb[kb * 32 + j] & 0xff) - 127 : (hyp == 1 ? (sb[j] & 0xff) - 127 : 0);
              s += std::ldexp(blk, ea + eb);
            }
            worst = std::fmax(worst, std::fabs(s - C[i * 32 + j]));
            ref = std::fmax(ref, std::fabs(s));
          }
        printf("MFMA e2m3 x e2m3, format code %d, scale hypothesis %d (%s): max |C - ref| / max |ref| = %.3e\n", fmt, hyp,
               hyp == 0 ? "every lane's own byte 0" : (hyp == 1 ? "the half-0 lane's byte for both halves" : "scales ignored"), worst / ref);
      }
    }
  }
  // ---------------- 3. throughput ----------------
  const size_t nsrc = 64 * 4096;
  std::vector<uint32_t> h(nsrc * 4);
  for (auto& v : h) v = (((uint32_t)rand() << 16) ^ (uint32_t)rand()) & 0xBFBFBFBFu;
  uint4* dsrc; float* dsink;
  CK(hipMalloc(&dsrc, nsrc * 16)); CK(hipMalloc(&dsink, 1024));
  CK(hipMemcpy(dsrc, h.data(), nsrc * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000, grid = 256 * 4;
  for (int rep = 0; rep < 2; ++rep)
    for (int mix = 1; mix <= 2; ++mix) {
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0));
        if (mix == 1) hipLaunchKernelGGL(thr_kernel<1>, dim3(grid), dim3(256), 0, 0, dsrc, dsink, iters);
        else hipLaunchKernelGGL(thr_kernel<2>, dim3(grid), dim3(256), 0, 0, dsrc, dsink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("rep %d mix %c: %.3f ms per %d pair-steps x 16 tiles (%.2f ns per pair-step-tile per SIMD; 53.3 = four, 40.0 = three 32-cycle units at 2.4 GHz)\n", rep,
             mix == 1 ? 'B' : 'C', ms, iters, ms * 1e6 / ((double)iters * 16 * (grid / 256.0)));
    }
  printf("mix B = 32 fp16 + 16 scaled fp8 K=64 (f16f8); mix C = 32 fp16 + 16 scaled e2m3 K=64 (f16f6)\n");
  return 0;
}
