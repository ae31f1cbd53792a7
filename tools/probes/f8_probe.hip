// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/probes/f8_probe.hip -o /tmp/f8_probe && /tmp/f8_probe)
//  1. semantics: OCP e4m3 operands, which (lane, byte) holds which k, what the E8M0 scale operands do, and that
//     v_cvt_pk_fp8_f32 writes OCP e4m3 with round-to-nearest-even -- everything csrc/conv3_wino_f8.hip relies on, checked
//     against a host computation in double;
//  2. throughput under the chip's power limit, one wave per SIMD, 16 accumulator tiles (256 AccVGPRs), RANDOM operand values:
//        mix A: 48 x v_mfma_f32_32x32x16_bf16                         (bf16x3: the conv kernel's step today, 3 units of 32 cycles)
//        mix B: 32 x v_mfma_f32_32x32x16_f16 + 16 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 x fp8)   (2 + 2 units covering TWO steps' worth
//               of products: fp16 hi*hi + one K-concatenated fp8 MFMA for both cross terms of two taps)
//     reported as time per loop iteration and "bf16 units" per second; mix B does the work of 2 x mix A per 64 units.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- host-side OCP e4m3fn ------------------------------------------------------------------------------------------
static double e4m3_to_double(uint8_t b) {
  const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  double v = e == 0 ? std::ldexp((double)m, -9) : std::ldexp(1.0 + m / 8.0, e - 7);
  if (e == 15 && m == 7) v = NAN;
  return s ? -v : v;
}
static uint8_t double_to_e4m3(double x) {       // round to nearest even, saturating
  uint8_t best = 0; double bd = 1e300;
  for (int b = 0; b < 256; ++b) {
    if ((b & 0x7f) == 0x7f) continue;
    const double d = std::fabs(e4m3_to_double((uint8_t)b) - x);
    if (d < bd || (d == bd && !(b & 1))) { bd = d; best = (uint8_t)b; }
  }
  return best;
}

// ---- 1. semantics ---------------------------------------------------------------------------------------------------
// A is [32][64] bytes (row i, k), B is [64][32] (k, col j) stored as Bt[j][k].  Lane l = (kb = l >> 5, i = l & 31) holds 32
// bytes; hypothesis H: byte m of lane (kb, i) is k = 32 kb + m for both operands.
__global__ void sem_kernel(const uint8_t* A, const uint8_t* Bt, float* C, int scale_a, int scale_b, const float* Cin) {
  const int l = threadIdx.x, i = l & 31, kb = l >> 5;
  i32x8 a, b;
  for (int q = 0; q < 8; ++q) {
    a[q] = *(const int*)(A + i * 64 + kb * 32 + q * 4);
    b[q] = *(const int*)(Bt + i * 64 + kb * 32 + q * 4);
  }
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = Cin ? Cin[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] : 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  // C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = c[r];
}

__global__ void cvt_kernel(const float* x, uint8_t* out, int n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * t + 1 >= n + 1) return;
  const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * t], x[2 * t + 1], 0, false);
  out[2 * t] = (uint8_t)(packed & 0xff);
  out[2 * t + 1] = (uint8_t)((packed >> 8) & 0xff);
}

// ---- 2. throughput ---------------------------------------------------------------------------------------------------
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
  // operands: 8 A + 8 B 16-byte fragments (bf16 / fp16), and for mix B 4 + 4 32-byte fp8 fragments; random bits from memory
  uint4 ra[8], rb[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    ra[q] = src[(blockIdx.x & 63) * 4096 + q * 256 + tid];
    rb[q] = src[(blockIdx.x & 63) * 4096 + 2048 + q * 256 + tid];
  }
  for (int it = 0; it < iters; ++it) {
    if constexpr (MIX == 0) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int t = 0; t < 16; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ra[(t >> 2) * 2 + (p == 0)]),
                                                           __builtin_bit_cast(bf16x8, rb[(t & 3) * 2 + (p == 1)]), acc[t], 0, 0, 0);
    } else {
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
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 127 - 11, 0, 127);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][7];
  if (s == 123.456f) sink[tid] = s;
}

int main() {
  // ---------------- semantics ----------------
  std::vector<uint8_t> A(32 * 64), Bt(32 * 64);
  srand(1);
  for (auto& v : A) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v ^= 0x10; }
  for (auto& v : Bt) { v = (uint8_t)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v ^= 0x10; }
  uint8_t *dA, *dB; float* dC;
  CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, Bt.size())); CK(hipMalloc(&dC, 32 * 32 * 4));
  CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice));
  // trials 0-2: random bytes (magnitudes over 15 binades), C = 0: the layout / scale check -- a wrong (lane, byte) -> k map or a
  //   wrong scale gives O(1) relative errors; what remains measures the width of the adder's alignment window.
  // trials 3-4: operands of ONE binade (|x| in [1, 2)), C = 0: the dot product of like-sized terms.
  // trials 5-6: the conv kernel's use: C = a running fp32 sum in [1, 2) (all 24 bits used), products scaled by 2^-11 (cross
  //   terms of a split product): is C + sum rounded like an fp32 add, or is the accumulator itself truncated?
  float* dCin; CK(hipMalloc(&dCin, 32 * 32 * 4));
  std::vector<float> Cin(32 * 32);
  for (auto& v : Cin) v = (float)(1.0 + (rand() / (double)RAND_MAX)) * ((rand() & 1) ? 1.f : -1.f);
  CK(hipMemcpy(dCin, Cin.data(), Cin.size() * 4, hipMemcpyHostToDevice));
  for (int trial = 0; trial < 7; ++trial) {
    const int sa = trial == 1 ? 127 - 11 : (trial == 2 ? 127 - 3 : (trial >= 5 ? 127 - 11 : 127)), sb = trial == 2 ? 127 + 5 : (trial == 6 ? 127 - 4 : 127);
    if (trial == 3) {
      for (auto& v : A) v = (uint8_t)((v & 0x87) | (7 << 3));      // exponent field 7: |x| in [1, 2)
      for (auto& v : Bt) v = (uint8_t)((v & 0x87) | (7 << 3));
      CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, Bt.data(), Bt.size(), hipMemcpyHostToDevice));
    }
    const bool with_c = trial >= 4;
    hipLaunchKernelGGL(sem_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, sa, sb, with_c ? dCin : (const float*)nullptr);
    std::vector<float> Cc(32 * 32);
    CK(hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, ref_norm = 0, worst_ulp = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0;
        for (int k = 0; k < 64; ++k) s += e4m3_to_double(A[i * 64 + k]) * e4m3_to_double(Bt[j * 64 + k]);
        s = std::ldexp(s, (sa - 127) + (sb - 127)) + (with_c ? (double)Cin[i * 32 + j] : 0.0);
        worst = std::fmax(worst, std::fabs(s - Cc[i * 32 + j]));
        ref_norm = std::fmax(ref_norm, std::fabs(s));
        const double ulp = std::ldexp(1.0, std::ilogb(s == 0 ? 1e-30 : s) - 23);
        worst_ulp = std::fmax(worst_ulp, std::fabs(s - Cc[i * 32 + j]) / ulp);
      }
    printf("semantics trial %d (scale_a 2^%d, scale_b 2^%d, %s, %s): max |C - ref| = %.3e (max |ref| %.3e, rel %.2e, worst %.2f fp32 ulp of the result)\n",
           trial, sa - 127, sb - 127, trial < 3 ? "random bytes" : "one binade", with_c ? "C in [1,2)" : "C = 0", worst, ref_norm, worst / ref_norm, worst_ulp);
  }
  {   // v_cvt_pk_fp8_f32 vs the host encoder
    const int n = 4096;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) {
      const double u = (rand() / (double)RAND_MAX) * 2 - 1;
      x[i] = (float)(u * std::ldexp(1.0, (rand() % 20) - 12));
      if (i < 16) x[i] = (float)((i & 1 ? -1 : 1) * (440.0 + 4.0 * i));      // around and beyond the maximum 448
    }
    float* dx; uint8_t* dq;
    CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dq, n));
    CK(hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(n / 2 / 64), dim3(64), 0, 0, dx, dq, n);
    std::vector<uint8_t> q(n);
    CK(hipMemcpy(q.data(), dq, n, hipMemcpyDeviceToHost));
    int bad = 0, sat_nan = 0;
    for (int i = 0; i < n; ++i) {
      const uint8_t want = double_to_e4m3(std::fmin(std::fmax((double)x[i], -448.0), 448.0));
      if ((q[i] & 0x7f) == 0x7f) { ++sat_nan; continue; }
      if (q[i] != want && !(e4m3_to_double(q[i]) == 0 && e4m3_to_double(want) == 0)) {
        if (bad < 8) printf("  cvt mismatch x = %.9g: device 0x%02x (%.6g) host 0x%02x (%.6g)\n", x[i], q[i], e4m3_to_double(q[i]), want,
                            e4m3_to_double(want));
        ++bad;
      }
    }
    printf("v_cvt_pk_fp8_f32: %d of %d differ from host RNE e4m3; %d results are NaN encodings (|x| > 448 without a clamp: ", bad, n, sat_nan);
    for (int i = 0; i < 6; ++i) printf("%.0f->0x%02x ", x[i], q[i]);
    printf(")\n");
  }
  // ---------------- throughput ----------------
  const size_t nsrc = 64 * 4096;
  std::vector<uint32_t> h(nsrc * 4);
  for (auto& v : h) {
    // random finite patterns in every view of the same bits: bit 6 of every byte cleared = bf16 / fp16 exponent MSB cleared
    // (|x| < 2) and no e4m3 NaN code (0x7f / 0xff need bit 6)
    v = (((uint32_t)rand() << 16) ^ (uint32_t)rand()) & 0xBFBFBFBFu;
  }
  uint4* dsrc; float* dsink;
  CK(hipMalloc(&dsrc, nsrc * 16)); CK(hipMalloc(&dsink, 1024));
  CK(hipMemcpy(dsrc, h.data(), nsrc * 16, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000, grid = 256 * 4;
  for (int rep = 0; rep < 2; ++rep) {
    for (int mix = 0; mix < 2; ++mix) {
      for (int w = 0; w < 2; ++w) {      // the second launch is the timed one (same clocks / caches)
        CK(hipEventRecord(e0));
        if (mix == 0) hipLaunchKernelGGL(thr_kernel<0>, dim3(grid), dim3(256), 0, 0, dsrc, dsink, iters);
        else hipLaunchKernelGGL(thr_kernel<1>, dim3(grid), dim3(256), 0, 0, dsrc, dsink, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
      }
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double units = (mix == 0 ? 48.0 : 64.0) * iters * 4.0 * grid;       // 32-cycle bf16-MFMA equivalents
      const double per_unit_flop = 2.0 * 32 * 32 * 16;
      const double per_simd = (mix == 0 ? 48.0 : 64.0) * iters * (grid / 256.0);   // units one SIMD executes (one wave at a time)
      printf("rep %d mix %c: %.3f ms  %.1f TF/s issued (bf16-equivalent)  %.2f ns per 32-cycle unit per SIMD (13.33 at 2.4 GHz)\n", rep,
             mix ? 'B' : 'A', ms, units * per_unit_flop / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd);
    }
  }
  printf("mix A = 48 bf16 MFMAs (3 units per product-step); mix B = 32 fp16 + 16 scaled fp8 K=64 (4 units per TWO product-steps)\n");
  return 0;
}
