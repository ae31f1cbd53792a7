// Probe of ds_read_b64_tr_b16 on gfx950 (run on the GPU box: hipcc --offload-arch=gfx950 -O3 tr_probe.hip -o tr_probe && ./tr_probe)
//  1. semantics: LDS holds u16 element index; every lane supplies an address; which 4 elements does it get back?
//  2. bank behaviour: cycles per wave-instruction for candidate LDS images of an MFMA 32x32x16 operand fragment whose
//     k index is strided in memory ([k][channel] images, csrc/wgrad_wino.hip), 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));
#define LDSP(p) ((__attribute__((address_space(3))) v4i16*)(p))

__device__ __forceinline__ int pattern_addr(int pat, int l) {   // byte address of lane l
  const int g = l >> 4, i = l & 15, sub = g & 1, seg = g >> 1;
  switch (pat) {
    case 0: return l * 8;                                                        // 512 contiguous bytes
    case 1: return sub * 384 + seg * 3072 + (i >> 2) * 32 + (i & 3) * 8;          // [16ch sub-image: 8 k x 32 B, padded to 384 B]
    case 2: return sub * 256 + seg * 2048 + (i >> 2) * 32 + (i & 3) * 8;          // unpadded sub-images
    case 3: return seg * 2048 + (i >> 2) * 64 + sub * 32 + (i & 3) * 8;           // [k][32 ch] rows of 64 B
    case 4: return seg * 2048 + (i >> 2) * 256 + sub * 32 + (i & 3) * 8;          // [k][128 ch] rows of 256 B
    case 5: return seg * 2560 + (i >> 2) * 320 + sub * 32 + (i & 3) * 8;          // rows of 320 B
    case 6: return seg * 2176 + (i >> 2) * 272 + sub * 32 + (i & 3) * 8;          // rows of 272 B
    case 7: return sub * 384 + seg * (3072 + 128) + (i >> 2) * 32 + (i & 3) * 8;  // pattern 1 with the segments 128 B apart in bank space
    case 8: return sub * 128 + seg * 2048 + (i >> 2) * 32 + (i & 3) * 8;          // sub-images interleaved every 4 k: [k/4][sub][4 k][16 ch]
    default: return l * 8;
  }
}

__global__ void semantics(unsigned short* out, int pat) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int a = pattern_addr(pat, l);
  v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP((char*)lds + a));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}

template <int TR>
__global__ __launch_bounds__(256) void timing(long long* cyc, int pat, int iters, int* sink) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4 * 8192];
  for (int i = threadIdx.x; i < 4 * 8192; i += 256) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
  char* base = (char*)lds + w * 16384 + pattern_addr(pat, l);
  int acc = 0;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (TR) {
        v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDSP(base + (u & 1) * 128 + (u >> 1) * 768));
        acc ^= r[0] ^ r[3];
      } else {
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        u32x2 r = *(__attribute__((address_space(3))) u32x2*)(base + (u & 1) * 128 + (u >> 1) * 768);
        acc ^= r[0] ^ r[1];
      }
    }
  }
  const long long t1 = clock64();
  if (l == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
  if (acc == 0x12345678) sink[0] = acc;
}

int main() {
  unsigned short* d_out; long long* d_cyc; int* d_sink;
  hipMalloc(&d_out, 64 * 4 * 2); hipMalloc(&d_cyc, 1024 * 8); hipMalloc(&d_sink, 4);
  std::vector<unsigned short> h(256);
  for (int pat = 0; pat <= 1; ++pat) {
    semantics<<<1, 64>>>(d_out, pat);
    hipMemcpy(h.data(), d_out, 512, hipMemcpyDeviceToHost);
    printf("== semantics, pattern %d: lane: byte address supplied -> 4 element indices returned\n", pat);
    for (int l = 0; l < 64; ++l) {
      int g = l >> 4, i = l & 15, sub = g & 1, seg = g >> 1;
      int a = pat == 0 ? l * 8 : sub * 384 + seg * 3072 + (i >> 2) * 32 + (i & 3) * 8;
      printf("lane %2d addr %5d (elem %4d): %4d %4d %4d %4d\n", l, a, a / 2, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
  }
  const int iters = 2000;
  std::vector<long long> c(1024);
  for (int tr = 1; tr >= 0; --tr)
    for (int pat = 0; pat <= 8; ++pat) {
      for (int rep = 0; rep < 2; ++rep) {
        if (tr) timing<1><<<256, 256>>>(d_cyc, pat, iters, d_sink); else timing<0><<<256, 256>>>(d_cyc, pat, iters, d_sink);
        hipDeviceSynchronize();
      }
      hipMemcpy(c.data(), d_cyc, 1024 * 8, hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 1024; ++i) s += (double)c[i];
      printf("timing %s pattern %d: %.2f clock64 ticks per wave-instruction (4 waves per CU sharing the LDS)\n",
             tr ? "ds_read_b64_tr_b16" : "ds_read_b64       ", pat, s / 1024 / (iters * 16.0));
    }
  return 0;
}
