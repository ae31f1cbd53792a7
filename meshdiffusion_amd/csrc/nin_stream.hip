// md_nin_f32: the ResnetBlock shortcut NIN_0 (1x1x1 channel mixing, lib/diffusion/models/layers.py:573-582 used at
// :667,:688) as a barrier-free streaming kernel for the shapes that matter: Cout = 128, Cin <= 256 over a 64^3 / 32^3 /
// 128^3 grid, where the layer is purely HBM-bound (AI ~ 40 FLOP/B).
//
//   out[b][co][p] = sum_ci W[ci][co] * x[b][ci][p] + bias[co]        x = torch.cat(parts, 1), read as fp32 F32B
//
// One persistent workgroup per CU (8 waves).  The WHOLE packed weight matrix (WPK tiles of MD_CFG_G1_128: [K/32][4][2][128]
// 16-byte items, <= 128 KB for K <= 256) is copied to LDS once per workgroup and stays there; after that single barrier the
// waves never synchronise again.  A wave owns 32 positions per tile: its B operand (8 fp32 channels of one position per lane
// = 32 contiguous bytes of the F32B layout) goes HBM -> registers directly, is split into bf16 hi/lo with v_cvt_pk_bf16_f32,
// and feeds 4 row tiles x 3 MFMAs per 16-channel step; K travels in groups of 4 steps through two register sets, each refilled
// with the group two ahead (of this tile or the next) as soon as it is consumed, so 8-16 x 1 KB of loads per wave are always in flight.
// Same bf16x3 arithmetic and the same fp32 accumulation order over K as md_gemm_conv on the pre-split operand.
#include "md_common.h"

namespace {
constexpr int NS_ROWS = 128, NS_WAVES = 8, NS_THREADS = NS_WAVES * 64, NS_TILE = NS_WAVES * 32;   // 256 positions per tile

// A wave-uniform address, told to the compiler: lands in an SGPR pair so the access uses the saddr + 32-bit voffset form.
// The round trip through integers loses the address space, so the accessors below cast back to global (1) explicitly --
// a generic pointer would turn into flat_load, which also counts against lgkmcnt and serialises with the LDS reads.
__device__ __forceinline__ uint64_t ns_uniform(const void* p) {
  const uint64_t v = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
typedef uint32_t ns_u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const ns_u32x4* ns_gload_t;
typedef __attribute__((address_space(1))) f32x4* ns_gstore_t;
__device__ __forceinline__ uint4 ns_load(uint64_t base, uint32_t off) {
  return __builtin_bit_cast(uint4, *(ns_gload_t)(base + off));
}
__device__ __forceinline__ void ns_store(uint64_t base, uint32_t off, f32x4 v) { *(ns_gstore_t)(base + off) = v; }
}  // namespace

template <int KSTEPS>   // K / 16: 8 (K = 128) or 16 (K = 256)
__global__ __launch_bounds__(NS_THREADS) void md_nin_f32_kernel(const uint4* __restrict__ x1, const uint4* __restrict__ x2,
                                                                int split8, const uint4* __restrict__ wpk,
                                                                const float* __restrict__ bias, float* __restrict__ out,
                                                                int64_t P, int n_tiles, int tiles_per_sample, int c1, int c2) {
  constexpr int GS = 4;                  // 16-channel steps per register set
  constexpr int NG = KSTEPS / GS;        // groups per tile (2 or 4): group g of a tile travels through set g & 1
  __shared__ __attribute__((aligned(16))) uint4 wl[(KSTEPS / 2) * 4 * 2 * NS_ROWS];    // KSTEPS/2 tiles of 16 KB
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  for (int i = tid; i < (KSTEPS / 2) * 1024; i += NS_THREADS) wl[i] = wpk[i];
  __syncthreads();
  const unsigned char* wb = (const unsigned char*)wl;

  uint4 s0[2 * GS], s1[2 * GS];
  // Addresses are (wave-uniform 64-bit base in SGPRs) + (one loop-invariant 32-bit byte offset per lane): the lane's position
  // inside the tile and its half h of the 16-channel step.  Keeps the address VGPRs at one register instead of a pair per load.
  const uint32_t loff = (uint32_t)(((int64_t)h * P + wid * 32 + j) * 32);
  // group g of tile t -> register set: channel group 2 ks + h of the concatenated input at the lane's position of sample b
  auto issue = [&](uint4 (&st)[2 * GS], int t, int g) {
    const int b = t / tiles_per_sample;
    const int64_t p0 = (int64_t)(t - b * tiles_per_sample) * NS_TILE;
#pragma unroll
    for (int k = 0; k < GS; ++k) {
      const int ks = g * GS + k;
      const uint64_t ub = ns_uniform((2 * ks < split8) ? (const char*)x1 + (((int64_t)b * (c1 >> 3) + 2 * ks) * P + p0) * 32
                                                    : (const char*)x2 + (((int64_t)b * (c2 >> 3) + (2 * ks - split8)) * P + p0) * 32);
      st[2 * k] = ns_load(ub, loff);
      st[2 * k + 1] = ns_load(ub, loff + 16);
    }
  };
  f32x16 acc[4];
  auto step = [&](int ks, const uint4& r0, const uint4& r1) {
    uint32_t hw[4], lw[4];
    md_split2(__uint_as_float(r0.x), __uint_as_float(r0.y), hw[0], lw[0]);
    md_split2(__uint_as_float(r0.z), __uint_as_float(r0.w), hw[1], lw[1]);
    md_split2(__uint_as_float(r1.x), __uint_as_float(r1.y), hw[2], lw[2]);
    md_split2(__uint_as_float(r1.z), __uint_as_float(r1.w), hw[3], lw[3]);
    const bf16x8 bhi = __builtin_bit_cast(bf16x8, make_uint4(hw[0], hw[1], hw[2], hw[3]));
    const bf16x8 blo = __builtin_bit_cast(bf16x8, make_uint4(lw[0], lw[1], lw[2], lw[3]));
    // weight tile cc = ks / 2, group g = 2 (ks & 1) + h inside it: byte ((g*2 + plane) * 128 + row) * 16
    const unsigned char* wt = wb + (ks >> 1) * 16384 + (((2 * (ks & 1) + h) * 2) * NS_ROWS + j) * 16;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
      const bf16x8 ahi = *(const bf16x8*)(wt + rt * 32 * 16);
      const bf16x8 alo = *(const bf16x8*)(wt + (NS_ROWS + rt * 32) * 16);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bhi, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, blo, acc[rt], 0, 0, 0);
      acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, bhi, acc[rt], 0, 0, 0);
    }
    // one scheduling region per 16-channel step: without it the scheduler hoists the LDS reads of all steps (512 VGPRs)
    __builtin_amdgcn_sched_barrier(0);
  };
  auto consume = [&](uint4 (&st)[2 * GS], int g) {
#pragma unroll
    for (int k = 0; k < GS; ++k) step(g * GS + k, st[2 * k], st[2 * k + 1]);
  };

  int t = blockIdx.x;
  if (t < n_tiles) { issue(s0, t, 0); issue(s1, t, 1); }
  for (; t < n_tiles; t += gridDim.x) {
    const int tn = t + gridDim.x;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // consume group g, then refill its set with the group two ahead (this tile's g + 2, or the next tile's g + 2 - NG)
      if (g & 1) consume(s1, g); else consume(s0, g);
      const bool same = g + 2 < NG;
      if (same || tn < n_tiles) {
        if (g & 1) issue(s1, same ? t : tn, same ? g + 2 : g + 2 - NG);
        else issue(s0, same ? t : tn, same ? g + 2 : g + 2 - NG);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue: + bias, 16-byte stores into the F32B layout (4 consecutive channels of one position per lane)
    const int b = t / tiles_per_sample;
    const int64_t p0 = (int64_t)(t - b * tiles_per_sample) * NS_TILE;
    const uint32_t soff = (uint32_t)((wid * 32 + j) * 32 + 16 * h);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rt * 32 + 8 * q + 4 * h;
        const f32x4 bv = *(const f32x4*)(bias + row);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) o4[e] = acc[rt][q * 4 + e] + bv[e];
        const uint64_t ub = ns_uniform((char*)out + (((int64_t)b * (NS_ROWS / 8) + rt * 4 + q) * P + p0) * 32);
        ns_store(ub, soff, o4);
      }
  }
}

extern "C" int md_nin_f32(const float* x1, const float* x2, int32_t c1, int32_t c2, const void* wpk, const float* bias,
                          float* out, int32_t batch, int32_t cout, int64_t P, int32_t n_cu, void* stream) {
  if (!x1 || !wpk || !bias || !out || batch <= 0 || P <= 0 || c1 <= 0 || c2 < 0) return MD_ERR_BAD_ARG;
  const int K = c1 + c2;
  if (cout != NS_ROWS || (K != 128 && K != 256) || (c1 % 16) || (c2 % 16) || (P % NS_TILE) || (c2 > 0 && !x2))
    return MD_ERR_UNSUPPORTED;
  const int64_t n_tiles64 = (int64_t)batch * (P / NS_TILE);
  if (n_tiles64 > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  const int n_tiles = (int)n_tiles64;
  int blocks = n_cu > 0 ? n_cu : 256;
  if (blocks > n_tiles) blocks = n_tiles;
  MD_HIP_CLEAR_ERROR();
  if (K == 256)
    hipLaunchKernelGGL((md_nin_f32_kernel<16>), dim3((unsigned)blocks), dim3(NS_THREADS), 0, (hipStream_t)stream, (const uint4*)x1,
                       (const uint4*)x2, c1 >> 3, (const uint4*)wpk, bias, out, P, n_tiles, (int)(P / NS_TILE), c1, c2);
  else
    hipLaunchKernelGGL((md_nin_f32_kernel<8>), dim3((unsigned)blocks), dim3(NS_THREADS), 0, (hipStream_t)stream, (const uint4*)x1,
                       (const uint4*)x2, c1 >> 3, (const uint4*)wpk, bias, out, P, n_tiles, (int)(P / NS_TILE), c1, c2);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
