// md_attn_fwd: fused single-head self-attention over the D*H*W tokens of a 16^3 level (N = 4096, head dim C = 256):
//   o[c][q] = sum_key v[c][key] * softmax_key( C^-1/2 * sum_c' k[c'][key] q[c'][q] ) + b_v[c]
// QK^T, the softmax and PV run in one kernel with an ONLINE softmax, so the [B][N][N] score matrix (67 MB fp32 per
// sample plus its split-bf16 copy) never exists.  Both contractions use the bf16x3 operand split (hi*hi + hi*lo + lo*hi,
// fp32 accumulate) like every other contraction of the path.
//
// Reference: AttnBlock.forward, lib/diffusion/models/layers.py:595-608 (the two einsums :602,:606 and F.softmax :604).
//
// Mapping (MFMA D[i][j] = sum_k A[i][k] B[k][j], lane (j = lane & 31, h = lane >> 5) holds rows 8r + 4h + {0..3}):
//   S^T[key][query]: A = K (rows = keys, k = channels), B = Q (cols = queries)  -> a lane owns ONE query column and 16
//                    keys of a 32-key tile: the softmax statistics of a query live in one lane pair (j, j + 32).
//   O[c][query]    : A = V (rows = channels, k = keys),  B = P (cols = queries) -> P comes straight from the S^T
//                    accumulator registers: MFMA k-slot (h, i) of key step ks is key 16 ks + 8 (i / 4) + 4 h + i % 4, and
//                    the V fragment is read from LDS in that same key order (two 8-byte pieces), so no cross-lane traffic.
// One workgroup = 128 queries (4 waves x 32), one wave per SIMD (512 registers: 128 O accumulators in AGPRs + the wave's
// whole Q operand, 128 VGPRs, stay resident); K and V stream through LDS in 32-key tiles (32 KB each, double buffered;
// the next tile's K is prefetched into registers behind the S^T MFMAs, its V behind the softmax and the PV MFMAs, one barrier per tile).
#include "md_common.h"

namespace {
constexpr int AT_C = 256;            // head dim = channels
constexpr int AT_CG = AT_C / 8;      // 32 channel groups
constexpr int AT_TK = 32;            // keys per tile
constexpr int AT_QW = 32;            // queries per wave
constexpr int AT_WAVES = 4;
constexpr int AT_THREADS = AT_WAVES * 64;
constexpr int AT_K_ITEMS = AT_CG * 2 * AT_TK;        // 2048 uint4 (32 KB)
constexpr int AT_V_ITEMS = (AT_TK / 8) * 2 * AT_C;   // 2048 uint4 (32 KB)
constexpr int AT_PF = AT_K_ITEMS / AT_THREADS;       // 8 K items + 8 V items per thread and tile
constexpr int AT_BUF_BYTES = (AT_K_ITEMS + AT_V_ITEMS) * 16;   // 64 KB per buffer
typedef __attribute__((ext_vector_type(4))) short bf16x4;

// Round 5: the wave's Q operand (128 registers) lives in AccVGPRs and feeds the S^T MFMAs from there.  The MFMA encoding takes either
// register file for its A / B sources, but hipcc only ever allocates the accumulator there: with Q in the architectural file the kernel
// needed ~330 VGPRs, so hipcc parked half of Q in AccVGPRs itself and copied it back before every use -- 208 v_accvgpr_read out of
// ~620 instructions per 32-key tile, in a kernel that is bound by the instruction issue of its single wave per SIMD.  The S^T MFMAs are
// therefore written as inline asm with an "a" (AccVGPR) B operand; Q is pinned there once after its load.  AT_Q_AGPR=0: the round-4 form.
#ifndef AT_Q_AGPR
#define AT_Q_AGPR 1
#endif
// acc += A * B with B in AccVGPRs.  Three of these on one accumulator issue back to back like the builtin's (same opcode, SrcC = vDst:
// no wait state needed); the accumulators are read by VALU only behind at_mfma_fence().
__device__ __forceinline__ void at_mfma_bq(f32x16& acc, const bf16x8& a, const bf16x8& bq) {
#if AT_Q_AGPR
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(bq));
#else
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
#endif
}
// hipcc does not see inside the asm: the wait states between an MFMA's register write and a VALU read of it (up to 19 for a 16-pass
// MFMA; the compiler inserts them for its own MFMAs) are part of the SAME asm statement as the last MFMA of each accumulator (round 6,
// in a separate statement behind the loop nothing kept hipcc from placing a register copy of the accumulator between the
// MFMA and the nops).  Behind s0's last MFMA the 20 idle issue cycles sit under the matrix pipe's 32 per MFMA; behind s1's they are the
// wait the softmax needs anyway.
__device__ __forceinline__ void at_mfma_bq_last(f32x16& acc, const bf16x8& a, const bf16x8& bq) {
#if AT_Q_AGPR
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 3" : "+v"(acc) : "v"(a), "a"(bq));
#else
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bq, acc, 0, 0, 0);
#endif
}
// compiler-level fence only: nothing that touches the score accumulators moves across the end of the S^T block
__device__ __forceinline__ void at_mfma_fence(f32x16& s0, f32x16& s1) {
#if AT_Q_AGPR
  asm volatile("" : "+v"(s0), "+v"(s1));
#endif
}
}  // namespace

__global__ __launch_bounds__(AT_THREADS) void md_attn_fwd_kernel(const uint4* __restrict__ qk, const uint4* __restrict__ vT,
                                                                 uint16_t* __restrict__ out, const float* __restrict__ bias_v,
                                                                 int N, int batch, float scale_log2e) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * AT_BUF_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 31, h = lane >> 5;
  // workgroup w runs on XCD w % 8: give every XCD one sample (all of its query tiles stream the same K / V through that L2)
  const int b = blockIdx.x % batch;
  const int qt = blockIdx.x / batch;
  const int q0 = qt * (AT_WAVES * AT_QW) + wid * AT_QW;
  const uint4* qkb = qk + (int64_t)b * (2 * AT_CG) * 2 * N;          // [2C/8][2][N] uint4
  const uint4* vb = vT + (int64_t)b * (N / 8) * 2 * AT_C;            // [N/8][2][C] uint4

  // ---- the wave's Q operand: B fragments for all 16 channel steps, hi and lo (128 VGPRs) ----
  bf16x8 qhi[AT_C / 16], qlo[AT_C / 16];
#pragma unroll
  for (int ks = 0; ks < AT_C / 16; ++ks) {
    const int g = 2 * ks + h;
    qhi[ks] = __builtin_bit_cast(bf16x8, qkb[((int64_t)(g * 2 + 0)) * N + q0 + j]);
    qlo[ks] = __builtin_bit_cast(bf16x8, qkb[((int64_t)(g * 2 + 1)) * N + q0 + j]);
  }
#if AT_Q_AGPR
#pragma unroll
  for (int ks = 0; ks < AT_C / 16; ++ks) asm volatile("" : "+a"(qhi[ks]), "+a"(qlo[ks]));      // Q -> AccVGPRs, for good
#endif
  f32x16 oacc[AT_C / 32];
#pragma unroll
  for (int rt = 0; rt < AT_C / 32; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[rt][r] = 0.f;
  float m_run = -1e30f, l_run = 0.f;       // running max (log2 domain) and this lane's partial denominator

  // ---- tile prefetch: global -> registers -> LDS, in two halves that share ONE set of 8 registers x 4: the K items are requested at the
  // start of a tile and stored behind its S^T MFMAs, the V items are requested then and stored at the end of the tile.  With both
  // sets live (64 registers) the kernel needed 510 registers and hipcc serialised half of the requests with their stores ("global_load;
  // s_waitcnt vmcnt(0); ds_write" eight times per tile: 0.81 ms per launch; this form 0.47).
  // Addresses: one 32-bit lane offset + a wave-uniform base per item (64-bit per-lane pointers cost 32 registers).
  uint4 pf[AT_PF];
  const uint4* kpart = qkb + (int64_t)AT_CG * 2 * N;                 // k = channels C .. 2C-1 of the fused q|k tensor
  const uint32_t k_lane = (uint32_t)(tid >> 5) * (uint32_t)N + (uint32_t)(tid & 31);      // item (g*2+plane) = tid / 32 + 8 i, key = tid % 32
  auto issue_k = [&](int t) {
    const uint4* base = kpart + t * AT_TK;
#pragma unroll
    for (int i = 0; i < AT_PF; ++i) pf[i] = (base + (int64_t)i * 8 * N)[k_lane];
  };
  auto issue_v = [&](int t) {
    const uint4* base = vb + (int64_t)t * (AT_TK / 8) * 2 * AT_C;      // ((kg*2+plane) * C + c) = tid + 256 i
#pragma unroll
    for (int i = 0; i < AT_PF; ++i) pf[i] = (base + i * AT_THREADS)[tid];
  };
  auto commit_k = [&](int buf) {
    unsigned char* base = lds + buf * AT_BUF_BYTES + tid * 16;
#pragma unroll
    for (int i = 0; i < AT_PF; ++i) *(uint4*)(base + i * AT_THREADS * 16) = pf[i];
  };
  auto commit_v = [&](int buf) {
    // V: thread tid holds channel c = tid of all 8 (key group, plane) items of the tile.  The MFMA wants, per key step ks and
    // lane half h, keys 16ks + 4h + {0..3} (group 2ks, half h) followed by 16ks + 8 + 4h + {0..3} (group 2ks + 1, half h):
    // those 16 bytes are assembled here and stored as ONE item [ks][plane][h][c], so that a half-wave reads 32 consecutive
    // 16-byte items with one ds_read_b128 (the 8-byte pieces compiled to ds_read2_b64: half the LDS rate).
    static_assert(AT_PF == 8 && AT_THREADS == AT_C, "one thread = one channel of all 4 key groups x 2 planes");
    unsigned char* vbase = lds + buf * AT_BUF_BYTES + AT_K_ITEMS * 16 + tid * 16;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const uint4 g0 = pf[(2 * ks) * 2 + p], g1 = pf[(2 * ks + 1) * 2 + p];
        *(uint4*)(vbase + (((ks * 2 + p) * 2 + 0) * AT_C) * 16) = make_uint4(g0.x, g0.y, g1.x, g1.y);
        *(uint4*)(vbase + (((ks * 2 + p) * 2 + 1) * AT_C) * 16) = make_uint4(g0.z, g0.w, g1.z, g1.w);
      }
  };

  const int ntiles = N / AT_TK;
  issue_k(0);
  commit_k(0);
  issue_v(0);
  commit_v(0);
  __syncthreads();
  for (int t = 0; t < ntiles; ++t) {
    const unsigned char* kb = lds + (t & 1) * AT_BUF_BYTES;
    const unsigned char* vbuf = kb + AT_K_ITEMS * 16;
    const bool more = t + 1 < ntiles;
    if (more) issue_k(t + 1);
    // ---- S^T tile: 32 keys x 32 queries, K = 256 channels; two accumulators (even / odd channel steps) ----
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    // K fragments one channel step ahead, in two register sets, the order pinned by scheduling groups (hipcc otherwise issues every
    // ds_read_b128 directly in front of the MFMA that needs it: 0.50 -> 0.475 ms per launch)
    bf16x8 kf[2][2];
    kf[0][0] = *(const bf16x8*)(kb + (((0 + h) * 2 + 0) * AT_TK + j) * 16);
    kf[0][1] = *(const bf16x8*)(kb + (((0 + h) * 2 + 1) * AT_TK + j) * 16);
#pragma unroll
    for (int ks = 0; ks < AT_C / 16; ++ks) {
      if (ks + 1 < AT_C / 16) {
        const int g = 2 * (ks + 1) + h;
        kf[(ks + 1) & 1][0] = *(const bf16x8*)(kb + ((g * 2 + 0) * AT_TK + j) * 16);
        kf[(ks + 1) & 1][1] = *(const bf16x8*)(kb + ((g * 2 + 1) * AT_TK + j) * 16);
      }
      const bf16x8 khi = kf[ks & 1][0], klo = kf[ks & 1][1];
      const bool last = ks + 2 >= AT_C / 16;            // this accumulator's last channel step (compile-time: the loop is unrolled)
      if (ks & 1) {
        at_mfma_bq(s1, klo, qhi[ks]);
        at_mfma_bq(s1, khi, qlo[ks]);
        if (last) at_mfma_bq_last(s1, khi, qhi[ks]); else at_mfma_bq(s1, khi, qhi[ks]);
      } else {
        at_mfma_bq(s0, klo, qhi[ks]);
        at_mfma_bq(s0, khi, qlo[ks]);
        if (last) at_mfma_bq_last(s0, khi, qhi[ks]); else at_mfma_bq(s0, khi, qhi[ks]);
      }
#if !AT_Q_AGPR
      if (ks + 1 < AT_C / 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
#endif
    }
    at_mfma_fence(s0, s1);
    // the other buffer's last readers (tile t - 1) are behind the previous barrier: K of tile t + 1 goes in now, its V is requested
    if (more) { commit_k((t + 1) & 1); issue_v(t + 1); }
    // ---- online softmax of this query column (log2 domain: exp(x) = 2^(x log2 e), the 1/sqrt(C) scale folded in) ----
    float tv[16];
    float mloc = -1e30f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { tv[r] = (s0[r] + s1[r]) * scale_log2e; mloc = fmaxf(mloc, tv[r]); }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    // Lazy rescale: the reference point m_run only moves when some query's tile maximum exceeds it by more than 2^8
    // (p <= 256 is harmless in fp32 / split bf16), so the 128 accumulator registers are rescaled a few times per
    // kernel instead of once per tile.  Wave-uniform decision: every lane then applies its own factor (1 if unmoved).
    if (__builtin_amdgcn_ballot_w64(mloc > m_run + 8.0f) != 0) {
      const float m_new = fmaxf(m_run, mloc);
      const float corr = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= corr;
#if AT_Q_AGPR
      // the accumulators live in AccVGPRs; hipcc hoisted the 128 v_accvgpr_read of this rarely taken branch to the loop head
      // (every tile paid them).  An opaque redefinition inside the branch keeps the copies where they are needed.
#pragma unroll
      for (int rt = 0; rt < AT_C / 32; ++rt) asm volatile("" : "+a"(oacc[rt]));
#endif
#pragma unroll
      for (int rt = 0; rt < AT_C / 32; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[rt][r] *= corr;
    }
    float psum = 0.f;
    uint32_t phi[8], plo[8];            // packed pairs of accumulator rows (2r, 2r+1)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float p0 = __builtin_amdgcn_exp2f(tv[2 * r] - m_run), p1 = __builtin_amdgcn_exp2f(tv[2 * r + 1] - m_run);
      psum += p0 + p1;
      md_split2(p0, p1, phi[r], plo[r]);
    }
    l_run += psum;
    // P as MFMA B fragments: key step ks takes accumulator rows {2ks, 2ks+1} x 4
    bf16x8 pbh[2], pbl[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      pbh[ks] = __builtin_bit_cast(bf16x8, make_uint4(phi[4 * ks], phi[4 * ks + 1], phi[4 * ks + 2], phi[4 * ks + 3]));
      pbl[ks] = __builtin_bit_cast(bf16x8, make_uint4(plo[4 * ks], plo[4 * ks + 1], plo[4 * ks + 2], plo[4 * ks + 3]));
    }
    // ---- O += V P : 8 channel row tiles x 2 key steps x 3 MFMAs ----
#pragma unroll
    for (int rt = 0; rt < AT_C / 32; ++rt) {
      const int c = rt * 32 + j;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        // V fragment in the P key order: item [ks][plane][h][c] = keys 16ks + 4h + {0..3}, 16ks + 8 + 4h + {0..3}
        const bf16x8 vhi = *(const bf16x8*)(vbuf + (((ks * 2 + 0) * 2 + h) * AT_C + c) * 16);
        const bf16x8 vlo = *(const bf16x8*)(vbuf + (((ks * 2 + 1) * 2 + h) * AT_C + c) * 16);
        oacc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vlo, pbh[ks], oacc[rt], 0, 0, 0);
        oacc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, pbl[ks], oacc[rt], 0, 0, 0);
        oacc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vhi, pbh[ks], oacc[rt], 0, 0, 0);
      }
    }
    if (more) commit_v((t + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue: o = O / l + b_v, written as split bf16 (S16B [B][C/8][2][N][8]) ----
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
  uint16_t* ob = out + (int64_t)b * AT_CG * 2 * N * 8;
  const int64_t qpos = q0 + j;
#pragma unroll
  for (int rt = 0; rt < AT_C / 32; ++rt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rt * 32 + 8 * q + 4 * h;          // 4 consecutive channels
      uint32_t hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) md_split(oacc[rt][q * 4 + e] * inv + bias_v[row + e], hi[e], lo[e]);
      const int64_t o = (((int64_t)(row >> 3) * 2) * N + qpos) * 8 + (row & 7);
      *(uint2*)(ob + o) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
      *(uint2*)(ob + o + (int64_t)N * 8) = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    }
}

extern "C" int md_attn_fwd(const void* qk, const void* vT, void* out, const float* bias_v, int32_t batch, int32_t C,
                           int32_t N, float scale, void* stream) {
  if (!qk || !vT || !out || !bias_v || batch <= 0) return MD_ERR_BAD_ARG;
  if (C != AT_C || N <= 0 || (N % (AT_WAVES * AT_QW))) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_attn_fwd_kernel, dim3((unsigned)(batch * (N / (AT_WAVES * AT_QW)))), dim3(AT_THREADS), 0,
                     (hipStream_t)stream, (const uint4*)qk, (const uint4*)vT, (uint16_t*)out, bias_v, N, batch,
                     scale * 1.4426950408889634f);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
