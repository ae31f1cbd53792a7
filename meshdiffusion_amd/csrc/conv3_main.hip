// md_conv3_main_kernel: the hot 3x3x3 stride-1 convolution (75% of a res64 U-Net evaluation),
// specialised from md_gemm_conv_kernel<128,32,4,8,8,27,1,2,4,SW=1> with the index arithmetic
// removed from the steady state.
//
// Ablations on MI355X (tools/bench_conv.py, DESIGN.md "Kernel notes") showed that the generic
// kernel was bound neither by LDS bank conflicts, nor by exposed LDS latency, nor by the halo
// bubble, but by instruction issue: ~250 integer/branch instructions per tap per wavefront next to
// 24 MFMAs, executed in lock step by both wavefronts of a SIMD.  Here:
//   * the 27-tap loop is fully unrolled, so every LDS address is `one VGPR + immediate`
//     (tap, K-half, hi/lo plane, row tile, column tile are all compile-time offsets);
//   * weight tiles advance by a scalar pointer bump per tap (WPK tiles are contiguous in step order);
//   * halo prefetch addresses / validity are computed once per workgroup, not per chunk;
//   * the epilogue batches its bias / residual loads instead of waiting on each.
// Same data layouts, same MFMA order per output element => bit-identical to the generic kernel.
//
// Reference op: nn.Conv3d 3x3x3 pad 1 (lib/diffusion/models/layers.py:118-124), optionally on the
// nearest-x2 upsampled input (layers.py:618-623, `ups`).
#include "md_common.h"

#ifndef MD_SETPRIO_VALUE
#define MD_SETPRIO_VALUE 0
#endif
namespace {
constexpr bool MD_SETPRIO = MD_SETPRIO_VALUE != 0;   // s_setprio 1/0 around each MFMA group: measured neutral (-DMD_SETPRIO_VALUE=1 to A/B)
constexpr int NT = 128, KC = 32, TZ = 4, TY = 8, TX = 8, TAPS = 27;
constexpr int ZH = 6, YH = 10, XH = 10;
constexpr int HS = 3 * YH * 24;          // 720 halo slots (odd z-planes interleaved at +12, y stride 24)
constexpr int HPOS = ZH * YH * XH;       // 600 valid halo positions
constexpr int KG = KC / 8;               // 4
constexpr int W_ITEMS = KG * 2 * NT;     // 1024 uint4 per weight tile (16 KiB)
constexpr int A_ITEMS = KG * 2 * HPOS;   // 4800 uint4 per halo tile
constexpr int NTHREADS = 512;
constexpr int A_PER_THREAD = (A_ITEMS + NTHREADS - 1) / NTHREADS;  // 10
constexpr int W_LDS_BYTES = 2 * W_ITEMS * 16;                     // 32 KiB (double buffer)
constexpr int LDS_ITEMS = 2 * W_ITEMS + KG * 2 * HS;              // 124928 B
constexpr int PF_TAP = 20;               // tap at which the next chunk's halo loads are issued
constexpr int PF_TAP_BF = 15;            // ... in the fused-operand mode: 10 taps of transform work follow (taps 16-25)

__device__ __forceinline__ int slot_of(int hz, int hy, int hx) {
  return (hz >> 1) * (YH * 24) + hy * 24 + (hz & 1) * 12 + hx;
}
}  // namespace

// ABL: timing-only ablations (results invalid): 1 = no LDS fragment reads, 3 = no barriers / weight commits,
// 4 = no weight global loads, 6 = no barriers only; 7 = (valid results) no MFMA/DS interleave hint
// PREC: MD_PREC_BF16X3 (both operands split bf16, 3 MFMAs/product) or MD_PREC_FP16X2 (weights split fp16,
// activations one fp16 plane, 2 MFMAs/product; the halo tile then has one plane and half the LDS/L2 traffic).
// A.stagger > 0 phase-staggers the first workgroup of every CU by up to that many shader cycles (all 256 CUs otherwise
// reach their epilogues -- 256 KB of residual reads + output stores per workgroup -- at the same moment, tile after tile:
// an HBM burst during which no matrix core works).
// VAR (A/B switch, valid results): bit 1 = commit the next tap's weight tile to LDS at the TOP of the tap (behind it 12
// MFMAs cover the ds_write latency) instead of right before the barrier.
// BF = 1 (MD_B_F32B_GN): the B operand is read as fp32 (F32B, up to two channel-concatenated parts) and GroupNorm affine
// + SiLU + the bf16 hi/lo split are applied while the halo tile is staged: one thread = one 8-channel group (kg = tid / 128,
// wave-uniform) x 5 halo positions; the transform runs half an item per tap at taps 16-25 of the chunk before, between MFMAs
// (measured alternatives: a whole item per tap over 5 taps = same cost; the two waves of a SIMD taking turns, waves 0-3 at
// taps 16-20 and waves 4-7 at taps 21-25 = 2 % slower).
template <int ABL, int PREC, int VAR = 0, int BF = 0>
__global__ __launch_bounds__(NTHREADS) void md_conv3_main_kernel(const MdGemmConvArgs A) {
  constexpr int PL = (PREC == MD_PREC_FP16X2) ? 1 : 2;        // activation planes staged in LDS
  static_assert(BF == 0 || PL == 2, "the fused operand transform produces the bf16x3 format");
  constexpr int A_ITEMS_P = KG * PL * HPOS;
  constexpr int BF_IT = (HPOS + 127) / 128;                    // 5 (kg, position) items per thread in BF mode
  constexpr int A_PT = BF ? 2 * BF_IT : (A_ITEMS_P + NTHREADS - 1) / NTHREADS;  // uint4 registers of the halo prefetch (10 / 5)
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * W_ITEMS + KG * PL * HS];
  unsigned char* lds = (unsigned char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform => SALU
  const int wr = wid >> 2, wc = wid & 3;
  const int j = lane & 31, h = lane >> 5;

  {
    // One workgroup per CU (122 KB of LDS): the first 256 workgroups of the grid land on the 256 CUs, and every later one
    // starts when its CU's predecessor retires, so a start-up delay of the first generation persists as that CU's phase.
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    if (A.stagger > 0 && lin < 256u) {
      const unsigned ph = (lin * 0x9E3779B1u) >> 24;                       // 0..255, decorrelated from the dispatch order
      const uint64_t t_end = __builtin_amdgcn_s_memtime() + ((uint64_t)ph * (uint64_t)A.stagger >> 8);
      while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(16);
    }
  }

  // ---- tile coordinates (scalar) ----------------------------------------------------------------
  const int D = A.D, H = A.H, W = A.W;
  const int64_t P = (int64_t)D * H * W;
  const int ntx = W / TX, nty = H / TY, ntz = D / TZ;
  const int tiles = ntx * nty * ntz;
  int Di = D, Hi = H, Wi = W;
  if (A.ups) { Di = D >> 1; Hi = H >> 1; Wi = W >> 1; }
  int bid = blockIdx.x;  // XCD-aware order: one contiguous run of tiles per XCD (block b runs on XCD b%8)
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles;
  const int t = bid % tiles;
  const int x0 = (t % ntx) * TX, y0 = ((t / ntx) % nty) * TY, z0 = (t / (ntx * nty)) * TZ;
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  const int rt = blockIdx.y;
  const int ncc_total = A.kdim / KC;
  const int ksplit = A.ksplit > 1 ? A.ksplit : 1;  // split-K: this workgroup owns chunks [cc_lo, cc_lo+ncc)
  const int cc_lo = (int)(((int64_t)blockIdx.z * ncc_total) / ksplit);
  const int ncc = (int)(((int64_t)(blockIdx.z + 1) * ncc_total) / ksplit) - cc_lo;
  const int nsteps = ncc * TAPS;

  const uint4* bptr = (const uint4*)A.b + (int64_t)b * (A.b_bstride / 8) + (int64_t)cc_lo * (KG * 2) * Pin;
  // BF: fp32 parts, 4 floats per uint4; channel group g8 (8 channels) of part p starts at p_base + g8 * Pin * 2 uint4
  const int bf_kg = __builtin_amdgcn_readfirstlane(tid >> 7);
  const uint4* bf_p1 = (const uint4*)A.b + (int64_t)b * (A.b_bstride / 4);
  const uint4* bf_p2 = (const uint4*)A.b2 + (int64_t)b * (A.b2_bstride / 4);
  const int bf_split8 = A.b_split >> 3;
  // tiles are contiguous in step order
  const uint4* wbase = (const uint4*)A.a + ((int64_t)rt * ncc_total + cc_lo) * TAPS * W_ITEMS;

  // ---- halo prefetch descriptors: computed ONCE (source offset in uint4 units relative to the chunk
  //      base, -1 = outside the grid => zero fill; LDS destination byte offset) ---------------------
  constexpr int N_DESC = BF ? BF_IT : A_PT;
  int hsrc[N_DESC], hdst[N_DESC];
#pragma unroll
  for (int i = 0; i < N_DESC; ++i) {
    const int item = BF ? (bf_kg * 2 * HPOS + (tid & 127) + i * 128) : (tid + i * NTHREADS);
    hsrc[i] = -1; hdst[i] = -1;
    if (BF ? ((tid & 127) + i * 128 < HPOS) : (item < A_ITEMS_P)) {
      const int gl = item / HPOS, r = item % HPOS;      // LDS plane index: (g*PL + part); BF: the hi plane of group bf_kg
      const int gp = (PL == 2) ? gl : gl * 2;            // global plane index: (g*2 + part); fp16x2 reads hi only
      const int hx = r % XH, hy = (r / XH) % YH, hz = r / (XH * YH);
      int uz = z0 + hz - 1, uy = y0 + hy - 1, ux = x0 + hx - 1;
      bool inb;
      if (A.ups) {
        inb = (uz >= 0) & (uz < D) & (uy >= 0) & (uy < H) & (ux >= 0) & (ux < W);
        uz >>= 1; uy >>= 1; ux >>= 1;
      } else {
        inb = (uz >= 0) & (uz < Di) & (uy >= 0) & (uy < Hi) & (ux >= 0) & (ux < Wi);
      }
      hdst[i] = W_LDS_BYTES + (gl * HS + slot_of(hz, hy, hx)) * 16;
      if (inb) hsrc[i] = BF ? (int)((((int64_t)uz * Hi + uy) * Wi + ux) * 2) : (int)(gp * Pin + ((int64_t)uz * Hi + uy) * Wi + ux);
    }
  }
  uint4 hreg[A_PT];
  f32x4 bf_ac[BF ? 4 : 1];    // BF: (a, c) of this thread's 8 channels for the chunk being prefetched: a0 c0 a1 c1 | ...
  auto act_issue = [&](int cc) {
    if constexpr (BF) {
      const int g8 = (cc_lo + cc) * KG + bf_kg;                       // 8-channel group in the concatenated input (scalar)
      const uint4* cb = (g8 < bf_split8) ? bf_p1 + (int64_t)g8 * Pin * 2 : bf_p2 + (int64_t)(g8 - bf_split8) * Pin * 2;
#pragma unroll
      for (int i = 0; i < BF_IT; ++i) {
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
        if (hsrc[i] >= 0) { v0 = cb[hsrc[i]]; v1 = cb[hsrc[i] + 1]; }
        hreg[2 * i] = v0; hreg[2 * i + 1] = v1;
      }
      if (A.b_ac != nullptr) {
        const f32x4* ap = (const f32x4*)(A.b_ac + ((int64_t)b * A.kdim + (int64_t)g8 * 8) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) bf_ac[q] = ap[q];
      }
    } else {
      const uint4* cb = bptr + (int64_t)cc * (KG * 2) * Pin;  // scalar chunk base
#pragma unroll
      for (int i = 0; i < A_PT; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (hsrc[i] >= 0) v = cb[hsrc[i]];
        hreg[i] = v;
      }
    }
  };
  // BF: fp32 x 8 channels of item i -> (hi plane uint4, lo plane uint4) in place.  y = x*a + c, SiLU = y / (1 + 2^(-y log2 e))
  // with the hardware exp2 / rcp (1 ulp each), hi = bf16(y) (v_cvt_pk_bf16_f32, RNE), lo = bf16(y - hi).
  // One call handles HALF an item (4 channels = one fp32 uint4 -> (hi pair, lo pair) in place: .xy = hi, .zw = lo), so the
  // VALU work of a chunk is spread thinly over 10 taps (about 1.5 VALU instructions per MFMA and wave).
  auto act_transform = [&](int hidx) {
    if constexpr (BF) {
      const int i = hidx >> 1;
      const uint4 r = hreg[hidx];
      const float v[4] = {__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
      float yv[4];
      const bool live = hsrc[i] >= 0;                     // outside the grid the ACTIVATED tensor is zero padded
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ce = (hidx & 1) * 4 + e;                // channel within the group of 8
        float y = v[e];
        if (A.b_ac != nullptr) {
          y = y * bf_ac[ce >> 1][(ce & 1) * 2] + bf_ac[ce >> 1][(ce & 1) * 2 + 1];
          if (A.b_silu) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
        }
        yv[e] = live ? y : 0.f;
      }
      uint32_t h01, l01, h23, l23;                       // packed pairs: one v_cvt_pk_bf16_f32 per plane and pair
      md_split2(yv[0], yv[1], h01, l01);
      md_split2(yv[2], yv[3], h23, l23);
      hreg[hidx] = make_uint4(h01, h23, l01, l23);
    }
  };
  auto act_commit = [&]() {
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < BF_IT; ++i)
        if (hdst[i] >= 0) {
          const uint4 r0 = hreg[2 * i], r1 = hreg[2 * i + 1];
          *(uint4*)(lds + hdst[i]) = make_uint4(r0.x, r0.y, r1.x, r1.y);
          *(uint4*)(lds + hdst[i] + HS * 16) = make_uint4(r0.z, r0.w, r1.z, r1.w);
        }
    } else {
#pragma unroll
      for (int i = 0; i < A_PT; ++i)
        if (hdst[i] >= 0) *(uint4*)(lds + hdst[i]) = hreg[i];
    }
  };

  // ---- fragment addresses: one VGPR each, everything else is an immediate ------------------------
  // weights  [KG][2][NT][8]: byte = ((ks*2+h)*2+part)*NT*16 + (wr*64 + rm*32 + j)*16 (+ buffer)
  const int vA = (h * 2 * NT + wr * 64 + j) * 16;
  // halo     [KG][2][HS][8]: byte = W_LDS_BYTES + ((ks*2+h)*2+part)*HS*16 + slot*16,
  //          slot = zterm(z+dz) + (y+dy)*24 + (x+dx),  y = cm*4 + (j>>3), x = j&7, z = wc
  const int laneB = W_LDS_BYTES + (h * PL * HS + (j >> 3) * 24 + (j & 7)) * 16;
  int vB[3];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    const int zz = wc + dz;
    vB[dz] = laneB + ((zz >> 1) * (YH * 24) + (zz & 1) * 12) * 16;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int cm = 0; cm < 2; ++cm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rm][cm][r] = 0.f;
  // Pin the accumulators to the AccVGPR half of the register file: MFMA C/D traffic (4 KiB read + 4 KiB
  // written per instruction) then stays off the architectural-VGPR ports that the LDS returns use.
  if constexpr (ABL != 8) {
#pragma unroll
    for (int rm = 0; rm < 2; ++rm)
#pragma unroll
      for (int cm = 0; cm < 2; ++cm) asm volatile("" : "+a"(acc[rm][cm]));
  }

  // ---- prologue -------------------------------------------------------------------------------------
  uint4 wreg0, wreg1;
  {
    wreg0 = wbase[tid]; wreg1 = wbase[tid + NTHREADS];
    act_issue(0);
    if constexpr (BF) {
#pragma unroll
      for (int i = 0; i < 2 * BF_IT; ++i) act_transform(i);
    }
    act_commit();
    *(uint4*)(lds + tid * 16) = wreg0;
    *(uint4*)(lds + (tid + NTHREADS) * 16) = wreg1;
    if (nsteps > 1) { wreg0 = wbase[W_ITEMS + tid]; wreg1 = wbase[W_ITEMS + tid + NTHREADS]; }
  }
  __syncthreads();

  // ---- main loop -----------------------------------------------------------------------------------------
  // Each tap is two K=16 half-steps.  Fragment set F0 feeds the first half, F1 the second, and the one
  // barrier of a tap sits BETWEEN the two MFMA groups:
  //   read F1(s) | MFMA F0(s) | commit W(s+1), request W(s+2) | barrier | read F0(s+1) | MFMA F1(s)
  // Every LDS read is issued a full MFMA group (12 MFMAs) before its first use, and right after the
  // barrier the matrix pipe already has 12 register-resident MFMAs to run.
  struct Frags { bf16x8 ahi[2], alo[2], bhi[2], blo[2]; };
  Frags F0, F1;
  if constexpr (ABL == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        F0.ahi[i][e] = F0.alo[i][e] = F1.ahi[i][e] = F1.alo[i][e] = (short)(0x3f80 + lane + e);
        F0.bhi[i][e] = F0.blo[i][e] = F1.bhi[i][e] = F1.blo[i][e] = (short)(0x3c00 + lane * 3 + e);
      }
  }
#define MD_LOAD_FRAGS(F, PA, PB, KS)                                                         \
  if constexpr (ABL != 1)                                                                    \
  _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) {                                         \
    F.ahi[rm] = *(const bf16x8*)((PA) + (((KS) * 4 + 0) * NT + rm * 32) * 16);               \
    F.alo[rm] = *(const bf16x8*)((PA) + (((KS) * 4 + 1) * NT + rm * 32) * 16);               \
  }                                                                                          \
  if constexpr (ABL != 1)                                                                    \
  _Pragma("unroll") for (int cm = 0; cm < 2; ++cm) {                                         \
    F.bhi[cm] = *(const bf16x8*)((PB) + (((KS) * 2 * PL + 0) * HS + cm * 4 * 24) * 16);      \
    if constexpr (PL == 2)                                                                   \
      F.blo[cm] = *(const bf16x8*)((PB) + (((KS) * 2 * PL + 1) * HS + cm * 4 * 24) * 16);    \
  }
#define MD_MFMA_BF16(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, c_, 0, 0, 0)
#define MD_MFMA_F16(a_, b_, c_) \
  c_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_), __builtin_bit_cast(f16x8, b_), c_, 0, 0, 0)
#define MD_MMA(F)                                                                                            \
  if constexpr (PL == 2) {                                                                                   \
    _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)        \
      MD_MFMA_BF16(F.alo[rm], F.bhi[cm], acc[rm][cm]);                                                       \
    _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)        \
      MD_MFMA_BF16(F.ahi[rm], F.blo[cm], acc[rm][cm]);                                                       \
    _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)        \
      MD_MFMA_BF16(F.ahi[rm], F.bhi[cm], acc[rm][cm]);                                                       \
  } else {                                                                                                   \
    _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)        \
      MD_MFMA_F16(F.alo[rm], F.bhi[cm], acc[rm][cm]);                                                        \
    _Pragma("unroll") for (int rm = 0; rm < 2; ++rm) _Pragma("unroll") for (int cm = 0; cm < 2; ++cm)        \
      MD_MFMA_F16(F.ahi[rm], F.bhi[cm], acc[rm][cm]);                                                        \
  }

  // All wavefronts run the same code in step, so a block of 8 back-to-back ds_read_b128 per wave arrives at
  // the LDS as a 64-instruction burst and every wave's (in-order) MFMA issue stalls behind its own queued
  // reads: measured LDS time was purely additive to MFMA time.  Ask the scheduler for 3 MFMA : 2 DS-read
  // groups so each read is issued in the shadow of the previous MFMA.
#define MD_INTERLEAVE()                                                  \
  if constexpr (ABL != 7) {                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < (PL == 2 ? 4 : 3); ++i_) {   \
      __builtin_amdgcn_sched_group_barrier(0x008, (PL == 2 ? 3 : 3), 0); \
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 \
    }                                                                    \
  }
  int cur = 0;                                       // byte offset of the weight buffer holding W(s)
  int s = 0;
  {
    const unsigned char* pa = lds + vA;
    const unsigned char* pb = lds + vB[0];
    MD_LOAD_FRAGS(F0, pa, pb, 0)
  }
  // The loop body is branch-free (one scheduling region per half-step): past the end, weight tile
  // indices and the prefetched chunk are clamped, so the redundant loads / LDS writes hit valid
  // addresses and are simply never consumed.
  const int last_tile = nsteps - 1;
  for (int cc = 0; cc < ncc; ++cc) {
    const int cpre = (cc + 1 < ncc) ? cc + 1 : cc;   // chunk whose halo is prefetched during this chunk
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int tn = (tap + 1) % TAPS;                 // next step's tap (same offsets in the next chunk)
      const int ndz = tn / 9, ndy = (tn / 3) % 3, ndx = tn % 3;
      const int nxt = cur ^ (W_ITEMS * 16);
      if constexpr ((VAR & 2) == 0) {
        const unsigned char* pa = lds + cur + vA;
        const unsigned char* pb = lds + vB[dz] + (dy * 24 + dx) * 16;
        MD_LOAD_FRAGS(F1, pa, pb, 1)
      }
      if constexpr ((VAR & 2) != 0) {
        // W(s+1) goes to LDS first: the buffer's last readers passed the barrier of tap s-1, its data was requested a
        // whole tap ago, and the 12 MFMAs below run while the LDS store path (13 cycles per ds_write_b128) drains
        *(uint4*)(lds + nxt + tid * 16) = wreg0;
        *(uint4*)(lds + nxt + (tid + NTHREADS) * 16) = wreg1;
        const int ts = (s + 2 < last_tile) ? s + 2 : last_tile;
        const uint4* wt = wbase + (int64_t)ts * W_ITEMS;
        wreg0 = wt[tid]; wreg1 = wt[tid + NTHREADS];
        __builtin_amdgcn_sched_barrier(0);   // keep the store ahead of this tap's fragment reads and MFMAs
        const unsigned char* pa = lds + cur + vA;
        const unsigned char* pb = lds + vB[dz] + (dy * 24 + dx) * 16;
        MD_LOAD_FRAGS(F1, pa, pb, 1)
      }
      if constexpr (MD_SETPRIO) __builtin_amdgcn_s_setprio(1);
      MD_MMA(F0)
      if constexpr (MD_SETPRIO) __builtin_amdgcn_s_setprio(0);
      MD_INTERLEAVE()
      if constexpr (ABL != 3 && (VAR & 2) == 0) {  // W(s+1): registers -> LDS (the other buffer; its last readers passed a barrier)
        *(uint4*)(lds + nxt + tid * 16) = wreg0;
        *(uint4*)(lds + nxt + (tid + NTHREADS) * 16) = wreg1;
      }
      if constexpr (ABL != 4 && (VAR & 2) == 0) {
        const int ts = (s + 2 < last_tile) ? s + 2 : last_tile;
        const uint4* wt = wbase + (int64_t)ts * W_ITEMS;
        wreg0 = wt[tid]; wreg1 = wt[tid + NTHREADS];
      }
      if (tap == (BF ? PF_TAP_BF : PF_TAP)) act_issue(cpre);
      if constexpr (BF) {
        if (tap > PF_TAP_BF && tap <= PF_TAP_BF + 2 * BF_IT) act_transform(tap - PF_TAP_BF - 1);   // half an item per tap, beside the MFMAs
      }
      if (tap == TAPS - 1) {
        __syncthreads();  // every wave has issued and completed its reads of this chunk's halo tile
        act_commit();
      }
      if constexpr (ABL != 3 && ABL != 6) __syncthreads();
      {
        const unsigned char* pa = lds + nxt + vA;
        const unsigned char* pb = lds + vB[ndz] + (ndy * 24 + ndx) * 16;
        MD_LOAD_FRAGS(F0, pa, pb, 0)
      }
      if constexpr (MD_SETPRIO) __builtin_amdgcn_s_setprio(1);
      MD_MMA(F1)
      if constexpr (MD_SETPRIO) __builtin_amdgcn_s_setprio(0);
      MD_INTERLEAVE()
      cur = nxt;
      ++s;
    }
  }
#undef MD_LOAD_FRAGS
#undef MD_MMA
#undef MD_MFMA_BF16
#undef MD_MFMA_F16
#undef MD_INTERLEAVE

  // ---- epilogue: bias + residual loads batched, 16-byte stores into the F32B layout -------------------
  const bool partial = ksplit > 1;  // raw partial sums go to the workspace slice; md_splitk_reduce finishes
  const float alpha = partial ? 1.f : A.alpha;
  const int rows = A.rows, rows_alloc = A.rows_alloc;
  const int rg_alloc = rows_alloc / 8;
  float* outp = partial ? A.partial + ((int64_t)blockIdx.z * A.batch + b) * rg_alloc * P * 8
                        : (float*)A.out + (int64_t)b * rg_alloc * P * 8;
  const float* resp = (A.residual && !partial) ? A.residual + (int64_t)b * A.res_bstride : nullptr;
  const float* biasp = (A.bias && !partial) ? A.bias + (int64_t)b * A.bias_bstride : nullptr;
  f32x4 bv[2][4];
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = rt * NT + wr * 64 + rm * 32 + 8 * q + 4 * h;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (biasp != nullptr) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (row + e < rows) v[e] = biasp[row + e];
      }
      bv[rm][q] = v;
    }
  const bool want_stats = (ABL == 0) && A.stats != nullptr && !partial;   // ablation builds carry no statistics code
  float st1[2][4][4], st2[2][4][4];   // per lane: sum / sum of squares of its 32 output channels over its 2 positions
#pragma unroll
  for (int rm = 0; rm < 2; ++rm)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) { st1[rm][q][e] = 0.f; st2[rm][q][e] = 0.f; }
#pragma unroll
  for (int cm = 0; cm < 2; ++cm) {
    const int y = cm * 4 + (j >> 3), x = j & 7;
    const int64_t gp = ((int64_t)(z0 + wc) * H + (y0 + y)) * W + (x0 + x);
    f32x4 rv[2][4];
#pragma unroll
    for (int rm = 0; rm < 2; ++rm)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rt * NT + wr * 64 + rm * 32 + 8 * q + 4 * h;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (resp != nullptr && row < rows_alloc) v = *(const f32x4*)(resp + ((int64_t)(row >> 3) * P + gp) * 8 + (row & 7));
        rv[rm][q] = v;
      }
#pragma unroll
    for (int rm = 0; rm < 2; ++rm)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rt * NT + wr * 64 + rm * 32 + 8 * q + 4 * h;
        if (row >= rows_alloc) continue;
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // same association as the generic kernel: (alpha*acc + bias) + residual
          float v = alpha * acc[rm][cm][q * 4 + e];
          v += bv[rm][q][e];
          v += rv[rm][q][e];
          o4[e] = v;
          st1[rm][q][e] += v;
          st2[rm][q][e] += v * v;
        }
        *(f32x4*)(outp + ((int64_t)(row >> 3) * P + gp) * 8 + (row & 7)) = o4;
      }
  }
  // ---- optional: GroupNorm statistics of the tensor just written (consumer skips its md_gn_stats pass) ----------
  if constexpr (ABL == 0) if (want_stats) {   // workgroup-uniform
    // A 16-lane DPP row holds the same 32 channels at 16 different positions: four DPP adds (quad_perm xor 1, xor 2,
    // row_half_mirror, row_mirror -- VALU only, no LDS crossbar traffic) leave the row total in every lane; lane 0 of
    // each row parks it in LDS (free: all waves are past their last fragment read after the barrier), where the two
    // rows of a half-wave and the 4 wc waves of a row half are added up.
    auto row_sum = [](float v) {
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
      return v;
    };
#pragma unroll
    for (int rm = 0; rm < 2; ++rm)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          st1[rm][q][e] = row_sum(st1[rm][q][e]);
          st2[rm][q][e] = row_sum(st2[rm][q][e]);
        }
    __syncthreads();
    float* red = (float*)lds;   // [8 waves][2 rows of a half-wave][64 channels][2]
    if ((lane & 15) == 0) {
      const int jr = (lane >> 4) & 1;
#pragma unroll
      for (int rm = 0; rm < 2; ++rm)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ch = rm * 32 + 8 * q + 4 * h + e;
            red[((wid * 2 + jr) * 64 + ch) * 2] = st1[rm][q][e];
            red[((wid * 2 + jr) * 64 + ch) * 2 + 1] = st2[rm][q][e];
          }
    }
    __syncthreads();
    if (tid < 256) {
      const int ch = tid >> 1, which = tid & 1;          // channel within the 128-row tile
      const int w0 = (ch >> 6) * 4, c64 = ch & 63;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) sum += red[(((w0 + (k >> 1)) * 2 + (k & 1)) * 64 + c64) * 2 + which];
      const int row = rt * NT + ch;
      if (row < rows_alloc) atomicAdd(A.stats + ((int64_t)b * rows_alloc + row) * 2 + which, (double)sum);
    }
  }
}

int md_launch_splitk_reduce(const MdGemmConvArgs& a, hipStream_t stream);  // gemm_conv.hip

// launched from gemm_conv.hip (MD_CFG_C3_128_FAST)
int md_launch_conv3_main(const MdGemmConvArgs& a, hipStream_t stream) {
  if (a.kdim % KC != 0 || a.kdim <= 0 || a.rows <= 0 || a.rows_alloc % 8 != 0 || a.batch <= 0) return MD_ERR_BAD_ARG;
  if (a.D % TZ || a.H % TY || a.W % TX) return MD_ERR_BAD_ARG;
  if (a.ups && ((a.D | a.H | a.W) & 1)) return MD_ERR_BAD_ARG;
  if (a.a_src != MD_A_PACKED || a.out_mode != MD_OUT_F32B) return MD_ERR_UNSUPPORTED;
  if (a.b_mode != MD_B_S16B && a.b_mode != MD_B_F32B_GN) return MD_ERR_BAD_ARG;
  if (a.b_mode == MD_B_F32B_GN) {
    if (a.prec != MD_PREC_BF16X3 || (a.b_split & 7) || a.b_split <= 0 || (a.b_split < a.kdim && a.b2 == nullptr)) return MD_ERR_BAD_ARG;
    if (a.b_silu && a.b_ac == nullptr) return MD_ERR_BAD_ARG;   // SiLU is applied together with the folded GroupNorm affine only
  }
  // (with split-K the statistics come from the finish kernel: md_splitk_reduce_stats_kernel)
  const int tiles = (a.D / TZ) * (a.H / TY) * (a.W / TX);
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  if (ks > 1 && (a.partial == nullptr || ks > a.kdim / KC)) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)(tiles * a.batch), (unsigned)((a.rows + NT - 1) / NT), (unsigned)ks);
  MD_HIP_CLEAR_ERROR();
  if (a.prec == MD_PREC_FP16X2) {
    hipLaunchKernelGGL((md_conv3_main_kernel<0, MD_PREC_FP16X2>), grid, dim3(NTHREADS), 0, stream, a);
  } else if (a.prec != MD_PREC_BF16X3) {
    return MD_ERR_BAD_ARG;
  } else {
    switch (a.cfg) {
#ifdef MD_BUILD_ABLATIONS   // timing-only variants for tools/bench_conv.py: each costs ~35 s of compile time
      case 111: if (a.b_mode != MD_B_S16B) return MD_ERR_UNSUPPORTED;
                hipLaunchKernelGGL((md_conv3_main_kernel<1, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
      case 113: hipLaunchKernelGGL((md_conv3_main_kernel<3, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
      case 114: hipLaunchKernelGGL((md_conv3_main_kernel<4, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
      case 116: hipLaunchKernelGGL((md_conv3_main_kernel<6, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
      case 117: hipLaunchKernelGGL((md_conv3_main_kernel<7, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
      case 118: hipLaunchKernelGGL((md_conv3_main_kernel<8, 0>), grid, dim3(NTHREADS), 0, stream, a); break;
#else
      case 111: case 113: case 114: case 116: case 117: case 118: return MD_ERR_UNSUPPORTED;   // MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build --force
#endif
      case 122: if (a.b_mode != MD_B_S16B) return MD_ERR_UNSUPPORTED;
                hipLaunchKernelGGL((md_conv3_main_kernel<0, 0, 2>), grid, dim3(NTHREADS), 0, stream, a); break;
      default:
        if (a.b_mode == MD_B_F32B_GN) hipLaunchKernelGGL((md_conv3_main_kernel<0, 0, 0, 1>), grid, dim3(NTHREADS), 0, stream, a);
        else hipLaunchKernelGGL((md_conv3_main_kernel<0, 0>), grid, dim3(NTHREADS), 0, stream, a);
        break;
    }
  }
  MD_HIP_CHECK_LAUNCH();
  if (ks > 1) return md_launch_splitk_reduce(a, stream);
  return MD_OK;
}
