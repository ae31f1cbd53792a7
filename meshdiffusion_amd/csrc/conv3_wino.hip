// 3x3x3 stride-1 convolution through a 1-D Winograd F(2,3) transform along the innermost (w) axis, for the inference path.
//
// Reference op: nn.Conv3d 3x3x3 pad 1 (lib/diffusion/models/layers.py:118-124) behind nn.GroupNorm + nn.SiLU
// (layers.py:676-681), optionally on torch.cat([h, skip], 1) (ddpm_res64.py:174-176) or on the nearest-x2 upsampled
// input (layers.py:618-623).
//
// For one output pair (x = 2i, 2i+1) with inputs d0..d3 = in[2i-1 .. 2i+2] and taps g0 g1 g2 along w:
//     m0 = (d0 - d2) g0          m1 = (d1 + d2) (g0+g1+g2)/2       m2 = (d2 - d1) (g0-g1+g2)/2       m3 = (d1 - d3) g2
//     y0 = m0 + m1 + m2          y1 = m1 - m2 - m3
// 4 products per 2 outputs instead of 6: the 27-tap contraction becomes 9 (kd, kh) taps x 4 "frequencies" on half the
// columns = 2/3 of the matrix-core work, still in bf16x3 (hi*hi + hi*lo + lo*hi) with fp32 accumulation; the
// transforms are exact-ish fp32 adds (measured error of one conv vs fp64: 5.5e-6, direct bf16x3: 4.5e-6).
//
// Three kernels:
//   md_wino_prep         fp32 F32B parts -> (GroupNorm affine, SiLU, zero pad, input transform, bf16 hi/lo split)
//                        -> T[B][C/8][f=4][plane=2][D][H][W/2][8 bf16]        (HBM-bound: reads 4 B, writes 8 B / element)
//   md_wino_pack_weights [Cout][Cin][3][3][3] fp32 -> G-transformed split tiles in MFMA A-fragment order
//   md_conv3_wino        the contraction.  One workgroup = 128 output channels x (4 x 8 x 8) positions = 128 rows x 128
//                        pairs x 4 frequencies of accumulators = the whole register file of a CU (4 waves x 512
//                        registers, 256 of them AccVGPRs).  WAVE f OWNS FREQUENCY f: its 128 x 128 tile needs only the
//                        freq-f slice of T (LDS-DMA into a private double-buffered halo, 2 x 15 KB) and the freq-f weight
//                        fragments (1 KB contiguous each, straight from L2 into registers -- no other wave wants them), so
//                        the main loop has NO barrier and no VALU work at all: per (kd, kh) step and 16-channel chunk a wave
//                        issues 48 MFMAs, 8 ds_read_b128 (0.17 per MFMA; conv3_main: 0.67) and 8 global loads.
//                        Only the epilogue meets: accumulators cross through LDS (m0..m3 of a pair live in 4 waves),
//                        y0 / y1 + bias + residual + GroupNorm sums as in md_conv3_main_kernel.
#include <type_traits>
#include "md_common.h"
#include "md_pack.h"

namespace {
constexpr int WN_THREADS = 256;
constexpr int WN_TZ = 4, WN_TY = 8, WN_TX = 8;          // output tile; 4 pairs along w
constexpr int WN_TPOS = 6 * 10 * 4;                      // 240 transformed halo entries (dz, hy, pair) per (k-group, plane)
constexpr int WN_HBUF = 4 * WN_TPOS * 16;                // 15360 B: [h 2][plane 2][240][16 B] = one chunk of one frequency
constexpr int WN_NDMA = WN_HBUF / 1024;                  // 15 pieces of 1 KB (64 lanes x 16 B) per chunk and wave
constexpr int WN_RED = 4 * 32 * 2;                       // floats: [wave][channel][sum, sumsq]
constexpr int WN_WAVE_LDS = 2 * WN_HBUF;                 // 30720 B private to a wave: two halo buffers
constexpr int WN_LDS_BYTES = 4 * WN_WAVE_LDS + 4 * WN_NDMA * 64 * 4;      // 138240 B: halo buffers + the f16f8 loop's offset tables; the epilogue's
                                                                          // exchange area (2 x 65792 B + statistics) aliases them

__device__ const uint4 wn_zero16 = {0u, 0u, 0u, 0u};     // source of halo entries outside the grid

typedef uint32_t wn_u32x4 __attribute__((ext_vector_type(4)));
// 16-byte load through the GLOBAL address space (a pointer that went through a select loses it and becomes a flat load,
// which also counts against lgkmcnt)
__device__ __forceinline__ uint4 wn_gload16(const void* p) {
  return __builtin_bit_cast(uint4, *(__attribute__((address_space(1))) const wn_u32x4*)(uintptr_t)p);
}
}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// md_wino_prep: one thread = one (sample, 8-channel group, z, y, pair)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void md_wino_prep_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                                           int c1, int c2, const float* __restrict__ ac, int silu, int ups,
                                                           uint4* __restrict__ T, int batch, int D, int H, int W,
                                                           uint32_t thr16, float drop_scale, uint64_t seed) {
  const int Wp = W >> 1;
  const int64_t Ph = (int64_t)D * H * Wp;
  const int CG = (c1 + c2) >> 3;
  const int64_t n = (int64_t)batch * CG * Ph;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  const int64_t pos2 = id % Ph;
  const int cg = (int)((id / Ph) % CG);
  const int b = (int)(id / (Ph * CG));
  const int pr = (int)(pos2 % Wp), y = (int)((pos2 / Wp) % H), z = (int)(pos2 / ((int64_t)Wp * H));
  int Di = D, Hi = H, Wi = W, zs = z, ys = y;
  if (ups) { Di >>= 1; Hi >>= 1; Wi >>= 1; zs >>= 1; ys >>= 1; }
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  const float* src = (cg * 8 < c1) ? x1 + ((int64_t)b * (c1 >> 3) + cg) * Pin * 8
                                   : x2 + ((int64_t)b * (c2 >> 3) + (cg - (c1 >> 3))) * Pin * 8;
  float a[8], c[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = 1.f; c[e] = 0.f; }
  if (ac != nullptr) {
    const f32x4* ap = (const f32x4*)(ac + ((int64_t)b * (c1 + c2) + cg * 8) * 2);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 v = ap[q];
      a[2 * q] = v[0]; c[2 * q] = v[1]; a[2 * q + 1] = v[2]; c[2 * q + 1] = v[3];
    }
  }
  float d[4][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int xw = 2 * pr - 1 + q;
    const bool live = xw >= 0 && xw < W;             // the conv pads the ACTIVATED tensor with zeros
    f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    const int xs = ups ? (xw >> 1) : xw;
    const int64_t spos = ((int64_t)zs * Hi + ys) * Wi + xs;
    if (live) {
      const f32x4* p = (const f32x4*)(src + spos * 8);
      v0 = p[0]; v1 = p[1];
    }
    uint64_t bits[2] = {0, 0};
    if (thr16) {      // training: the dropout mask of md_gn_apply (same counter-based hash of (seed, sample, channel quad, position))
      const uint64_t quad0 = (uint64_t)(((int64_t)b * (c1 + c2) + cg * 8) >> 2) * (uint64_t)Pin;
      bits[0] = md_drop_bits(seed, quad0 + (uint64_t)spos);
      bits[1] = md_drop_bits(seed, quad0 + (uint64_t)Pin + (uint64_t)spos);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float yv = e < 4 ? v0[e] : v1[e - 4];
      if (ac != nullptr) {
        yv = yv * a[e] + c[e];
        if (silu) yv = yv * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(yv * -1.4426950408889634f));
      }
      if (thr16) yv = md_drop_keep(bits[e >> 2], e & 3, thr16) ? yv * drop_scale : 0.f;   // nn.Dropout (layers.py:682)
      d[q][e] = live ? yv : 0.f;
    }
  }
  uint4* out = T + ((int64_t)b * CG + cg) * 8 * Ph + pos2;       // [f][plane][Ph] items of 16 B
#pragma unroll
  for (int f = 0; f < 4; ++f) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
      t[e] = f == 0 ? d[0][e] - d[2][e] : f == 1 ? d[1][e] + d[2][e] : f == 2 ? d[2][e] - d[1][e] : d[1][e] - d[3][e];
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) md_split2(t[2 * q], t[2 * q + 1], hw[q], lw[q]);
    out[(int64_t)(f * 2) * Ph] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    out[(int64_t)(f * 2 + 1) * Ph] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// md_wino_pack_weights: one thread = one 16-byte item (8 consecutive input channels of one output row, one plane)
//   layout [cout/128][cin/16][tap (kd,kh) 9][f 4][row tile 4][plane 2][h 2][row 32][8 bf16]
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void md_wino_pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ wpk,
                                                                   int cout, int cin, int64_t s_row, int64_t s_k, int flip) {
  const int64_t n = (int64_t)cout * cin * 36 / 4;        // items: cout * cin * 36 values * 2 planes / 8
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (id >= n) return;
  wpk[id] = md_pack_wino_item(w, cout, cin, s_row, s_k, flip, id);
}

// f16f8 weights: max |w| of the tensor (the pre-scale 2^sw is derived from it on the device: no host round trip), then the
// fragments; header behind the fragments: {max |w| (float bits), sw (int), 2^-sw (float), 0}
__global__ __launch_bounds__(256) void md_wino_amax_kernel(const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k,
                                                           uint32_t* __restrict__ hdr, const float* __restrict__ eq) {
  const int64_t n = (int64_t)cout * cin * 27;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int t = (int)(i % 27);
    const int64_t r = i / 27;
    float v = fabsf(w[(r / cin) * s_row + (r % cin) * s_k + t]);
    if (eq) v /= eq[r % cin];                // the equalised weights are what gets packed (md_wino_equaliser; powers of two: exact)
    m = v > m ? v : m;                       // NaN never wins: a non-finite weight tensor packs with sw = 0
  }
  m = md_wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(hdr, __float_as_uint(m));      // non-negative floats order like their bit patterns
}

__global__ __launch_bounds__(256) void md_wino_pack_weights_f8_kernel(const float* __restrict__ w, uint4* __restrict__ wpk, int cout, int cin,
                                                                      int64_t s_row, int64_t s_k, uint32_t* __restrict__ hdr,
                                                                      const float* __restrict__ eq) {
  const int64_t n = (int64_t)cout * cin * 9;             // 16-byte items: cout * cin * 36 values * 4 B / 16
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float wscale = md_wino_f8_wscale(__uint_as_float(hdr[0]));
  if (id == 0) {
    hdr[1] = (uint32_t)ilogbf(wscale);
    hdr[2] = __float_as_uint(1.0f / wscale);
    hdr[3] = 0u;
  }
  if (id >= n) return;
  wpk[id] = md_pack_wino_f8_item(w, cout, cin, s_row, s_k, wscale, id, eq);
}

__global__ __launch_bounds__(256) void md_wino_pack_weights_f6_kernel(const float* __restrict__ w, uint4* __restrict__ wpk, int cout, int cin,
                                                                      int64_t s_row, int64_t s_k, uint32_t* __restrict__ hdr,
                                                                      const float* __restrict__ eq) {
  const int64_t n = (int64_t)cout * cin * 9;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float wscale = md_wino_f8_wscale(__uint_as_float(hdr[0]));
  if (id == 0) {
    hdr[1] = (uint32_t)ilogbf(wscale);
    hdr[2] = __float_as_uint(1.0f / wscale);
    hdr[3] = 6u;                                          // the cross-term format of this buffer (md_conv3_wino_f6 checks nothing: host-side type)
  }
  if (id >= n) return;
  wpk[id] = md_pack_wino_f6_item(w, cout, cin, s_row, s_k, wscale, id, eq);
}

// ------------------------------------------------------------------------------------------------------------------
// md_conv3_wino
// ------------------------------------------------------------------------------------------------------------------
struct WnArgs {
  const uint4* T;          // [B][cin/8][4][2][D][H][W/2] items of 16 B
  const uint4* wpk;
  float* out;              // F32B [B][cout/8][P][8]
  const float* bias;       // may be null; per sample with stride bias_bstride (0 = shared)
  const float* residual;   // F32B like out, may be null
  double* stats;           // may be null
  const float* hdr;        // F8: the weight buffer's header {max |w|, sw, 2^-sw} behind the fragments (md_wino_pack_weights_f8)
  int64_t bias_bstride, res_bstride;
  int batch, cin, cout, D, H, W;
  float out_scale;         // F8: multiplies the weights' descale (1, or 1 / tscale of md_wino_prep_dual_f6: a power of two)
  const uint32_t* amax;    // F8, may be null: the operand was lifted by 2^md_dgrad_lift_log2(amax[0]) (md_wino_prep_dual_f6's dynamic form)
};

// ABL (timing only, results invalid; -DMD_BUILD_ABLATIONS, tools/bench_wino.py): bit 0 no halo traffic, bit 1 halo traffic in the
//   prologue only (real data stays in LDS), bit 2 weight loads in the prologue only (three real sets reused), bit 3 no LDS fragment
//   reads, bit 4 no epilogue, bit 5 halo read from a private L2-resident 30 KB, bit 6 halo read as a private contiguous
//   HBM stream, bit 7 (valid results, no statistics) per-wave s_memtime stamps of the first 1024 workgroups into the buffer
//   passed as `stats` (tools/bench_wino.py --stamps: where a workgroup's time goes).  NOTE: with bit 0 / 3 the MFMAs run on constant operands and the chip clocks higher (data-dependent power):
//   such runs bound the MFMA time from below, they do not price the removed traffic.
//
// F8 = the "f16f8" arithmetic (md_common.h md_split_f16f8; operand from md_wino_prep_f8, weights from md_wino_pack_weights_f8): a
// product is fp16(a) fp16(b) + [e4m3(a) e4m3(b_lo 2^11) + e4m3(a_lo 2^11) e4m3(b)] 2^-11.  Per TWO steps (a "pair-step": steps 2p, 2p+1
// of the sequence step = chunk * 9 + tap) and accumulator tile: two v_mfma_f32_32x32x16_f16 + ONE v_mfma_scale_f32_32x32x64_f8f6f4
// whose K = 64 is [lanes 0-31: step 2p | lanes 32-63: step 2p+1] x [16 channels x (e4m3(a), e4m3(a_lo))] -- 4 matrix-core units
// of 32 cycles where bf16x3 issues 6 (measured on random data, tools/probes/f8_probe.hip: 0.636 of the bf16x3 time).  Same T
// geometry, same halo image, same epilogue (times 2^-sw, the weights' power-of-two pre-scale).
typedef int wn_i32x8 __attribute__((ext_vector_type(8)));
typedef int wn_i32x4 __attribute__((ext_vector_type(4)));

// F6 (with F8) = the "f16f6" arithmetic (md_common.h md_split_f16f6; operand from md_wino_prep_f6, weights from md_wino_pack_weights_f6): the
// cross terms as MX block-scaled e2m3 x e2m3 (format code 2), which the same K = 64 MFMA executes at twice its e4m3 rate (measured
// in the fp16 mix, tools/probes/f6_probe.hip: 0.85 of the f16f8 pair-step on random data).  Same geometry, loads, LDS image and
// schedule: the 32-byte fragment a lane assembles from its two 16-byte items is [6 registers of codes | the block's E8M0 scale |
// 0], and register 6 of either fragment is that MFMA's per-lane scale operand.
// RES: the launch has a residual operand (compile-time: with the request behind a run-time branch hipcc cannot count the loads in
// flight and waits with vmcnt(0) -- for the NEXT round's prefetch and the round's own stores as well, profiles/r06_wino_epilogue_ab.txt)
template <int ABL, bool F8 = false, bool F6 = false, bool RES = false>
__global__ __launch_bounds__(WN_THREADS) void md_conv3_wino_kernel(const WnArgs A) {
  __shared__ __attribute__((aligned(16))) unsigned char wn_smem[WN_LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);       // = frequency f of this wave
  const int j = lane & 31, h = lane >> 5;
  uint64_t stamp[12];      // 0..8: s_memtime (shader cycles) at the phase boundaries; 9: hardware ids; 10, 11: s_memrealtime (100 MHz) at start / end
  auto mark = [&](int k) { if constexpr (ABL & 128) stamp[k] = __builtin_amdgcn_s_memtime(); };
  if constexpr (ABL & 128) stamp[10] = __builtin_amdgcn_s_memrealtime();
  mark(0);

#ifndef W8_STAGGER
#define W8_STAGGER 0      // A/B: > 0 delays the first workgroup of every CU by a hashed phase of up to that many shader cycles (see conv3_main.hip)
#endif
  if constexpr (W8_STAGGER > 0) {
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
    if (lin < 256u) {
      const unsigned ph = (lin * 0x9E3779B1u) >> 24;
      const uint64_t t_end = __builtin_amdgcn_s_memtime() + ((uint64_t)ph * (uint64_t)W8_STAGGER >> 8);
      while (__builtin_amdgcn_s_memtime() < t_end) __builtin_amdgcn_s_sleep(16);
    }
  }
  const int D = A.D, H = A.H, W = A.W, Wp = W >> 1;
  const int64_t P = (int64_t)D * H * W, Ph = P >> 1;
  const int ntx = W / WN_TX, nty = H / WN_TY, ntz = D / WN_TZ;
  const int tiles = ntx * nty * ntz;
  int bid = blockIdx.x;     // XCD-aware order: one contiguous run of tiles per XCD (block b runs on XCD b % 8)
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles, t = bid % tiles;
  const int x0 = (t % ntx) * WN_TX, y0 = ((t / ntx) % nty) * WN_TY, z0 = (t / (ntx * nty)) * WN_TZ;
  const int rtb = blockIdx.y;                                      // 128-row block of output channels
  const int CG = A.cin >> 3, nchunk = A.cin >> 4;
  const int nsteps = nchunk * 9;

  // ---- this wave's private LDS: two halo buffers (one chunk of one frequency each) -------------------------------------
  unsigned char* my_smem = wn_smem + wid * WN_WAVE_LDS;

  // ---- halo pieces: piece k = entries [64 k, 64 k + 64) of the linear image [h][plane][dz][hy][pair], one entry per lane;
  // source offset in 16-byte items relative to the (sample, chunk, freq) base, -1 = outside the grid (zero).  Recomputed
  // where a piece is requested (a table would hold 15 registers for the whole kernel).
  // (measured alternative: hi / lo planes side by side in T, so that a (dz, hy) row is one whole 128-byte line and a piece
  // 8 lines instead of 16 half lines: conv +4 %, prep +20 % slower -- reverted)
  auto halo_off = [&](int k) -> int {
    const int e = k * 64 + lane;
    const int hp = e / WN_TPOS, tp = e % WN_TPOS;
    const int dz = tp / 40, hy = (tp >> 2) % 10, pr = tp & 3;
    const int z = z0 + dz - 1, y = y0 + hy - 1;
    const bool live = (z >= 0) & (z < D) & (y >= 0) & (y < H);
    // cg = 2 chunk + (hp >> 1); inside a cg: [f][plane][Ph]; the f term is in the base
    if constexpr (F8) {      // 32-bit and unconditional (16 Ph < 2^31 is checked at launch): stays a select, no branch in the loop
      const int Phi = (int)Ph;
      const int off = (hp >> 1) * 8 * Phi + (hp & 1) * Phi + (z * H + y) * Wp + (x0 >> 1) + pr;
      return live ? off : -1;
    }
    return live ? (int)((int64_t)(hp >> 1) * 8 * Ph + (int64_t)(hp & 1) * Ph + ((int64_t)z * H + y) * Wp + (x0 >> 1) + pr) : -1;
  };
  const uint4* tbase = A.T + ((int64_t)b * CG * 8 + wid * 2) * Ph;                              // + chunk * 16 * Ph
  const uint4* zsrc = &wn_zero16;
  auto halo_load = [&](int chunk, int k) -> uint4 {
    const int dk = halo_off(k);
    const uint4* src = dk >= 0 ? tbase + (int64_t)chunk * 16 * Ph + dk : zsrc;
    if constexpr (ABL & 32)      // timing only: a private 30 KB per workgroup slot and wave that stays in L2
      src = A.T + (((blockIdx.x & 255) * 4 + wid) * 2 + (chunk & 1)) * 960 + k * 64 + lane;
    if constexpr (ABL & 64)      // timing only: a private contiguous 15 KB per (workgroup, wave, chunk) (128 ch @ 64^3, B = 8 only)
      src = A.T + ((((((int64_t)bid * 4 + wid) * nchunk + chunk) * 960) & (((int64_t)1 << 26) - 1)) + k * 64 + lane);
    if constexpr (ABL & 1) return make_uint4(0, 0, 0, 0);
    return wn_gload16(src);
  };
  auto halo_store = [&](int buf, int k, const uint4& v) {
    if constexpr (!(ABL & 1)) *(uint4*)(my_smem + buf * WN_HBUF + k * 1024 + lane * 16) = v;
  };
  // weights: step s = chunk * 9 + tap; this wave's 8 fragments (row tile 4 x plane 2) of step s are 8 KB contiguous, a
  // fragment (32 rows x 16 k, one plane) 1 KB = one 16-byte load per lane: they go straight to registers (no other wave
  // wants them), ring of three sets, requested two steps ahead
  const uint4* wbase = A.wpk + (((int64_t)rtb * nsteps) * 4 + wid) * 512 + lane;               // + s * 2048 + i * 64
  auto load_A = [&](int s, bf16x8 (&dst)[8]) {
    const uint4* wp = wbase + (int64_t)s * 2048;
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = __builtin_bit_cast(bf16x8, wp[i * 64]);
  };
  // halo fragment of column tile ct (= output plane z0 + ct), tap (kd, kh):  entry (ct + kd) * 40 + kh * 4 + j
  const unsigned char* vB = my_smem + h * 2 * WN_TPOS * 16 + j * 16;
  auto read_B = [&](int tap, int buf, bf16x8 (&dst)[8], bool force) {
    const int kd = tap / 3, kh = tap % 3;
    const unsigned char* p = vB + buf * WN_HBUF + (kd * 40 + kh * 4) * 16;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      if (force || !(ABL & 8)) {
        dst[ct * 2] = *(const bf16x8*)(p + ct * 40 * 16);
        dst[ct * 2 + 1] = *(const bf16x8*)(p + WN_TPOS * 16 + ct * 40 * 16);
      }
  };

  f32x16 acc[4][4];
  auto zero_acc = [&]() {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[rt][ct][r] = 0.f;
        asm volatile("" : "+a"(acc[rt][ct]));          // the 256 accumulator registers are the AccVGPR half of the file
      }
  };
#ifndef W8_PRO_FENCE
#define W8_PRO_FENCE 1      // A/B: 0 = the prologue order hipcc chooses by itself (accumulators zeroed and all offsets computed before the first request)
#endif
  if constexpr (!W8_PRO_FENCE) zero_acc();
  float descale = 1.f;
  if constexpr (F8) {
    static_assert(ABL == 0 || ABL == 128, "timing ablations exist for the bf16x3 loop only");
    // ---- F8 main loop -------------------------------------------------------------------------------------------------
    // Pair-step u of a body (2 chunks = 18 steps = 9 pair-steps; chunk c0 in halo buffer 0, c0 + 1 in buffer 1; pair-step 4 is
    // tap 8 of the first and tap 0 of the second chunk) = 4 GROUPS, one per column tile ct (output plane z0 + ct):
    //     12 MFMAs: fp16 step 2u x 4 row tiles, fp16 step 2u+1 x 4 row tiles, fp8 (both steps) x 4 row tiles   (512 cycles)
    //  || the 4 B fragments of the NEXT group (16 registers, double-buffered per group: LDS latency is one group at most)
    //  || 8 of the 16 weight pieces of the NEXT pair-step in groups 0 and 1 (two register sets of 64: a piece is a 1 KB-contiguous
    //     16-byte load per lane as before; this wave's 16 pieces of a pair-step are 16 KB contiguous)
    //  || up to four halo pieces of a later chunk requested (groups 2, 3), one requested two pair-steps earlier stored per group (two
    //     register slots of 4 pieces):
    //     chunk c0 + 1 (buffer 1, free from pair-step 0 on: its last reader is pair-step 8 of the body before): requested at
    //     pair-steps 7, 8 (of the body before; the prologue for chunk 1) and 0, 1, stored at 0, 1, 2, 3 -- the first reader is the
    //     prefetch of pair-step 4's fragments in the last group of pair-step 3, issued after that group's store;
    //     chunk c0 + 2 (buffer 0, free after pair-step 4): requested at 3, 4, 5, 6, stored at 5, 6, 7, 8.
    // A body has an odd number of pair-steps, so two bodies (PB = 0 / 1: which weight set holds pair-step 0) are unrolled.
    // W8_HG: the halo request schedule of the pair-step loop.  A wave's loads return in order and it can only keep as many requests
    // in flight as it has registers for them: with 8 pieces (two groups of four, requested two pair-steps before their store) the loop
    // measured 86.7 k cycles per workgroup against 75.2 k with the same requests served from a resident kilobyte (s_memtime stamps,
    // profiles/r04_f8_halo_latency.txt) -- a latency x parallelism limit, not a placement one (1 and 2 below measured neutral):
    //   0: requests 2 + 2 in groups 2 and 3, entry q stored by group q, two pair-steps later
    //   1: all 16 weight requests in group 0, the four halo requests in group 1, entries 0 1 | 2 3 stored by groups 0 | 1
    //   2: weights in groups 0 1, the four halo requests in group 2, entries 0 | 1 | 2 3 stored by groups 0 | 1 | 2
    //   3: as 2 with THREE groups in flight, requested three pair-steps before their store (12 pieces = 12 KB per wave, +16 registers)
    // (a store of entry q always precedes the request that refills it: pass 0 of a group runs before its pass 1)
#ifndef W8_HG
#define W8_HG 3
#endif
    // W8_HG = 3, slots by interval colouring over a body's 9 pair-steps (request u_r -> store u_s, ' = next body): slot 0: 0 -> 3, 3 -> 6,
    // 6 -> 0'; slot 1: 2 -> 5, 5 -> 8, 8 -> 2'; slot 2: 4 -> 7, 7 -> 1'
    auto slot3_of_store = [](int u) constexpr -> int { return (u == 0 || u == 3 || u == 6) ? 0 : ((u == 2 || u == 5 || u == 8) ? 1 : 2); };
    auto slot3_of_request = [](int u) constexpr -> int { return (u == 0 || u == 3 || u == 6) ? 0 : ((u == 2 || u == 5 || u == 8) ? 1 : 2); };
    const int npairs = nsteps >> 1;
    const uint4* wbase8 = A.wpk + (((int64_t)rtb * npairs) * 4 + wid) * 1024 + lane;       // + p * 4096 + piece * 64
    // weights [set][row tile]: the two fp16 fragments (step 2p, 2p+1) and the 32-byte fp8 fragment (8 consecutive registers: its two
    // 16-byte loads write the halves directly); halo fragments of one group [group parity] likewise; halo pieces in flight [slot][group]
    uint4 A8h[2][4][2], B8h[2][2], hs8[W8_HG == 3 ? 3 : 2][4];
    wn_i32x8 A8f[2][4], B8f[2];
    // The source offsets of the 15 halo pieces (loop-invariant, ~45 VALU each to recompute, 15 registers to keep) live in a
    // 3.75 KB table per wave behind the halo buffers (the epilogue's exchange area reuses the space): one ds_read_b32 per request.
    // Bit k of live_mask: this lane's entry of piece k lies inside the grid (the compiler keeps the 15 masks in SGPR pairs).
    uint32_t* otab = (uint32_t*)(wn_smem + 4 * WN_WAVE_LDS) + wid * (WN_NDMA * 64) + lane;      // unsigned: no sign extension behind the read
    static_assert(4 * WN_WAVE_LDS + 4 * WN_NDMA * 64 * 4 <= WN_LDS_BYTES, "offset tables fit behind the halo buffers");
    uint32_t live_mask = 0;
    constexpr int D1 = 64, D2 = (40 - 8) * 16, D3 = WN_HBUF - (2 * 40 + 2 * 4) * 16;      // address of step 2u+1 minus step 2u
    const unsigned char* vBh = my_smem + h * 2 * WN_TPOS * 16 + j * 16;                    // k-group h, plane 0 (fp16)
    const unsigned char* vB8a = my_smem + WN_TPOS * 16 + j * 16 + h * D1;                  // k-group 0, plane 1 (fp8), step 2u + h
    const unsigned char* vB8b = my_smem + WN_TPOS * 16 + j * 16 + h * D2;
    const unsigned char* vB8c = my_smem + WN_TPOS * 16 + j * 16 + h * D3;
    auto step_off = [](int s) constexpr -> int { return (s / 9) * WN_HBUF + (((s % 9) / 3) * 40 + ((s % 9) % 3) * 4) * 16; };
    auto cat8 = [](const uint4& x, const uint4& y) -> wn_i32x8 {
      const wn_i32x4 a = __builtin_bit_cast(wn_i32x4, x), b = __builtin_bit_cast(wn_i32x4, y);
      return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    };
    auto halo_store8 = [&](int buf, int k, uint4 v) {
      const bool lv = (live_mask >> k) & 1u;
      v.x = lv ? v.x : 0u; v.y = lv ? v.y : 0u; v.z = lv ? v.z : 0u; v.w = lv ? v.w : 0u;
      *(uint4*)(my_smem + buf * WN_HBUF + k * 1024 + lane * 16) = v;
    };
    auto read_B8 = [&](auto uc, auto ctc, uint4 (&dh)[2], wn_i32x8& df) {
      constexpr int u = decltype(uc)::value, ct = decltype(ctc)::value;
      constexpr int o0 = step_off(2 * u) + ct * 40 * 16, o1 = step_off(2 * u + 1) + ct * 40 * 16, dk = o1 - o0;
      static_assert(dk == D1 || dk == D2 || dk == D3, "three kinds of step pairs");
      const unsigned char* v8 = dk == D1 ? vB8a : (dk == D2 ? vB8b : vB8c);
      dh[0] = *(const uint4*)(vBh + o0);
      dh[1] = *(const uint4*)(vBh + o1);
      df = cat8(*(const uint4*)(v8 + o0), *(const uint4*)(v8 + o0 + 2 * WN_TPOS * 16));
    };
#ifndef W8_ABL
#define W8_ABL 0      // timing-only A/B builds of the f16f8 loop (results invalid): 1 weights always those of pair-step 0, 2 halo requests from one
#endif                // resident 1 KB per wave, 4 no halo stores in the loop, 8 no fragment reads in the loop (tools/ab_f8_sched.sh)
    auto load_A8 = [&](int p, int set, int rt0, int nrt) {      // row tiles rt0 .. rt0 + nrt - 1 of pair-step p
      const uint4* wp = wbase8 + (int64_t)((W8_ABL & 1) ? 0 : p) * 4096;
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
        if (rt >= rt0 && rt < rt0 + nrt) {
          A8h[set][rt][0] = wp[(rt * 4 + 0) * 64];
          A8h[set][rt][1] = wp[(rt * 4 + 1) * 64];
          A8f[set][rt] = cat8(wp[(rt * 4 + 2) * 64], wp[(rt * 4 + 3) * 64]);
        }
    };
    // ---- prologue: weights of pair-step 0; chunk 0 -> buffer 0 and pieces 0..7 of chunk 1 into the two slots; the offset table.
    // Order FENCED (hipcc otherwise computes all 15 offsets and zeroes the 256 accumulator registers -- ~800 VALU, ~2 us -- before
    // the first request leaves, and sinks the weight requests behind the halo stores: profiles/r04_mid_*, ISA of f0db013): the 16
    // weight requests first (their address is scalar + lane), then offset k -> request k, the accumulators are zeroed while the
    // requests are in flight, and only then the stores wait for the data ----
    {
      uint4 h0[WN_NDMA];
      load_A8(0, 0, 0, 4);
      if constexpr (W8_PRO_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < WN_NDMA; ++k) {
        const int dk = halo_off(k);
        const uint32_t o = dk >= 0 ? (uint32_t)dk : 0u;      // outside the grid: a valid address (entry 0 of the chunk), zeroed on its way to LDS
        live_mask |= (dk >= 0 ? 1u : 0u) << k;
        otab[k * 64] = o;
        h0[k] = wn_gload16(tbase + o);
#ifndef W8_PRO_SPLIT
#define W8_PRO_SPLIT 1      // A/B: 0 = chunk 1's first pieces requested interleaved with chunk 0's (the wait for chunk 0 then covers them: loads return in order)
#endif
        // chunk 1's first groups of four (stored by pair-steps 0, 1 (, 2)): slots as the loop's store side expects them
        if constexpr (!W8_PRO_SPLIT)
          if (k < (W8_HG == 3 ? 12 : 8)) hs8[W8_HG == 3 ? slot3_of_store(k >> 2) : (k >> 2)][k & 3] = wn_gload16(tbase + (int64_t)16 * Ph + o);
        if constexpr (W8_PRO_FENCE) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (W8_PRO_SPLIT) {      // behind ALL of chunk 0 in the memory queue; the offsets come back out of the table
#pragma unroll
        for (int k = 0; k < (W8_HG == 3 ? 12 : 8); ++k)
          hs8[W8_HG == 3 ? slot3_of_store(k >> 2) : (k >> 2)][k & 3] = wn_gload16(tbase + (int64_t)16 * Ph + otab[k * 64]);
        if constexpr (W8_PRO_FENCE) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (W8_PRO_FENCE) zero_acc();
      descale = A.hdr[2] * A.out_scale;
      if (A.amax != nullptr) descale *= ldexpf(1.f, -md_dgrad_lift_log2(A.amax[0]));
      if constexpr (W8_PRO_FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int k = 0; k < WN_NDMA; ++k) halo_store8(0, k, h0[k]);
    }
    read_B8(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, B8h[0], B8f[0]);
    mark(1);
    const int scale_a = 127 - 11, scale_b = 127;       // E8M0: the fp8 products carry 2^-11
    // first of the (up to) four halo pieces pair-step u requests, two in group 2 and two in group 3 (99 = none): pieces 8..14 of
    // chunk c0 + 1 at pair-steps 0, 1; all of c0 + 2 at 3..6; 0..7 of c0 + 3 at 7, 8
    auto base_of = [](int u) constexpr -> int {
      if (W8_HG == 3)      // three pair-steps ahead: piece 12.. of c0 + 1 at pair-step 0; all of c0 + 2 at 2..5; 0..11 of c0 + 3 at 6, 7, 8
        return u == 0 ? 12 : (u >= 2 && u <= 5 ? (u - 2) * 4 : (u >= 6 ? (u - 6) * 4 : 99));
      return u == 0 ? 8 : (u == 1 ? 12 : (u >= 3 && u <= 6 ? (u - 3) * 4 : (u >= 7 ? (u - 7) * 4 : 99)));
    };
    uint32_t off_next[4] = {0u, 0u, 0u, 0u};             // source offsets of the next group's requests, read one group ahead
    auto body = [&](auto pbc, int c0) {
      constexpr int PB = decltype(pbc)::value;
      const int p0 = (c0 * 9) >> 1;
      const int cn1 = c0 + 1, cn2 = c0 + 2 < nchunk ? c0 + 2 : nchunk - 1, cn3 = c0 + 3 < nchunk ? c0 + 3 : nchunk - 1;
      // the last body's requests for chunks that do not exist keep the loop one basic block; W8_TAIL_RES = 1 points them all at ONE resident
      // line (entry 0 of the last chunk: the offsets are masked to zero) instead of re-reading that chunk's halo: with 8 chunks they are a
      // fifth of all halo requests, and the weight waits behind them cover their latency like any other's
#ifndef W8_TAIL_RES
#define W8_TAIL_RES 1
#endif
      const uint32_t m2 = (!W8_TAIL_RES || c0 + 2 < nchunk) ? ~0u : 0u, m3 = (!W8_TAIL_RES || c0 + 3 < nchunk) ? ~0u : 0u;      // wave-uniform offset masks
      auto group = [&](auto uc, auto ctc) {
        constexpr int u = decltype(uc)::value, ct = decltype(ctc)::value;
        constexpr int aset = (PB + u) & 1, bset = ct & 1;
        constexpr int un = ct < 3 ? u : (u + 1) % 9, ctn = ct < 3 ? ct + 1 : 0;      // the group after this one
        // Three fenced parts, one per MFMA pass over the 4 row tiles (hipcc otherwise lines the three MFMAs of an accumulator up
        // back to back -- a dependent chain with hazard nops between the fp16 and the fp8 form -- and clumps the memory operations):
        // -- pass 0 (fp16, step 2u): the halo piece requested two pair-steps ago goes to LDS, then the next group's fragments are
        //    read (LDS executes a wave's accesses in order: the first reader of a refilled buffer is issued behind its last store)
        // W8_HG (defined above): where a pair-step's four halo requests and stores sit, and how many request groups are in flight
        constexpr int HG = W8_HG, RG = HG == 1 ? 1 : 2;
        constexpr int sq_lo = HG == 0 ? ct : (HG == 1 ? (ct < 2 ? 2 * ct : 4) : (ct < 2 ? ct : (ct == 2 ? 2 : 4)));
        constexpr int sq_hi = HG == 0 ? ct + 1 : (HG == 1 ? (ct < 2 ? 2 * ct + 2 : 4) : (ct < 2 ? ct + 1 : 4));
        constexpr int st_base = u <= 3 ? u * 4 : (u >= 5 ? (u - 5) * 4 : 99);
        constexpr int st_slot = HG == 3 ? slot3_of_store(u) : (u <= 3 ? (u & 1) : ((u - 5) & 1));
        constexpr int n_st = (sq_lo < sq_hi && st_base + sq_lo < WN_NDMA) + (sq_lo + 1 < sq_hi && st_base + sq_lo + 1 < WN_NDMA);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q >= sq_lo && q < sq_hi && st_base + q < WN_NDMA) {
            if constexpr (W8_ABL & 4) asm volatile("" :: "v"(hs8[st_slot][q].x), "v"(hs8[st_slot][q].w));
            else halo_store8(u <= 3 ? 1 : 0, st_base + q, hs8[st_slot][q]);
          }
        if constexpr (!(W8_ABL & 8)) read_B8(std::integral_constant<int, un>{}, std::integral_constant<int, ctn>{}, B8h[bset ^ 1], B8f[bset ^ 1]);
        else asm volatile("" : "+v"(B8h[bset ^ 1][0]), "+v"(B8h[bset ^ 1][1]), "+v"(B8f[bset ^ 1]));
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A8h[aset][rt][0]), __builtin_bit_cast(f16x8, B8h[bset][0]),
                                                               acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n_st > 0 && !(W8_ABL & 4)) __builtin_amdgcn_sched_group_barrier(0x200, n_st, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- pass 1 (fp16, step 2u + 1): groups 2 and 3 request two halo pieces of a later chunk each, from the offsets read out of
        //    the table one group earlier -- BEHIND the weight loads of groups 0 and 1 in the memory queue: loads return in order,
        //    and the wait for the weights at the next pair-step must not cover a halo request (an HBM access) issued just before them
        constexpr int ld_base = base_of(u), ldn_base = base_of(un);
        constexpr int ld_slot = HG == 3 ? slot3_of_request(u) : (u == 0 ? 0 : (u == 1 ? 1 : ((u - 3) & 1)));      // HG < 3: alternates over the eight requesting pair-steps 0 1 3 4 5 6 7 8
        // entries [rq_lo, rq_hi) are requested by this group, [rn_lo, rn_hi) by the next one (its offsets are read out of the table now)
        constexpr int rq_lo = HG == 0 ? (ct >= 2 ? 2 * (ct - 2) : 4) : (ct == RG ? 0 : 4), rq_hi = HG == 0 ? (ct >= 2 ? 2 * (ct - 2) + 2 : 4) : 4;
        constexpr int rn_lo = HG == 0 ? (ctn >= 2 ? 2 * (ctn - 2) : 4) : (ctn == RG ? 0 : 4), rn_hi = HG == 0 ? (ctn >= 2 ? 2 * (ctn - 2) + 2 : 4) : 4;
        constexpr int n_ld = (rq_lo < rq_hi ? ((ld_base + rq_lo < WN_NDMA) + (ld_base + rq_lo + 1 < WN_NDMA && rq_lo + 1 < rq_hi) +
                                               (ld_base + rq_lo + 2 < WN_NDMA && rq_lo + 2 < rq_hi) + (ld_base + rq_lo + 3 < WN_NDMA && rq_lo + 3 < rq_hi)) : 0);
        constexpr int n_rd = (rn_lo < rn_hi ? ((ldn_base + rn_lo < WN_NDMA) + (ldn_base + rn_lo + 1 < WN_NDMA && rn_lo + 1 < rn_hi) +
                                               (ldn_base + rn_lo + 2 < WN_NDMA && rn_lo + 2 < rn_hi) + (ldn_base + rn_lo + 3 < WN_NDMA && rn_lo + 3 < rn_hi)) : 0);
        if constexpr (n_ld > 0) {
          constexpr int which = HG == 3 ? (u == 0 ? 1 : (u <= 5 ? 2 : 3)) : (u <= 1 ? 1 : (u <= 6 ? 2 : 3));
          const uint4* cb = tbase + (int64_t)(which == 1 ? cn1 : (which == 2 ? cn2 : cn3)) * 16 * Ph;
          const uint32_t om = which == 1 ? ~0u : (which == 2 ? m2 : m3);      // 0: every lane reads entry 0 of the (clamped) chunk -- one resident line
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (i < n_ld) hs8[ld_slot][rq_lo + i] = wn_gload16((W8_ABL & 2) ? tbase + lane : cb + (off_next[i] & om));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (i < n_rd) off_next[i] = otab[(ldn_base + rn_lo + i) * 64];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A8h[aset][rt][1]), __builtin_bit_cast(f16x8, B8h[bset][1]),
                                                               acc[rt][ct], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n_rd > 0) __builtin_amdgcn_sched_group_barrier(0x100, n_rd, 0);
        if constexpr (n_ld > 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n_ld > 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n_ld > 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (n_ld > 3) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        // -- pass 2 (fp8, both steps; 64 cycles per MFMA): in the first two groups the 16 weight pieces of the next pair-step
#ifndef W8_NA
#define W8_NA 2
#endif
        constexpr int NA = HG == 1 ? 1 : W8_NA;      // groups that carry the 16 weight loads (1 = all in group 0)
        if constexpr (ct < NA) {
          const int pn = p0 + u + 1 < npairs ? p0 + u + 1 : npairs - 1;      // clamped: the redundant tail request is never used
          load_A8(pn, aset ^ 1, ct * (4 / NA), 4 / NA);
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
          if constexpr (F6)
            acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8f[aset][rt], B8f[bset], acc[rt][ct], 2, 2, 0, A8f[aset][rt][6], 0,
                                                                          B8f[bset][6]);
          else
            acc[rt][ct] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A8f[aset][rt], B8f[bset], acc[rt][ct], 0, 0, 0, scale_a, 0, scale_b);
        if constexpr (ct < NA) {
#pragma unroll
          for (int i_ = 0; i_ < 4; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 8 / NA / 2, 0);      // 16 / NA loads over the 4 MFMAs
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto pair_step = [&](auto uc) {
        group(uc, std::integral_constant<int, 0>{});
        group(uc, std::integral_constant<int, 1>{});
        group(uc, std::integral_constant<int, 2>{});
        group(uc, std::integral_constant<int, 3>{});
      };
      pair_step(std::integral_constant<int, 0>{}); pair_step(std::integral_constant<int, 1>{}); pair_step(std::integral_constant<int, 2>{});
      pair_step(std::integral_constant<int, 3>{}); pair_step(std::integral_constant<int, 4>{}); pair_step(std::integral_constant<int, 5>{});
      pair_step(std::integral_constant<int, 6>{}); pair_step(std::integral_constant<int, 7>{}); pair_step(std::integral_constant<int, 8>{});
      (void)cn1;
    };
    for (int c0 = 0; c0 < nchunk; c0 += 4) {
      body(std::integral_constant<int, 0>{}, c0);
      if (c0 + 2 >= nchunk) break;
      body(std::integral_constant<int, 1>{}, c0 + 2);
    }
  } else {
  bf16x8 Ar[3][8];      // weights  [step % 3][rt * 2 + plane]
    bf16x8 Bf[2][8];      // halo     [step parity][ct * 2 + plane]
    uint4 hst[2][3];      // halo pieces on their way from global memory to LDS: [tap parity][piece of the step]
  
    // ---- prologue: chunk 0 of the halo (15 pieces, all requested before the first is stored), weights of steps 0 and 1; as in the
    // f16f8 prologue the order is fenced: weight requests, offset k -> request k, the accumulators zeroed while the requests fly ----
    {
      uint4 h0[WN_NDMA];
      if constexpr (W8_PRO_FENCE) {
        load_A(0, Ar[0]);
        load_A(nsteps > 1 ? 1 : 0, Ar[1]);
        __builtin_amdgcn_sched_barrier(0);
      }
  #pragma unroll
      for (int k = 0; k < WN_NDMA; ++k) {
        h0[k] = halo_load(0, k);
        if constexpr (W8_PRO_FENCE) __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (W8_PRO_FENCE) {
        zero_acc();
        __builtin_amdgcn_sched_barrier(0);
      } else {
        load_A(0, Ar[0]);
        load_A(nsteps > 1 ? 1 : 0, Ar[1]);
      }
      if constexpr (ABL & 4) load_A(nsteps > 2 ? 2 : 0, Ar[2]);      // timing only: three real weight sets, reused for every step
  #pragma unroll
      for (int k = 0; k < WN_NDMA; ++k) halo_store(0, k, h0[k]);
      if constexpr (ABL & 2) {      // timing only: real data in both buffers, no halo traffic after this
  #pragma unroll
        for (int k = 0; k < WN_NDMA; ++k) halo_store(1, k, h0[k]);
      }
    }
    read_B(0, 0, Bf[0], true);
    mark(1);
  
    // ---- main loop: two chunks (18 steps) per iteration so that every register-set index is a compile-time constant ----
    // No LDS-DMA: an LDS-DMA instruction costs the issuing wave 150-230 cycles (measured; with one wave per SIMD nobody else
    // feeds the matrix pipe meanwhile), a plain load + ds_write_b128 a fraction of that, and hipcc counts plain loads exactly.
    // Step s = 48 MFMAs (three passes a_lo*b_hi, a_hi*b_lo, a_hi*b_hi over the 16 accumulator tiles):
    //   pass 0   || 8 weight loads of step s+2, 8 ds_reads of the halo fragments of step s+1
    //   pass 1-2 || taps 0..4: 3 halo pieces of the NEXT chunk requested (after this step's weight loads, so that the wait
    //               for those weights two steps later does not cover them); taps 2..6: the 3 pieces requested two steps
    //               earlier stored to the other halo buffer at the END of the step -- ~2.7 steps (2 us) after the request.
    // The other halo buffer is free from tap 0 on (its last reads, issued at tap 7 of the chunk before, returned at tap 8);
    // its first reader is the fragment read of tap 8, two steps after the last store (LDS executes a wave's accesses in order).
  #define WN_MFMA(PASS, IDX, a, bq)                                                                                       \
    acc[(IDX) >> 2][(IDX) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[((IDX) >> 2) * 2 + ((PASS) == 0 ? 1 : 0)],   \
                                                                         bq[((IDX) & 3) * 2 + ((PASS) == 1 ? 1 : 0)],    \
                                                                         acc[(IDX) >> 2][(IDX) & 3], 0, 0, 0)
    for (int c0 = 0; c0 < nchunk; c0 += 2) {
  #pragma unroll
      for (int u = 0; u < 18; ++u) {
        const int tap = u % 9, cpar = u / 9;             // chunk c0 + cpar lives in halo buffer cpar
        const int s = c0 * 9 + u;
        const int sw = s + 2 < nsteps ? s + 2 : nsteps - 1;                     // clamped: the redundant tail requests are never used
        const int cn = c0 + cpar + 1 < nchunk ? c0 + cpar + 1 : nchunk - 1;     // clamped likewise
        bf16x8 (&Aw)[8] = Ar[u % 3], (&Bc)[8] = Bf[u & 1], (&Bn)[8] = Bf[(u + 1) & 1];
        if constexpr (!(ABL & 4)) load_A(sw, Ar[(u + 2) % 3]);
        if (tap < 8) read_B(tap + 1, cpar, Bn, false);
        else read_B(0, cpar ^ 1, Bn, false);
        // (MFMA order inside a step measured neutral, profiles/r03_wino_epilogue_ab.txt: column-tile-major passes, and the three
        // products of a tile adjacent -- the kernel is power-limited, issue order does not change the energy)
  #pragma unroll
        for (int m = 0; m < 16; ++m) WN_MFMA(0, m, Aw, Bc);
  #pragma unroll
        for (int i_ = 0; i_ < 8; ++i_) {                 // 8 LDS reads and 8 global loads under the 16 MFMAs of the first pass
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (tap >= 2 && tap <= 6 && !(ABL & 2)) {
  #pragma unroll
          for (int q = 0; q < 3; ++q) halo_store(cpar ^ 1, (tap - 2) * 3 + q, hst[tap & 1][q]);
        }
        if (tap <= 4 && !(ABL & 2)) {
  #pragma unroll
          for (int q = 0; q < 3; ++q) hst[tap & 1][q] = halo_load(cn, tap * 3 + q);
        }
  #pragma unroll
        for (int m = 0; m < 32; ++m) {
          if (m < 16) WN_MFMA(1, m, Aw, Bc); else WN_MFMA(2, m - 16, Aw, Bc);
        }
        // inside the region: the 3 loads early (one per 2 MFMAs), the 3 stores behind the last MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 20, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  #undef WN_MFMA
  }

  mark(2);
  // ---- epilogue: the four frequencies of an output pair meet through LDS, one 32-row tile per round ----------------------
  if constexpr (ABL & 16) {                              // timing only: keep the accumulators alive, write nothing
    float keep = 0.f;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) keep += acc[rt][ct][0];
    if (keep == 123.456f) A.out[tid] = keep;
    return;
  }
  // Exchange area (round 6): one region = [f 4][col 128][8 items of 16 B] with NO padding; item k (rows 4k .. 4k+3 of the round's
  // 32) of column col sits at slot k ^ sigma(col), sigma = ((col & 1) << 2) | ((col >> 1) & 3), and frequency f starts at
  // f * 1024 + {0, 0, 8, 16}[f] items: with that both sides are conflict-free (tests/test_cpu_kernel_layouts.py replays the
  // ds_write_b128 / ds_read_b128 lane groups).
  // Write side (MFMA layout): lane (j, h) holds column j of a column tile, rows 8 q + 4 h + {0..3} = item 2 q + h.
  // Read side: this wave finishes column tile `wid` (output plane z0 + wid).  A lane owns ONE position of a tile row and ONE
  // 16-byte half of its 8-channel item: lane bits [0] half, [3:1] x, [5:4] channel-group pair; its 8 slots are the 8 rows y0 + yr.
  // So a global load / store instruction covers, per channel group, the 256 contiguous bytes of a whole tile row (8 full
  // 128-byte lines per instruction).  Round 5 gave a lane 4 channels x the 8 positions of a row (y0 in one store, y1 in the
  // next): 32 separate 32-byte runs per instruction, and the epilogue was bound by the address path -- 36 loads + 32 stores
  // per wave at ~32 lines each (profiles/r06_wino_epilogue_ab.txt: removing every exposed latency from the rounds left the
  // tile time where it was).  Price: an output needs 3 of the 4 frequencies (y0 = m0 + m1 + m2, y1 = m1 - m2 - m3), so a
  // lane reads 3 items per output instead of 4 per two -- 24 ds_read_b128 per round instead of 16, from an LDS that idles.
  constexpr int XF_ITEMS = 1024, XREG_FLOATS = (4 * XF_ITEMS + 16) * 4;
  static_assert(2 * XREG_FLOATS * 4 + 2 * WN_RED * 4 <= WN_LDS_BYTES, "two exchange regions + the statistics area fit");
  float* xreg = (float*)wn_smem;
  float* red = xreg + 2 * XREG_FLOATS;
  const int rows_total = A.cout;
  float* outp = A.out + (int64_t)b * rows_total * P;
  const float* resp = A.residual ? A.residual + (int64_t)b * A.res_bstride : nullptr;
  const float* biasp = A.bias ? A.bias + (int64_t)b * A.bias_bstride : nullptr;
  const bool want_stats = (ABL & 128) ? false : A.stats != nullptr;
  const int half = lane & 1, xq = (lane >> 1) & 7, cgp = lane >> 4;      // read side
  const int cq = 2 * cgp + half, prr = xq >> 1;
  const bool odd = xq & 1;                                                // y1 lanes
  const float sgn = odd ? -1.f : 1.f;
  // positions: plane z0 + wid, row y0 + yr, x = x0 + xq
  const int64_t gp0 = ((int64_t)(z0 + wid) * H + y0) * W + x0 + xq;
  auto flush_stats = [&](int r) {     // after the barrier that follows round r's red[] writes: 64 fp64 atomics
    if (tid < 64) {
      const int ch = tid >> 1, which = tid & 1;
      const float* rb = red + (r & 1) * WN_RED;
      float sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) sum += rb[(k * 32 + ch) * 2 + which];
      const int row = rtb * 128 + r * 32 + ch;
      atomicAdd(A.stats + ((int64_t)b * rows_total + row) * 2 + which, (double)sum);
    }
  };
  // bias of all four rounds and the residual of rounds 0 and 1 are requested here, before the first barrier; round r + 2's
  // residual at the end of round r, into the registers round r has just consumed (two rounds = ~5 k cycles ahead of its use).
  // Everything is straight-line code, so hipcc counts the loads exactly (a wait never covers a later request or the round's own
  // stores).  RES is a template parameter for the same reason: behind a run-time `if (residual)` the waits degrade to vmcnt(0).
  // (All four rounds' residual up front = 128 registers next to the 96 of a round's exchange reads: 65 spilled.)
  const bool has_bias = biasp != nullptr;
  f32x4 pbias[4], pres[RES ? 2 : 1][8];
  auto prefetch_res = [&](int r) {
    if constexpr (RES) {
      const float* rp = resp + ((int64_t)(rtb * 16 + r * 4 + cgp) * P + gp0) * 8 + 4 * half;
#pragma unroll
      for (int yr = 0; yr < 8; ++yr) pres[r & 1][yr] = *(const f32x4*)(rp + (int64_t)yr * W * 8);
    }
  };
  {
    const float* bsrc = has_bias ? biasp : (const float*)A.wpk;      // always a valid address: the value is dropped at its use
#pragma unroll
    for (int r = 0; r < 4; ++r) pbias[r] = *(const f32x4*)(bsrc + rtb * 128 + r * 32 + 4 * cq);
    prefetch_res(0);
    prefetch_res(1);
  }
  auto row_sum8 = [](float v) {       // over the 8 lanes of a DPP row with the same lane & 1 (row_ror 8, 4, 2)
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xF, 0xF, true));
    return v;
  };
  // write side: item addresses of this lane's four q (the xor term), + ct * 32 columns as an immediate
  const int sj = ((j & 1) << 2) | ((j >> 1) & 3);
  const int wbase_items = wid * XF_ITEMS + (wid == 2 ? 8 : (wid == 3 ? 16 : 0)) + j * 8;
  auto write_round = [&](int r) {
    float* xw = xreg + (r & 1) * XREG_FLOATS;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float* pq = xw + (wbase_items + ((2 * q + h) ^ sj)) * 4;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[r][ct][q * 4 + e];
        *(f32x4*)(pq + ct * 32 * 8 * 4) = v;
      }
    }
  };
  // read side: column wid * 32 + yr * 4 + prr; sigma(col) = ((prr & 1) << 2) | ((yr & 1) << 1) | (prr >> 1)
  const int rcol_items = (wid * 32 + prr) * 8;
  const int rx0 = cq ^ (((prr & 1) << 2) | (prr >> 1)), rx1 = rx0 ^ 2;                       // rows yr even / odd
  const int offA = odd ? 2 * XF_ITEMS + 8 : 0, offC = odd ? 3 * XF_ITEMS + 16 : 2 * XF_ITEMS + 8;   // y0: m0, m1, m2; y1: m2, m1, m3
  __syncthreads();                                       // every wave is done with its private buffers (the exchange area aliases them)
  mark(3);
#ifndef W8_EPI_EARLYW
#define W8_EPI_EARLYW 1     // A/B: 0 = round r + 1's accumulators go to LDS behind round r's combine and stores (their latency then sits in front of the barrier)
#endif
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float* xr = xreg + (r & 1) * XREG_FLOATS;
    if (!W8_EPI_EARLYW || r == 0) write_round(r);
    __syncthreads();
    f32x4 mA[8], mB[8], mC[8];
#pragma unroll
    for (int yr = 0; yr < 8; ++yr) {
      const float* pb = xr + (rcol_items + yr * 32 + ((yr & 1) ? rx1 : rx0)) * 4;
      mA[yr] = *(const f32x4*)(pb + offA * 4);
      mB[yr] = *(const f32x4*)(pb + XF_ITEMS * 4);
      mC[yr] = *(const f32x4*)(pb + offC * 4);
    }
    // the statistics of the round before (wave 0 only: 4 LDS reads + 64 fp64 atomics) go BEHIND this round's reads in the LDS queue
    // (issued first they delayed wave 0 -- and with it the next barrier -- by their latency every round) and in front of the early write
    if (want_stats && r > 0) flush_stats(r - 1);
    if (W8_EPI_EARLYW && r < 3) {
      // the other region's last readers (round r - 1) are behind this round's barrier: round r + 1's accumulators follow this
      // round's reads into the LDS queue, and their write latency runs under the combine and the stores instead of in front of
      // the next barrier (LDS executes a wave's accesses in order: the reads return first)
      __builtin_amdgcn_sched_barrier(0);
      write_round(r + 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    float* op = outp + ((int64_t)(rtb * 16 + r * 4 + cgp) * P + gp0) * 8 + 4 * half;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bv = has_bias ? pbias[r] : zero4;
    f32x4 s1 = zero4, s2 = s1;
#pragma unroll
    for (int yr = 0; yr < 8; ++yr) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        // y0 = (m1 + m0) + m2, y1 = (m1 - m2) - m3: one code path, the sign is the lane's (exact: fma with +-1)
        const float y = __builtin_fmaf(sgn, mC[yr][e], __builtin_fmaf(sgn, mA[yr][e], mB[yr][e]));
        if constexpr (F8) o[e] = y * descale + bv[e];       // the accumulators hold 2^sw x the products (the weights' pre-scale): an exact power of two
        else o[e] = y + bv[e];
      }
      if constexpr (RES) o += pres[r & 1][yr];
      *(f32x4*)(op + (int64_t)yr * W * 8) = o;
      s1 += o;
      s2 += o * o;
    }
    if (r < 2) prefetch_res(r + 2);
    if (want_stats) {
      f32x4 a1, a2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { a1[e] = row_sum8(s1[e]); a2[e] = row_sum8(s2[e]); }
      if ((lane & 14) == 0) {         // one lane per channel quad: 8 consecutive floats of red[]
        float* rb = red + (r & 1) * WN_RED + (wid * 32 + 4 * cq) * 2;
        const f32x4 w0 = {a1[0], a2[0], a1[1], a2[1]}, w1 = {a1[2], a2[2], a1[3], a2[3]};
        *(f32x4*)rb = w0;
        *(f32x4*)(rb + 4) = w1;
      }
    }
    mark(4 + r);
  }
  if (want_stats) {
    __syncthreads();
    flush_stats(3);
  }
  if constexpr (ABL & 128) {      // stamps [workgroup][wave][10] of the first 1024 workgroups (dispatch order)
    const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
#ifdef W8_STAMPS_LATE      // the LAST 1024 workgroups in dispatch order (steady state) instead of the cold first generation
    const unsigned total = gridDim.x * gridDim.y;
    const unsigned lin0 = total > 1024u ? total - 1024u : 0u;
#else
    const unsigned lin0 = 0u;
#endif
    if (lin >= lin0 && lin - lin0 < 1024u && lane == 0) {
      uint64_t* dst = (uint64_t*)A.stats + ((size_t)(lin - lin0) * 4 + wid) * 12;
      // the stores are waited for by the end of the kernel, so the last stamp is taken before them
      stamp[8] = __builtin_amdgcn_s_memtime();
      stamp[11] = __builtin_amdgcn_s_memrealtime();
      stamp[9] = (uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 4) |                 // HW_REG_HW_ID (CU / SE of the wave)
                 ((uint64_t)(uint32_t)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);        // HW_REG_XCC_ID
#pragma unroll
      for (int k = 0; k < 12; ++k) dst[k] = stamp[k];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
extern "C" int64_t md_wino_operand_bytes(int32_t batch, int32_t cin, int32_t D, int32_t H, int32_t W) {
  if (batch <= 0 || cin <= 0 || (cin & 7) || D <= 0 || H <= 0 || W <= 0 || (W & 1)) return MD_ERR_BAD_ARG;
  return (int64_t)batch * (cin / 8) * 8 * ((int64_t)D * H * (W / 2)) * 16;
}

extern "C" int md_wino_prep(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu,
                            int32_t ups, void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, float drop_p,
                            uint64_t drop_seed, void* stream) {
  if (!x1 || !t_out || batch <= 0 || c1 <= 0 || c2 < 0 || (c1 & 7) || (c2 & 7) || (c2 > 0 && !x2)) return MD_ERR_BAD_ARG;
  if (silu && !ac) return MD_ERR_BAD_ARG;      // SiLU is applied together with the folded GroupNorm affine only
  if (!(drop_p >= 0.f && drop_p < 1.f) || (drop_p > 0.f && (ups || c2 > 0))) return MD_ERR_BAD_ARG;
  if (D <= 0 || H <= 0 || W <= 0 || (W & 1) || (ups && ((D | H | W) & 1))) return MD_ERR_BAD_ARG;
  const int64_t n = (int64_t)batch * ((c1 + c2) / 8) * D * H * (W / 2);
  const int64_t blocks = (n + 255) / 256;
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x1, x2, c1, c2, ac, silu,
                     ups, (uint4*)t_out, batch, D, H, W, md_drop_thr16(drop_p), 1.0f / (1.0f - drop_p), drop_seed);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int64_t md_wino_weight_bytes(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  return (int64_t)cout * cin * 36 * 4;
}

extern "C" int md_wino_pack_weights(const float* w, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                                    int32_t flip, void* stream) {
  if (!w || !wpk || cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  const int64_t n = (int64_t)cout * cin * 9;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wino_pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (uint4*)wpk, cout, cin, s_row, s_k, flip);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int64_t md_wino_weight_bytes_f8(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  return (int64_t)cout * cin * 36 * 4 + 256;             // fragments + header
}

static int md_wino_pack_weights_f8_launch(bool f6, const float* w, const float* eq, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k,
                                          void* stream) {
  if (!w || !wpk || cout <= 0 || cin <= 0 || (cout % 128) || (cin % 32)) return MD_ERR_BAD_ARG;
  const int64_t n = (int64_t)cout * cin * 9;
  uint32_t* hdr = (uint32_t*)((unsigned char*)wpk + n * 16);
  MD_HIP_CLEAR_ERROR();
  hipError_t e = hipMemsetAsync(hdr, 0, 256, (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  const int64_t nel = (int64_t)cout * cin * 27;
  int ab = (int)((nel + 255) / 256);
  if (ab > 1024) ab = 1024;
  hipLaunchKernelGGL(md_wino_amax_kernel, dim3((unsigned)ab), dim3(256), 0, (hipStream_t)stream, w, cout, cin, s_row, s_k, hdr, eq);
  MD_HIP_CHECK_LAUNCH();
  if (f6)
    hipLaunchKernelGGL(md_wino_pack_weights_f6_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wpk,
                       cout, cin, s_row, s_k, hdr, eq);
  else
    hipLaunchKernelGGL(md_wino_pack_weights_f8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wpk,
                       cout, cin, s_row, s_k, hdr, eq);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_wino_pack_weights_f8(const float* w, const float* eq, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, void* stream) {
  return md_wino_pack_weights_f8_launch(false, w, eq, wpk, cout, cin, s_row, s_k, stream);
}

extern "C" int md_wino_pack_weights_f6(const float* w, const float* eq, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, void* stream) {
  return md_wino_pack_weights_f8_launch(true, w, eq, wpk, cout, cin, s_row, s_k, stream);
}

static int md_conv3_wino_f8_launch(bool f6, const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                                   const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                                   int32_t cout, int32_t D, int32_t H, int32_t W, void* stream, float out_scale = 1.0f,
                                   const uint32_t* amax = nullptr) {
  if (!t_in || !wpk || !out || batch <= 0) return MD_ERR_BAD_ARG;
  if (cin <= 0 || cout <= 0 || (cin % 32) || (cout % 128)) return MD_ERR_UNSUPPORTED;
  if (D <= 0 || H <= 0 || W <= 0 || (D % WN_TZ) || (H % WN_TY) || (W % WN_TX)) return MD_ERR_UNSUPPORTED;
  if ((int64_t)D * H * W * 8 >= (int64_t)1 << 31) return MD_ERR_UNSUPPORTED;
  WnArgs a;
  a.T = (const uint4*)t_in; a.wpk = (const uint4*)wpk; a.out = out; a.bias = bias; a.residual = residual; a.stats = stats;
  a.hdr = (const float*)((const unsigned char*)wpk + (int64_t)cout * cin * 36 * 4);
  a.out_scale = out_scale; a.amax = amax;
  a.bias_bstride = bias_bstride; a.res_bstride = res_bstride;
  a.batch = batch; a.cin = cin; a.cout = cout; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / WN_TZ) * (H / WN_TY) * (W / WN_TX);
  const dim3 grid((unsigned)(tiles * batch), (unsigned)(cout / 128));
  MD_HIP_CLEAR_ERROR();
#ifdef W8_STAMPS      // A/B build only (tools/bench_wino.py --f8 --stamps): `stats` receives the per-wave s_memtime stamps, no statistics
  constexpr int ABL_ = 128;
#else
  constexpr int ABL_ = 0;
#endif
#define WN_LAUNCH8(F6_, RES_) hipLaunchKernelGGL((md_conv3_wino_kernel<ABL_, true, F6_, RES_>), grid, dim3(WN_THREADS), 0, (hipStream_t)stream, a)
  if (f6) { if (residual) WN_LAUNCH8(true, true); else WN_LAUNCH8(true, false); }
  else { if (residual) WN_LAUNCH8(false, true); else WN_LAUNCH8(false, false); }
#undef WN_LAUNCH8
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int md_conv3_wino_f8(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                                const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                                int32_t cout, int32_t D, int32_t H, int32_t W, void* stream) {
  return md_conv3_wino_f8_launch(false, t_in, wpk, out, bias, bias_bstride, residual, res_bstride, stats, batch, cin, cout, D, H, W, stream);
}

extern "C" int md_conv3_wino_f6(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                                const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                                int32_t cout, int32_t D, int32_t H, int32_t W, void* stream) {
  return md_conv3_wino_f8_launch(true, t_in, wpk, out, bias, bias_bstride, residual, res_bstride, stats, batch, cin, cout, D, H, W, stream);
}

extern "C" int md_conv3_wino_f6_scaled(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                                       const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                                       int32_t cout, int32_t D, int32_t H, int32_t W, float out_scale, const uint32_t* amax_bits, void* stream) {
  if (!(out_scale > 0.f)) return MD_ERR_BAD_ARG;
  return md_conv3_wino_f8_launch(true, t_in, wpk, out, bias, bias_bstride, residual, res_bstride, stats, batch, cin, cout, D, H, W, stream, out_scale,
                                 amax_bits);
}

extern "C" int md_conv3_wino(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                             const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin,
                             int32_t cout, int32_t D, int32_t H, int32_t W, int32_t variant, void* stream) {
  if (!t_in || !wpk || !out || batch <= 0) return MD_ERR_BAD_ARG;
  if (cin <= 0 || cout <= 0 || (cin % 32) || (cout % 128)) return MD_ERR_UNSUPPORTED;
  if (D <= 0 || H <= 0 || W <= 0 || (D % WN_TZ) || (H % WN_TY) || (W % WN_TX)) return MD_ERR_UNSUPPORTED;
  if ((int64_t)D * H * W * 8 >= (int64_t)1 << 31) return MD_ERR_UNSUPPORTED;       // 32-bit halo offsets (16 Ph items)
  WnArgs a;
  a.T = (const uint4*)t_in; a.wpk = (const uint4*)wpk; a.out = out; a.bias = bias; a.residual = residual; a.stats = stats;
  a.hdr = nullptr;
  a.out_scale = 1.0f; a.amax = nullptr;
  a.bias_bstride = bias_bstride; a.res_bstride = res_bstride;
  a.batch = batch; a.cin = cin; a.cout = cout; a.D = D; a.H = H; a.W = W;
  const int tiles = (D / WN_TZ) * (H / WN_TY) * (W / WN_TX);
  MD_HIP_CLEAR_ERROR();
  const dim3 grid((unsigned)(tiles * batch), (unsigned)(cout / 128));
#define WN_LAUNCH(A_) hipLaunchKernelGGL((md_conv3_wino_kernel<A_, false, false, true>), grid, dim3(WN_THREADS), 0, (hipStream_t)stream, a)
  if (variant != 0 && residual == nullptr) return MD_ERR_UNSUPPORTED;      // the timing-only variants are built in the residual form
  switch (variant) {
    case 0:
      if (residual) WN_LAUNCH(0);
      else hipLaunchKernelGGL((md_conv3_wino_kernel<0, false, false, false>), grid, dim3(WN_THREADS), 0, (hipStream_t)stream, a);
      break;
#ifdef MD_BUILD_ABLATIONS      // timing-only variants for tools/bench_wino.py
    case 1: WN_LAUNCH(1); break;
    case 2: WN_LAUNCH(2); break;
    case 4: WN_LAUNCH(4); break;
    case 6: WN_LAUNCH(6); break;
    case 22: WN_LAUNCH(22); break;
    case 9: WN_LAUNCH(9); break;
    case 16: WN_LAUNCH(16); break;
    case 25: WN_LAUNCH(25); break;
    case 32: WN_LAUNCH(32); break;
    case 64: WN_LAUNCH(64); break;
    case 128: WN_LAUNCH(128); break;
#endif
    default: return MD_ERR_UNSUPPORTED;
  }
#undef WN_LAUNCH
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
