// Weight gradient of the 3x3x3 stride-1 convolutions in the Winograd F(2,3)-along-w domain (autograd of
// lib/diffusion/models/layers.py:118-124 for the layers whose forward / data gradient run through csrc/conv3_wino.hip).
//
// For an output pair (x = 2i, 2i+1) with output gradients dy0 dy1 and the activated inputs d0..d3 = a[2i-1 .. 2i+2]:
//     dg_k = sum_pairs (dy0 d_k + dy1 d_{k+1}),  k = 0, 1, 2                       (6 products per pair)
// is the transpose of the forward's bilinear form, so it has the transposed minimal algorithm (4 products per pair):
//     t = B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)        -- exactly the operand T the forward conv consumed
//     u = (dy0, dy0 + dy1, dy0 - dy1, dy1)                    -- md_wino_prep_dual's second output
//     n_f = sum_pairs u_f t_f ;   dg0 = n0 + (n1 + n2)/2,  dg1 = (n1 - n2)/2,  dg2 = (n1 + n2)/2 - n3.
// The 27-tap contraction over positions becomes 9 (kd, kh) row shifts x 4 frequencies over HALF the columns = 2/3 of the
// matrix-core work of csrc/wgrad.hip, and the w shift of the taps is gone: a (kd, kh) tap is a plain ROW shift of T.
//
// Operands stay in the channel-innermost layout the conv kernels use, [B][C/8][f 4][plane 2][D][H][W/2][8 bf16] (hi / lo
// planes): no PB16 re-layout pass.  The contraction index of an MFMA (8 consecutive k per lane) is the PAIR index here,
// which is strided in that layout; the LDS image of a 16-pair element is [plane][8-pair segment][32-channel block][pair 8]
// [32 ch] (the four 16-byte chunks of a 64-byte [pair] row XOR-swizzled by pair >> 1 so that the staging stores are
// conflict-free too) and fragments come out of it with ds_read_b64_tr_b16 (gfx950 transpose read: a 16-lane group reads a
// [4 pair][16 channel] block and each lane receives one channel's 4 pairs; semantics pinned by tools/probes/tr_probe.hip).
//
// Structure (as md_conv3_wino: nothing is shared between the waves of a workgroup, so the main loop has NO barrier): one wave =
// one frequency, one kd, a 64 co x 64 ci tile and the THREE kh taps (192 accumulator registers) over a
// range of (sample, z) planes.  Its operands stream through 32 KB of LDS private to the wave in elements of 16 pairs (one MFMA
// k-step; a row is NH = 2 (64^3) or 1 (32^3) elements): U double-buffered, T in a ring of 2 NH + 2 elements because element n
// of dY meets T[n - NH], T[n], T[n + NH] (rows y - 1, y, y + 1 of plane z + kd - 1).  Row -1 / H of a plane is an all-zero T
// element shared by consecutive planes and paired with an all-zero U element, so the whole range is ONE uniform stream of
// identical steps:  { store the element pair requested three steps ago ; request the pair three steps ahead (plain 16-byte loads
// into a ring of four register sets: ~32 KB in flight per wave, the memory round trip is ~1.5 us) ; 36 MFMAs on U[n] x T[n - NH], T[n], T[n + NH], the 32 transpose reads of the NEXT
// operands issued between them }.  A wave's LDS accesses execute in order, which is all the synchronisation there is.
// Partial sums go to a workspace; md_wgrad_wino_reduce sums the K ranges in a fixed order, applies the output transform and
// accumulates into dW (deterministic).
//
// Arithmetic: bf16x3 (lo*hi + hi*lo + hi*hi, fp32 accumulate), like the forward.
#include "md_common.h"
#include <cstdlib>

namespace {

constexpr int WW_THREADS = 256;
constexpr int WW_ELB = 4096;        // LDS bytes of one element: [plane 2][segment 2][32-ch block 2][pair 8][64 B]
typedef short ww_v4i16 __attribute__((ext_vector_type(4)));

struct WwArgs {
  const uint4* U;      // transformed dY    [B][co/8][4][2][Ph] items of 16 B
  const uint4* T;      // transformed input [B][ci/8][4][2][Ph]
  float* partial;      // [ksplit][f 4][kd 3][kh 3][co][ci]
  int batch, co, ci, D, H, Wp;
  int co_tiles, ci_tiles, ksplit;      // tiles of 64 channels
};

__device__ __forceinline__ uint4 ww_gload16(const uint4* p) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  return __builtin_bit_cast(uint4, *(__attribute__((address_space(1))) const u32x4*)(uintptr_t)p);
}

// 8 consecutive pairs of one channel: two transpose reads (pairs 0-3 and 4-7 of the lane's segment; d = byte distance of the
// second read's lane address from the first's: 256 +- the swizzle term)
__device__ __forceinline__ bf16x8 ww_frag(const unsigned char* p, int d) {
  typedef __attribute__((address_space(3))) ww_v4i16 lds_v4;
  const ww_v4i16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p));
  const ww_v4i16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(p + d));
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3]; r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}

// Stream element cursor: element (valid plane v = (b, zi), j, h): j = 0 the zero row between planes, j >= 1 row j - 1; h = half
// row.  Branch-free (scalar selects): the main loop is one basic block per step.
template <int NH>
struct WwCursor {
  int v, b, zi, j, h;
  __device__ void next(int H, int nz) {
    const int h1 = h + 1;
    const int wh = h1 == NH;
    h = wh ? 0 : h1;
    const int j1 = j + wh;
    const int wj = j1 > H;
    j = wj ? 0 : j1;
    v += wj;
    const int z1 = zi + wj;
    const int wz = z1 == nz;
    zi = wz ? 0 : z1;
    b += wz;
  }
};

__device__ const uint4 ww_zero16 = {0u, 0u, 0u, 0u};     // source of the all-zero elements

// NH = 16-pair elements per row: 2 (W = 64) or 1 (W = 32)
// DBG (timing only, results invalid; -DMD_BUILD_ABLATIONS + env MD_WW_DBG, tools/bench_wgrad_wino.py): bit 0 no global loads /
//   LDS stores in the loop, bit 1 no MFMAs, bit 2 no fragment reads (stale registers), bit 4 loads but no LDS stores, bit 5
//   LDS stores (of zeros) but no loads
template <int NH, int DBG = 0>
__global__ __launch_bounds__(WW_THREADS) void md_wgrad_wino_kernel(const WwArgs g) {
  constexpr int R = 2 * NH + 2;                    // T ring elements
  constexpr int WAVEB = (R + 2) * WW_ELB;          // LDS private to a wave: T ring + 2 U elements (32 KB / 24 KB)
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WAVEB];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const tring = smem + wv * WAVEB;
  unsigned char* const ubuf = tring + R * WW_ELB;

  // ---- work item: runs of 32 consecutive items per XCD (block b lands on XCD b % 8), so the kd / tile units of one K range --
  //      which read the same rows -- share an L2
  // unit = (128 x 128 tile, f, kd); the 4 waves of a workgroup are the 64 x 64 quadrants of the tile: every row is requested
  // by two waves of the same CU (measured against "4 waves = 4 frequencies of one 64 x 64 tile": -17 % L2 requests, -4 % time)
  const int ct2 = g.ci_tiles >> 1;                 // 128-column tiles
  const int units = (g.co_tiles >> 1) * ct2 * 12;
  const int bx = blockIdx.x, slot = bx >> 3;
  const int w = (slot >> 5) * 256 + (bx & 7) * 32 + (slot & 31);
  if (w >= g.ksplit * units) return;
  const int r = w / units, u = w - r * units;
  const int kd = u % 3, f = (u / 3) & 3;
  const int tci = ((u / 12) % ct2) * 2 + (wv & 1), tco = ((u / 12) / ct2) * 2 + (wv >> 1);
  const int D = g.D, H = g.H, Wp = g.Wp;
  const int64_t Ph = (int64_t)D * H * Wp;
  // valid planes of this kd: (b, z) with 0 <= z + kd - 1 < D; index v -> b = v / nz, z = v % nz + z_first.  (Measured
  // alternative: the same plane ranges for the three kd units, invalid planes multiplied with zeros, so that they walk the
  // same U rows in lock step: the L2 does not merge their simultaneous misses -- 20 % hits instead of 50 %, 20 GB instead
  // of 12.6 GB from HBM per 128 -> 128 @ 64^3 launch, same time; a start stagger of 1-30 us changes nothing.)
  const int nz = kd == 1 ? D : D - 1, z_first = kd == 0 ? 1 : 0;
  const int NV = g.batch * nz;
  const int v0 = (int)((int64_t)NV * r / g.ksplit), v1 = (int)((int64_t)NV * (r + 1) / g.ksplit);

  // ---- per-lane constants of the staging loads: instruction q of an element = (plane, 8-pair segment); lane = (channel group,
  //      pair of the segment): 8 consecutive lanes read 128 contiguous bytes
  int64_t goff[4];
  int loff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int plane = q & 1, sg = q >> 1;
    const int cg = lane >> 3, k = lane & 7;
    goff[q] = ((int64_t)cg * 8 + f * 2 + plane) * Ph + sg * 8 + k;
    // row k of a 32-channel block is 64 B = four 16-byte chunks (cg & 3), stored at chunk ^ (k >> 1): the 8 lanes of a
    // ds_write_b128 group (k = 0..7 of one cg) then hit 8 different 16-byte bank groups instead of 2 (4-way conflict)
    loff[q] = plane * 2048 + sg * 1024 + (cg >> 2) * 512 + k * 64 + (((cg & 3) ^ (k >> 1)) * 16);
  }
  const uint4* const ubase = g.U + ((int64_t)tco * 8) * 8 * Ph;       // + b * (co/8) * 8 * Ph + row * Wp + h * 16 + goff
  const uint4* const tbase = g.T + ((int64_t)tci * 8) * 8 * Ph;
  const int64_t u_bstride = (int64_t)(g.co >> 3) * 8 * Ph, t_bstride = (int64_t)(g.ci >> 3) * 8 * Ph;
  auto u_ptr = [&](const WwCursor<NH>& c) -> const uint4* {           // c.j >= 1
    return ubase + (int64_t)c.b * u_bstride + ((int64_t)(c.zi + z_first) * H + (c.j - 1)) * Wp + c.h * 16;
  };
  auto t_ptr = [&](const WwCursor<NH>& c) -> const uint4* {
    return tbase + (int64_t)c.b * t_bstride + ((int64_t)(c.zi + z_first + kd - 1) * H + (c.j - 1)) * Wp + c.h * 16;
  };

  // ---- fragment addresses: lane = (8-pair segment s of the k-step, 16-channel half sub, i); see the file header
  const int s = lane >> 5, sub = (lane >> 4) & 1, i = lane & 15;
  const int chunk = sub * 2 + ((i & 3) >> 1);      // 16-byte chunk of the lane's 8 bytes inside the 64-byte row, before the swizzle
  const int frag_lo = s * 1024 + (i >> 2) * 64 + ((chunk ^ (i >> 3)) * 16) + (i & 1) * 8;                 // pairs 0-3: k = i >> 2
  const int frag_d = 256 + (((chunk ^ (2 + (i >> 3))) - (chunk ^ (i >> 3))) * 16);                          // pairs 4-7: k = 4 + (i >> 2)

  f32x16 acc[3][2][2];
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[kh][rt][ct][e] = 0.f;
        asm volatile("" : "+a"(acc[kh][rt][ct]));      // the 192 accumulator registers are AccVGPRs
      }

  // stream: element n = (v, j, h); nsteps elements incl. the closing zero row, rounded up to the unroll factor (the extra
  // elements are zeros on both sides)
  const int nreal = (v1 - v0) * (H + 1) * NH + NH;
  const int nsteps = (nreal + 3) / 4 * 4;
  uint4 st_u[4][4], st_t[4][4];                    // element pairs on their way from global memory to LDS (three steps ahead)
  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  // every request is 8 unconditional 16-byte loads: a zero element reads the same 16 zero bytes in every lane
  auto request = [&](int set, const WwCursor<NH>& ct, const WwCursor<NH>& cu) {
    const bool t_live = ct.j != 0 && ct.v < v1, u_live = cu.j != 0 && cu.v < v1;
    const uint4* tp = t_ptr(ct);
    const uint4* up = u_ptr(cu);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (DBG & (1 | 32)) { st_t[set][q] = zero4; st_u[set][q] = zero4; continue; }
      st_t[set][q] = ww_gload16(t_live ? tp + goff[q] : &ww_zero16);      // per-lane select: the step stays one basic block
      st_u[set][q] = ww_gload16(u_live ? up + goff[q] : &ww_zero16);
    }
  };
  auto store = [&](int set, int tslot, int uslot) {
    if (DBG & 1) return;
    if (DBG & 16) {                                  // timing only: the loads are consumed, nothing is stored
#pragma unroll
      for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(st_t[set][q].x), "v"(st_t[set][q].w), "v"(st_u[set][q].x), "v"(st_u[set][q].w));
      return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *(uint4*)(tring + tslot * WW_ELB + loff[q]) = st_t[set][q];
      *(uint4*)(ubuf + uslot * WW_ELB + loff[q]) = st_u[set][q];
    }
  };
  // fragments: A = U element (2 row tiles x hi / lo), B = one T element (2 column tiles x hi / lo)
  auto read4 = [&](const unsigned char* el, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      dst[t2 * 2] = ww_frag(el + t2 * 512 + frag_lo, frag_d);
      dst[t2 * 2 + 1] = ww_frag(el + t2 * 512 + 2048 + frag_lo, frag_d);
    }
  };

  if (v1 > v0) {
    // ---- prologue: T[-NH .. NH-1] = zeros (the first NH steps multiply the zero U elements with T[-NH ..]: no NaN from
    //      uninitialised LDS), T[NH] = first half row; U[0] = zero; request sets 0, 1, 2 = the pairs that steps 0, 1, 2 store:
    //      (T[NH+1], U[1]), (T[NH+2], U[2]), (T[NH+3], U[3])
    WwCursor<NH> ct = {v0, v0 / nz, v0 % nz, 1, 0}, cu = {v0, v0 / nz, v0 % nz, 0, 0};
    {
      request(0, ct, cu);                          // T[NH] (row 0, first half), U[0] (zero)
#pragma unroll
      for (int e = 0; e < R + 2; ++e)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *(uint4*)(tring + e * WW_ELB + loff[q]) = (DBG & 1) ? make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u) : zero4;
      store(0, NH, 0);
      ct.next(H, nz); cu.next(H, nz);
      request(0, ct, cu); ct.next(H, nz); cu.next(H, nz);  // T[NH+1], U[1]
      request(1, ct, cu); ct.next(H, nz); cu.next(H, nz);  // T[NH+2], U[2]
      request(2, ct, cu); ct.next(H, nz); cu.next(H, nz);  // T[NH+3], U[3]
    }
    bf16x8 A[2][4], Bq[2][4];                      // [buffer][tile * 2 + plane]
    int c = 0;                                     // n % R
    read4(ubuf + 0 * WW_ELB, A[0]);
    read4(tring + ((R - NH) % R) * WW_ELB, Bq[0]); // T[-NH]

#define WW_MFMA12(KH, AF, BF)                                                                                          \
    if (!(DBG & 2)) {                                                                                                 \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                   \
        acc[KH][m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[(m >> 1) * 2 + 1], BF[(m & 1) * 2], acc[KH][m >> 1][m & 1], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                   \
        acc[KH][m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[(m >> 1) * 2], BF[(m & 1) * 2 + 1], acc[KH][m >> 1][m & 1], 0, 0, 0); \
      _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                                   \
        acc[KH][m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AF[(m >> 1) * 2], BF[(m & 1) * 2], acc[KH][m >> 1][m & 1], 0, 0, 0); \
    } else {                                                                                                          \
      asm volatile("" :: "v"(AF[0]), "v"(AF[1]), "v"(AF[2]), "v"(AF[3]), "v"(BF[0]), "v"(BF[1]), "v"(BF[2]), "v"(BF[3])); \
    }

    for (int n0 = 0; n0 < nsteps; n0 += 4) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        // ---- step n = n0 + uu; c = n % R.  T[n - NH], T[n], T[n + NH] live in ring slots c - NH, c, c + NH; this step
        //      stores T[n + NH + 1] / U[n + 1] (register set n % 4) and requests T[n + NH + 4] / U[n + 4] (set (n + 3) % 4).
        //      Fragment buffers: A[n & 1] = U[n]; Bq[n & 1] holds T[n - NH] on entry.
        const int cm = c >= NH ? c - NH : c - NH + R, cp = c + NH >= R ? c + NH - R : c + NH;
        const int cw = cp + 1 >= R ? cp + 1 - R : cp + 1, cm1 = cm + 1 >= R ? cm + 1 - R : cm + 1;
        const int pb = uu & 1;
        bf16x8 (&Ac)[4] = A[pb], (&An)[4] = A[pb ^ 1], (&B0)[4] = Bq[pb], (&B1)[4] = Bq[pb ^ 1];
        const bool rd = !((DBG & 4) && n0 > 0);
        // kh = 0 on T[n - NH] || the 8 stores of this step and the reads of T[n]
        store(uu, cw, pb ^ 1);
        if (rd) read4(tring + c * WW_ELB, B1);
        WW_MFMA12(0, Ac, B0)
        if (!(DBG & 5)) {
#pragma unroll
          for (int i_ = 0; i_ < 8; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // kh = 1 on T[n] || the 8 requests two steps ahead and the reads of T[n + NH]
        request((uu + 3) % 4, ct, cu);
        ct.next(H, nz); cu.next(H, nz);
        if (rd) read4(tring + cp * WW_ELB, B0);
        WW_MFMA12(1, Ac, B1)
        if (!(DBG & 5)) {
#pragma unroll
          for (int i_ = 0; i_ < 8; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // kh = 2 on T[n + NH] || the reads of the next step's first operands U[n + 1] (stored above) and T[n + 1 - NH]
        if (rd) { read4(ubuf + (pb ^ 1) * WW_ELB, An); read4(tring + cm1 * WW_ELB, B1); }
        WW_MFMA12(2, Ac, B0)
        if (!(DBG & 4)) {
#pragma unroll
          for (int i_ = 0; i_ < 4; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
#pragma unroll
          for (int i_ = 0; i_ < 8; ++i_) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        c = c + 1 >= R ? 0 : c + 1;
      }
    }
#undef WW_MFMA12
  }

  // ---- partial sums: [r][f][kd][kh][co][ci], lane -> ci (128-byte rows)
  const int RT = g.co_tiles * 64, CT = g.ci_tiles * 64;
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float* o = g.partial + (((((int64_t)r * 4 + f) * 3 + kd) * 3 + kh) * RT + tco * 64 + rt * 32) * CT + tci * 64 + ct * 32 + l31;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
          o[(int64_t)row * CT] = acc[kh][rt][ct][e];
        }
      }
}

// dw[co*s_row + ci*s_k + ((kd*3 + kh)*3 + kw)*s_tap] += output transform of sum_r partial[r][f][kd][kh][co][ci]
// One thread = one (co, ci, kd, kh) and its 3 kw taps (measured alternatives: one thread per (co, ci) and all 27 taps as one
// contiguous run: 6x slower -- 9x fewer threads; the LDS-transposing form of md_wgrad_reduce27_kernel: 2.3x slower here --
// these layers have 128-256 channels, i.e. 256-1024 blocks of it).
__global__ void md_wgrad_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int co, int ci, int ksplit,
                                            int64_t s_row, int64_t s_k, int64_t s_tap) {
  const int64_t total = (int64_t)9 * co * ci;
  const int64_t tile = (int64_t)co * ci;           // one (f, kd, kh) slab
  const int64_t slab = 36 * tile;                  // one K range
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(id % ci);
    const int row = (int)((id / ci) % co);
    const int tap = (int)(id / tile);              // kd * 3 + kh
    float n[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
      const float* p = partial + ((int64_t)f * 9 + tap) * tile + (int64_t)row * ci + col;
      float sum = 0.f;
      for (int r = 0; r < ksplit; ++r) sum += p[r * slab];
      n[f] = sum;
    }
    const float h12 = 0.5f * (n[1] + n[2]);
    float* d = dw + row * s_row + col * s_k + (int64_t)(tap * 3) * s_tap;
    d[0] += n[0] + h12;
    d[s_tap] += 0.5f * (n[1] - n[2]);
    d[2 * s_tap] += h12 - n[3];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// md_wgrad_nin: weight gradient of a 1x1x1 NIN layer (autograd of layers.py:573-582) straight from the S16B tensors the
// backward already holds -- dW[ci][co] += sum_{sample, position} x[ci][pos] dY[co][pos] -- with the same machinery: channel-
// innermost operands, transpose reads, wave-private element streams (16 positions per step), no barrier.  Replaces
// md_to_pb16 (x and dY) + md_wgrad<taps = 1> for the ResnetBlock shortcuts: the re-layout passes were 3x the contraction.
// One wave = a 64 co x 64 ci tile over a range of elements; 12 MFMAs per 8 KB of operands: HBM / L2 bound.
struct WnArgsNin {
  const uint4* DY;     // S16B [B][co/8][2][P] items of 16 B
  const uint4* X;      // S16B [B][ci/8][2][P]
  float* partial;      // [ksplit][co][ci]
  int batch, co, ci;
  int64_t P;
  int ksplit;
};

__global__ __launch_bounds__(WW_THREADS) void md_wgrad_nin_kernel(const WnArgsNin g) {
  constexpr int WAVEB = 4 * WW_ELB;                // two U and two T elements
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WAVEB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* const tbuf = smem + wv * WAVEB;
  unsigned char* const ubuf = tbuf + 2 * WW_ELB;
  const int ct2 = g.ci >> 7;
  const int units = (g.co >> 7) * ct2;
  const int bx = blockIdx.x;
  if (bx >= g.ksplit * units) return;
  const int r = bx / units, u = bx - r * units;
  const int tci = (u % ct2) * 2 + (wv & 1), tco = (u / ct2) * 2 + (wv >> 1);
  const int64_t P = g.P;
  const int epb = (int)(P >> 4);                   // elements per sample
  const int64_t NE = (int64_t)g.batch * epb;
  const int e0 = (int)(NE * r / g.ksplit), e1 = (int)(NE * (r + 1) / g.ksplit);

  int64_t goff[4];
  int loff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int plane = q & 1, sg = q >> 1;
    const int cg = lane >> 3, k = lane & 7;
    goff[q] = ((int64_t)cg * 2 + plane) * P + sg * 8 + k;
    loff[q] = plane * 2048 + sg * 1024 + (cg >> 2) * 512 + k * 64 + (((cg & 3) ^ (k >> 1)) * 16);
  }
  const uint4* const ubase = g.DY + ((int64_t)tco * 8) * 2 * P;
  const uint4* const tbase = g.X + ((int64_t)tci * 8) * 2 * P;
  const int64_t u_bstride = (int64_t)(g.co >> 3) * 2 * P, t_bstride = (int64_t)(g.ci >> 3) * 2 * P;
  const int s = lane >> 5, sub = (lane >> 4) & 1, i = lane & 15;
  const int chunk = sub * 2 + ((i & 3) >> 1);
  const int frag_lo = s * 1024 + (i >> 2) * 64 + ((chunk ^ (i >> 3)) * 16) + (i & 1) * 8;
  const int frag_d = 256 + (((chunk ^ (2 + (i >> 3))) - (chunk ^ (i >> 3))) * 16);

  f32x16 acc[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rt][ct][e] = 0.f;

  const int nreal = e1 - e0;
  const int nsteps = (nreal + 3) / 4 * 4;
  uint4 st_u[4][4], st_t[4][4];
  // element e of the range (beyond its end: the 16 zero bytes)
  auto request = [&](int set, int e) {
    const bool live = e < e1;
    const int b = e / epb, p0 = (e - b * epb) << 4;
    const uint4* up = ubase + (int64_t)b * u_bstride + p0;
    const uint4* tp = tbase + (int64_t)b * t_bstride + p0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      st_u[set][q] = ww_gload16(live ? up + goff[q] : &ww_zero16);
      st_t[set][q] = ww_gload16(live ? tp + goff[q] : &ww_zero16);
    }
  };
  auto store = [&](int set, int slot) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *(uint4*)(ubuf + slot * WW_ELB + loff[q]) = st_u[set][q];
      *(uint4*)(tbuf + slot * WW_ELB + loff[q]) = st_t[set][q];
    }
  };
  auto read4 = [&](const unsigned char* el, bf16x8 (&dst)[4]) {
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      dst[t2 * 2] = ww_frag(el + t2 * 512 + frag_lo, frag_d);
      dst[t2 * 2 + 1] = ww_frag(el + t2 * 512 + 2048 + frag_lo, frag_d);
    }
  };
  if (nreal > 0) {
    // prologue: element 0 in LDS slot 0, sets 1, 2, 3, 0 = elements 1..4 in flight (element m travels in set m % 4)
    request(0, e0);
    store(0, 0);
    request(1, e0 + 1); request(2, e0 + 2); request(3, e0 + 3); request(0, e0 + 4);
    bf16x8 A[2][4], Bq[2][4];
    read4(ubuf, A[0]);
    read4(tbuf, Bq[0]);
    for (int n0 = 0; n0 < nsteps; n0 += 4) {
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        // step n = n0 + uu on element n (fragments in A / Bq[n & 1], LDS slot n & 1): stores element n + 1 (register set
        // (n + 1) % 4) into the other slot, requests element n + 4 into the set just stored
        const int n = n0 + uu, pb = uu & 1;
        store((uu + 1) & 3, pb ^ 1);
        request((uu + 1) & 3, e0 + n + 5);
        read4(ubuf + (pb ^ 1) * WW_ELB, A[pb ^ 1]);
        read4(tbuf + (pb ^ 1) * WW_ELB, Bq[pb ^ 1]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pb][(m >> 1) * 2 + 1], Bq[pb][(m & 1) * 2], acc[m >> 1][m & 1], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pb][(m >> 1) * 2], Bq[pb][(m & 1) * 2 + 1], acc[m >> 1][m & 1], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m >> 1][m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[pb][(m >> 1) * 2], Bq[pb][(m & 1) * 2], acc[m >> 1][m & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const int half = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      float* o = g.partial + ((int64_t)r * g.co + tco * 64 + rt * 32) * g.ci + tci * 64 + ct * 32 + l31;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * half;
        o[(int64_t)row * g.ci] = acc[rt][ct][e];
      }
    }
}

// dw[co*s_row + ci*s_k] += sum_r partial[r][co][ci]
__global__ void md_wgrad_nin_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int co, int ci, int ksplit,
                                           int64_t s_row, int64_t s_k) {
  const int64_t tile = (int64_t)co * ci;
  for (int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; id < tile; id += (int64_t)gridDim.x * blockDim.x) {
    // consecutive threads walk the FASTER dimension of dw (NIN: W[ci][co], s_row = 1): the slabs are then read with stride ci,
    // a few MB in all -- better than scattering the read-modify-writes
    int row, col;
    if (s_row <= s_k) { row = (int)(id % co); col = (int)(id / co); } else { col = (int)(id % ci); row = (int)(id / ci); }
    const float* p = partial + (int64_t)row * ci + col;
    float sum = 0.f;
    for (int r = 0; r < ksplit; ++r) sum += p[r * tile];
    dw[row * s_row + col * s_k] += sum;
  }
}
}  // namespace

extern "C" int64_t md_wgrad_wino_workspace_bytes(int32_t co, int32_t ci, int32_t ksplit) {
  if (co <= 0 || ci <= 0 || (co % 128) || (ci % 128) || ksplit <= 0) return MD_ERR_BAD_ARG;
  return (int64_t)ksplit * 36 * co * ci * 4;
}

extern "C" int md_wgrad_wino(const void* u_dy, const void* t_act, float* dw, void* workspace, int64_t workspace_bytes,
                             int32_t batch, int32_t co, int32_t ci, int32_t D, int32_t H, int32_t W, int32_t ksplit,
                             int64_t s_row, int64_t s_k, int64_t s_tap, void* stream) {
  if (!u_dy || !t_act || !dw || !workspace || batch <= 0 || ksplit <= 0) return MD_ERR_BAD_ARG;
  if (co <= 0 || ci <= 0 || (co % 128) || (ci % 128)) return MD_ERR_UNSUPPORTED;
  if (D < 2 || H < 2 || (W != 64 && W != 32)) return MD_ERR_UNSUPPORTED;      // rows of 32 or 16 pairs
  if (workspace_bytes < md_wgrad_wino_workspace_bytes(co, ci, ksplit)) return MD_ERR_BAD_ARG;
  if (ksplit > batch * (D - 1)) return MD_ERR_BAD_ARG;                         // every K range owns at least one plane
  WwArgs g;
  g.U = (const uint4*)u_dy; g.T = (const uint4*)t_act; g.partial = (float*)workspace;
  g.batch = batch; g.co = co; g.ci = ci; g.D = D; g.H = H; g.Wp = W / 2;
  g.co_tiles = co / 64; g.ci_tiles = ci / 64; g.ksplit = ksplit;
  const int64_t items = (int64_t)ksplit * (co / 128) * (ci / 128) * 12;
  const int64_t blocks = (items + 255) / 256 * 256;
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  const dim3 grid((unsigned)blocks), blk(WW_THREADS);
  int dbg = 0;
#ifdef MD_BUILD_ABLATIONS
  if (const char* e = getenv("MD_WW_DBG")) dbg = atoi(e);
#define WW_DBG_CASE(V) case V: if (W == 64) hipLaunchKernelGGL((md_wgrad_wino_kernel<2, V>), grid, blk, 0, (hipStream_t)stream, g); \
                               else hipLaunchKernelGGL((md_wgrad_wino_kernel<1, V>), grid, blk, 0, (hipStream_t)stream, g); break;
  switch (dbg) { WW_DBG_CASE(1) WW_DBG_CASE(2) WW_DBG_CASE(4) WW_DBG_CASE(5) WW_DBG_CASE(16) WW_DBG_CASE(32) WW_DBG_CASE(18) WW_DBG_CASE(34) WW_DBG_CASE(20) WW_DBG_CASE(36) default: dbg = 0; }
#undef WW_DBG_CASE
#endif
  if (dbg == 0) {
    if (W == 64) hipLaunchKernelGGL((md_wgrad_wino_kernel<2>), grid, blk, 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((md_wgrad_wino_kernel<1>), grid, blk, 0, (hipStream_t)stream, g);
  }
  MD_HIP_CHECK_LAUNCH();
  const int64_t total = (int64_t)9 * co * ci;
  int rb = (int)((total + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(md_wgrad_wino_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace,
                     dw, co, ci, ksplit, s_row, s_k, s_tap);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int64_t md_wgrad_nin_workspace_bytes(int32_t co, int32_t ci, int32_t ksplit) {
  if (co <= 0 || ci <= 0 || (co % 128) || (ci % 128) || ksplit <= 0) return MD_ERR_BAD_ARG;
  return (int64_t)ksplit * co * ci * 4;
}

extern "C" int md_wgrad_nin(const void* dy_s16, const void* x_s16, float* dw, void* workspace, int64_t workspace_bytes,
                            int32_t batch, int32_t co, int32_t ci, int64_t P, int32_t ksplit, int64_t s_row, int64_t s_k,
                            void* stream) {
  if (!dy_s16 || !x_s16 || !dw || !workspace || batch <= 0 || ksplit <= 0 || P <= 0) return MD_ERR_BAD_ARG;
  if ((co % 128) || (ci % 128) || co <= 0 || ci <= 0 || (P % 16)) return MD_ERR_UNSUPPORTED;
  if ((int64_t)batch * (P / 16) >= ((int64_t)1 << 31) || ksplit > (int64_t)batch * (P / 16)) return MD_ERR_BAD_ARG;
  if (workspace_bytes < md_wgrad_nin_workspace_bytes(co, ci, ksplit)) return MD_ERR_BAD_ARG;
  WnArgsNin g;
  g.DY = (const uint4*)dy_s16; g.X = (const uint4*)x_s16; g.partial = (float*)workspace;
  g.batch = batch; g.co = co; g.ci = ci; g.P = P; g.ksplit = ksplit;
  const int64_t blocks = (int64_t)ksplit * (co / 128) * (ci / 128);
  if (blocks > 0x7fffffff) return MD_ERR_UNSUPPORTED;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_wgrad_nin_kernel, dim3((unsigned)blocks), dim3(WW_THREADS), 0, (hipStream_t)stream, g);
  MD_HIP_CHECK_LAUNCH();
  const int64_t total = (int64_t)co * ci;
  int rb = (int)((total + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(md_wgrad_nin_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw,
                     co, ci, ksplit, s_row, s_k);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
