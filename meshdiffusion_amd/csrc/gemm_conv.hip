// md_gemm_conv: implicit-GEMM 3x3x3 / 1x1x1 convolution and batched GEMM on the
// gfx950 matrix cores with a bf16x3 operand split (hi*hi + hi*lo + lo*hi, fp32
// accumulate in v_mfma_f32_32x32x16_bf16).
//
// Replaces (reference, Python/ATen): nn.Conv3d 3x3x3 lib/diffusion/models/layers.py:118-124,
// Downsample :626-643, Upsample :611-623, NIN :573-582, attention einsums :602,:606.
//
// Mapping   D[i][j] = sum_{tap,k} A[i][tap][k] * B[k][pos_j + tap]
//   i (MFMA rows)  = output channel,  A = packed weight tiles (or an S16B tensor)
//   j (MFMA cols)  = spatial position, B = S16B activations
//   A workgroup owns NT rows x (TZ x TY x TX) positions.  Per K-chunk of KC input
//   channels the input tile WITH HALO is staged once in LDS ([KC/8][hi|lo][slot][8]
//   bf16, 16 B per slot => x-runs are bank-conflict free for ds_read_b128) and is
//   re-used by all 27 taps; per tap a [KC/8][hi|lo][NT][8] weight tile (a linear
//   copy of one WPK tile) is double-buffered through LDS with the global loads of
//   tap t+1 in flight behind the MFMAs of tap t.
//   Accumulator layout (32x32): lane holds col = lane&31 (position) and rows
//   8q + 4*(lane>>5) + {0..3}: four consecutive output channels of one position,
//   i.e. one 16-byte store into the 8-channel blocked F32B layout.
#include "md_common.h"
#include "md_pack.h"

template <int NT_, int KC_, int TZ_, int TY_, int TX_, int TAPS_, int STRIDE_, int WR_, int WC_,
          int SW_ = 0, int PIPE_ = 0, int ABL_ = 0, int BF_ = 0>
struct GCfg {
  // BF = 1: a conv configuration (TAPS > 1) that also takes its B operand as fp32 F32B parts (MD_B_F32B_GN) and applies the
  // folded GroupNorm affine + SiLU + zero padding + the bf16 split while staging the halo tile -- what md_conv3_main_kernel's
  // BF mode does for the 4x8x8 tile, here for the 4^3 level (MD_CFG_C3_LOW): no md_gn_apply pass.
  // One thread = one 8-channel group (kg = tid / (NTHREADS / KG)) x BF_IT halo positions.
  static constexpr int BF = BF_;
  static constexpr int ABL = ABL_;  // timing-only ablations (results invalid): 1 no LDS frag reads, 2 no MFMA,
                                    // 3 no barriers/weight commits, 4 no weight global loads
  static constexpr int NT = NT_, KC = KC_, TZ = TZ_, TY = TY_, TX = TX_, TAPS = TAPS_,
                       STRIDE = STRIDE_, WR = WR_, WC = WC_, SW = SW_, PIPE = PIPE_;
  static constexpr int MT = TZ * TY * TX;
  static constexpr int NW = WR * WC;
  static constexpr int NTHREADS = NW * 64;
  static constexpr int RM = NT / (32 * WR);
  static constexpr int CM = MT / (32 * WC);
  // kernel extent: cubic k^3 (27, 125), or k x k x 1 (9, 25) = one dx column of a k^3 kernel -- the dx-folded head,
  // whose output rows are (co, dx) pairs summed with an x shift by md_fold_dx
  static constexpr int KSZ = (TAPS == 125 || TAPS == 25) ? 5 : (TAPS == 27 || TAPS == 9) ? 3 : 1;
  static constexpr int KSX = (TAPS == 25 || TAPS == 9) ? 1 : KSZ;
  static constexpr int ZH = (TAPS > 1) ? (TZ - 1) * STRIDE + KSZ : 1;
  static constexpr int YH = (TAPS > 1) ? (TY - 1) * STRIDE + KSZ : 1;
  static constexpr int XH = (TAPS > 1) ? (TX - 1) * STRIDE + KSX : MT;
  static constexpr int XHP = XH;
  // SW=1 (tile x-extent 8, stride 1): y-rows are 24 slots apart with odd z-planes interleaved at +12,
  // so a 32-position fragment (8 x by 4 y) touches every 16-byte LDS slot class exactly twice, once
  // per ds_read_b128 lane group => conflict free for every tap (24 = 8 mod 16).
  static constexpr int HS = (SW == 1) ? ((ZH + 1) / 2) * YH * 24 : ZH * YH * XHP;  // halo slots
  static constexpr int HPOS = ZH * YH * XH;                                          // valid halo positions
  static __device__ __forceinline__ int slot_of(int hz, int hy, int hx) {
    if constexpr (SW == 1) return (hz >> 1) * (YH * 24) + hy * 24 + (hz & 1) * 12 + hx;
    else return (hz * YH + hy) * XHP + hx;
  }
  static constexpr int KG = KC / 8;         // 8-channel groups per K chunk
  static constexpr int W_ITEMS = KG * 2 * NT;  // uint4 items of one weight tile
  static constexpr int A_ITEMS = KG * 2 * HPOS;  // uint4 items loaded per activation halo tile
  static constexpr int W_PER_THREAD = (W_ITEMS + NTHREADS - 1) / NTHREADS;
  static constexpr int BF_TPK = NTHREADS / KG;                          // threads per channel group in BF mode
  static constexpr int BF_IT = (HPOS + BF_TPK - 1) / BF_TPK;            // halo positions per thread in BF mode
  static constexpr int A_PER_THREAD_S16 = (A_ITEMS + NTHREADS - 1) / NTHREADS;
  static constexpr int A_PER_THREAD = (BF && 2 * BF_IT > A_PER_THREAD_S16) ? 2 * BF_IT : A_PER_THREAD_S16;
  static constexpr int W_LDS_ITEMS = 2 * W_ITEMS;
  static constexpr int LDS_ITEMS = W_LDS_ITEMS + KG * 2 * HS;
  static constexpr int LDS_BYTES = LDS_ITEMS * 16;
  static constexpr int PADLO = (STRIDE == 2) ? 0 : (KSZ - 1) / 2;  // 'same' padding; Downsample pads (0,1)
  static constexpr int PADLO_X = (STRIDE == 2) ? 0 : (KSX - 1) / 2;
  static_assert(TAPS == 1 || TAPS == 27 || TAPS == 125 || TAPS == 9 || TAPS == 25, "1x1x1, 3x3x3, 5x5x5, 3x3x1 or 5x5x1");
  static_assert(RM >= 1 && CM >= 1, "tile too small for the wave grid");
  static_assert(NT % (32 * WR) == 0 && MT % (32 * WC) == 0, "tile / wave grid mismatch");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
  static_assert(SW == 0 || (TAPS == 27 && STRIDE == 1 && TX == 8), "SW=1 layout is for 3x3x3, x-extent 8, stride 1");
  static_assert(BF == 0 || (TAPS > 1 && STRIDE == 1 && NTHREADS % KG == 0), "BF: stride-1 conv configurations, whole thread groups per channel group");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void md_gemm_conv_kernel(const MdGemmConvArgs A) {
  __shared__ __attribute__((aligned(16))) uint4 smem[C::LDS_ITEMS];
  uint4* wl = smem;                   // [2][KG*2][NT]
  uint4* al = smem + C::W_LDS_ITEMS;  // [KG*2][HS]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = tid >> 6;
  const int wr = wid / C::WC, wc = wid % C::WC;
  const int j = lane & 31, h = lane >> 5;

  // ---- tile coordinates -------------------------------------------------
  const int D = A.D, H = A.H, W = A.W;
  const int64_t P = (int64_t)D * H * W;
  int tiles, z0 = 0, y0 = 0, x0 = 0;
  int Di = D, Hi = H, Wi = W;  // input extents
  if constexpr (C::TAPS > 1) {
    const int ntx = W / C::TX, nty = H / C::TY, ntz = D / C::TZ;
    tiles = ntx * nty * ntz;
    if constexpr (C::STRIDE == 2) { Di = 2 * D; Hi = 2 * H; Wi = 2 * W; }
    if (A.ups) { Di = D >> 1; Hi = H >> 1; Wi = W >> 1; }
  } else {
    tiles = W / C::MT;
  }
  // XCD-aware block order: workgroup b runs on XCD b % 8, so give each XCD one contiguous run of
  // tiles (neighbouring tiles share halos, every tile shares the weights) to keep re-reads in its L2.
  int bid = blockIdx.x;
  if ((gridDim.x & 7) == 0) bid = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int b = bid / tiles;
  const int t = bid % tiles;
  if constexpr (C::TAPS > 1) {
    const int ntx = W / C::TX, nty = H / C::TY;
    x0 = (t % ntx) * C::TX;
    y0 = ((t / ntx) % nty) * C::TY;
    z0 = (t / (ntx * nty)) * C::TZ;
  }
  const int64_t Pin = (int64_t)Di * Hi * Wi;
  const int rt = blockIdx.y;
  const int ncc_total = A.kdim / C::KC;
  // split-K: this workgroup owns K chunks [cc_lo, cc_lo + ncc)
  const int ksplit = A.ksplit > 1 ? A.ksplit : 1;
  const int cc_lo = (int)(((int64_t)blockIdx.z * ncc_total) / ksplit);
  const int ncc = (int)(((int64_t)(blockIdx.z + 1) * ncc_total) / ksplit) - cc_lo;
  const int nsteps = ncc * C::TAPS;

  const uint4* bptr = (const uint4*)A.b + (int64_t)b * (A.b_bstride / 8);
  const uint4* aptr = (const uint4*)A.a;
  if (A.a_src == MD_A_S16B) aptr += (int64_t)b * (A.a_bstride / 8);

  // ---- per-lane fragment addresses ---------------------------------------
  int a_row[C::RM];
#pragma unroll
  for (int rm = 0; rm < C::RM; ++rm) a_row[rm] = (wr * C::RM + rm) * 32 + j;
  int bz[C::CM], by[C::CM], bx[C::CM];  // halo coordinates (tap 0) of this lane's column in each col tile
#pragma unroll
  for (int cm = 0; cm < C::CM; ++cm) {
    const int p = (wc * C::CM + cm) * 32 + j;
    if constexpr (C::TAPS > 1) {
      bx[cm] = (p % C::TX) * C::STRIDE; by[cm] = ((p / C::TX) % C::TY) * C::STRIDE; bz[cm] = (p / (C::TX * C::TY)) * C::STRIDE;
    } else {
      bx[cm] = p; by[cm] = 0; bz[cm] = 0;
    }
  }

  f32x16 acc[C::RM][C::CM];
#pragma unroll
  for (int rm = 0; rm < C::RM; ++rm)
#pragma unroll
    for (int cm = 0; cm < C::CM; ++cm)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rm][cm][r] = 0.f;

  // ---- weight-tile prefetch (global -> registers) ------------------------
  uint4 wreg[C::W_PER_THREAD];
  auto w_issue = [&](int cc, int tap) {
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      const int item = tid + i * C::NTHREADS;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (item < C::W_ITEMS) {
        if (A.a_src == MD_A_PACKED) {
          v = aptr[((int64_t)(rt * ncc_total + cc_lo + cc) * C::TAPS + tap) * C::W_ITEMS + item];
        } else {
          const int gp = item / C::NT, r = item % C::NT;
          const int row = rt * C::NT + r;
          if (row < A.a_rows)
            v = aptr[((int64_t)((cc_lo + cc) * C::KG + (gp >> 1)) * 2 + (gp & 1)) * A.a_rows + row];
        }
      }
      wreg[i] = v;
    }
  };
  auto w_commit = [&](int buf) {
#pragma unroll
    for (int i = 0; i < C::W_PER_THREAD; ++i) {
      const int item = tid + i * C::NTHREADS;
      if (item < C::W_ITEMS) wl[buf * C::W_ITEMS + item] = wreg[i];
    }
  };

  // ---- activation halo tile: global -> registers -> LDS (zero fill outside the grid) ----
  uint4 hreg[C::A_PER_THREAD];
  // MD_B_F32B_GN on a 1x1x1 configuration (the ResnetBlock shortcut NIN_0 reading the raw block input, layers.py:688):
  // one thread = (8-channel group, position) pairs; two fp32 uint4 in, the hi and the lo plane item out.  No affine /
  // SiLU here (b_ac must be NULL): the shortcut takes the un-normalised input.
  const bool bf32 = (C::TAPS == 1) && A.b_mode == MD_B_F32B_GN;
  const uint4* bf_p1 = (const uint4*)A.b + (int64_t)b * (A.b_bstride / 4);
  const uint4* bf_p2 = (const uint4*)A.b2 + (int64_t)b * (A.b2_bstride / 4);
  auto act_issue_f32 = [&](auto& hr, int cc) {
    if constexpr (C::TAPS == 1 && (C::A_PER_THREAD % 2) == 0) {
#pragma unroll
      for (int i = 0; i < C::A_PER_THREAD / 2; ++i) {
        const int q = tid + i * C::NTHREADS;                 // pair index: (kg, position)
        const int kg = q / C::HPOS, r = q % C::HPOS;
        const int g8 = (cc_lo + cc) * C::KG + kg;
        const uint4* cb = (g8 < (A.b_split >> 3)) ? bf_p1 + (int64_t)g8 * Pin * 2 : bf_p2 + (int64_t)(g8 - (A.b_split >> 3)) * Pin * 2;
        const int64_t src = ((int64_t)t * C::MT + r) * 2;
        const uint4 r0 = cb[src], r1 = cb[src + 1];
        const float v[8] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w),
                            __uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w)};
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) md_split2(v[2 * e], v[2 * e + 1], hi[e], lo[e]);
        hr[2 * i] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        hr[2 * i + 1] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  };
  auto act_commit_f32 = [&](auto& hr) {
    if constexpr (C::TAPS == 1 && (C::A_PER_THREAD % 2) == 0) {
#pragma unroll
      for (int i = 0; i < C::A_PER_THREAD / 2; ++i) {
        const int q = tid + i * C::NTHREADS;
        const int kg = q / C::HPOS, r = q % C::HPOS;
        al[(kg * 2) * C::HS + r] = hr[2 * i];
        al[(kg * 2 + 1) * C::HS + r] = hr[2 * i + 1];
      }
    }
  };
  // BF (conv configurations): kg and the halo positions of this thread are fixed; per chunk it loads its 8 channels' folded
  // affine (a, c) and BF_IT x 8 fp32 values; the transform runs right before the LDS store (act_commit)
  const bool bf3 = C::BF && (C::TAPS > 1) && A.b_mode == MD_B_F32B_GN;
  const int bf_kg = C::BF ? tid / C::BF_TPK : 0;
  f32x4 bf_ac[C::BF ? 4 : 1];
  unsigned bf_live = 0;              // bit i: halo position i of the chunk in flight lies inside the grid
  auto act_issue_bf = [&](auto& hr, int cc) {
    if constexpr (C::BF) {
      const int g8 = (cc_lo + cc) * C::KG + bf_kg;
      const uint4* cb = (g8 < (A.b_split >> 3)) ? bf_p1 + (int64_t)g8 * Pin * 2 : bf_p2 + (int64_t)(g8 - (A.b_split >> 3)) * Pin * 2;
      bf_live = 0;
#pragma unroll
      for (int i = 0; i < C::BF_IT; ++i) {
        const int r = (tid % C::BF_TPK) + i * C::BF_TPK;
        const int hx = r % C::XH, hy = (r / C::XH) % C::YH, hz = r / (C::XH * C::YH);
        const int uz = z0 + hz - C::PADLO, uy = y0 + hy - C::PADLO, ux = x0 + hx - C::PADLO_X;
        const bool inb = (r < C::HPOS) & (uz >= 0) & (uz < Di) & (uy >= 0) & (uy < Hi) & (ux >= 0) & (ux < Wi);
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = v0;
        if (inb) {
          const int64_t src = (((int64_t)uz * Hi + uy) * Wi + ux) * 2;
          v0 = cb[src]; v1 = cb[src + 1];
          bf_live |= 1u << i;
        }
        hr[2 * i] = v0; hr[2 * i + 1] = v1;
      }
      if (A.b_ac != nullptr) {
        const f32x4* ap = (const f32x4*)(A.b_ac + ((int64_t)b * A.kdim + (int64_t)g8 * 8) * 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) bf_ac[q] = ap[q];
      }
    }
  };
  auto act_commit_bf = [&](auto& hr) {
    if constexpr (C::BF) {
#pragma unroll
      for (int i = 0; i < C::BF_IT; ++i) {
        const int r = (tid % C::BF_TPK) + i * C::BF_TPK;
        if (r >= C::HPOS) continue;
        const uint4 r0 = hr[2 * i], r1 = hr[2 * i + 1];
        const float v[8] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w),
                            __uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z), __uint_as_float(r1.w)};
        const bool live = (bf_live >> i) & 1u;      // outside the grid the ACTIVATED tensor is zero padded
        float yv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float y = v[e];
          if (A.b_ac != nullptr) {
            y = y * bf_ac[e >> 1][(e & 1) * 2] + bf_ac[e >> 1][(e & 1) * 2 + 1];
            if (A.b_silu) y = y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(y * -1.4426950408889634f));
          }
          yv[e] = live ? y : 0.f;
        }
        uint32_t hi[4], lo[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) md_split2(yv[2 * e], yv[2 * e + 1], hi[e], lo[e]);
        const int slot = C::slot_of(r / (C::XH * C::YH), (r / C::XH) % C::YH, r % C::XH);
        al[(bf_kg * 2) * C::HS + slot] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        al[(bf_kg * 2 + 1) * C::HS + slot] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
  };
  auto act_issue = [&](auto& hr, int cc) {
    if (bf32) { act_issue_f32(hr, cc); return; }
    if (bf3) { act_issue_bf(hr, cc); return; }
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD_S16; ++i) {
      const int item = tid + i * C::NTHREADS;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (item < C::A_ITEMS) {
        const int gp = item / C::HPOS, r = item % C::HPOS;
        int64_t src;
        bool inb = true;
        if constexpr (C::TAPS > 1) {
          const int hx = r % C::XH, hy = (r / C::XH) % C::YH, hz = r / (C::XH * C::YH);
          int uz = z0 * C::STRIDE + hz - C::PADLO;
          int uy = y0 * C::STRIDE + hy - C::PADLO;
          int ux = x0 * C::STRIDE + hx - C::PADLO_X;
          if (A.ups) {
            inb = (uz >= 0) & (uz < D) & (uy >= 0) & (uy < H) & (ux >= 0) & (ux < W);
            uz >>= 1; uy >>= 1; ux >>= 1;
          } else {
            inb = (uz >= 0) & (uz < Di) & (uy >= 0) & (uy < Hi) & (ux >= 0) & (ux < Wi);
          }
          src = ((int64_t)uz * Hi + uy) * Wi + ux;
        } else {
          src = (int64_t)t * C::MT + r;
        }
        if (inb) v = bptr[((int64_t)((cc_lo + cc) * C::KG + (gp >> 1)) * 2 + (gp & 1)) * Pin + src];
      }
      hr[i] = v;
    }
  };
  auto act_commit = [&](auto& hr) {
    if (bf32) { act_commit_f32(hr); return; }
    if (bf3) { act_commit_bf(hr); return; }
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD_S16; ++i) {
      const int item = tid + i * C::NTHREADS;
      if (item < C::A_ITEMS) {
        const int gp = item / C::HPOS, r = item % C::HPOS;
        int slot = r;
        if constexpr (C::TAPS > 1) slot = C::slot_of(r / (C::XH * C::YH), (r / C::XH) % C::YH, r % C::XH);
        al[gp * C::HS + slot] = hr[i];
      }
    }
  };

  const bf16x8* wlf = (const bf16x8*)wl;
  const bf16x8* alf = (const bf16x8*)al;

  // ---- one (K chunk, tap) step of MFMAs ----------------------------------------
  bf16x8 abl_a[C::RM], abl_b[C::CM];  // ABL==1 only
  if constexpr (C::ABL == 1 || C::ABL == 5) {
#pragma unroll
    for (int rm = 0; rm < C::RM; ++rm)
#pragma unroll
      for (int e = 0; e < 8; ++e) abl_a[rm][e] = (short)(0x3f80 + lane + e);
#pragma unroll
    for (int cm = 0; cm < C::CM; ++cm)
#pragma unroll
      for (int e = 0; e < 8; ++e) abl_b[cm][e] = (short)(0x3c00 + lane * 3 + e);
  }
  auto compute = [&](int buf, int dz, int dy, int dx) {
    const bf16x8* wb = wlf + buf * C::W_ITEMS;
    int bs[C::CM];
#pragma unroll
    for (int cm = 0; cm < C::CM; ++cm) bs[cm] = C::slot_of(bz[cm] + dz, by[cm] + dy, bx[cm] + dx);
#pragma unroll
    for (int ks = 0; ks < C::KC / 16; ++ks) {
      const int g = ks * 2 + h;
      bf16x8 ahi[C::RM], alo[C::RM], bhi[C::CM], blo[C::CM];
#pragma unroll
      for (int rm = 0; rm < C::RM; ++rm) {
        if constexpr (C::ABL == 1 || C::ABL == 5) { ahi[rm] = abl_a[rm]; alo[rm] = abl_a[rm]; }
        else {
          ahi[rm] = wb[(g * 2 + 0) * C::NT + a_row[rm]];
          alo[rm] = wb[(g * 2 + 1) * C::NT + a_row[rm]];
        }
      }
#pragma unroll
      for (int cm = 0; cm < C::CM; ++cm) {
        if constexpr (C::ABL == 1 || C::ABL == 5) { bhi[cm] = abl_b[cm]; blo[cm] = abl_b[cm]; }
        else {
          bhi[cm] = alf[(g * 2 + 0) * C::HS + bs[cm]];
          blo[cm] = alf[(g * 2 + 1) * C::HS + bs[cm]];
        }
      }
      if constexpr (C::ABL == 2) {
#pragma unroll
        for (int rm = 0; rm < C::RM; ++rm) { asm volatile("" ::"v"(ahi[rm]), "v"(alo[rm])); }
#pragma unroll
        for (int cm = 0; cm < C::CM; ++cm) { asm volatile("" ::"v"(bhi[cm]), "v"(blo[cm])); }
      } else {
#pragma unroll
        for (int rm = 0; rm < C::RM; ++rm)
#pragma unroll
          for (int cm = 0; cm < C::CM; ++cm) {
            acc[rm][cm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo[rm], bhi[cm], acc[rm][cm], 0, 0, 0);
            acc[rm][cm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[rm], blo[cm], acc[rm][cm], 0, 0, 0);
            acc[rm][cm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi[rm], bhi[cm], acc[rm][cm], 0, 0, 0);
          }
      }
    }
  };
  auto advance = [&](int& cc, int& tap, int& dz, int& dy, int& dx) {
    ++tap; ++dx;
    if constexpr (C::TAPS > 1) {
      if (dx == C::KSX) { dx = 0; ++dy; }
      if (dy == C::KSZ) { dy = 0; ++dz; }
    }
    if (tap == C::TAPS) { tap = 0; dz = dy = dx = 0; ++cc; }
  };

  // ---- main loop over (K chunk, tap) ---------------------------------------
  int cc = 0, tap = 0, dz = 0, dy = 0, dx = 0;
  if constexpr (C::PIPE == 0) {
    w_issue(0, 0);
    for (int s = 0; s < nsteps; ++s) {
      if (tap == 0) {
        __syncthreads();  // everyone finished reading the previous halo tile
        act_issue(hreg, cc);
        act_commit(hreg);
      }
      const int buf = s & 1;
      w_commit(buf);
      __syncthreads();
      int ncc_ = cc, ntap = tap, ndz = dz, ndy = dy, ndx = dx;
      advance(ncc_, ntap, ndz, ndy, ndx);
      if (s + 1 < nsteps) w_issue(ncc_, ntap);  // next step's weights fly behind the MFMAs
      compute(buf, dz, dy, dx);
      cc = ncc_; tap = ntap; dz = ndz; dy = ndy; dx = ndx;
    }
  } else {
    // Software pipeline: W(s+1) is written to LDS and W(s+2) is requested from L2 *inside* step s,
    // the next K-chunk's halo tile is requested 3 taps early into registers, and there is exactly
    // one barrier per step (two at a chunk switch).
    constexpr int PF = (C::TAPS >= 3) ? C::TAPS - 3 : 0;
    int c1 = 0, t1 = 0, z1 = 0, y1 = 0, x1 = 0;  // coordinates of step s+2 (for w_issue)
    w_issue(0, 0);
    act_issue(hreg, 0);
    act_commit(hreg);
    w_commit(0);
    advance(c1, t1, z1, y1, x1);
    if (nsteps > 1) w_issue(c1, t1);
    advance(c1, t1, z1, y1, x1);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (C::ABL != 3 && C::ABL != 5 && s + 1 < nsteps) w_commit((s + 1) & 1);
      if (C::ABL != 4 && C::ABL != 5 && s + 2 < nsteps) w_issue(c1, t1);
      advance(c1, t1, z1, y1, x1);
      const bool more = cc + 1 < ncc;
      if (C::ABL != 5 && tap == PF && more) act_issue(hreg, cc + 1);
      compute(s & 1, dz, dy, dx);
      if (C::ABL != 5 && tap == C::TAPS - 1 && more) {
        __syncthreads();  // everyone finished reading this chunk's halo tile
        act_commit(hreg);
      }
      if constexpr (C::ABL != 3 && C::ABL != 5) __syncthreads();
      advance(cc, tap, dz, dy, dx);
    }
  }

  // ---- epilogue --------------------------------------------------------------
  if (ksplit > 1) {  // raw partial sums -> workspace slice (F32B layout); md_splitk_reduce finishes the job
    const int rga = A.rows_alloc / 8;
    float* part = A.partial + ((int64_t)blockIdx.z * A.batch + b) * rga * P * 8;
#pragma unroll
    for (int cm = 0; cm < C::CM; ++cm) {
      const int p = (wc * C::CM + cm) * 32 + j;
      int64_t gp;
      if constexpr (C::TAPS > 1) {
        const int x = p % C::TX, y = (p / C::TX) % C::TY, z = p / (C::TX * C::TY);
        gp = ((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
      } else {
        gp = (int64_t)t * C::MT + p;
      }
#pragma unroll
      for (int rm = 0; rm < C::RM; ++rm)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = rt * C::NT + (wr * C::RM + rm) * 32 + 8 * q + 4 * h;
          if (row >= A.rows_alloc) continue;
          f32x4 o4 = {acc[rm][cm][q * 4 + 0], acc[rm][cm][q * 4 + 1], acc[rm][cm][q * 4 + 2], acc[rm][cm][q * 4 + 3]};
          *(f32x4*)(part + ((int64_t)(row >> 3) * P + gp) * 8 + (row & 7)) = o4;
        }
    }
    return;
  }
  const float alpha = A.alpha;
  const int rows = A.rows, rows_alloc = A.rows_alloc;
  const int rg_alloc = rows_alloc / 8;
#pragma unroll
  for (int cm = 0; cm < C::CM; ++cm) {
    const int p = (wc * C::CM + cm) * 32 + j;
    int64_t gp;
    if constexpr (C::TAPS > 1) {
      const int x = p % C::TX, y = (p / C::TX) % C::TY, z = p / (C::TX * C::TY);
      gp = ((int64_t)(z0 + z) * H + (y0 + y)) * W + (x0 + x);
    } else {
      gp = (int64_t)t * C::MT + p;
    }
#pragma unroll
    for (int rm = 0; rm < C::RM; ++rm) {
      const int rowbase = rt * C::NT + (wr * C::RM + rm) * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = rowbase + 8 * q + 4 * h;  // first of 4 consecutive rows
        if (row >= rows_alloc && A.out_mode != MD_OUT_NCDHW) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float x = alpha * acc[rm][cm][q * 4 + e];
          if (A.bias != nullptr && row + e < rows) x += A.bias[(int64_t)b * A.bias_bstride + row + e];
          v[e] = x;
        }
        if (A.out_mode == MD_OUT_F32B) {
          const int64_t o = (((int64_t)b * rg_alloc + (row >> 3)) * P + gp) * 8 + (row & 7);
          if (A.residual != nullptr) {
            const int64_t ro = (int64_t)b * A.res_bstride + (((int64_t)(row >> 3)) * P + gp) * 8 + (row & 7);
            const f32x4 r4 = *(const f32x4*)(A.residual + ro);
            v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3];
          }
          f32x4 o4 = {v[0], v[1], v[2], v[3]};
          *(f32x4*)((float*)A.out + o) = o4;
        } else if (A.out_mode == MD_OUT_S16B) {
          uint32_t hi[4], lo[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) md_split(v[e], hi[e], lo[e]);
          const int64_t o = ((((int64_t)b * rg_alloc + (row >> 3)) * 2) * P + gp) * 8 + (row & 7);
          uint2 h2 = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
          uint2 l2 = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
          *(uint2*)((uint16_t*)A.out + o) = h2;
          *(uint2*)((uint16_t*)A.out + o + P * 8) = l2;
        } else {  // NCDHW fp32 [B][rows][P]
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (row + e < rows) ((float*)A.out)[((int64_t)b * rows + row + e) * P + gp] = v[e];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------
using Cfg_C3_128 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4>;
using Cfg_C3_128_V2 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1>;
using Cfg_C3_128_SW = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 0>;
using Cfg_C3_128_PIPE = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 0, 1>;
using Cfg_C5_128_K16 = GCfg<128, 16, 4, 8, 8, 125, 1, 2, 4>;  // 5x5x5 stem of ddpm_res128 (Cin<=16)
using Cfg_C5_32_K16 = GCfg<32, 16, 4, 8, 8, 125, 1, 1, 8>;    // 5x5x5 head of ddpm_res128 (Cout<=32)
using Cfg_ABL1 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1, 1>;
using Cfg_ABL2 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1, 2>;
using Cfg_ABL3 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1, 3>;
using Cfg_ABL4 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1, 4>;
using Cfg_ABL5 = GCfg<128, 32, 4, 8, 8, 27, 1, 2, 4, 1, 1, 5>;
using Cfg_C3_128_K16 = GCfg<128, 16, 4, 8, 8, 27, 1, 2, 4>;
using Cfg_C3_32 = GCfg<32, 32, 4, 8, 8, 27, 1, 1, 8>;
using Cfg_C3X_32 = GCfg<32, 32, 4, 8, 8, 9, 1, 1, 8>;        // (BF = 1 measured on the res64 head: 0.67 -> 1.39 ms, more than the md_gn_apply pass it saves: the unpipelined loader exposes the transform)        // 3x3x1 taps: dx-folded 3x3x3 head (rows = (co, dx))
using Cfg_C5X_32_K16 = GCfg<32, 16, 4, 8, 8, 25, 1, 1, 8>;   // 5x5x1 taps: dx-folded 5x5x5 head of ddpm_res128
using Cfg_C3X_128_K16 = GCfg<128, 16, 4, 8, 8, 9, 1, 2, 4>;   // (PIPE=1 measured slower here: 1.02 vs 0.67 ms)  // dx-folded 3x3x3 stem: K = 4 ch x 3 dx (12 -> 16)
using Cfg_C5X_128 = GCfg<128, 32, 4, 8, 8, 25, 1, 2, 4>;     // dx-folded 5x5x5 stem: K = 4 ch x 5 dx (20 -> 32)
using Cfg_C3_LOW = GCfg<128, 32, 4, 4, 4, 27, 1, 4, 2, 0, 1, 0, 1>;   // PIPE=1: weight / halo loads requested ahead of the MFMAs
// experiment: 4-wave workgroups on a 4x4x8 tile (78.8 KB of LDS => two independent workgroups per CU instead of one
// 8-wave workgroup): +4.5 % on 128->128 @64^3, -3 % on 256->128 against Cfg_C3_128 -- decoupling the barriers does not pay
// for the larger halo re-read factor (2.8 vs 2.3)
using Cfg_C3_128_W4 = GCfg<128, 32, 4, 4, 8, 27, 1, 2, 2>;
using Cfg_C3_S2 = GCfg<128, 32, 4, 4, 4, 27, 2, 4, 2, 0, 1>;
// PIPE=1: next chunk's weight and activation tiles are requested before the MFMAs of the current one (the 1x1x1 / GEMM
// launches are HBM-bound: the unpipelined loop left the memory system idle during every compute phase).  Two activation
// tiles in flight instead of one measured the same (3.72 ms per step for the four 256->128 @64^3 shortcuts either way)
using Cfg_G1_128 = GCfg<128, 32, 1, 1, 256, 1, 1, 2, 4, 0, 1>;
using Cfg_G1_128_LOW = GCfg<128, 32, 1, 1, 64, 1, 1, 4, 2, 0, 1>;
// 128 columns per workgroup (48 KB of LDS: three workgroups per CU): the HBM-bound ResnetBlock shortcut at 64^3 / 32^3 --
// one workgroup's output burst (64 KB) overlaps the other workgroups' loads
using Cfg_G1_128_N128 = GCfg<128, 32, 1, 1, 128, 1, 1, 2, 4, 0, 1>;
using Cfg_G1_64_LOW = GCfg<64, 32, 1, 1, 64, 1, 1, 2, 2, 0, 1>;

// ---- split-K finish: out = alpha * sum_z partial[z] + bias + residual (slices added in order) ----
__global__ void md_splitk_reduce_kernel(const MdGemmConvArgs A, int64_t P) {
  const int rga = A.rows_alloc / 8;
  const int64_t n4 = (int64_t)A.batch * rga * P * 2;  // float4 items
  const int64_t slice = (int64_t)A.batch * rga * P * 8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e0 = i * 4;
    f32x4 s = *(const f32x4*)(A.partial + e0);
    for (int z = 1; z < A.ksplit; ++z) {
      const f32x4 v = *(const f32x4*)(A.partial + (int64_t)z * slice + e0);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int64_t per_b = (int64_t)rga * P * 8;
    const int b = (int)(e0 / per_b);
    const int64_t r = e0 % per_b;
    const int row = (int)(r / (P * 8)) * 8 + (int)(r & 7);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x = A.alpha * s[e];
      if (A.bias != nullptr && row + e < A.rows) x += A.bias[(int64_t)b * A.bias_bstride + row + e];
      o[e] = x;
    }
    if (A.residual != nullptr) {
      const f32x4 rv = *(const f32x4*)(A.residual + (int64_t)b * A.res_bstride + r);
      o[0] += rv[0]; o[1] += rv[1]; o[2] += rv[2]; o[3] += rv[3];
    }
    *(f32x4*)((float*)A.out + e0) = o;
  }
}

// The same finish with the GroupNorm sums of the result (MdGemmConvArgs.stats) for the small grids that run split-K (8^3 / 4^3
// levels): one block per (sample, 8-channel group) walks that slab's P positions; a thread keeps one channel quad (its float4
// index parity is constant over the block stride), so the sums are per-thread adds, one wave reduction, 16 plain adds to stats
// -- instead of a md_gn_stats launch over the tensor just written.
__global__ __launch_bounds__(256) void md_splitk_reduce_stats_kernel(const MdGemmConvArgs A, int64_t P) {
  __shared__ float red[4][16];
  const int rga = A.rows_alloc / 8;
  const int cg = blockIdx.x % rga, b = blockIdx.x / rga;
  const int64_t slab = ((int64_t)b * rga + cg) * P * 8;                // first float of this (sample, channel group)
  const int64_t slice = (int64_t)A.batch * rga * P * 8;
  const int row = cg * 8 + (threadIdx.x & 1) * 4;
  f32x4 bias = {0.f, 0.f, 0.f, 0.f};
  if (A.bias != nullptr) {
#pragma unroll
    for (int e = 0; e < 4; ++e) if (row + e < A.rows) bias[e] = A.bias[(int64_t)b * A.bias_bstride + row + e];
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  for (int64_t i = threadIdx.x; i < P * 2; i += 256) {                 // float4 items of the slab: (position, channel half)
    const int64_t e0 = slab + i * 4;
    f32x4 s = *(const f32x4*)(A.partial + e0);
    for (int z = 1; z < A.ksplit; ++z) {
      const f32x4 v = *(const f32x4*)(A.partial + (int64_t)z * slice + e0);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = A.alpha * s[e] + bias[e];       // same association as md_splitk_reduce_kernel
    if (A.residual != nullptr) {
      const f32x4 rv = *(const f32x4*)(A.residual + (int64_t)b * A.res_bstride + (int64_t)cg * P * 8 + i * 4);
      o[0] += rv[0]; o[1] += rv[1]; o[2] += rv[2]; o[3] += rv[3];
    }
    *(f32x4*)((float*)A.out + e0) = o;
    s1 += o; s2 += o * o;
  }
  // lanes of equal parity hold the same channel quad: xor-reduce over lane bits 1..5, then over the 4 waves through LDS
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int o = 2; o < 64; o <<= 1) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane < 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[wv][(lane * 4 + e) * 2] = s1[e]; red[wv][(lane * 4 + e) * 2 + 1] = s2[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 16) {
    const float t = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    A.stats[((int64_t)b * A.rows_alloc + cg * 8) * 2 + threadIdx.x] += (double)t;     // this block owns the 8 channels
  }
}

int md_launch_splitk_reduce(const MdGemmConvArgs& a, hipStream_t stream) {
  const int64_t P = (int64_t)a.D * a.H * a.W;
  if (a.stats != nullptr) {
    MD_HIP_CLEAR_ERROR();
    hipLaunchKernelGGL(md_splitk_reduce_stats_kernel, dim3((unsigned)(a.batch * (a.rows_alloc / 8))), dim3(256), 0, stream, a, P);
    MD_HIP_CHECK_LAUNCH();
    return MD_OK;
  }
  const int64_t n4 = (int64_t)a.batch * (a.rows_alloc / 8) * P * 2;
  int blocks = (int)((n4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a, P);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}

extern "C" int64_t md_gemm_conv_partial_bytes(const MdGemmConvArgs* a) {
  if (a == nullptr) return MD_ERR_BAD_ARG;
  if (a->ksplit <= 1) return 0;
  return (int64_t)a->ksplit * a->batch * (a->rows_alloc / 8) * ((int64_t)a->D * a->H * a->W) * 8 * 4;
}

template <class C>
static int launch_cfg(const MdGemmConvArgs& a, hipStream_t stream) {
  if (a.kdim % C::KC != 0 || a.kdim <= 0) return MD_ERR_BAD_ARG;
  if (a.rows <= 0 || a.rows_alloc % 8 != 0 || a.batch <= 0) return MD_ERR_BAD_ARG;
  int tiles;
  if (C::TAPS > 1) {
    if (a.D % C::TZ || a.H % C::TY || a.W % C::TX) return MD_ERR_BAD_ARG;
    if (a.ups && ((a.D | a.H | a.W) & 1)) return MD_ERR_BAD_ARG;
    if (a.ups && C::STRIDE != 1) return MD_ERR_BAD_ARG;
    tiles = (a.D / C::TZ) * (a.H / C::TY) * (a.W / C::TX);
  } else {
    if (a.D != 1 || a.H != 1 || a.W % C::MT || a.ups) return MD_ERR_BAD_ARG;
    tiles = a.W / C::MT;
  }
  if (a.a_src == MD_A_S16B && a.a_rows <= 0) return MD_ERR_BAD_ARG;
  if (a.prec != MD_PREC_BF16X3) return MD_ERR_UNSUPPORTED;  // fp16x2 lives in the dedicated conv kernel only
  if (a.b_mode != MD_B_S16B) {
    if (a.b_mode != MD_B_F32B_GN || (a.b_split & 7) || a.b_split <= 0 || (a.b_split < a.kdim && a.b2 == nullptr)) return MD_ERR_UNSUPPORTED;
    if (C::TAPS == 1) {          // fp32 operand: split-only, 1x1x1 configurations whose tile is a whole number of pairs per thread
      if ((C::A_PER_THREAD % 2) || (C::A_ITEMS % (2 * C::NTHREADS)) || a.b_ac != nullptr) return MD_ERR_UNSUPPORTED;
    } else {                     // conv configurations built with BF: GroupNorm affine (+ SiLU) + split in the halo loader
      if (!C::BF || a.ups) return MD_ERR_UNSUPPORTED;
      if (a.b_silu && a.b_ac == nullptr) return MD_ERR_BAD_ARG;
    }
  }
  const int row_tiles = (a.rows + C::NT - 1) / C::NT;
  const int ks = a.ksplit > 1 ? a.ksplit : 1;
  if (ks > 1 && (a.partial == nullptr || a.out_mode != MD_OUT_F32B || ks > a.kdim / C::KC)) return MD_ERR_BAD_ARG;
  dim3 grid((unsigned)(tiles * a.batch), (unsigned)row_tiles, (unsigned)ks);
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_gemm_conv_kernel<C>, grid, dim3(C::NTHREADS), 0, stream, a);
  MD_HIP_CHECK_LAUNCH();
  if (ks > 1) return md_launch_splitk_reduce(a, stream);
  return MD_OK;
}

template <class C>
static void cfg_info(int32_t* nt, int32_t* kc, int32_t* cols, int32_t* taps, int32_t* lds, int32_t* thr) {
  if (nt) *nt = C::NT;
  if (kc) *kc = C::KC;
  if (cols) *cols = C::MT;
  if (taps) *taps = C::TAPS;
  if (lds) *lds = C::LDS_BYTES;
  if (thr) *thr = C::NTHREADS;
}

// Production configurations; the A/B baselines of tools/bench_conv.py (V2/SW/PIPE/W4 and the timing-only ABL
// variants) are instantiated only with -DMD_BUILD_ABLATIONS (MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build --force).
#define MD_CFG_CASES_PROD(F, ...)                                   \
    case MD_CFG_C3_128: F<Cfg_C3_128>(__VA_ARGS__); break;          \
    case MD_CFG_C3_128_K16: F<Cfg_C3_128_K16>(__VA_ARGS__); break;  \
    case MD_CFG_C3_32: F<Cfg_C3_32>(__VA_ARGS__); break;            \
    case MD_CFG_C3X_32: F<Cfg_C3X_32>(__VA_ARGS__); break;          \
    case MD_CFG_C5X_32_K16: F<Cfg_C5X_32_K16>(__VA_ARGS__); break;  \
    case MD_CFG_C3X_128_K16: F<Cfg_C3X_128_K16>(__VA_ARGS__); break; \
    case MD_CFG_C5X_128: F<Cfg_C5X_128>(__VA_ARGS__); break;        \
    case MD_CFG_C3_LOW: F<Cfg_C3_LOW>(__VA_ARGS__); break;          \
    case MD_CFG_C3_S2: F<Cfg_C3_S2>(__VA_ARGS__); break;            \
    case MD_CFG_G1_128: F<Cfg_G1_128>(__VA_ARGS__); break;          \
    case MD_CFG_G1_128_LOW: F<Cfg_G1_128_LOW>(__VA_ARGS__); break;  \
    case MD_CFG_G1_128_N128: F<Cfg_G1_128_N128>(__VA_ARGS__); break; \
    case MD_CFG_G1_64_LOW: F<Cfg_G1_64_LOW>(__VA_ARGS__); break;    \
    case MD_CFG_C5_128_K16: F<Cfg_C5_128_K16>(__VA_ARGS__); break;  \
    case MD_CFG_C5_32_K16: F<Cfg_C5_32_K16>(__VA_ARGS__); break;
#define MD_CFG_CASES_ABL(F, ...)                                    \
    case MD_CFG_C3_128_W4: F<Cfg_C3_128_W4>(__VA_ARGS__); break;    \
    case MD_CFG_C3_128_V2: F<Cfg_C3_128_V2>(__VA_ARGS__); break;    \
    case MD_CFG_C3_128_SW: F<Cfg_C3_128_SW>(__VA_ARGS__); break;    \
    case MD_CFG_C3_128_PIPE: F<Cfg_C3_128_PIPE>(__VA_ARGS__); break;
#define MD_CFG_CASES_TIMING(F, ...)                                 \
    case 101: F<Cfg_ABL1>(__VA_ARGS__); break;                      \
    case 102: F<Cfg_ABL2>(__VA_ARGS__); break;                      \
    case 103: F<Cfg_ABL3>(__VA_ARGS__); break;                      \
    case 104: F<Cfg_ABL4>(__VA_ARGS__); break;                      \
    case 105: F<Cfg_ABL5>(__VA_ARGS__); break;
#ifdef MD_BUILD_ABLATIONS
#define MD_CFG_SWITCH_LAUNCH(cfg, F, ...) \
  switch (cfg) { MD_CFG_CASES_PROD(F, __VA_ARGS__) MD_CFG_CASES_ABL(F, __VA_ARGS__) MD_CFG_CASES_TIMING(F, __VA_ARGS__) default: return MD_ERR_UNSUPPORTED; }
#else
#define MD_CFG_SWITCH_LAUNCH(cfg, F, ...) \
  switch (cfg) { MD_CFG_CASES_PROD(F, __VA_ARGS__) default: return MD_ERR_UNSUPPORTED; }
#endif
// geometry queries never instantiate a kernel: every configuration of the header answers (the timing-only ids 101..127 in
// ablation builds only)
#ifdef MD_BUILD_ABLATIONS
#define MD_CFG_SWITCH_INFO(cfg, F, ...) \
  switch (cfg) { MD_CFG_CASES_PROD(F, __VA_ARGS__) MD_CFG_CASES_ABL(F, __VA_ARGS__) MD_CFG_CASES_TIMING(F, __VA_ARGS__) default: return MD_ERR_UNSUPPORTED; }
#define MD_CFG_IS_TIMING_FAST(c) ((c) >= 111 && (c) <= 127)
#else
#define MD_CFG_SWITCH_INFO(cfg, F, ...) \
  switch (cfg) { MD_CFG_CASES_PROD(F, __VA_ARGS__) MD_CFG_CASES_ABL(F, __VA_ARGS__) default: return MD_ERR_UNSUPPORTED; }
#define MD_CFG_IS_TIMING_FAST(c) false
#endif

int md_launch_conv3_main(const MdGemmConvArgs& a, hipStream_t stream);  // conv3_main.hip

extern "C" int md_gemm_conv(const MdGemmConvArgs* args, void* stream) {
  if (args == nullptr || args->a == nullptr || args->b == nullptr || args->out == nullptr)
    return MD_ERR_BAD_ARG;
  int rc = MD_OK;
  hipStream_t st = (hipStream_t)stream;
  if (args->cfg == MD_CFG_C3_128_FAST || MD_CFG_IS_TIMING_FAST(args->cfg)) return md_launch_conv3_main(*args, st);
  if (args->stats != nullptr && !(args->ksplit > 1 && args->out_mode == MD_OUT_F32B))
    return MD_ERR_UNSUPPORTED;   // statistics: the dedicated 3x3x3 kernel's epilogue, or the split-K finish (md_splitk_reduce_stats_kernel)
  MD_CFG_SWITCH_LAUNCH(args->cfg, rc = launch_cfg, *args, st);
  return rc;
}

extern "C" int md_gemm_conv_cfg_info(int32_t cfg, int32_t* nt, int32_t* kc, int32_t* cols,
                                     int32_t* taps, int32_t* lds_bytes, int32_t* threads) {
  if (cfg == MD_CFG_C3_128_FAST || MD_CFG_IS_TIMING_FAST(cfg)) cfg = MD_CFG_C3_128_V2;  // same tile geometry and LDS image
  MD_CFG_SWITCH_INFO(cfg, cfg_info, nt, kc, cols, taps, lds_bytes, threads);
  return MD_OK;
}

// ---- weight packing: fp32 -> WPK split-bf16 tiles ------------------------------------
__global__ void md_pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ out,
                                       int rows, int kdim, int taps, int64_t s_row, int64_t s_k,
                                       int64_t s_tap, int nt, int kc, int64_t n_items, int prec) {
  for (int64_t item = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; item < n_items;
       item += (int64_t)gridDim.x * blockDim.x)
    out[item] = md_pack_wpk_item(w, rows, kdim, taps, s_row, s_k, s_tap, nt, kc, prec, item);
}

extern "C" int64_t md_packed_weight_bytes(int32_t rows, int32_t kdim, int32_t taps, int32_t nt,
                                          int32_t kc) {
  if (rows <= 0 || kdim <= 0 || taps <= 0 || nt <= 0 || kc <= 0 || (kc % 8)) return MD_ERR_BAD_ARG;
  const int64_t rt = (rows + nt - 1) / nt, ncc = (kdim + kc - 1) / kc;
  return rt * ncc * taps * (int64_t)(kc / 8) * 2 * nt * 16;
}

extern "C" int md_pack_weights(const float* w, void* wpk, int32_t rows, int32_t kdim, int32_t taps,
                               int64_t s_row, int64_t s_k, int64_t s_tap, int32_t nt, int32_t kc,
                               int32_t prec, void* stream) {
  const int64_t bytes = md_packed_weight_bytes(rows, kdim, taps, nt, kc);
  if (bytes < 0 || w == nullptr || wpk == nullptr || prec < 0 || prec > 1) return MD_ERR_BAD_ARG;
  const int64_t n_items = bytes / 16;
  int blocks = (int)((n_items + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w,
                     (uint4*)wpk, rows, kdim, taps, s_row, s_k, s_tap, nt, kc, n_items, prec);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
