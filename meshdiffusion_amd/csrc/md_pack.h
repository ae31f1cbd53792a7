// Per-item bodies of the two weight-packing kernels, shared by the single-weight launches (md_pack_weights in gemm_conv.hip,
// md_wino_pack_weights in conv3_wino.hip) and the batched launch md_pack_batch (pack_batch.hip): same arithmetic, same bits.
#pragma once
#include "md_common.h"

// WPK tiles [rows/NT][K/KC][taps][KC/8][2][NT][8]: item = 16 bytes = 8 consecutive k of one row, one plane
__device__ __forceinline__ uint4 md_pack_wpk_item(const float* __restrict__ w, int rows, int kdim, int taps, int64_t s_row, int64_t s_k,
                                                  int64_t s_tap, int nt, int kc, int prec, int64_t item) {
  const int kg = kc / 8;
  const int ncc = (kdim + kc - 1) / kc;
  int64_t r = item;
  const int rr = (int)(r % nt); r /= nt;
  const int part = (int)(r % 2); r /= 2;
  const int g = (int)(r % kg); r /= kg;
  const int tap = (int)(r % taps); r /= taps;
  const int cc = (int)(r % ncc); r /= ncc;
  const int rt = (int)r;
  const int row = rt * nt + rr;
  uint32_t v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = cc * kc + g * 8 + e;
    float x = 0.f;
    if (row < rows && k < kdim) x = w[row * s_row + k * s_k + tap * s_tap];
    uint32_t hi, lo;
    if (prec == MD_PREC_FP16X2) md_split_f16(x, hi, lo); else md_split(x, hi, lo);
    v[e] = part ? lo : hi;
  }
  return make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
}

// Winograd F(2,3) weight fragments [cout/128][cin/16][tap (kd,kh) 9][f 4][row tile 4][plane 2][h 2][row 32][8 bf16]
__device__ __forceinline__ uint4 md_pack_wino_item(const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k, int flip,
                                                   int64_t id) {
  int64_t r = id;
  const int row = (int)(r % 32); r /= 32;
  const int h = (int)(r % 2); r /= 2;
  const int plane = (int)(r % 2); r /= 2;
  const int rtile = (int)(r % 4); r /= 4;
  const int f = (int)(r % 4); r /= 4;
  const int tap = (int)(r % 9); r /= 9;
  const int nchunk = cin / 16;
  const int chunk = (int)(r % nchunk); r /= nchunk;
  const int ct = (int)r;
  const int co = (ct * 4 + rtile) * 32 + row;
  uint32_t word[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t half[2];
#pragma unroll
    for (int e2 = 0; e2 < 2; ++e2) {
      const int ci = chunk * 16 + h * 8 + 2 * q + e2;
      // element (row, k, kd, kh, kw) of the convolution being packed = w[row * s_row + k * s_k + t27], t27 = (kd*3+kh)*3+kw,
      // or 26 - t27 when `flip` (data gradient: W'[ci][co][t] = W[co][ci][26 - t], read in place)
      const float* g = w + (int64_t)co * s_row + (int64_t)ci * s_k;
      const int t0 = tap * 3;
      const float g0 = g[flip ? 26 - t0 : t0], g1 = g[flip ? 25 - t0 : t0 + 1], g2 = g[flip ? 24 - t0 : t0 + 2];
      const float G = f == 0 ? g0 : f == 1 ? (g0 + g1 + g2) * 0.5f : f == 2 ? (g0 - g1 + g2) * 0.5f : g2;
      uint32_t hi, lo;
      md_split(G, hi, lo);
      half[e2] = plane ? lo : hi;
    }
    word[q] = half[0] | (half[1] << 16);
  }
  (void)cout;
  return make_uint4(word[0], word[1], word[2], word[3]);
}

// Winograd F(2,3) weight fragments of the "f16f8" arithmetic (md_conv3_wino_f8; md_split_f16f8 in md_common.h):
//   [cout/128][pair p = step / 2][f 4][row tile 4][piece 4][lane 64][16 B],   step s = chunk * 9 + (kd, kh) tap, cin / 16 chunks
//   piece 0 / 1: the fp16 MFMA A fragment of step 2p / 2p + 1: lane (row = lane & 31, h = lane >> 5) holds hi(G') of channels
//                chunk * 16 + 8 h + 0..7
//   piece 2 / 3: the two halves of the K-concatenated fp8 A fragment: lane (row, h) belongs to step 2p + h and holds
//                [e4m3(lo(G') 2^11) ch 0..7 | e4m3(G') ch 0..7] (piece 2) and the same of channels 8..15 (piece 3) -- lo first:
//                it meets the activation item's e4m3(t) half, the plain image the e4m3(lo(t) 2^11) half.
// G' = G * wscale with wscale = 2^sw chosen from the tensor's max |w| so that max |G'| lies in [128, 256) (e4m3's top binades,
// far from fp16's subnormals whatever the layer's weight scale); the conv's epilogue multiplies by 2^-sw.
__device__ __forceinline__ float md_wino_f8_wscale(float amax) {       // 2^sw; amax = max |w| of the raw 3x3x3 weights
  if (!(amax > 0.f) || !(amax < 1e30f)) return 1.f;
  const int e = ilogbf(1.5f * amax);                                   // |G| <= 1.5 max |w|
  int sw = 7 - e;
  sw = sw < -100 ? -100 : (sw > 100 ? 100 : sw);
  return ldexpf(1.f, sw);
}
// eq (may be null): the per-input-channel equaliser of md_wino_equaliser; the fragments hold G' / eq[ci] (eq is a power of two: exact).
// flip: the data-gradient orientation W'[ci][co][t] = W[co][ci][26 - t], read in place (as md_pack_wino_item)
__device__ __forceinline__ uint4 md_pack_wino_f8_item(const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k, float wscale,
                                                      int64_t id, const float* __restrict__ eq = nullptr, int flip = 0) {
  int64_t r = id;
  const int lane = (int)(r % 64); r /= 64;
  const int piece = (int)(r % 4); r /= 4;
  const int rtile = (int)(r % 4); r /= 4;
  const int f = (int)(r % 4); r /= 4;
  const int npairs = (cin / 16) * 9 / 2;
  const int p = (int)(r % npairs); r /= npairs;
  const int rtb = (int)r;
  const int row = lane & 31, h = lane >> 5;
  const int co = (rtb * 4 + rtile) * 32 + row;
  const int step = 2 * p + (piece < 2 ? piece : h);
  const int chunk = step / 9, tap = step % 9;
  const int ci0 = chunk * 16 + (piece < 2 ? h : piece - 2) * 8;
  float g8[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float* g = w + (int64_t)co * s_row + (int64_t)(ci0 + e) * s_k;
    const int t0 = tap * 3;
    const float g0 = g[flip ? 26 - t0 : t0], g1 = g[flip ? 25 - t0 : t0 + 1], g2 = g[flip ? 24 - t0 : t0 + 2];
    const float G = f == 0 ? g0 : f == 1 ? (g0 + g1 + g2) * 0.5f : f == 2 ? (g0 - g1 + g2) * 0.5f : g2;
    g8[e] = G * (eq ? wscale / eq[ci0 + e] : wscale);
  }
  uint4 hi;
  uint32_t q[2], ql[2];
  md_split_f16f8(g8, hi, q, ql);
  (void)cout;
  return piece < 2 ? hi : make_uint4(ql[0], ql[1], q[0], q[1]);
}

// "f16f6" weight fragments (md_conv3_wino_f6; md_split_f16f6 in md_common.h): the layout of md_pack_wino_f8_item, pieces 2 / 3 =
// the two halves of the lane's 32-byte MX record: lane (row, h) belongs to step 2p + h and its K block is the 16 input channels of
// that step's chunk: [e2m3 codes of (lo(G') 2^11, G') x 16, interleaved | E8M0 byte of the block, 2^-11 folded in | 0].
__device__ __forceinline__ uint4 md_pack_wino_f6_item(const float* __restrict__ w, int cout, int cin, int64_t s_row, int64_t s_k, float wscale,
                                                      int64_t id, const float* __restrict__ eq = nullptr, int flip = 0) {
  const int piece = (int)((id / 64) % 4);
  if (piece < 2) return md_pack_wino_f8_item(w, cout, cin, s_row, s_k, wscale, id, eq, flip);      // the fp16 fragments are the same
  int64_t r = id;
  const int lane = (int)(r % 64); r /= 64;
  r /= 4;
  const int rtile = (int)(r % 4); r /= 4;
  const int f = (int)(r % 4); r /= 4;
  const int npairs = (cin / 16) * 9 / 2;
  const int p = (int)(r % npairs); r /= npairs;
  const int rtb = (int)r;
  const int row = lane & 31, h = lane >> 5;
  const int co = (rtb * 4 + rtile) * 32 + row;
  const int step = 2 * p + h;
  const int chunk = step / 9, tap = step % 9;
  float g16[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float* g = w + (int64_t)co * s_row + (int64_t)(chunk * 16 + e) * s_k;
    const int t0 = tap * 3;
    const float g0 = g[flip ? 26 - t0 : t0], g1 = g[flip ? 25 - t0 : t0 + 1], g2 = g[flip ? 24 - t0 : t0 + 2];
    const float G = f == 0 ? g0 : f == 1 ? (g0 + g1 + g2) * 0.5f : f == 2 ? (g0 - g1 + g2) * 0.5f : g2;
    g16[e] = G * (eq ? wscale / eq[chunk * 16 + e] : wscale);
  }
  uint4 h0, h1, r0, r1;
  md_split_f16f6(g16, true, -11, h0, h1, r0, r1);
  return piece == 2 ? r0 : r1;
}
