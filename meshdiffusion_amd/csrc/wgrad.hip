// Weight gradient of the 3x3x3 convolutions and the 1x1x1 NIN layers (autograd of layers.py:118-124, :573-582):
//
//   dW[co][ci][dz][dy][dx] = sum_{sample, position} dY[co][pos] * A[ci][pos + off(dz,dy,dx)]
//
// on PB16 operands (backward.hip): bf16 hi/lo planes, [position on the zero-padded grid][B/8][plane][channel][8 samples].
// The contraction index is (position, sample) with the 8 samples innermost, so every tap is a plain shift of the
// position index and one MFMA k-group (8 consecutive k of one lane) is "8 samples of one channel at one position".
//
// Why a dedicated kernel: the contraction is ~2 M long at 64^3 x 8 samples while the output is tiny, so the only
// reuse there is comes from holding many (co, ci, tap) accumulators per workgroup.  One workgroup owns a 128 co x
// 128 ci tile for the THREE dx taps of one (dz, dy) group (96 accumulator VGPRs per lane, 8 waves) and walks a
// contiguous range of positions: per position it stages 4 KB of dY and 4 KB of A through LDS for 3 x 128 x 128 x 8
// MACs -- the same A positions serve dx = -1, 0, +1 (window of 10 positions per 8 of dY).  The nine (dz, dy) groups
// and the K ranges are separate workgroups; a block's XCD is chosen so that the nine groups of one K range share an
// L2 (they read the same dY and row-shifted A).  Partial sums go to a workspace and a second kernel reduces them in
// a fixed order into dW (deterministic; no float atomics).
//
// Arithmetic: bf16x3 (lo*hi + hi*lo + hi*hi, fp32 accumulate), like the forward.
#include "md_common.h"

namespace {

constexpr int WG_THREADS = 512;
constexpr int WG_TILE = 128;          // co and ci tile edge
constexpr int WG_STAGE = 8;           // dY positions per LDS stage (k = 64): one x-segment of one interior grid row
constexpr int WG_A_VEC = WG_STAGE * 2 * WG_TILE;            // 16-byte vectors of dY per stage      (2048)

struct WgArgs {
  const uint16_t* dy;    // PB16 of dY, pointing at padded position 0 (guard skipped)
  const uint16_t* act;   // PB16 of A, same
  float* partial;        // [ksplit][ngroups][NTAP][co_tiles*128][ci_tiles*128]
  int a_ch, b_ch;        // channel counts of the two PB16 tensors
  int co_tiles, ci_tiles, ngroups, ksplit;
  int rows, cols;        // valid co / ci (waves whose whole sub-tile lies outside skip their MFMAs)
  int bgn;               // B / 8
  int S;                 // grid edge in y and x; the padded edge is S + 2 pad
  int Dz;                // grid depth (z): S, or S / zsplit when samples are cut into z-slabs (md_to_pb16)
  int pad;               // halo of the PB16 grids: 1 (3x3x3, 1x1x1) or 2 (5x5x5)
  int ksz;               // 1, 3 or 5
  int spr;               // stages per grid row = ceil(S / 8)
};

// The contraction walks the INTERIOR rows of the padded grid only (dY is zero on the halo): stage st of sample
// group bg = row (z, y), x-segment seg -> padded positions p0 .. p0+7, p0 = ((z+1)(S+2) + y+1)(S+2) + 1 + 8 seg.
struct WgCursor {
  int bg, z, y, seg;
  __device__ void init(int st, int S, int Dz, int spr) {
    seg = st % spr;
    int row = st / spr;
    y = row % S;
    row /= S;
    z = row % Dz;
    bg = row / Dz;
  }
  __device__ void next(int S, int Dz, int spr) {
    if (++seg == spr) {
      seg = 0;
      if (++y == S) {
        y = 0;
        if (++z == Dz) { z = 0; ++bg; }
      }
    }
  }
  __device__ int64_t p0(int S, int pad) const {
    const int sp = S + 2 * pad;
    return ((int64_t)(z + pad) * sp + (y + pad)) * sp + pad + seg * WG_STAGE;
  }
};

// One half of a stage (4 of its 8 dY positions = two k-steps) for one wave.  Fragments: lanes 0-31 take position q,
// lanes 32-63 position q + 2.  k-step k2 uses the dY pair (base, base+2), base = 4 hq + k2, and tap t pairs it with the
// A-window pair (base + t, base + t + 2): NTAP + 1 distinct A fragments serve the 2 NTAP (k-step, tap) combinations.
// NS = row sub-tiles this wave owns (2, or 1 when the second lies outside `rows`).
template <int NTAP, int NS>
__device__ __forceinline__ void wg_half(f32x16 (&acc)[2][NTAP], const uint4* sa, const uint4* sb, const int hq, const int a_idx,
                                        const int b_idx) {
  bf16x8 bh[NTAP + 1], bl[NTAP + 1];
#pragma unroll
  for (int j = 0; j < NTAP + 1; ++j) {
    const int idx = b_idx + (hq * 4 + j) * 2 * WG_TILE;
    bh[j] = __builtin_bit_cast(bf16x8, sb[idx]);
    bl[j] = __builtin_bit_cast(bf16x8, sb[idx + WG_TILE]);
  }
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2) {
    bf16x8 ah[NS], al[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int idx = a_idx + (hq * 4 + k2) * 2 * WG_TILE + s * 32;
      ah[s] = __builtin_bit_cast(bf16x8, sa[idx]);
      al[s] = __builtin_bit_cast(bf16x8, sa[idx + WG_TILE]);
    }
    // three passes over the independent (s, t) accumulators: no back-to-back dependent MFMAs
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[s], bh[k2 + t], acc[s][t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bl[k2 + t], acc[s][t], 0, 0, 0);
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int t = 0; t < NTAP; ++t) acc[s][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[s], bh[k2 + t], acc[s][t], 0, 0, 0);
  }
}

// FULL: rows, cols multiples of 128 and S a multiple of 8: neither the MFMA section nor the global loads carry any
// predication (exec-mask branches between MFMA groups cost the scheduler its freedom to overlap LDS reads and
// memory instructions with the matrix pipe).
// DBG (tools/bench_wgrad.py only): 1 = no global loads in the loop, 2 = no MFMAs; results are then meaningless.
template <int NTAP, bool FULL, int DBG>
__global__ __launch_bounds__(WG_THREADS) void md_wgrad_kernel(const WgArgs g) {
  constexpr int WIN = NTAP == 3 ? WG_STAGE + 2 : WG_STAGE;   // A positions per stage
  constexpr int B_VEC = WIN * 2 * WG_TILE;
  constexpr int A_ITERS = WG_A_VEC / WG_THREADS, B_ITERS = B_VEC / WG_THREADS;
  __shared__ __attribute__((aligned(16))) uint4 smem[2 * (WG_A_VEC + B_VEC)];

  // ---- which (K range, unit) is this block.  Work items w = r * units + u are dealt to the XCDs in runs of 32 (the
  //      CUs of one XCD; block b lands on XCD b % 8): the units of one K range share an L2, and no XCD is handed more
  //      workgroups than it has CUs while others idle (one 144 KB-LDS workgroup fits per CU).
  const int units = g.co_tiles * g.ci_tiles * g.ngroups;
  const int b = blockIdx.x, slot = b >> 3;
  const int w = (slot >> 5) * 256 + (b & 7) * 32 + (slot & 31);
  if (w >= g.ksplit * units) return;
  const int r = w / units, u = w - r * units;
  const int grp = u % g.ngroups;
  const int tci = (u / g.ngroups) % g.ci_tiles, tco = u / (g.ngroups * g.ci_tiles);
  const int co0 = tco * WG_TILE, ci0 = tci * WG_TILE;
  const int S = g.S, sp = S + 2 * g.pad;
  const int total = g.bgn * g.Dz * S * g.spr;
  const int per = (total + g.ksplit - 1) / g.ksplit;
  const int st0 = min(total, r * per), st1 = min(total, st0 + per);
  int64_t off = 0;
  if (NTAP == 3) {
    if (g.ksz == 3) {          // group = (dz, dy); the window starts at dx = -1
      off = ((int64_t)(grp / 3 - 1) * sp + (grp % 3 - 1)) * sp - 1;
    } else {                   // 5x5x5: group = ((dz, dy), h); window h covers dx = -2 + 3h .. -2 + 3h + 2 (dx = +3 is discarded)
      const int zy = grp >> 1, h = grp & 1;
      off = ((int64_t)(zy / 5 - 2) * sp + (zy % 5 - 2)) * sp - 2 + 3 * h;
    }
  }

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3, half = lane >> 5, l31 = lane & 31;
  const bool row_on[2] = {co0 + wr * 64 < g.rows, co0 + wr * 64 + 32 < g.rows};
  const bool col_on = ci0 + wc * 32 < g.cols;

  f32x16 acc[2][NTAP];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[s][t][i] = 0.f;

  // ---- global -> register staging (vector v = (position, plane, channel row) of the tile) ----
  uint4 ra[A_ITERS], rb[B_ITERS];
  const int64_t a_pos = (int64_t)g.bgn * 2 * g.a_ch * 8, b_pos = (int64_t)g.bgn * 2 * g.b_ch * 8;   // bf16 per position
  const int frow = tid & (WG_TILE - 1), fpl = (tid >> 7) & 1, fpos = tid >> 8;   // v = tid + i*512: pos = fpos + 2 i
  const bool a_ok = co0 + frow < g.a_ch, b_ok = ci0 + frow < g.b_ch;
  WgCursor cur;
  cur.init(st0, S, g.Dz, g.spr);
  auto fetch = [&]() {   // the stage `cur` points at
    const int64_t p0 = cur.p0(S, g.pad);
    const int valid = min(WG_STAGE, S - cur.seg * WG_STAGE);
    const uint16_t* pa = g.dy + (p0 + fpos) * a_pos + ((int64_t)(cur.bg * 2 + fpl) * g.a_ch + co0 + frow) * 8;
    const uint16_t* pb = g.act + (p0 + off + fpos) * b_pos + ((int64_t)(cur.bg * 2 + fpl) * g.b_ch + ci0 + frow) * 8;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) {
      uint4 x = make_uint4(0, 0, 0, 0);
      if (FULL || (a_ok && fpos + 2 * i < valid)) x = *(const uint4*)(pa + 2 * i * a_pos);
      ra[i] = x;
    }
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) {
      uint4 x = make_uint4(0, 0, 0, 0);
      if (FULL || b_ok) x = *(const uint4*)(pb + 2 * i * b_pos);
      rb[i] = x;
    }
  };
  auto stash = [&](int buf) {
    uint4* sa = smem + buf * (WG_A_VEC + B_VEC);
    uint4* sb = sa + WG_A_VEC;
#pragma unroll
    for (int i = 0; i < A_ITERS; ++i) sa[tid + i * WG_THREADS] = ra[i];
#pragma unroll
    for (int i = 0; i < B_ITERS; ++i) sb[tid + i * WG_THREADS] = rb[i];
  };

  if (st0 < st1) {
    fetch();
    stash(0);
  }
  __syncthreads();
  // wave-uniform: does this wave own any output at all / both row sub-tiles
  const bool active = (FULL || (col_on && row_on[0])) && !(DBG & 2);
  const bool both = FULL || row_on[1];
  const int a_idx = (2 * half) * 2 * WG_TILE + wr * 64 + l31, b_idx = (2 * half) * 2 * WG_TILE + wc * 32 + l31;
  int st = st0;
  // steady state: a next stage exists.  Its operands are requested first, written to the other LDS buffer between
  // the two halves (that buffer is free since the last barrier; the writes overlap the second half's MFMAs).
  for (; st + 1 < st1; ++st) {
    const int buf = (st - st0) & 1;
    if (!(DBG & 1)) {
      cur.next(S, g.Dz, g.spr);
      fetch();
    }
    // keep the requests up here: left alone the scheduler sinks them next to their use (the LDS writes below) to
    // save registers, which exposes the whole memory latency
    __builtin_amdgcn_sched_barrier(0);
    const uint4* sa = smem + buf * (WG_A_VEC + B_VEC);
    const uint4* sb = sa + WG_A_VEC;
    if (active) {
      if (both) wg_half<NTAP, 2>(acc, sa, sb, 0, a_idx, b_idx);
      else wg_half<NTAP, 1>(acc, sa, sb, 0, a_idx, b_idx);
    }
    stash(buf ^ 1);
    if (active) {
      if (both) wg_half<NTAP, 2>(acc, sa, sb, 1, a_idx, b_idx);
      else wg_half<NTAP, 1>(acc, sa, sb, 1, a_idx, b_idx);
    }
    __syncthreads();
  }
  if (st < st1 && active) {   // last stage: nothing left to prefetch
    const int buf = (st - st0) & 1;
    const uint4* sa = smem + buf * (WG_A_VEC + B_VEC);
    const uint4* sb = sa + WG_A_VEC;
    if (both) {
      wg_half<NTAP, 2>(acc, sa, sb, 0, a_idx, b_idx);
      wg_half<NTAP, 2>(acc, sa, sb, 1, a_idx, b_idx);
    } else {
      wg_half<NTAP, 1>(acc, sa, sb, 0, a_idx, b_idx);
      wg_half<NTAP, 1>(acc, sa, sb, 1, a_idx, b_idx);
    }
  }

  // ---- partial sums: [r][grp][t][co][ci], lane -> ci (coalesced 128 B rows) ----
  const int RT = g.co_tiles * WG_TILE, CT = g.ci_tiles * WG_TILE;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      float* o = g.partial + ((((int64_t)r * g.ngroups + grp) * NTAP + t) * RT + co0 + wr * 64 + s * 32) * CT + ci0 + wc * 32 + l31;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        o[(int64_t)row * CT] = acc[s][t][i];
      }
    }
}

// dw[row*s_row + col*s_k + tap*s_tap] += sum_r partial[r][slot][row][col]; slot = grp * NTAP + t maps to
//   ksz 1: tap 0;  ksz 3: tap = slot;  ksz 5: grp = (dz*5+dy)*2 + h, dx = 3h + t (dx = 5 does not exist: skipped)
// (measured alternative: one thread per (row, col) writing its taps as one contiguous run -- 0.12 ms instead of 0.07 ms per
// launch: 27x fewer threads cost more than the scattered read-modify-writes)
__global__ void md_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int rows, int cols, int RT,
                                       int CT, int nslots, int ksz, int ksplit, int64_t s_row, int64_t s_k, int64_t s_tap) {
  const int64_t total = (int64_t)nslots * rows * cols;
  const int64_t slab = (int64_t)nslots * RT * CT;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % cols);
    const int row = (int)((i / cols) % rows);
    const int slot = (int)(i / ((int64_t)cols * rows));
    int tap = slot;
    if (ksz == 5) {
      const int grp = slot / 3, t = slot - grp * 3, dx = 3 * (grp & 1) + t;
      if (dx >= 5) continue;
      tap = (grp >> 1) * 5 + dx;
    }
    const float* p = partial + ((int64_t)slot * RT + row) * CT + col;
    float s = 0.f;
    for (int r = 0; r < ksplit; ++r) s += p[r * slab];
    dw[row * s_row + col * s_k + tap * s_tap] += s;
  }
}


// The same reduction for the 3x3x3 conv layout (dw[co][ci][27]: s_k == 27, s_tap == 1) with an LDS transpose: a block takes
// one row and 64 columns, reads the 27 slot slabs in 256-byte runs along col and writes the 64 x 27 = 1728 floats of dW as ONE
// contiguous run (the generic kernel's adjacent threads write 108 bytes apart: 16x write amplification, 0.1 ms for a
// 512 x 512 layer, 6 ms per training step over the 8^3 / 4^3 levels).
__global__ __launch_bounds__(256) void md_wgrad_reduce27_kernel(const float* __restrict__ partial, float* __restrict__ dw, int rows,
                                                                int cols, int RT, int CT, int ksplit, int64_t s_row) {
  __shared__ float tile[64 * 28];
  const int tid = threadIdx.x;
  const int cb = cols >> 6;
  const int row = blockIdx.x / cb, c0 = (blockIdx.x % cb) << 6;
  const int64_t plane = (int64_t)RT * CT, slab = 27 * plane;
  const int c = tid & 63;
  for (int slot = tid >> 6; slot < 27; slot += 4) {
    const float* p = partial + (int64_t)slot * plane + (int64_t)row * CT + c0 + c;
    float sum = 0.f;
    for (int r = 0; r < ksplit; ++r) sum += p[r * slab];
    tile[c * 28 + slot] = sum;
  }
  __syncthreads();
  float* d = dw + (int64_t)row * s_row + (int64_t)c0 * 27;
  for (int e = tid; e < 64 * 27; e += 256) d[e] += tile[(e / 27) * 28 + (e % 27)];
}

}  // namespace

#ifdef MD_BUILD_ABLATIONS      // include/meshdiffusion_hip_experimental.h: a process-wide knob, absent from the default library
static int md_wgrad_debug = 0;
extern "C" void md_wgrad_set_debug(int32_t flags) { md_wgrad_debug = flags; }
#else
static constexpr int md_wgrad_debug = 0;
#endif

static int md_wgrad_slots(int taps) { return taps == 1 ? 1 : (taps == 27 ? 27 : (taps == 125 ? 150 : -1)); }

extern "C" int64_t md_wgrad_workspace_bytes(int32_t rows, int32_t cols, int32_t taps, int32_t ksplit) {
  if (rows <= 0 || cols <= 0 || md_wgrad_slots(taps) < 0 || ksplit <= 0) return MD_ERR_BAD_ARG;
  const int64_t RT = (rows + WG_TILE - 1) / WG_TILE * WG_TILE, CT = (cols + WG_TILE - 1) / WG_TILE * WG_TILE;
  return (int64_t)ksplit * md_wgrad_slots(taps) * RT * CT * 4;
}

extern "C" int md_wgrad(const void* dy_pb, const void* act_pb, float* dw, void* workspace, int64_t workspace_bytes,
                        int32_t batch, int32_t a_ch, int32_t b_ch, int32_t rows, int32_t cols, int32_t D, int32_t H,
                        int32_t W, int32_t guard, int32_t taps, int32_t ksplit, int64_t s_row, int64_t s_k, int64_t s_tap,
                        void* stream) {
  if (!dy_pb || !act_pb || !dw || !workspace || batch <= 0 || a_ch <= 0 || b_ch <= 0 || (a_ch % 8) ||
      (b_ch % 8) || rows <= 0 || rows > a_ch || cols <= 0 || cols > b_ch || D <= 0 || H <= 0 || W != H ||
      md_wgrad_slots(taps) < 0 || ksplit <= 0)
    return MD_ERR_BAD_ARG;
  const int pad = taps == 125 ? 2 : 1;
  const int sp = H + 2 * pad;   // y / x padded edge (z only sets the number of rows)
  // a stage may run up to 7 positions past the end of its row (masked dY, but the A window is read): stay in the guard
  if (guard < pad * (sp * sp + sp + 1) + WG_STAGE + 2) return MD_ERR_BAD_ARG;
  if (workspace_bytes < md_wgrad_workspace_bytes(rows, cols, taps, ksplit)) return MD_ERR_BAD_ARG;
  WgArgs g;
  g.bgn = (batch + 7) / 8;
  g.a_ch = a_ch; g.b_ch = b_ch;
  g.dy = (const uint16_t*)dy_pb + (int64_t)guard * g.bgn * 2 * a_ch * 8;
  g.act = (const uint16_t*)act_pb + (int64_t)guard * g.bgn * 2 * b_ch * 8;
  g.partial = (float*)workspace;
  g.co_tiles = (rows + WG_TILE - 1) / WG_TILE; g.ci_tiles = (cols + WG_TILE - 1) / WG_TILE;
  g.ngroups = taps == 27 ? 9 : (taps == 125 ? 50 : 1);
  g.pad = pad;
  g.ksz = taps == 27 ? 3 : (taps == 125 ? 5 : 1);
  g.ksplit = ksplit;
  g.rows = rows; g.cols = cols;
  g.S = H;
  g.Dz = D;
  g.spr = (H + WG_STAGE - 1) / WG_STAGE;
  const int units = g.co_tiles * g.ci_tiles * g.ngroups;
  const int64_t blocks = ((int64_t)ksplit * units + 255) / 256 * 256;
  if (blocks > 0x7fffffff) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  // FULL: no predication anywhere -- whole 128-channel tiles on both operands and whole 8-position stages
  const bool full = (rows % WG_TILE) == 0 && (cols % WG_TILE) == 0 && (H % WG_STAGE) == 0;
  const dim3 grid((unsigned)blocks), blk(WG_THREADS);
  const hipStream_t hs = (hipStream_t)stream;
  if (taps != 1 && full) {
    switch (md_wgrad_debug) {   // timing ablations exist for the dominant instantiation only
#ifdef MD_BUILD_ABLATIONS
      case 1: hipLaunchKernelGGL((md_wgrad_kernel<3, true, 1>), grid, blk, 0, hs, g); break;
      case 2: hipLaunchKernelGGL((md_wgrad_kernel<3, true, 2>), grid, blk, 0, hs, g); break;
      case 3: hipLaunchKernelGGL((md_wgrad_kernel<3, true, 3>), grid, blk, 0, hs, g); break;
#endif
      default: hipLaunchKernelGGL((md_wgrad_kernel<3, true, 0>), grid, blk, 0, hs, g); break;
    }
  } else if (taps != 1) hipLaunchKernelGGL((md_wgrad_kernel<3, false, 0>), grid, blk, 0, hs, g);
  else if (full) hipLaunchKernelGGL((md_wgrad_kernel<1, true, 0>), grid, blk, 0, hs, g);
  else hipLaunchKernelGGL((md_wgrad_kernel<1, false, 0>), grid, blk, 0, hs, g);
  MD_HIP_CHECK_LAUNCH();
  const int nslots = md_wgrad_slots(taps);
  if (taps == 27 && s_tap == 1 && s_k == 27 && (cols % 64) == 0 && (int64_t)rows * (cols / 64) <= 0x7fffffff) {
    hipLaunchKernelGGL(md_wgrad_reduce27_kernel, dim3((unsigned)(rows * (cols / 64))), dim3(256), 0, (hipStream_t)stream,
                       (const float*)workspace, dw, rows, cols, g.co_tiles * WG_TILE, g.ci_tiles * WG_TILE, ksplit, s_row);
    MD_HIP_CHECK_LAUNCH();
    return MD_OK;
  }
  const int64_t total = (int64_t)nslots * rows * cols;
  int rb = (int)((total + 255) / 256);
  if (rb > 4096) rb = 4096;
  hipLaunchKernelGGL(md_wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, dw,
                     rows, cols, g.co_tiles * WG_TILE, g.ci_tiles * WG_TILE, nslots, g.ksz, ksplit, s_row, s_k, s_tap);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
