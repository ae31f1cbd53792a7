// md_pack_batch: every stale packed weight of a training step in ONE launch.
//
// A training step changes all 365 M parameters, so each weight is re-packed once per step (WPK tiles for the direct / GEMM
// kernels, Winograd fragments for md_conv3_wino, forward and data-gradient forms): ~260 launches of md_pack_weights /
// md_wino_pack_weights per res64 step, 22-29 us each whatever their size (profiles/r03_train_kt.summary.txt: 6.8 ms per
// step).  The host queues the jobs and hands them over as one table; a block finds its job by binary search over the
// jobs' first block and runs the same per-item code as the single-weight kernels (md_pack.h): bit-identical tiles.
#include "md_common.h"
#include "md_pack.h"

__global__ __launch_bounds__(256) void md_pack_batch_kernel(const MdPackJob* __restrict__ jobs, int n_jobs) {
  int lo = 0, hi = n_jobs - 1;                 // last job whose first block <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const MdPackJob J = jobs[lo];
  const int64_t item = ((int64_t)blockIdx.x - J.block0) * 256 + threadIdx.x;
  if (item >= J.n_items) return;
  uint4* out = (uint4*)J.out;
  if (J.kind == MD_PACK_WINO) out[item] = md_pack_wino_item(J.w, J.rows, J.kdim, J.s_row, J.s_k, J.flip, item);
  else if (J.kind == MD_PACK_WINO_F6) {
    // f16f6 fragments with a FIXED power-of-two pre-scale 2^prec (no max |w| pass: the training step re-packs every weight, the MX
    // records carry their own block scales and 2^prec only has to keep the fp16 fragments off the subnormals); the last 16 items of
    // the job are the 256-byte header {0, sw, 2^-sw, 6} md_conv3_wino_f6 reads its descale from
    const int64_t n_frag = J.n_items - 16;
    if (item < n_frag) out[item] = md_pack_wino_f6_item(J.w, J.rows, J.kdim, J.s_row, J.s_k, ldexpf(1.f, J.prec), item, nullptr, J.flip);
    else out[item] = item == n_frag ? make_uint4(0u, (uint32_t)J.prec, __float_as_uint(ldexpf(1.f, -J.prec)), 6u) : make_uint4(0u, 0u, 0u, 0u);
  }
  else out[item] = md_pack_wpk_item(J.w, J.rows, J.kdim, J.taps, J.s_row, J.s_k, J.s_tap, J.nt, J.kc, J.prec, item);
}

// MD_PACK_WPK jobs with taps > 1 (conv / conv_dgrad weights: most of the 365 M parameters sit in the 512-channel 3x3x3
// layers), block-cooperative: the per-item form gathers 8 floats 108 B (or a whole output-channel stride) apart per 16-byte
// item -- 1.1 TB/s over the step's 5 GB.  Here a block takes 16 rows x one K chunk x all taps: the source is read in
// contiguous runs (rows x taps or k x taps, whichever is contiguous in memory) into LDS, then every 16 lanes write one
// 256-byte run of a (tap, channel group, plane) tile row.  Same values, same split: bit-identical tiles.
namespace {
constexpr int PT_ROWS = 16;
constexpr int PT_MAX_FLOATS = 13824;              // 16 rows x 32 k x 27 taps
}  // namespace

// TAPS / KC compile-time (index arithmetic by constants); TAPS = 0: generic (runtime taps and kc, scalar loads).
// Interior blocks of 16-byte aligned weights read the source as float4.
template <int TAPS, int KC>
__global__ __launch_bounds__(256) void md_pack_tiled_kernel(const MdPackJob* __restrict__ jobs, int n_jobs) {
  __shared__ __attribute__((aligned(16))) float tile[PT_MAX_FLOATS + 4 * PT_ROWS];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const MdPackJob J = jobs[lo];
  const int taps = TAPS ? TAPS : J.taps, kc = TAPS ? KC : J.kc, nt = J.nt, kg = kc / 8;
  const int ncc = (J.kdim + kc - 1) / kc;
  const int nrb = ((J.rows + nt - 1) / nt) * (nt / PT_ROWS);          // 16-row blocks of the padded row range
  const int lb = (int)((int64_t)blockIdx.x - J.block0);
  const int rb = lb % nrb, cc = lb / nrb;
  const int row0 = rb * PT_ROWS;
  const int rstride = kc * taps + 4;                                  // +4: the 16 rows of a store wave hit 16 different banks, rows stay 16-byte aligned
  const int n = PT_ROWS * kc * taps;
  const bool k_inner = (J.s_k < 0 ? -J.s_k : J.s_k) <= (J.s_row < 0 ? -J.s_row : J.s_row);   // which of (row, k) is the closer stride
  // the taps of a (row, k) pair are contiguous in memory, ascending or (s_tap = -1: data-gradient weights, w points at the
  // last tap) descending: the tile holds them in MEMORY order, the flip happens where they are read back
  const bool flip = J.s_tap < 0;
  const float* wraw = flip ? J.w - (taps - 1) : J.w;
  const int64_t chunk0 = (int64_t)row0 * J.s_row + (int64_t)cc * kc * J.s_k;   // first element of the block's source region
  // memory runs: k_inner: one per row (kc * taps floats, row stride s_row); else one per k (16 * taps floats, stride s_k)
  const int run = (k_inner ? kc : PT_ROWS) * taps;
  const int64_t run_stride = k_inner ? J.s_row : J.s_k;
  const bool interior = row0 + PT_ROWS <= J.rows && (cc + 1) * kc <= J.kdim;
  const bool vec = TAPS != 0 && interior && (((uintptr_t)(wraw + chunk0) | (uintptr_t)(run_stride * 4)) & 15) == 0 && (run & 3) == 0 &&
                   (k_inner ? J.s_k == taps : J.s_row == taps);
  if (vec) {
    const int run4 = run >> 2, n4 = n >> 2;
    for (int i4 = threadIdx.x; i4 < n4; i4 += 256) {
      const int ri = i4 / run4, off4 = i4 - ri * run4;               // run index (row or k), float4 inside the run
      const f32x4 v = *(const f32x4*)(wraw + chunk0 + (int64_t)ri * run_stride + off4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = off4 * 4 + e;                                   // element inside the run: (inner index, tap)
        const int in = o / taps, t = o - in * taps;
        const int r = k_inner ? ri : in, k = k_inner ? in : ri;
        tile[r * rstride + k * taps + t] = v[e];
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < n; idx += 256) {
      const int t = idx % taps;
      const int q = idx / taps;
      const int r = k_inner ? q / kc : q % PT_ROWS;
      const int k = k_inner ? q % kc : q / PT_ROWS;
      const int row = row0 + r, kk = cc * kc + k;
      float v = 0.f;
      if (row < J.rows && kk < J.kdim) v = wraw[(int64_t)row * J.s_row + (int64_t)kk * J.s_k + t];
      tile[r * rstride + k * taps + t] = v;
    }
  }
  __syncthreads();
  uint4* out = (uint4*)J.out;
  const int tasks = taps * kg * PT_ROWS;
  for (int task = threadIdx.x; task < tasks; task += 256) {
    const int rr = task % PT_ROWS;
    const int tg = task / PT_ROWS;
    const int g = tg % kg, tap = tg / kg;
    const int mt = flip ? taps - 1 - tap : tap;                      // memory position of output tap `tap`
    uint32_t h[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float x = tile[rr * rstride + (g * 8 + e) * taps + mt];
      if (J.prec == MD_PREC_FP16X2) md_split_f16(x, h[e], l[e]); else md_split(x, h[e], l[e]);
    }
    const int row = row0 + rr;
    const int rt = row / nt, rin = row % nt;
    const int64_t base = ((((int64_t)(rt * ncc + cc) * taps + tap) * kg + g) * 2) * nt + rin;
    out[base] = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    out[base + nt] = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
  }
}

// MD_PACK_WINO jobs, block-cooperative: a block = one (128-row block, 16-channel chunk, 32-row tile) of the fragment layout
// [cout/128][cin/16][tap 9][f 4][row tile 4][plane 2][h 2][row 32][8]: 32 rows x 16 channels x 27 source taps through LDS
// (float4 reads of the contiguous runs), then 64 lanes write the 1 KB of one (tap, f, plane).  Same G and split as
// md_pack_wino_item: bit-identical fragments.
__global__ __launch_bounds__(256) void md_pack_wino_tiled_kernel(const MdPackJob* __restrict__ jobs, int n_jobs) {
  constexpr int ROWS = 32, KC = 16, TAPS = 27, RSTRIDE = KC * TAPS + 4;
  __shared__ __attribute__((aligned(16))) float tile[ROWS * RSTRIDE];
  int lo = 0, hi = n_jobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].block0 <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const MdPackJob J = jobs[lo];
  const int nchunk = J.kdim / KC;
  const int lb = (int)((int64_t)blockIdx.x - J.block0);
  const int rtile = lb & 3, chunk = (lb >> 2) % nchunk, ct = (lb >> 2) / nchunk;
  const int row0 = (ct * 4 + rtile) * ROWS;
  const bool k_inner = J.s_k <= J.s_row;                  // forward: channels of a row contiguous; data gradient: rows of a channel
  const int64_t chunk0 = (int64_t)row0 * J.s_row + (int64_t)chunk * KC * J.s_k;
  const int run = (k_inner ? KC : ROWS) * TAPS, nrun = k_inner ? ROWS : KC;
  const int64_t run_stride = k_inner ? J.s_row : J.s_k;
  const bool vec = (((uintptr_t)(J.w + chunk0) | (uintptr_t)(run_stride * 4)) & 15) == 0 && (k_inner ? J.s_k == TAPS : J.s_row == TAPS);
  if (vec) {
    const int run4 = run >> 2;
    for (int i4 = threadIdx.x; i4 < nrun * run4; i4 += 256) {
      const int ri = i4 / run4, off4 = i4 - ri * run4;
      const f32x4 v = *(const f32x4*)(J.w + chunk0 + (int64_t)ri * run_stride + off4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int o = off4 * 4 + e;
        const int in = o / TAPS, t = o - in * TAPS;
        const int r = k_inner ? ri : in, k = k_inner ? in : ri;
        tile[r * RSTRIDE + k * TAPS + t] = v[e];
      }
    }
  } else {
    for (int idx = threadIdx.x; idx < ROWS * KC * TAPS; idx += 256) {
      const int t = idx % TAPS, q = idx / TAPS;
      const int r = q / KC, k = q % KC;
      tile[r * RSTRIDE + k * TAPS + t] = J.w[(int64_t)(row0 + r) * J.s_row + (int64_t)(chunk * KC + k) * J.s_k + t];
    }
  }
  __syncthreads();
  uint4* out = (uint4*)J.out + ((int64_t)(ct * nchunk + chunk) * 36 * 4 + rtile) * 128;      // + (tap * 4 + f) * 512 + plane * 64 + h * 32 + row
  for (int task = threadIdx.x; task < 36 * 64; task += 256) {
    const int row = task & 31, h = (task >> 5) & 1, tf = task >> 6;
    const int f = tf & 3, tap = tf >> 2;
    const int t0 = tap * 3;
    uint32_t hw[4], lw[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t hh[2], ll[2];
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const float* g = tile + row * RSTRIDE + (h * 8 + 2 * q + e2) * TAPS;
        const float g0 = g[J.flip ? 26 - t0 : t0], g1 = g[J.flip ? 25 - t0 : t0 + 1], g2 = g[J.flip ? 24 - t0 : t0 + 2];
        const float G = f == 0 ? g0 : f == 1 ? (g0 + g1 + g2) * 0.5f : f == 2 ? (g0 - g1 + g2) * 0.5f : g2;
        md_split(G, hh[e2], ll[e2]);
      }
      hw[q] = hh[0] | (hh[1] << 16);
      lw[q] = ll[0] | (ll[1] << 16);
    }
    uint4* o = out + (int64_t)(tap * 4 + f) * 512 + h * 32 + row;
    o[0] = make_uint4(hw[0], hw[1], hw[2], hw[3]);
    o[64] = make_uint4(lw[0], lw[1], lw[2], lw[3]);
  }
}

extern "C" int md_pack_batch(const MdPackJob* jobs_dev, int32_t n_jobs, int64_t total_blocks, int32_t tiled, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffff) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  const dim3 grid((unsigned)total_blocks), blk(256);
  hipStream_t st = (hipStream_t)stream;
  switch (tiled) {     // 1: WPK, runtime taps / kc; taps * 100 + kc: WPK, every job of the table has that geometry; 3: Winograd fragments
    case 0: hipLaunchKernelGGL(md_pack_batch_kernel, grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 1: hipLaunchKernelGGL((md_pack_tiled_kernel<0, 0>), grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 3: hipLaunchKernelGGL(md_pack_wino_tiled_kernel, grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 2732: hipLaunchKernelGGL((md_pack_tiled_kernel<27, 32>), grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 2716: hipLaunchKernelGGL((md_pack_tiled_kernel<27, 16>), grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 932: hipLaunchKernelGGL((md_pack_tiled_kernel<9, 32>), grid, blk, 0, st, jobs_dev, n_jobs); break;
    case 916: hipLaunchKernelGGL((md_pack_tiled_kernel<9, 16>), grid, blk, 0, st, jobs_dev, n_jobs); break;
    default: return MD_ERR_UNSUPPORTED;
  }
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
