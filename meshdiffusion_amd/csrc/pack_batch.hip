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
  else out[item] = md_pack_wpk_item(J.w, J.rows, J.kdim, J.taps, J.s_row, J.s_k, J.s_tap, J.nt, J.kc, J.prec, item);
}

extern "C" int md_pack_batch(const MdPackJob* jobs_dev, int32_t n_jobs, int64_t total_blocks, void* stream) {
  if (!jobs_dev || n_jobs <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffff) return MD_ERR_BAD_ARG;
  MD_HIP_CLEAR_ERROR();
  hipLaunchKernelGGL(md_pack_batch_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n_jobs);
  MD_HIP_CHECK_LAUNCH();
  return MD_OK;
}
