/* A plain C host of libmeshdiffusion_hip.so: proves that include/meshdiffusion_hip.h is valid C99 and that the library
 * links and answers without Python or torch (argument validation only -- no device work, so it runs without a GPU). */
#include <stdio.h>
#include <stddef.h>
#include "meshdiffusion_hip.h"

int main(void) {
  MdGemmConvArgs a;
  int fails = 0;
  if (md_abi_version() != MD_ABI_VERSION) { printf("abi %d != %d\n", md_abi_version(), MD_ABI_VERSION); ++fails; }
  if (md_packed_weight_bytes(128, 128, 27, 128, 32) != (int64_t)27 * 4 * 4 * 2 * 128 * 16) { printf("packed bytes\n"); ++fails; }
  if (md_packed_weight_bytes(0, 128, 27, 128, 32) != MD_ERR_BAD_ARG) { printf("bad arg not rejected\n"); ++fails; }
  if (md_wgrad_workspace_bytes(128, 128, 27, 28) != (int64_t)28 * 27 * 128 * 128 * 4) { printf("wgrad ws\n"); ++fails; }
  if (md_pb16_bytes(8, 128, 64, 64, 64, 0, 1) != (int64_t)66 * 66 * 66 * 2 * 128 * 8 * 2) { printf("pb16 bytes\n"); ++fails; }
  if (md_gemm_conv(NULL, NULL) != MD_ERR_BAD_ARG) { printf("null args\n"); ++fails; }
  if (md_gn_stats(NULL, NULL, 1, 8, 1, 8, 0, NULL) != MD_ERR_BAD_ARG) { printf("gn_stats null\n"); ++fails; }
  /* round-3 entry points: argument validation and the job record of md_pack_batch */
  {
    MdPackJob j;
    char buf[64];
    if (sizeof j != 88 || offsetof(MdPackJob, block0) != 48 || offsetof(MdPackJob, kind) != 84) { printf("MdPackJob layout\n"); ++fails; }
    if (md_pack_batch(NULL, 1, 1, 0, NULL) != MD_ERR_BAD_ARG) { printf("pack_batch null\n"); ++fails; }
    if (md_conv3_s2((const float*)buf, buf, (float*)buf, NULL, 0, NULL, 1, 48, 64, 64, 8, 8, 8, NULL) != MD_ERR_UNSUPPORTED) { printf("s2 cin\n"); ++fails; }
    if (md_conv3_head((const float*)buf, NULL, buf, (float*)buf, 1, 64, 16, 8, 8, 8, NULL) != MD_ERR_BAD_ARG) { printf("head ac\n"); ++fails; }
    if (md_conv3_stem(buf, buf, (float*)buf, NULL, NULL, NULL, 1, 12, 8, 8, 8, NULL) != MD_ERR_UNSUPPORTED) { printf("stem cout\n"); ++fails; }
  }
  /* round-4 entry points: the f16f8 / f16f6 Winograd calls and the chunked mesher's workspace */
  {
    char buf[64];
    if (md_wino_prep_f8(NULL, NULL, 16, 0, NULL, 0, 0, NULL, buf, 1, 8, 8, 8, NULL) != MD_ERR_BAD_ARG) { printf("prep_f8 null\n"); ++fails; }
    if (md_wino_prep_f6((const float*)buf, NULL, 24, 0, NULL, 0, 0, NULL, buf, 1, 8, 8, 8, NULL) != MD_ERR_BAD_ARG) { printf("prep_f6 blocks\n"); ++fails; }
    if (md_wino_weight_bytes_f8(128, 64) != (int64_t)128 * 64 * 36 * 4 + 256) { printf("f8 weight bytes\n"); ++fails; }
    if (md_wino_pack_weights_f6((const float*)buf, NULL, buf, 96, 64, 64 * 27, 27, NULL) != MD_ERR_BAD_ARG) { printf("pack_f6 rows\n"); ++fails; }
    if (md_conv3_wino_f6(buf, buf, (float*)buf, NULL, 0, NULL, 0, NULL, 1, 48, 128, 8, 8, 8, NULL) != MD_ERR_UNSUPPORTED) { printf("wino_f6 cin\n"); ++fails; }
    if (md_conv3_wino_f8(NULL, buf, (float*)buf, NULL, 0, NULL, 0, NULL, 1, 32, 128, 8, 8, 8, NULL) != MD_ERR_BAD_ARG) { printf("wino_f8 null\n"); ++fails; }
    if (md_wino_equaliser(NULL, (const float*)buf, (const float*)buf, 128, 64, 64 * 27, 27, (float*)buf, NULL) != MD_ERR_BAD_ARG) { printf("equaliser null\n"); ++fails; }
    if (md_marching_tets_workspace_bytes(32, 195331, 159330) != (int64_t)32 * (195331 + 191 + 2 * 156) * 4) { printf("mt workspace\n"); ++fails; }
  }
  printf("sizeof(MdGemmConvArgs)=%zu stats@%zu %s\n", sizeof a, offsetof(MdGemmConvArgs, stats), fails ? "FAIL" : "ok");
  return fails;
}
