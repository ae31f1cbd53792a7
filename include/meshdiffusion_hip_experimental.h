/*
 * meshdiffusion_hip_experimental.h -- entry points of EXPERIMENTAL kernels that are NOT part of the default
 * libmeshdiffusion_hip.so (built only with MD_BUILD_EXPERIMENTAL=1 python -m meshdiffusion_amd.build; sources under
 * meshdiffusion_amd/csrc/experimental/).  Same conventions as meshdiffusion_hip.h.  Nothing on the product path calls them.
 */
#ifndef MESHDIFFUSION_HIP_EXPERIMENTAL_H
#define MESHDIFFUSION_HIP_EXPERIMENTAL_H
#include "meshdiffusion_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/*
 * EXPERIMENTAL (not used by default; tools/bench_wino.py --f43): the same convolution through Winograd F(4,3) along w
 * (csrc/experimental/conv3_wino43.hip): 6 products per 4 outputs = 1/2 of the direct MFMA work, T = 1.5x the input
 * (T[B][C/8][6][2][D][H][W/4][8 bf16]), weight tiles [Cout/128][Cin/16][kd*3+kh][6][row tile 4][plane 2][k-group 2][row 32][8].
 * Arguments as md_wino_* (no dropout, W % 8 == 0 for the conv); error of one conv ~1.3e-5 (F(2,3): 5.5e-6).
 */
int64_t md_wino43_operand_bytes(int32_t batch, int32_t cin, int32_t D, int32_t H, int32_t W);
int md_wino43_prep(const float* x1, const float* x2, int32_t c1, int32_t c2, const float* ac, int32_t silu, int32_t ups,
                   void* t_out, int32_t batch, int32_t D, int32_t H, int32_t W, void* stream);
int64_t md_wino43_weight_bytes(int32_t cout, int32_t cin);
int md_wino43_pack_weights(const float* w, void* wpk, int32_t cout, int32_t cin, int64_t s_row, int64_t s_k, int32_t flip,
                           void* stream);
int md_conv3_wino43(const void* t_in, const void* wpk, float* out, const float* bias, int64_t bias_bstride,
                    const float* residual, int64_t res_bstride, double* stats, int32_t batch, int32_t cin, int32_t cout,
                    int32_t D, int32_t H, int32_t W, void* stream);

/*
 * ABLATION BUILDS ONLY (MD_BUILD_ABLATIONS=1; tools/bench_wgrad.py): selects a timing-only instantiation of md_wgrad for the
 * following launches of this process (0 = normal).  A process-wide knob, which is why it is not part of the default library,
 * whose entry points are stateless.  The same builds accept the timing-only md_gemm_conv configuration ids 101..122
 * (tools/bench_conv.py) and the md_conv3_wino `variant` values other than 0 (tools/bench_wino.py).
 */
void md_wgrad_set_debug(int32_t flags);

#ifdef __cplusplus
}
#endif
#endif
