/*
 * meshdiffusion_hip_experimental.h -- entry points that are NOT part of the default libmeshdiffusion_hip.so: the timing-only
 * knob of MD_BUILD_ABLATIONS=1 builds.  Same conventions as meshdiffusion_hip.h.  Nothing on the product path calls it.
 * (The F(4,3) Winograd prototype that lived here through rounds 2-4 was retired in round 5: DESIGN.md section 8 keeps its
 * measurements, the git history its source.)
 */
#ifndef MESHDIFFUSION_HIP_EXPERIMENTAL_H
#define MESHDIFFUSION_HIP_EXPERIMENTAL_H
#include "meshdiffusion_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

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
