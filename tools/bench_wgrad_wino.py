"""Micro-benchmark of the Winograd weight gradient (csrc/wgrad_wino.hip) against the PB16 kernel (csrc/wgrad.hip + md_to_pb16)
on the res64 training shapes (B = 8).  With a MD_BUILD_ABLATIONS=1 build, --dbg times the timing-only variants (MD_WW_DBG bits:
1 no global loads / LDS stores, 2 no MFMAs, 4 no fragment reads, 8 no barrier).

    python tools/bench_wgrad_wino.py [--dbg 1,2,4,5,8,13] [--shapes 128:128:64,256:128:64]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops as ops  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import backward as bw  # noqa: E402


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dbg", default="")
    ap.add_argument("--shapes", default="128:128:64,256:128:64,128:128:32,256:256:32,384:128:32")
    a = ap.parse_args()
    B = a.batch
    rows = []
    for sh in a.shapes.split(","):
        ci, co, S = [int(v) for v in sh.split(":")]
        P = S ** 3
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((B, ci // 8, P, 8), device="cuda", generator=g)
        dy = torch.randn((B, co // 8, P, 8), device="cuda", generator=g) * 0.1
        t_act = ops.wino_prep([(x, ci)], None, False, False, B, S, keep=True)
        _, u_dy = ops.wino_prep([(dy, co)], None, False, False, B, S, dual=True)
        u_dy = u_dy.clone()
        dw = torch.zeros((co, ci, 3, 3, 3), device="cuda")
        flops = 2.0 * B * co * ci * 27 * P
        os.environ.pop("MD_WW_DBG", None)
        ms = timed(lambda: ops.wgrad_wino(u_dy, t_act, B, co, ci, S, dw))
        rows.append(dict(shape=sh, kernel="md_wgrad_wino", ms=round(ms, 3), tflops_alg=round(flops / ms / 1e9, 1),
                         frac_issued=round(2 * flops / ms / 1e9 / 2500e3 * 1e3, 3)))
        for d in [v for v in a.dbg.split(",") if v]:
            os.environ["MD_WW_DBG"] = d
            ms_d = timed(lambda: ops.wgrad_wino(u_dy, t_act, B, co, ci, S, dw))
            rows.append(dict(shape=sh, kernel=f"md_wgrad_wino dbg={d}", ms=round(ms_d, 3)))
        os.environ.pop("MD_WW_DBG", None)
        ms_p = timed(lambda: ops.wino_prep([(dy, co)], None, False, False, B, S, dual=True))
        ms_p1 = timed(lambda: ops.wino_prep([(dy, co)], None, False, False, B, S))
        sm = torch.zeros((B, co), device="cuda")
        ms_ps = timed(lambda: ops.wino_prep([(dy, co)], None, False, False, B, S, dual=True, sums=sm))
        ms_cs = timed(lambda: bw.channel_sums(dy, B, co, P))
        rows.append(dict(shape=sh, kernel="md_wino_prep_dual(dy)", ms=round(ms_p, 3), single_ms=round(ms_p1, 3),
                         with_sums_ms=round(ms_ps, 3), channel_sums_ms=round(ms_cs, 3),
                         tbs=round((4.0 + 16.0) * B * co * P / ms_p / 1e9, 2)))
        # the PB16 path
        xs = bw.split_f32b(x, B, ci, P)
        dy_pb = bw.to_pb16(dy, B, co, S, 0, zhalo=False)
        act_pb = bw.to_pb16(xs, B, ci, S, 1)
        ms_o = timed(lambda: bw.wgrad(dy_pb, act_pb, B, co, ci, S, 27, dw, ci * 27, 27, 1))
        ms_t = timed(lambda: (bw.to_pb16(dy, B, co, S, 0, zhalo=False), bw.to_pb16(xs, B, ci, S, 1)))
        rows.append(dict(shape=sh, kernel="md_wgrad (PB16)", ms=round(ms_o, 3), tflops_alg=round(flops / ms_o / 1e9, 1),
                         to_pb16_ms=round(ms_t, 3)))
        del x, dy, t_act, u_dy, xs, dy_pb, act_pb
        torch.cuda.empty_cache()
    for r in rows:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
