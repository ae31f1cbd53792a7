"""Microbenchmark of Downsample's stride-2 conv: md_conv3_s2 (fp32 operand, slab-wise K loop) against the path it replaces
(md_gn_apply split pass + md_gemm_conv MD_CFG_C3_S2).  HIP events on the launch stream, median of `--reps`.

    python tools/bench_s2.py [--shapes cin:cout:S_out:B,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops as ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=9)
    ap.add_argument("--shapes", default="128:128:32:8,128:128:16:8,256:256:8:8,128:128:64:2")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for sh in a.shapes.split(","):
        cin, cout, S, B = [int(v) for v in sh.split(":")]
        g = torch.Generator().manual_seed(1)
        x = ops.ncdhw_to_f32b(torch.randn((B, cin, 2 * S, 2 * S, 2 * S), generator=g).to(dev))
        w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.03).to(dev)
        bias = torch.randn((cout,), generator=g).to(dev)
        flops = 2.0 * B * cout * cin * 27 * S ** 3

        def timed(fn):
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        pw = ops.PackedWeight(w, "conv", ops.CFG_S2_PACK, dev)
        stats = torch.zeros((B, cout, 2), dtype=torch.float64, device=dev)
        out = ops.f32b_empty(B, cout, S ** 3, dev)
        ms = timed(lambda: ops.conv3_s2(pw, x, B, S, bias=bias, stats=stats, out=out))
        print(json.dumps(dict(shape=sh, kernel="md_conv3_s2", ms=round(ms, 4), tflops_alg=round(flops / ms / 1e9, 1))), flush=True)
        pw_old = ops.PackedWeight(w, "conv", ops.CFG_C3_S2, dev)
        P_in = (2 * S) ** 3
        ms_split = timed(lambda: ops.gn_apply([(x, cin)], None, B, P_in, norm=False, silu=False))
        s16 = ops.gn_apply([(x, cin)], None, B, P_in, norm=False, silu=False)
        ks = ops.ksplit_for(ops.CFG_C3_S2, B, cout, cin, S)
        ms_old = timed(lambda: ops.gemm_conv(cfg=ops.CFG_C3_S2, a=pw_old.data, b=s16, out=out, batch=B, rows=cout, rows_alloc=cout,
                                             kdim=cin, dims=(S, S, S), bias=bias, ksplit=ks))
        print(json.dumps(dict(shape=sh, kernel="md_gn_apply(split) + md_gemm_conv(C3_S2)", ksplit=ks, ms_split=round(ms_split, 4),
                              ms_conv=round(ms_old, 4), ms=round(ms_split + ms_old, 4))), flush=True)


if __name__ == "__main__":
    main()
