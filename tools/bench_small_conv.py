"""Micro-benchmark of the direct 3x3x3 kernel on the small grids (8^3, 16^3) where it runs with split-K: production vs the timing-only
ablations (library built with MD_BUILD_ABLATIONS=1: F4 = no weight global loads, F3 = no barriers / weight commits, F1 = no LDS
fragment reads).  HIP events, median of --reps.

    MD_LIB_SUFFIX=_abl MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build && MD_LIB=.../libmeshdiffusion_hip_abl.so python tools/bench_small_conv.py
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
    ap.add_argument("--reps", type=int, default=15)
    ap.add_argument("--shapes", default="512:512:8:8,512:512:8:1,1024:512:8:8,256:256:16:1,512:512:4:8,512:512:4:1")
    ap.add_argument("--cfgs", default="C3_128_FAST,F4,F3,F1")
    a = ap.parse_args()
    dev = "cuda"
    for sh in a.shapes.split(","):
        cin, cout, S, B = [int(v) for v in sh.split(":")]
        P = S ** 3
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((B, cin // 8, P, 8), device=dev, generator=g)
        act = ops.gn_apply([(x, cin)], None, B, P, norm=False, silu=False)
        w = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=g) * 0.05
        base_cfg = ops.conv_cfg_for(S)
        ks = ops.ksplit_for(base_cfg, B, cout, cin, S)
        for name in a.cfgs.split(","):
            cfg = getattr(ops, "CFG_" + name) if name != "C3_128_FAST" else base_cfg
            if S % 8 and name != "C3_128_FAST":
                continue
            try:
                pw = ops.PackedWeight(w, "conv", cfg, dev)
                out = ops.f32b_empty(B, cout, P, dev)

                def run():
                    ops.gemm_conv(cfg=cfg, a=pw.data, b=act, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin, dims=(S, S, S), ksplit=ks)
                run(); torch.cuda.synchronize()
                ts = []
                for _ in range(a.reps):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); run(); e1.record(); torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1))
                ms = sorted(ts)[len(ts) // 2]
            except Exception as e:  # configuration not built
                print(f"{sh} {name}: {e}")
                continue
            wbytes = cout * cin * 27 * 4
            print(json.dumps(dict(shape=sh, cfg=name, ksplit=ks, ms=round(ms, 4), weight_gbs=round(wbytes / ms / 1e6, 1),
                                  tflops=round(2.0 * B * P * cin * cout * 27 / ms / 1e9, 1))), flush=True)


if __name__ == "__main__":
    main()
