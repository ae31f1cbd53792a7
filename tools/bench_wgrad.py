"""Micro-benchmark of md_wgrad (csrc/wgrad.hip) on the dominant res64 shapes, with ablations.
    python tools/bench_wgrad.py [--debug 0|1|2|3] [--blocks 256]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import _lib, hip_ops as ops  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import backward as bw  # noqa: E402

SHAPES = [(128, 128, 64, 27), (128, 256, 64, 27), (128, 128, 32, 27), (256, 256, 16, 27), (512, 512, 8, 27), (512, 512, 4, 27),
          (128, 256, 64, 1), (8, 128, 64, 27), (128, 16, 64, 27), (512, 1024, 4, 27), (512, 1024, 8, 27), (256, 768, 16, 27)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--debug", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=256)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--only", type=int, default=-1, help="index into SHAPES")
    a = ap.parse_args()
    lib = _lib.load()
    if not hasattr(lib, "md_wgrad_set_debug"):
        raise SystemExit("--debug needs an ablation build: MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build --force")
    lib.md_wgrad_set_debug(a.debug)
    bw.WGRAD_BLOCKS = a.blocks
    B = 8
    dev = torch.device("cuda")
    for co, ci, S, taps in (SHAPES if a.only < 0 else [SHAPES[a.only]]):
        P = S ** 3
        dy = torch.randn((B, co // 8, P, 8), device=dev)
        act = torch.randn((B, ci // 8, P, 8), device=dev)
        dy_pb = bw.to_pb16(dy, B, co, S, 0, zhalo=False)
        act_pb = bw.to_pb16(act, B, ci, S, 0)
        dw = torch.zeros((co, ci, taps), device=dev)
        bw.wgrad(dy_pb, act_pb, B, co, ci, S, taps, dw, ci * taps, taps, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            bw.wgrad(dy_pb, act_pb, B, co, ci, S, taps, dw, ci * taps, taps, 1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        fl = 2.0 * co * ci * taps * P * B
        print(f"co={co:4d} ci={ci:4d} S={S:3d} taps={taps:2d}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TF/s algorithmic "
              f"({3 * fl / ms / 1e9 / 2500 * 100:5.1f}% of bf16 peak issued)", flush=True)


if __name__ == "__main__":
    main()
