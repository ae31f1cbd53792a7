"""Achieved HBM bandwidth of the streaming kernels at the dominant res64 shape (128 channels, 64^3, batch 8).
    python tools/bench_stream.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops as ops  # noqa: E402


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B, C, S = 8, 128, 64
    P = S ** 3
    dev = torch.device("cuda")
    x = torch.randn((B, C // 8, P, 8), device=dev)
    gn = torch.nn.GroupNorm(32, C, eps=1e-6).to(dev)
    n = B * C * P
    ops.FUSE_GN_STATS = False
    prm = ops.gn_params([(x, C)], gn.weight, gn.bias, B, P)
    out = ops.s16b_empty(B, C, P, dev)
    y = torch.empty_like(x)
    rows = [
        ("md_gn_stats + finalize (read 4 B/elt)", lambda: ops.gn_params([(x, C)], gn.weight, gn.bias, B, P), 4 * n),
        ("md_gn_apply norm+silu (read 4, write 4)", lambda: ops.gn_apply([(x, C)], prm, B, P, out=out), 8 * n),
        ("md_gn_apply + raw split (read 4, write 8)", lambda: ops.gn_apply([(x, C)], prm, B, P, out=out, want_raw=True), 12 * n),
        ("torch copy_ (read 4, write 4)", lambda: y.copy_(x), 8 * n),
    ]
    for name, fn, nbytes in rows:
        ms = timed(fn)
        print(f"{name:45s} {ms:7.3f} ms  {nbytes / ms / 1e9:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
