"""Within-process A/B micro-benchmark of md_gemm_conv 3x3x3 configurations on the dominant
res64 shapes (HIP-event timed, interleaved rounds) + bit-equality check against the baseline cfg.

    python tools/bench_conv.py [--cfgs C3_128,C3_128_V2,...] [--rounds 5]

The timing-only ablations of the dedicated kernel (cfgs F1/F3/F4/F6/F7/F8 = 111..118) are compiled only with
`MD_BUILD_ABLATIONS=1 python -m meshdiffusion_amd.build --force` (they add ~4 minutes of compile time).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops as ops  # noqa: E402

SHAPES = [  # (B, Cin, Cout, S) -- SURVEY 8(a) row 2 dominant shapes at the bench batch
    (8, 128, 128, 64), (8, 256, 128, 64), (8, 128, 128, 32), (8, 256, 256, 16), (8, 512, 512, 8),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfgs", default="C3_128,C3_128_SW,C3_128_PIPE,C3_128_V2")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--shapes", default="0,1,2,3,4")
    ap.add_argument("--stagger", type=int, default=0, help="A.stagger (shader cycles) for the *_STAG variants")
    ap.add_argument("--no-residual", action="store_true", help="launch without the residual operand (Conv_0 shape of work)")
    a = ap.parse_args()
    cfgs = [(n, getattr(ops, "CFG_" + n.replace("_STAG", ""))) for n in a.cfgs.split(",")]   # NAME_STAG = NAME with --stagger
    print(f"# cfgs {a.cfgs} stagger {a.stagger} residual {not a.no_residual}")
    dev = "cuda"
    for si in [int(i) for i in a.shapes.split(",")]:
        B, cin, cout, S = SHAPES[si]
        P = S ** 3
        g = torch.Generator(device=dev).manual_seed(si)
        x = torch.randn((B, cin // 8, P, 8), device=dev, generator=g)
        act = ops.gn_apply([(x, cin)], None, B, P, norm=False, silu=False)
        w = torch.randn((cout, cin, 3, 3, 3), device=dev, generator=g) * 0.05
        bias = torch.randn((B, cout), device=dev, generator=g)
        res = torch.randn((B, cout // 8, P, 8), device=dev, generator=g)
        flops = 2.0 * B * P * cin * cout * 27
        outs, times = {}, {n: [] for n, _ in cfgs}
        pws = {n: ops.PackedWeight(w, "conv", c, dev) for n, c in cfgs}

        def run(n, c):
            out = ops.f32b_empty(B, cout, P, dev)
            ops.gemm_conv(cfg=c, a=pws[n].data, b=act, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                          dims=(S, S, S), bias=bias, bias_bstride=cout, residual=None if a.no_residual else res,
                          res_bstride=cout * P, stagger=a.stagger if "STAG" in n else 0)
            return out

        for n, c in cfgs:
            outs[n] = run(n, c)
        torch.cuda.synchronize()
        base = outs[cfgs[0][0]]
        for n, _ in cfgs[1:]:
            same = torch.equal(outs[n], base)
            err = float((outs[n].double() - base.double()).norm() / base.double().norm())
            print(f"  shape {SHAPES[si]} {n} vs {cfgs[0][0]}: bit-equal={same} rel={err:.2e}")
        del outs
        for r in range(a.rounds):
            for n, c in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(n, c); e1.record(); torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1))
        for n, _ in cfgs:
            t = sorted(times[n])
            med = t[len(t) // 2]
            print(f"shape B{B} {cin}->{cout} @{S}^3  {n:14s} min {t[0]:8.3f} ms  med {med:8.3f} ms  "
                  f"{flops / t[0] / 1e9:8.1f} TF/s(alg)  {3 * flops / t[0] / 1e9 / 2500 * 100:5.1f}% of bf16 MFMA peak (issued)")


if __name__ == "__main__":
    main()
