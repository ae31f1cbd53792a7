"""Microbenchmark of the Winograd operand pass (csrc/wino_prep2.hip) in its four forms -- bf16 hi / lo (training forward), dual (T + U +
channel sums: the backward's pass over an output gradient), f16f8 and f16f6 (inference, with the layer's equaliser) -- on the hot
shapes.  HIP events on the launch stream, median of --reps; algorithmic bytes = 4 B read + 8 B (dual: 16 B) written per element.
A/B of a kernel change on one box: build the other variant as a second library and alternate the two
    MD_LIB_SUFFIX=_p2old MD_EXTRA_DEFINES=-DP2_OLD_IMAGE python -m meshdiffusion_amd.build
    for i in 1 2 3; do python tools/bench_prep.py; MD_LIB=meshdiffusion_amd/libmeshdiffusion_hip_p2old.so python tools/bench_prep.py; done
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
    ap.add_argument("--shapes", default="128:64:8,256:64:8,256:32:8")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {"lib": os.environ.get("MD_LIB", "default")}
    for sh in a.shapes.split(","):
        cin, S, B = [int(v) for v in sh.split(":")]
        g = torch.Generator().manual_seed(1)
        x = ops.ncdhw_to_f32b(torch.randn((B, cin, S, S, S), generator=g).to(dev))
        ac = torch.stack([1.0 + 0.1 * torch.randn((B, cin), generator=g), 0.1 * torch.randn((B, cin), generator=g)], -1).contiguous().to(dev)
        eq = torch.exp2(torch.randint(-2, 3, (cin,), generator=g).float()).to(dev)
        sums = torch.zeros((B, cin), dtype=torch.float32, device=dev)
        n = B * cin * S ** 3

        def timed(fn):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(a.reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            return sorted(ts)[len(ts) // 2]

        forms = {"bf16": (lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S), 12.0),
                 "dual": (lambda: ops.wino_prep([(x, cin)], None, False, False, B, S, dual=True, sums=sums), 20.0),
                 "f8": (lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S, f8="f8", eq=eq), 12.0),
                 "f6": (lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S, f8="f6", eq=eq), 12.0),
                 "f6_no_eq": (lambda: ops.wino_prep([(x, cin)], ac, True, False, B, S, f8="f6"), 12.0)}
        for name, (fn, bpe) in forms.items():
            ms = timed(fn)
            out[f"{cin}@{S}^3xB{B}/{name}"] = {"ms": round(ms, 4), "GBps": round(bpe * n / ms / 1e6, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
