"""CPU model of the weight-gradient arithmetic (round 6 study; no GPU): how much do operand roundings cost in dW = sum_positions dy * x?

Runs the oracle U-Net (oracle/unet_oracle.py, torch autograd, float64 accumulation) on the small config with the seeded weights, captures the
(input, output-gradient) pair of every 3x3x3 stride-1 conv of a DDPM loss backward, and recomputes each layer's weight gradient from ROUNDED
operands -- the rounding error of a product is random, so in the sum over positions it averages out; how far depends on how much the products
cancel, which only real (activation, gradient) pairs show.  Variants:
    bf16x3   both operands as bf16 hi + lo, three products (what md_wgrad_wino computes)
    f16x2    x as ONE fp16, dy as fp16 hi + lo (lifted: max |dy| 2^k in [16, 32))        2 matrix-core products instead of 3
    f16x1    both as ONE fp16 (dy lifted)                                                 1 product
    bf16x1   both as one bf16 (for scale)

    python tools/wgrad_numerics.py [--size 32] [--batch 2] [--weights sensitised|trained]
"""
import argparse
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import synth  # noqa: E402
from oracle import unet_oracle as uo  # noqa: E402


def bf16(x):
    return x.float().to(torch.bfloat16).double()


def f16(x):
    return x.float().to(torch.float16).double()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=32)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--weights", default="sensitised", choices=["sensitised", "trained"])
    a = ap.parse_args()
    torch.manual_seed(0)
    cfg = synth.small_config(image_size=a.size)
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg.device = torch.device("cpu")
    model = mutils.get_model(cfg.model.name)(cfg)
    make = synth.trained_like_state_dict if a.weights == "trained" else synth.sensitised_state_dict
    sd = make(model.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(a.size))
    sd = {k: v.float().requires_grad_(True) if v.dtype.is_floating_point else v for k, v in sd.items()}
    pairs = []
    real = F.conv3d

    def spy(x, w, b=None, stride=1, padding=0, *rest, **kw):
        y = real(x, w, b, stride, padding, *rest, **kw)
        if tuple(w.shape[2:]) == (3, 3, 3) and stride in (1, (1, 1, 1)) and w.shape[1] >= 16:
            rec = {"x": x.detach().double(), "w": tuple(w.shape)}
            y.register_hook(lambda g, rec=rec: rec.__setitem__("dy", g.detach().double()))
            pairs.append(rec)
        return y

    F.conv3d = spy
    uo.F.conv3d = spy if hasattr(uo, "F") else real
    try:
        R, B = a.size, a.batch
        mask = synth.synthetic_grid_mask(R).view(1, 1, R, R, R).float()
        x0 = synth.synthetic_inputs(B, 4, R, seed=5).float() * mask
        noise = torch.randn(x0.shape)
        t = torch.tensor([300.0, 700.0][:B] + [500.0] * max(0, B - 2))
        a_bar = torch.exp(-0.25 * (t / 999) ** 2 * 19.9 - 0.5 * (t / 999) * 0.1).view(-1, 1, 1, 1, 1)
        xt = (a_bar * x0 + (1 - a_bar ** 2).sqrt() * noise) * mask
        eps = uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), xt, t)
        loss = (0.5 * ((eps - noise) * mask) ** 2).sum() / B
        loss.backward()
    finally:
        F.conv3d = real
        if hasattr(uo, "F"):
            uo.F.conv3d = real
    print(f"{len(pairs)} convs captured; loss {float(loss):.4f}")
    tot = {k: [0.0, 0.0] for k in ("bf16x3", "f16x2", "f16x1", "bf16x1")}
    worst = {k: 0.0 for k in tot}
    for i, rec in enumerate(pairs):
        x, dy = rec["x"], rec.get("dy")
        if dy is None:
            continue
        co, ci = rec["w"][0], rec["w"][1]

        def wg(xx, gg):
            return torch.nn.grad.conv3d_weight(xx, rec["w"], gg, padding=1)

        ref = wg(x, dy)
        k = 4 - math.floor(math.log2(float(dy.abs().max()))) if float(dy.abs().max()) > 0 else 0
        L = 2.0 ** k
        xh, xl = bf16(x), None
        xl = bf16(x - xh)
        gh = bf16(dy); gl = bf16(dy - gh)
        v = {"bf16x3": wg(xh, gh) + wg(xh, gl) + wg(xl, gh),
             "bf16x1": wg(xh, gh)}
        x16 = f16(x)
        g16h = f16(dy * L); g16l = f16(dy * L - g16h)
        v["f16x2"] = (wg(x16, g16h) + wg(x16, g16l)) / L
        v["f16x1"] = wg(x16, g16h) / L
        n = float(ref.norm())
        canc = float((wg(x.abs(), dy.abs())).norm()) / max(n, 1e-300)      # sum |p| over |sum p|: how much the products cancel
        line = f"conv {i:2d} {ci:4d}->{co:4d} @{x.shape[-1]:3d}^3  |dy|max {float(dy.abs().max()):.2e}  sum|p|/|sum p| {canc:8.1f} "
        for name, val in v.items():
            e = float((val - ref).norm()) / max(n, 1e-300)
            tot[name][0] += float((val - ref).norm()) ** 2
            tot[name][1] += n ** 2
            worst[name] = max(worst[name], e)
            line += f" {name} {e:.2e}"
        print(line)
    print("all conv weight gradients together (rel-L2) / worst layer:")
    for name in tot:
        print(f"  {name:7s} {math.sqrt(tot[name][0] / tot[name][1]):.2e} / {worst[name]:.2e}")


if __name__ == "__main__":
    main()
