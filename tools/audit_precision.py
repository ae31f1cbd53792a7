"""Per-layer audit of the reduced-precision conv arithmetic on a GIVEN set of weights (a checkpoint or the synthetic ones).

The f16f8 / f16f6 arithmetic of the inference Winograd convs is validated on synthetic weights only -- the trained
checkpoints are external downloads.  This tool is what to run once on a real checkpoint: it evaluates the model at a few timesteps with
`hip_ops.AUDIT` on, which repeats every reduced-precision conv launch in bf16x3 on the same operands and records the relative
difference, prints the per-layer table, and lists the convs whose error exceeds --bar; `--apply` marks those convs
(`layer.md_bf16x3_sites`) so that they run bf16x3 from then on, and reports the whole-network difference before / after.

    python tools/audit_precision.py [--ckpt checkpoint.pth] [--weights sensitised|trained_like] [--precision f16f6] [--bar 6e-5]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import hip_ops, synth  # noqa: E402
from meshdiffusion_amd.config import get_config_res64, get_config_res128  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, ddpm_res128, utils as mutils  # noqa: F401,E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="res64", choices=["res64", "res128"])
    ap.add_argument("--ckpt", default=None, help="a reference-format checkpoint ({'model': state_dict, 'ema': ...}); default: synthetic weights")
    ap.add_argument("--weights", default="trained_like", choices=["sensitised", "trained_like"])
    ap.add_argument("--precision", default=None, help="default: the config's hip_precision")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--timesteps", default="999,500.3,37.8")
    ap.add_argument("--bar", type=float, default=6e-5, help="per-layer relative difference above which a conv is listed (and, with --apply, moved to bf16x3)")
    ap.add_argument("--apply", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = get_config_res64() if a.config == "res64" else get_config_res128()
    cfg.device = dev
    if a.precision:
        cfg.model.hip_precision = a.precision
    R = cfg.data.image_size
    model = mutils.create_model(cfg).eval()
    if a.ckpt:
        # what sampling uses (evaler._setup): the checkpoint's raw weights loaded STRICTLY (a key-prefix mismatch must not leave the
        # random initialisation in place unnoticed), then the EMA shadow parameters copied over them
        from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
        ck = torch.load(a.ckpt, map_location="cpu")
        sd = ck.get("model", ck)
        missing, unexpected = model.load_state_dict(sd, strict=False)
        if missing or unexpected:
            raise SystemExit(f"--ckpt: state-dict keys do not match the model: missing {list(missing)[:5]} ({len(missing)}), "
                             f"unexpected {list(unexpected)[:5]} ({len(unexpected)})")
        if "ema" in ck:
            ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
            ema.load_state_dict(ck["ema"])
            ema.copy_to(model.parameters())
            hip_ops.bump_param_epoch()
            print("audit_precision: EMA shadow parameters applied (the weights sampling uses)")
        else:
            print("audit_precision: the checkpoint has no 'ema' entry -- auditing its raw weights")
    else:
        make = synth.trained_like_state_dict if a.weights == "trained_like" else synth.sensitised_state_dict
        model.module.load_state_dict(make(model.module.state_dict(), grid_mask=synth.synthetic_grid_mask(R)), strict=True)
    names = {id(m): n for n, m in model.module.named_modules()}
    x = (synth.synthetic_inputs(a.batch, 4, R, seed=5) * synth.synthetic_grid_mask(R).view(1, 1, R, R, R)).to(dev)
    worst = {}

    def evaluate(t):
        with torch.no_grad():
            return model(x, torch.full((a.batch,), float(t), device=dev))

    def whole(t):
        keep = model.module.hip_precision
        y = evaluate(t)
        model.module.hip_precision = "bf16x3"
        try:
            y3 = evaluate(t)
        finally:
            model.module.hip_precision = keep
        return float(((y.double() - y3.double()).norm() / y3.double().norm()).item())

    for t in [float(v) for v in a.timesteps.split(",")]:
        hip_ops.AUDIT = []
        try:
            evaluate(t)
            recs = hip_ops.AUDIT
        finally:
            hip_ops.AUDIT = None
        for r in recs:
            key = (names.get(id(r["owner"]), "?"), r["site"])
            if key not in worst or r["rel_l2"] > worst[key]["rel_l2"]:
                worst[key] = dict(r, t=t)
        print(f"t = {t}: {len(recs)} reduced-precision conv launches, worst layer {max(r['rel_l2'] for r in recs):.2e}, "
              f"whole network vs bf16x3 {whole(t):.2e}", flush=True)
    print(f"\nper conv, worst over the timesteps ({model.module.hip_precision} against bf16x3 on the same operands):")
    flagged = []
    for (name, site), r in sorted(worst.items(), key=lambda kv: -kv[1]["rel_l2"]):
        mark = " <-- above the bar" if r["rel_l2"] > a.bar else ""
        print(f"  {name:22s} {site:3s} {r['cin']:5d}->{r['cout']:4d} @{r['S']:3d}^3  {r['fmt']}  {r['rel_l2']:.2e} (t = {r['t']}){mark}")
        if r["rel_l2"] > a.bar:
            flagged.append(r)
    print(f"\n{len(flagged)} of {len(worst)} convs above {a.bar:g}")
    if a.apply and flagged:
        for r in flagged:
            r["owner"].md_bf16x3_sites = tuple(set(getattr(r["owner"], "md_bf16x3_sites", ())) | {r["site"]})
        for t in [float(v) for v in a.timesteps.split(",")]:
            print(f"after moving them to bf16x3: t = {t}: whole network vs bf16x3 {whole(t):.2e}")


if __name__ == "__main__":
    main()
