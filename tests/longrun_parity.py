"""Parity checker script (test infrastructure: it imports the oracle; ~4 GPU-minutes, so not part of `pytest -m gpu`).
End-to-end check of the BASELINE target: "sampled SDF+deform grids matching the reference within 1e-3
rel-L2 under fixed seed", over the full ancestral schedule, on the GPU box.

The reference tree is not available on the GPU box, so the comparison partner is the oracle restatement
(pinned to the imported reference at 3e-6 by oracle/gen_golden.py) evaluated with PyTorch fp32 ops on the
same GPU.  Both trajectories consume the SAME noise tensors (drawn once per step from the device
generator), so the difference isolates the arithmetic of the HIP path.

    python tests/longrun_parity.py [--steps 999] [--config res64|small] [--batch 1]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import synth  # noqa: E402
from meshdiffusion_amd.config import get_config_res64  # noqa: E402
from meshdiffusion_amd.lib.diffusion import sampling, sde_lib  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401,E402
from oracle import unet_oracle as uo  # noqa: E402


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=999)
    ap.add_argument("--config", default="res64")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--out", default=None)
    ap.add_argument("--seeds", default="42", help="comma-separated seeds: one full trajectory pair per seed")
    ap.add_argument("--partner", default="oracle", choices=["oracle", "direct"],
                    help="oracle: the fp32 PyTorch restatement on the same GPU (default); direct: the HIP path with the direct 27-tap "
                         "kernels (MD_WINO=0) -- the build whose 999-step parity vs the oracle is on record -- so that a B = 8 run "
                         "of the full schedule costs minutes instead of half an hour of fp32 torch convolutions")
    ap.add_argument("--precision", default=None, help="config.model.hip_precision of the HIP path (default: the config's: f16f6)")
    ap.add_argument("--weights", default="sensitised", choices=["sensitised", "trained_like"],
                    help="synth.sensitised_state_dict (i.i.d.) or the adversarial synth.trained_like_state_dict")
    ap.add_argument("--calibrate", action="store_true", help="models.utils.calibrate_model at the run's batch before sampling (what evaler does "
                                                              "after restore_checkpoint): measured equalisers, Upsample convs on the reduced-precision path")
    ap.add_argument("--oracle-samples", default=None,
                    help="comma-separated sample indices: the HIP path runs the whole batch (the graded B = 8 launches), the fp32 oracle "
                         "only these samples of it on the same noise (samples are independent: GroupNorm is per sample) -- a 999-step "
                         "B = 8 run then costs 2/8 of the 27 GPU-minutes of fp32 torch convolutions")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = get_config_res64() if a.config == "res64" else synth.small_config()
    cfg.device = dev
    if a.precision:
        cfg.model.hip_precision = a.precision
    a.hip_precision = cfg.model.hip_precision
    a.sel = [int(v) for v in a.oracle_samples.split(",")] if a.oracle_samples else None
    R = cfg.data.image_size
    model = mutils.create_model(cfg).eval()
    if a.weights == "trained_like":
        sd = synth.trained_like_state_dict(model.module.state_dict(), seed=4321, grid_mask=synth.synthetic_grid_mask(R))
    else:
        sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    a.calibration = None
    if a.calibrate:
        rep = mutils.calibrate_model(model, cfg, batch=a.batch)
        a.calibration = None if rep is None else dict(measured=rep["measured"], worst=rep["worst"], demoted=[list(d[:2]) for d in rep["demoted"]])
        print("calibrated:", a.calibration, flush=True)
    sd_gpu = {k: v.to(dev) for k, v in sd.items()}
    del sd
    ocfg = synth.oracle_cfg(cfg)
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R).to(dev)
    shape = (a.batch, 4, R, R, R)
    st = sampling.AncestralStepper(sde, shape, device=dev, grid_mask=mask)
    model_fn = mutils.get_model_fn(model, train=False)
    runs = []
    for seed in [int(v) for v in a.seeds.split(",")]:
        runs.append(one_seed(a, seed, st, model_fn, sd_gpu, ocfg, shape, dev, mask))
    out = runs[0] if len(runs) == 1 else {"config": a.config, "batch": a.batch, "steps": a.steps, "target": 1e-3,
                                           "worst_final_x_mean_rel_l2": max(r["final_x_mean_rel_l2"] for r in runs), "runs": runs}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


def one_seed(a, seed, st, model_fn, sd_gpu, ocfg, shape, dev, mask):
    torch.manual_seed(seed)
    x_h = st.prior()
    sel = a.sel if a.sel is not None else list(range(shape[0]))
    x_o = x_h[sel].clone()
    ts = st.timesteps
    marks = sorted(set([1, 10, 50, 100, 200, 400, 600, 800, a.steps]))
    log = []
    t_h = t_o = 0.0
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    with torch.no_grad():
        for i in range(a.steps):
            z = torch.randn(shape, device=dev)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            x_h, xm_h = st.step(model_fn, x_h, i, draw=lambda _t: z)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            if a.partner == "direct":
                from meshdiffusion_amd import hip_ops
                keep, hip_ops.WINO = hip_ops.WINO, False
                try:
                    e = model_fn(x_o, st.labels[i])
                finally:
                    hip_ops.WINO = keep
            else:
                e = torch.cat([uo.unet_res64_forward(sd_gpu, ocfg, x_o[b:b + 1], st.labels[i][sel[b]:sel[b] + 1]) for b in range(len(sel))])
            # per-evaluation error of the HIP U-Net on the PARTNER's state (no trajectory feedback)
            if (i + 1) in marks:
                if a.sel is None:
                    e_h = model_fn(x_o, st.labels[i])
                else:                              # the partner's samples inside a full batch of the HIP state
                    xb = x_h.clone(); xb[sel] = x_o
                    e_h = model_fn(xb, st.labels[i])[sel]
                eval_err = rel(e_h, e)
            x_o, xm_o = uo_step(x_o, e, z[sel], st, i, mask)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            t_h += t1 - t0; t_o += t2 - t1
            if (i + 1) in marks:
                rec = {"step": i + 1, "x_rel_l2": rel(x_h[sel], x_o), "x_mean_rel_l2": rel(xm_h[sel], xm_o),
                       "unet_eval_rel_l2": eval_err}
                log.append(rec)
                print(json.dumps(rec), flush=True)
    from meshdiffusion_amd import hip_ops
    return {"config": a.config, "batch": a.batch, "steps": a.steps, "seed": seed, "partner": a.partner,
            "hip_precision": hip_ops.FORCE_PRECISION or a.hip_precision, "weights": a.weights, "oracle_samples": sel,
            "calibration": a.calibration,
            "final_x_mean_rel_l2": log[-1]["x_mean_rel_l2"],
            "target": 1e-3, "hip_s_per_step": t_h / a.steps, "oracle_gpu_s_per_step": t_o / a.steps, "trace": log}


def uo_step(x, e, z, st, i, mask):
    c = st.coef[i][0]
    beta, sigma = c[0], c[1]
    score = -e / sigma
    x_mean = (x + beta * score) / torch.sqrt(1.0 - beta)
    x_new = x_mean + torch.sqrt(beta) * z
    return x_new * mask, x_mean * mask


if __name__ == "__main__":
    main()
