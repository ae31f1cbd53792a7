"""Short measured runs of the other BASELINE.json configs on one MI355X (bench.py covers configs[1]).

  #4  res128 4-ch grid, batch=2, ancestral sampling steps        -> sample-steps/s, peak HBM
  #5  marching tets, 32 meshes per launch (res64 tet grid)        -> meshes/s
      cond_gen inpainting sampler, res64, batch=32, a few iterations -> ms/iteration, peak HBM
  #1  res64 uncond, batch=1, first 10 steps                        -> ms/step
Prints one JSON object.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import synth  # noqa: E402
from meshdiffusion_amd.config import get_config_res64, get_config_res128  # noqa: E402
from meshdiffusion_amd.dmtet import GridMesher  # noqa: E402
from meshdiffusion_amd.lib.diffusion import sampling, sde_lib  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, ddpm_res128, utils as mutils  # noqa: F401,E402

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def build(cfg, R, seed):
    cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg).eval()
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=seed, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    return model


def time_steps(model, cfg, B, R, steps, warm, cond=False):
    mutils.calibrate_model(model, cfg, batch=min(B, 8))      # as evaler does after loading a checkpoint (untimed)
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = synth.synthetic_grid_mask(R).cuda()
    shape = (B, 4, R, R, R)
    torch.cuda.reset_peak_memory_stats()
    if not cond:
        st = sampling.AncestralStepper(sde, shape, device="cuda", grid_mask=mask.view(1, R, R, R))
        fn = mutils.get_model_fn(model)
        with torch.no_grad():
            x = st.prior()
            for i in range(warm):
                x, _ = st.step(fn, x, i)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(steps):
                x, xm = st.step(fn, x, warm + i)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert bool(torch.isfinite(xm).all())
    else:
        g = torch.Generator().manual_seed(5)
        partial = torch.sign(torch.randn((1, 1, R, R, R), generator=g)).cuda()
        pmask = ((torch.rand((1, 1, R, R, R), generator=g) < 0.5).float().cuda()) * mask.view(1, 1, R, R, R)
        pmask[..., R // 2:] = 0          # half-space x < R/2
        fn = sampling.get_sampling_fn(cfg, sde, shape, lambda t: t, 1e-3, grid_mask=mask.view(1, 1, R, R, R))
        fn(model, partial=partial, partial_mask=pmask, freeze_iters=950, n_iters=warm)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _ = fn(model, partial=partial, partial_mask=pmask, freeze_iters=950, n_iters=steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        assert bool(torch.isfinite(out).all())
    return dt / steps, torch.cuda.max_memory_allocated() / 2 ** 30


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-res128", action="store_true")
    ap.add_argument("--only-res128", action="store_true", help="config #4 only (profiling)")
    a = ap.parse_args()
    res = {}
    if a.only_res128:
        cfg = get_config_res128()
        model = build(cfg, 128, 99)
        s, peak = time_steps(model, cfg, 2, 128, 2, 1)
        print(json.dumps({"config4_res128_b2": {"ms_per_step": round(s * 1e3, 1), "peak_hbm_gib": round(peak, 1)}}))
        return
    cfg = get_config_res64()
    model = build(cfg, 64, 1234)
    s, _ = time_steps(model, cfg, 1, 64, 10, 2)
    res["config1_res64_b1_10steps"] = {"ms_per_step": round(s * 1e3, 2), "sample_steps_per_s": round(1 / s, 2)}
    s, peak = time_steps(model, cfg, 32, 64, 3, 1, cond=True)
    res["config5_cond_gen_res64_b32"] = {"ms_per_iteration": round(s * 1e3, 1), "sample_steps_per_s": round(32 / s, 2),
                                         "peak_hbm_gib": round(peak, 1)}
    del model
    torch.cuda.empty_cache()
    # marching tets, 32 meshes per launch
    tet = np.load(os.path.join(GOLD, "64_tets_cropped.npz"))
    mesher = GridMesher(tet["vertices"], tet["indices"], 64)
    g = torch.Generator().manual_seed(11)
    ax = torch.linspace(-1, 1, 64)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    grids = torch.empty(32, 4, 64, 64, 64)
    for m in range(32):
        grids[m, 0] = 0.3 + 0.01 * m - (X ** 2 + Y ** 2 + Z ** 2).sqrt() + 0.05 * torch.sin(9 * X + m)
        grids[m, 1:] = torch.randn(3, 64, 64, 64, generator=g) * 0.7
    gd = grids.cuda()
    meshes = mesher(gd)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        meshes = mesher(gd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    res["config5_marching_tets_b32"] = {"ms_per_call_32_meshes": round(dt * 1e3, 2), "meshes_per_s": round(32 / dt, 1),
                                        "verts_faces_mesh0": [int(meshes[0][0].shape[0]), int(meshes[0][1].shape[0])]}
    if not a.skip_res128:
        cfg = get_config_res128()
        model = build(cfg, 128, 99)
        s, peak = time_steps(model, cfg, 2, 128, 2, 1)
        res["config4_res128_b2"] = {"ms_per_step": round(s * 1e3, 1), "sample_steps_per_s": round(2 / s, 3),
                                    "peak_hbm_gib": round(peak, 1), "mask": "synthetic period-4 lattice (asset missing upstream)"}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
