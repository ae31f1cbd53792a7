"""Generation drivers -- mirror of the reference's lib/diffusion/evaler.py
(`uncond_gen` :14-60, `cond_gen` :134-211).  Same behaviour and file outputs
(`{eval_dir}/{idx}.npy`, float32 [B,4,R,R,R]); the grid mask is read from
`./data/grid_mask_{R}.pt` relative to the cwd like the reference, with `map_location` so that the
CUDA-saved tensor loads on any host.
"""
import os

import numpy as np
import torch

from . import losses, sampling, sde_lib
from .models import ddpm_res64, ddpm_res128  # noqa: F401  (registers the models, like trainer.py:7 in the reference)
from .models import utils as mutils
from .models.ema import ExponentialMovingAverage
from .utils import restore_checkpoint


def _setup(config, mask_shape):
    eval_dir, ckpt_path = config.eval.eval_dir, config.eval.ckpt_path
    os.makedirs(eval_dir, exist_ok=True)
    score_model = mutils.create_model(config)
    optimizer = losses.get_optimizer(config, score_model.parameters())
    ema = ExponentialMovingAverage(score_model.parameters(), decay=config.model.ema_rate)
    state = dict(optimizer=optimizer, model=score_model, ema=ema, step=0)
    sde = sde_lib.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max,
                        N=config.model.num_scales, device=config.device)
    R = config.data.image_size
    mask_path = getattr(config.eval, "grid_mask_path", None) or f"./data/grid_mask_{R}.pt"
    grid_mask = torch.load(mask_path, map_location=config.device).view(*mask_shape(R)).to(config.device)
    shape = (config.eval.batch_size, config.data.num_channels, R, R, R)
    sampling_fn = sampling.get_sampling_fn(config, sde, shape, lambda x: x, 1e-3, grid_mask=grid_mask)
    assert os.path.exists(ckpt_path), ckpt_path
    print("ckpt path:", ckpt_path)
    state = restore_checkpoint(ckpt_path, state, device=config.device)
    ema.copy_to(score_model.parameters())
    print(f"loaded model is trained till iter {state['step'] // config.training.iter_size}")
    return score_model, sampling_fn, eval_dir


def uncond_gen(config, idx=0):
    """Unconditional generation: N-1 ancestral steps from the masked prior, saves {idx}.npy."""
    with torch.no_grad():
        model, sampling_fn, eval_dir = _setup(config, lambda R: (1, R, R, R))
        samples, _ = sampling_fn(model)
        np.save(os.path.join(eval_dir, f"{idx}.npy"), samples.cpu().numpy())


def tet_vertices_to_grid_index(vertices):
    """Map tet-grid vertex positions to integer cubic-grid coordinates (evaler.py:187-195)."""
    uniq = vertices[:].unique()
    dx = uniq[1] - uniq[0]
    return torch.round((vertices - vertices.min()) / dx).long()


def cond_gen(config, save_fname="0"):
    """Conditional generation from a partial DMTet (2.5D view) scattered into the cubic grid."""
    with torch.no_grad():
        model, sampling_fn, eval_dir = _setup(config, lambda R: (1, 1, R, R, R))
        R = config.data.image_size
        partial = torch.load(config.eval.partial_dmtet_path, map_location="cpu", weights_only=False)
        tet = np.load(config.eval.tet_path)
        idx = tet_vertices_to_grid_index(torch.tensor(tet["vertices"]))
        sdf_grid = torch.zeros((1, 1, R, R, R))
        sdf_grid[0, 0, idx[:, 0], idx[:, 1], idx[:, 2]] = partial["sdf"].cpu().float()
        vis_grid = torch.zeros((1, 1, R, R, R))
        vis_grid[0, 0, idx[:, 0], idx[:, 1], idx[:, 2]] = partial["vis"].cpu().float()
        samples, _ = sampling_fn(model, partial=sdf_grid.to(config.device), partial_mask=vis_grid.to(config.device),
                                 freeze_iters=config.eval.freeze_iters)
        np.save(os.path.join(eval_dir, f"{save_fname}.npy"), samples.cpu().numpy())
