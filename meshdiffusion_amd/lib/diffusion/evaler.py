"""Generation drivers -- mirror of the reference's lib/diffusion/evaler.py
(`uncond_gen` :14-60, `cond_gen` :134-211).  Same behaviour and file outputs
(`{eval_dir}/{idx}.npy`, float32 [B,4,R,R,R]); the grid mask is read from
`./data/grid_mask_{R}.pt` relative to the cwd like the reference, with `map_location` so that the
CUDA-saved tensor loads on any host.

Multi-GPU (replaces the reference's `torch.nn.DataParallel` inside `create_model`, models/utils.py:88-96, that
evaler.py:29 relies on): under `torchrun --nproc-per-node N main_diffusion.py --mode=uncond_gen|cond_gen` every rank
builds a static replica on `cuda:LOCAL_RANK`, samples its shard of `config.eval.batch_size` (parallel.shard_sizes) with
the global RNG seeded `config.seed + rank`, and rank 0 gathers the shards (the only collective: 4.2 MB per sample) and
writes ONE `{idx}.npy` holding the full batch in rank order.  Parity under sharding is per shard: rank r's
rows equal a single-process run with batch `sizes[r]` after `torch.manual_seed(config.seed + r)`; the reference
itself draws a whole batch from one unseeded generator, so no stronger statement exists.  A single process behaves
exactly as before (no seeding, the caller's RNG state decides the draw).
"""
import os

import numpy as np
import torch

from . import losses, parallel, sampling, sde_lib
from .models import ddpm_res64, ddpm_res128  # noqa: F401  (registers the models, like trainer.py:7 in the reference)
from .models import utils as mutils
from .models.ema import ExponentialMovingAverage
from .utils import restore_checkpoint


def _setup(config, mask_shape, local_batch=None):
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
    shape = (config.eval.batch_size if local_batch is None else local_batch, config.data.num_channels, R, R, R)
    sampling_fn = sampling.get_sampling_fn(config, sde, shape, lambda x: x, 1e-3, grid_mask=grid_mask)
    assert os.path.exists(ckpt_path), ckpt_path
    print("ckpt path:", ckpt_path)
    state = restore_checkpoint(ckpt_path, state, device=config.device)
    ema.copy_to(score_model.parameters())
    print(f"loaded model is trained till iter {state['step'] // config.training.iter_size}")
    # the checkpoint's weights meet the reduced-precision conv formats here for the first time: measure every conv's operand on a
    # few noise batches, rebuild the equalisers from the measurement, audit each conv against its bf16x3 form and demote the ones
    # above the bar (models/utils.calibrate_model; config.eval.calibrate = False skips it)
    if getattr(config.eval, "calibrate", True) and torch.device(config.device).type == "cuda":
        rep = mutils.calibrate_model(score_model, config, batch=shape[0])      # at the sampling batch: which convs take the Winograd path depends on it
        if rep is not None:
            print(f"calibrated the {score_model.module.hip_precision} convs: {rep['measured']} measured, worst kept {rep['worst']:.2e}, "
                  f"demoted to bf16x3: {rep['demoted']}")
    return score_model, sampling_fn, eval_dir


def _ranks(config):
    """(rank, world) of this process; under torchrun (WORLD_SIZE > 1) joins the process group and moves the run to this
    rank's GPU."""
    rank, world, local_rank = parallel.init_distributed()
    if world > 1 and torch.device(config.device).type == "cuda":
        torch.cuda.set_device(local_rank)
        config.device = torch.device("cuda", local_rank)
    return rank, world


def _generate(config, mask_shape, fname, run):
    """Shard `config.eval.batch_size` over the ranks, run `run(model, sampling_fn) -> samples` on each shard, gather to
    rank 0 and write `{eval_dir}/{fname}.npy` once.  Returns the full batch on rank 0 (None elsewhere)."""
    rank, world = _ranks(config)
    total = int(config.eval.batch_size)

    def shard(local_batch, seed):
        if world > 1:
            torch.manual_seed(seed)                     # CPU generator (prior) and this rank's device generator (per-step noise)
        model, sampling_fn, _ = _setup(config, mask_shape, local_batch=local_batch)
        return run(model, sampling_fn)

    with torch.no_grad():
        samples = parallel.sharded_sample(shard, total, int(getattr(config, "seed", 0)))
    if rank == 0:
        os.makedirs(config.eval.eval_dir, exist_ok=True)
        np.save(os.path.join(config.eval.eval_dir, f"{fname}.npy"), samples.cpu().numpy())
    if world > 1:
        parallel.barrier()                              # nobody leaves (and tears the group down) before the file exists
    ops_release()
    return samples


def ops_release():
    from ... import hip_ops as ops
    ops.release_scratch()                               # the Winograd operand buffer (GBs at 64^3) is not needed after a run


def uncond_gen(config, idx=0):
    """Unconditional generation: N-1 ancestral steps from the masked prior, saves {idx}.npy."""
    return _generate(config, lambda R: (1, R, R, R), idx, lambda model, sampling_fn: sampling_fn(model)[0])


def tet_vertices_to_grid_index(vertices):
    """Map tet-grid vertex positions to integer cubic-grid coordinates (evaler.py:187-195)."""
    uniq = vertices[:].unique()
    dx = uniq[1] - uniq[0]
    return torch.round((vertices - vertices.min()) / dx).long()


def cond_gen(config, save_fname="0"):
    """Conditional generation from a partial DMTet (2.5D view) scattered into the cubic grid."""
    R = config.data.image_size
    partial = torch.load(config.eval.partial_dmtet_path, map_location="cpu", weights_only=False)
    tet = np.load(config.eval.tet_path)
    idx = tet_vertices_to_grid_index(torch.tensor(tet["vertices"]))
    sdf_grid = torch.zeros((1, 1, R, R, R))
    sdf_grid[0, 0, idx[:, 0], idx[:, 1], idx[:, 2]] = partial["sdf"].cpu().float()
    vis_grid = torch.zeros((1, 1, R, R, R))
    vis_grid[0, 0, idx[:, 0], idx[:, 1], idx[:, 2]] = partial["vis"].cpu().float()

    def run(model, sampling_fn):
        return sampling_fn(model, partial=sdf_grid.to(config.device), partial_mask=vis_grid.to(config.device),
                           freeze_iters=config.eval.freeze_iters)[0]

    return _generate(config, lambda R: (1, 1, R, R, R), save_fname, run)
