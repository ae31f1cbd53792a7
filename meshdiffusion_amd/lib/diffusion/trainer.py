"""Training loop -- mirror of the reference's lib/diffusion/trainer.py:18-137 (`train(config)`), one process
per GPU.

Same working-directory layout (`{train_dir}/checkpoints/checkpoint_{step}.pth`,
`{train_dir}/checkpoints-meta/checkpoint.pth`, `{train_dir}/tensorboard`), same checkpoint dict, same step /
iter_size / logging / snapshot cadence, same `./data/grid_mask_{R}.pt` lookup relative to the cwd.

What differs by design: the reference runs ONE process that scatters
`config.training.batch_size` samples over all visible GPUs with `torch.nn.DataParallel`, re-broadcasting
1.46 GB of weights every call and reducing gradients to GPU 0.  Here `torchrun --nproc-per-node N
main_diffusion.py --mode=train ...` starts one rank per GPU; each rank holds a static replica, draws its own
disjoint slice of every shuffled epoch (`batch_size // N` samples per step, so the GLOBAL batch is the
reference's), and the only exchange is one RCCL all-reduce (mean) of the flat fp32 gradient buffer before the
optimizer step.  Every rank then applies the identical update, so replicas stay bit-identical without any
parameter broadcast after start-up.  Rank 0 alone logs and writes checkpoints.
"""
import logging
import os
import sys

import torch
import torch.distributed as dist

from . import losses, parallel, sde_lib
from ... import hip_ops as ops
from .models import ddpm_res64, ddpm_res128  # noqa: F401  (registers the models: reference trainer.py:7)
from .models import utils as mutils
from .models.ema import ExponentialMovingAverage
from .utils import restore_checkpoint, save_checkpoint
from ..dataset.shapenet_dmtet_dataset import ShapeNetDMTetDataset


class RankShardSampler(torch.utils.data.Sampler):
    """Epoch-shuffled indices, rank r of W takes positions r, r+W, ... of the common permutation (tail dropped
    so all ranks see the same number of items).  The permutation is seeded by (seed, epoch) so every rank
    draws the same one without communicating.  Deviation from the reference loader (shuffle=True, drop_last=False on
    one process): up to W-1 items per epoch and the last partial batch are skipped, i.e. an epoch is
    floor(floor(n / W) / local_batch) steps -- every step then has the full global batch on every rank."""

    def __init__(self, n, rank, world, seed):
        self.n, self.rank, self.world, self.seed, self.epoch = int(n), int(rank), int(world), int(seed), 0

    def __len__(self):
        return self.n // self.world

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.seed * 100003 + self.epoch)
        self.epoch += 1
        perm = torch.randperm(self.n, generator=g)[: (self.n // self.world) * self.world]
        return iter(perm[self.rank::self.world].tolist())


def _summary_writer(path):
    try:
        from torch.utils import tensorboard
        return tensorboard.SummaryWriter(path)
    except Exception as e:  # tensorboard is an optional dependency of torch
        logging.warning(f"tensorboard unavailable ({e.__class__.__name__}); scalars are logged as text only")
        return None


def train(config):
    rank, world, local = parallel.init_distributed()
    if not torch.cuda.is_available():
        raise RuntimeError("training runs on the HIP path only: no GPU is visible")
    torch.cuda.set_device(local)
    config.device = torch.device("cuda", local)
    # the reference never consumes config.seed; here it pins, per rank, the noise / timestep draws,
    # the dropout mask seeds and the DataLoader workers' base seed, so multi-rank runs are reproducible
    torch.manual_seed(int(config.seed) + rank)
    workdir = config.training.train_dir
    logging.info("working dir: {:s}".format(workdir))
    writer = _summary_writer(os.path.join(workdir, "tensorboard")) if rank == 0 else None
    resolution = config.data.image_size

    score_model = mutils.create_model(config)
    # all replicas start from rank 0's initialisation: the only parameter broadcast of the whole run
    parallel.broadcast_params_(score_model.parameters())
    ema = ExponentialMovingAverage(score_model.parameters(), decay=config.model.ema_rate)
    optimizer = losses.get_optimizer(config, score_model.parameters())
    state = dict(optimizer=optimizer, model=score_model, ema=ema, step=0)

    checkpoint_dir = os.path.join(workdir, "checkpoints")
    checkpoint_meta_dir = os.path.join(workdir, "checkpoints-meta", "checkpoint.pth")
    os.makedirs(checkpoint_dir, exist_ok=True)
    os.makedirs(os.path.dirname(checkpoint_meta_dir), exist_ok=True)
    state = restore_checkpoint(checkpoint_meta_dir, state, config.device)
    initial_step = int(state["step"])

    mask = torch.load(f"./data/grid_mask_{resolution}.pt", map_location="cpu").view(
        1, 1, resolution, resolution, resolution).to(config.device)
    score_model.module.mask.data[:] = mask[:]
    ops.bump_param_epoch()

    if config.training.batch_size % world:
        raise ValueError(f"training.batch_size={config.training.batch_size} must divide over {world} ranks")
    local_batch = config.training.batch_size // world
    dataset = ShapeNetDMTetDataset(config.data.meta_path, deform_scale=config.model.deform_scale, aug=True,
                                   grid_mask=mask, filter_meta_path=config.data.filter_meta_path,
                                   normalize_sdf=config.data.normalize_sdf, extension=config.data.extension)
    sampler = RankShardSampler(len(dataset), rank, world, seed=config.seed)
    loader = torch.utils.data.DataLoader(dataset, batch_size=local_batch, sampler=sampler, drop_last=True,
                                         num_workers=config.data.num_workers, pin_memory=True)
    if len(loader) == 0:
        raise ValueError(f"dataset of {len(dataset)} grids is smaller than one global batch "
                         f"({config.training.batch_size})")
    # a resumed run continues the permutation sequence where the checkpoint left it (state["step"] counts consumed
    # local batches) instead of replaying epoch 0, 1, ...; the position inside the interrupted epoch is not restored
    sampler.epoch = initial_step // len(loader)
    data_iter = iter(loader)

    sde = sde_lib.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
    optimize_fn = losses.optimization_manager(config)
    train_step_fn = losses.get_step_fn(sde, train=True, optimize_fn=optimize_fn, mask=mask,
                                       loss_type=config.training.loss_type)
    iter_size = config.training.iter_size
    num_train_steps = config.training.n_iters
    logging.info("Starting training loop at step %d." % (initial_step // iter_size,))

    for step in range(initial_step // iter_size, num_train_steps + 1):
        tmp_loss = 0.0
        for inner in range(iter_size):
            try:
                batch = next(data_iter)
            except StopIteration:
                data_iter = iter(loader)
                batch = next(data_iter)
            batch = batch.to(config.device, non_blocking=True)
            loss = train_step_fn(state, batch, clear_grad=(inner == 0), update_param=(inner == iter_size - 1))["loss"]
            tmp_loss += loss.item()
        tmp_loss /= iter_size
        if world > 1:
            t = torch.tensor([tmp_loss], dtype=torch.float64, device=config.device)
            dist.all_reduce(t)
            tmp_loss = float(t) / world
        if rank == 0 and step % config.training.log_freq == 0:
            logging.info("step: %d, training_loss: %.5e" % (step, tmp_loss))
            sys.stdout.flush()
            if writer is not None:
                writer.add_scalar("training_loss", tmp_loss, step)
        if rank == 0 and step != 0 and step % config.training.snapshot_freq_for_preemption == 0:
            logging.info(f"save meta at iter {step}")
            save_checkpoint(checkpoint_meta_dir, state)
        if rank == 0 and (step != 0 and step % config.training.snapshot_freq == 0 or step == num_train_steps):
            logging.info(f"save model: {step}-th")
            save_checkpoint(os.path.join(checkpoint_dir, f"checkpoint_{step}.pth"), state)
    if world > 1:
        dist.barrier()
    return state
