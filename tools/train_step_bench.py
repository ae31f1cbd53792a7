"""BASELINE config #3 on the GPUs that are visible: res64 training step (forward + loss + backward + gradient
all-reduce + fused clip/Adam/EMA), batch 8 per GPU, dropout as configured (0.1).  One process per GPU:

    python tools/train_step_bench.py --steps 2 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_step_bench.py

Prints one JSON line from rank 0 (samples/s = world * batch * steps / wall; time split of the last step).
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from meshdiffusion_amd import synth  # noqa: E402
from meshdiffusion_amd.config import get_config_res64, get_config_res128  # noqa: E402
from meshdiffusion_amd.lib.diffusion import losses, parallel, sde_lib  # noqa: E402
from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, ddpm_res128, utils as mutils  # noqa: F401,E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--small", action="store_true", help="small U-Net (CI-sized)")
    ap.add_argument("--arch", default="res64", choices=["res64", "res128"])
    a = ap.parse_args()
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.arch == "res128":
        cfg = synth.small_config_res128() if a.small else get_config_res128()
    else:
        cfg = synth.small_config() if a.small else get_config_res64()
    cfg.device = dev
    R = cfg.data.image_size
    model = mutils.create_model(cfg)
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    del sd
    model.train()
    opt = losses.FusedAdamEMA(model.parameters(), lr=cfg.optim.lr, beta1=cfg.optim.beta1, eps=cfg.optim.eps,
                              weight_decay=cfg.optim.weight_decay, ema_decay=cfg.model.ema_rate,
                              grad_clip=cfg.optim.grad_clip, warmup=cfg.optim.warmup)
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    mask = synth.synthetic_grid_mask(R).view(1, 1, R, R, R).to(dev)
    loss_fn = losses.get_ddpm_loss_fn(sde, train=True, mask=mask)
    g = torch.Generator().manual_seed(100 + rank)
    x0 = torch.sign(torch.randn((a.batch, 1, R, R, R), generator=g))
    batch = (torch.cat([x0, torch.rand((a.batch, 3, R, R, R), generator=g) * 2 - 1], 1) * mask.cpu()).to(dev)
    torch.manual_seed(7 + rank)
    split = {}

    def step(i):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss = loss_fn(model, batch)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        parallel.allreduce_grads_(opt.grad)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        opt.step(i + 1)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        split.update(fwd_loss_ms=(t1 - t0) * 1e3, bwd_ms=(t2 - t1) * 1e3, allreduce_ms=(t3 - t2) * 1e3, opt_ms=(t4 - t3) * 1e3)
        return float(loss.detach())

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    ls = [step(a.warmup + i) for i in range(a.steps)]
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    wall = time.perf_counter() - t0
    wt = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(wt, op=torch.distributed.ReduceOp.MAX)
    if rank == 0:
        w = float(wt)
        print(json.dumps({"metric": f"{a.arch} training step (fwd+bwd+Adam/EMA), samples/s", "value": round(world * a.batch * a.steps / w, 3),
                          "n_gpus": world, "batch_per_gpu": a.batch, "steps": a.steps, "s_per_step": round(w / a.steps, 3),
                          "losses": [round(v, 5) for v in ls], "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                          "last_step_split_ms": {k: round(v, 1) for k, v in split.items()}, "dropout": cfg.model.dropout,
                          "model": ("small " if a.small else "") + cfg.model.name}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
