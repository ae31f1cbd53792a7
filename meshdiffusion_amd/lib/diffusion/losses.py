"""Optimizer factory / optimisation manager -- mirror of the reference's
lib/diffusion/losses.py:26-52.  (The DDPM loss + backward through the HIP U-Net, losses.py:54-141,
is a later SURVEY 8(a) row; see DESIGN.md "Scope".)"""
import numpy as np
import torch
import torch.optim as optim


def get_optimizer(config, params):
    if config.optim.optimizer != "Adam":
        raise NotImplementedError(f"Optimizer {config.optim.optimizer} not supported yet!")
    return optim.Adam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                      weight_decay=config.optim.weight_decay)


def optimization_manager(config):
    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                    grad_clip=config.optim.grad_clip):
        if warmup > 0:
            for g in optimizer.param_groups:
                g["lr"] = lr * np.minimum(step / warmup, 1.0)
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        optimizer.step()

    return optimize_fn
