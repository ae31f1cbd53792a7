"""Deterministic synthetic weights / inputs for benchmarks and parity tests.

The reference's default initialisation makes the network output ~0 (every Conv_1, NIN_3 and the
head are scaled by 1e-10: layers.py:90,593,662, ddpm_res64.py:121), so parity on default-init
weights is blind.  `sensitised_state_dict` therefore draws EVERY tensor from a
generator seeded by (seed, crc32(key)): independent of module construction order and identical on
every host with the same torch build, so the 1.46 GB of res64 weights never have to be shipped.
"""
import zlib

import numpy as np
import torch

from .config import get_config_res64


def small_config(image_size=16, nf=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,)):
    """A U-Net small enough for the CPU oracle to run in seconds but covering every kernel config."""
    c = get_config_res64()
    c.data.image_size = image_size
    c.model.update(nf=nf, ch_mult=ch_mult, num_res_blocks=num_res_blocks, attn_resolutions=attn_resolutions,
                   dropout=0.0)
    return c


def small_config_res128(image_size=16, nf=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,)):
    """Same idea for the res128 architecture (5x5x5 stem/head, no coords, 2 blocks at level 0)."""
    c = small_config(image_size, nf, ch_mult, num_res_blocks, attn_resolutions)
    c.model.name = "ddpm_res128"
    return c


def oracle_cfg(config):
    m = config.model
    out = dict(nf=m.nf, ch_mult=tuple(m.ch_mult), num_res_blocks=m.num_res_blocks,
               attn_resolutions=tuple(m.attn_resolutions), image_size=config.data.image_size)
    if m.name.startswith("ddpm_res128"):
        out["level0_blocks"] = 2
    return out


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) % (2 ** 63 - 1))
    return g


def sensitised_state_dict(template, seed=1234, grid_mask=None):
    """template: a state dict (shapes/keys).  Returns a new CPU fp32/fp64 state dict."""
    out = {}
    for key, ref in template.items():
        shape, g = tuple(ref.shape), _gen(seed, key)
        leaf = key.split(".")[-1]
        if key.endswith("sigmas"):
            out[key] = ref.detach().clone().cpu()
        elif key.endswith("coords"):
            out[key] = torch.rand(shape, generator=g) * 2.0 - 1.0
        elif key.endswith("mask") and len(shape) == 5:
            if grid_mask is not None:
                out[key] = grid_mask.reshape(shape).float().cpu().clone()
            else:
                out[key] = (torch.rand(shape, generator=g) < 0.116).float()
        elif "GroupNorm" in key or (len(shape) == 1 and leaf == "weight"):
            # GroupNorm affine (the final norm is a bare `all_modules.{i}.weight/.bias` of ndim 1)
            if leaf == "weight":
                out[key] = 1.0 + 0.05 * torch.randn(shape, generator=g)
            else:
                out[key] = 0.02 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            out[key] = 0.02 * torch.randn(shape, generator=g)
        else:
            recept = float(np.prod(shape)) / shape[0] / shape[1]
            fan_avg = (shape[0] + shape[1]) * recept / 2.0
            a = float(np.sqrt(3.0 / fan_avg))
            out[key] = (torch.rand(shape, generator=g) * 2.0 - 1.0) * a
    return out


def trained_like_state_dict(template, seed=4321, grid_mask=None, span=3.0, df=3.0):
    """A second, ADVERSARIAL sensitisation: what a trained checkpoint can look like and i.i.d. weights never do.
      * conv / NIN / Linear weights are heavy-tailed: Student-t(df) instead of uniform, same variance (outlier weights inside every
        16-channel block of a row);
      * every GroupNorm that feeds a convolution or the attention projections gets a per-channel scale s_c = 2^U(-span, span)
        on its affine (gamma_c, beta_c) *= s_c, compensated in the consumer: W[:, c] /= s_c -- the channels of a block differ by
        up to 2^(2 span) in magnitude while every channel keeps mattering equally (the case MX block scaling is weakest in).
    Same keys / shapes / determinism as sensitised_state_dict (trained checkpoints are external downloads: reference README.md:35-37)."""
    out = sensitised_state_dict(template, seed=seed, grid_mask=grid_mask)
    for key, v in list(out.items()):
        if v.dim() >= 2 and not key.endswith(("coords", "mask")):
            shape, g = tuple(v.shape), _gen(seed + 1, key)
            recept = float(np.prod(shape)) / shape[0] / shape[1]
            std = float(np.sqrt(1.0 / ((shape[0] + shape[1]) * recept / 2.0)))
            z = torch.randn(shape, generator=g)
            chi = torch.randn((int(df),) + shape, generator=g).pow(2).sum(0) / float(df)
            out[key] = z / chi.sqrt() * (std / float(np.sqrt(df / (df - 2.0))))
    n_mod = 1 + max(int(k.split(".")[1]) for k in out if k.startswith("all_modules."))

    def rescale(gn_prefix, consumers):
        gw, gb = gn_prefix + "weight", gn_prefix + "bias"
        s = torch.exp2((torch.rand(out[gw].shape, generator=_gen(seed + 2, gw)) * 2.0 - 1.0) * span)
        out[gw] = out[gw] * s
        out[gb] = out[gb] * s
        for ck, axis in consumers:
            w = out[ck]
            out[ck] = w / s.view([-1 if a == axis else 1 for a in range(w.dim())])

    for i in range(n_mod):
        p = f"all_modules.{i}."
        for j in (0, 1):
            if p + f"GroupNorm_{j}.weight" in out and p + f"Conv_{j}.weight" in out:
                rescale(p + f"GroupNorm_{j}.", [(p + f"Conv_{j}.weight", 1)])
        if p + "GroupNorm_0.weight" in out and p + "NIN_2.W" in out and p + "Conv_0.weight" not in out:      # AttnBlock: q, k, v
            rescale(p + "GroupNorm_0.", [(p + f"NIN_{j}.W", 0) for j in (0, 1, 2)])
    fin, head = f"all_modules.{n_mod - 2}.", f"all_modules.{n_mod - 1}.weight"
    if out[fin + "weight"].dim() == 1 and out[head].dim() == 5:
        rescale(fin, [(head, 1)])
    return out


def synthetic_grid_mask(R, seed=7):
    """Binary [R,R,R] mask with ~11.6% live cells on a period-4 lattice (the
    res128 asset is missing upstream, and tests must not depend on the reference tree)."""
    pat = {(0, 1, 1), (2, 1, 1), (3, 0, 3), (3, 2, 3), (1, 3, 0), (1, 3, 2), (1, 1, 3), (3, 3, 1)}
    idx = torch.arange(R)
    X, Y, Z = torch.meshgrid(idx, idx, idx, indexing="ij")
    m = torch.zeros(R, R, R)
    for (a, b, c) in pat:
        m[((X % 4) == a) & ((Y % 4) == b) & ((Z % 4) == c)] = 1.0
    m[R - 1, :, :] = 0; m[:, R - 1, :] = 0; m[:, :, R - 1] = 0
    return m


def synthetic_inputs(B, C, R, seed=42):
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return torch.randn((B, C, R, R, R), generator=g)
