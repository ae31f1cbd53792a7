"""TEST INFRASTRUCTURE ONLY -- CPU restatement (plain PyTorch fp32/fp64 ops) of the reference's
3-D U-Net score network, DDPM ancestral step and inpainting blend.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (meshdiffusion_amd/) never does.

Parity pin: `oracle/gen_golden.py` runs this restatement against the IMPORTED reference
(/root/reference, which is a Python repo and imports in the build container) on the same seeded
state dict and inputs, asserts agreement, and writes the golden fixtures in tests/golden/.
The reference itself holds no tests or golden vectors for this path.

Each function cites the reference code it restates (paths relative to the reference root).
The network is driven purely by a reference-format state dict (`all_modules.{i}.*` keys).
"""
import math

import torch
import torch.nn.functional as F


def timestep_embedding(t, dim):
    """lib/diffusion/models/layers.py:542-556."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32, device=t.device) * -scale)
    arg = t.float()[:, None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def nin(x, W, b):
    """layers.py:573-582: 1x1x1 channel contraction with W [in, out]."""
    return torch.einsum("bcdhw,co->bodhw", x, W) + b[None, :, None, None, None]


def group_norm(x, w, b):
    return F.group_norm(x, 32, w, b, eps=1e-6)


class _SD:
    """Prefix view on a state dict."""

    def __init__(self, sd, prefix=""):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[self.prefix + k]

    def __contains__(self, k):
        return (self.prefix + k) in self.sd

    def sub(self, p):
        return _SD(self.sd, self.prefix + p)


def resnet_block(p, x, temb, drop=None):
    """layers.py:672-689.  Eval mode (drop=None): dropout is the identity.  Training: `drop` is the explicit
    {0, 1/(1-p)} factor tensor nn.Dropout would have multiplied by (layers.py:682), supplied by the test."""
    h = F.silu(group_norm(x, p["GroupNorm_0.weight"], p["GroupNorm_0.bias"]))
    h = F.conv3d(h, p["Conv_0.weight"], p["Conv_0.bias"], padding=1)
    h = h + F.linear(F.silu(temb), p["Dense_0.weight"], p["Dense_0.bias"])[:, :, None, None, None]
    h = F.silu(group_norm(h, p["GroupNorm_1.weight"], p["GroupNorm_1.bias"]))
    if drop is not None:
        h = h * drop
    h = F.conv3d(h, p["Conv_1.weight"], p["Conv_1.bias"], padding=1)
    if "NIN_0.W" in p:
        x = nin(x, p["NIN_0.W"], p["NIN_0.b"])
    return x + h


def attn_block(p, x):
    """layers.py:595-608: single-head attention over D*H*W tokens."""
    B, C = x.shape[:2]
    h = group_norm(x, p["GroupNorm_0.weight"], p["GroupNorm_0.bias"])
    q = nin(h, p["NIN_0.W"], p["NIN_0.b"]).reshape(B, C, -1)
    k = nin(h, p["NIN_1.W"], p["NIN_1.b"]).reshape(B, C, -1)
    v = nin(h, p["NIN_2.W"], p["NIN_2.b"]).reshape(B, C, -1)
    w = torch.einsum("bcq,bck->bqk", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w, dim=-1)
    o = torch.einsum("bqk,bck->bcq", w, v).reshape(x.shape)
    return x + nin(o, p["NIN_3.W"], p["NIN_3.b"])


def downsample(p, x):
    """layers.py:633-643: pad (0,1) on every axis, then stride-2 3x3x3 conv."""
    return F.conv3d(F.pad(x, (0, 1, 0, 1, 0, 1)), p["Conv_0.weight"], p["Conv_0.bias"], stride=2)


def upsample(p, x):
    """layers.py:618-623: nearest x2 then 3x3x3 conv."""
    h = F.interpolate(x, scale_factor=2, mode="nearest")
    return F.conv3d(h, p["Conv_0.weight"], p["Conv_0.bias"], padding=1)


def unet_res64_forward(sd, cfg, x, labels):
    """lib/diffusion/models/ddpm_res64.py:126-199 and ddpm_res128.py:137-215 driven by a state dict.

    cfg: dict(nf, ch_mult, num_res_blocks, attn_resolutions, image_size[, level0_blocks]); sd keys may
    carry a leading 'module.' (DataParallel) prefix.  The res128 variant is recognised from the state
    dict itself: 5x5x5 stem/head weights (padding = k//2), no `coords` entry (ddpm_res128.py:77,162),
    and `level0_blocks` = 2 (ddpm_res128.py:98,118).
    """
    if any(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    root = _SD(sd)
    mod = lambda i: root.sub(f"all_modules.{i}.")  # noqa: E731
    nf, ch_mult, nrb = cfg["nf"], cfg["ch_mult"], cfg["num_res_blocks"]
    attn_res, R = cfg["attn_resolutions"], cfg["image_size"]
    nres = len(ch_mult)
    blocks_at = lambda lvl: cfg.get("level0_blocks") or nrb if lvl == 0 else nrb  # noqa: E731
    i = 0
    temb = timestep_embedding(labels, nf)
    temb = F.linear(temb, mod(i)["weight"], mod(i)["bias"]); i += 1
    temb = F.linear(F.silu(temb), mod(i)["weight"], mod(i)["bias"]); i += 1
    pad = mod(i)["weight"].shape[-1] // 2
    h0 = F.conv3d(x, mod(i)["weight"], mod(i)["bias"], padding=pad); i += 1
    if "coords" in sd:
        h0 = h0 + F.conv3d(sd["coords"], sd["pos_layer.weight"], sd["pos_layer.bias"], padding=pad)
    h0 = h0 + F.conv3d(sd["mask"], sd["mask_layer.weight"], sd["mask_layer.bias"], padding=pad)
    hs = [h0]
    for lvl in range(nres):
        for _ in range(blocks_at(lvl)):
            h = resnet_block(mod(i), hs[-1], temb); i += 1
            if h.shape[-1] in attn_res:
                h = attn_block(mod(i), h); i += 1
            hs.append(h)
        if lvl != nres - 1:
            hs.append(downsample(mod(i), hs[-1])); i += 1
    h = hs[-1]
    h = resnet_block(mod(i), h, temb); i += 1
    h = attn_block(mod(i), h); i += 1
    h = resnet_block(mod(i), h, temb); i += 1
    for lvl in reversed(range(nres)):
        for _ in range(blocks_at(lvl) + 1):
            h = resnet_block(mod(i), torch.cat([h, hs.pop()], dim=1), temb); i += 1
        if h.shape[-1] in attn_res:
            h = attn_block(mod(i), h); i += 1
        if lvl != 0:
            h = upsample(mod(i), h); i += 1
    assert not hs
    h = F.silu(group_norm(h, mod(i)["weight"], mod(i)["bias"])); i += 1
    h = F.conv3d(h, mod(i)["weight"], mod(i)["bias"], padding=mod(i)["weight"].shape[-1] // 2); i += 1
    assert f"all_modules.{i}.weight" not in sd
    return h


# ---- sampler pieces ------------------------------------------------------------------------------
def vpsde_tables(N=1000, beta_min=0.1, beta_max=20.0):
    """lib/diffusion/sde_lib.py:189-195."""
    betas = torch.linspace(beta_min / N, beta_max / N, N)
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    return betas, torch.sqrt(alphas_cumprod), torch.sqrt(1.0 - alphas_cumprod)


def ancestral_step(x, eps_hat, z, t, mask, N=1000):
    """models/utils.py:191-198 + sampling.py:222-230 + :476-478 for a batch-constant time t."""
    betas, _, sq1mac = vpsde_tables(N)
    k = (t * (N - 1)).long()
    beta, std = betas[k], sq1mac[k]
    score = -eps_hat / std
    x_mean = (x + beta * score) / torch.sqrt(1.0 - beta)
    x_new = x_mean + torch.sqrt(beta) * z
    if mask is not None:
        x_new, x_mean = x_new * mask, x_mean * mask
    return x_new, x_mean


def sample_k_steps(eps_fn, x0, noises, mask, N=1000, eps=1e-3):
    """First len(noises) iterations of the unconditional loop sampling.py:471-481."""
    timesteps = torch.linspace(1.0, eps, N)
    x, x_mean = x0, x0
    for i, z in enumerate(noises):
        t = timesteps[i]
        labels = torch.ones(x.shape[0]) * t * (N - 1)
        x, x_mean = ancestral_step(x, eps_fn(x, labels), z, t, mask, N)
    return x, x_mean


def marginal_prob_coef(t, beta_0=0.1, beta_1=20.0):
    """sde_lib.py:210-214."""
    lmc = -0.25 * t ** 2 * (beta_1 - beta_0) - 0.5 * t * beta_0
    return torch.exp(lmc), torch.sqrt(1.0 - torch.exp(2.0 * lmc))


def ddim_step(x, eps_hat, t, tprev, N=1000):
    """lib/diffusion/sde_lib.py:113-140 (`discretize_ddim`, use_clip False) for batch time vectors t, tprev; `eps_hat` is
    the network output (the reference builds this sampler's score_fn with std_scale=False, models/utils.py:185-189).
    Returns (x_new, x0_pred), both float64 like the reference."""
    _, sa, s1 = vpsde_tables(N)
    k, kp = (t * (N - 1) / 1).long(), (tprev * (N - 1) / 1).long()
    a1, a2 = sa[k][:, None, None, None, None], s1[k][:, None, None, None, None]
    a1p, a2p = sa[kp][:, None, None, None, None], s1[kp][:, None, None, None, None]
    r1, r2 = a1p.double() / a1.double(), a2p.double() / a2.double()
    x0s = x.double() - a2.double() * eps_hat.double()
    sst = x - x0s
    x0_pred = x0s / a1
    x_new = r1.double() * x + (-r1 + r2.double()) * sst.double()
    return x_new, x0_pred


def ddim_sample(eps_fn, x_init, mask, N=1000, denoise=True, n_iters=None):
    """lib/diffusion/sampling.py:522-569 (`ddim_sampler`, 'quad' schedule, no partial grid) with `encode` read as False."""
    import numpy as np
    seq = [int(v) for v in list(np.linspace(0, np.sqrt(N * 0.8), 100) ** 2)]
    timesteps = torch.tensor(seq) / N
    x = x_init * mask
    x0_pred = x
    order = list(reversed(range(1, len(timesteps))))
    if n_iters is not None:
        order = order[:n_iters]
    for i in order:
        vec_t = torch.ones(x.shape[0]) * timesteps[i]
        vec_tprev = torch.ones(x.shape[0]) * timesteps[i - 1]
        e = eps_fn(x.float(), vec_t.float() * (N - 1))
        x, x0_pred = ddim_step(x, e, vec_t, vec_tprev, N)
        x, x0_pred = x * mask, x0_pred * mask
    return (x0_pred if denoise else x) * mask
