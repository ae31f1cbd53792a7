"""Model registry and score-function wrappers -- host-side mirror of the reference's
lib/diffusion/models/utils.py (register_model/get_model :27-47, get_sigmas :50-62,
create_model :88-96, get_model_fn :99-128, get_score_fn :167-203).

Differences (MI355X-first): `create_model` returns a light `ModelReplica` wrapper exposing
`.module` like torch.nn.DataParallel does, but it is ONE static replica per process (one
process per GPU); there is no per-call parameter broadcast.
"""
import numpy as np
import torch

from .. import sde_lib

_MODELS = {}


def register_model(cls=None, *, name=None):
    """Class decorator: register a score model under `name` (default: the class name)."""

    def _register(c):
        key = c.__name__ if name is None else name
        if key in _MODELS:
            raise ValueError(f"Already registered model with name: {key}")
        _MODELS[key] = c
        return c

    return _register if cls is None else _register(cls)


def get_model(name):
    return _MODELS[name]


def get_sigmas(config):
    """SMLD noise levels (geometric from sigma_max to sigma_min); kept because the models
    register it as the `sigmas` buffer that is part of the checkpoint format."""
    m = config.model
    return np.exp(np.linspace(np.log(m.sigma_max), np.log(m.sigma_min), m.num_scales))


class ModelReplica(torch.nn.Module):
    """Stands where the reference has torch.nn.DataParallel: `.module` is the score model and
    state-dict keys carry the 'module.' prefix, so reference checkpoints load unchanged."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, x, labels):
        return self.module(x, labels)


def create_model(config, use_parallel=True):
    model = get_model(config.model.name)(config)
    if use_parallel:
        model = ModelReplica(model).to(config.device)
    return model


def calibrate_model(model, config, batch=2, timesteps=(900.0, 400.0, 60.0), seed=20260930, bar=4e-5):
    """Load-time calibration of the reduced-precision conv arithmetic (DDPMUNet3D.calibrate) on what the sampler will feed the
    model: unit-variance noise on the grid mask at a few timesteps of the schedule (the ancestral trajectory starts as exactly that
    and keeps unit scale: the VP SDE is variance preserving).  Call it once after `load_state_dict` / `restore_checkpoint`;
    `evaler.uncond_gen / cond_gen` do.  Returns the calibration report, or None for a model without the method / in bf16x3."""
    net = model.module if hasattr(model, "module") else model
    if not hasattr(net, "calibrate") or getattr(net, "hip_precision", None) not in ("f16f8", "f16f6"):
        return None
    was_training = net.training
    net.eval()
    try:
        dev = next(net.parameters()).device
        R, C = config.data.image_size, config.data.num_channels
        g = torch.Generator().manual_seed(seed)
        mask = net.mask.detach().to(dev).view(1, 1, R, R, R) if hasattr(net, "mask") else 1.0
        xs, ls = [], []
        for t in timesteps:
            xs.append(torch.randn((batch, C, R, R, R), generator=g).to(dev) * mask)
            ls.append(torch.full((batch,), float(t), device=dev))
        return net.calibrate(xs, ls, bar=bar)
    finally:
        net.train(was_training)


def get_model_fn(model, train=False):
    def model_fn(x, labels):
        model.train() if train else model.eval()
        return model(x, labels)

    return model_fn


def get_score_fn(sde, model, train=False, continuous=False, std_scale=True):
    """score(x, t) = -eps_hat(x, t*(N-1)) / sqrt(1 - alpha_bar[long(t*(N-1))]) for the VP SDE."""
    if continuous:
        raise AssertionError("continuous-time training is not supported on this path (as in the reference)")
    if not isinstance(sde, sde_lib.VPSDE):
        raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
    model_fn = get_model_fn(model, train=train)

    def score_fn(x, t):
        labels = t * (sde.N - 1)
        out = model_fn(x, labels)
        if not std_scale:
            return out
        std = sde.sqrt_1m_alphas_cumprod.to(labels.device)[labels.long()]
        return -out / std[:, None, None, None, None]

    return score_fn


def to_flattened_numpy(x):
    return x.detach().cpu().numpy().reshape((-1,))


def from_flattened_numpy(x, shape):
    return torch.from_numpy(x.reshape(shape))
