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
