"""Config objects for the MeshDiffusion hot path without ml_collections/absl.

Mirrors the fields of the reference's configs (configs/default_configs.py:5-89,
configs/res64.py:6-63, configs/res128.py:6-62) and the `--config.a.b=v` dotted override
syntax of `config_flags.DEFINE_config_file(..., lock_config=False)` (main_diffusion.py:13-16).
Reference config *files* can also be loaded unchanged through `load_config_file`, which
injects a minimal `ml_collections` shim when the real package is absent.
"""
import ast
import importlib.util
import os
import sys
import types

import torch


class ConfigDict(dict):
    """Attribute-style nested dict (the subset of ml_collections.ConfigDict this path uses)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]

    def set_by_path(self, dotted, value):
        keys = dotted.split(".")
        node = self
        for k in keys[:-1]:
            if k not in node or not isinstance(node[k], ConfigDict):
                node[k] = ConfigDict()
            node = node[k]
        node[keys[-1]] = value

    def copy_and_resolve(self):
        out = ConfigDict()
        for k, v in self.items():
            out[k] = v.copy_and_resolve() if isinstance(v, ConfigDict) else v
        return out


def _literal(s):
    try:
        return ast.literal_eval(s)
    except (ValueError, SyntaxError):
        return s


def apply_overrides(config, argv):
    """Apply `--config.a.b=v` (or `--config.a.b v`) overrides; returns the unconsumed args."""
    rest, i = [], 0
    while i < len(argv):
        a = argv[i]
        if a.startswith("--config."):
            body = a[len("--config."):]
            if "=" in body:
                path, val = body.split("=", 1)
            else:
                path, val = body, argv[i + 1]
                i += 1
            config.set_by_path(path, _literal(val))
        else:
            rest.append(a)
        i += 1
    return rest


def get_default_configs():
    c = ConfigDict()
    c.training = ConfigDict(batch_size=64, n_iters=2400001, snapshot_freq=50000, log_freq=50, eval_freq=100,
                            snapshot_freq_for_preemption=5000, snapshot_sampling=True,
                            likelihood_weighting=False, continuous=True, reduce_mean=False, iter_size=1,
                            loss_type="l2", train_dir="PLACEHOLDER")
    c.sampling = ConfigDict(n_steps_each=1, noise_removal=True, probability_flow=False, snr=0.075)
    c.eval = ConfigDict(begin_ckpt=50, end_ckpt=96, batch_size=512, enable_sampling=True, num_samples=50000,
                        enable_loss=True, enable_bpd=False, bpd_dataset="test", ckpt_path="PLACEHOLDER",
                        partial_dmtet_path="PLACEHOLDER", tet_path="PLACEHOLDER", freeze_iters=950)
    c.data = ConfigDict(dataset="LSUN", image_size=256, random_flip=True, uniform_dequantization=False,
                        centered=False, num_channels=3, num_workers=4, normalize_sdf=True,
                        meta_path="PLACEHOLDER", filter_meta_path="PLACEHOLDER", extension="pt")
    c.model = ConfigDict(sigma_max=378, sigma_min=0.01, num_scales=2000, beta_min=0.1, beta_max=20.0,
                         dropout=0.0, embedding_type="fourier", deform_scale=1.0)
    c.optim = ConfigDict(weight_decay=0, optimizer="Adam", lr=2e-4, beta1=0.9, eps=1e-8, warmup=5000,
                         grad_clip=1.0)
    c.seed = 42
    c.device = torch.device("cuda:0") if torch.cuda.is_available() else torch.device("cpu")
    c.render = ConfigDict()
    return c


def get_config_res64():
    c = get_default_configs()
    c.training.update(sde="vpsde", continuous=False, reduce_mean=True, batch_size=48, lip_scale=None,
                      snapshot_freq_for_preemption=1000)
    c.sampling.update(method="pc", predictor="ancestral_sampling", corrector="none")
    c.data.update(dataset="ShapeNet", centered=True, image_size=64, num_channels=4, num_workers=4, aug=True)
    c.model.update(name="ddpm_res64", scale_by_sigma=False, num_scales=1000, ema_rate=0.9999,
                   normalization="GroupNorm", nonlinearity="swish", nf=128, ch_mult=(1, 1, 2, 4, 4),
                   num_res_blocks_first=2, num_res_blocks=3, attn_resolutions=(16,), resamp_with_conv=True,
                   conditional=True, dropout=0.1, hip_precision="f16f6")
    c.optim.lr = 2e-5
    c.eval.batch_size = 4
    c.eval.eval_dir = "PLACEHOLDER"
    c.seed = 42
    return c


def get_config_res128():
    c = get_default_configs()
    c.training.update(sde="vpsde", continuous=False, reduce_mean=True, batch_size=8, iter_size=4,
                      lip_scale=None, snapshot_freq_for_preemption=1000)
    c.sampling.update(method="pc", predictor="ancestral_sampling", corrector="none")
    c.data.update(dataset="ShapeNet", centered=True, image_size=128, num_channels=4, num_workers=4, aug=True)
    # NB the reference config names 'ddpm_res128_v2', which is not a registered model
    # (configs/res128.py:40 vs ddpm_res128.py:41); we register both names.
    c.model.update(name="ddpm_res128", scale_by_sigma=False, num_scales=1000, ema_rate=0.9999,
                   normalization="GroupNorm", nonlinearity="swish", nf=128, ch_mult=(1, 1, 2, 4, 4, 4), num_res_blocks_first=2,
                   num_res_blocks=2, attn_resolutions=(16,), resamp_with_conv=True, conditional=True,
                   dropout=0.1, hip_precision="f16f6")
    c.optim.lr = 2e-5
    c.eval.batch_size = 7
    c.eval.eval_dir = "PLACEHOLDER"
    c.seed = 42
    return c


def load_config_file(path):
    """Import a reference-style config file (defines get_config()) and return its config."""
    path = os.path.abspath(path)
    if "ml_collections" not in sys.modules:
        try:
            import ml_collections  # noqa: F401
        except ImportError:
            shim = types.ModuleType("ml_collections")
            shim.ConfigDict = ConfigDict
            sys.modules["ml_collections"] = shim
    root = os.path.dirname(os.path.dirname(path))
    added = False
    if root not in sys.path:  # reference files do `from configs.default_configs import ...`
        sys.path.insert(0, root)
        added = True
    try:
        spec = importlib.util.spec_from_file_location("_md_user_config", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod.get_config()
    finally:
        if added:
            sys.path.remove(root)
