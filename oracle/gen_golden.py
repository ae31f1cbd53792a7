"""Generate the golden fixtures in tests/golden/ by running the IMPORTED reference
(/root/reference, read-only) on CPU, and pin the oracle restatements against it.

Run in the build container only (the GPU box has no /root/reference):
    python oracle/gen_golden.py [--skip-res64]

What is recorded (all inputs are regenerated from seeds by meshdiffusion_amd.synth, so only
outputs are stored):
  unet_small.npz     reference DDPMRes64 (small config) eps_hat, full tensor
  unet_res64.npz     reference DDPMRes64 (res64, B=1) eps_hat: ::4 subsample + statistics
  unet_res64_trained.npz  the same on the adversarial "trained-like" weights (--only trained): B = 2, two timesteps
  sampler_small.npz  unmodified reference pc_sampler, first K iterations, uncond + inpainting
  ddim.npz           reference discretize_ddim on seeded inputs, the quad schedule, the whole DDIM sampler (small config)
  sampler_res64.npz  BASELINE config #1: res64, B=1, first 10 of 1000 ancestral steps (live cells + statistics)
  sampler_cond_res64_b32.npz  BASELINE config #5: cond_gen res64, B=32, first 5 iterations of the inpainting sampler
  sampler_res128_b2.npz       BASELINE config #4: res128, B=2, first 2 ancestral steps
  train_grads.npz    reference loss function (train mode): loss + per-parameter gradient norms / samples
  train_grads_b2.npz the same for the real res64 network at B = 2 (--only train_b2)
  train_grads_trained.npz the same at B = 1 on the adversarial trained-like weights (--only train_trained)
  dataset.npz        reference ShapeNetDMTetDataset items (augmentation on/off) for seeded on-disk grids
  dmtet.npz          reference DMTet.__call__ on the shipped 64-grid: counts, hashes, samples
  64_tets_cropped.npz  the tet-grid DATA asset (vertices/indices), copied verbatim
Every reference result is also compared with the oracle restatement (assert), which is the pin
that lets the GPU box use the oracle as the checker.
"""
import argparse
import hashlib
import os
import shutil
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from meshdiffusion_amd import synth  # noqa: E402
from meshdiffusion_amd.config import ConfigDict  # noqa: E402
from oracle import dmtet_oracle, unet_oracle  # noqa: E402


def rel_l2(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def import_reference():
    """Import lib.diffusion from the reference on a GPU-less host."""
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self  # sde_lib.py / sampling.py hard-code .cuda()
    sys.path.insert(0, REF)
    from lib.diffusion import sampling as rsampling, sde_lib as rsde  # noqa: E402
    from lib.diffusion.models import ddpm_res64 as rmodel, ddpm_res128 as rmodel128, utils as rmutils  # noqa: F401,E402
    return rsampling, rsde, rmutils


def ref_model(rmutils, config, sd):
    m = rmutils.create_model(config, use_parallel=False)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.eval()


def make_sd(config, R, seed=1234):
    # template shapes from OUR module tree (identical keys/shapes are asserted by strict load above)
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, ddpm_res128, utils as mutils  # noqa: F401
    c = ConfigDict(config)
    tmpl = mutils.create_model(c, use_parallel=False).state_dict()
    return synth.sensitised_state_dict(tmpl, seed=seed, grid_mask=synth.synthetic_grid_mask(R))


def run_ref_sampler(rsampling, rsde, config, model, shape, mask, K, seed, cond=None):
    sde = rsde.VPSDE(beta_min=config.model.beta_min, beta_max=config.model.beta_max, N=config.model.num_scales)
    fn = rsampling.get_sampling_fn(config, sde, shape, lambda x: x, 1e-3, grid_mask=mask)
    old = rsampling.tqdm.trange
    rsampling.tqdm.trange = lambda n, *a, **k: range(min(n, K))  # first K iterations of the real loop
    try:
        torch.manual_seed(seed)
        if cond is None:
            out, _ = fn(model)
        else:
            out, _ = fn(model, partial=cond[0], partial_mask=cond[1], freeze_iters=cond[2])
    finally:
        rsampling.tqdm.trange = old
    return out


def oracle_sampler(sd, ocfg, shape, mask, K, seed, N=1000):
    torch.manual_seed(seed)
    x0 = torch.randn(*shape) * mask
    noises = []

    def eps_fn(x, labels):
        return unet_oracle.unet_res64_forward(sd, ocfg, x, labels)

    timesteps = torch.linspace(1.0, 1e-3, N)
    x, x_mean = x0, x0
    for i in range(K):
        labels = torch.ones(shape[0]) * timesteps[i] * (N - 1)
        e = eps_fn(x, labels)
        z = torch.randn_like(x)
        x, x_mean = unet_oracle.ancestral_step(x, e, z, timesteps[i], mask, N)
    return x_mean


def gen_unet_and_sampler(skip_res64):
    rsampling, rsde, rmutils = import_reference()
    with torch.no_grad():
        # ---------------- small config ----------------
        cfg = synth.small_config(); cfg.device = torch.device("cpu")
        R = cfg.data.image_size
        sd = make_sd(cfg, R)
        model = ref_model(rmutils, cfg, sd)
        x = synth.synthetic_inputs(2, 4, R, seed=42)
        labels = torch.tensor([500.3, 12.7])
        y_ref = model(x, labels)
        y_or = unet_oracle.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, labels)
        e = rel_l2(y_or, y_ref)
        print(f"[small] oracle vs reference U-Net rel-L2 = {e:.3e}; out std {float(y_ref.std()):.3f}")
        assert e < 1e-5
        np.savez_compressed(os.path.join(GOLD, "unet_small.npz"), y=y_ref.numpy(), labels=labels.numpy(),
                            x_seed=42, sd_seed=1234)

        mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
        K = 6
        xm_ref = run_ref_sampler(rsampling, rsde, cfg, model, (2, 4, R, R, R), mask, K, seed=77)
        xm_or = oracle_sampler(sd, synth.oracle_cfg(cfg), (2, 4, R, R, R), mask, K, seed=77)
        e = rel_l2(xm_or, xm_ref)
        print(f"[small] oracle vs reference {K}-step sampler rel-L2 = {e:.3e}")
        assert e < 1e-5
        # inpainting branch of the unmodified reference sampler
        g = torch.Generator().manual_seed(5)
        partial = torch.sign(torch.randn((1, 1, R, R, R), generator=g))
        pmask = (torch.rand((1, 1, R, R, R), generator=g) < 0.5).float() * mask.view(1, 1, R, R, R)
        mask5 = mask.view(1, 1, R, R, R)
        xc_ref = run_ref_sampler(rsampling, rsde, cfg, model, (2, 4, R, R, R), mask5, K, seed=78,
                                 cond=(partial, pmask, 4))
        np.savez_compressed(os.path.join(GOLD, "sampler_small.npz"), uncond=xm_ref.numpy(), cond=xc_ref.numpy(),
                            K=K, uncond_seed=77, cond_seed=78, cond_data_seed=5, freeze_iters=4)

        # ---------------- res128 architecture, small config ----------------
        cfg = synth.small_config_res128(); cfg.device = torch.device("cpu")
        R = cfg.data.image_size
        sd = make_sd(cfg, R, seed=4321)
        model = ref_model(rmutils, cfg, sd)
        x = synth.synthetic_inputs(2, 4, R, seed=44)
        labels = torch.tensor([700.1, 3.3])
        y_ref = model(x, labels)
        y_or = unet_oracle.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, labels)
        e = rel_l2(y_or, y_ref)
        print(f"[small res128] oracle vs reference U-Net rel-L2 = {e:.3e}; out std {float(y_ref.std()):.3f}")
        assert e < 1e-5
        np.savez_compressed(os.path.join(GOLD, "unet_small_res128.npz"), y=y_ref.numpy(), labels=labels.numpy(),
                            x_seed=44, sd_seed=4321)

        if skip_res64:
            return
        # ---------------- res64 ----------------
        from meshdiffusion_amd.config import get_config_res64
        cfg = get_config_res64(); cfg.device = torch.device("cpu")
        R = 64
        t0 = time.time()
        sd = make_sd(cfg, R)
        model = ref_model(rmutils, cfg, sd)
        print(f"[res64] weights ready in {time.time() - t0:.1f}s")
        x = synth.synthetic_inputs(1, 4, R, seed=42)
        labels = torch.tensor([500.3])
        t0 = time.time(); y_ref = model(x, labels); t_ref = time.time() - t0
        t0 = time.time(); y_or = unet_oracle.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, labels); t_or = time.time() - t0
        e = rel_l2(y_or, y_ref)
        print(f"[res64] oracle vs reference rel-L2 = {e:.3e}; ref {t_ref:.1f}s oracle {t_or:.1f}s; std {float(y_ref.std()):.3f}")
        assert e < 1e-5
        np.savez_compressed(os.path.join(GOLD, "unet_res64.npz"), y_sub=y_ref[:, :, ::4, ::4, ::4].numpy(),
                            y_norm=float(y_ref.double().norm()), y_sum=y_ref.double().sum(dim=(0, 2, 3, 4)).numpy(),
                            y_row=y_ref[0, :, 31, 17, :].numpy(), labels=labels.numpy(), x_seed=42, sd_seed=1234)
        if os.environ.get("MD_GOLD_RES128", "1") == "1":
            gen_res128_full(rmutils)
        mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
        t0 = time.time()
        xm = run_ref_sampler(rsampling, rsde, cfg, model, (1, 4, R, R, R), mask, 10, seed=42)
        print(f"[res64] reference 10-step sampler {time.time() - t0:.1f}s")
        np.savez_compressed(os.path.join(GOLD, "sampler_res64.npz"), xm_norm=float(xm.double().norm()),
                            xm_row=xm[0, :, 33, 17, :].numpy(), K=10, seed=42,
                            **sample_stats(xm, synth.synthetic_grid_mask(R), 4))


def gen_trained_like():
    """unet_res64_trained.npz: the unmodified reference DDPMRes64 (res64, B = 1, t = 500.3 and a late step t = 37.8) on the ADVERSARIAL
    sensitisation synth.trained_like_state_dict (Student-t weights, GroupNorm gammas 2^U(-3, 3) compensated in the consuming conv /
    NIN): the gate under the reduced-precision conv arithmetics of the HIP path.  A 25-step B = 8 sampler run on
    these weights is checked against the oracle on the GPU (tests/test_gpu_graded.py); the oracle is pinned here."""
    rsampling, rsde, rmutils = import_reference()
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    with torch.no_grad():
        cfg = get_config_res64(); cfg.device = torch.device("cpu")
        R = 64
        tmpl = mutils.create_model(ConfigDict(cfg), use_parallel=False).state_dict()
        sd = synth.trained_like_state_dict(tmpl, seed=4321, grid_mask=synth.synthetic_grid_mask(R))
        del tmpl
        model = ref_model(rmutils, cfg, sd)
        x = synth.synthetic_inputs(2, 4, R, seed=52)
        labels = torch.tensor([500.3, 37.8])
        t0 = time.time(); y_ref = model(x, labels); t_ref = time.time() - t0
        y_or = unet_oracle.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, labels)
        e = rel_l2(y_or, y_ref)
        print(f"[res64 trained-like] oracle vs reference rel-L2 = {e:.3e}; ref {t_ref:.1f}s; std {float(y_ref.std()):.3f}", flush=True)
        assert e < 1e-5
        np.savez_compressed(os.path.join(GOLD, "unet_res64_trained.npz"), y_sub=y_ref[:, :, ::4, ::4, ::4].numpy(),
                            y_norm=y_ref.double().flatten(1).norm(dim=1).numpy(), y_sum=y_ref.double().sum(dim=(2, 3, 4)).numpy(),
                            y_row=y_ref[:, :, 31, 17, :].numpy(), labels=labels.numpy(), x_seed=52, sd_seed=4321)
        del model, sd
        # unet_res128_trained.npz: the same gate for ddpm_res128 at 128^3 (configs[3]'s network), one evaluation
        from meshdiffusion_amd.config import get_config_res128
        from meshdiffusion_amd.lib.diffusion.models import ddpm_res128  # noqa: F401
        cfg = get_config_res128(); cfg.device = torch.device("cpu")
        tmpl = mutils.create_model(ConfigDict(cfg), use_parallel=False).state_dict()
        sd = synth.trained_like_state_dict(tmpl, seed=777, grid_mask=synth.synthetic_grid_mask(128))
        del tmpl
        model = ref_model(rmutils, cfg, sd)
        x = synth.synthetic_inputs(1, 4, 128, seed=53)
        labels = torch.tensor([612.4])
        t0 = time.time(); y = model(x, labels); dt = time.time() - t0
        print(f"[res128 trained-like] reference forward {dt:.1f}s; std {float(y.std()):.3f}", flush=True)
        np.savez_compressed(os.path.join(GOLD, "unet_res128_trained.npz"), y_sub=y[:, :, ::8, ::8, ::8].numpy(),
                            y_norm=float(y.double().norm()), y_row=y[0, :, 63, 17, :].numpy(), labels=labels.numpy(),
                            x_seed=53, sd_seed=777)


def live_cells(mask, stride):
    """Flat indices of every `stride`-th LIVE cell of a [R,R,R] grid mask.  The sampler multiplies its state by the
    mask, and the lattice has no live cell on a ::4 sub-grid, so fixtures of sampled grids record live cells."""
    idx = torch.nonzero(torch.as_tensor(mask).reshape(-1) > 0).reshape(-1)
    return idx[::stride]


def sample_stats(x, mask, stride):
    """What a sampled-grid fixture stores: values at every `stride`-th live cell (all samples, all channels), the
    per-sample norms and the per-(sample, channel) sums."""
    x = torch.as_tensor(x)
    B, C = x.shape[:2]
    li = live_cells(mask, stride)
    flat = x.reshape(B, C, -1)
    return dict(live=flat[:, :, li].numpy().copy(), norms=flat.double().reshape(B, -1).norm(dim=1).numpy(),
                sums=flat.double().sum(dim=2).numpy(), l1=flat.double().abs().sum(dim=2).numpy(), stride=np.int64(stride))


def cond_inputs(R, mask, seed=5):
    """BASELINE config #5 conditioning: `partial` = sign of a smooth SDF (channel 0 of a held-out grid),
    `partial_mask` = half-space x < R/2 intersected with the grid mask; both [1,1,R,R,R].  Shared with tests/."""
    g = torch.Generator().manual_seed(seed)
    ax = torch.linspace(-1, 1, R)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    c = torch.rand(3, generator=g) * 0.3 - 0.15
    sdf = 0.55 - ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2).sqrt() + 0.08 * torch.sin(7 * X + 3 * Y)
    m5 = torch.as_tensor(mask).reshape(1, 1, R, R, R).float()
    partial = torch.where(sdf > 0, torch.ones(()), -torch.ones(())).reshape(1, 1, R, R, R)
    half = (torch.arange(R).view(R, 1, 1) < R // 2).float().expand(R, R, R).reshape(1, 1, R, R, R)
    return partial, half * m5


def gen_graded(which):
    """Sampler fixtures at the BASELINE batch sizes, from the UNMODIFIED reference sampler on the CPU:
      cond32  : configs[4] cond_gen res64, B=32, first 5 iterations (freeze_iters 950: blend + re-noise every iteration)
      res128  : configs[3] res128, B=2, first 2 ancestral steps (synthetic 128 mask: the asset is missing upstream)
      config1 : configs[0] res64 B=1 first 10 steps, re-recorded on live cells"""
    rsampling, rsde, rmutils = import_reference()
    from meshdiffusion_amd.config import get_config_res64, get_config_res128
    with torch.no_grad():
        if "config1" in which or "cond32" in which:
            cfg = get_config_res64(); cfg.device = torch.device("cpu")
            R = 64
            sd = make_sd(cfg, R)
            model = ref_model(rmutils, cfg, sd)
            del sd
            mask = synth.synthetic_grid_mask(R)
        if "config1" in which:
            t0 = time.time()
            xm = run_ref_sampler(rsampling, rsde, cfg, model, (1, 4, R, R, R), mask.view(1, R, R, R), 10, seed=42)
            print(f"[res64] reference 10-step sampler {time.time() - t0:.1f}s", flush=True)
            np.savez_compressed(os.path.join(GOLD, "sampler_res64.npz"), xm_norm=float(xm.double().norm()),
                                xm_row=xm[0, :, 33, 17, :].numpy(), K=10, seed=42, **sample_stats(xm, mask, 4))
        if "cond32" in which:
            B, K = 32, 5
            partial, pmask = cond_inputs(R, mask)
            t0 = time.time()
            xc = run_ref_sampler(rsampling, rsde, cfg, model, (B, 4, R, R, R), mask.view(1, 1, R, R, R), K, seed=91,
                                 cond=(partial, pmask, 950))
            print(f"[res64 cond_gen B={B}] reference {K} iterations {time.time() - t0:.1f}s", flush=True)
            np.savez_compressed(os.path.join(GOLD, "sampler_cond_res64_b32.npz"), B=B, K=K, seed=91, cond_seed=5,
                                freeze_iters=950, **sample_stats(xc, mask, 32))
        if "res128" in which:
            cfg = get_config_res128(); cfg.device = torch.device("cpu")
            sd = make_sd(cfg, 128, seed=99)
            model = ref_model(rmutils, cfg, sd)
            del sd
            mask = synth.synthetic_grid_mask(128)
            t0 = time.time()
            xm = run_ref_sampler(rsampling, rsde, cfg, model, (2, 4, 128, 128, 128), mask.view(1, 128, 128, 128), 2, seed=17)
            print(f"[res128 B=2] reference 2-step sampler {time.time() - t0:.1f}s", flush=True)
            np.savez_compressed(os.path.join(GOLD, "sampler_res128_b2.npz"), B=2, K=2, seed=17, sd_seed=99,
                                **sample_stats(xm, mask, 32))


def gen_ddim():
    """DDIM.  Pinned against the unmodified reference:
      * `discretize_ddim` (sde_lib.py:113-140) on seeded x / eps for three (t, tprev) pairs of the quad schedule;
      * the whole `ddim_sampler` (sampling.py:522-569) on the small config with config.sampling.noise_removal = False --
        the only setting the reference can run: with noise_removal True its return statement reads the undefined name
        `encode` (sampling.py:569) and raises NameError, which is asserted here.
    The oracle restatement (unet_oracle.ddim_step / ddim_sample) is asserted equal to both."""
    rsampling, rsde, rmutils = import_reference()
    out = {}
    with torch.no_grad():
        cfg = synth.small_config(); cfg.device = torch.device("cpu")
        R = cfg.data.image_size
        sde = rsde.VPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
        g = torch.Generator().manual_seed(31)
        x = torch.randn((2, 4, 4, 4, 4), generator=g)
        eps_fix = torch.randn((2, 4, 4, 4, 4), generator=g)
        rs = sde.reverse(lambda xx, tt: eps_fix, probability_flow=False)
        seq = [int(v) for v in list(np.linspace(0, np.sqrt(sde.N * 0.8), 100) ** 2)]
        ts = torch.tensor(seq) / sde.N
        out["seq"] = np.array(seq, dtype=np.int64)
        for n, i in enumerate((99, 50, 5)):
            vt, vp = torch.ones(2) * ts[i], torch.ones(2) * ts[i - 1]
            xin = x if n == 0 else x.double() * 0.7          # float32 state (first update) and float64 states (later ones)
            xn, x0p = rs.discretize_ddim(xin, vt, tprev=vp)
            assert xn.dtype == torch.float64 and x0p.dtype == torch.float64
            xo, x0o = unet_oracle.ddim_step(xin, eps_fix, vt, vp, sde.N)
            assert torch.equal(xo, xn) and torch.equal(x0o, x0p), "oracle ddim_step != reference discretize_ddim"
            out[f"step{n}_x_new"], out[f"step{n}_x0_pred"], out[f"step{n}_i"] = xn.numpy(), x0p.numpy(), np.int64(i)
        out["step_seed"] = np.int64(31)
        # ---- whole sampler, small config ----
        sd = make_sd(cfg, R)
        model = ref_model(rmutils, cfg, sd)
        mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
        cfg.sampling.method = "ddim"
        for nr, expect_error in ((True, True), (False, False)):
            cfg.sampling.noise_removal = nr
            fn = rsampling.get_sampling_fn(cfg, sde, (2, 4, R, R, R), lambda v: v, 1e-3, grid_mask=mask)
            torch.manual_seed(55)
            try:
                res, _ = fn(model)
                failed = False
            except NameError as e:
                failed = True
                print(f"[ddim] reference with noise_removal={nr}: NameError({e})")
            assert failed == expect_error
        torch.manual_seed(55)
        x_init = torch.randn(2, 4, R, R, R)
        xo = unet_oracle.ddim_sample(lambda xx, lb: unet_oracle.unet_res64_forward(sd, synth.oracle_cfg(cfg), xx, lb), x_init,
                                     mask, sde.N, denoise=False)
        e = rel_l2(xo, res)
        print(f"[ddim] oracle vs reference 99-evaluation DDIM sampler (small config) rel-L2 = {e:.3e}; out std {float(res.std()):.3f}")
        assert e < 1e-6 and res.dtype == torch.float64
        out["sampler_small"], out["sampler_seed"] = res.numpy().astype(np.float64), np.int64(55)
    np.savez_compressed(os.path.join(GOLD, "ddim.npz"), **out)


def gen_res128_full(rmutils):
    """Reference DDPMRes128 at 128^3 (BASELINE config #4 shape), one evaluation: ::8 subsample + statistics."""
    from meshdiffusion_amd.config import get_config_res128
    cfg = get_config_res128(); cfg.device = torch.device("cpu")
    sd = make_sd(cfg, 128, seed=99)
    model = ref_model(rmutils, cfg, sd)
    x = synth.synthetic_inputs(1, 4, 128, seed=5)
    labels = torch.tensor([321.5])
    t0 = time.time(); y = model(x, labels); dt = time.time() - t0
    print(f"[res128] reference forward {dt:.1f}s; std {float(y.std()):.3f}")
    np.savez_compressed(os.path.join(GOLD, "unet_res128.npz"), y_sub=y[:, :, ::8, ::8, ::8].numpy(),
                        y_norm=float(y.double().norm()), y_row=y[0, :, 63, 17, :].numpy(), labels=labels.numpy(),
                        x_seed=5, sd_seed=99)
    del model, sd


# -------------------------------------------------------------------------------------------------
def import_ref_dmtet():
    for name in ("kaolin", "pytorch3d", "pytorch3d.ops", "nvdiffrast", "nvdiffrast.torch", "imageio",
                 "tinycudann", "xatlas"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, os.path.join(REF, "nvdiffrec"))
    # render/ pulls in the CUDA plugin; DMTet.__call__ needs none of it: stub the whole sub-package
    render = types.ModuleType("lib.render")
    for sub in ("mesh", "render", "regularizer", "util", "renderutils"):
        m = types.ModuleType(f"lib.render.{sub}")
        setattr(render, sub, m)
        sys.modules[f"lib.render.{sub}"] = m
    sys.modules["lib.render"] = render
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dmtet", os.path.join(REF, "nvdiffrec/lib/geometry/dmtet.py"),
                                                  submodule_search_locations=None)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = "lib.geometry"
    sys.modules.setdefault("lib", types.ModuleType("lib"))
    sys.modules.setdefault("lib.geometry", types.ModuleType("lib.geometry"))
    spec.loader.exec_module(mod)
    return mod


class _CudaToCpu(torch.overrides.TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device", None) in ("cuda", torch.device("cuda")):
            kwargs["device"] = "cpu"
        return func(*args, **kwargs)


def dmtet_cases(verts, seed=0):
    """Deterministic (pos, sdf) inputs on the 64 tet grid; shared with tests/."""
    g = torch.Generator().manual_seed(seed)
    v = torch.as_tensor(verts, dtype=torch.float32) * 2.1
    deform = (torch.rand(v.shape, generator=g) * 2 - 1) * (2 / 128) * 2.0
    pos = v + deform
    r = v.norm(dim=1)
    cases = {
        "sphere": torch.sign(0.6 - r),
        "noise": torch.sign(torch.randn(v.shape[0], generator=g)),
        "sinus": torch.sign(torch.sin(7 * v[:, 0]) * torch.cos(5 * v[:, 1]) + 0.3 * torch.sin(9 * v[:, 2])),
        "box_zeros": torch.where((v.abs().max(dim=1).values < 0.5), torch.ones(v.shape[0]),
                                 torch.where(v.abs().max(dim=1).values < 0.6, torch.zeros(v.shape[0]),
                                             -torch.ones(v.shape[0]))),
        "smooth": 0.55 - r + 0.1 * torch.sin(11 * v[:, 0]),   # non-unit sdf: exercises the interpolation weights
    }
    return pos, cases


def ref_auto_normals():
    """The reference's `auto_normals` (nvdiffrec/lib/render/mesh.py:200-229) as a callable (verts, faces) -> (v_nrm, f_nrm):
    its function body is compiled from the file where it lies (the module itself imports the renderer) and run against the
    real `util` module (nvdiffrast / imageio stubbed: neither is touched by dot / safe_normalize)."""
    import ast
    import importlib.util
    for name in ("nvdiffrast", "nvdiffrast.torch", "imageio"):
        sys.modules.setdefault(name, types.ModuleType(name))
    spec = importlib.util.spec_from_file_location("ref_render_util", os.path.join(REF, "nvdiffrec/lib/render/util.py"))
    util = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(util)
    tree = ast.parse(open(os.path.join(REF, "nvdiffrec/lib/render/mesh.py")).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "auto_normals"][0]
    ns = {"torch": torch, "util": util, "Mesh": lambda *a, **k: types.SimpleNamespace(**k)}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "mesh.py:auto_normals", "exec"), ns)

    def call(verts, faces):
        with _CudaToCpu():
            m = ns["auto_normals"](types.SimpleNamespace(v_pos=verts, t_pos_idx=faces))
        return m.v_nrm, m.f_nrm
    return call


def gen_dmtet():
    src = os.path.join(REF, "nvdiffrec/data/tets/64_tets_cropped.npz")
    shutil.copyfile(src, os.path.join(GOLD, "64_tets_cropped.npz"))
    tet = np.load(src)
    verts, idx = tet["vertices"], tet["indices"]
    # the grid mask asset equals the tet-vertex occupancy (data/get_tet_mask.py)
    from meshdiffusion_amd.dmtet import grid_mask_from_tets
    ref_mask = torch.load(os.path.join(REF, "data/grid_mask_64.pt"), map_location="cpu").float()
    assert torch.equal(grid_mask_from_tets(verts, 64), ref_mask.view(64, 64, 64)), "grid mask mismatch"
    print("[dmtet] grid_mask_from_tets == data/grid_mask_64.pt ; live cells", int(ref_mask.sum()))
    mod = import_ref_dmtet()
    normals_ref = ref_auto_normals()
    pos, cases = dmtet_cases(verts)
    tets_t = torch.as_tensor(idx, dtype=torch.long)
    out = {}
    with _CudaToCpu():
        dm = mod.DMTet()
        for name, sdf in cases.items():
            t0 = time.time()
            v, f, uvs, uv_idx, ftet, vvi = dm(pos, sdf.clone(), tets_t)
            dt = time.time() - t0
            vo, fo, fto = dmtet_oracle.marching_tets(pos.numpy(), sdf.numpy(), idx)
            assert np.array_equal(fo, f.numpy()), name
            assert np.array_equal(fto, ftet.numpy()), name
            assert np.array_equal(vo, v.numpy()), (name, np.abs(vo - v.numpy()).max())
            print(f"[dmtet] {name}: V={v.shape[0]} F={f.shape[0]} ref {dt:.2f}s  oracle == reference (faces, verts bit-equal)")
            out[f"{name}_counts"] = np.array([v.shape[0], f.shape[0]])
            out[f"{name}_faces_sha"] = sha(f.numpy().astype(np.int64))
            out[f"{name}_verts_sha"] = sha(v.numpy().astype(np.float32))
            out[f"{name}_uv_idx_sha"] = sha(uv_idx.numpy().astype(np.int64))
            out[f"{name}_vvi_sha"] = sha(vvi.numpy().astype(np.int64))
            vn_ref, fn_ref = normals_ref(v, f)
            vn_or, fn_or = dmtet_oracle.auto_normals(v.numpy(), f.numpy())
            # (torch.cross contracts a1*b2 - a2*b1 into an fma on this host, numpy does not: 1 ulp, not bit-equal)
            ok = dmtet_oracle.well_conditioned_normals(v.numpy(), f.numpy())
            assert rel_l2(fn_or, fn_ref) < 1e-6 and np.abs(vn_or - vn_ref.numpy())[ok].max() < 1e-4 and ok.mean() > 0.85, name
            out[f"{name}_vnrm_head"] = vn_ref.numpy()[:256]
            out[f"{name}_vnrm_sum"] = (vn_ref.double() * torch.as_tensor(ok)[:, None]).sum(0).numpy()
            out[f"{name}_fnrm_head"], out[f"{name}_fnrm_sum"] = fn_ref.numpy()[:256], fn_ref.double().sum(0).numpy()
            out[f"{name}_uvs_sha"] = sha(uvs.numpy().astype(np.float32))
            out[f"{name}_ftet_sha"] = sha(ftet.numpy().astype(np.int64))
            out[f"{name}_uvs_shape"] = np.array(uvs.shape)
            out[f"{name}_faces_head"] = f.numpy()[:64]
            out[f"{name}_faces_tail"] = f.numpy()[-64:]
            out[f"{name}_verts_head"] = v.numpy()[:64]
    np.savez_compressed(os.path.join(GOLD, "dmtet.npz"), **out)


def train_step_inputs(B, R, seed):
    """Seeded (batch, labels, noise) of one DDPM training step; the GPU test regenerates them and feeds the same
    tensors to the HIP loss function by patching torch.randint / torch.randn_like exactly as done here."""
    g = torch.Generator().manual_seed(seed)
    mask = synth.synthetic_grid_mask(R).view(1, 1, R, R, R)
    batch = synth.synthetic_inputs(B, 4, R, seed=seed + 1) * mask
    labels = torch.randint(0, 1000, (B,), generator=g)
    noise = torch.randn((B, 4, R, R, R), generator=g)
    return batch, labels, noise, mask


class fixed_draws:
    """Make `torch.randint` / `torch.randn_like` return the given tensors once each (the loss function's draws,
    lib/diffusion/losses.py:59,62), then restore them."""

    def __init__(self, labels, noise):
        self.labels, self.noise = labels, noise

    def __enter__(self):
        self.ri, self.rl = torch.randint, torch.randn_like
        torch.randint = lambda *a, **k: self.labels.to(k.get("device", "cpu"))
        torch.randn_like = lambda t, *a, **k: self.noise.to(t.device)
        return self

    def __exit__(self, *exc):
        torch.randint, torch.randn_like = self.ri, self.rl


def gen_train(full, b2_only=False, trained_only=False):
    """Loss and parameter gradients of the UNMODIFIED reference loss function (lib/diffusion/losses.py:54-85, train
    mode, dropout 0) for the small res64/res128 configs and -- `full` -- the real res64 network at B=1.
    Stored per parameter: gradient norm and a strided sample of <= 256 entries."""
    rsampling, rsde, rmutils = import_reference()
    from lib.diffusion import losses as rlosses
    cases = [("small_res64", synth.small_config(), 8, 1234), ("small_res128", synth.small_config_res128(), 8, 4321)]
    if full:
        from meshdiffusion_amd.config import get_config_res64
        cases.append(("res64", get_config_res64(), 1, 1234))
    if b2_only:
        # train_grads_b2.npz: the real res64 network at B = 2 -- at that batch the 32^3 levels of the HIP path run through the
        # Winograd kernels too (hip_ops.wino_ok), so forward + data-gradient Winograd convs at 64^3 AND 32^3 are pinned to
        # the reference's autograd, not to one block
        from meshdiffusion_amd.config import get_config_res64
        cases = [("res64_b2", get_config_res64(), 2, 1234)]
    if trained_only:
        # train_grads_trained.npz: the real res64 network at B = 1 on the adversarial trained-like weights (the training path runs
        # bf16x3, which has no block scale to upset: this pins that claim to the reference's autograd)
        from meshdiffusion_amd.config import get_config_res64
        cases = [("res64_trained", get_config_res64(), 1, 4321)]
    out = {}
    for name, cfg, B, sd_seed in cases:
        cfg.device = torch.device("cpu")
        cfg.model.dropout = 0.0
        R = cfg.data.image_size
        if name.endswith("_trained"):
            from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
            tmpl = mutils.create_model(ConfigDict(cfg), use_parallel=False).state_dict()
            sd = synth.trained_like_state_dict(tmpl, seed=sd_seed, grid_mask=synth.synthetic_grid_mask(R))
            del tmpl
        else:
            sd = make_sd(cfg, R, seed=sd_seed)
        model = ref_model(rmutils, cfg, sd)
        batch, labels, noise, mask = train_step_inputs(B, R, seed=2024)
        sde = rsde.VPSDE(beta_min=cfg.model.beta_min, beta_max=cfg.model.beta_max, N=cfg.model.num_scales)
        loss_fn = rlosses.get_ddpm_loss_fn(sde, train=True, mask=mask)
        t0 = time.time()
        with fixed_draws(labels, noise):
            loss = loss_fn(model, batch)
        loss.backward()
        print(f"[train {name}] reference loss {float(loss):.6f}  ({time.time() - t0:.1f} s)")
        out[f"{name}_loss"] = np.float64(float(loss))
        out[f"{name}_B"], out[f"{name}_sd_seed"] = np.int64(B), np.int64(sd_seed)
        gsq = 0.0
        for n, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach().flatten()
            stride = max(1, g.numel() // 256)
            out[f"{name}/{n}/norm"] = np.float64(float(g.double().norm()))
            out[f"{name}/{n}/sample"] = g[::stride][:256].numpy().copy()
            gsq += float(g.double().square().sum())
        out[f"{name}_gnorm"] = np.float64(gsq ** 0.5)
    fname = "train_grads_trained.npz" if trained_only else ("train_grads_b2.npz" if b2_only else "train_grads.npz")
    np.savez_compressed(os.path.join(GOLD, fname), **out)


def dataset_inputs(tmp, R=8):
    """Three seeded grids on disk (two at r=R, one smaller than the model grid), a path list and an id
    filter, in the reference's on-disk format (data/tets_to_3dgrid.py:49).  Shared with the CPU test."""
    import json
    g = torch.Generator().manual_seed(99)
    paths = []
    for i, r in enumerate((R, R // 2, R, R)):
        grid = torch.randn((4, r, r, r), generator=g)
        grid[0, :, 1::2] = 0.0                      # exact zeros exercise the sign(0) -> +1 rule
        grid[1:, :, :, ::3] = 0.0                   # empty cells exercise the non-empty test of the augmentation
        path = os.path.join(tmp, f"grid_{i:05d}.pt")
        torch.save(grid, path)
        paths.append(path)
    meta, keep = os.path.join(tmp, "meta.json"), os.path.join(tmp, "keep.json")
    json.dump(paths, open(meta, "w"))
    json.dump([0, 1, 3], open(keep, "w"))
    mask = (torch.rand((1, 1, R, R, R), generator=g) < 0.6).float()
    return meta, keep, mask


def gen_dataset():
    """Reference ShapeNetDMTetDataset items (lib/dataset/shapenet_dmtet_dataset.py) under a fixed global seed."""
    import tempfile
    sys.path.insert(0, REF)
    from lib.dataset.shapenet_dmtet_dataset import ShapeNetDMTetDataset as RefDS
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        meta, keep, mask = dataset_inputs(tmp)
        for tag, kw in (("aug_norm", dict(aug=True, normalize_sdf=True, filter_meta_path=keep)),
                        ("plain", dict(aug=False, normalize_sdf=False, filter_meta_path=None))):
            ds = RefDS(meta, grid_mask=mask, extension="pt", **kw)
            out[f"{tag}_len"] = np.int64(len(ds))
            torch.manual_seed(4321)
            for i in range(len(ds)):
                out[f"{tag}_{i}"] = ds[i].numpy()
    np.savez_compressed(os.path.join(GOLD, "dataset.npz"), **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-res64", action="store_true")
    ap.add_argument("--only", choices=["unet", "dmtet", "dataset", "train", "train_b2", "train_trained", "graded", "ddim", "trained"], default=None)
    ap.add_argument("--graded", default="config1,cond32,res128", help="which graded-size sampler fixtures to (re)generate")
    a = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(os.cpu_count())
    if a.only in (None, "dmtet"):
        gen_dmtet()
    if a.only in (None, "dataset"):
        gen_dataset()
    if a.only in (None, "train"):
        gen_train(full=not a.skip_res64)
    if a.only in (None, "train_b2"):
        gen_train(full=False, b2_only=True)
    if a.only == "train_trained" or (a.only is None and not a.skip_res64):
        gen_train(full=False, trained_only=True)
    if a.only in (None, "unet"):
        gen_unet_and_sampler(a.skip_res64)
    if a.only in (None, "ddim"):
        gen_ddim()
    if a.only == "trained" or (a.only is None and not a.skip_res64):
        gen_trained_like()
    if a.only == "graded" or (a.only is None and not a.skip_res64):
        gen_graded(a.graded.split(","))
    print("golden fixtures written to", GOLD)
