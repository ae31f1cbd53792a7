"""DDIM sampler (reference sde_lib.py:113-140 `discretize_ddim`, sampling.py:500-570) on the HIP path:
md_ddim_step against the unmodified reference's update (float64; bit-exact against the same op chain on the same host), the whole 99-evaluation sampler against the
unmodified reference run (noise_removal=False: the only setting upstream can execute) and against the oracle for the
repaired noise_removal=True path, and `--config.sampling.method=ddim` from the CLI."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import GOLD, ROOT, rel_l2

pytestmark = pytest.mark.gpu


def _step_inputs(seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((2, 4, 4, 4, 4), generator=g)
    eps = torch.randn((2, 4, 4, 4, 4), generator=g)
    return x, eps


def test_ddim_step_vs_reference_golden(hip_lib):
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    gold = np.load(os.path.join(GOLD, "ddim.npz"))
    sde = sde_lib.VPSDE(0.1, 20.0, 1000, device="cuda")
    ts = sampling.ddim_schedule(1000)
    assert np.array_equal((ts * 1000).round().long().numpy(), gold["seq"])
    pred = sampling.DDIMPredictor(sde, None)
    x, eps = _step_inputs(int(gold["step_seed"]))
    for n in range(3):
        i = int(gold[f"step{n}_i"])
        vt, vp = (torch.ones(2) * ts[i]).cuda(), (torch.ones(2) * ts[i - 1]).cuda()
        xin = x.double() if n == 0 else x.double() * 0.7
        coef = pred.coefficients(vt, vp)
        xn, x0p, x32 = ops.ddim_step(xin.cuda(), eps.cuda(), None, coef)
        # (a) bit for bit (float64) against the reference's op chain evaluated with torch on this host from the same
        #     table entries (sde_lib.py:129-139)
        a1, a2, r1, r2 = [coef.cpu()[:, j][:, None, None, None, None] for j in range(4)]
        x0s = xin - a2 * eps.double()
        sst = xin - x0s
        assert torch.equal(xn.cpu(), r1 * xin + ((-r1) + r2) * sst), n
        assert torch.equal(x0p.cpu(), x0s / a1), n
        assert torch.equal(x32.cpu(), xn.cpu().float())
        # (b) against the unmodified reference's recorded output.  Not bitwise: the float32 VP tables come from
        #     torch.linspace, whose vectorised CPU kernel rounds a few entries differently from host to host (1e-7)
        assert rel_l2(xn.cpu(), gold[f"step{n}_x_new"]) < 2e-6 and rel_l2(x0p.cpu(), gold[f"step{n}_x0_pred"]) < 2e-6, n
    # mask and inpainting blend (sampling.py:560-564) against the same ops in torch
    P = 64
    mask = (torch.rand(P, generator=torch.Generator().manual_seed(1)) < 0.5).float()
    part = torch.sign(torch.randn(P, generator=torch.Generator().manual_seed(2)))
    pm = (torch.rand(P, generator=torch.Generator().manual_seed(3)) < 0.3).float()
    vt, vp = (torch.ones(2) * ts[40]).cuda(), (torch.ones(2) * ts[39]).cuda()
    xn, x0p, _ = ops.ddim_step(x.double().cuda(), eps.cuda(), mask.cuda(), pred.coefficients(vt, vp), part.cuda(), pm.cuda(), 1)
    r_xn, r_x0, _ = ops.ddim_step(x.double().cuda(), eps.cuda(), None, pred.coefficients(vt, vp))
    r_xn, r_x0 = r_xn.cpu() * mask.view(4, 4, 4), r_x0.cpu() * mask.view(4, 4, 4)
    for r in (r_xn, r_x0):
        r[:, 1] = r[:, 1] * (1 - pm.view(4, 4, 4)) + part.view(4, 4, 4) * pm.view(4, 4, 4)
    assert torch.equal(xn.cpu(), r_xn) and torch.equal(x0p.cpu(), r_x0)


def _small(method, noise_removal):
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    cfg = synth.small_config(); cfg.device = torch.device("cuda")
    cfg.sampling.method, cfg.sampling.noise_removal = method, noise_removal
    model = mutils.create_model(cfg).eval()
    R = cfg.data.image_size
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    return cfg, model, sd


def test_ddim_sampler_small_vs_reference_and_oracle(hip_lib):
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from oracle import unet_oracle as uo
    gold = np.load(os.path.join(GOLD, "ddim.npz"))
    cfg, model, sd = _small("ddim", False)
    R = cfg.data.image_size
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
    fn = sampling.get_sampling_fn(cfg, sde, (2, 4, R, R, R), lambda v: v, 1e-3, grid_mask=mask.cuda())
    torch.manual_seed(int(gold["sampler_seed"]))
    out, nfe = fn(model)                                   # 99 U-Net evaluations on the quadratic schedule
    assert out.dtype == torch.float64 and nfe == 2000
    e = rel_l2(out.cpu(), gold["sampler_small"])
    print(f"DDIM sampler (99 evaluations, noise_removal=False) vs the unmodified reference: {e:.3e}")
    assert e < 1e-3
    assert float((out.cpu() * (1 - mask)).abs().max()) == 0.0
    # repaired noise_removal=True path (upstream: NameError 'encode'): last x0 prediction, vs the oracle restatement
    cfg.sampling.noise_removal = True
    fn = sampling.get_sampling_fn(cfg, sde, (2, 4, R, R, R), lambda v: v, 1e-3, grid_mask=mask.cuda())
    torch.manual_seed(7)
    x_init = torch.randn(2, 4, R, R, R)
    out, _ = fn(model, x0=x_init.cuda(), n_iters=12)
    with torch.no_grad():
        ref = uo.ddim_sample(lambda xx, lb: uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), xx, lb), x_init, mask, 1000,
                             denoise=True, n_iters=12)
    e = rel_l2(out.cpu(), ref)
    print(f"DDIM sampler (12 evaluations, noise_removal=True -> x0 prediction) vs oracle: {e:.3e}")
    assert e < 1e-3


def test_cli_ddim_uncond_gen(hip_lib, tmp_path, monkeypatch):
    """`main_diffusion.py --mode=uncond_gen --config.sampling.method=ddim` (evaler.py:44-58 with the 'ddim' branch of
    get_sampling_fn, sampling.py:118-128) writes the .npy like the pc sampler does."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import main_diffusion
    from meshdiffusion_amd import synth
    from test_gpu_cli import _write_ckpt_and_mask
    cfg = synth.small_config(); cfg.device = torch.device("cuda")
    ck = _write_ckpt_and_mask(tmp_path, cfg, synth)
    cdir = tmp_path / "configs"; cdir.mkdir()
    (cdir / "small.py").write_text("from meshdiffusion_amd import synth\n\ndef get_config():\n    return synth.small_config()\n")
    monkeypatch.chdir(tmp_path)
    out = tmp_path / "out"
    torch.manual_seed(0)
    main_diffusion.main(["--config", str(cdir / "small.py"), "--mode=uncond_gen", f"--config.eval.eval_dir={out}",
                         f"--config.eval.ckpt_path={ck}", "--config.eval.batch_size=2", "--config.sampling.method=ddim"])
    x = np.load(out / "0.npy")
    R = cfg.data.image_size
    m = synth.synthetic_grid_mask(R).numpy()
    assert x.shape == (2, 4, R, R, R) and np.isfinite(x).all() and np.abs(x * (1 - m)).max() == 0.0 and np.abs(x).max() > 0


def test_ddim_res64_batch8_10_evaluations_vs_oracle_fp32_on_gpu(hip_lib):
    """the DDIM sampler at the graded size -- the real res64 network, B = 8, the first K = 10
    evaluations of the 100-point quadratic schedule (sampling.py:522-569) -- against the oracle restatement whose U-Net
    runs with PyTorch fp32 ops on the same GPU (the float64 update itself on the host, as the oracle defines it)."""
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from oracle import unet_oracle as uo
    cfg = get_config_res64(); cfg.device = torch.device("cuda")
    cfg.sampling.method = "ddim"
    B, K, R = 8, 10, 64
    model = mutils.create_model(cfg).eval()
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    sd_gpu = {k: v.cuda() for k, v in sd.items()}
    del sd
    ocfg = synth.oracle_cfg(cfg)
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
    x_init = torch.randn((B, 4, R, R, R), generator=torch.Generator().manual_seed(42))
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False

    def eps_oracle(x, labels):     # one sample at a time (the fp32 torch ops at B = 8 would need 8x the workspace)
        with torch.no_grad():
            return torch.cat([uo.unet_res64_forward(sd_gpu, ocfg, x[b:b + 1].cuda(), labels[b:b + 1].cuda()) for b in range(B)]).cpu()

    for denoise in (False, True):
        cfg.sampling.noise_removal = denoise
        fn = sampling.get_sampling_fn(cfg, sde, (B, 4, R, R, R), lambda v: v, 1e-3, grid_mask=mask.cuda())
        out, _ = fn(model, x0=x_init.cuda(), n_iters=K)
        ref = uo.ddim_sample(eps_oracle, x_init, mask, 1000, denoise=denoise, n_iters=K)
        e = rel_l2(out.cpu(), ref)
        per = max(rel_l2(out[b].cpu(), ref[b]) for b in range(B))
        print(f"DDIM res64 B=8, {K} evaluations, noise_removal={denoise}: vs fp32 oracle {e:.3e} (worst sample {per:.3e})")
        assert out.dtype == torch.float64 and e < 1e-4 and per < 1e-4
        assert float((out.cpu() * (1 - mask)).abs().max()) == 0.0
