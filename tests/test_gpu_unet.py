"""Layer-, network- and sampler-level parity of the HIP path (through the reference-compatible
Python API) against the oracle and the golden fixtures produced by the imported reference.

Tolerances (rel-L2): one layer / one U-Net evaluation 1e-4 (bf16x3 operands; the budget implied by
BASELINE's 1e-3 end-to-end target is ~5e-5 systematic per evaluation); K-step sampled
grids 1e-3 as stated in BASELINE.json north_star."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, rel_l2

pytestmark = pytest.mark.gpu

TOL_EVAL = 1e-4
TOL_SAMPLE = 1e-3


@pytest.fixture(scope="module")
def env(hip_lib):
    assert torch.cuda.is_available()
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, ddpm_res128, layers, utils as mutils  # noqa: F401
    from oracle import unet_oracle as uo
    return dict(synth=synth, layers=layers, mutils=mutils, uo=uo)


def _small_model(env, dev="cuda"):
    synth, mutils = env["synth"], env["mutils"]
    cfg = synth.small_config(); cfg.device = torch.device(dev)
    model = mutils.create_model(cfg)           # ModelReplica(...).to(device), like the reference
    R = cfg.data.image_size
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    return cfg, model.eval(), sd


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _layer_sd(layer, seed):
    from meshdiffusion_amd import synth
    return synth.sensitised_state_dict(layer.state_dict(), seed=seed)


@pytest.mark.parametrize("cin,cout,S", [(32, 32, 8), (64, 32, 8), (96, 64, 4), (64, 64, 16)])
def test_resnet_block(env, cin, cout, S):
    layers, uo = env["layers"], env["uo"]
    blk = layers.ResnetBlockDDPM(act=torch.nn.SiLU(), in_ch=cin, out_ch=cout, temb_dim=128, dropout=0.0)
    sd = _layer_sd(blk, 3)
    blk.load_state_dict(sd); blk = blk.cuda().eval()
    x, temb = _randn((2, cin, S, S, S), 1), _randn((2, 128), 2)
    with torch.no_grad():
        y = blk(x.cuda(), temb.cuda()).cpu()
        ref = uo.resnet_block(sd, x, temb)
    assert rel_l2(y, ref) < TOL_EVAL


@pytest.mark.parametrize("Cc,S", [(64, 8), (64, 4), (256, 8), (256, 16)])
def test_attn_block(env, Cc, S):
    """(256, 8) and (256, 16) run the fused attention kernel (md_attn_fwd: C = 256, N % 128 == 0; 16^3 is the res64 shape),
    the others the GEMM + softmax path."""
    layers, uo = env["layers"], env["uo"]
    blk = layers.AttnBlock(channels=Cc)
    sd = _layer_sd(blk, 4)
    blk.load_state_dict(sd); blk = blk.cuda().eval()
    x = _randn((2, Cc, S, S, S), 5)
    with torch.no_grad():
        y = blk(x.cuda()).cpu()
        ref = uo.attn_block(sd, x)
    assert rel_l2(y, ref) < TOL_EVAL
    assert rel_l2(y - x, ref - x) < 5e-4      # the attention branch itself, not just the residual
    from meshdiffusion_amd import hip_ops
    if hip_ops.attn_fused_ok(Cc, S ** 3):     # the fused kernel against the GEMM + md_softmax_keys path on the same input
        hip_ops.FUSE_ATTN = False
        try:
            with torch.no_grad():
                y2 = blk(x.cuda()).cpu()
        finally:
            hip_ops.FUSE_ATTN = True
        e = rel_l2(y - x, y2 - x)
        print(f"fused attention vs GEMM+softmax path (C={Cc}, N={S ** 3}): branch rel-L2 {e:.2e}; vs oracle {rel_l2(y - x, ref - x):.2e} / {rel_l2(y2 - x, ref - x):.2e}")
        assert e < 1e-4


def test_up_down_nin(env):
    layers, uo = env["layers"], env["uo"]
    x = _randn((2, 32, 8, 8, 8), 6)
    with torch.no_grad():
        up = layers.Upsample(32, with_conv=True); sd = _layer_sd(up, 7); up.load_state_dict(sd)
        assert rel_l2(up.cuda()(x.cuda()).cpu(), uo.upsample(sd, x)) < TOL_EVAL
        dn = layers.Downsample(32, with_conv=True); sd = _layer_sd(dn, 8); dn.load_state_dict(sd)
        assert rel_l2(dn.cuda()(x.cuda()).cpu(), uo.downsample(sd, x)) < TOL_EVAL
        n = layers.NIN(32, 64); sd = _layer_sd(n, 9); n.load_state_dict(sd)
        assert rel_l2(n.cuda()(x.cuda()).cpu(), uo.nin(x, sd["W"], sd["b"])) < TOL_EVAL


def test_unet_small_vs_golden_and_oracle(env):
    cfg, model, sd = _small_model(env)
    gold = np.load(os.path.join(GOLD, "unet_small.npz"))
    x = env["synth"].synthetic_inputs(2, 4, cfg.data.image_size, seed=int(gold["x_seed"]))
    labels = torch.tensor(gold["labels"])
    with torch.no_grad():
        y = model(x.cuda(), labels.cuda()).cpu()
        y_or = env["uo"].unet_res64_forward(sd, env["synth"].oracle_cfg(cfg), x, labels)
    e_gold, e_or = rel_l2(y, gold["y"]), rel_l2(y, y_or)
    print(f"small U-Net: vs reference golden {e_gold:.3e}, vs oracle {e_or:.3e}")
    assert e_gold < TOL_EVAL and e_or < TOL_EVAL


def test_unet_small_res128_vs_golden_and_oracle(env):
    """ddpm_res128 architecture (5x5x5 stem/head, no coords, 2 level-0 blocks) on a small grid."""
    synth, mutils = env["synth"], env["mutils"]
    cfg = synth.small_config_res128(); cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg).eval()
    R = cfg.data.image_size
    gold = np.load(os.path.join(GOLD, "unet_small_res128.npz"))
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=int(gold["sd_seed"]), grid_mask=synth.synthetic_grid_mask(R))
    assert "coords" not in sd and tuple(sd["all_modules.2.weight"].shape[2:]) == (5, 5, 5)
    model.module.load_state_dict(sd, strict=True)
    x = synth.synthetic_inputs(2, 4, R, seed=int(gold["x_seed"]))
    labels = torch.tensor(gold["labels"])
    with torch.no_grad():
        y = model(x.cuda(), labels.cuda()).cpu()
        y_or = env["uo"].unet_res64_forward(sd, synth.oracle_cfg(cfg), x, labels)
    e_gold, e_or = rel_l2(y, gold["y"]), rel_l2(y, y_or)
    print(f"small res128 U-Net: vs reference golden {e_gold:.3e}, vs oracle {e_or:.3e}")
    assert e_gold < TOL_EVAL and e_or < TOL_EVAL


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "unet_res128.npz")), reason="res128 golden not generated")
def test_unet_res128_full_size_vs_reference_golden(env):
    """BASELINE config #4 shape: ddpm_res128 at 128^3 (390.6 M parameters), one evaluation, against the
    reference's own output (CPU, recorded by oracle/gen_golden.py).  The res128 grid-mask asset is missing
    upstream, so the mask is the synthetic period-4 lattice.  (Measured once against the oracle run with
    PyTorch fp32 ops on the same GPU: 1.8e-5, peak HBM 17.5 GiB.)"""
    from meshdiffusion_amd.config import get_config_res128
    synth, mutils = env["synth"], env["mutils"]
    gold = np.load(os.path.join(GOLD, "unet_res128.npz"))
    cfg = get_config_res128(); cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg).eval()
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=int(gold["sd_seed"]),
                                     grid_mask=synth.synthetic_grid_mask(128))
    model.module.load_state_dict(sd, strict=True)
    del sd
    x = synth.synthetic_inputs(1, 4, 128, seed=int(gold["x_seed"])).cuda()
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        y = model(x, torch.tensor(gold["labels"]).cuda()).cpu()
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    e_sub = rel_l2(y[:, :, ::8, ::8, ::8], gold["y_sub"])
    e_row = rel_l2(y[0, :, 63, 17, :], gold["y_row"])
    e_norm = abs(float(y.double().norm()) - float(gold["y_norm"])) / float(gold["y_norm"])
    print(f"res128 full size vs reference golden: sub {e_sub:.3e} row {e_row:.3e} norm {e_norm:.3e}; peak HBM {peak:.1f} GiB")
    assert e_sub < TOL_EVAL and e_row < TOL_EVAL and e_norm < TOL_EVAL


def test_weight_update_invalidates_packed_cache(env):
    cfg, model, sd = _small_model(env)
    x = env["synth"].synthetic_inputs(1, 4, cfg.data.image_size, seed=3)
    labels = torch.tensor([123.4])
    with torch.no_grad():
        y0 = model(x.cuda(), labels.cuda()).clone()
        for p in model.parameters():
            if p.requires_grad:
                p.mul_(1.01)
        y1 = model(x.cuda(), labels.cuda())
        sd2 = {k: v.cpu() for k, v in model.module.state_dict().items()}
        ref = env["uo"].unet_res64_forward(sd2, env["synth"].oracle_cfg(cfg), x, labels)
    assert rel_l2(y1.cpu(), ref) < TOL_EVAL and rel_l2(y0.cpu(), ref) > 1e-3


def test_sampler_small_uncond_and_cond_vs_reference_golden(env):
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    cfg, model, sd = _small_model(env)
    gold = np.load(os.path.join(GOLD, "sampler_small.npz"))
    R, K = cfg.data.image_size, int(gold["K"])
    synth = env["synth"]
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    shape = (2, 4, R, R, R)

    def cpu_noise(x):   # replay the reference's CPU generator stream on the host, ship to the GPU
        return torch.randn(x.shape).to(x.device)

    mask = synth.synthetic_grid_mask(R).view(1, R, R, R).cuda()
    fn = sampling.get_sampling_fn(cfg, sde, shape, lambda x: x, 1e-3, grid_mask=mask)
    torch.manual_seed(int(gold["uncond_seed"]))
    out, nfe = fn(model, n_iters=K, noise_fn=cpu_noise)
    e = rel_l2(out.cpu(), gold["uncond"])
    print(f"{K}-step uncond sampler vs reference: {e:.3e}")
    assert e < TOL_SAMPLE and nfe == 2000
    assert float((out.cpu() * (1 - mask.cpu())).abs().max()) == 0.0   # masked cells exactly zero

    g = torch.Generator().manual_seed(int(gold["cond_data_seed"]))
    partial = torch.sign(torch.randn((1, 1, R, R, R), generator=g))
    pmask = (torch.rand((1, 1, R, R, R), generator=g) < 0.5).float() * mask.cpu().view(1, 1, R, R, R)
    fn5 = sampling.get_sampling_fn(cfg, sde, shape, lambda x: x, 1e-3, grid_mask=mask.view(1, 1, R, R, R))
    torch.manual_seed(int(gold["cond_seed"]))
    outc, _ = fn5(model, partial=partial.cuda(), partial_mask=pmask.cuda(), freeze_iters=int(gold["freeze_iters"]),
                  n_iters=K, noise_fn=cpu_noise)
    e = rel_l2(outc.cpu(), gold["cond"])
    print(f"{K}-step inpainting sampler vs reference: {e:.3e}")
    assert e < TOL_SAMPLE


def test_fp16x2_mode_small_unet_and_sampler(env):
    """Opt-in fast arithmetic (config.model.hip_precision = "fp16x2"): per evaluation ~1e-3, and the K-step
    sampled grids stay inside BASELINE's 1e-3 (profiles/: 6.9e-5 after the full 999 steps at res64)."""
    from meshdiffusion_amd import hip_ops
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    if hip_ops.FORCE_PRECISION:
        pytest.skip("MD_FORCE_PRECISION overrides the model's fp16x2")
    synth, mutils = env["synth"], env["mutils"]
    cfg = synth.small_config(); cfg.device = torch.device("cuda"); cfg.model.hip_precision = "fp16x2"
    model = mutils.create_model(cfg).eval()
    R = cfg.data.image_size
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    try:
        gold = np.load(os.path.join(GOLD, "unet_small.npz"))
        x = synth.synthetic_inputs(2, 4, R, seed=int(gold["x_seed"]))
        with torch.no_grad():
            y = model(x.cuda(), torch.tensor(gold["labels"]).cuda()).cpu()
        e = rel_l2(y, gold["y"])
        print(f"fp16x2 small U-Net vs reference golden: {e:.3e}")
        assert 2e-5 < e < 3e-3          # really in the fast mode, and within its per-evaluation class
        g2 = np.load(os.path.join(GOLD, "sampler_small.npz"))
        sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
        mask = synth.synthetic_grid_mask(R).view(1, R, R, R).cuda()
        fn = sampling.get_sampling_fn(cfg, sde, (2, 4, R, R, R), lambda t: t, 1e-3, grid_mask=mask)
        torch.manual_seed(int(g2["uncond_seed"]))
        out, _ = fn(model, n_iters=int(g2["K"]), noise_fn=lambda t: torch.randn(t.shape).to(t.device))
        es = rel_l2(out.cpu(), g2["uncond"])
        print(f"fp16x2 {int(g2['K'])}-step sampler vs reference: {es:.3e}")
        assert es < TOL_SAMPLE
    finally:
        assert not hip_ops._SCOPES and hip_ops.PRECISION != "fp16x2"      # the model's scope is left again: nothing to restore


def test_graphed_stepper_matches_eager(env):
    """hipGraph replay of a denoise step == the eager step (same noise stream from the same seed)."""
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    cfg, model, sd = _small_model(env)
    R = cfg.data.image_size
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = env["synth"].synthetic_grid_mask(R).view(1, R, R, R).cuda()
    st = sampling.AncestralStepper(sde, (2, 4, R, R, R), device="cuda", grid_mask=mask)
    model_fn = env["mutils"].get_model_fn(model, train=False)
    x0 = (env["synth"].synthetic_inputs(2, 4, R, seed=3) * mask.cpu()).cuda()
    with torch.no_grad():
        torch.manual_seed(5)
        xe = x0
        for i in range(5):
            xe, xme = st.step(model_fn, xe, i)
        gs = sampling.GraphedStepper(st, model_fn, warmup=1)
        torch.manual_seed(5)
        xg = x0
        for i in range(5):
            xg, xmg = gs.step(xg, i)
    assert gs.graph is not None
    assert rel_l2(xmg.cpu(), xme.cpu()) < 1e-5      # same noise stream; only fp64-atomic order may differ
    assert bool(torch.isfinite(xmg).all()) and float((xmg.cpu() * (1 - mask.cpu())).abs().max()) == 0.0


def _res64_model(env):
    from meshdiffusion_amd.config import get_config_res64
    synth, mutils = env["synth"], env["mutils"]
    cfg = get_config_res64(); cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg)
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(64))
    model.module.load_state_dict(sd, strict=True)
    del sd
    return cfg, model.eval()


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "unet_res64.npz")), reason="res64 golden not generated")
def test_unet_res64_vs_reference_golden(env):
    cfg, model = _res64_model(env)
    gold = np.load(os.path.join(GOLD, "unet_res64.npz"))
    x = env["synth"].synthetic_inputs(1, 4, 64, seed=int(gold["x_seed"]))
    with torch.no_grad():
        y = model(x.cuda(), torch.tensor(gold["labels"]).cuda()).cpu()
    e_sub = rel_l2(y[:, :, ::4, ::4, ::4], gold["y_sub"])
    e_row = rel_l2(y[0, :, 31, 17, :], gold["y_row"])
    e_norm = abs(float(y.double().norm()) - float(gold["y_norm"])) / float(gold["y_norm"])
    print(f"res64 U-Net vs reference golden: sub {e_sub:.3e} row {e_row:.3e} norm {e_norm:.3e}")
    assert e_sub < TOL_EVAL and e_row < TOL_EVAL and e_norm < TOL_EVAL
    # batch consistency at the bench batch size: sample 0 of a B=8 batch == the B=1 result
    xb = torch.cat([x, env["synth"].synthetic_inputs(7, 4, 64, seed=9)], 0)
    lb = torch.cat([torch.tensor(gold["labels"]), torch.linspace(10.0, 990.0, 7)])
    with torch.no_grad():
        yb = model(xb.cuda(), lb.cuda())
    # not bit-equal: the split-K factor of the 4^3/8^3 convs depends on the batch, and fp32 accumulation order
    # over K up to 27648 terms moves results at the 1e-5 level (the same size as the arithmetic's own error)
    assert rel_l2(yb[0:1].cpu(), y) < 5e-5


TOL_EVAL_TRAINED = 6e-5     # per-evaluation budget on the adversarial weights, in the DEFAULT arithmetic


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "unet_res64_trained.npz")), reason="trained-like golden not generated")
def test_unet_res64_trained_like_weights_vs_reference_golden(env):
    """The adversarial gate under the default arithmetic: the real res64 network on synth.trained_like_state_dict (Student-t weights,
    every GroupNorm that feeds a conv / the attention projections scaled per channel by 2^U(-3, 3) and compensated in the consumer --
    a 64x spread of activation magnitudes inside the 16-channel K blocks of the f16f6 format) against the UNMODIFIED reference's
    output on the same weights (oracle/gen_golden.py --only trained), two timesteps.  Without the static equaliser f16f6 measures
    2.5e-4 here on the CPU model (tools/f16f8_numerics.py --unet --weights trained), with it 4.6e-5; f16f8 3.1e-5; bf16x3 1.4e-5.
    The default configuration must stay under 6e-5; the other arithmetics are measured on the same model and printed."""
    from meshdiffusion_amd import hip_ops
    from meshdiffusion_amd.config import get_config_res64
    synth, mutils = env["synth"], env["mutils"]
    gold = np.load(os.path.join(GOLD, "unet_res64_trained.npz"))
    cfg = get_config_res64(); cfg.device = torch.device("cuda")
    default_mode = cfg.model.hip_precision
    model = mutils.create_model(cfg).eval()
    sd = synth.trained_like_state_dict(model.module.state_dict(), seed=int(gold["sd_seed"]), grid_mask=synth.synthetic_grid_mask(64))
    model.module.load_state_dict(sd, strict=True)
    del sd
    x = synth.synthetic_inputs(2, 4, 64, seed=int(gold["x_seed"])).cuda()
    labels = torch.tensor(gold["labels"]).cuda()

    def errors(mode):
        model.module.hip_precision = mode
        hip_ops.PROFILE = []
        try:
            with torch.no_grad():
                y = model(x, labels).cpu()
            tags = [r[5] for r in hip_ops.PROFILE if r[0] == "wino"]
        finally:
            hip_ops.PROFILE = None
        es = []
        for b in range(2):
            e_sub = rel_l2(y[b:b + 1, :, ::4, ::4, ::4], gold["y_sub"][b:b + 1])
            e_row = rel_l2(y[b, :, 31, 17, :], gold["y_row"][b])
            e_norm = abs(float(y[b].double().norm()) - float(gold["y_norm"][b])) / float(gold["y_norm"][b])
            es.append((e_sub, e_row, e_norm))
        return es, tags

    out = {}
    for mode in dict.fromkeys([default_mode, "f16f6", "f16f8", "bf16x3"]):
        if hip_ops.FORCE_PRECISION and mode != default_mode:
            continue
        out[mode], tags = errors(mode)
        fmt = mode[3:] if mode in ("f16f8", "f16f6") else None
        n_fmt = sum(t.endswith("/" + fmt) for t in tags) if fmt else 0
        print(f"res64 U-Net, trained-like weights, {mode}: sub / row / norm per sample {[tuple(f'{v:.2e}' for v in e) for e in out[mode]]}; "
              f"{len(tags)} Winograd conv launches, {n_fmt} of them in the reduced-precision format")
        if fmt and not hip_ops.FORCE_PRECISION:
            assert n_fmt >= 20 and len(tags) - n_fmt <= 3        # all but the Upsample convs (raw residual stream: bf16x3); B = 2: the 64^3 and 32^3 levels
    model.module.hip_precision = default_mode
    for e_sub, e_row, e_norm in out[default_mode]:
        assert e_sub < TOL_EVAL_TRAINED and e_norm < TOL_EVAL_TRAINED and e_row < TOL_EVAL


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "unet_res64_trained.npz")), reason="trained-like golden not generated")
def test_calibrate_measured_equalisers_audit_and_upsample_convs_on_f16f6(env):
    """DDPMUNet3D.calibrate on the adversarial weights: one measured + one audited evaluation per batch.
    After it (a) every Winograd conv -- the three Upsample convs on the raw residual stream included -- runs in the configured reduced
    precision unless the audit demoted it, (b) no conv kept on that path differs from its bf16x3 form by more than the bar, (c) the
    evaluation against the UNMODIFIED reference's golden is no worse than uncalibrated (and under the same 6e-5 budget)."""
    from meshdiffusion_amd import hip_ops
    from meshdiffusion_amd.config import get_config_res64
    synth, mutils = env["synth"], env["mutils"]
    if hip_ops.FORCE_PRECISION:
        pytest.skip("the calibration is about the configured default arithmetic")
    gold = np.load(os.path.join(GOLD, "unet_res64_trained.npz"))
    cfg = get_config_res64(); cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg).eval()
    sd = synth.trained_like_state_dict(model.module.state_dict(), seed=int(gold["sd_seed"]), grid_mask=synth.synthetic_grid_mask(64))
    model.module.load_state_dict(sd, strict=True)
    del sd
    x = synth.synthetic_inputs(2, 4, 64, seed=int(gold["x_seed"])).cuda()
    labels = torch.tensor(gold["labels"]).cuda()

    def evaluate():
        hip_ops.PROFILE = []
        try:
            with torch.no_grad():
                y = model(x, labels).cpu()
            tags = [r[5] for r in hip_ops.PROFILE if r[0] == "wino"]
        finally:
            hip_ops.PROFILE = None
        e = max(max(rel_l2(y[b:b + 1, :, ::4, ::4, ::4], gold["y_sub"][b:b + 1]),
                    abs(float(y[b].double().norm()) - float(gold["y_norm"][b])) / float(gold["y_norm"][b])) for b in range(2))
        return e, tags

    e0, tags0 = evaluate()
    fmt = cfg.model.hip_precision[3:]
    n0 = sum(t.endswith("/" + fmt) for t in tags0)
    # calibration batches: the golden's inputs are NOT among them (a noise batch at three other timesteps)
    xc = synth.synthetic_inputs(2, 4, 64, seed=77).cuda() * synth.synthetic_grid_mask(64).view(1, 1, 64, 64, 64).cuda()
    rep = model.module.calibrate([xc, xc, xc], [torch.full((2,), t, device="cuda") for t in (900.0, 400.0, 60.0)], bar=4e-5)
    e1, tags1 = evaluate()
    n1 = sum(t.endswith("/" + fmt) for t in tags1)
    print(f"calibrate(): {rep['measured']} convs measured, {rep['audited']} launches audited, worst kept {rep['worst']:.2e}, demoted {rep['demoted']}; "
          f"golden error {e0:.2e} -> {e1:.2e}; reduced-precision Winograd launches {n0} -> {n1} of {len(tags1)}")
    assert rep["measured"] >= len(tags0) and rep["worst"] <= 4e-5
    assert n1 == len(tags1) - sum(1 for d in rep["demoted"]) or n1 >= n0          # demoted sites run bf16x3, everything else reduced precision
    assert n1 + len(rep["demoted"]) >= len(tags1)
    assert e1 < TOL_EVAL_TRAINED and e1 < e0 * 1.1


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "unet_res128_trained.npz")), reason="trained-like res128 golden not generated")
def test_unet_res128_full_size_trained_like_weights_vs_reference_golden(env):
    """The same adversarial gate for configs[3]'s network: ddpm_res128 at 128^3 on synth.trained_like_state_dict against the
    unmodified reference's output (oracle/gen_golden.py --only trained), in the config's own arithmetic."""
    from meshdiffusion_amd.config import get_config_res128
    synth, mutils = env["synth"], env["mutils"]
    gold = np.load(os.path.join(GOLD, "unet_res128_trained.npz"))
    cfg = get_config_res128(); cfg.device = torch.device("cuda")
    model = mutils.create_model(cfg).eval()
    sd = synth.trained_like_state_dict(model.module.state_dict(), seed=int(gold["sd_seed"]), grid_mask=synth.synthetic_grid_mask(128))
    model.module.load_state_dict(sd, strict=True)
    del sd
    x = synth.synthetic_inputs(1, 4, 128, seed=int(gold["x_seed"])).cuda()
    with torch.no_grad():
        y = model(x, torch.tensor(gold["labels"]).cuda()).cpu()
    e_sub = rel_l2(y[:, :, ::8, ::8, ::8], gold["y_sub"])
    e_row = rel_l2(y[0, :, 63, 17, :], gold["y_row"])
    e_norm = abs(float(y.double().norm()) - float(gold["y_norm"])) / float(gold["y_norm"])
    print(f"res128 full size, trained-like weights ({cfg.model.hip_precision}) vs reference golden: sub {e_sub:.3e} row {e_row:.3e} norm {e_norm:.3e}")
    assert e_sub < TOL_EVAL and e_row < TOL_EVAL and e_norm < TOL_EVAL


def test_precision_is_a_property_of_the_model_not_of_the_process(env):
    """res64 (f16f8 here, to tell it from the other) and res128 (its config's own hip_precision) evaluated in both
    orders: every model's Winograd launches are the ones ITS config names, whatever ran before; the process default is untouched and
    a training-mode forward inside is bf16x3."""
    from meshdiffusion_amd import hip_ops
    from meshdiffusion_amd.config import get_config_res64, get_config_res128
    synth, mutils = env["synth"], env["mutils"]
    assert get_config_res128().model.hip_precision in ("bf16x3", "f16f8", "f16f6")          # stated, not inherited
    if hip_ops.FORCE_PRECISION:
        pytest.skip("MD_FORCE_PRECISION overrides every model's arithmetic")
    c64 = get_config_res64(); c64.device = torch.device("cuda"); c64.model.hip_precision = "f16f8"
    c128 = get_config_res128(); c128.device = torch.device("cuda")
    fmt128 = c128.model.hip_precision[3:] if c128.model.hip_precision != "bf16x3" else None
    m64, m128 = mutils.create_model(c64).eval(), mutils.create_model(c128).eval()
    x64, x128 = synth.synthetic_inputs(1, 4, 64, seed=1).cuda(), synth.synthetic_inputs(1, 4, 128, seed=2).cuda()
    lab = torch.tensor([400.0]).cuda()
    before = (hip_ops.PRECISION, hip_ops.WINO_F8, hip_ops.DEFAULT_PRECISION)

    def launches(model, x):
        hip_ops.PROFILE = []
        try:
            with torch.no_grad():
                model(x, lab)
            return [r[5] for r in hip_ops.PROFILE if r[0] == "wino"]
        finally:
            hip_ops.PROFILE = None

    def check(tags, fmt):
        kinds = {t.rsplit("/", 1)[1] if t.endswith(("/f8", "/f6")) else "bf16x3" for t in tags}
        n_fmt = sum(t.endswith("/" + fmt) for t in tags) if fmt else 0
        assert kinds <= ({fmt, "bf16x3"} if fmt else {"bf16x3"}), kinds
        assert (n_fmt >= 10 and len(tags) - n_fmt <= 4) if fmt else tags, (fmt, len(tags), n_fmt)     # B = 1: the 64^3 level (+ 128^3 / 64^3 of res128)

    for order in ((m64, m128), (m128, m64), (m64, m128)):
        for m in order:
            check(launches(m, x64 if m is m64 else x128), "f8" if m is m64 else fmt128)
            assert (hip_ops.PRECISION, hip_ops.WINO_F8, hip_ops.DEFAULT_PRECISION) == before and not hip_ops._SCOPES


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLD, "sampler_res64.npz")), reason="res64 golden not generated")
def test_config1_res64_10_steps_vs_reference_golden(env):
    """BASELINE config #1: res64 uncond, B=1, first 10 of the 1000 ancestral steps."""
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    cfg, model = _res64_model(env)
    gold = np.load(os.path.join(GOLD, "sampler_res64.npz"))
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = env["synth"].synthetic_grid_mask(64).view(1, 64, 64, 64).cuda()
    fn = sampling.get_sampling_fn(cfg, sde, (1, 4, 64, 64, 64), lambda x: x, 1e-3, grid_mask=mask)
    torch.manual_seed(int(gold["seed"]))
    out, _ = fn(model, n_iters=int(gold["K"]), noise_fn=lambda x: torch.randn(x.shape).to(x.device))
    out = out.cpu()
    from oracle.gen_golden import sample_stats
    mine = sample_stats(out, env["synth"].synthetic_grid_mask(64), int(gold["stride"]))
    assert float(np.abs(gold["live"]).max()) > 0.1          # every 4th live cell of the lattice: not a blind sample
    e_live = rel_l2(mine["live"], gold["live"])
    e_row = rel_l2(out[0, :, 33, 17, :], gold["xm_row"])
    e_norm = abs(float(out.double().norm()) - float(gold["xm_norm"])) / float(gold["xm_norm"])
    e_sum = float((np.abs(mine["sums"] - gold["sums"]) / mine["l1"]).max())
    print(f"config #1 (res64, 10 steps) vs reference: live cells {e_live:.3e} row {e_row:.3e} norm {e_norm:.3e} sums {e_sum:.3e}")
    assert e_live < TOL_SAMPLE and e_row < TOL_SAMPLE and e_norm < TOL_SAMPLE and e_sum < TOL_SAMPLE


def test_odd_batch_sampler_and_fused_attention(env):
    """A batch that is no multiple of anything (B = 3): 4 ancestral steps of the small model vs the oracle, and the fused
    attention kernel (its sample-to-XCD mapping is `block % batch`) at C = 256, N = 512 vs the oracle."""
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    cfg, model, sd = _small_model(env)
    R, B = cfg.data.image_size, 3
    synth, uo = env["synth"], env["uo"]
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
    st = sampling.AncestralStepper(sde, (B, 4, R, R, R), device="cuda", grid_mask=mask.cuda())
    fn = env["mutils"].get_model_fn(model, train=False)
    x = (synth.synthetic_inputs(B, 4, R, seed=21) * mask).cuda()
    xo = x.cpu()
    g = torch.Generator().manual_seed(22)
    ts = torch.linspace(1.0, 1e-3, 1000)
    with torch.no_grad():
        for i in range(4):
            z = torch.randn((B, 4, R, R, R), generator=g)
            x, xm = st.step(fn, x, i, draw=lambda _t: z.cuda())
            e = uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), xo, torch.ones(B) * ts[i] * 999)
            xo, xmo = uo.ancestral_step(xo, e, z, ts[i], mask)
    assert rel_l2(xm.cpu(), xmo) < TOL_EVAL
    blk = env["layers"].AttnBlock(channels=256)
    sda = _layer_sd(blk, 4)
    blk.load_state_dict(sda); blk = blk.cuda().eval()
    xa = _randn((B, 256, 8, 8, 8), 6)
    with torch.no_grad():
        y = blk(xa.cuda()).cpu()
        ref = uo.attn_block(sda, xa)
    assert rel_l2(y - xa, ref - xa) < 5e-4
