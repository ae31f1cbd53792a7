"""End-to-end drop-in check of the CLI path: `main_diffusion.py --mode=uncond_gen|cond_gen` with a synthetic
reference-format checkpoint (DataParallel 'module.' keys, EMA shadow params), cwd-relative grid mask, .npy out
(reference: main_diffusion.py:19-25, lib/diffusion/evaler.py:14-60,134-211)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _write_ckpt_and_mask(tmp_path, cfg, synth):
    from meshdiffusion_amd.lib.diffusion import losses
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
    from meshdiffusion_amd.lib.diffusion.utils import save_checkpoint
    R = cfg.data.image_size
    model = mutils.create_model(cfg)
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    state = dict(optimizer=losses.get_optimizer(cfg, model.parameters()), model=model, ema=ema, step=12345)
    ck = tmp_path / "ckpt" / "checkpoint.pth"
    os.makedirs(ck.parent)
    save_checkpoint(str(ck), state)
    os.makedirs(tmp_path / "data")
    torch.save(synth.synthetic_grid_mask(R).cuda(), tmp_path / "data" / f"grid_mask_{R}.pt")  # CUDA-saved, like upstream
    return str(ck)


def test_cli_uncond_and_cond_gen(hip_lib, tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import main_diffusion
    from meshdiffusion_amd import config as mdc, synth
    cfg = synth.small_config(); cfg.device = torch.device("cuda")
    cfg.model.num_scales = 40                      # a 40-level schedule keeps the full loop short (39 / 40 iterations)
    ck = _write_ckpt_and_mask(tmp_path, cfg, synth)
    # a reference-style config FILE that goes through the ml_collections shim
    cdir = tmp_path / "configs"; cdir.mkdir()
    (cdir / "small.py").write_text(
        "from meshdiffusion_amd import synth\n\ndef get_config():\n    c = synth.small_config()\n"
        "    c.model.num_scales = 40\n    return c\n")
    monkeypatch.chdir(tmp_path)                    # the mask path is cwd-relative in the reference
    out = tmp_path / "out"
    torch.manual_seed(0)
    main_diffusion.main(["--config", str(cdir / "small.py"), "--mode=uncond_gen", f"--config.eval.eval_dir={out}",
                         f"--config.eval.ckpt_path={ck}", "--config.eval.batch_size=2"])
    x = np.load(out / "0.npy")
    R = cfg.data.image_size
    assert x.shape == (2, 4, R, R, R) and x.dtype == np.float32 and np.isfinite(x).all()
    m = synth.synthetic_grid_mask(R).numpy()
    assert np.abs(x * (1 - m)).max() == 0.0 and np.abs(x).max() > 0

    # cond_gen: synthetic partial DMTet on a tet grid whose vertices are the live lattice cells
    idx = np.argwhere(m > 0).astype(np.float32)
    verts = (idx / (R - 1) - 0.5).astype(np.float32)          # regular spacing -> integer grid coordinates
    tet_path = tmp_path / "tets.npz"
    np.savez(tet_path, vertices=verts, indices=np.zeros((1, 4), np.int32))
    g = torch.Generator().manual_seed(1)
    part = {"sdf": torch.sign(torch.randn(len(verts), generator=g)), "vis": torch.rand(len(verts), generator=g) < 0.5}
    ppath = tmp_path / "dmtet.pt"
    torch.save(part, ppath)
    main_diffusion.main(["--config", str(cdir / "small.py"), "--mode=cond_gen", f"--config.eval.eval_dir={out}",
                         f"--config.eval.ckpt_path={ck}", "--config.eval.batch_size=2",
                         f"--config.eval.partial_dmtet_path={ppath}", f"--config.eval.tet_path={tet_path}",
                         "--config.eval.freeze_iters=30"])
    xc = np.load(out / "0.npy")
    assert xc.shape == (2, 4, R, R, R) and np.isfinite(xc).all() and np.abs(xc * (1 - m)).max() == 0.0


def test_cli_train_mode_runs_checkpoints_and_resumes(hip_lib, tmp_path, monkeypatch):
    """`main_diffusion.py --mode=train` (reference trainer.py:18-137) on a small U-Net and a synthetic on-disk
    dataset: steps through the HIP forward/backward, logs finite losses, writes the reference-format
    checkpoints, and a second invocation resumes from checkpoints-meta instead of restarting."""
    import json
    sys.path.insert(0, ROOT)
    import main_diffusion
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import trainer
    R = 16
    (tmp_path / "data").mkdir()
    gm = synth.synthetic_grid_mask(R)
    torch.save(gm, tmp_path / "data" / f"grid_mask_{R}.pt")
    g = torch.Generator().manual_seed(5)
    paths = []
    for i in range(24):
        grid = torch.cat([torch.sign(torch.randn((1, R, R, R), generator=g)), torch.rand((3, R, R, R), generator=g) - 0.5])
        p = tmp_path / f"grid_{i:05d}.pt"
        torch.save(grid * gm, p)
        paths.append(str(p))
    json.dump(paths, open(tmp_path / "meta.json", "w"))
    json.dump(list(range(20)), open(tmp_path / "keep.json", "w"))
    cdir = tmp_path / "configs"; cdir.mkdir()
    (cdir / "small.py").write_text(
        "from meshdiffusion_amd import synth\n\ndef get_config():\n    c = synth.small_config()\n    return c\n")
    monkeypatch.chdir(tmp_path)
    wd = tmp_path / "run"
    common = ["--config", str(cdir / "small.py"), "--mode=train", f"--config.training.train_dir={wd}",
              f"--config.data.meta_path={tmp_path / 'meta.json'}", f"--config.data.filter_meta_path={tmp_path / 'keep.json'}",
              "--config.training.batch_size=8", "--config.data.num_workers=0", "--config.training.log_freq=1",
              "--config.training.snapshot_freq_for_preemption=2", "--config.training.snapshot_freq=100",
              "--config.optim.warmup=2", "--config.optim.lr=1e-3"]
    torch.manual_seed(0)
    main_diffusion.main(common + ["--config.training.n_iters=3"])
    ck = torch.load(wd / "checkpoints" / "checkpoint_3.pth", weights_only=False)
    assert set(ck) == {"optimizer", "model", "ema", "step"} and ck["step"] == 4       # steps 0..3 inclusive
    assert all(k.startswith("module.") for k in ck["model"])
    meta = torch.load(wd / "checkpoints-meta" / "checkpoint.pth", weights_only=False)
    assert meta["step"] == 3                                                           # saved after step index 2
    w0 = ck["model"]["module.all_modules.2.weight"].clone()
    # resume: starts at step 3 (from the meta checkpoint), runs up to 5
    seen = []
    real = trainer.losses.get_step_fn

    def spy(*a, **k):
        fn = real(*a, **k)

        def wrapped(state, batch, **kw):
            seen.append(int(state["step"]))
            out = fn(state, batch, **kw)
            assert torch.isfinite(out["loss"]).all()
            return out
        return wrapped
    monkeypatch.setattr(trainer.losses, "get_step_fn", spy)
    main_diffusion.main(common + ["--config.training.n_iters=5"])
    assert seen == [3, 4, 5]
    ck2 = torch.load(wd / "checkpoints" / "checkpoint_5.pth", weights_only=False)
    assert ck2["step"] == 6 and not torch.equal(ck2["model"]["module.all_modules.2.weight"], w0)
    assert ck2["ema"]["num_updates"] == 6


def test_bench_contract_on_gpu(hip_lib):
    """`python bench.py` (one GPU): exactly one JSON line with the driver's keys, the roofline of the dominant kernel
    measured from in-region HIP events, and the HBM-bound kernel figures (short run: no CPU baseline / fast mode)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--no-fast-mode"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - 8 * 2 / (d["ms_per_step"] * 2e-3)) / d["value"] < 1e-3
    r = d["roofline"]
    # algorithmic flops; the Winograd kernel issues 2 x that in bf16 MFMAs (3 products, 2/3 of the multiplications): ceiling 1/2
    assert r["bound"] == "mfma" and 0.05 < r["frac"] < 0.5 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert "md_conv3_wino" in r["kernel"]
    assert r["launches"] == 2 * 45            # 45 of the 60 3x3x3 stride-1 convs of a res64 evaluation (the 8^3 level stays direct)
    assert r["operand_prep"]["kernel"] == "md_wino_prep" and 0.3 < r["operand_prep"]["frac"] < 1.0
    assert r["direct_build"]["frac"] < r["frac"]
    h = d["hbm_bound_kernels"]
    assert h["md_gn_apply"]["bound"] == "hbm" and 0.3 < h["md_gn_apply"]["frac"] < 1.0
