"""bench.py -- denoise steps/sec on 64^3 x 4 DMTet grids (BASELINE.json metric), 1..8 MI355X.

A "step" is one DDPM ancestral denoise step of a batch: one res64 U-Net evaluation (HIP kernels)
plus the fused ancestral update.  Workload = BASELINE.json configs[1]: res64 4-channel grid,
batch 8 per GPU, iterations of the 1000-step sampler; random-init ("sensitised") weights, synthetic
data.  Sampling shards over GPUs as independent sample batches: no data-path collective
("scaling": "weak", batch per GPU fixed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus 8 --steps 20 --warmup 3      # no launcher around it: starts the 8 ranks itself (self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = N * B * K / wall  (sample-steps per second, whole job); the timed
region carries no instrumentation.  Extra objects, all measured AFTER the timed region:
  roofline      dominant kernel (md_conv3_wino: the 3x3x3 convs as Winograd F(2,3) along w): HIP events around every launch of
                a second, untimed pass; its operand pass md_wino_prep priced beside it (`operand_prep`), and the direct 27-tap
                kernel on the same convs in the same process (`direct_build`)
  train_step    BASELINE configs[2] per GPU (res64 training step, batch 8, dropout 0.1) through the trainer's step
                function; with N > 1 the gradients are exchanged by parallel.GradReducer over RCCL (the path's one
                real collective: 1.456 GB of fp32 gradients per step)
  other_configs configs[3] res128 B=2 sampling step, configs[0] res64 B=1, configs[4] cond_gen B=32 + marching tets x 32 (N = 1)
  cpu_baseline  the CPU oracle restatement of the reference on the host cores (rank 0, N = 1 only)
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of one res64 U-Net forward per sample
FLOPS_PER_SAMPLE_STEP = 5.763e12
ACT_BYTES_PER_SAMPLE_STEP = 8.21e9
WEIGHT_BYTES_PER_STEP = 1.456e9
GRAD_BYTES = 1.456e9
FLOPS_PER_SAMPLE_STEP_RES128 = 3.453e13
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "conv_traffic.json")   # PMC-derived HBM bytes per launch, keyed by kernel source


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (configs[1]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--no-calibrate", action="store_true", help="skip the load-time calibration of the reduced-precision convs (the Upsample "
                                                                "convs then stay in bf16x3 and the equalisers are the static ones)")
    ap.add_argument("--precision", default="f16f6", choices=["bf16x3", "fp16x2", "f16f8", "f16f6"],
                    help="arithmetic of the hot conv kernel for the headline number (DESIGN.md section 3): f16f6 = the package default "
                         "(config.model.hip_precision), f16f8 = its e4m3 form, bf16x3 = the round-1..3 arithmetic")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the second measurement in the other arithmetic (bf16x3_mode)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the configs[2] training-step measurement")
    ap.add_argument("--train-steps", type=int, default=3)
    ap.add_argument("--no-res128", action="store_true", help="skip the configs[3] res128 B=2 measurement")
    ap.add_argument("--config", default="res64", choices=["res64", "res128"],
                    help="res128: only the configs[3] measurement (profiling); the headline is always res64")
    ap.add_argument("--weights", default="sensitised", choices=["sensitised", "trained_like"],
                    help="synthetic weights of the headline model: i.i.d. 'sensitised' (default) or the adversarial synth.trained_like_state_dict "
                         "(heavy tails, 2^U(-3,3) GroupNorm gammas): the step time must not depend on it")
    ap.add_argument("--graph", action="store_true", help="replay the denoise step from a captured hipGraph")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only check of the multi-process control flow (gloo, no kernels, fake timing)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` without a launcher around it: re-execute this command line under
    torch.distributed.run with N ranks on this node (one process per GPU, RCCL; gloo for --dry-run) and pass its exit
    code on.  Fails loudly when the node has fewer than N devices: a 1-rank line must never stand in for an N-GPU record."""
    import socket
    import subprocess
    if not a.dry_run:
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < a.gpus:
            raise SystemExit(f"bench.py --gpus {a.gpus}: this node has {have} visible GPU(s); refusing to run fewer ranks")
    with socket.socket() as so:          # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    a = parse()
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if a.steps < 1 or a.warmup < 0:
        raise SystemExit("--steps must be >= 1 and --warmup >= 0: the line reports the time of exactly --steps timed steps")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(a)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {a.gpus} "
                         "(or leave WORLD_SIZE unset and bench.py starts the ranks itself)")
    if a.dry_run:
        return dry_run(a, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        if dist.get_world_size() != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the process group has {dist.get_world_size()} ranks")
        if torch.cuda.device_count() < world:
            raise SystemExit(f"{world} ranks but {torch.cuda.device_count()} visible GPU(s): one process per GPU")

    from meshdiffusion_amd import hip_ops, synth
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401

    if a.config == "res128":
        print(json.dumps({"other_configs": {"res128_b2": res128_step(dev, steps=a.steps, warmup=a.warmup)}}), flush=True)
        return

    cfg = get_config_res64()
    cfg.device = dev
    cfg.model.hip_precision = a.precision
    cfg.eval.batch_size = a.batch
    R, B = cfg.data.image_size, a.batch
    t_setup = time.time()
    model = mutils.create_model(cfg).eval()
    if a.weights == "trained_like":
        sd_cpu = synth.trained_like_state_dict(model.module.state_dict(), seed=4321, grid_mask=synth.synthetic_grid_mask(R))
    else:
        sd_cpu = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd_cpu, strict=True)
    # load-time calibration of the reduced-precision convs, as evaler.uncond_gen / cond_gen run it after restoring a checkpoint
    # (measured equalisers, per-conv audit against bf16x3; untimed: it happens once per weight set): models/utils.calibrate_model
    calibration = None if a.no_calibrate else mutils.calibrate_model(model, cfg, batch=B)      # at the bench batch: the calibration's launches have the timed launches' shapes (kernel-trace averages stay those of the workload)
    if not (rank == 0 and world == 1 and not a.no_cpu_baseline):
        sd_cpu = None
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R).to(dev)
    shape = (B, cfg.data.num_channels, R, R, R)
    stepper = sampling.AncestralStepper(sde, shape, eps=1e-3, device=dev, grid_mask=mask)
    model_fn = mutils.get_model_fn(model, train=False)
    torch.manual_seed(42 + rank)
    t_setup = time.time() - t_setup

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if a.graph:
        gstep = sampling.GraphedStepper(stepper, model_fn, warmup=1)
        a.no_kernel_events = True
        a.warmup = max(a.warmup, 3)        # eager warm-up + capture happen in the untimed steps

        class _G:                          # same call shape as AncestralStepper.step
            @staticmethod
            def step(_fn, x_, i_):
                return gstep.step(x_, i_)
        run = _G
    else:
        run = stepper
    assert hip_ops.PROFILE is None
    with torch.no_grad():
        x = stepper.prior()
        it = 0
        for _ in range(a.warmup):          # untimed: packs weights, warms allocator and caches
            x, _ = run.step(model_fn, x, it); it += 1
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            x, xm = run.step(model_fn, x, it); it += 1
        barrier()
        wall = time.perf_counter() - t0
        assert bool(torch.isfinite(xm).all()), "non-finite samples"
        # ---- second, UNTIMED pass of the same steps with HIP events around every GEMM / conv launch ----
        events = events_unfused = None
        if not a.no_kernel_events:
            hip_ops.PROFILE = []
            t1 = time.perf_counter()
            for _ in range(min(a.steps, 5)):
                x, xm = run.step(model_fn, x, it); it += 1
            torch.cuda.synchronize()
            wall_prof = (time.perf_counter() - t1) / min(a.steps, 5)
            events, hip_ops.PROFILE = hip_ops.PROFILE, None
            # the same convolutions through the DIRECT 27-tap kernel (fused operand loader, no Winograd transform), 2 steps:
            # the r02 mid-round build, measured in the same process on the same box
            if hip_ops.WINO and a.precision in ("bf16x3", "f16f8", "f16f6"):
                hip_ops.WINO = False
                x2, _ = run.step(model_fn, x, it)                     # packs the direct kernel's weight tiles (untimed)
                hip_ops.PROFILE = []
                for _ in range(2):
                    x2, _ = run.step(model_fn, x2, it)
                torch.cuda.synchronize()
                events_unfused, hip_ops.PROFILE = hip_ops.PROFILE, None
                hip_ops.WINO = True

    # ---- second measurement on the same workload, same process: the round-1..3 arithmetic (bf16x3 in the Winograd convs too) ----
    fast = None
    if a.precision in ("f16f8", "f16f6") and not a.no_fast_mode and world == 1 and a.steps > 0:
        model.module.hip_precision = "bf16x3"      # the arithmetic is a property of the model call (hip_ops.precision_scope): nothing global to restore
        try:
            with torch.no_grad():
                xf = stepper.prior()
                for k in range(max(a.warmup, 2)):      # packs the bf16x3 Winograd fragments (untimed)
                    xf, _ = stepper.step(model_fn, xf, k)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for k in range(a.steps):
                    xf, _ = stepper.step(model_fn, xf, a.warmup + k)
                torch.cuda.synchronize()
                wf = time.perf_counter() - t1
        finally:
            model.module.hip_precision = a.precision
        del xf
        fast = {"precision": "bf16x3 everywhere (three bf16 MFMAs per product; config.model.hip_precision = \"bf16x3\"): the headline "
                             "arithmetic of rounds 1-3, measured here in the same process on the same box",
                "value": round(B * a.steps / wf, 3), "unit": "sample-steps/s", "ms_per_step": round(wf / a.steps * 1e3, 3),
                "parity": "per U-Net evaluation 1.2-2.2e-5, 999-step sampled grids 8.4-8.6e-6 rel-L2 vs the fp32 oracle (profiles/r02_*, r03_longrun_*); "
                          "f16f8 / f16f6 (with the static equaliser): 3.4e-5 / 4.9e-5 per evaluation on the trained-like weights vs the reference golden, "
                          "long-run records in profiles/r04_longrun_*, r05_longrun_*"}
    # ---- BASELINE configs[0] on the GPU: res64, batch 1 (single-sample latency), same weights ----
    b1 = None
    if world == 1 and B != 1 and not a.no_res128:
        st1 = sampling.AncestralStepper(sde, (1, cfg.data.num_channels, R, R, R), eps=1e-3, device=dev, grid_mask=mask)
        with torch.no_grad():
            x1 = st1.prior()
            for k in range(3):
                x1, _ = st1.step(model_fn, x1, k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(10):
                x1, xm1 = st1.step(model_fn, x1, 3 + k)
            torch.cuda.synchronize()
            d1 = (time.perf_counter() - t1) / 10
        b1 = {"workload": "BASELINE configs[0] on the GPU: res64 4-ch grid, batch=1, DDPM ancestral sampling steps",
              "dtype": launch_arithmetic(hip_ops, lambda: st1.step(model_fn, x1, 13), model.module.hip_precision),
              "ms_per_step": round(d1 * 1e3, 2), "sample_steps_per_s": round(1.0 / d1, 3), "steps": 10,
              "mfma_frac_step": round(FLOPS_PER_SAMPLE_STEP / d1 / (PEAK_BF16_TFLOPS * 1e12), 4)}
        del st1, x1, xm1

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    # ---- BASELINE configs[4] (N = 1): cond_gen inpainting at batch 32 and the marching-tets launch ----
    cfg4 = None
    if world == 1 and not a.no_res128:
        model.module.hip_precision = a.precision
        cfg4 = {"cond_gen_res64_b32": cond_gen_bench(dev, model, cfg), "marching_tets_b32": marching_tets_bench(dev)}
        torch.cuda.empty_cache()

    # ---- BASELINE configs[2]: the training step of this rank's batch shard (all ranks; RCCL gradient exchange) ----
    train = None
    if not a.no_train_step:
        del stepper, x, xm
        train = train_step_bench(a, cfg, model, rank, world, dev, dist, barrier)
    del model
    torch.cuda.empty_cache()

    if rank == 0:
        sample_steps = world * B * a.steps
        value = sample_steps / wall
        ms_per_step = wall / a.steps * 1e3
        roof = roofline(events, hip_ops, a, B, wall_prof) if events else None
        if roof and events_unfused:
            mu = [(f, s.elapsed_time(e) * 1e-3) for (c, f, s, e, _, _) in events_unfused if c == hip_ops.CFG_C3_128_FAST]
            au = sum(f for f, _ in mu) / sum(t for _, t in mu) / 1e12
            roof["direct_build"] = {"kernel": "md_conv3_main_kernel<0,0,0,1> (direct 27-tap implicit GEMM, GroupNorm affine + SiLU + split "
                                              "in the halo loader): every 3x3x3 conv of the step with MD_WINO=0",
                                    "achieved": round(au, 2), "frac": round(au / PEAK_BF16_TFLOPS, 4),
                                    "avg_launch_ms": round(sum(t for _, t in mu) / len(mu) * 1e3, 4),
                                    "conv_ms_per_step": round(sum(t for _, t in mu) / 2 * 1e3, 2),
                                    "note": "2 instrumented steps right after switching paths (allocator churn: their wall time is not "
                                            "a step time); whole-step A/B of the two paths: profiles/r02_wino_first_bench_{on,off}.json"}
        step_flops = B * FLOPS_PER_SAMPLE_STEP
        step_bytes = B * ACT_BYTES_PER_SAMPLE_STEP + WEIGHT_BYTES_PER_STEP
        whole = {"mfma_frac_step": round(step_flops / (wall / a.steps) / (PEAK_BF16_TFLOPS * 1e12), 4),
                 "hbm_frac_step": round(step_bytes / (wall / a.steps) / (PEAK_HBM_GBS * 1e9), 4),
                 "ms_per_unet_eval_per_sample": round(ms_per_step / B, 3)}
        if roof:     # flat copies: the driver's `parsed` record keeps scalar members of `roofline` only
            roof["whole_step_mfma_frac"], roof["whole_step_hbm_frac"] = whole["mfma_frac_step"], whole["hbm_frac_step"]
            if roof.get("operand_prep"):
                op = roof["operand_prep"]
                roof["operand_prep_ms_per_step"], roof["operand_prep_gbs"] = op["ms_per_step"], op["achieved"]
                roof["operand_prep_hbm_frac"], roof["conv_plus_prep_frac"] = op["frac"], op["conv_plus_prep_frac"]
            if train:
                roof["train_step_ms"] = train.get("ms_per_step")
                roof["train_step_mfma_frac"] = train.get("mfma_frac_step")
        hbm_kernels = hbm_bound_kernels(dev, B) if world == 1 else None
        other = None
        if world == 1 and not a.no_res128:
            other = {"res128_b2": res128_step(dev)}
            if b1 is not None:
                other["res64_b1"] = b1
            if cfg4 is not None:
                other.update(cfg4)
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(sd_cpu, cfg, synth)
        line = {
            "metric": "denoise steps/sec on 64^3x4 DMTet grids (sample-steps/s = n_gpus*batch*steps/wall)",
            "value": round(value, 3), "unit": "sample-steps/s",
            "n_gpus": dist.get_world_size() if dist is not None else 1, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "parity": "B = 8, calibrated model, trained-like weights: 200 ancestral steps vs the fp32 oracle 1.8e-5 (tests/test_gpu_graded.py, driver-run); "
                      "all 999 steps (builder-run) 1.87e-5 (profiles/r06_longrun_999step_b8_f16f6_calibrated_trained_like_vs_oracle.json; uncalibrated, "
                      "round 5: 1.7e-5), 1.9e-5 on the i.i.d. weights (profiles/r04_longrun_*); one U-Net evaluation vs the unmodified reference on the trained-like weights "
                      "4.6-5.4e-5 (tests/golden/unet_res64_trained.npz); DESIGN.md sections 3 and 5 (target 1e-3 rel-L2 on sampled grids)",
            "dtype": {"bf16x3": "bf16x3 (split-bf16 MFMA operands, fp32 accumulate/IO)",
                                            "fp16x2": "fp16x2 (weights split fp16, activations fp16, fp32 accumulate/IO)",
                                            "f16f8": "f16f8 in the Winograd convs behind a GroupNorm (fp16 hi*hi MFMA + e4m3 cross terms in a K-concatenated scaled "
                                                     "fp8 MFMA: 2 matrix-core units per product; operands and weights equalised per input channel by a static "
                                                     "power of two, md_wino_equaliser), bf16x3 elsewhere; fp32 accumulate/IO",
                                            "f16f6": "f16f6 in the Winograd convs behind a GroupNorm (fp16 hi*hi MFMA + MX block-scaled e2m3 cross terms in a "
                                                     "K-concatenated scaled MFMA at twice the e4m3 rate; operands and weights equalised per input channel by a "
                                                     "power of two: md_wino_equaliser, after the load-time calibration from the measured operand statistics -- "
                                                     "which also puts the Upsample convs on the raw residual stream on this path; uncalibrated they stay "
                                                     "bf16x3), bf16x3 elsewhere; fp32 accumulate/IO"}[a.precision],
            "data": f"synthetic (seeded prior noise, {a.weights.replace('_', '-')} random-init res64 weights, synthetic grid mask)",
            "config": {"workload": "BASELINE configs[1]: res64 4-ch grid DDPM ancestral sampling steps, batch=8 per GPU",
                       "batch_per_gpu": B, "grid": [cfg.data.num_channels, R, R, R],
                       "sharding": "independent sample shards per GPU, no data-path collective",
                       "launch": "hipGraph replay" if a.graph else "eager launches through the C ABI"},
            "roofline": roof, "whole_step": whole, "train_step": train, "other_configs": other,
            "hbm_bound_kernels": hbm_kernels, "cpu_baseline": cpu, "bf16x3_mode": fast,
            "setup_s": round(t_setup, 1),
            "calibration": None if calibration is None else {
                "what": "DDPMUNet3D.calibrate at load time (untimed): per-conv operand mean squares measured on noise batches at 3 timesteps, "
                        "equalisers rebuilt from them, every reduced-precision conv audited against bf16x3; bar 4e-5",
                "convs_measured": calibration["measured"], "worst_kept_rel_l2": round(calibration["worst"], 7),
                "demoted_to_bf16x3": [list(d[:2]) for d in calibration["demoted"]]},
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def launch_arithmetic(hip_ops, step, configured):
    """What one more (untimed) call of `step` actually launched: the model's configured arithmetic and the formats of its Winograd
    conv launches counted from the launch tags (f16f8 / f16f6 apply behind a GroupNorm; raw-operand convs and everything outside the
    Winograd path run bf16x3)."""
    assert hip_ops.PROFILE is None
    hip_ops.PROFILE = []
    try:
        with torch.no_grad():
            step()
        torch.cuda.synchronize()
        tags = [r[5] for r in hip_ops.PROFILE if r[0] == "wino"]
    finally:
        hip_ops.PROFILE = None
    n = {"f8": sum(t.endswith("/f8") for t in tags), "f6": sum(t.endswith("/f6") for t in tags)}
    return {"configured": configured, "wino_conv_launches": {"f16f6": n["f6"], "f16f8": n["f8"], "bf16x3": len(tags) - n["f6"] - n["f8"]},
            "elsewhere": "bf16x3"}


def conv_source_key(src="conv3_wino.hip"):
    """Identifies the build of the dominant kernel: sha256 of its source files (the PMC traffic figure in
    profiles/conv_traffic.json was measured on one such build and goes stale when the kernel changes)."""
    h = hashlib.sha256()
    for f in (src, "md_common.h"):
        with open(os.path.join(ROOT, "meshdiffusion_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def roofline(events, hip_ops, a, B, wall_prof):
    """Dominant kernel from the HIP events of the untimed second pass (events on the launch stream)."""
    wino = any(c == "wino" for (c, *_rest) in events)
    dom = "wino" if wino else hip_ops.CFG_C3_128_FAST
    main = [(f, s.elapsed_time(e) * 1e-3, ab) for (c, f, s, e, ab, _) in events if c == dom]
    prep_t = sum(s.elapsed_time(e) * 1e-3 for (c, _, s, e, _, _) in events if c == "wino_prep")
    prep_b = sum(ab for (c, _, _, _, ab, _) in events if c == "wino_prep")
    tot_f, tot_t = sum(f for f, _, _ in main), sum(t for _, t, _ in main)
    alg_bytes = sum(ab for _, _, ab in main) / max(len(main), 1)
    allt = sum(s.elapsed_time(e) * 1e-3 for (_, _, s, e, _, _) in events)
    # per-shape breakdown of every GEMM / conv launch of one step: count, ms per step, algorithmic TFLOP/s
    shapes = {}
    for (c, f, s, e, _, tag) in events:
        k = f"cfg{c}:{tag}"
        v = shapes.setdefault(k, [0, 0.0, 0.0])
        v[0] += 1; v[1] += s.elapsed_time(e); v[2] += f
    ns = min(a.steps, 5)
    per_shape = {k: {"n_per_step": v[0] // ns, "ms_per_step": round(v[1] / ns, 3), "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)}
                 for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])}
    n_steps = min(a.steps, 5)
    ach = tot_f / tot_t / 1e12
    traffic, traffic_src = None, None
    key = conv_source_key("conv3_wino.hip" if wino else "conv3_main.hip")
    try:
        with open(TRAFFIC_FILE) as fh:
            tr = json.load(fh)
        ent = tr.get(key)
        if ent and a.precision == ent.get("precision", "bf16x3") and B == ent.get("batch", 8):
            traffic, traffic_src = ent["hbm_bytes_per_launch"], f"profiles/conv_traffic.json[{key}] <- {ent.get('source')}"
        else:
            traffic_src = f"profiles/conv_traffic.json has no entry for kernel build {key} (batch {B}, {a.precision}): re-profile"
    except OSError:
        traffic_src = "profiles/conv_traffic.json missing"
    fused = bool(hip_ops.FUSE_GN_APPLY and a.precision in ("bf16x3", "f16f8", "f16f6"))
    f8 = wino and a.precision == "f16f8"
    f6 = wino and a.precision == "f16f6"
    if f6:
        kname = ("md_conv3_wino_kernel<0, true, true, RES> (both forms: RES = true with a residual operand, false without; md_conv3_wino_f6: "
                 "3x3x3 conv as Winograd F(2,3) along w, 9 taps x 4 frequencies, one "
                 "frequency per wave; f16f6 arithmetic: per two steps and accumulator tile two v_mfma_f32_32x32x16_f16 + one K-concatenated "
                 "v_mfma_scale_f32_32x32x64_f8f6f4 on MX block-scaled e2m3 cross terms; operand prepared by md_wino_prep_f6)")
    elif f8:
        kname = ("md_conv3_wino_kernel<0, true, false, RES> (md_conv3_wino_f8: 3x3x3 conv as Winograd F(2,3) along w, 9 taps x 4 frequencies, one frequency "
                 "per wave; f16f8 arithmetic: per two steps and accumulator tile two v_mfma_f32_32x32x16_f16 + one K-concatenated "
                 "v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 cross terms); operand prepared by md_wino_prep_f8)")
    elif wino:
        kname = ("md_conv3_wino_kernel<0, false, false, RES> (3x3x3 conv as Winograd F(2,3) along w: 9 taps x 4 frequencies, bf16x3 MFMA, one frequency "
                 "per wave; operand prepared by md_wino_prep)")
    elif fused:
        kname = ("md_conv3_main_kernel<0,0,0,1> (3x3x3 conv, implicit GEMM, bf16x3 MFMA; fp32 operand with GroupNorm "
                 "affine + SiLU + bf16 split applied in the halo loader)")
    else:
        kname = "md_conv3_main_kernel<0,0,0,0> (3x3x3 conv, implicit GEMM, bf16x3 MFMA)"
    return {"bound": "mfma", "kernel": kname,
            "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_BF16_TFLOPS, 4),
            # HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
            # gfx950, + WRITE_SIZE) of the kernel build identified by `kernel_build`
            "traffic": traffic, "traffic_source": traffic_src, "kernel_build": key,
            "algorithmic_bytes_per_launch": round(alg_bytes),
            "launches": len(main), "avg_launch_ms": round(tot_t / max(len(main), 1) * 1e3, 4),
            "kernel_time_share_of_step": round(tot_t / n_steps / wall_prof, 4),
            "all_gemm_conv_time_share_of_step": round(allt / n_steps / wall_prof, 4),
            # what bounds `frac` in practice (measured once per round with s_memtime + s_memrealtime stamps, not by this run): the kernel is
            # power-bound -- matrix-core duty x shader clock is constant across its variants and shapes, so stall removal returns ~1/3
            "power_bound": {"mfma_duty_x_clock_ghz": [1.14, 1.23], "mfma_peak_clock_ghz": 2.4, "shader_clock_ghz_under_kernel": [1.43, 1.69],
                            "evidence": "profiles/r06_wino_epilogue_ab.txt (step 4); DESIGN.md section 8, round 6"},
            # the Winograd path's operand pass (GroupNorm affine + SiLU + input transform + split), HBM-bound, priced separately;
            # `with_prep` = the same algorithmic flops over conv + prep time
            "operand_prep": ({"kernel": "md_wino_prep", "bound": "hbm", "ms_per_step": round(prep_t / n_steps * 1e3, 3),
                              "achieved": round(prep_b / prep_t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(prep_b / prep_t / 1e9 / PEAK_HBM_GBS, 4),
                              "conv_plus_prep_tflops": round(tot_f / (tot_t + prep_t) / 1e12, 2),
                              "conv_plus_prep_frac": round(tot_f / (tot_t + prep_t) / 1e12 / PEAK_BF16_TFLOPS, 4)} if wino and prep_t > 0 else None),
            "per_shape": per_shape,
            "issued_matrix_core_frac": round(ach * (1.0 if f6 else (4.0 / 3.0 if f8 else 2.0)) / PEAK_BF16_TFLOPS, 4) if wino else None,
            "note": "achieved = algorithmic 2*27*Cin*Cout*P flops of the convolution (1x: not the matrix-core units issued per product, "
                    "and not reduced by the Winograd factor 2/3) / HIP-event time of every launch of this kernel in an untimed "
                    "pass right after the timed region (the timed region itself carries no events); the kernel ISSUES "
                    + ("achieved * 2 * 2/3 = 4/3 * achieved of 32-cycle matrix-core units (one fp16 MFMA + half an fp8 K = 64 MFMA per "
                       "product; issued_matrix_core_frac prices them at the bf16 peak)" if f8 else
                       ("achieved * 1.5 * 2/3 = achieved of 32-cycle matrix-core units at the instructions' nominal rates (one fp16 MFMA + half an "
                        "e2m3 K = 64 MFMA of 32 cycles per product; measured in the fp16 mix the e2m3 form costs 1.4 units: "
                        "tools/probes/f6_probe.hip)" if f6 else "achieved * 3 * 2/3 = 2 * achieved of bf16 MFMA work"))}


def train_step_bench(a, cfg, model, rank, world, dev, dist, barrier):
    """BASELINE configs[2] per GPU: res64 training step (forward + masked loss + backward + gradient exchange + clip +
    Adam + EMA), batch 8 per GPU, dropout 0.1, through lib.diffusion.losses.get_step_fn exactly as trainer.py drives it.
    With N > 1 ranks the gradients (1.456 GB fp32) are averaged by parallel.GradReducer: in-place bucket all-reduces
    on the flat gradient buffer, launched from the backward pass as layers finish."""
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import losses, parallel, sde_lib
    from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
    R, B = cfg.data.image_size, a.batch
    cfg.model.hip_precision = "bf16x3"
    model.module.hip_precision = None
    model.train()
    parallel.broadcast_params_(model.parameters())
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    opt = losses.get_optimizer(cfg, model.parameters())
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    mask = synth.synthetic_grid_mask(R).view(1, 1, R, R, R).to(dev)
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), mask=mask)
    g = torch.Generator().manual_seed(100 + rank)
    x0 = torch.sign(torch.randn((B, 1, R, R, R), generator=g))
    batch = (torch.cat([x0, torch.rand((B, 3, R, R, R), generator=g) * 2 - 1], 1) * mask.cpu()).to(dev)
    state = dict(optimizer=opt, model=model, ema=ema, step=1)
    torch.manual_seed(7 + rank)
    step_fn(state, batch)                                   # warm-up: packs dgrad weights, allocates Adam state
    step_fn(state, batch)                                   # second warm-up: the caching allocator has seen the step's pattern
    barrier()
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    losses_seen = [step_fn(state, batch)["loss"] for _ in range(a.train_steps)]
    barrier()
    wall = time.perf_counter() - t0
    state["timers"] = {}                                    # one more step with device syncs between its phases
    step_fn(state, batch)
    split = state.pop("timers")
    wt = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wt, op=dist.ReduceOp.MAX)
    s_per_step = float(wt) / a.train_steps
    ex = state.get("exchange") or {}
    return {"workload": "BASELINE configs[2] per GPU: res64 training step (fwd + loss + bwd + grad exchange + clip + Adam + EMA), "
                        f"batch {B} per GPU, dropout {cfg.model.dropout}",
            "value": round(world * B / s_per_step, 3), "unit": "samples/s", "n_gpus": world, "steps": a.train_steps,
            "ms_per_step": round(s_per_step * 1e3, 2),
            "dtype": "bf16x3 (training forward and weight gradients: hip_ops.precision_scope(training=True), whatever the model's inference "
                     "hip_precision is); the data-gradient convs of the Winograd layers in f16f6 behind a power-of-two lift (hip_ops.DGRAD_F6, "
                     "round 6: 1.7e-5 per conv vs fp64, the 494 gradients vs the reference at unchanged tolerances)",
            "mfma_frac_step": round(3 * FLOPS_PER_SAMPLE_STEP * B / s_per_step / (PEAK_BF16_TFLOPS * 1e12), 4),
            "split_ms": {k: round(v, 2) for k, v in split.items()},
            "exchange": exchange_record(ex, split, world, dist, dev),
            "loss": [round(float(v.detach()), 5) for v in losses_seen],
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), "optimizer": "reference objects (torch.optim.Adam state, EMA shadow params, optimization_manager hyper-parameters) executed by "
                         "md_grad_sqnorm + md_adam_ema_step over flat buffers (MD_FUSED_OPT=0: torch kernels)"}


def exchange_record(ex, split, world, dist, dev, cap_bytes=128 << 20):
    """The `train_step.exchange` object: what the gradient exchange of the instrumented step did on EVERY rank (each rank's own
    bucket count / bytes / exposed wait, gathered to rank 0) and the world size the process group itself reports.  Shared by the
    GPU path (RCCL) and `--dry-run` (gloo, CPU): tests/test_dist_cpu.py checks its shape without a GPU."""
    # what each rank's OWN process sees of the node: the device it computes on (index inside its visible set, the visible count, and
    # the device's PCI bus id as a number -- eight distinct values on a first 8-GPU line prove that RCCL ran over eight devices and
    # not eight ranks on one), HIP_VISIBLE_DEVICES as a hash (-1: unset)
    on_gpu = torch.device(dev).type == "cuda"
    dev_index = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()) if on_gpu else -1
    n_visible = torch.cuda.device_count() if on_gpu else 0
    bus = -1
    if on_gpu:
        try:
            p = torch.cuda.get_device_properties(dev_index)
            bus = (int(getattr(p, "pci_domain_id", 0)) << 16) | (int(getattr(p, "pci_bus_id", -1)) << 8) | int(getattr(p, "pci_device_id", 0))
        except Exception:
            bus = -1
    vis = os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES"))
    vis_code = -1 if vis is None else int.from_bytes(vis.encode()[:6].ljust(6, b"\0"), "little")
    mine = torch.tensor([float(ex.get("buckets", 0)), float(ex.get("bytes", 0)), float(split.get("exchange_exposed", 0.0)),
                         float(split.get("bwd", 0.0)), float(dev_index), float(n_visible), float(bus), float(vis_code)],
                        dtype=torch.float64, device=dev)
    per_rank = [mine]
    if dist is not None and world > 1:
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)

    def _vis(code):
        return None if code < 0 else int(code).to_bytes(6, "little").rstrip(b"\0").decode(errors="replace")

    per_rank = [{"rank": r, "buckets": int(v[0]), "bytes": int(v[1]), "exchange_exposed_ms": round(float(v[2]), 2),
                 "backward_ms": round(float(v[3]), 2), "device_index": int(v[4]), "visible_devices": int(v[5]),
                 "pci_bus_id": int(v[6]), "HIP_VISIBLE_DEVICES": _vis(float(v[7]))} for r, v in enumerate(per_rank)]
    multi = dist is not None and world > 1
    backend = dist.get_backend() if multi else None
    return {"collective": ("RCCL" if backend == "nccl" else str(backend)) + " all-reduce (AVG) of fp32 gradients, in place on the flat gradient buffer"
                          if multi else "none (single rank)",
            "backend": backend, "world_size": dist.get_world_size() if multi else 1,
            "buckets": ex.get("buckets", 0), "bytes": ex.get("bytes", 0), "bucket_cap_bytes": cap_bytes,
            "exposed_ms": round(split.get("exchange_exposed", 0.0), 2), "per_rank": per_rank,
            "distinct_devices": len({(r["pci_bus_id"], r["device_index"], r["HIP_VISIBLE_DEVICES"]) for r in per_rank}) if on_gpu else 0}


def res128_step(dev, steps=3, warmup=2):
    """BASELINE configs[3]: ddpm_res128 at 128^3, batch 2, ancestral sampling steps on one GPU (not the headline)."""
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.config import get_config_res128
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res128, utils as mutils  # noqa: F401
    cfg = get_config_res128(); cfg.device = dev
    R, B = 128, 2
    model = mutils.create_model(cfg).eval()
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=99, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)
    del sd
    calibration = mutils.calibrate_model(model, cfg, batch=B)      # at the measured batch: a level below the Winograd floor at B = 1 would go unmeasured
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    st = sampling.AncestralStepper(sde, (B, 4, R, R, R), device=dev, grid_mask=synth.synthetic_grid_mask(R).view(1, R, R, R).to(dev))
    fn = mutils.get_model_fn(model)
    torch.cuda.reset_peak_memory_stats()
    with torch.no_grad():
        x = st.prior()
        for i in range(warmup):
            x, _ = st.step(fn, x, i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            x, xm = st.step(fn, x, warmup + i)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    assert bool(torch.isfinite(xm).all())
    from meshdiffusion_amd import hip_ops
    out = {"workload": "BASELINE configs[3]: res128 4-ch grid, batch=2, DDPM ancestral sampling steps (synthetic mask)",
           "dtype": launch_arithmetic(hip_ops, lambda: st.step(fn, x, warmup + steps), model.module.hip_precision),
           "ms_per_step": round(dt * 1e3, 2), "sample_steps_per_s": round(B / dt, 3), "steps": steps,
           "mfma_frac_step": round(B * FLOPS_PER_SAMPLE_STEP_RES128 / dt / (PEAK_BF16_TFLOPS * 1e12), 4),
           "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
           "calibration": None if calibration is None else {"convs_measured": calibration["measured"], "worst_kept_rel_l2": round(calibration["worst"], 7),
                                                            "demoted_to_bf16x3": [list(d[:2]) for d in calibration["demoted"]]}}
    del model, st, x, xm
    torch.cuda.empty_cache()
    return out


def kuhn_tet_grid(n):
    """A synthetic tet grid of the reference grid's size class (the real 64-resolution grid is a data asset of the reference:
    159 330 tets / 195 331 edges): an n^3 cube lattice, every cube split into the 6 Kuhn tetrahedra."""
    ax = torch.arange(n + 1)
    X, Y, Z = torch.meshgrid(ax, ax, ax, indexing="ij")
    verts = torch.stack([X, Y, Z], -1).reshape(-1, 3).float() / n * 2 - 1
    vid = lambda x, y, z: (x * (n + 1) + y) * (n + 1) + z      # noqa: E731
    cx, cy, cz = [t.reshape(-1) for t in torch.meshgrid(torch.arange(n), torch.arange(n), torch.arange(n), indexing="ij")]
    import itertools
    tets = []
    for perm in itertools.permutations(range(3)):
        p = [cx, cy, cz]
        corners = [vid(*p)]
        for axis in perm:
            p = [c + (1 if k == axis else 0) for k, c in enumerate(p)]
            corners.append(vid(*p))
        tets.append(torch.stack(corners, -1))
    return verts, torch.cat(tets, 0)


def marching_tets_bench(dev, M=32):
    """BASELINE configs[4], first half: md_marching_tets on M = 32 meshes per launch (deformed tet grid + SDF per mesh)."""
    from meshdiffusion_amd import dmtet
    verts, tets = kuhn_tet_grid(30)
    tables = dmtet.TetTables(tets, dev)
    g = torch.Generator().manual_seed(11)
    N = verts.shape[0]
    pos = (verts[None] + 0.01 * torch.randn((M, N, 3), generator=g)).to(dev)
    r = verts.norm(dim=1)
    sdf = torch.stack([0.55 + 0.01 * m - r + 0.05 * torch.sin(9 * verts[:, 0] + m) for m in range(M)]).to(dev)
    meshes, cnt = dmtet.marching_tets_batch(pos, sdf, tables)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        meshes, cnt = dmtet.marching_tets_batch(pos, sdf, tables)       # no host sync inside: the counts are copied behind the kernels
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    V, F = int(cnt[:, 0].sum()), int(cnt[:, 1].sum())
    T, E = tables.n_tets, tables.n_edges
    alg = M * (N * 16 + T * 40 + E * 8) + V * 12 + F * 32       # per mesh: sdf + pos, the static tables (L2-resident), verts + faces + face->tet out
    return {"workload": f"BASELINE configs[4]: marching tetrahedra, {M} meshes per launch (synthetic Kuhn tet grid: {T} tets, {E} unique "
                        f"edges, {N} vertices; the reference's 64-resolution grid has 159330 / 195331 / 36562)",
            "ms_per_launch": round(dt * 1e3, 3), "meshes_per_s": round(M / dt, 1), "verts_total": V, "faces_total": F,
            "algorithmic_bytes_per_launch": alg, "achieved_gbs": round(alg / dt / 1e9, 1), "hbm_frac": round(alg / dt / 1e9 / PEAK_HBM_GBS, 4),
            "note": "the call + its asynchronous device->host copy of the per-mesh vertex / face counts (pinned buffer + event; the meshes "
                    "are trimmed lazily at their first host access, outside the call)"}


def cond_gen_bench(dev, model, cfg, B=32, iters=3):
    """BASELINE configs[4], second half: the pc sampler with a partial grid (cond_gen inpainting: blend + re-noise every
    iteration), res64, batch 32 on one GPU; a few iterations from the start of the schedule."""
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    R = cfg.data.image_size
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device=dev)
    mask = synth.synthetic_grid_mask(R).to(dev)
    g = torch.Generator().manual_seed(5)
    partial = torch.sign(torch.randn((1, 1, R, R, R), generator=g)).to(dev)
    pmask = (torch.rand((1, 1, R, R, R), generator=g) < 0.5).float().to(dev) * mask.view(1, 1, R, R, R)
    pmask[..., R // 2:] = 0
    fn = sampling.get_sampling_fn(cfg, sde, (B, 4, R, R, R), lambda t: t, 1e-3, grid_mask=mask.view(1, 1, R, R, R))
    torch.cuda.reset_peak_memory_stats()
    fn(model, partial=partial, partial_mask=pmask, freeze_iters=950, n_iters=2)      # untimed: weight fragments, allocator pools (4 GB blocks at this batch)

    def timed(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _ = fn(model, partial=partial, partial_mask=pmask, freeze_iters=950, n_iters=n)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        return time.perf_counter() - t0

    # a call = set-up (the prior is drawn on the CPU like the reference's: 33 M normals at this batch, ~0.1-0.3 s; initial conditioning) + n
    # iterations; the per-iteration time of the 1000-iteration sampler is the DIFFERENCE of two calls, the set-up is reported beside it
    t_short, t_long = timed(1), timed(1 + iters)
    dt = (t_long - t_short) / iters
    setup_s = max(t_short - dt, 0.0)
    return {"workload": f"BASELINE configs[4]: cond_gen partial-grid inpainting sampler (pc, blend + re-noise), res64, batch {B}",
            "dtype": f"{model.module.hip_precision} (the model's config.model.hip_precision; see res64_b1.dtype for the launch formats)",
            "ms_per_iteration": round(dt * 1e3, 1), "sample_steps_per_s": round(B / dt, 2), "iterations": iters,
            "setup_ms_per_call": round(setup_s * 1e3, 1),
            "mfma_frac_step": round(B * FLOPS_PER_SAMPLE_STEP / dt / (PEAK_BF16_TFLOPS * 1e12), 4),
            "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}


def hbm_bound_kernels(dev, B):
    """The step's HBM-bound kernels (GroupNorm statistics / apply: ~10 % of the step) timed on their own, outside the
    timed region, at the dominant shape (128 channels, 64^3, this batch): algorithmic bytes / HIP-event time vs the
    8 TB/s HBM peak."""
    from meshdiffusion_amd import hip_ops as ops
    C_, S_ = 128, 64
    P_ = S_ ** 3
    x = torch.randn((B, C_ // 8, P_, 8), device=dev)
    gamma, beta = torch.ones(C_, device=dev), torch.zeros(C_, device=dev)
    fuse, ops.FUSE_GN_STATS = ops.FUSE_GN_STATS, False
    try:
        prm = ops.gn_params([(x, C_)], gamma, beta, B, P_)
        out = ops.s16b_empty(B, C_, P_, dev)
        n = B * C_ * P_
        res = {}
        for name, fn, nbytes in (("md_gn_stats", lambda: ops.gn_params([(x, C_)], gamma, beta, B, P_), 4 * n),
                                 ("md_gn_apply", lambda: ops.gn_apply([(x, C_)], prm, B, P_, out=out), 8 * n)):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            gbs = nbytes / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9
            res[name] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 3), "algorithmic_bytes_per_launch": nbytes}
        return res
    finally:
        ops.FUSE_GN_STATS = fuse


def dry_run(a, rank, world):
    """Exercise rendezvous, barrier, MAX-reduction and the rank-0 JSON line without a GPU."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
    wall_t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    ranks = 1
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        dist.barrier()
        ranks = dist.get_world_size()        # what the line reports is what the process group holds, not the flag
        if ranks != a.gpus:
            raise SystemExit(f"--gpus {a.gpus} but the process group has {ranks} ranks")
    # the training step's exchange through the REAL reducer (parallel.FlatGrads + GradReducer: in-place prefix buckets) on a
    # stand-in flat buffer, so that the `train_step.exchange` object an N-GPU run prints is exercised without a GPU
    from meshdiffusion_amd.lib.diffusion import parallel
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (300_000, 50_000, 700_000, 1_000)]
    fg = parallel.FlatGrads(ps)
    fg.attach()
    fg.flat.fill_(float(rank + 1))
    cap = 1 << 20
    t0 = time.perf_counter()
    red = parallel.GradReducer(cap_bytes=cap, flat=fg)
    for p in ps[:3]:
        red.ready([p])
    t1 = time.perf_counter()
    red.finish(ps)
    t2 = time.perf_counter()
    want = sum(range(1, world + 1)) / world
    grads_ok = bool(torch.all(fg.flat == want)) if world > 1 else bool(torch.all(fg.flat == 1.0))
    exchange = exchange_record(red.stats, {"exchange_exposed": (t2 - t1) * 1e3, "bwd": (t1 - t0) * 1e3}, world,
                               dist if world > 1 else None, torch.device("cpu"), cap_bytes=cap)
    if rank == 0:
        wall = float(wall_t.item())
        print(json.dumps({"metric": "dry-run", "value": ranks * a.batch * a.steps / wall, "n_gpus": ranks,
                          "steps": a.steps, "warmup": a.warmup, "max_wall": wall,
                          "train_step": {"exchange": exchange, "grads_averaged": grads_ok, "grad_bytes": fg.n * 4}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd_cpu, cfg, synth):
    """The oracle (CPU restatement of the reference, pinned to it by oracle/gen_golden.py) timed on the host cores:
    denoise steps at B=1 (one res64 U-Net evaluation + ancestral update each = 1/8 of one unit batch of the GPU
    workload), one warm-up step then >= 3 timed ones, median."""
    from oracle import unet_oracle as uo
    ocfg = synth.oracle_cfg(cfg)
    R = cfg.data.image_size
    state = {"x": synth.synthetic_inputs(1, 4, R, seed=42), "i": 0}
    z = synth.synthetic_inputs(1, 4, R, seed=43)
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)

    def one_step():
        t0 = time.perf_counter()
        t = torch.tensor(1.0 - state["i"] * 1e-3)
        e = uo.unet_res64_forward(sd_cpu, ocfg, state["x"], torch.ones(1) * t * 999)
        state["x"], _ = uo.ancestral_step(state["x"], e, z, t, mask)
        state["i"] += 1
        return time.perf_counter() - t0

    # Thread sweep: oneDNN convs of this size do not scale to the whole host (r02: 256 threads 98 s per
    # step against 3.5 s at 32), so the baseline is the BEST thread count of a bounded sweep, reported with the sweep.  One
    # warm-up + one timed step per count; the sweep stops once a count is 1.5x slower than the best so far (past the optimum),
    # which also keeps the whole-host count out of the default run when it is hopeless.
    ncpu = os.cpu_count() or 1
    cands = [t for t in (16, 32, 64, 128, 256) if t <= ncpu] or [ncpu]
    if ncpu > cands[-1]:
        cands.append(ncpu)
    sweep, best = {}, None
    with torch.no_grad():
        for t in cands:
            torch.set_num_threads(t)
            warm = one_step()
            s = one_step() if warm < 40.0 else warm      # a very slow count: its warm-up step is its measurement
            sweep[str(t)] = round(s, 2)
            if best is None or s < best[1]:
                best = (t, s)
            elif s > 1.5 * best[1]:
                break
        torch.set_num_threads(best[0])
        timed = [one_step() for _ in range(3)] if best[1] < 20.0 else [best[1]]
    med = statistics.median(timed)
    return {"value": round(1.0 / med, 4), "unit": "sample-steps/s", "cores": best[0], "kind": "port",
            "host_cpu_count": ncpu, "host_cpu_model": cpu_model_name(), "torch_threads": torch.get_num_threads(),
            "s_per_sample_step": round(med, 2), "timed_steps_s": [round(v, 2) for v in timed],
            "thread_sweep_s_per_step": sweep,
            "sample": f"{len(timed)} timed denoise steps (res64 U-Net eval + ancestral update) at batch=1 on the host CPU at the best "
                      "thread count of the sweep in thread_sweep_s_per_step (one warm-up + one timed step per count), median; PyTorch "
                      "fp32 oracle restatement (oracle/unet_oracle.py, pinned to the imported reference by oracle/gen_golden.py; the "
                      "reference itself is absent on the GPU box); a full 999-step run would take ~999x this"}


if __name__ == "__main__":
    main()
