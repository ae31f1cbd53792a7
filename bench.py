"""bench.py -- denoise steps/sec on 64^3 x 4 DMTet grids (BASELINE.json metric), 1..8 MI355X.

A "step" is one DDPM ancestral denoise step of a batch: one res64 U-Net evaluation (HIP kernels)
plus the fused ancestral update.  Workload = BASELINE.json configs[1]: res64 4-channel grid,
batch 8 per GPU, iterations of the 1000-step sampler; random-init ("sensitised") weights, synthetic
data.  Sampling shards over GPUs as independent sample batches: no data-path collective
("scaling": "weak", batch per GPU fixed).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = N * B * K / wall  (sample-steps per second, whole job).
Extra objects: `roofline` (dominant kernel = the 3x3x3 implicit-GEMM conv, HIP-event timed inside
the timed region) and `cpu_baseline` (the CPU oracle restatement of the reference on the host cores,
rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work of one res64 U-Net forward per sample
FLOPS_PER_SAMPLE_STEP = 5.763e12
ACT_BYTES_PER_SAMPLE_STEP = 8.21e9
WEIGHT_BYTES_PER_STEP = 1.456e9
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
PMC_TRAFFIC_BYTES_PER_LAUNCH = 1.480e9   # measured: (2*FETCH_SIZE + WRITE_SIZE) KiB per md_conv3_main_kernel launch


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (configs[1]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "fp16x2"],
                    help="arithmetic of the hot conv kernel for the headline number (DESIGN.md section 3)")
    ap.add_argument("--no-fast-mode", action="store_true", help="skip the extra fp16x2 measurement")
    ap.add_argument("--graph", action="store_true", help="replay the denoise step from a captured hipGraph")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU-only check of the multi-process control flow (gloo, no kernels, fake timing)")
    return ap.parse_args()


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
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

    from meshdiffusion_amd import hip_ops, synth
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion import sampling, sde_lib
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401

    cfg = get_config_res64()
    cfg.device = dev
    cfg.model.hip_precision = a.precision
    cfg.eval.batch_size = a.batch
    R, B = cfg.data.image_size, a.batch
    t_setup = time.time()
    model = mutils.create_model(cfg).eval()
    sd_cpu = synth.sensitised_state_dict(model.module.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd_cpu, strict=True)
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
    with torch.no_grad():
        x = stepper.prior()
        it = 0
        for _ in range(a.warmup):          # untimed: packs weights, warms allocator and caches
            x, _ = run.step(model_fn, x, it); it += 1
        if not a.no_kernel_events:
            hip_ops.PROFILE = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            x, xm = run.step(model_fn, x, it); it += 1
        barrier()
        wall = time.perf_counter() - t0
    events, hip_ops.PROFILE = hip_ops.PROFILE, None
    assert bool(torch.isfinite(xm).all()), "non-finite samples"

    # ---- optional second measurement: the opt-in fp16x2 arithmetic on the same workload ----
    fast = None
    if a.precision == "bf16x3" and not a.no_fast_mode and world == 1:
        model.module.hip_precision = "fp16x2"
        with torch.no_grad():
            xf = stepper.prior()
            for k in range(max(a.warmup, 1)):
                xf, _ = stepper.step(model_fn, xf, k)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for k in range(a.steps):
                xf, xmf = stepper.step(model_fn, xf, a.warmup + k)
            torch.cuda.synchronize()
            wf = time.perf_counter() - t1
        model.module.hip_precision = a.precision
        hip_ops.set_precision(a.precision)
        fast = {"precision": "fp16x2 (weights split fp16, activations fp16, 2 MFMAs/product; opt-in: "
                             "config.model.hip_precision)", "value": round(B * a.steps / wf, 3),
                "unit": "sample-steps/s", "ms_per_step": round(wf / a.steps * 1e3, 3),
                "parity": "999-step sampled grids 6.9e-5 rel-L2 vs fp32 (profiles/r01_longrun_999step_act_fp16_experiment.json); "
                          "~1e-3 per U-Net evaluation"}

    wall_t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
    wall = float(wall_t.item())

    if rank == 0:
        sample_steps = world * B * a.steps
        value = sample_steps / wall
        ms_per_step = wall / a.steps * 1e3
        # ---- roofline of the dominant kernel from in-region HIP events ----
        roof = None
        if events:
            main = [(f, s.elapsed_time(e) * 1e-3, ab) for (c, f, s, e, ab) in events if c == hip_ops.CFG_C3_128_FAST]
            tot_f, tot_t = sum(f for f, _, _ in main), sum(t for _, t, _ in main)
            alg_bytes = sum(ab for _, _, ab in main) / max(len(main), 1)
            allt = sum(s.elapsed_time(e) * 1e-3 for (_, _, s, e, _) in events)
            ach = tot_f / tot_t / 1e12
            roof = {"bound": "mfma", "kernel": "md_conv3_main_kernel<0> (3x3x3 conv, implicit GEMM, bf16x3 MFMA)",
                    "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4),
                    # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled as
                    # MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE): profiles/r01_final_pmc_*.summary.txt
                    "traffic": PMC_TRAFFIC_BYTES_PER_LAUNCH if a.precision == "bf16x3" and B == 8 else None,
                    "algorithmic_bytes_per_launch": round(alg_bytes),
                    "launches": len(main), "avg_launch_ms": round(tot_t / max(len(main), 1) * 1e3, 4),
                    "kernel_time_share_of_step": round(tot_t / wall, 4),
                    "all_gemm_conv_time_share_of_step": round(allt / wall, 4),
                    "note": "achieved = algorithmic 2*M*N*K flops (1x, not the 3 bf16 MFMAs issued per product) "
                            "/ HIP-event time of every launch of this kernel inside the timed region; "
                            "ceiling of the bf16x3 scheme is 1/3 of peak"}
        step_flops = B * FLOPS_PER_SAMPLE_STEP
        step_bytes = B * ACT_BYTES_PER_SAMPLE_STEP + WEIGHT_BYTES_PER_STEP
        whole = {"mfma_frac_step": round(step_flops / (wall / a.steps) / (PEAK_BF16_TFLOPS * 1e12), 4),
                 "hbm_frac_step": round(step_bytes / (wall / a.steps) / (PEAK_HBM_GBS * 1e9), 4),
                 "ms_per_unet_eval_per_sample": round(ms_per_step / B, 3)}
        cpu = None
        if world == 1 and not a.no_cpu_baseline:
            cpu = cpu_baseline(sd_cpu, cfg, synth)
        hbm_kernels = hbm_bound_kernels(dev, B) if world == 1 else None
        line = {
            "metric": "denoise steps/sec on 64^3x4 DMTet grids (sample-steps/s = n_gpus*batch*steps/wall)",
            "value": round(value, 3), "unit": "sample-steps/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": ("bf16x3 (split-bf16 MFMA operands, fp32 accumulate/IO)" if a.precision == "bf16x3" else
                                           "fp16x2 (weights split fp16, activations fp16, fp32 accumulate/IO)"),
            "data": "synthetic (seeded prior noise, sensitised random-init res64 weights, synthetic grid mask)",
            "config": {"workload": "BASELINE configs[1]: res64 4-ch grid DDPM ancestral sampling steps, batch=8 per GPU",
                       "batch_per_gpu": B, "grid": [cfg.data.num_channels, R, R, R],
                       "sharding": "independent sample shards per GPU, no data-path collective",
                       "launch": "hipGraph replay" if a.graph else "eager launches through the C ABI"},
            "roofline": roof, "whole_step": whole, "hbm_bound_kernels": hbm_kernels, "cpu_baseline": cpu, "fast_mode": fast,
            "setup_s": round(t_setup, 1),
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def hbm_bound_kernels(dev, B):
    """The step's HBM-bound kernels (GroupNorm statistics / apply: ~10 % of the step) timed on their own, outside the
    timed region, at the dominant shape (128 channels, 64^3, this batch): algorithmic bytes / HIP-event time vs the
    8 TB/s HBM peak."""
    import torch
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
    if world > 1:
        dist.all_reduce(wall_t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        wall = float(wall_t.item())
        print(json.dumps({"metric": "dry-run", "value": world * a.batch * a.steps / wall, "n_gpus": world,
                          "steps": a.steps, "warmup": a.warmup, "max_wall": wall}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(sd_cpu, cfg, synth):
    """The oracle (CPU restatement of the reference, pinned to it by oracle/gen_golden.py) timed on
    the host cores: one denoise step at B=1 = 1/(8*K) of the GPU workload's unit count."""
    from oracle import unet_oracle as uo
    # oneDNN convs of this size get slower beyond a few dozen threads (256 threads: 98 s/step measured)
    torch.set_num_threads(min(os.cpu_count(), 32))
    ocfg = synth.oracle_cfg(cfg)
    R = cfg.data.image_size
    x = synth.synthetic_inputs(1, 4, R, seed=42)
    z = synth.synthetic_inputs(1, 4, R, seed=43)
    mask = synth.synthetic_grid_mask(R).view(1, R, R, R)
    times = []
    with torch.no_grad():
        for i in range(2):
            if times and times[0] > 12.0:
                break            # keep the default bench run within minutes
            t0 = time.perf_counter()
            t = torch.tensor(1.0 - i * 1e-3)
            e = uo.unet_res64_forward(sd_cpu, ocfg, x, torch.ones(1) * t * 999)
            x, _ = uo.ancestral_step(x, e, z, t, mask)
            times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(1.0 / best, 4), "unit": "sample-steps/s", "cores": torch.get_num_threads(),
            "kind": "port", "s_per_sample_step": round(best, 2),
            "sample": f"{len(times)} denoise step(s) (res64 U-Net eval + ancestral update) at batch=1 on the host CPU, best; "
                      "PyTorch fp32 oracle restatement (oracle/unet_oracle.py)"}


if __name__ == "__main__":
    main()
