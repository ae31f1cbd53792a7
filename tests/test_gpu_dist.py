"""Two training replicas on ONE GPU (gloo carries the exchange; on a node each rank has its own GPU and RCCL):
the real HIP forward/backward + start-up broadcast + overlapped bucket reducer + optimizer must leave both replicas
with identical parameters, equal to a single process stepping on the whole batch."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
B_TOTAL = 16


def _reap(procs):
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _one_step(rank, world, shard, micro=1):
    """Build the small U-Net (per-rank init seed), broadcast, run one training step on `shard` of the common batch
    (`micro` > 1: the shard in that many accumulated micro-steps, trainer.py:94-116 with training.iter_size = micro)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle.gen_golden import fixed_draws, train_step_inputs
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import losses, parallel, sde_lib
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
    torch.cuda.set_device(0)
    cfg = synth.small_config(); cfg.device = torch.device("cuda:0")
    cfg.optim.warmup, cfg.optim.lr = 1, 1e-3
    R = cfg.data.image_size
    model = mutils.create_model(cfg)
    sd = synth.sensitised_state_dict(model.module.state_dict(), seed=100 + rank, grid_mask=synth.synthetic_grid_mask(R))
    model.module.load_state_dict(sd, strict=True)          # replicas start DIFFERENT on purpose
    parallel.broadcast_params_(model.parameters())          # ... and leave the broadcast as rank 0's copy
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    opt = losses.get_optimizer(cfg, model.parameters())
    batch, labels, noise, mask = train_step_inputs(B_TOTAL, R, seed=77)
    sde = sde_lib.VPSDE(cfg.model.beta_min, cfg.model.beta_max, cfg.model.num_scales, device="cuda")
    step_fn = losses.get_step_fn(sde, train=True, optimize_fn=losses.optimization_manager(cfg), mask=mask.cuda())
    state = dict(model=model, ema=ema, optimizer=opt, step=1)
    idx = list(range(B_TOTAL))[shard] if isinstance(shard, slice) else list(shard)
    per = len(idx) // micro
    loss = 0.0
    for m in range(micro):
        sel = idx[m * per:(m + 1) * per]
        with fixed_draws(labels[sel].cuda(), noise[sel].cuda()):
            loss = loss + step_fn(state, batch[sel].cuda(), clear_grad=(m == 0), update_param=(m == micro - 1))["loss"] / micro
    return (float(loss.detach()), [p.detach().cpu().clone() for p in model.parameters()],
            [p.grad.detach().cpu().clone() for p in model.parameters() if p.grad is not None])


def _worker(rank, world, port, q, micro=1):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from meshdiffusion_amd.lib.diffusion import parallel
    parallel.init_distributed(backend="gloo")
    per = B_TOTAL // world
    # micro > 1: rank r's micro-step m takes samples [m*B/micro + r*per/micro, ...): the union over ranks of micro-step m is
    # the m-th contiguous block of the batch, i.e. what a single process accumulating `micro` steps of B/micro would see
    if micro > 1:
        pm = per // micro
        sel = [m * (B_TOTAL // micro) + rank * pm + k for m in range(micro) for k in range(pm)]
        loss, params, grads = _one_step(rank, world, sel, micro=micro)
    else:
        loss, params, grads = _one_step(rank, world, slice(rank * per, (rank + 1) * per))
    q.put((rank, loss, [t.numpy() for t in params], [t.numpy() for t in grads]))       # by value: the child exits before the parent reads
    dist.barrier()
    dist.destroy_process_group()


def test_two_replicas_one_step_equals_full_batch_step(hip_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            rank, loss, params, grads = q.get(timeout=300)
            res[rank] = (loss, [torch.from_numpy(a) for a in params], [torch.from_numpy(a) for a in grads])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    # replicas identical after the step (same averaged gradient, same update)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    # single process, whole batch, rank 0's initialisation: the averaged shard gradients are the full-batch gradient
    loss_full, params_full, grads_full = _one_step(0, 1, slice(0, B_TOTAL))
    assert abs(0.5 * (res[0][0] + res[1][0]) - loss_full) < 1e-5 * abs(loss_full)

    def rel(xs, ys):
        num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(xs, ys))
        return (num / sum(float(b.double().pow(2).sum()) for b in ys)) ** 0.5

    assert len(res[0][2]) == len(grads_full) and rel(res[0][2], grads_full) < 1e-5
    # ... and so is the update, up to Adam's first step turning 1e-7 differences of near-zero gradients into +-lr
    assert rel(res[0][1], params_full) < 1e-4


def test_two_replicas_gradient_accumulation_equals_single_process_accumulation(hip_lib):
    """VERDICT r02 item 3(iv) on the HIP path: training.iter_size = 2 on two replicas (micro-step 1 accumulates locally, the
    in-place bucket all-reduces ride under micro-step 2's backward) == one process accumulating the same two micro-batches."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, 2)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            rank, loss, params, grads = q.get(timeout=300)
            res[rank] = (loss, [torch.from_numpy(a) for a in params], [torch.from_numpy(a) for a in grads])
        for p in procs:
            p.join(timeout=120)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    for a, b in zip(res[0][1], res[1][1]):
        assert torch.equal(a, b)
    loss_full, params_full, grads_full = _one_step(0, 1, slice(0, B_TOTAL), micro=2)

    def rel(xs, ys):
        num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(xs, ys))
        return (num / sum(float(b.double().pow(2).sum()) for b in ys)) ** 0.5

    assert abs(0.5 * (res[0][0] + res[1][0]) - loss_full) < 1e-5 * abs(loss_full)
    assert len(res[0][2]) == len(grads_full) and rel(res[0][2], grads_full) < 1e-5
    assert rel(res[0][1], params_full) < 1e-4


def _nccl_world1_worker(port, q):
    """RCCL ("nccl" backend) in a world of ONE rank on the one GPU of this box: the backend branch of
    parallel.init_distributed, GradReducer's in-place AVG bucket all-reduces on the flat gradient buffer (launched from
    the HIP backward) and the copying fallback all execute on RCCL; averaging over one rank must change nothing."""
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from meshdiffusion_amd.lib.diffusion import parallel
    rank, world, _ = parallel.init_distributed(backend="nccl", force=True)
    assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
    # plain buffers through both reducer modes
    g = torch.Generator().manual_seed(1)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in ((300, 7), (5,), (64, 64, 3))]
    grads = [torch.randn(p.shape, generator=g).cuda() for p in ps]
    fg = parallel.FlatGrads(ps)
    fg.attach()
    for p, gr in zip(ps, grads):
        p.grad.copy_(gr)
    red = parallel.GradReducer(cap_bytes=4096, flat=fg, force=True)
    assert red.active and red.avg is not None
    red.ready(ps[:2]); red.finish(ps)
    torch.cuda.synchronize()
    assert red.stats["buckets"] == 2 and all(torch.equal(p.grad, gr) for p, gr in zip(ps, grads))
    for p, gr in zip(ps, grads):
        p.grad = gr.clone()
    red = parallel.GradReducer(cap_bytes=4096, force=True)
    red.ready(ps[2:]); red.finish(ps)
    torch.cuda.synchronize()
    assert all(torch.equal(p.grad, gr) for p, gr in zip(ps, grads))
    # one real training step with the exchange forced on == the same step without it
    plain = _one_step(0, 1, slice(0, B_TOTAL))
    parallel.FORCE_EXCHANGE = True
    forced = _one_step(0, 1, slice(0, B_TOTAL))
    def rel(xs, ys):
        num = sum(float((a.double() - b.double()).pow(2).sum()) for a, b in zip(xs, ys))
        return (num / sum(float(b.double().pow(2).sum()) for b in ys)) ** 0.5
    # (two runs of the step differ by the order of the fp64 GroupNorm-statistics atomics: compare to 1e-6, not bitwise)
    q.put(("ok", plain[0], forced[0], max(rel(forced[2], plain[2]), rel(forced[1], plain[1]))))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_world_size_1_gradient_exchange(hip_lib):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), q))
    p.start()
    try:
        tag, l0, l1, same = q.get(timeout=300)
        p.join(timeout=120)
        assert p.exitcode == 0
    finally:
        _reap([p])
    # gradients through RCCL AVG over one rank are unchanged; so are the updated parameters
    assert tag == "ok" and abs(l0 - l1) <= 1e-6 * abs(l0) and same < 1e-6, (l0, l1, same)


def _cli_sampling_worker(rank, world, port, tmp, total):
    """One rank of `torchrun --nproc-per-node 2 main_diffusion.py --mode=uncond_gen` (both ranks on the one GPU of this box:
    LOCAL_RANK 0, gloo carries the final gather)."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      MD_DIST_BACKEND="gloo")
    sys.path.insert(0, ROOT)
    os.chdir(tmp)
    import torch.distributed as dist
    import main_diffusion
    main_diffusion.main(["--config", os.path.join(tmp, "small.py"), "--mode=uncond_gen", f"--config.eval.eval_dir={tmp}/out",
                         f"--config.eval.ckpt_path={tmp}/ckpt/checkpoint.pth", f"--config.eval.batch_size={total}",
                         "--config.seed=11"])
    dist.destroy_process_group()


def test_cli_uncond_gen_two_ranks_per_shard_parity(hip_lib, tmp_path, monkeypatch):
    """VERDICT r02 item 4 on the real HIP path: two ranks sample a batch of 3 through main_diffusion.py; rank 0's single
    output file must hold, in rank order, exactly what single-process runs with batch sizes[r] and seed config.seed + r
    produce (the per-shard parity definition of SURVEY 8e; reference: evaler.py:14-60 + models/utils.py:88-96)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import main_diffusion
    from test_gpu_cli import _write_ckpt_and_mask
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import parallel
    tmp = str(tmp_path)
    cfg = synth.small_config(); cfg.device = torch.device("cuda")
    cfg.model.num_scales = 40                      # (below ~21 levels the discrete betas exceed 1: NaN for any sampler)
    _write_ckpt_and_mask(tmp_path, cfg, synth)
    with open(os.path.join(tmp, "small.py"), "w") as f:
        f.write("from meshdiffusion_amd import synth\n\ndef get_config():\n    c = synth.small_config()\n"
                "    c.model.num_scales = 40\n    return c\n")
    total, R = 3, cfg.data.image_size
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_cli_sampling_worker, args=(r, 2, _free_port_pair(), tmp, total)) for r in range(2)]
    try:
        for p in procs:
            p.start()
        for p in procs:
            p.join(timeout=300)
            assert p.exitcode == 0
    finally:
        _reap(procs)
    assert sorted(os.listdir(os.path.join(tmp, "out"))) == ["0.npy"]
    x = np.load(os.path.join(tmp, "out", "0.npy"))
    assert x.shape == (total, 4, R, R, R) and np.isfinite(x).all() and np.abs(x).max() > 0
    # the same shards from single-process runs (no process group: evaler does not seed, the caller does)
    monkeypatch.chdir(tmp_path)
    sizes, row = parallel.shard_sizes(total, 2), 0
    for r, n in enumerate(sizes):
        torch.manual_seed(11 + r)
        main_diffusion.main(["--config", os.path.join(tmp, "small.py"), "--mode=uncond_gen", f"--config.eval.eval_dir={tmp}/single{r}",
                             f"--config.eval.ckpt_path={tmp}/ckpt/checkpoint.pth", f"--config.eval.batch_size={n}"])
        ref = np.load(os.path.join(tmp, f"single{r}", "0.npy"))
        err = np.linalg.norm(x[row:row + n].astype(np.float64) - ref) / np.linalg.norm(ref)
        print(f"rank {r}: shard of {n} vs its single-process run: rel-L2 {err:.2e}")
        assert err < 1e-5       # same kernels, same draws; only the fp64-atomic order of the GroupNorm sums differs
        row += n


_PORT = []


def _free_port_pair():
    """Both workers of one test must get the SAME port."""
    if not _PORT:
        _PORT.append(_free_port())
    return _PORT[0]
