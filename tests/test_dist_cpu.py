"""world_size-2 gloo tests of the multi-GPU sampling host logic (no kernels: a stub sampler stands in
for the HIP U-Net, which the -m gpu tests cover)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wire(o):
    """Results travel through the multiprocessing queue BY VALUE (numpy): a torch tensor is sent as a shared-memory handle that
    dies with the worker -- a parent that reads the queue after the worker has exited gets FileNotFoundError (seen 1 run in 5)."""
    if isinstance(o, torch.Tensor):
        return ("__t__", o.detach().cpu().numpy().copy())
    if isinstance(o, (list, tuple)):
        return type(o)(_wire(v) for v in o)
    return o


def _unwire(o):
    if isinstance(o, tuple) and len(o) == 2 and isinstance(o[0], str) and o[0] == "__t__":
        return torch.from_numpy(o[1])
    if isinstance(o, (list, tuple)):
        return type(o)(_unwire(v) for v in o)
    return o


def retry_rendezvous(fn):
    """The multi-process tests rendezvous over 127.0.0.1 on a port that was free a moment ago; on a busy host another
    process can grab it first (seen right after large file transfers).  One retry with a fresh port."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        try:
            return fn(*a, **k)
        except Exception as e:   # noqa: BLE001
            print(f"first attempt failed ({type(e).__name__}: {e}); retrying once", flush=True)
            return fn(*a, **k)
    return wrapped


def _reap(procs):
    for p in procs:
        if p.is_alive():
            p.terminate()
            p.join(timeout=10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stub_sampler(local_batch, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn((local_batch, 4, 4, 4, 4), generator=g)


def _stub_sampler_f64(local_batch, seed):      # the DDIM sampler's state is float64 like the reference's
    return _stub_sampler(local_batch, seed).double() / 3.0


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from meshdiffusion_amd.lib.diffusion import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    out = parallel.sharded_sample(_stub_sampler, total, seed=100)
    local = parallel.sharded_sample(_stub_sampler, total, seed=100, gather=False)
    out64 = parallel.sharded_sample(_stub_sampler_f64, total, seed=100)
    if rank == 0:     # the gathered file keeps the sampler's dtype and bits whatever the world size
        sizes = parallel.shard_sizes(total, world)
        want = torch.cat([_stub_sampler_f64(sizes[r], 100 + r) for r in range(world)], 0)
        assert out64.dtype == torch.float64 and torch.equal(out64, want)
    q.put(_wire((rank, None if out is None else out.clone(), local.clone())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 5])
@retry_rendezvous
def test_sharded_sampling_two_ranks_gloo(total):
    from meshdiffusion_amd.lib.diffusion import parallel
    assert parallel.shard_sizes(5, 2) == [3, 2] and parallel.shard_sizes(8, 8) == [1] * 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, out, local = _unwire(q.get(timeout=120))
        res[rank] = (out, local)
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    _reap(procs)
    assert codes == [0] * len(procs), codes
    sizes = parallel.shard_sizes(total, 2)
    expect = torch.cat([_stub_sampler(sizes[r], 100 + r) for r in range(2)], 0)
    assert res[1][0] is None
    assert torch.equal(res[0][0], expect)                       # rank-0 gather == per-shard reference runs
    assert torch.equal(res[1][1], _stub_sampler(sizes[1], 101))  # each rank == a run with its own seed


def _grad_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from meshdiffusion_amd.lib.diffusion import parallel
    parallel.init_distributed(backend="gloo")
    # local "gradient" of a mean-over-local-batch loss on this rank's shard of a batch of 8
    g = torch.Generator().manual_seed(3)
    data = torch.randn((8, 1000), generator=g)
    shard = data[rank * 4:(rank + 1) * 4]
    flat = shard.mean(dim=0).clone()
    parallel.allreduce_grads_(flat)
    q.put(_wire((rank, flat.clone())))
    dist.barrier()
    dist.destroy_process_group()


@retry_rendezvous
def test_grad_allreduce_equals_large_batch_gradient_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(_unwire(q.get(timeout=120)) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    _reap(procs)
    assert codes == [0] * len(procs), codes
    g = torch.Generator().manual_seed(3)
    full = torch.randn((8, 1000), generator=g).mean(dim=0)
    assert torch.allclose(res[0], full, atol=1e-6) and torch.equal(res[0], res[1])


def _replica_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from meshdiffusion_amd.lib.diffusion import parallel
    parallel.init_distributed(backend="gloo")
    torch.manual_seed(10 + rank)                                   # replicas start DIFFERENT on purpose
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    parallel.broadcast_params_(net.parameters(), cap_bytes=64)     # tiny cap: several buckets
    start = [p.detach().clone() for p in net.parameters()]
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn((8, 6), generator=g), torch.randn((8, 3), generator=g)
    sl = slice(rank * 4, rank * 4 + 4)
    ((net(x[sl]) - y[sl]) ** 2).mean().backward()
    local = [p.grad.clone() for p in net.parameters()]
    parallel.allreduce_param_grads_(net.parameters(), cap_bytes=64)
    reduced = [p.grad.clone() for p in net.parameters()]
    # the overlapped reducer used by the training step: layers announced in backward order, late parameters
    # (here: the first layer's bias) only known at finish()
    for p, g in zip(net.parameters(), local):
        p.grad.copy_(g)
    red = parallel.GradReducer(cap_bytes=64)
    assert red.active
    red.ready(list(net[2].parameters()))
    red.ready([net[0].weight])
    red.finish(net.parameters())
    for p, g in zip(net.parameters(), reduced):
        assert torch.allclose(p.grad, g, atol=1e-7)
    # the same exchange IN PLACE on one flat gradient buffer laid out in completion order (what losses.get_step_fn does):
    # finished layers are a growing prefix whose new part goes out once it reaches the bucket size
    order = list(net[2].parameters()) + [net[0].weight, net[0].bias]
    fg = parallel.FlatGrads(order)
    assert fg.matches(net.parameters()) and fg.n == sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        p.grad = None
    fg.attach()
    for p, g in zip(net.parameters(), local):
        assert p.grad.data_ptr() == fg.views[fg.slices[id(p)][0]].data_ptr()
        p.grad.copy_(g)
    red = parallel.GradReducer(cap_bytes=48, flat=fg)
    red.ready([net[0].weight])                      # announced out of order: nothing contiguous is final yet
    assert red.sent == 0 and not red.pending
    red.ready(list(net[2].parameters()))            # prefix = Linear(5,3) (18 floats) + net[0].weight (30): one bucket
    assert red.sent == 48 and len(red.pending) == 1
    red.finish(net.parameters())
    assert red.stats["buckets"] == 2 and red.stats["bytes"] == 4 * fg.n
    for p, g in zip(net.parameters(), reduced):
        assert torch.allclose(p.grad, g, atol=1e-7)
    q.put(_wire((rank, start, reduced)))
    dist.barrier()
    dist.destroy_process_group()


@retry_rendezvous
def test_replica_broadcast_and_bucketed_grad_allreduce_gloo():
    """trainer.py's exchange: start-up broadcast makes the replicas identical; the bucketed mean of per-rank
    gradients equals the gradient of the global batch (training.batch_size = sum of the shards)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_replica_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, start, grads = _unwire(q.get(timeout=120))
        res[rank] = (start, grads)
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    _reap(procs)
    assert codes == [0] * len(procs), codes
    torch.manual_seed(10)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    g = torch.Generator().manual_seed(3)
    x, y = torch.randn((8, 6), generator=g), torch.randn((8, 3), generator=g)
    ((net(x) - y) ** 2).mean().backward()
    for k, p in enumerate(net.parameters()):
        assert torch.equal(res[0][0][k], p.detach()) and torch.equal(res[1][0][k], p.detach())
        assert torch.allclose(res[0][1][k], p.grad, atol=1e-6) and torch.equal(res[0][1][k], res[1][1][k])


def _check_dry_run_exchange(d, world):
    """the `train_step.exchange` object of an N-rank line -- per-rank exposed wait, bucket count / bytes and the
    world size the PROCESS GROUP reports -- produced by the real reducer (parallel.FlatGrads + GradReducer) under gloo."""
    ts = d["train_step"]
    ex = ts["exchange"]
    assert ts["grads_averaged"] is True
    assert ex["world_size"] == world and ex["backend"] == "gloo" and "all-reduce" in ex["collective"]
    assert ex["bytes"] == ts["grad_bytes"] == (300_000 + 50_000 + 700_000 + 1_000) * 4          # every gradient byte exactly once
    assert ex["bucket_cap_bytes"] == 1 << 20 and 2 <= ex["buckets"] <= 5                        # in-place prefix buckets of >= 1 MB + the rest
    assert [r["rank"] for r in ex["per_rank"]] == list(range(world))
    for r in ex["per_rank"]:
        assert r["buckets"] == ex["buckets"] and r["bytes"] == ex["bytes"]
        assert 0.0 <= r["exchange_exposed_ms"] < 60_000 and r["backward_ms"] >= 0.0
        # each rank echoes the device it ran on (CPU dry run: none) so that a first 8-GPU line shows 8 distinct ones
        assert r["device_index"] == -1 and r["visible_devices"] == 0 and "pci_bus_id" in r and "HIP_VISIBLE_DEVICES" in r
    assert ex["distinct_devices"] == 0
    assert ex["exposed_ms"] == ex["per_rank"][0]["exchange_exposed_ms"]


@retry_rendezvous
def test_bench_multi_process_control_flow_dry_run():
    """bench.py under torch.distributed.run with 2 ranks: rendezvous on 127.0.0.1, barrier, MAX over ranks,
    exactly one JSON line from rank 0 (the GPU work itself is covered by the -m gpu suite and `bench.py`)."""
    import json
    import subprocess
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--dry-run"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=180, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and abs(d["max_wall"] - 0.2) < 1e-9 and d["value"] == 2 * 8 * 3 / 0.2
    _check_dry_run_exchange(d, 2)


@retry_rendezvous
def test_bench_gpus_flag_starts_its_own_ranks():
    """plain `python bench.py --gpus 2` (no torchrun around it, WORLD_SIZE unset) must come up with two
    ranks -- it re-executes itself under torch.distributed.run -- and report n_gpus from the process group."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                          "--dry-run"], capture_output=True, text=True, timeout=240, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and abs(d["max_wall"] - 0.2) < 1e-9 and d["value"] == 2 * 8 * 3 / 0.2
    _check_dry_run_exchange(d, 2)


def test_bench_refuses_fewer_ranks_than_gpus_flag():
    """A 1-rank run must never print a line for --gpus N: a WORLD_SIZE that disagrees with the flag (also WORLD_SIZE=1),
    or fewer devices than ranks on the real path, exits non-zero without a JSON line."""
    import subprocess
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE=1" in out.stderr and "{" not in out.stdout
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, timeout=120, cwd=ROOT, env=env)     # no --dry-run: this container has no GPU at all
    import torch
    if torch.cuda.device_count() < 2:
        assert out.returncode != 0 and "refusing to run fewer ranks" in out.stderr and "{" not in out.stdout


def _evaler_worker(rank, world, port, tmp, total):
    """One rank of `main_diffusion.py --mode=uncond_gen` / cond_gen under torchrun; the sampler is a stub (the HIP path
    needs a GPU; tests/test_gpu_dist.py runs the same flow through the real kernels)."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MD_DIST_BACKEND="gloo")
    os.chdir(tmp)
    import numpy as np
    import main_diffusion
    from meshdiffusion_amd.lib.diffusion import sampling

    def fake_get_sampling_fn(config, sde, shape, inverse_scaler, eps, grid_mask=None, return_traj=False):
        def fn(model, partial=None, partial_mask=None, partial_channel=0, freeze_iters=None):
            # rows carry (rank's global seed, the local batch the sampler was built for, conditioning flag)
            x = torch.zeros(shape)
            x[:, 0] = float(torch.initial_seed())
            x[:, 1] = float(shape[0])
            x[:, 2] = 0.0 if partial is None else float(partial.abs().sum())
            return x, 0
        return fn

    sampling.get_sampling_fn = fake_get_sampling_fn
    common = ["--config", os.path.join(tmp, "small.py"), f"--config.eval.eval_dir={tmp}/out",
              f"--config.eval.ckpt_path={tmp}/ckpt.pth", f"--config.eval.batch_size={total}", "--config.seed=7"]
    main_diffusion.main(common + ["--mode=uncond_gen"])
    if rank == 0:
        x = np.load(os.path.join(tmp, "out", "0.npy"))
        np.save(os.path.join(tmp, "uncond.npy"), x)
    main_diffusion.main(common + ["--mode=cond_gen", f"--config.eval.partial_dmtet_path={tmp}/dmtet.pt",
                                  f"--config.eval.tet_path={tmp}/tets.npz", "--config.eval.freeze_iters=3"])
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 1])
@retry_rendezvous
def test_cli_generation_shards_over_two_ranks_and_writes_one_file_gloo(total, tmp_path):
    """`torchrun --nproc-per-node 2 main_diffusion.py --mode=uncond_gen|cond_gen` -- every rank samples
    its shard of eval.batch_size with seed config.seed + rank, rank 0 gathers and writes ONE {idx}.npy with the full batch
    in rank order (reference: evaler.py:14-60 + DataParallel, models/utils.py:88-96).  total = 1: rank 1 has no sample."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.lib.diffusion import losses, parallel
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from meshdiffusion_amd.lib.diffusion.models.ema import ExponentialMovingAverage
    from meshdiffusion_amd.lib.diffusion.utils import save_checkpoint
    tmp = str(tmp_path)
    cfg = synth.small_config(); cfg.device = torch.device("cpu")
    R = cfg.data.image_size
    model = mutils.create_model(cfg)
    ema = ExponentialMovingAverage(model.parameters(), decay=cfg.model.ema_rate)
    save_checkpoint(os.path.join(tmp, "ckpt.pth"),
                    dict(optimizer=losses.get_optimizer(cfg, model.parameters()), model=model, ema=ema, step=3))
    os.makedirs(os.path.join(tmp, "data"), exist_ok=True)
    m = synth.synthetic_grid_mask(R)
    torch.save(m, os.path.join(tmp, "data", f"grid_mask_{R}.pt"))
    with open(os.path.join(tmp, "small.py"), "w") as f:
        f.write("import torch\nfrom meshdiffusion_amd import synth\n\ndef get_config():\n    c = synth.small_config()\n"
                "    c.device = torch.device('cpu')\n    return c\n")
    idx = np.argwhere(m.numpy() > 0).astype(np.float32)
    np.savez(os.path.join(tmp, "tets.npz"), vertices=(idx / (R - 1) - 0.5).astype(np.float32), indices=np.zeros((1, 4), np.int32))
    g = torch.Generator().manual_seed(1)
    part = {"sdf": torch.sign(torch.randn(len(idx), generator=g)), "vis": torch.rand(len(idx), generator=g) < 0.5}
    torch.save(part, os.path.join(tmp, "dmtet.pt"))
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_evaler_worker, args=(r, 2, port, tmp, total)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    codes = [p.exitcode for p in procs]
    _reap(procs)
    assert codes == [0, 0], codes
    sizes = parallel.shard_sizes(total, 2)
    files = sorted(os.listdir(os.path.join(tmp, "out")))
    assert files == ["0.npy"], files                                      # ONE file, written by rank 0
    for name, cond in (("uncond.npy", False), (os.path.join("out", "0.npy"), True)):
        x = np.load(os.path.join(tmp, name))
        assert x.shape == (total, 4, R, R, R) and x.dtype == np.float32
        row = 0
        for r, n in enumerate(sizes):
            for _ in range(n):
                assert x[row, 0].flat[0] == 7 + r and x[row, 1].flat[0] == n    # seed + rank, built for the local batch
                assert (x[row, 2].flat[0] != 0) == cond
                row += 1


def _accum_worker(rank, world, port, q):
    """Gradient accumulation (trainer.py:94-116, training.iter_size = 2) through the in-place flat reducer, the way
    losses.get_step_fn drives it: micro-step 1 only accumulates (no reducer exists); micro-step 2 accumulates layer by layer
    and announces each finished layer, so bucket all-reduces of the finished prefix are in flight while later layers of the
    SAME flat buffer are still being written."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from meshdiffusion_amd.lib.diffusion import parallel
    parallel.init_distributed(backend="gloo")
    shapes = [(40, 30), (30,), (64, 16), (16,), (7, 5, 3), (200,)]
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    fg = parallel.FlatGrads(params)                     # completion order = list order
    g = torch.Generator().manual_seed(50 + rank)
    micro = [[torch.randn(s, generator=g) for s in shapes] for _ in range(2)]
    fg.zero_(); fg.attach()                             # clear_grad on the first micro-step
    for p, gr in zip(params, micro[0]):                 # micro-step 1: update_param=False -> no GradReducer
        p.grad.add_(gr)
    red = parallel.GradReducer(cap_bytes=1024, flat=fg)  # micro-step 2 (update_param=True)
    fg.attach()
    assert red.active
    launched = []
    for p, gr in zip(params, micro[1]):
        p.grad.add_(gr)                                 # this layer's gradient becomes final ...
        red.ready([p])                                  # ... and is announced; earlier buckets may already be on the wire
        launched.append(len(red.pending))
    red.finish(params)
    assert launched[0] >= 1 and red.stats["buckets"] >= 3 and red.stats["bytes"] == 4 * fg.n
    q.put(_wire((rank, [m for m in micro], fg.flat.clone())))
    dist.barrier()
    dist.destroy_process_group()


@retry_rendezvous
def test_gradient_accumulation_two_micro_steps_flat_reducer_gloo():
    """iter_size = 2 with the in-place prefix all-reduces -- the exchanged buffer must hold the
    rank-mean of the SUM of both micro-steps' gradients (the reference accumulates without dividing, trainer.py:94-116)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(2):
        rank, micro, flat = _unwire(q.get(timeout=120))
        res[rank] = (micro, flat)
    for p in procs:
        p.join(timeout=60)
    codes = [p.exitcode for p in procs]
    _reap(procs)
    assert codes == [0, 0], codes
    expect = torch.cat([(0.5 * (res[0][0][0][k] + res[0][0][1][k] + res[1][0][0][k] + res[1][0][1][k])).reshape(-1)
                        for k in range(6)])
    assert torch.allclose(res[0][1], expect, atol=1e-6) and torch.equal(res[0][1], res[1][1])
