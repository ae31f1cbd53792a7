"""Multi-GPU sampling: one process per GPU, independent sample shards, no data-path collective.

Replaces the reference's single-process `torch.nn.DataParallel` (lib/diffusion/models/utils.py:88-96),
which re-broadcasts all 1.46 GB of parameters on every denoise step.  Each rank keeps a
static weight replica and draws its own shard with `seed + rank`; the only collective is the optional
final gather of the finished `[B,4,R,R,R]` grids (4.2 MB/sample) to rank 0 over RCCL (backend "nccl" on
ROCm) or gloo (CPU tests).
"""
import os

import torch
import torch.distributed as dist

from ... import hip_ops as ops


def shard_sizes(total, world):
    """Split `total` samples over `world` ranks as evenly as possible (first ranks get the remainder)."""
    base, rem = divmod(int(total), int(world))
    return [base + (1 if r < rem else 0) for r in range(world)]


# Self-test switch: run the gradient exchange's collectives even in a world of one rank (RCCL all-reduce of every
# bucket onto itself) -- tests/test_gpu_dist.py uses it to execute the "nccl" code path on a single-GPU box.
FORCE_EXCHANGE = False


def init_distributed(backend=None, force=False):
    """Initialise from the torchrun environment; returns (rank, world, local_rank).  No-op for world 1 unless `force`."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:     # MD_DIST_BACKEND=gloo: several ranks on ONE GPU (tests; RCCL refuses two ranks per device)
            backend = os.environ.get("MD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        kw = dict(device_id=torch.device("cuda", local_rank)) if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank


def barrier():
    """dist.barrier() on this rank's device (no-op outside a multi-rank run)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def sharded_sample(sample_fn, total_batch, seed, gather=True):
    """Run `sample_fn(local_batch, seed + rank) -> tensor[local_batch, ...]` on every rank.

    Returns the concatenated `[total_batch, ...]` tensor on rank 0 (None elsewhere) when `gather`,
    else the local shard.  Rank r owns samples [sum(sizes[:r]), sum(sizes[:r+1])).
    """
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    sizes = shard_sizes(total_batch, world)
    local = sample_fn(sizes[rank], int(seed) + rank) if sizes[rank] > 0 else None
    if not gather or world == 1:
        return local
    # all ranks must contribute equal-shaped buffers: pad the shard to the largest size
    ref = local if local is not None else None
    # slot 0: rank count of dims, 1..6: shape, 7: dtype code -- the gather keeps the shard's OWN dtype (the DDIM sampler
    # returns float64 state like the reference: a world-size-dependent cast would change the written .npy)
    codes = {torch.float32: 1, torch.float64: 2, torch.float16: 3, torch.bfloat16: 4}
    # slot 8: MINUS the dtype code (MAX-reduced like the rest: -min over the ranks that hold samples) -- every rank learns whether the
    # ranks disagree and ALL of them raise; a rank that raised alone would leave the others hanging in the gather
    shape_t = torch.zeros(9, dtype=torch.int64)
    shape_t[8] = -(1 << 40)
    if ref is not None:
        assert ref.dim() <= 6 and ref.dtype in codes, (ref.shape, ref.dtype)
        shape_t[0] = ref.dim()
        shape_t[1:1 + ref.dim()] = torch.tensor(ref.shape)
        shape_t[7] = codes[ref.dtype]
        shape_t[8] = -codes[ref.dtype]
    # RCCL moves device buffers; gloo gathers through host memory (also when the shards live on a GPU)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    shape_t = shape_t.to(dev)
    dist.all_reduce(shape_t, op=dist.ReduceOp.MAX)
    nd = int(shape_t[0])
    full = [int(v) for v in shape_t[1:1 + nd]]
    if int(shape_t[7]) == 0:            # no rank holds a sample (total_batch = 0): nothing to gather, on every rank alike
        return None
    if int(shape_t[7]) != -int(shape_t[8]):
        names = {v: k for k, v in codes.items()}
        raise RuntimeError(f"sharded_sample: the ranks sampled different dtypes ({names[-int(shape_t[8])]} .. {names[int(shape_t[7])]}); "
                           f"rank {rank} holds {None if local is None else local.dtype}")
    dtype = {v: k for k, v in codes.items()}[int(shape_t[7])]
    buf = torch.zeros([max(sizes)] + full[1:], dtype=dtype, device=dev)
    if local is not None:
        buf[:sizes[rank]] = local.to(dev)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0)
    if rank != 0:
        return None
    return torch.cat([o[:n] for o, n in zip(out, sizes)], dim=0)


def allreduce_grads_(flat_grad):
    """Average a flat fp32 gradient buffer over all ranks in place (one RCCL all-reduce of 1.456 GB for
    res64 instead of DataParallel's reduce-to-GPU0).  Equal shards + mean-over-local-batch
    loss => the average reproduces the large-batch gradient.  No-op for world size 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return flat_grad
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    flat_grad.div_(dist.get_world_size())
    return flat_grad


def broadcast_params_(params, src=0, cap_bytes=256 << 20):
    """Start-up only: copy rank `src`'s parameters to every replica (bucketed, <= 256 MB per broadcast)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    for bucket in _buckets([p.data for p in params], cap_bytes):
        flat = torch._utils._flatten_dense_tensors(bucket)
        dist.broadcast(flat, src=src)
        for t, f in zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)):
            t.copy_(f)
    ops.bump_param_epoch()   # `.data.copy_` does not bump `_version`: packed-weight caches must not survive the broadcast


def _buckets(tensors, cap_bytes=256 << 20):
    out, cur, size = [], [], 0
    for t in tensors:
        n = t.numel() * t.element_size()
        if cur and size + n > cap_bytes:
            out.append(cur)
            cur, size = [], 0
        cur.append(t)
        size += n
    if cur:
        out.append(cur)
    return out


def allreduce_param_grads_(params, cap_bytes=256 << 20):
    """Mean of `p.grad` over ranks for separately allocated gradients (the reference-format optimizer path):
    a few large flattened buckets (<= 256 MB: xGMI rings are per-link bound, so few big messages), launched
    asynchronously back to back and unpacked once all are in flight.  If the gradients are already views of
    one flat buffer use `allreduce_grads_` on it instead.  No-op for world size 1."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    grads = [p.grad for p in params if p.requires_grad and p.grad is not None]
    pending = []
    for bucket in _buckets(grads, cap_bytes):
        flat = torch._utils._flatten_dense_tensors(bucket)
        pending.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
    for work, flat, bucket in pending:
        work.wait()
        flat.div_(world)
        for g, f in zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)):
            g.copy_(f)


class FlatGrads:
    """Every trainable parameter's `.grad` as a view of ONE flat fp32 buffer, laid out in the order the backward pass
    finishes them (`DDPMUNet3D.grad_completion_order`).  A finished prefix of the buffer is then a contiguous slice that
    `GradReducer` all-reduces IN PLACE: no flatten copy, no copy-back, and (RCCL: `ReduceOp.AVG`) no division pass --
    the exchange touches the 1.456 GB of res64 gradients exactly once.  `torch.optim.Adam` / `clip_grad_norm_` work on
    the views unchanged."""

    def __init__(self, ordered_params):
        self.params = [p for p in ordered_params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        self.slices, off = {}, 0
        for i, p in enumerate(self.params):
            if id(p) in self.slices:
                raise ValueError("parameter listed twice")
            self.slices[id(p)] = (i, off, p.numel())
            off += p.numel()
        self.n = off
        self.flat = torch.zeros(self.n, dtype=torch.float32, device=dev)
        self.views = [self.flat[o:o + n].view_as(p) for p, (_, o, n) in ((p, self.slices[id(p)]) for p in self.params)]

    def matches(self, params):
        ps = [p for p in params if p.requires_grad]
        return (len(ps) == len(self.params) and all(id(p) in self.slices and self.slices[id(p)][2] == p.numel() for p in ps)
                and ps[0].device == self.flat.device)

    def attach(self):
        """(Re-)point every `.grad` at its view (optimizer.zero_grad() sets them to None)."""
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def zero_(self):
        self.flat.zero_()


def flat_grads_for(net):
    """The FlatGrads of a network that knows its gradient completion order (cached on the module; rebuilt when the
    parameters were re-created or moved)."""
    order = net.grad_completion_order()      # every parameter the backward writes (unused ones keep .grad = None)
    fg = net.__dict__.get("_md_flat_grads")
    if fg is None or not fg.matches(order):
        fg = FlatGrads(order)
        net.__dict__["_md_flat_grads"] = fg
    return fg


class GradReducer:
    """Bucket-wise gradient averaging that overlaps with the backward pass.

    The HIP backward (`DDPMUNet3D.backward`) walks the layers in reverse and calls `ready(params)` as soon as a
    layer's parameter gradients are final.  With `flat` (a FlatGrads whose layout IS that completion order) the
    finished gradients form a growing prefix of one buffer: every `cap_bytes` the new part of the prefix goes out as
    ONE asynchronous in-place all-reduce (RCCL runs it on its own stream behind the kernels already queued, so it rides
    under the remaining backward; `ReduceOp.AVG` on RCCL, SUM + one in-place division on gloo).  `finish(all_params)`
    sends the rest (FiLM / timestep-MLP / stem gradients are only complete at the very end) and waits.  Without `flat`
    (gradients allocated separately) buckets are flattened copies that are copied back after the wait.  xGMI rings are
    per-link bound, so buckets are large (default 128 MB, 12 messages for res64) rather than DDP's 25 MB.
    World size 1: every call is a no-op (set `force=True` to run the collectives anyway: world-1 RCCL self-test).
    """

    def __init__(self, cap_bytes=128 << 20, flat=None, force=False):
        self.cap = int(cap_bytes)
        self.active = dist.is_initialized() and (dist.get_world_size() > 1 or force or FORCE_EXCHANGE)
        self.flat = flat
        self.cur, self.cur_bytes, self.pending, self.done = [], 0, [], set()
        self.frontier = self.sent = 0          # flat mode: params [0, frontier) are final, elements [0, sent) are on the wire
        self.sent_idx = 0
        self.stats = dict(buckets=0, bytes=0)
        self.avg = None
        if self.active:
            self.avg = dist.ReduceOp.AVG if dist.get_backend() == "nccl" else None   # gloo has no AVG

    def _allreduce(self, t):
        self.stats["buckets"] += 1
        self.stats["bytes"] += t.numel() * t.element_size()
        return dist.all_reduce(t, op=self.avg if self.avg is not None else dist.ReduceOp.SUM, async_op=True)

    def ready(self, params):
        if not self.active:
            return
        if self.flat is not None:
            fl = self.flat
            for p in params:
                if id(p) in fl.slices:
                    self.done.add(id(p))
            while self.frontier < len(fl.params) and id(fl.params[self.frontier]) in self.done:
                self.frontier += 1
            end = fl.n if self.frontier == len(fl.params) else fl.slices[id(fl.params[self.frontier])][1]
            if (end - self.sent) * 4 >= self.cap:
                self._launch_flat(end)
            return
        for p in params:
            if p.requires_grad and p.grad is not None and id(p) not in self.done:
                self.done.add(id(p))
                self.cur.append(p.grad)
                self.cur_bytes += p.grad.numel() * p.grad.element_size()
        if self.cur_bytes >= self.cap:
            self._launch()

    def _launch_flat(self, end):
        if end > self.sent:
            sl = self.flat.flat[self.sent:end]
            self.pending.append((self._allreduce(sl), sl, None))
            self.sent = end

    def _launch(self):
        if not self.cur:
            return
        flat = torch._utils._flatten_dense_tensors(self.cur)
        self.pending.append((self._allreduce(flat), flat, self.cur))
        self.cur, self.cur_bytes = [], 0

    def finish(self, all_params):
        if not self.active:
            return
        world = dist.get_world_size()
        if self.flat is not None:
            self._launch_flat(self.flat.n)
            # parameters outside the flat buffer (none for the registered models) take the copying path
            self.flat, rest = None, [p for p in all_params if id(p) not in self.flat.slices]
            self.ready(rest)
        else:
            self.ready(list(all_params))
        self._launch()
        for work, flat, bucket in self.pending:
            work.wait()
            if self.avg is None:
                flat.div_(world)
            if bucket is not None:
                for g, f in zip(bucket, torch._utils._unflatten_dense_tensors(flat, bucket)):
                    g.copy_(f)
        self.pending = []
