"""DDPM loss, step function and optimisation manager -- mirror of the reference's
lib/diffusion/losses.py (`get_optimizer` :26-35, `optimization_manager` :38-52, `get_ddpm_loss_fn` :54-85,
`get_step_fn` :87-141).

On the HIP path: forward noising, the masked loss and its gradient w.r.t. eps_hat, the U-Net forward AND
backward (models/backward.py), the global grad-norm and the fused clip + Adam + EMA update.  `loss.backward()`
works because the U-Net and the loss are each one opaque autograd node whose backward is the HIP code; the
reference's `optimize_fn` (torch.optim.Adam + clip_grad_norm_) can then be used unchanged, or `FusedAdamEMA`.
Dropout (ResnetBlockDDPM, p = config.model.dropout) runs inside the GroupNorm+SiLU kernel from a counter-based
mask that the backward regenerates.  Both architectures and any per-GPU batch (the wgrad blocks samples by 8: a partial block is zero-filled).
"""
import ctypes as C

import numpy as np
import torch
import torch.optim as optim

from . import parallel
from .models import utils as mutils
from .sde_lib import VPSDE
from ... import _lib
from ... import hip_ops as ops


def get_optimizer(config, params):
    if config.optim.optimizer != "Adam":
        raise NotImplementedError(f"Optimizer {config.optim.optimizer} not supported yet!")
    return optim.Adam(params, lr=config.optim.lr, betas=(config.optim.beta1, 0.999), eps=config.optim.eps,
                      weight_decay=config.optim.weight_decay)


def optimization_manager(config):
    def optimize_fn(optimizer, params, step, lr=config.optim.lr, warmup=config.optim.warmup,
                    grad_clip=config.optim.grad_clip):
        if warmup > 0:
            for g in optimizer.param_groups:
                g["lr"] = lr * np.minimum(step / warmup, 1.0)
        if grad_clip >= 0:
            torch.nn.utils.clip_grad_norm_(params, max_norm=grad_clip)
        optimizer.step()

    optimize_fn._md_hparams = True     # (lr, warmup, grad_clip) = optimize_fn.__defaults__: lets the step function fuse it
    return optimize_fn


FUSED_OPT = __import__("os").environ.get("MD_FUSED_OPT", "1") == "1"


class _FlatOptState:
    """torch.optim.Adam + clip_grad_norm_ + ExponentialMovingAverage.update as the two launches of md_grad_sqnorm /
    md_adam_ema_step over flat buffers -- WITHOUT changing what the caller holds: parameters, Adam's `exp_avg` /
    `exp_avg_sq` and the EMA's `shadow_params` become views of flat buffers laid out like parallel.FlatGrads, so
    `optimizer.state_dict()`, `ema.state_dict()` and therefore the reference checkpoint format keep working, and a
    `load_state_dict` (new tensors) is detected and re-imported on the next step."""

    def __init__(self, fg, optimizer, ema):
        self.fg = fg
        dev = fg.flat.device
        self.p = torch.empty_like(fg.flat)
        self.m = torch.zeros_like(fg.flat)
        self.v = torch.zeros_like(fg.flat)
        self.e = torch.empty_like(fg.flat)
        self.sq = torch.zeros(1, dtype=torch.float64, device=dev)
        # one 0-dim `step` tensor PER parameter (torch.optim.Adam advances each entry's step in place: a tensor shared by
        # all entries would be advanced once per parameter after a resume through the torch optimizer)
        self.step_ts = [torch.zeros((), dtype=torch.float32) for _ in fg.params]
        self.ema_index = None
        for p in fg.params:                                          # parameters -> views of the flat buffer
            _, o, n = fg.slices[id(p)]
            self.p[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.p[o:o + n].view_as(p)
        ops.bump_param_epoch()
        self.opt_steps = 0
        self._import(optimizer, ema)

    def _views(self, flat):
        return [flat[o:o + n].view_as(p) for p, (_, o, n) in ((p, self.fg.slices[id(p)]) for p in self.fg.params)]

    def _import(self, optimizer, ema):
        """Adopt whatever state the torch objects currently hold (fresh, or just loaded from a checkpoint)."""
        mv, vv = self._views(self.m), self._views(self.v)
        self.opt_steps = 0
        for p, m_, v_, s_ in zip(self.fg.params, mv, vv, self.step_ts):
            st = optimizer.state.get(p)
            if st and "exp_avg" in st:
                m_.copy_(st["exp_avg"]); v_.copy_(st["exp_avg_sq"])
                self.opt_steps = int(float(st["step"]))
            else:
                m_.zero_(); v_.zero_()
            optimizer.state[p] = dict(step=s_, exp_avg=m_, exp_avg_sq=v_)
        for s_ in self.step_ts:
            s_.fill_(float(self.opt_steps))
        live = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        idx = {id(p): i for i, p in enumerate(live)}
        ev = self._views(self.e)
        for p, e_ in zip(self.fg.params, ev):
            i = idx[id(p)]
            e_.copy_(ema.shadow_params[i])
            ema.shadow_params[i] = e_
        self.m0, self.e0, self.ema_i0 = mv[0], ev[0], idx[id(self.fg.params[0])]
        self.live_key = (id(optimizer), id(ema), len(live), len(ema.shadow_params))

    def params_alias_flat(self):
        """The first and the last parameter still are views of the flat buffer (model.to() / float() / a manual
        `p.data = ...` / load_state_dict(assign=True) replace the storage: md_adam_ema_step would then update an orphan)."""
        for p in (self.fg.params[0], self.fg.params[-1]):
            _, o, n = self.fg.slices[id(p)]
            if p.data_ptr() != self.p.data_ptr() + 4 * o or p.numel() != n:
                return False
        return True

    def usable(self, optimizer, ema):
        if not self.params_alias_flat():
            return False                          # the caller rebuilds the flat state from the live parameters
        p0 = self.fg.params[0]
        st = optimizer.state.get(p0)
        if not st or st.get("exp_avg") is None or st["exp_avg"].data_ptr() != self.m0.data_ptr() \
                or ema.shadow_params[self.ema_i0].data_ptr() != self.e0.data_ptr():
            self._import(optimizer, ema)          # state was replaced (load_state_dict / restore_checkpoint)
        return True

    def step(self, optimizer, ema, sched_step, lr0, warmup, grad_clip):
        lib = _lib.load()
        g = optimizer.param_groups[0]
        lr = float(lr0 * np.minimum(sched_step / warmup, 1.0)) if warmup > 0 else float(g["lr"])
        g["lr"] = lr
        self.opt_steps += 1
        torch._foreach_add_(self.step_ts, 1.0)
        d = ema._effective_decay()
        sq = None
        if grad_clip >= 0:
            self.sq.zero_()
            _lib.check(lib.md_grad_sqnorm(ops._ptr(self.fg.flat), self.fg.n, ops._ptr(self.sq), ops._stream()), "md_grad_sqnorm")
            sq = self.sq
        b1, b2 = g["betas"]
        _lib.check(lib.md_adam_ema_step(ops._ptr(self.p), ops._ptr(self.fg.flat), ops._ptr(self.m), ops._ptr(self.v),
                                        ops._ptr(self.e), self.fg.n, lr, float(b1), float(b2), float(g["eps"]),
                                        float(g["weight_decay"]), self.opt_steps, float(d), ops._ptr(sq), float(grad_clip),
                                        ops._stream()), "md_adam_ema_step")
        ops.bump_param_epoch()   # raw-pointer update: packed-weight caches are keyed on data_ptr/_version


def _fused_opt_for(net, fg, optimizer, ema, optimize_fn):
    """The _FlatOptState of `net` when the step's optimizer / EMA are the reference configuration it can replace:
    plain torch.optim.Adam (one group, no amsgrad / maximize) over exactly the parameters of `fg`, our EMA class and an
    optimize_fn made by optimization_manager.  Otherwise None (the torch objects run as they are)."""
    from .models.ema import ExponentialMovingAverage
    if not (FUSED_OPT and fg is not None and getattr(optimize_fn, "_md_hparams", False)):
        return None
    if type(optimizer) is not optim.Adam or len(optimizer.param_groups) != 1 or not isinstance(ema, ExponentialMovingAverage):
        return None
    g = optimizer.param_groups[0]
    if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
        return None
    live = [p for p in g["params"] if p.requires_grad]
    fs = net.__dict__.get("_md_flat_opt")
    key = (id(optimizer), id(ema), len(live), len(ema.shadow_params))
    if fs is None or fs.fg is not fg or fs.live_key != key:      # eligibility is re-derived only when an object changed
        live_ids = {id(q) for q in live}
        if len(ema.shadow_params) != len(live) or any(id(p) not in live_ids for p in fg.params):
            return None
        fs = None
    if any(p.grad is not None for p in live if id(p) not in fg.slices):
        return None                                  # a parameter outside the flat buffer received a gradient
    if fs is None or not fs.usable(optimizer, ema):
        fs = _FlatOptState(fg, optimizer, ema)
        net.__dict__["_md_flat_opt"] = fs
        fs.usable(optimizer, ema)
    return fs


def ddpm_perturb(vpsde, batch, labels, noise, mask_flat):
    """x_t = (sqrt(abar_l) x0 + sqrt(1-abar_l) noise) * mask  -- md_ddpm_perturb."""
    lib = _lib.load()
    dev = batch.device
    coef = torch.stack([vpsde.sqrt_alphas_cumprod.to(dev)[labels], vpsde.sqrt_1m_alphas_cumprod.to(dev)[labels]],
                       dim=1).contiguous()
    batch, noise = batch.contiguous(), noise.contiguous()
    out = torch.empty_like(batch)
    B, Cc, P = batch.shape[0], batch.shape[1], batch[0, 0].numel()
    _lib.check(lib.md_ddpm_perturb(ops._ptr(batch), ops._ptr(noise), ops._ptr(mask_flat), ops._ptr(coef), ops._ptr(out),
                                   B, Cc, P, ops._stream()), "md_ddpm_perturb")
    return out


def masked_sq_err(eps_hat, noise, mask_flat, want_grad=False, gscale=1.0):
    """Per-sample sum of (eps_hat-noise)^2 * mask in fp64 (+ optional dLoss/d eps_hat) -- md_masked_sq_err."""
    lib = _lib.load()
    B, Cc, P = eps_hat.shape[0], eps_hat.shape[1], eps_hat[0, 0].numel()
    sums = torch.zeros(B, dtype=torch.float64, device=eps_hat.device)
    eps_hat, noise = eps_hat.contiguous(), noise.contiguous()   # named: the copies must outlive the launch
    grad = torch.empty_like(eps_hat) if want_grad else None
    _lib.check(lib.md_masked_sq_err(ops._ptr(eps_hat), ops._ptr(noise), ops._ptr(mask_flat),
                                    ops._ptr(sums), ops._ptr(grad), float(gscale), B, Cc, P, ops._stream()),
               "md_masked_sq_err")
    return sums, grad


def get_ddpm_loss_fn(vpsde, train, mask=None, loss_type="l2"):
    if loss_type != "l2":
        raise NotImplementedError("only the l2 loss of the configured path is implemented")
    assert isinstance(vpsde, VPSDE)

    def loss_fn(model, batch):
        model_fn = mutils.get_model_fn(model, train=train)
        labels = torch.randint(0, vpsde.N, (batch.shape[0],), device=batch.device)
        noise = torch.randn_like(batch)
        mask_flat = mask.reshape(-1).to(batch.device, torch.float32).contiguous() if mask is not None else None
        perturbed = ddpm_perturb(vpsde, batch, labels, noise, mask_flat)
        score = model_fn(perturbed, labels)
        norm = 1.0
        if mask is not None:
            norm = float(np.prod(mask.size())) / float(mask.sum())
        return _MaskedDDPMLoss.apply(score, noise, mask_flat, norm)

    return loss_fn


class _MaskedDDPMLoss(torch.autograd.Function):
    """loss = mean_b( mean_{c,p}( (score-noise)^2 * mask ) ) * norm, with norm = mask.numel()/mask.sum()
    (losses.py:68-78).  Forward and the gradient w.r.t. `score` come from one md_masked_sq_err launch."""

    @staticmethod
    def forward(ctx, score, noise, mask_flat, norm):
        B = score.shape[0]
        per = float(score[0].numel())
        want = score.requires_grad
        sums, grad = masked_sq_err(score.detach(), noise, mask_flat, want_grad=want, gscale=norm / (B * per))
        ctx.grad = grad
        loss = (sums / per).to(torch.float32).mean() * norm
        return loss

    @staticmethod
    def backward(ctx, g):
        return (ctx.grad * g if ctx.grad is not None else None), None, None, None


def get_step_fn(sde, train, optimize_fn=None, mask=None, loss_type="l2"):
    loss_fn = get_ddpm_loss_fn(sde, train, mask=mask, loss_type=loss_type)

    def step_fn(state, batch, clear_grad=True, update_param=True):
        model = state["model"]
        tm = state.get("timers")              # bench.py: a dict -> device-synchronised phase times of this step (ms)
        clock = [0.0]

        def mark(name):
            if tm is not None:
                import time
                torch.cuda.synchronize()
                now = time.perf_counter()
                if name is not None:
                    tm[name] = (now - clock[0]) * 1e3
                clock[0] = now
        mark(None)
        if train:
            optimizer = state["optimizer"]
            net = getattr(model, "module", model)
            # gradients live in ONE flat buffer laid out in backward-completion order (parallel.FlatGrads): the
            # exchange below all-reduces slices of it in place, Adam and the clip read the views
            fg = parallel.flat_grads_for(net) if hasattr(net, "grad_completion_order") else None
            if clear_grad:
                optimizer.zero_grad()
                if fg is not None:
                    fg.zero_()
            if fg is not None:
                fg.attach()
            loss = loss_fn(model, batch)
            mark("fwd_loss")
            # one process per GPU: the replicas' gradients meet here.  On the step that updates the parameters the
            # U-Net backward announces finished layers to the reducer, whose bucket all-reduces overlap the rest of
            # the backward (no-op for a single process).
            reducer = parallel.GradReducer(flat=fg) if update_param else None
            if reducer is not None and reducer.active:
                net._grad_ready_hook = reducer.ready
            try:
                loss.backward()
            finally:
                if hasattr(net, "_grad_ready_hook"):
                    del net._grad_ready_hook
            mark("bwd")                                # includes the bucket all-reduces that ran under it
            if update_param:
                reducer.finish(model.parameters())
                mark("exchange_exposed")               # what the backward did not hide
                state["exchange"] = reducer.stats      # buckets / bytes of this step's all-reduces (bench.py reports them)
                fs = _fused_opt_for(net, fg, optimizer, state["ema"], optimize_fn)
                if fs is not None:      # clip + Adam + EMA: two launches over the flat buffers (same arithmetic)
                    fs.step(optimizer, state["ema"], state["step"], *optimize_fn.__defaults__)
                    state["step"] += 1
                    mark("clip_adam_ema")
                    return {"loss": loss}
                optimize_fn(optimizer, model.parameters(), step=state["step"])
            state["step"] += 1
            state["ema"].update(model.parameters())
            mark("clip_adam_ema")
        else:
            with torch.no_grad():
                ema = state["ema"]
                ema.store(model.parameters())
                ema.copy_to(model.parameters())
                loss = loss_fn(model, batch)
                ema.restore(model.parameters())
        return {"loss": loss}

    return step_fn


class FusedAdamEMA:
    """clip_grad_norm_ + torch.optim.Adam + ExponentialMovingAverage.update as TWO kernel launches over flat
    buffers (md_grad_sqnorm, md_adam_ema_step) instead of ~12 passes over 1.46 GB of state.

    Parameters are re-pointed at views of one flat fp32 buffer (so are their .grad), m / v / ema are flat too.
    `lr` warm-up follows optimization_manager: lr * min(step / warmup, 1).
    """

    def __init__(self, params, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, ema_decay=0.9999,
                 grad_clip=1.0, warmup=5000):
        self.params = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        if dev.type != "cuda":
            raise _lib.MeshDiffusionHipError("FusedAdamEMA runs on the GPU only")
        self.sizes = [p.numel() for p in self.params]
        self.n = sum(self.sizes)
        self.flat = torch.empty(self.n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.n, dtype=torch.float32, device=dev)
        off = 0
        for p, n in zip(self.params, self.sizes):
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p)
            p.grad = self.grad[off:off + n].view_as(p)
            off += n
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.ema = self.flat.clone()
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, beta1, beta2, eps, weight_decay
        self.ema_decay, self.grad_clip, self.warmup = ema_decay, grad_clip, warmup
        self.opt_steps = 0       # optimizer steps taken (Adam bias correction)
        self.ema_updates = 0     # EMA updates (decay warm-up min(d, (1+n)/(10+n)))
        self._sq = torch.zeros(1, dtype=torch.float64, device=dev)

    def zero_grad(self):
        self.grad.zero_()

    def step(self, sched_step):
        """`sched_step` = state['step'] before the increment (reference: lr warm-up uses it)."""
        lib = _lib.load()
        lr = self.lr * float(np.minimum(sched_step / self.warmup, 1.0)) if self.warmup > 0 else self.lr
        self.opt_steps += 1
        self.ema_updates += 1
        d = min(self.ema_decay, (1 + self.ema_updates) / (10 + self.ema_updates))
        sq = None
        if self.grad_clip >= 0:
            self._sq.zero_()
            _lib.check(lib.md_grad_sqnorm(ops._ptr(self.grad), self.n, ops._ptr(self._sq), ops._stream()), "md_grad_sqnorm")
            sq = self._sq
        _lib.check(lib.md_adam_ema_step(ops._ptr(self.flat), ops._ptr(self.grad), ops._ptr(self.m), ops._ptr(self.v),
                                        ops._ptr(self.ema), self.n, lr, self.b1, self.b2, self.eps, self.wd,
                                        self.opt_steps, d, ops._ptr(sq), float(self.grad_clip), ops._stream()),
                   "md_adam_ema_step")
        ops.bump_param_epoch()   # packed-weight caches must be rebuilt: raw-pointer updates do not bump _version

    def _views(self, flat):
        out, off = [], 0
        for p, n in zip(self.params, self.sizes):
            out.append(flat[off:off + n].view_as(p))
            off += n
        return out

    def state_dict(self):
        """`torch.optim.Adam.state_dict()`-compatible (state[i] = {step, exp_avg, exp_avg_sq} per parameter, one param
        group), so a checkpoint written from this optimizer restores into the reference's `optim.Adam` and back."""
        m, v = self._views(self.m), self._views(self.v)
        state = {i: dict(step=torch.tensor(float(self.opt_steps)), exp_avg=m[i].clone(), exp_avg_sq=v[i].clone())
                 for i in range(len(self.params))} if self.opt_steps > 0 else {}
        group = dict(lr=self.lr, betas=(self.b1, self.b2), eps=self.eps, weight_decay=self.wd, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(self.params))))
        return dict(state=state, param_groups=[group])

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, (self.b1, self.b2), self.eps, self.wd = g["lr"], g["betas"], g["eps"], g["weight_decay"]
        st = sd["state"]
        self.opt_steps = 0
        self.m.zero_(); self.v.zero_()
        for i, (mv, vv) in enumerate(zip(self._views(self.m), self._views(self.v))):
            e = st.get(i, st.get(str(i)))
            if e is None:
                continue
            mv.copy_(e["exp_avg"]); vv.copy_(e["exp_avg_sq"])
            self.opt_steps = int(float(e["step"]))

    def ema_state_dict(self):
        """The reference EMA's state dict {decay, num_updates, shadow_params} (models/ema.py:91-97)."""
        return dict(decay=self.ema_decay, num_updates=self.ema_updates, shadow_params=[t.clone() for t in self.ema_shadow_params()])

    def load_ema_state_dict(self, sd):
        self.ema_decay, self.ema_updates = sd["decay"], int(sd["num_updates"] or 0)
        for dst, src in zip(self.ema_shadow_params(), sd["shadow_params"]):
            dst.copy_(src)

    def ema_shadow_params(self):
        """List of views in parameters() order == the reference EMA's `shadow_params` (checkpoint format)."""
        out, off = [], 0
        for p, n in zip(self.params, self.sizes):
            out.append(self.ema[off:off + n].view_as(p))
            off += n
        return out


_ = C
