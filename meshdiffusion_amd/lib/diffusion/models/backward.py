"""Backward pass of the U-Net layers on the HIP path ("first correct version").

Every contraction re-uses md_gemm_conv:
  * dgrad of a conv / NIN  = the forward kernels on WPK tiles packed from the flipped + transposed weight;
  * wgrad                  = split-K GEMM over (position, sample) on PB16 operands (csrc/backward.hip), one
                             B-pointer offset per tap, the three dx taps of a (dz, dy) row batched per launch;
GroupNorm/SiLU backward, bias sums and gradient resampling are streaming kernels.  Glue that is not on the
FLOP/byte path (slicing a concatenated gradient, adding two gradients, the [B,512] timestep-MLP algebra) uses
torch tensor ops on the device.

Gradients are F32B tensors shaped like the forward activations; parameter gradients accumulate into `.grad`.
Batch must be a multiple of 8 (the PB16 contraction blocks 8 samples).
"""
import torch

from .... import _lib
from .... import hip_ops as ops
from ....hip_ops import _ptr, _stream, check

GUARD_EXTRA = 8


def _grad_of(p):
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def split_f32b(t, B, C, P, fp16=False):
    """F32B -> S16B operand (bf16 split) of a gradient / activation tensor."""
    return ops.gn_apply([(t, C)], None, B, P, norm=False, silu=False, fp16=fp16)


# ---------------------------------------------------------------------------------------------------------
# PB16 operands and the wgrad GEMM
# ---------------------------------------------------------------------------------------------------------
def _guard(S):
    g = (S + 2) * (S + 2) + (S + 2) + 1 + GUARD_EXTRA
    return ((g + 3) // 4) * 4


def to_pb16(src, B, C, S, mode, up=0, stuff=0):
    """src: F32B (mode 0) or S16B (mode 1) on an S^3 grid (or (S/2)^3 when up/stuff) -> PB16 on the padded S^3 grid."""
    lib = _lib.load()
    g = _guard(S)
    nbytes = lib.md_pb16_bytes(B, C, S, S, S, g)
    if nbytes <= 0:
        raise _lib.MeshDiffusionHipError("md_pb16_bytes failed (batch must be a multiple of 8)")
    out = torch.empty(nbytes // 2 + 4 * 2 * C * 8 * (B // 8), dtype=torch.bfloat16, device=src.device)
    out[nbytes // 2:].zero_()      # tail so that K rounded up to 4 positions stays in bounds
    check(lib.md_to_pb16(_ptr(src), _ptr(out), B, C, S, S, S, g, mode, up, stuff, _stream()), "md_to_pb16")
    return out


def wgrad(dy_pb, act_pb, B, co, ci, S, taps, dw, s_row, s_k, s_tap):
    """dw[co][ci][tap] += sum_{pos,b} dy[co][pos,b] * act[ci][pos + off(tap), b]   (taps = 27 or 1)."""
    lib = _lib.load()
    g = _guard(S)
    Sp = S + 2
    Pp = Sp ** 3
    kpos = ((Pp + 3) // 4) * 4
    bg = B // 8
    kdim = kpos * bg * 8
    a_pos = bg * 2 * co * 8          # bf16 elements per position of the A (dy) operand
    b_pos = bg * 2 * ci * 8
    cfg = ops.CFG_G1_128_LOW
    assert ci % 64 == 0, "wgrad needs the activation channel count to be a multiple of 64 (pad with zeros)"
    rows8 = ((co + 7) // 8) * 8
    tiles = (ci // 64) * ((co + 127) // 128)
    ksplit = 1
    while tiles * 3 * ksplit < 1024 and ksplit < 256:
        ksplit *= 2
    a_base = dy_pb[g * a_pos:]
    groups = [(dz, dyy) for dz in range(3) for dyy in range(3)] if taps == 27 else [(1, 1)]
    nb = 3 if taps == 27 else 1
    for dz, dyy in groups:
        off = ((dz - 1) * Sp + (dyy - 1)) * Sp + (-1 if taps == 27 else 0)
        b_base = act_pb[(g + off) * b_pos:]
        out = torch.empty((nb, rows8 // 8, ci, 8), dtype=torch.float32, device=dy_pb.device)
        ops.gemm_conv(cfg=cfg, a=a_base, b=b_base, out=out, batch=nb, rows=co, rows_alloc=rows8, kdim=kdim,
                      dims=(1, 1, ci), a_src=ops.A_S16B, a_rows=co, a_bstride=0, b_bstride=b_pos, ksplit=ksplit)
        tap0 = (dz * 3 + dyy) * 3 if taps == 27 else 0
        check(lib.md_wgrad_finish(_ptr(out), _ptr(dw), co, ci, ci, nb, tap0, s_row, s_k, s_tap, _stream()),
              "md_wgrad_finish")


def channel_sums(t, B, C, P):
    lib = _lib.load()
    out = torch.zeros((B, C), dtype=torch.float32, device=t.device)
    check(lib.md_channel_sums(_ptr(t), _ptr(out), B, C, P, _stream()), "md_channel_sums")
    return out


def resample(t, B, C, Sc, mode, accumulate_into=None):
    """mode 0: fine (2Sc)^3 -> coarse Sc^3 sum of children; mode 1: coarse -> fine zero-stuffed at odd positions."""
    lib = _lib.load()
    Pc = Sc ** 3
    if mode == 0:
        out = accumulate_into if accumulate_into is not None else ops.f32b_empty(B, C, Pc, t.device)
    else:
        out = ops.f32b_empty(B, C, Pc * 8, t.device)
    check(lib.md_grad_resample(_ptr(t), _ptr(out), B, C, Sc, Sc, Sc, mode, 1 if accumulate_into is not None else 0,
                               _stream()), "md_grad_resample")
    return out


# ---------------------------------------------------------------------------------------------------------
# GroupNorm (+SiLU) backward over concatenated parts
# ---------------------------------------------------------------------------------------------------------
def gn_backward(parts, dy, params, gn, B, P, silu, d_into=None):
    """parts: forward inputs [(F32B, C)]; dy: F32B [B][Ctot][P]; returns one F32B [B][Ctot][P] gradient
    (written into / accumulated onto `d_into` when given) and accumulates gn.weight/.bias grads."""
    lib = _lib.load()
    dev = dy.device
    ctot = sum(c for _, c in parts)
    sums = torch.zeros((B, ctot, 2), dtype=torch.float64, device=dev)
    off = 0
    for t, c in parts:
        check(lib.md_gn_bwd_stats(_ptr(t), _ptr(dy), _ptr(params), _ptr(sums), B, c, P, ctot, off, ctot, 1 if silu else 0,
                                  _stream()), "md_gn_bwd_stats")
        off += c
    coef = torch.empty((B, ctot, 4), dtype=torch.float32, device=dev)
    check(lib.md_gn_bwd_finalize(_ptr(sums), _ptr(params), _ptr(gn.weight), _ptr(coef), _ptr(_grad_of(gn.weight)),
                                 _ptr(_grad_of(gn.bias)), B, ctot, gn.num_groups, P, _stream()), "md_gn_bwd_finalize")
    acc = d_into is not None
    outs = []
    off = 0
    dxcat = d_into if acc else ops.f32b_empty(B, ctot, P, dev)
    for t, c in parts:
        # each part's gradient is written at its channel offset of the concatenated gradient tensor
        dx_view = dxcat.view(B, ctot // 8, P, 8)[:, off // 8:(off + c) // 8]
        if len(parts) == 1:
            check(lib.md_gn_bwd_apply(_ptr(t), _ptr(dy), _ptr(params), _ptr(coef), _ptr(dxcat), B, c, P, ctot, 0, ctot,
                                      1 if silu else 0, 1 if acc else 0, _stream()), "md_gn_bwd_apply")
        else:
            tmp = dx_view.contiguous() if acc else ops.f32b_empty(B, c, P, dev)
            check(lib.md_gn_bwd_apply(_ptr(t), _ptr(dy), _ptr(params), _ptr(coef), _ptr(tmp), B, c, P, ctot, off, ctot,
                                      1 if silu else 0, 1 if acc else 0, _stream()), "md_gn_bwd_apply")
            outs.append(tmp)
        off += c
    if len(parts) == 1:
        return [dxcat]
    return outs


# ---------------------------------------------------------------------------------------------------------
# convolution / NIN backward
# ---------------------------------------------------------------------------------------------------------
def dgrad_weight(layer, name, conv, cfg):
    """WPK tiles of the data-gradient conv: W'[ci][co][k] = W[co][ci][flip(k)]."""
    def build():
        w = conv.weight.detach().flip(2, 3, 4).transpose(0, 1).contiguous()
        return ops.PackedWeight(w, "conv", cfg, w.device)
    return layer._cached(f"{name}/dgrad{cfg}", [conv.weight], build)


def conv3_backward(layer, name, conv, dy, act_s16, B, S_out, ups=0, stride=1, need_dx=True, act_channels=None):
    """Backward of y = conv3x3x3(act) (+bias).  dy: F32B [B][co][S_out^3]; act_s16: S16B input operand of the
    forward (coarse grid when ups, fine grid 2*S_out when stride 2).  Returns dx (F32B) or None."""
    from . import layers
    co, ci = conv.weight.shape[0], conv.weight.shape[1]
    P = S_out ** 3
    dev = dy.device
    # bias
    _grad_of(conv.bias).add_(channel_sums(dy, B, co, P).sum(0))
    # weight gradient
    S_fine = S_out * stride
    if stride == 2:
        dy_pb = to_pb16(dy, B, co, S_fine, 0, stuff=1)
    else:
        dy_pb = to_pb16(dy, B, co, S_out, 0)
    ci_pad = act_channels if act_channels is not None else ci
    act_pb = to_pb16(act_s16, B, ci_pad, S_fine, 1, up=ups)
    dw = _grad_of(conv.weight)
    if ci_pad != ci:      # stem: the operand was zero padded to 64 channels; accumulate into a padded scratch
        scratch = torch.zeros((co, ci_pad, 27), dtype=torch.float32, device=dev)
        wgrad(dy_pb, act_pb, B, co, ci_pad, S_fine, 27, scratch, ci_pad * 27, 27, 1)
        dw.add_(scratch[:, :ci].reshape(dw.shape))
    else:
        wgrad(dy_pb, act_pb, B, co, ci, S_fine, 27, dw, ci * 27, 27, 1)
    del dy_pb, act_pb
    if not need_dx:
        return None
    # data gradient: same conv kernels, flipped/transposed weights
    if stride == 2:
        dyz = resample(dy, B, co, S_out, 1)                       # fine grid, dy at odd positions
        cfg = ops.conv_cfg_for(S_fine)
        pw = dgrad_weight(layer, name, conv, cfg)
        return layers.run_conv3(pw, split_f32b(dyz, B, co, S_fine ** 3), B, S_fine)
    cfg = ops.conv_cfg_for(S_out)
    if cfg == ops.CFG_C3_128_FAST and co % 32 != 0:
        cfg = ops.CFG_C3_128_K16 if co <= 16 else cfg
    pw = dgrad_weight(layer, name, conv, cfg)
    dyc = dy
    co_k = co
    if cfg == ops.CFG_C3_128_K16:          # head: dy has 4 (padded to 8) channels -> K padded to 16
        dy16 = torch.zeros((B, 2, P, 8), dtype=torch.float32, device=dev)
        dy16[:, :dy.shape[1]] = dy
        dyc, co_k = dy16, 16
    dx = layers.run_conv3(pw, split_f32b(dyc, B, co_k, P), B, S_out)
    if ups:
        dx = resample(dx, B, ci, S_out // 2, 0)
    return dx


def nin_backward(nin, dy, xs_s16, B, P, S, need_dx=True):
    """Backward of y[co] = sum_ci x[ci] W[ci][co] + b.  xs_s16: S16B of the forward input."""
    from . import layers
    ci, co = nin.W.shape
    _grad_of(nin.b).add_(channel_sums(dy, B, co, P).sum(0))
    dy_pb = to_pb16(dy, B, co, S, 0)
    x_pb = to_pb16(xs_s16, B, ci, S, 1)
    wgrad(dy_pb, x_pb, B, co, ci, S, 1, _grad_of(nin.W), 1, co, 0)
    del dy_pb, x_pb
    if not need_dx:
        return None
    cfg = ops.gemm_cfg_for(P, ci)
    pw = nin._cached(f"dgrad{cfg}", [nin.W], lambda: ops.PackedWeight(nin.W, "rows", cfg, nin.W.device))
    return layers.run_gemm(pw, split_f32b(dy, B, co, P), B, P)
