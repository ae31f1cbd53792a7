"""Backward pass of the U-Net layers on the HIP path ("first correct version").

Every contraction re-uses md_gemm_conv:
  * dgrad of a conv / NIN  = the forward kernels on WPK tiles packed from the flipped + transposed weight;
  * wgrad                  = md_wgrad (csrc/wgrad.hip): a dedicated MFMA kernel contracting over (position, sample)
                             on PB16 operands, 128 co x 128 ci x 3 dx taps per workgroup, deterministic split-K;
GroupNorm/SiLU backward, bias sums and gradient resampling are streaming kernels.  Glue that is not on the
FLOP/byte path (slicing a concatenated gradient, adding two gradients, the [B,512] timestep-MLP algebra) uses
torch tensor ops on the device.

Gradients are F32B tensors shaped like the forward activations; parameter gradients accumulate into `.grad`.
Any batch size works.  The wgrad contraction blocks 8 samples per MFMA k-group; a batch that is not a
multiple of 8 fills its blocks with z-slabs of each sample instead (`zsplit_for`: 8 / gcd(B, 8) slabs; the sum over
positions is a sum over slabs), so no MFMA work is spent on zero padding.
"""
import torch

from .... import _lib
from .... import hip_ops as ops
from ....hip_ops import _ptr, _stream, check

import os

GUARD_EXTRA = 12
# A/B switch: bias-gradient channel sums of dY from the dual operand pass (md_wino_prep_dual) instead of md_channel_sums
FUSE_DY_SUMS = os.environ.get("MD_FUSE_DY_SUMS", "1") == "1"
WGRAD_BLOCKS = 256     # one 512-thread wgrad workgroup per CU


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
def _guard(S, pad=1):
    sp = S + 2 * pad
    g = pad * (sp * sp + sp + 1) + GUARD_EXTRA
    return ((g + 3) // 4) * 4


def zsplit_for(B, S):
    """z-slabs per sample so that B * zsplit virtual samples fill whole 8-sample blocks of the wgrad operands:
    8 / gcd(B, 8) (1 for multiples of 8, 4 for B = 6, 8 for odd B), as far as the grid depth S divides."""
    import math
    zs = 8 // math.gcd(int(B), 8)
    while zs > 1 and S % zs:
        zs //= 2
    return zs


def to_pb16(src, B, C, S, mode, up=0, stuff=0, c_src=None, pad=1, zhalo=True):
    """src: F32B (mode 0) or S16B (mode 1) on an S^3 grid (or (S/2)^3 when up/stuff) -> PB16 on the padded S^3 grid.
    c_src: channels actually present in `src` (the PB16 tensor is zero for channels c_src..C-1).
    pad: halo of the padded grid = kernel size // 2 of the conv whose wgrad consumes it (1 for NIN).
    zhalo: True for an activation operand, False for a dY operand (see md_to_pb16: z-slab virtual samples)."""
    lib = _lib.load()
    g = _guard(S, pad)
    zs = zsplit_for(B, S)
    VB, Dz = B * zs, S // zs
    nbytes = lib.md_pb16_bytes(VB, C, Dz, S, S, g, pad)
    if nbytes <= 0:
        raise _lib.MeshDiffusionHipError("md_pb16_bytes failed")
    out = torch.empty(nbytes // 2 + 4 * 2 * C * 8 * ((VB + 7) // 8), dtype=torch.bfloat16, device=src.device)
    out[nbytes // 2:].zero_()      # tail so that K rounded up to 4 positions stays in bounds
    check(lib.md_to_pb16(_ptr(src), _ptr(out), B, C, C if c_src is None else c_src, Dz, S, S, g, pad, mode, up, stuff,
                         zs, 1 if zhalo else 0,
                         _stream()), "md_to_pb16")
    return out


def wgrad(dy_pb, act_pb, B, co, ci, S, taps, dw, s_row, s_k, s_tap, a_ch=None, b_ch=None):
    """dw[co*s_row + ci*s_k + tap*s_tap] += sum_{pos,b} dy[co][pos,b] * act[ci][pos + off(tap), b]  (taps = 125, 27 or 1)
    -- md_wgrad (csrc/wgrad.hip).  a_ch / b_ch: channel counts of the two PB16 tensors when they exceed co / ci."""
    lib = _lib.load()
    g = _guard(S, 2 if taps == 125 else 1)
    a_ch = co if a_ch is None else a_ch
    b_ch = ci if b_ch is None else b_ch
    zs = zsplit_for(B, S)
    VB, Dz = B * zs, S // zs
    stages = ((VB + 7) // 8) * Dz * S * ((S + 7) // 8)
    units = ((co + 127) // 128) * ((ci + 127) // 128) * {125: 50, 27: 9, 1: 1}[taps]
    ksplit = max(1, min(WGRAD_BLOCKS // units, stages // 4))
    nbytes = lib.md_wgrad_workspace_bytes(co, ci, taps, ksplit)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dy_pb.device)
    check(lib.md_wgrad(_ptr(dy_pb), _ptr(act_pb), _ptr(dw), _ptr(ws), nbytes, VB, a_ch, b_ch, co, ci, Dz, S, S, g, taps,
                       ksplit, s_row, s_k, s_tap, _stream()), "md_wgrad")


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
def gn_backward(parts, dy, params, gn, B, P, silu, d_into=None, drop=None, residual=None, sums_out=None, want_amax=False):
    """parts: forward inputs [(F32B, C)]; dy: F32B [B][Ctot][P]; returns one F32B [B][Ctot][P] gradient
    (written into / accumulated onto `d_into` when given) and accumulates gn.weight/.bias grads.
    drop=(p, seed): the forward applied dropout after the activation (same mask regenerated here).
    Single-part extras: residual (F32B): result = residual + gradient (identity shortcut, no separate copy);
    sums_out (zeroed float [B, Ctot]): += per-(sample, channel) sums of the gradient (bias / FiLM gradients);
    want_amax: the result carries `_md_amax`, the device word with its max |.| (hip_ops.absmax_word's, without the extra pass) -- for
    a consumer that is the f16f6 data-gradient conv and reads the tensor unmodified."""
    lib = _lib.load()
    dp, dseed = (float(drop[0]), int(drop[1])) if drop else (0.0, 0)
    dev = dy.device
    ctot = sum(c for _, c in parts)
    sums = torch.zeros((B, ctot, 2), dtype=torch.float64, device=dev)
    off = 0
    for t, c in parts:
        check(lib.md_gn_bwd_stats(_ptr(t), _ptr(dy), _ptr(params), _ptr(sums), B, c, P, ctot, off, ctot, 1 if silu else 0,
                                  dp, dseed, _stream()), "md_gn_bwd_stats")
        off += c
    coef = torch.empty((B, ctot, 4), dtype=torch.float32, device=dev)
    check(lib.md_gn_bwd_finalize(_ptr(sums), _ptr(params), _ptr(gn.weight), _ptr(coef), _ptr(_grad_of(gn.weight)),
                                 _ptr(_grad_of(gn.bias)), B, ctot, gn.num_groups, P, _stream()), "md_gn_bwd_finalize")
    acc = d_into is not None
    assert (residual is None and sums_out is None) or (len(parts) == 1 and not acc)
    outs = []
    off = 0
    dxcat = d_into if acc else ops.f32b_empty(B, ctot, P, dev)
    for t, c in parts:
        # each part's gradient is written at its channel offset of the concatenated gradient tensor
        dx_view = dxcat.view(B, ctot // 8, P, 8)[:, off // 8:(off + c) // 8]
        if len(parts) == 1:
            amax = ops.amax_slot(dev) if (want_amax and not acc) else None
            check(lib.md_gn_bwd_apply(_ptr(t), _ptr(dy), _ptr(params), _ptr(coef), _ptr(dxcat), B, c, P, ctot, 0, ctot,
                                      1 if silu else 0, 1 if acc else 0, dp, dseed, _ptr(residual), _ptr(sums_out), _ptr(amax),
                                      _stream()), "md_gn_bwd_apply")
            if amax is not None:
                dxcat._md_amax = amax
        else:
            tmp = dx_view.contiguous() if acc else ops.f32b_empty(B, c, P, dev)
            check(lib.md_gn_bwd_apply(_ptr(t), _ptr(dy), _ptr(params), _ptr(coef), _ptr(tmp), B, c, P, ctot, off, ctot,
                                      1 if silu else 0, 1 if acc else 0, dp, dseed, None, None, None, _stream()),
                  "md_gn_bwd_apply")
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
        return ops.PackedWeight(conv.weight, "conv_dgrad", cfg, conv.weight.device)
    return layer._cached(f"{name}/dgrad{cfg}", [conv.weight], build)


def dgrad_wino_weight(layer, name, conv):
    """Winograd tiles of the data-gradient conv (hip_ops.WinoWeight, kind "conv_dgrad")."""
    return layer._cached(f"{name}/dgrad_wino", [conv.weight],
                         lambda: ops.WinoWeight(conv.weight, conv.weight.device, kind="conv_dgrad"))


def dgrad_wino_weight_f6(layer, name, conv):
    """f16f6 fragments of the data-gradient conv (hip_ops.WinoWeightF6Dgrad: packed by the step's md_pack_batch table)."""
    return layer._cached(f"{name}/dgrad_wino_f6", [conv.weight], lambda: ops.WinoWeightF6Dgrad(conv.weight, conv.weight.device))


def conv3_backward(layer, name, conv, dy, act_s16, B, S_out, ups=0, stride=1, need_dx=True, act_channels=None,
                   bias_sums=None, shared=None, t_act=None, bias_sums_out=False):
    """Backward of y = conv k^3 (act) (+bias), k = 3 (any layer) or 5 (stem / head of ddpm_res128, stride 1).
    dy: F32B [B][co][S_out^3]; act_s16: S16B input operand of the forward (coarse grid when ups, fine grid 2*S_out
    when stride 2).  Returns dx (F32B) or None.
    shared: dict cache of tensors derived from `dy` (its PB16 operand, its S16B split) when another layer consumes the same
    gradient (the ResnetBlock's Conv_1 and shortcut NIN_0 both start from the block's output gradient).
    t_act: the forward conv's Winograd operand T (layers on the Winograd path, hip_ops.wgrad_wino_ok) in place of `act_s16`:
    ONE pass over dy (md_wino_prep_dual) feeds the Winograd data-gradient conv and the Winograd weight gradient (md_wgrad_wino);
    no S16B / PB16 tensors at all.  bias_sums_out (with t_act): `bias_sums` is a ZEROED [B, co] float tensor that this call
    fills (the operand pass over dy adds up its channels) for the caller's other consumers of the same sums."""
    from . import layers
    co, ci, ksz = conv.weight.shape[0], conv.weight.shape[1], conv.weight.shape[-1]
    taps, pad = ksz ** 3, ksz // 2
    assert ksz == 3 or (ksz == 5 and stride == 1 and not ups)
    P = S_out ** 3
    dev = dy.device
    co_t = dy.shape[1] * 8            # channels of the dy tensor (co rounded up to 8)
    # bias
    fused_sums = FUSE_DY_SUMS and t_act is not None and bias_sums is not False and (bias_sums is None or bias_sums_out)
    if bias_sums_out and not fused_sums:          # the caller handed a zeroed tensor to fill: fill it the separate way
        bias_sums.copy_(channel_sums(dy, B, co_t, P))
    if bias_sums is not False and not fused_sums:   # False: the caller owns the bias gradient (ResnetBlock Conv_0: FiLM shares the sums)
        bs = bias_sums if bias_sums is not None else channel_sums(dy, B, co_t, P)
        _grad_of(conv.bias).add_(bs.sum(0)[:co])
    if t_act is not None:
        assert ksz == 3 and stride == 1 and co == co_t
        bs = None
        if fused_sums:      # the operand pass over dy also adds up its channels: no md_channel_sums pass
            bs = bias_sums if bias_sums is not None else torch.zeros((B, co), dtype=torch.float32, device=dev)
        # round 6: the data-gradient conv in f16f6 (hip_ops.DGRAD_F6): the dual operand pass writes T in that format, lifted by a
        # power of two that follows the tensor's own maximum; U (the weight gradient's operand) and the channel sums are those of
        # the bf16x3 path, bit for bit
        f6 = ops.DGRAD_F6 and need_dx and co % 32 == 0 and ci % 128 == 0 and 256 % S_out == 0
        amax = None
        if f6:
            if ops.DGRAD_LIFT == "dyn":
                # the lift follows this tensor's own largest element (device-side: no host sync): the word its producer left with it
                # (gn_backward(want_amax=True): the tensor comes straight from md_gn_bwd_apply), or one more read of it
                amax = getattr(dy, "_md_amax", None)
                if amax is None:
                    amax = ops.absmax_word(dy)
            t_dy, u_dy = ops.wino_prep([(dy, co)], None, False, False, B, S_out, dual=True, sums=bs, f8="f6", tscale=ops.DGRAD_TSCALE, amax=amax)
        else:
            t_dy, u_dy = ops.wino_prep([(dy, co)], None, False, False, B, S_out, dual=True, sums=bs)
        if fused_sums:
            _grad_of(conv.bias).add_(bs.sum(0)[:co])
        ops.wgrad_wino(u_dy, t_act, B, co, ci, S_out, _grad_of(conv.weight))
        if not need_dx:
            return None
        if f6:
            dx = ops.conv3_wino(dgrad_wino_weight_f6(layer, name, conv), t_dy, B, S_out,
                                out_scale=1.0 if amax is not None else 1.0 / ops.DGRAD_TSCALE, amax=amax)
        else:
            dx = ops.conv3_wino(dgrad_wino_weight(layer, name, conv), t_dy, B, S_out)
        if ups:
            dx = resample(dx, B, ci, S_out // 2, 0)
        return dx
    # weight gradient
    S_fine = S_out * stride
    if stride == 2:
        dy_pb = to_pb16(dy, B, co_t, S_fine, 0, stuff=1, zhalo=False)
    elif shared is not None and pad == 1:
        dy_pb = shared.get("dy_pb")
        if dy_pb is None:
            dy_pb = shared["dy_pb"] = to_pb16(dy, B, co_t, S_out, 0, pad=pad, zhalo=False)
    else:
        dy_pb = to_pb16(dy, B, co_t, S_out, 0, pad=pad, zhalo=False)
    c_src = act_channels if act_channels is not None else ci      # channels of the S16B operand tensor
    act_pb = to_pb16(act_s16, B, c_src, S_fine, 1, up=ups, pad=pad)
    wgrad(dy_pb, act_pb, B, co, ci, S_fine, taps, _grad_of(conv.weight), ci * taps, taps, 1, a_ch=co_t, b_ch=c_src)
    del dy_pb, act_pb
    if not need_dx:
        return None
    # data gradient: same conv kernels, flipped/transposed weights
    if stride == 2:
        dyz = resample(dy, B, co, S_out, 1)                       # fine grid, dy at odd positions
        cfg = ops.conv_cfg_for(S_fine)
        pw = dgrad_weight(layer, name, conv, cfg)
        return layers.run_conv3(pw, split_f32b(dyz, B, co_t, S_fine ** 3), B, S_fine)
    if ksz == 3 and co == co_t and ops.wino_ok(ci, co, S_out, B):
        # Winograd path (2/3 of the MFMA work): the operand pass reads dy as it is (fp32, no affine) in place of the bf16 split
        t = ops.wino_prep([(dy, co)], None, False, False, B, S_out)
        dx = ops.conv3_wino(dgrad_wino_weight(layer, name, conv), t, B, S_out)
        del t
        if ups:
            dx = resample(dx, B, ci, S_out // 2, 0)
        return dx
    cfg = ops.conv_cfg_for(S_out)
    k16 = ops.CFG_C3_128_K16 if ksz == 3 else ops.CFG_C5_128_K16
    if ksz == 5:
        assert co <= 16, "5x5x5 data gradients are only needed for the 4-channel head"
        cfg = k16
    elif cfg == ops.CFG_C3_128_FAST and co % 32 != 0:
        cfg = k16 if co <= 16 else cfg
    pw = dgrad_weight(layer, name, conv, cfg)
    dyc = dy
    co_k = co
    if cfg == k16:                         # head: dy has 4 (padded to 8) channels -> K padded to 16
        dy16 = torch.zeros((B, 2, P, 8), dtype=torch.float32, device=dev)
        dy16[:, :dy.shape[1]] = dy
        dyc, co_k = dy16, 16
    if shared is not None and dyc is dy:
        dy16 = shared.get("dy_s16")
        if dy16 is None:
            dy16 = shared["dy_s16"] = split_f32b(dyc, B, co_t, P)
    else:
        dy16 = split_f32b(dyc, B, co_k if cfg == k16 else co_t, P)
    dx = layers.run_conv3(pw, dy16, B, S_out)
    if ups:
        dx = resample(dx, B, ci, S_out // 2, 0)
    return dx


def s16b_transpose(t, B, R, Cn):
    """S16B [B][R/8][2][Cn][8] -> [B][Cn/8][2][R][8]."""
    lib = _lib.load()
    out = torch.empty((B, Cn // 8, 2, R, 8), dtype=torch.bfloat16, device=t.device)
    check(lib.md_s16b_transpose(_ptr(t), _ptr(out), B, R, Cn, _stream()), "md_s16b_transpose")
    return out


def bgemm(a_s16, a_rows, b_s16, cols, kdim, B, out_rows_alloc=None, alpha=1.0):
    """Batched out[i][j] = alpha * sum_k A[i][k] B[j][k] with both operands S16B ([K/8][2][rows|cols][8] per batch)."""
    cfg = ops.gemm_cfg_for(cols, a_rows)
    ra = out_rows_alloc if out_rows_alloc is not None else ((a_rows + 7) // 8) * 8
    out = torch.empty((B, ra // 8, cols, 8), dtype=torch.float32, device=a_s16.device)
    ops.gemm_conv(cfg=cfg, a=a_s16, b=b_s16, out=out, batch=B, rows=a_rows, rows_alloc=ra, kdim=kdim, dims=(1, 1, cols),
                  a_src=ops.A_S16B, a_rows=a_rows, a_bstride=(kdim // 8) * 2 * a_rows * 8,
                  b_bstride=(kdim // 8) * 2 * cols * 8, alpha=alpha)
    return out


def softmax_keys_bwd(p_s16, dp, B, nk, nq, alpha):
    lib = _lib.load()
    ds = torch.empty_like(p_s16)
    check(lib.md_softmax_keys_bwd(_ptr(p_s16), _ptr(dp), _ptr(ds), B, nk, nq, float(alpha), _stream()), "md_softmax_keys_bwd")
    return ds


def wgrad_nin(dy_pb, xs_s16, B, co, ci, S, dw):
    """dw[ci][co] += sum x[ci] dy[co]."""
    x_pb = to_pb16(xs_s16, B, ci, S, 1)
    wgrad(dy_pb, x_pb, B, co, ci, S, 1, dw, 1, co, 0)


def nin_backward(nin, dy, xs_s16, B, P, S, need_dx=True, with_bias=True, bias_sums=None, shared=None, accumulate_into=None):
    """Backward of y[co] = sum_ci x[ci] W[ci][co] + b.  xs_s16: S16B of the forward input.
    bias_sums / shared: per-(sample, channel) sums and dy-derived operands already computed for another consumer of `dy`.
    accumulate_into: [(F32B tensor, channels), ...] covering the input channels in order -- the data gradient is ADDED to those
    tensors by one GEMM per part (rows of W sliced, the tensor as residual and output) instead of being returned: no
    concatenated gradient tensor and no separate add passes (the ResnetBlock shortcut: the parts already hold the GroupNorm_0
    path's gradient)."""
    from . import layers
    ci, co = nin.W.shape
    if with_bias:
        bs = bias_sums if bias_sums is not None else channel_sums(dy, B, co, P)
        _grad_of(nin.b).add_(bs.sum(0)[:co])
    dy16 = shared.get("dy_s16") if shared is not None else None
    if ops.wgrad_nin_ok(co, ci, P):
        # both operands as they are (S16B): no PB16 re-layout; the same split of dy feeds the data-gradient GEMM below
        if dy16 is None:
            dy16 = split_f32b(dy, B, co, P)
            if shared is not None:
                shared["dy_s16"] = dy16
        ops.wgrad_nin(dy16, xs_s16, B, co, ci, P, _grad_of(nin.W))
    else:
        dy_pb = shared.get("dy_pb") if shared is not None else None
        if dy_pb is None:
            dy_pb = to_pb16(dy, B, co, S, 0, zhalo=False)
            if shared is not None:
                shared["dy_pb"] = dy_pb
        wgrad_nin(dy_pb, xs_s16, B, co, ci, S, _grad_of(nin.W))
        del dy_pb
    if not need_dx:
        return None
    if dy16 is None:
        dy16 = split_f32b(dy, B, co, P)
        if shared is not None:
            shared["dy_s16"] = dy16
    if accumulate_into is not None:
        off = 0
        for t, c in accumulate_into:
            cfg = ops.gemm_cfg_for(P, c)
            pw = nin._cached(f"dgrad{cfg}/{off}+{c}", [nin.W],
                             lambda off=off, c=c, cfg=cfg: ops.PackedWeight(nin.W[off:off + c], "rows", cfg, nin.W.device))
            layers.run_gemm(pw, dy16, B, P, residual=t, out=t)
            off += c
        assert off == ci
        return None
    cfg = ops.gemm_cfg_for(P, ci)
    pw = nin._cached(f"dgrad{cfg}", [nin.W], lambda: ops.PackedWeight(nin.W, "rows", cfg, nin.W.device))
    return layers.run_gemm(pw, dy16, B, P)


def attn_backward(blk, sv, dy):
    """Backward of AttnBlock.forward_blocked (single head, keys blocked by 8; see layers.AttnBlock)."""
    from . import layers
    B, P, S, Cc = sv["B"], sv["P"], sv["S"], blk.channels
    x, prm, hN, qk, vT, pr, o = sv["x"], sv["prm"], sv["hN"], sv["qk"], sv["vT"], sv["pr"], sv["o"]
    alpha = ops.attn_scale(Cc)
    # y = NIN_3(o) + x
    d_o = nin_backward(blk.NIN_3, dy, o, B, P, S)                               # F32B [C][P]
    _grad_of(blk.NIN_2.b).add_(channel_sums(d_o, B, Cc, P).sum(0))              # o = V P + b_v
    d_o16 = split_f32b(d_o, B, Cc, P)                                           # [C/8][2][q][8c]
    v_cb = s16b_transpose(vT, B, P, Cc)                                         # [C/8][2][key][8c]
    dP = bgemm(v_cb, P, d_o16, P, Cc, B)                                        # [key/8][q][8]
    dS = softmax_keys_bwd(pr, dP, B, P, P, alpha)                               # S16B [key/8][2][q][8key]
    del dP
    d_o_q = s16b_transpose(d_o16, B, Cc, P)                                     # [q/8][2][c][8q]
    pr_q = s16b_transpose(pr, B, P, P)                                          # [q/8][2][key][8q]
    dV = bgemm(d_o_q, Cc, pr_q, P, P, B)                                        # F32B [c/8][key][8]
    del pr_q, d_o_q
    q16 = qk[:, :Cc // 8].contiguous()
    k16 = qk[:, Cc // 8:].contiguous()
    kT = s16b_transpose(k16, B, Cc, P)                                          # [key/8][2][c][8key]
    dq = bgemm(kT, Cc, dS, P, P, B)                                             # [c/8][q][8]
    qT = s16b_transpose(q16, B, Cc, P)                                          # [q/8][2][c][8q]
    dS_q = s16b_transpose(dS, B, P, P)                                          # [q/8][2][key][8q]
    dk = bgemm(qT, Cc, dS_q, P, P, B)                                           # [c/8][key][8]
    del dS, dS_q, kT, qT
    dqk = torch.cat([dq, dk], dim=1).contiguous()                               # F32B [2C][P]
    # q|k = NIN_0|NIN_1 (h)
    wqk = torch.cat([blk.NIN_0.W.detach(), blk.NIN_1.W.detach()], dim=1).contiguous()   # [C][2C]
    bsum = channel_sums(dqk, B, 2 * Cc, P).sum(0)
    _grad_of(blk.NIN_0.b).add_(bsum[:Cc]); _grad_of(blk.NIN_1.b).add_(bsum[Cc:])
    dw = torch.zeros_like(wqk)
    dqk_pb = to_pb16(dqk, B, 2 * Cc, S, 0, zhalo=False)
    wgrad_nin(dqk_pb, hN, B, 2 * Cc, Cc, S, dw)
    _grad_of(blk.NIN_0.W).add_(dw[:, :Cc]); _grad_of(blk.NIN_1.W).add_(dw[:, Cc:])
    del dqk_pb
    cfg = ops.gemm_cfg_for(P, Cc)
    pw = ops.PackedWeight(wqk, "rows", cfg, wqk.device)
    d_h = layers.run_gemm(pw, split_f32b(dqk, B, 2 * Cc, P), B, P)
    # v = NIN_2(h) (bias handled above)
    dv_pb = to_pb16(dV, B, Cc, S, 0, zhalo=False)
    wgrad_nin(dv_pb, hN, B, Cc, Cc, S, _grad_of(blk.NIN_2.W))
    del dv_pb
    pw2 = blk.NIN_2._cached(f"dgrad{cfg}", [blk.NIN_2.W], lambda: ops.PackedWeight(blk.NIN_2.W, "rows", cfg, blk.NIN_2.W.device))
    d_h.add_(layers.run_gemm(pw2, split_f32b(dV, B, Cc, P), B, P))
    dx = dy.clone()
    gn_backward([(x, Cc)], d_h, prm, blk.GroupNorm_0, B, P, silu=False, d_into=dx)
    return dx
