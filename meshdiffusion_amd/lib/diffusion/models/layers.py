"""DDPM U-Net layers for the MI355X path.

Host-side mirror of the live part of the reference's lib/diffusion/models/layers.py:
`default_init` :88-91, `ddpm_conv3x3` :118-124, `get_timestep_embedding` :542-556, `NIN` :573-582,
`AttnBlock` :585-608, `Upsample` :611-623, `Downsample` :626-643, `ResnetBlockDDPM` :646-689.
Class, attribute and parameter names/shapes are identical so reference checkpoints load
unchanged; everything numerical runs in libmeshdiffusion_hip.so through `hip_ops`.

Each layer has two entry points:
  forward(x_ncdhw, ...)        -- reference-compatible signature (converts layouts at the edge)
  forward_blocked(parts, ...)  -- used by the fused U-Net: F32B tensors in, F32B tensor out;
                                  `parts` is a list of (F32B tensor, channels) standing for
                                  torch.cat(parts, dim=1) without materialising the concat.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .... import hip_ops as ops


# --------------------------------------------------------------------------------------------
# initialisers (same distributions as the reference; fresh implementation)
# --------------------------------------------------------------------------------------------
def variance_scaling(scale, mode, distribution, in_axis=1, out_axis=0, dtype=torch.float32, device="cpu"):
    def init(shape, dtype=dtype, device=device):
        receptive = np.prod(shape) / shape[in_axis] / shape[out_axis]
        fan_in, fan_out = shape[in_axis] * receptive, shape[out_axis] * receptive
        denom = {"fan_in": fan_in, "fan_out": fan_out, "fan_avg": (fan_in + fan_out) / 2}[mode]
        var = scale / denom
        if distribution == "normal":
            return torch.randn(*shape, dtype=dtype, device=device) * np.sqrt(var)
        if distribution == "uniform":
            return (torch.rand(*shape, dtype=dtype, device=device) * 2.0 - 1.0) * np.sqrt(3 * var)
        raise ValueError("invalid distribution for variance scaling initializer")

    return init


def default_init(scale=1.0):
    return variance_scaling(1e-10 if scale == 0 else scale, "fan_avg", "uniform")


def get_act(config):
    name = config.model.nonlinearity.lower()
    if name == "swish":
        return nn.SiLU()
    raise NotImplementedError("only the 'swish' nonlinearity is implemented on the HIP path "
                              "(both registered reference models use it)")


def _conv(in_planes, out_planes, k, stride=1, padding=1, init_scale=1.0):
    conv = nn.Conv3d(in_planes, out_planes, kernel_size=k, stride=stride, padding=padding, bias=True)
    conv.weight.data = default_init(init_scale)(conv.weight.data.shape)
    nn.init.zeros_(conv.bias)
    return conv


def ddpm_conv3x3(in_planes, out_planes, stride=1, bias=True, dilation=1, init_scale=1.0, padding=1):
    assert bias and dilation == 1
    return _conv(in_planes, out_planes, 3, stride, padding, init_scale)


def ddpm_conv5x5(in_planes, out_planes, stride=2, bias=True, dilation=1, init_scale=1.0, padding=2):
    assert bias and dilation == 1
    return _conv(in_planes, out_planes, 5, stride, padding, init_scale)


def get_timestep_embedding(timesteps, embedding_dim, max_positions=10000):
    """Sinusoidal embedding [B] -> [B, dim] computed by md_timestep_embedding."""
    assert len(timesteps.shape) == 1 and max_positions == 10000
    return ops.timestep_embedding(timesteps, embedding_dim)


# --------------------------------------------------------------------------------------------
# packed-weight cache
# --------------------------------------------------------------------------------------------
class HipLayer(nn.Module):
    """Caches device-side packed weights; rebuilt when a parameter is modified in place,
    re-assigned or moved (key = data_ptr + version counter)."""

    def _cached(self, name, params, builder, mark=True):
        cache = self.__dict__.setdefault("_md_cache", {})
        key = (ops.PARAM_EPOCH,) + tuple((p.data_ptr(), p._version, str(p.device)) for p in params)
        hit = cache.get(name)
        if hit is None or hit[0] != key:
            hit = (key, builder())
            cache[name] = hit
        if mark:
            ops.note_pack_use(self, name, params, builder)     # hip_ops.prewarm_packs: rebuilt in one batch before the next training forward
        return hit[1]


def _spatial_edge(P):
    s = round(P ** (1.0 / 3.0))
    assert s * s * s == P, "cubic grids only"
    return s


def _parts_of(x):
    """NCDHW tensor -> ([(F32B, C)], B, P, spatial)."""
    B, Cc = x.shape[0], x.shape[1]
    spatial = tuple(x.shape[2:])
    return [(ops.ncdhw_to_f32b(x), Cc)], B, int(np.prod(spatial)), spatial


def conv3_packed(layer, name, conv, cfg):
    prec = ops.fast_prec(cfg)
    return layer._cached(f"{name}/p{prec}", [conv.weight],
                         lambda: ops.PackedWeight(conv.weight, "conv", cfg, conv.weight.device, prec))


def conv3_wino_packed(layer, name, conv, gn=None):
    """Lazy builder of the Winograd-transformed weight tiles of `conv` (cached like conv3_packed); f8: the fragments of the
    f16f8 / f16f6 arithmetic (hip_ops.WinoWeightF8, inference).  gn: the nn.GroupNorm whose output (through SiLU) the conv reads --
    the f16f8 / f16f6 fragments are then packed with the pair's equaliser (`build.eq()`: the vector the operand pass of the
    same launch needs; hip_ops.wino_equaliser), rebuilt whenever the weight or the GroupNorm affine changes.  After a calibration
    (DDPMUNet3D.calibrate) the layer holds the MEASURED per-channel mean squares of this conv's operand (`layer._md_act_ms[name]`):
    the equaliser is then built from them (also for a conv no GroupNorm precedes: `build.measured()` is what lets an Upsample
    conv onto the reduced-precision path)."""
    def measured():
        return layer.__dict__.get("_md_act_ms", {}).get(name)

    def eq():
        ms = measured() if ops.WINO_EQ else None
        if ms is None and not (gn is not None and ops.WINO_EQ):
            return None
        gp = [gn.weight, gn.bias] if gn is not None else []
        if ms is None:
            return layer._cached(f"{name}/wino_eq", [conv.weight] + gp, lambda: ops.wino_equaliser(gn.weight, gn.bias, conv.weight), mark=False)
        return layer._cached(f"{name}/wino_eqm", [conv.weight] + gp + [ms],
                             lambda: ops.wino_equaliser(gn.weight if gn is not None else None, gn.bias if gn is not None else None,
                                                        conv.weight, a2m=ms), mark=False)

    def build(f8=False):
        if f8:
            fmt = "f6" if f8 == "f6" else "f8"
            ms = measured() if ops.WINO_EQ else None
            use_eq = ms is not None or (gn is not None and ops.WINO_EQ)
            deps = [conv.weight] + ([gn.weight, gn.bias] if (use_eq and gn is not None) else []) + ([ms] if ms is not None else [])
            # mark=False (here and for the equalisers): inference-only entries -- a training step's prewarm_packs must not rebuild them
            # after every evaluation in between
            return layer._cached(f"{name}/wino_{fmt}{('m' if ms is not None else 'e') if use_eq else ''}", deps,
                                 lambda: ops.WinoWeightF8(conv.weight, conv.weight.device, fmt, eq=eq()), mark=False)
        return layer._cached(f"{name}/wino", [conv.weight], lambda: ops.WinoWeight(conv.weight, conv.weight.device))
    build.eq, build.measured = eq, measured
    build.owner, build.site = layer, name      # identifies the conv for per-layer overrides (layer.md_bf16x3_sites), the audit and the calibration
    return build


def fused_operand_ok(pw, ups=0):
    """True when a conv on packed weights `pw` can take its input as fp32 F32B parts and apply GroupNorm + SiLU + the
    bf16 split itself (MD_B_F32B_GN: the dedicated kernel, or the 4^3-level tile of the generic kernel; bf16x3 arithmetic).
    The generic tile's fused loader has no nearest-x2 fold (launch_cfg: MD_ERR_UNSUPPORTED for BF && ups), so an Upsample
    onto a 4^3 grid stays on the split path."""
    cfgs = (ops.CFG_C3_128_FAST,) if ups else (ops.CFG_C3_128_FAST, ops.CFG_C3_LOW)
    return ops.FUSE_GN_APPLY and pw.cfg in cfgs and pw.prec == ops.PREC_BF16X3


def run_conv3(pw, act_s16, B, S_out, *, bias=None, bias_bstride=0, residual=None, res_bstride=None, ups=0,
              out=None, out_mode=ops.OUT_F32B, rows_alloc=None, want_stats=False, b_f32=None, wino=None):
    """3x3x3 conv of an S16B activation tensor with packed weights `pw` on an S_out^3 output grid.
    want_stats: the output feeds a GroupNorm -- when the launch allows it (dedicated kernel, no split-K) its
    epilogue also accumulates the per-(sample, channel) sums, attached to the result as `_md_sums`.
    b_f32 (see hip_ops.gemm_conv): fp32 parts + folded GroupNorm affine instead of `act_s16` (fused_operand_ok); with
    `wino_only` (training: `drop` = (p, seed) allowed) it only describes the operand of the Winograd path and the direct
    kernel keeps reading `act_s16`.
    wino (with b_f32): builder of the layer's WinoWeight (conv3_wino_packed); where hip_ops.wino_ok says so the conv runs as
    md_wino_prep + md_conv3_wino (Winograd F(2,3) along w: 2/3 of the matrix-core work) instead of the direct kernel.
    b_f32["keep"] (training, layers with hip_ops.wgrad_wino_ok): the operand T is allocated on its own and handed back as
    b_f32["t_out"]; `act_s16` may then be None (nothing else reads the S16B activation of such a layer)."""
    P = S_out ** 3
    dev = act_s16.device if act_s16 is not None else b_f32["parts"][0][0].device
    rows_alloc = rows_alloc if rows_alloc is not None else ((pw.rows + 7) // 8) * 8
    if out is None:
        out = ops.f32b_empty(B, rows_alloc, P, dev)
    if residual is not None and res_bstride is None:
        res_bstride = rows_alloc * P
    if (b_f32 is not None and wino is not None and out_mode == ops.OUT_F32B and rows_alloc == pw.rows
            and pw.prec == ops.PREC_BF16X3 and ops.wino_ok(pw.rows, pw.kdim, S_out, B)):
        stats = ops.stats_zeros(B, rows_alloc, dev) if want_stats and ops.FUSE_GN_STATS else None
        # f16f8 / f16f6: inference operands that come out of a GroupNorm (the layer's static equaliser flattens their channels), or
        # any operand whose layer holds a measured equaliser (DDPMUNet3D.calibrate); an uncalibrated raw residual stream (Upsample:
        # ac None) stays in bf16x3
        owner = getattr(wino, "owner", None)
        if ops.CALIBRATE is not None and owner is not None and not b_f32.get("wino_only"):
            # calibration evaluation: what this conv's operand really looks like, per input channel (nearest-x2 upsampling does not
            # change a channel's mean square: measured on the coarse tensor)
            ms = ops.wino_operand_ms(b_f32["parts"], b_f32.get("ac"), b_f32.get("silu"), B, b_f32["parts"][0][0].shape[2])
            rec = ops.CALIBRATE.setdefault((id(owner), wino.site), [owner, wino.site, torch.zeros_like(ms), 0])
            rec[2] += ms
            rec[3] += 1
        f8 = (not b_f32.get("wino_only")) and ops.wino_f8_ok(
            S_out, drop=b_f32.get("drop"), keep=bool(b_f32.get("keep")), parts=b_f32["parts"],
            normalised=b_f32.get("ac") is not None or (hasattr(wino, "measured") and wino.measured() is not None))
        if f8 and owner is not None and wino.site in getattr(owner, "md_bf16x3_sites", ()):
            f8 = False                   # this conv was taken off the reduced-precision path (layer.md_bf16x3_sites: tools/audit_precision.py)
        t = ops.wino_prep(b_f32["parts"], b_f32.get("ac"), b_f32.get("silu"), ups, B, S_out, drop=b_f32.get("drop"),
                          keep=bool(b_f32.get("keep")), f8=f8, eq=wino.eq() if f8 and hasattr(wino, "eq") else None)
        if b_f32.get("keep"):
            b_f32["t_out"] = t           # training: the Winograd weight gradient reads the operand again (tape)
        ops.conv3_wino(wino(f8) if f8 else wino(), t, B, S_out, bias=bias, bias_bstride=bias_bstride, residual=residual,
                       res_bstride=res_bstride or 0, stats=stats, out=out)
        if f8 and ops.AUDIT is not None:
            # diagnostic mode (tools/audit_precision.py): the same launch once more in bf16x3; the pair's relative difference is the
            # error the reduced-precision format adds on THIS layer with THESE weights and activations
            t3 = ops.wino_prep(b_f32["parts"], b_f32.get("ac"), b_f32.get("silu"), ups, B, S_out)
            ref = ops.conv3_wino(wino(), t3, B, S_out, bias=bias, bias_bstride=bias_bstride, residual=residual, res_bstride=res_bstride or 0)
            ops.AUDIT.append(dict(owner=getattr(wino, "owner", None), site=getattr(wino, "site", None), fmt=f8, cin=pw.kdim, cout=pw.rows,
                                  S=S_out, rel_l2=float((torch.linalg.vector_norm(out - ref, dtype=torch.float64)
                                                         / torch.linalg.vector_norm(ref, dtype=torch.float64).clamp_min(1e-300)).item())))
            del t3, ref
        if stats is not None:
            out._md_sums = stats
        elif hasattr(out, "_md_sums"):
            del out._md_sums
        return out
    if b_f32 is not None and b_f32.get("wino_only"):
        b_f32 = None                     # training: the direct kernel takes the taped S16B activation (`act_s16`)
    ksplit = ops.ksplit_for(pw.cfg, B, pw.rows, pw.kdim, S_out) if out_mode == ops.OUT_F32B else 1
    stats = None
    # statistics of the output: the dedicated kernel's epilogue, or (8^3 / 4^3 levels) the split-K finish kernel
    # (one block per (sample, 8-channel group) there: only where that fills the chip -- at B = 1 the 32^3 level would run on 16 blocks)
    if (want_stats and ops.FUSE_GN_STATS and ((pw.cfg == ops.CFG_C3_128_FAST and ksplit == 1)
                                              or (ksplit > 1 and ops.SPLITK_STATS and P <= 512 and B * rows_alloc >= 2048))
            and out_mode == ops.OUT_F32B and rows_alloc == pw.rows):
        stats = ops.stats_zeros(B, rows_alloc, dev)
    ops.gemm_conv(cfg=pw.cfg, a=pw.data, b=act_s16, out=out, batch=B, rows=pw.rows,
                  rows_alloc=rows_alloc, kdim=pw.kdim, dims=(S_out, S_out, S_out), bias=bias,
                  bias_bstride=bias_bstride, residual=residual, res_bstride=res_bstride or 0, ups=ups,
                  out_mode=out_mode, ksplit=ksplit, prec=pw.prec, stats=stats, b_f32=b_f32)
    if stats is not None:
        out._md_sums = stats
    elif hasattr(out, "_md_sums"):
        del out._md_sums
    return out


def run_gemm(pw, act_s16, B, P, *, bias=None, bias_bstride=0, residual=None, out=None, out_mode=ops.OUT_F32B,
             alpha=1.0, b_bstride=None, b_f32=None):
    """1x1x1 conv / GEMM with packed weights: out[rows][P].  b_f32: fp32 F32B parts split by the kernel's loader
    (CFG_G1_128 only; hip_ops.gemm_conv) instead of the S16B operand `act_s16`."""
    rows_alloc = ((pw.rows + 7) // 8) * 8
    dev = act_s16.device if act_s16 is not None else b_f32["parts"][0][0].device
    if out is None:
        out = (ops.f32b_empty if out_mode == ops.OUT_F32B else ops.s16b_empty)(B, rows_alloc, P, dev)
    return ops.gemm_conv(cfg=pw.cfg, a=pw.data, b=act_s16, out=out, batch=B, rows=pw.rows,
                         rows_alloc=rows_alloc, kdim=pw.kdim, dims=(1, 1, P), bias=bias,
                         bias_bstride=bias_bstride, residual=residual,
                         res_bstride=rows_alloc * P if residual is not None else 0, alpha=alpha,
                         out_mode=out_mode, b_bstride=b_bstride, b_f32=b_f32)


# --------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------
class NIN(HipLayer):
    """1x1x1 channel mixing: y[..., o] = sum_i x[..., i] W[i, o] + b[o]  (W is [in, out])."""

    def __init__(self, in_dim, num_units, init_scale=0.1):
        super().__init__()
        self.W = nn.Parameter(default_init(scale=init_scale)((in_dim, num_units)), requires_grad=True)
        self.b = nn.Parameter(torch.zeros(num_units), requires_grad=True)

    def packed(self, P, hbm_bound=False):
        rows = self.W.shape[1]
        cfg = ops.gemm_cfg_for(P, rows, hbm_bound)
        return self._cached(f"w{cfg}", [self.W], lambda: ops.PackedWeight(self.W, "nin", cfg, self.W.device))

    def forward_s16(self, act_s16, B, P, residual=None, out_mode=ops.OUT_F32B):
        return run_gemm(self.packed(P), act_s16, B, P, bias=self.b, residual=residual, out_mode=out_mode)

    def forward(self, x):
        parts, B, P, spatial = _parts_of(x)
        act = ops.gn_apply(parts, None, B, P, norm=False, silu=False)
        return ops.f32b_to_ncdhw(self.forward_s16(act, B, P), spatial)


class AttnBlock(HipLayer):
    """Single-head self-attention over the D*H*W tokens (head dim = C)."""

    def __init__(self, channels):
        super().__init__()
        self.GroupNorm_0 = nn.GroupNorm(num_groups=32, num_channels=channels, eps=1e-6)
        self.NIN_0 = NIN(channels, channels)
        self.NIN_1 = NIN(channels, channels)
        self.NIN_2 = NIN(channels, channels)
        self.NIN_3 = NIN(channels, channels, init_scale=0.0)
        self.channels = channels

    def _qk_packed(self, P):
        cfg = ops.gemm_cfg_for(P, 2 * self.channels)

        def build():
            w = torch.cat([self.NIN_0.W.detach(), self.NIN_1.W.detach()], dim=1)  # [C][2C]
            return ops.PackedWeight(w, "nin", cfg, w.device)

        return self._cached(f"qk{cfg}", [self.NIN_0.W, self.NIN_1.W], build)

    def _qk_bias(self):
        return self._cached("qkb", [self.NIN_0.b, self.NIN_1.b],
                            lambda: torch.cat([self.NIN_0.b.detach(), self.NIN_1.b.detach()]).contiguous())

    def _wv_s16(self):
        return self._cached("wv", [self.NIN_2.W], lambda: ops.pack_s16b_from_matrix(self.NIN_2.W, self.NIN_2.W.device))

    def forward_blocked(self, x, B, P, tape=None):
        Cc = self.channels
        dev = x.device
        gn = self.GroupNorm_0
        params = ops.gn_params([(x, Cc)], gn.weight, gn.bias, B, P, eps=gn.eps, groups=gn.num_groups)
        h = ops.gn_apply([(x, Cc)], params, B, P, norm=True, silu=False)          # S16B [B][C][P]
        # q | k  = NIN_0 | NIN_1 in one GEMM: S16B [B][2C][P]
        qk = run_gemm(self._qk_packed(P), h, B, P, bias=self._qk_bias(), out_mode=ops.OUT_S16B)
        qk_bstride = (2 * Cc // 8) * 2 * P * 8
        k_view = qk.view(-1)[(Cc // 8) * 2 * P * 8:]
        # vT[token][c] = sum_i h[i][token] Wv[i][c]  -> S16B over tokens: [B][P/8][2][C][8]
        cfg_v = ops.gemm_cfg_for(Cc, P)
        vT = ops.s16b_empty(B, P, Cc, dev)
        ops.gemm_conv(cfg=cfg_v, a=h, b=self._wv_s16(), out=vT, batch=B, rows=P, rows_alloc=P, kdim=Cc,
                      dims=(1, 1, Cc), a_src=ops.A_S16B, a_rows=P, a_bstride=(Cc // 8) * 2 * P * 8,
                      b_bstride=0, out_mode=ops.OUT_S16B)
        if tape is None and ops.attn_fused_ok(Cc, P):
            # inference at the 16^3 levels: QK^T, online softmax and PV in one kernel (no [B][P][P] score matrix)
            o = ops.attn_fwd(qk, vT, self.NIN_2.b, B, Cc, P, ops.attn_scale(Cc))
            return self.NIN_3.forward_s16(o, B, P, residual=x)
        # S^T[key][query] = C^-1/2 * sum_c k[c][key] q[c][query]   (fp32, keys blocked by 8)
        cfg_s = ops.gemm_cfg_for(P, P)
        sT = ops.f32b_empty(B, P, P, dev)
        ops.gemm_conv(cfg=cfg_s, a=k_view, b=qk, out=sT, batch=B, rows=P, rows_alloc=P, kdim=Cc,
                      dims=(1, 1, P), a_src=ops.A_S16B, a_rows=P, a_bstride=qk_bstride, b_bstride=qk_bstride,
                      alpha=ops.attn_scale(Cc))
        pr = ops.softmax_keys(sT, B, P, P)                                          # S16B [B][P keys][P q]
        # o[c][q] = sum_key vT[key][c] p[key][q] + b_v[c]   (rows of softmax sum to 1)
        cfg_o = ops.gemm_cfg_for(P, Cc)
        o = ops.s16b_empty(B, Cc, P, dev)
        ops.gemm_conv(cfg=cfg_o, a=vT, b=pr, out=o, batch=B, rows=Cc, rows_alloc=Cc, kdim=P, dims=(1, 1, P),
                      a_src=ops.A_S16B, a_rows=Cc, a_bstride=(P // 8) * 2 * Cc * 8, bias=self.NIN_2.b,
                      out_mode=ops.OUT_S16B)
        if tape is not None:
            tape.append(dict(layer=self, x=x, prm=params, hN=h, qk=qk, vT=vT, pr=pr, o=o, B=B, P=P, S=_spatial_edge(P)))
        return self.NIN_3.forward_s16(o, B, P, residual=x)

    def backward_blocked(self, sv, dy):
        from . import backward as bw
        return bw.attn_backward(self, sv, dy)

    def forward(self, x):
        parts, B, P, spatial = _parts_of(x)
        return ops.f32b_to_ncdhw(self.forward_blocked(parts[0][0], B, P), spatial)


class Upsample(HipLayer):
    def __init__(self, channels, with_conv=False):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Upsample without conv is not used by the registered models")
        self.Conv_0 = ddpm_conv3x3(channels, channels)
        self.with_conv = with_conv

    def forward_blocked(self, x, Cc, B, P, tape=None):
        s_out = 2 * _spatial_edge(P)
        pw = conv3_packed(self, "w", self.Conv_0, ops.conv_cfg_for(s_out))
        if tape is None and fused_operand_ok(pw, ups=1) and pw.kdim == Cc:   # the conv splits the raw fp32 input while loading it
            return run_conv3(pw, None, B, s_out, bias=self.Conv_0.bias, ups=1, want_stats=True,
                             b_f32=dict(parts=[(x, Cc)], ac=None, silu=False), wino=conv3_wino_packed(self, "w", self.Conv_0))
        fw = None
        ww = (tape is not None and pw.prec == ops.PREC_BF16X3 and pw.kdim == Cc
              and ops.wgrad_wino_ok(pw.rows, pw.kdim, s_out, B))      # Winograd weight gradient: T instead of the S16B operand
        act = None if ww else ops.gn_apply([(x, Cc)], None, B, P, norm=False, silu=False, fp16=pw.prec == ops.PREC_FP16X2)
        if tape is not None:
            assert pw.prec == ops.PREC_BF16X3
            if ops.WINO_TRAIN_FWD:
                fw = dict(parts=[(x, Cc)], ac=None, silu=False, wino_only=True, keep=ww)
        out = run_conv3(pw, act, B, s_out, bias=self.Conv_0.bias, ups=1, want_stats=True, b_f32=fw,
                        wino=conv3_wino_packed(self, "w", self.Conv_0) if fw is not None else None)
        if tape is not None:
            tape.append(dict(layer=self, act=act, t=fw.get("t_out") if fw else None, B=B, S_out=s_out))
        return out

    def backward_blocked(self, sv, dy):
        from . import backward as bw
        return bw.conv3_backward(self, "w", self.Conv_0, dy, sv["act"], sv["B"], sv["S_out"], ups=1, t_act=sv.get("t"))

    def forward(self, x):
        parts, B, P, spatial = _parts_of(x)
        return ops.f32b_to_ncdhw(self.forward_blocked(parts[0][0], parts[0][1], B, P), tuple(2 * s for s in spatial))


class Downsample(HipLayer):
    def __init__(self, channels, with_conv=False):
        super().__init__()
        if not with_conv:
            raise NotImplementedError("Downsample without conv is not used by the registered models")
        self.Conv_0 = ddpm_conv3x3(channels, channels, stride=2, padding=0)
        self.with_conv = with_conv

    def forward_blocked(self, x, Cc, B, P, tape=None):
        s_out = _spatial_edge(P) // 2
        if ops.conv3_s2_ok(self.Conv_0.weight.shape[0], Cc, s_out, B):
            # one kernel on the raw fp32 tensor (inference: no split pass at all), GroupNorm sums of the output from its epilogue
            pw = self._cached("w/s2", [self.Conv_0.weight],
                              lambda: ops.PackedWeight(self.Conv_0.weight, "conv", ops.CFG_S2_PACK, self.Conv_0.weight.device))
            stats = ops.stats_zeros(B, pw.rows, x.device) if ops.FUSE_GN_STATS and pw.rows % 8 == 0 else None
            out = ops.conv3_s2(pw, x, B, s_out, bias=self.Conv_0.bias, stats=stats)
            if stats is not None:
                out._md_sums = stats
            if tape is not None:     # training: the weight gradient reads the split operand (md_to_pb16 <- S16B)
                tape.append(dict(layer=self, act=ops.gn_apply([(x, Cc)], None, B, P, norm=False, silu=False), B=B, S_out=s_out))
            return out
        act = ops.gn_apply([(x, Cc)], None, B, P, norm=False, silu=False)
        pw = conv3_packed(self, "w", self.Conv_0, ops.conv_cfg_for(s_out, stride=2))
        if tape is not None:
            tape.append(dict(layer=self, act=act, B=B, S_out=s_out))
        return run_conv3(pw, act, B, s_out, bias=self.Conv_0.bias)

    def backward_blocked(self, sv, dy):
        from . import backward as bw
        return bw.conv3_backward(self, "w", self.Conv_0, dy, sv["act"], sv["B"], sv["S_out"], stride=2)

    def forward(self, x):
        parts, B, P, spatial = _parts_of(x)
        return ops.f32b_to_ncdhw(self.forward_blocked(parts[0][0], parts[0][1], B, P), tuple(s // 2 for s in spatial))


class ResnetBlockDDPM(HipLayer):
    """GN-SiLU-conv3 (+temb bias) - GN-SiLU-dropout-conv3 + shortcut (identity or NIN)."""

    def __init__(self, act, in_ch, out_ch=None, temb_dim=None, conv_shortcut=False, dropout=0.1):
        super().__init__()
        out_ch = in_ch if out_ch is None else out_ch
        if conv_shortcut:
            raise NotImplementedError("conv_shortcut is never enabled by the registered models")
        self.GroupNorm_0 = nn.GroupNorm(num_groups=32, num_channels=in_ch, eps=1e-6)
        self.act = act
        self.Conv_0 = ddpm_conv3x3(in_ch, out_ch)
        if temb_dim is not None:
            self.Dense_0 = nn.Linear(temb_dim, out_ch)
            self.Dense_0.weight.data = default_init()(self.Dense_0.weight.data.shape)
            nn.init.zeros_(self.Dense_0.bias)
        self.GroupNorm_1 = nn.GroupNorm(num_groups=32, num_channels=out_ch, eps=1e-6)
        self.Dropout_0 = nn.Dropout(dropout)
        self.Conv_1 = ddpm_conv3x3(out_ch, out_ch, init_scale=0.0)
        if in_ch != out_ch:
            self.NIN_0 = NIN(in_ch, out_ch)
        self.out_ch, self.in_ch, self.conv_shortcut = out_ch, in_ch, conv_shortcut

    def _bias0(self):
        """Conv_0.bias + Dense_0.bias, folded into one per-channel constant."""
        if hasattr(self, "Dense_0"):
            return self._cached("b0", [self.Conv_0.bias, self.Dense_0.bias],
                                lambda: (self.Conv_0.bias.detach() + self.Dense_0.bias.detach()).contiguous())
        return self.Conv_0.bias

    def forward_blocked(self, parts, B, P, temb=None, bias0=None, bias0_stride=None, tape=None):
        """bias0 (optional): precomputed Conv_0.bias + Dense_0(SiLU(temb)) rows [B, out_ch] with row stride
        `bias0_stride` floats (the U-Net computes all blocks' FiLM biases in one launch).
        tape (optional list): receives what `backward_blocked` needs (training)."""
        drop = (self.Dropout_0.p, ops.next_dropout_seed()) if self.training and self.Dropout_0.p > 0 else None
        S = _spatial_edge(P)
        cin = sum(c for _, c in parts)
        assert cin == self.in_ch
        cfg = ops.conv_cfg_for(S)
        g0, g1 = self.GroupNorm_0, self.GroupNorm_1
        pw0, pw1 = conv3_packed(self, "w0", self.Conv_0, cfg), conv3_packed(self, "w1", self.Conv_1, cfg)
        f16 = pw0.prec == ops.PREC_FP16X2   # operand format follows the kernel that consumes the tensor
        need_nin = self.in_ch != self.out_ch
        if (tape is None and drop is None and fused_operand_ok(pw0) and fused_operand_ok(pw1) and pw0.kdim == cin
                and pw1.kdim == self.out_ch and not ops.NIN_SIDE_STREAM):
            # inference: no GroupNorm-apply pass at all -- both convs read the fp32 tensors and normalise / activate /
            # split them in their halo loaders; the statistics come from the producing convs' epilogues
            _, ac0 = ops.gn_params(parts, g0.weight, g0.bias, B, P, eps=g0.eps, groups=g0.num_groups, want_ac=True)
            if bias0 is None:
                if temb is not None:
                    bias0, bias0_stride = ops.linear(temb, self.Dense_0.weight, self._bias0(), silu_in=True), self.out_ch
                else:
                    bias0, bias0_stride = self.Conv_0.bias, 0
            h = run_conv3(pw0, None, B, S, bias=bias0, bias_bstride=bias0_stride, want_stats=True,
                          b_f32=dict(parts=parts, ac=ac0, silu=True), wino=conv3_wino_packed(self, "w0", self.Conv_0, gn=g0))
            if need_nin:
                pwn = self.NIN_0.packed(P, hbm_bound=True)
                if pwn.kdim == cin and ops.nin_stream_ok(parts, self.out_ch, P):   # weights resident in LDS, input streamed once
                    res = ops.nin_f32(parts, pwn, self.NIN_0.b, B, P)
                elif pwn.cfg in (ops.CFG_G1_128, ops.CFG_G1_128_N128) and pwn.kdim == cin:   # the shortcut GEMM splits the raw fp32 parts itself
                    res = run_gemm(pwn, None, B, P, bias=self.NIN_0.b, b_f32=dict(parts=parts, ac=None, silu=False))
                else:                                               # small grids: one raw split pass of the block input
                    xs = ops.gn_apply(parts, None, B, P, norm=False, silu=False)
                    res = self.NIN_0.forward_s16(xs, B, P)
            else:
                res = parts[0][0]
            _, ac1 = ops.gn_params([(h, self.out_ch)], g1.weight, g1.bias, B, P, eps=g1.eps, groups=g1.num_groups, want_ac=True)
            return run_conv3(pw1, None, B, S, bias=self.Conv_1.bias, residual=res, want_stats=True,
                             b_f32=dict(parts=[(h, self.out_ch)], ac=ac1, silu=True), wino=conv3_wino_packed(self, "w1", self.Conv_1, gn=g1))
        # training (tape): the convs go through the Winograd path where it applies (its operand pass repeats GroupNorm + SiLU
        # + dropout from the fp32 tensors); the S16B activations are still written: the weight gradients read them
        wino_fwd = tape is not None and not f16 and ops.WINO_TRAIN_FWD
        # layers whose weight gradient runs in the Winograd domain (hip_ops.wgrad_wino_ok) keep the operand T of the forward
        # conv on the tape and write NO S16B activation: nothing else would read it
        ww0 = wino_fwd and pw0.kdim == cin and ops.wgrad_wino_ok(self.out_ch, cin, S, B)
        ww1 = wino_fwd and pw1.kdim == self.out_ch and ops.wgrad_wino_ok(self.out_ch, self.out_ch, S, B)
        prm, ac0 = ops.gn_params(parts, g0.weight, g0.bias, B, P, eps=g0.eps, groups=g0.num_groups, want_ac=True)
        f0 = dict(parts=parts, ac=ac0, silu=True, wino_only=True, keep=ww0) if wino_fwd else None
        w0 = conv3_wino_packed(self, "w0", self.Conv_0) if f0 is not None else None
        xs = None
        res = None
        if ww0:
            a0 = None
            if need_nin:                 # the shortcut NIN (forward GEMM and its weight gradient) still takes the raw bf16 split
                xs = ops.gn_apply(parts, None, B, P, norm=False, silu=False)
        else:
            a0 = ops.gn_apply(parts, prm, B, P, norm=True, silu=True, fp16=f16, want_raw=need_nin)
            if need_nin:
                a0, xs = a0
        if need_nin and ops.NIN_SIDE_STREAM:
            # The shortcut GEMM is HBM-bound and independent of Conv_0 (MFMA-bound): launch it on a second HIP
            # stream so that it shares the GPU with the convolution instead of preceding Conv_1 serially.
            main, side = torch.cuda.current_stream(), ops.side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                res = self.NIN_0.forward_s16(xs, B, P)
            res.record_stream(main)
            xs.record_stream(side)
        if bias0 is not None:
            h = run_conv3(pw0, a0, B, S, bias=bias0, bias_bstride=bias0_stride, want_stats=True, b_f32=f0, wino=w0)
        elif temb is not None:   # per-(sample, channel) additive bias = Conv_0.b + Dense_0(SiLU(temb))
            bias0 = ops.linear(temb, self.Dense_0.weight, self._bias0(), silu_in=True)
            h = run_conv3(pw0, a0, B, S, bias=bias0, bias_bstride=self.out_ch, want_stats=True, b_f32=f0, wino=w0)
        else:
            h = run_conv3(pw0, a0, B, S, bias=self.Conv_0.bias, want_stats=True, b_f32=f0, wino=w0)
        prm1, ac1 = ops.gn_params([(h, self.out_ch)], g1.weight, g1.bias, B, P, eps=g1.eps, groups=g1.num_groups, want_ac=True)
        a1 = None if ww1 else ops.gn_apply([(h, self.out_ch)], prm1, B, P, norm=True, silu=True, fp16=f16, drop=drop)
        f1 = dict(parts=[(h, self.out_ch)], ac=ac1, silu=True, drop=drop, wino_only=True, keep=ww1) if wino_fwd else None
        w1 = conv3_wino_packed(self, "w1", self.Conv_1) if f1 is not None else None
        if need_nin:
            if res is None:
                res = self.NIN_0.forward_s16(xs, B, P)
            else:
                torch.cuda.current_stream().wait_stream(ops.side_stream())
        else:
            assert len(parts) == 1
            res = parts[0][0]
        out = run_conv3(pw1, a1, B, S, bias=self.Conv_1.bias, residual=res, want_stats=True, b_f32=f1, wino=w1)
        if tape is not None:
            assert not f16, "the backward pass uses the bf16x3 operand format"
            tape.append(dict(layer=self, parts=parts, prm0=prm, a0=a0, h=h, prm1=prm1, a1=a1, xs=xs, B=B, P=P, S=S,
                             temb=temb, drop=drop, t0=f0.get("t_out") if f0 else None, t1=f1.get("t_out") if f1 else None))
        return out

    def backward_blocked(self, sv, dy):
        """dy: F32B [B][out_ch][P].  Returns ([grad per input part], dbias0 [B, out_ch]); accumulates .grad of
        Conv_0/Conv_1/GroupNorm_0/GroupNorm_1/NIN_0 (Dense_0 is handled by the caller from dbias0)."""
        from . import backward as bw
        B, P, S, parts = sv["B"], sv["P"], sv["S"], sv["parts"]
        # Conv_1 and the shortcut NIN_0 both start from the block's output gradient: its channel sums (both biases), its PB16
        # operand (both weight gradients) and its bf16 split (both data gradients) are computed once
        shared = {} if self.in_ch != self.out_ch else None
        if sv.get("t1") is not None:     # Winograd backward: Conv_1's operand pass over dy fills the channel sums on its way
            bsum = torch.zeros((B, self.out_ch), dtype=torch.float32, device=dy.device)
        else:
            bsum = bw.channel_sums(dy, B, self.out_ch, P)
        d_a1 = bw.conv3_backward(self, "w1", self.Conv_1, dy, sv["a1"], B, S, bias_sums=bsum, shared=shared, t_act=sv.get("t1"),
                                 bias_sums_out=sv.get("t1") is not None)
        dbias0 = torch.zeros((B, self.out_ch), dtype=torch.float32, device=dy.device)
        d_h = bw.gn_backward([(sv["h"], self.out_ch)], d_a1, sv["prm1"], self.GroupNorm_1, B, P, silu=True,
                             drop=sv.get("drop"), sums_out=dbias0,      # d(bias0) = channel sums of d_h, same pass
                             want_amax=ops.DGRAD_F6 and ops.DGRAD_LIFT == "dyn" and sv.get("t0") is not None)[0]
        del d_a1
        # Conv_0.bias gradient = batch sum of dbias0: the FiLM caller adds it once together with Dense_0's
        d_a0 = bw.conv3_backward(self, "w0", self.Conv_0, d_h, sv["a0"], B, S, bias_sums=False, t_act=sv.get("t0"))
        del d_h
        if self.in_ch != self.out_ch:
            dparts = bw.gn_backward(parts, d_a0, sv["prm0"], self.GroupNorm_0, B, P, silu=True)
            # the shortcut's data gradient is added onto the parts by the GEMMs themselves (no concatenated tensor, no add pass)
            bw.nin_backward(self.NIN_0, dy, sv["xs"], B, P, S, bias_sums=bsum, shared=shared,
                            accumulate_into=[(g, c) for g, (_, c) in zip(dparts, parts)])
            shared.clear()
            return dparts, dbias0
        d_x = bw.gn_backward(parts, d_a0, sv["prm0"], self.GroupNorm_0, B, P, silu=True, residual=dy)[0]
        return [d_x], dbias0

    def forward(self, x, temb=None):
        parts, B, P, spatial = _parts_of(x)
        return ops.f32b_to_ncdhw(self.forward_blocked(parts, B, P, temb), spatial)


_ = math
