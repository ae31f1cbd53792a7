"""Numerics of the fp16 + fp8-cross-term operand split ("f16f8") on the CPU (PyTorch emulation):
    a * b ~= fp16(a) * fp16(b)                                            one v_mfma_f32_32x32x16_f16
           + [ e4m3(a) * e4m3(b_lo 2^11) + e4m3(a_lo 2^11) * e4m3(b) ] 2^-11   HALF a v_mfma_scale_f32_32x32x64_f8f6f4 (K-concatenated)
 with a_lo = a - fp16(a): two 32-cycle matrix-core units per product instead of bf16x3's three.
 (1) one 128 -> 128 3x3x3 conv through Winograd F(2,3) along w, vs fp64:   bf16x3 | f16f8
 (2) --unet: one res64 U-Net evaluation (B = 1, sensitised weights, the SURVEY 7.1 experiment) where every conv that runs
     on the Winograd path (3x3x3, stride 1, >= 16^3, Cin % 32 == 0, Cout % 128 == 0) is computed in the emulated arithmetic and
     the rest in fp32, vs the same network in fp64.
    python tools/f16f8_numerics.py [--unet] [--threads 32]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def q8(x):      # OCP e4m3fn, round to nearest even, saturating at +-448 (the kernel clamps before v_cvt_pk_fp8_f32)
    return x.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def split_bf(x):
    hi = x.to(torch.bfloat16).float()
    return hi, (x - hi).to(torch.bfloat16).float()


def conv_bf16x3(t, g):
    th, tl = split_bf(t); gh, gl = split_bf(g)
    return F.conv3d(th, gh) + F.conv3d(th, gl) + F.conv3d(tl, gh)


def weight_exp(g):
    """Power-of-two pre-scale of a weight tensor for its e4m3 images: max |g| 2^e in [128, 256)."""
    m = float(g.abs().max())
    return 0 if m == 0 else 7 - int(torch.floor(torch.log2(torch.tensor(m))).item())


def conv_f16f8(t, g, sw=None):
    sw = weight_exp(g) if sw is None else sw
    th, gh = t.half().float(), g.half().float()
    tl, gl = t - th, g - gh
    cross = F.conv3d(q8(t), q8(gl * 2.0 ** (11 + sw))) + F.conv3d(q8(tl * 2.0 ** 11), q8(g * 2.0 ** sw))
    return F.conv3d(th, gh) + cross * 2.0 ** -(11 + sw)


# ---- "f16f6" study (not built): the cross terms in an MX block-scaled 6-bit format, which v_mfma_scale_f32_32x32x64_f8f6f4 runs at
# twice the e4m3 rate (32 cycles per K = 64).  A lane's K block is 32 values = [16 channels x (a, a_lo 2^11)] of one position (weights:
# one output row x 16 input channels x (g_lo, g)); the block shares ONE power-of-two scale (E8M0).
def q6(x, fmt):
    """Round to e2m3 (max 7.5) or e3m2 (max 28), nearest even, saturating; x is already divided by the block scale."""
    mbits, emin, vmax = (3, 0, 7.5) if fmt == "e2m3" else (2, -2, 28.0)
    ax = x.abs().clamp_min(1e-30)
    e = torch.floor(torch.log2(ax)).clamp_min(emin)
    step = torch.exp2(e - mbits)
    return (torch.round(x / step) * step).clamp(-vmax, vmax)


def block_scale(amax, fmt):
    emax = 2 if fmt == "e2m3" else 4
    return torch.exp2(torch.floor(torch.log2(amax.clamp_min(1e-30))) - emax)


def q6_act(t, tl11, fmt):
    """t, tl11: [B][C][...]: blocks of 16 channels per position, scale shared by the value and its scaled remainder."""
    B, C = t.shape[:2]
    sh = (B, C // 16, 16) + tuple(t.shape[2:])
    a, b = t.reshape(sh), tl11.reshape(sh)
    s = block_scale(torch.maximum(a.abs().amax(2, keepdim=True), b.abs().amax(2, keepdim=True)), fmt)
    return (q6(a / s, fmt) * s).reshape(t.shape), (q6(b / s, fmt) * s).reshape(t.shape)


def q6_wt(g, gl11, fmt):
    """g, gl11: [Co][Ci][3][3][1]: blocks of 16 input channels per (row, tap)."""
    Co, Ci = g.shape[:2]
    sh = (Co, Ci // 16, 16) + tuple(g.shape[2:])
    a, b = g.reshape(sh), gl11.reshape(sh)
    s = block_scale(torch.maximum(a.abs().amax(2, keepdim=True), b.abs().amax(2, keepdim=True)), fmt)
    return (q6(a / s, fmt) * s).reshape(g.shape), (q6(b / s, fmt) * s).reshape(g.shape)


def conv_f16f6(t, g, fmt="e2m3"):
    th, gh = t.half().float(), g.half().float()
    tq, tlq = q6_act(t, (t - th) * 2.0 ** 11, fmt)
    gq, glq = q6_wt(g, (g - gh) * 2.0 ** 11, fmt)
    return F.conv3d(th, gh) + (F.conv3d(tq, glq) + F.conv3d(tlq, gq)) * 2.0 ** -11


def equaliser_ref(gamma, beta, w, lo=-14, hi=14):
    """Restatement of md_wino_equaliser (csrc/wino_eq.hip): the static per-input-channel power-of-two s_c the f16f8 / f16f6 operand
    pass multiplies the activated operand by (the packed weights carry 1 / s_c):  s_c = 2^round(log2(g_c / a_c) / 2), with
    a_c = rms of silu(gamma_c z + beta_c) over z ~ N(0, 1) (64-point midpoint rule on [-6, 6]) and g_c = rms of w[:, c]."""
    z = (-6.0 + 12.0 * (torch.arange(64, dtype=torch.float64) + 0.5) / 64.0)
    pdf = torch.exp(-0.5 * z * z)
    y = gamma.double()[:, None] * z[None] + beta.double()[:, None]
    a2 = ((y * torch.sigmoid(y)) ** 2 * pdf[None]).sum(1) / pdf.sum()
    g2 = w.double().pow(2).mean(dim=(0, 2, 3, 4))
    e = torch.round(0.25 * (torch.log2(g2.clamp_min(1e-300)) - torch.log2(a2.clamp_min(1e-300)))).clamp(lo, hi)
    e = torch.where((a2 > 0) & (g2 > 0), e, torch.zeros_like(e))
    return torch.exp2(e).float()


def wino_conv(x, w, conv):
    """3x3x3 pad-1 conv as F(2,3) along w with `conv` for the four (3,3,1) frequency contractions."""
    B, Ci, D, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1, 1, 1)); Wd = W + 2
    d0 = xp[..., 0:Wd - 3:2]; d1 = xp[..., 1:Wd - 2:2]; d2 = xp[..., 2:Wd - 1:2]; d3 = xp[..., 3:Wd:2]
    T = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    G = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
    m = [conv(T[f].contiguous(), G[f][..., None].contiguous()) for f in range(4)]
    return torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], -1).reshape(B, w.shape[0], D, H, W)


def rel(y, ref):
    return float((y.double() - ref).norm() / ref.norm())


def one_conv():
    torch.manual_seed(0)
    B, Ci, Co, S = 1, 128, 128, 16
    x = F.silu(torch.randn(B, Ci, S, S, S) * 1.5 + 0.3)
    w = torch.randn(Co, Ci, 3, 3, 3) * (1.0 / (27 * Ci) ** 0.5)
    ref = F.conv3d(x.double(), w.double(), padding=1)
    print("one 128->128 conv at 16^3 (SiLU-like input), rel-L2 vs fp64:")
    print("  Winograd fp32        %.3e" % rel(wino_conv(x, w, F.conv3d), ref))
    print("  Winograd bf16x3      %.3e" % rel(wino_conv(x, w, conv_bf16x3), ref))
    print("  Winograd f16f8       %.3e" % rel(wino_conv(x, w, conv_f16f8), ref))
    print("  Winograd f16 + MX e2m3 cross terms (study)  %.3e" % rel(wino_conv(x, w, lambda t, g: conv_f16f6(t, g, "e2m3")), ref))
    print("  Winograd f16 + MX e3m2 cross terms (study)  %.3e" % rel(wino_conv(x, w, lambda t, g: conv_f16f6(t, g, "e3m2")), ref))
    for sw in (0, 3, 12):
        print("  Winograd f16f8, weight pre-scale 2^%-2d (auto: 2^%d)  %.3e" % (sw, weight_exp(w), rel(wino_conv(x, w, lambda t, g: conv_f16f8(t, g, sw)), ref)))


def unet(seeds, weights="sensitised", span=3.0):
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from oracle import unet_oracle as uo
    cfg = get_config_res64(); cfg.device = torch.device("cpu")
    R = cfg.data.image_size
    model = mutils.create_model(cfg, use_parallel=False)
    if weights == "trained":
        sd = synth.trained_like_state_dict(model.state_dict(), grid_mask=synth.synthetic_grid_mask(R), span=span)
        weights = f"trained-like (GroupNorm spread 2^+-{span:g})"
    else:
        sd = synth.sensitised_state_dict(model.state_dict(), seed=1234, grid_mask=synth.synthetic_grid_mask(R))
    del model
    ocfg = synth.oracle_cfg(cfg)
    real_conv3d = F.conv3d
    mode = {"m": None, "eq": False, "gn": None}
    real_gn, real_up = uo.group_norm, uo.upsample

    def gn_rec(x, w, b):             # the GroupNorm affine in front of the next conv (the static equaliser's input)
        mode["gn"] = (w, b)
        return real_gn(x, w, b)

    def up_rec(p, x):                # Upsample: no GroupNorm in front of its conv -> the product path runs it in bf16x3
        mode["gn"] = None
        return real_up(p, x)

    def patched(x, w, b=None, stride=1, padding=0, *a, **k):
        st = stride if isinstance(stride, int) else stride[0]
        pd = padding if isinstance(padding, int) else padding[0]
        if (mode["m"] is not None and x.dtype == torch.float32 and tuple(w.shape[2:]) == (3, 3, 3) and st == 1 and pd == 1
                and x.shape[-1] >= 16 and w.shape[1] % 32 == 0 and w.shape[0] % 128 == 0 and not a and not k):
            gn, mode["gn"] = mode["gn"], None
            if mode["eq"]:
                if gn is None or gn[0].numel() != w.shape[1]:
                    y = wino_conv(x, w, conv_bf16x3)               # un-normalised operand: bf16x3 (hip_ops.wino_f8_ok)
                else:
                    s = equaliser_ref(gn[0], gn[1], w).view(1, -1, 1, 1, 1)
                    y = wino_conv(x * s, w / s, mode["m"])
            else:
                y = wino_conv(x, w, mode["m"])
            return y if b is None else y + b[None, :, None, None, None]
        return real_conv3d(x, w, b, stride, padding, *a, **k)

    for seed in seeds:
        x = synth.synthetic_inputs(1, 4, R, seed=seed) * synth.synthetic_grid_mask(R).view(1, 1, R, R, R)
        lab = torch.tensor([500.3])
        with torch.no_grad():
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            te = uo.timestep_embedding
            uo.timestep_embedding = lambda t, dim: te(t, dim).double()      # the oracle builds it in fp32
            try:
                ref = uo.unet_res64_forward(sd64, ocfg, x.double(), lab.double())
            finally:
                uo.timestep_embedding = te
            F.conv3d, uo.group_norm, uo.upsample = patched, gn_rec, up_rec
            try:
                out = {}
                f6 = lambda t, g: conv_f16f6(t, g, "e2m3")  # noqa: E731
                for name, m, eq in (("fp32", None, False), ("bf16x3 (Winograd convs)", conv_bf16x3, False),
                                    ("f16f8", conv_f16f8, False), ("f16f6", f6, False),
                                    ("f16f8 + static equaliser, raw operands bf16x3", conv_f16f8, True),
                                    ("f16f6 + static equaliser, raw operands bf16x3", f6, True)):
                    mode["m"], mode["eq"] = m, eq
                    out[name] = rel(uo.unet_res64_forward(sd, ocfg, x, lab), ref)
                    print(f"  {name:48s} {out[name]:.3e}", flush=True)
            finally:
                F.conv3d, uo.group_norm, uo.upsample = real_conv3d, real_gn, real_up
        print(f"^ res64 U-Net evaluation, {weights} weights, B = 1, t = 500.3, input seed {seed}: rel-L2 of eps vs fp64")


def audit(weights):
    """How good is the equaliser's STATIC estimate?  One res64 U-Net evaluation on the CPU (fp32 oracle): for every GroupNorm -> SiLU ->
    3x3x3 conv pair the per-channel rms of the activated operand the conv really sees against a_c = rms of silu(gamma_c z + beta_c),
    z ~ N(0, 1) -- GroupNorm normalises a GROUP, not a channel, so a channel's own mean / variance inside its group are what the static
    model cannot know -- and the spread (max / min over the 16 channels of a K block) of the equalised operand, which is what e2m3's
    four binades have to hold."""
    from meshdiffusion_amd import synth
    from meshdiffusion_amd.config import get_config_res64
    from meshdiffusion_amd.lib.diffusion.models import ddpm_res64, utils as mutils  # noqa: F401
    from oracle import unet_oracle as uo
    cfg = get_config_res64(); cfg.device = torch.device("cpu")
    R = cfg.data.image_size
    model = mutils.create_model(cfg, use_parallel=False)
    make = synth.trained_like_state_dict if weights == "trained" else synth.sensitised_state_dict
    sd = make(model.state_dict(), grid_mask=synth.synthetic_grid_mask(R))
    del model
    real_conv3d, real_gn, real_up, real_down = F.conv3d, uo.group_norm, uo.upsample, uo.downsample
    st = {"gn": None, "rows": []}

    def down_rec(p, x):              # Downsample: no GroupNorm in front either (md_conv3_s2, bf16x3)
        st["gn"] = None
        return real_down(p, x)

    def gn_rec(x, w, b):
        st["gn"] = (w, b)
        return real_gn(x, w, b)

    def up_rec(p, x):
        st["gn"] = None
        return real_up(p, x)

    def conv_rec(x, w, b=None, stride=1, padding=0, *a, **k):
        gn, st["gn"] = st["gn"], None
        if gn is not None and tuple(w.shape[2:]) == (3, 3, 3) and gn[0].numel() == w.shape[1] and w.shape[1] % 16 == 0 and x.shape[-1] >= 16:
            s = equaliser_ref(gn[0], gn[1], w)
            z = -6.0 + 12.0 * (torch.arange(64, dtype=torch.float64) + 0.5) / 64.0
            pdf = torch.exp(-0.5 * z * z)
            y = gn[0].double()[:, None] * z[None] + gn[1].double()[:, None]
            a_est = (((y * torch.sigmoid(y)) ** 2 * pdf[None]).sum(1) / pdf.sum()).sqrt()
            a_real = x.double().pow(2).mean(dim=(0, 2, 3, 4)).sqrt()
            ratio = (a_real / a_est.clamp_min(1e-30))
            def blockspread(v):
                vb = v.reshape(-1, 16)
                return float((vb.max(1).values / vb.min(1).values.clamp_min(1e-30)).max())
            st["rows"].append((tuple(w.shape[:2]), x.shape[-1], float(ratio.min()), float(ratio.median()), float(ratio.max()),
                               blockspread(a_real), blockspread(a_real * s.double())))
        return real_conv3d(x, w, b, stride, padding, *a, **k)

    x = synth.synthetic_inputs(1, 4, R, seed=42) * synth.synthetic_grid_mask(R).view(1, 1, R, R, R)
    F.conv3d, uo.group_norm, uo.upsample, uo.downsample = conv_rec, gn_rec, up_rec, down_rec
    try:
        with torch.no_grad():
            uo.unet_res64_forward(sd, synth.oracle_cfg(cfg), x, torch.tensor([500.3]))
    finally:
        F.conv3d, uo.group_norm, uo.upsample, uo.downsample = real_conv3d, real_gn, real_up, real_down
    print(f"equaliser audit, {weights} weights: measured / estimated per-channel operand rms (min, median, max) and the worst 16-channel "
          "block spread of the operand before -> after equalisation")
    for (co, ci), S, lo, med, hi, sp0, sp1 in st["rows"]:
        print(f"  {ci:4d}->{co:4d} @{S:2d}^3   rms ratio {lo:5.2f} {med:5.2f} {hi:5.2f}   block spread {sp0:9.1f} -> {sp1:6.1f}")
    r = st["rows"]
    print(f"  over {len(r)} pairs: rms ratio in [{min(v[2] for v in r):.2f}, {max(v[4] for v in r):.2f}]; worst block spread "
          f"{max(v[5] for v in r):.0f} -> {max(v[6] for v in r):.1f}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--unet", action="store_true")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--seeds", default="42")
    ap.add_argument("--weights", default="sensitised", choices=["sensitised", "trained"],
                    help="synth.sensitised_state_dict (i.i.d.) or synth.trained_like_state_dict (heavy tails, 2^U(-3,3) GroupNorm gammas)")
    ap.add_argument("--span", type=float, default=3.0, help="--weights trained: per-channel GroupNorm scales 2^U(-span, span)")
    ap.add_argument("--audit", action="store_true", help="compare the equaliser's static activation estimate with the operands of one forward")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    if a.audit:
        audit(a.weights)
        sys.exit(0)
    one_conv()
    if a.unet:
        unet([int(s) for s in a.seeds.split(",")], a.weights, a.span)
