"""Winograd F(2,3)-along-w path of the 3x3x3 convolution (md_wino_prep + md_wino_pack_weights + md_conv3_wino) against
PyTorch fp32 on the CPU: nn.GroupNorm + nn.SiLU + nn.Conv3d (lib/diffusion/models/layers.py:676-681, :118-124), the channel
concat of two parts (ddpm_res64.py:174-176) and the nearest-x2 upsampled input (layers.py:618-623).  Tolerance: the bf16x3
budget of the direct kernels (3e-5 rel-L2); the transform adds ~1e-6 (oracle-side measurement in DESIGN.md)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_MFMA = 3e-5


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from meshdiffusion_amd import hip_ops
    return hip_ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _split_bf16(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


def test_wino_prep_layout_and_values(ops):
    """T[b][c/8][f][plane][z][y][pair][8]: the four transformed inputs of every output pair, zero padded AFTER the
    activation, split into bf16 hi / lo exactly like md_split2 (RNE) -- bit-exact against the same arithmetic in torch."""
    B, S, cs = 2, 8, [16, 8]
    cin = sum(cs)
    xs = [_rand((B, c, S, S, S), 30 + i) for i, c in enumerate(cs)]
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    t = ops.wino_prep(parts, None, False, False, B, S).cpu()
    t = t.view(B, cin // 8, 4, 2, S, S, S // 2, 8)
    x = torch.cat(xs, 1)
    xp = F.pad(x, (1, 1))                                    # w only: d_k = x[2i - 1 + k]
    d = [xp[..., k:k + S:2] for k in range(4)]               # each [B, C, S, S, S/2]
    tr = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]
    for f in range(4):
        hi, lo = _split_bf16(tr[f])
        for plane, ref in enumerate((hi, lo)):
            got = t[:, :, f, plane]                          # [B, C/8, S, S, S/2, 8]
            want = ref.view(B, cin // 8, 8, S, S, S // 2).permute(0, 1, 3, 4, 5, 2)
            assert torch.equal(got.view(torch.int16), want.contiguous().view(torch.int16)), (f, plane)


@pytest.mark.parametrize("case", ["plain_128", "two_parts_gn_silu_res_stats", "ups_256rows", "gn_no_silu_k64"])
def test_conv3_wino_vs_torch(ops, case):
    cfgs = {
        "plain_128": dict(cs=[128], cout=128, S=16, B=2, gn=False, silu=False, ups=False, res=False, stats=False),
        "two_parts_gn_silu_res_stats": dict(cs=[96, 32], cout=128, S=16, B=2, gn=True, silu=True, ups=False, res=True, stats=True),
        "ups_256rows": dict(cs=[64], cout=256, S=16, B=1, gn=False, silu=False, ups=True, res=False, stats=True),
        "gn_no_silu_k64": dict(cs=[32, 32], cout=128, S=8, B=3, gn=True, silu=False, ups=False, res=True, stats=False),
    }
    c = cfgs[case]
    cs, cout, S, B = c["cs"], c["cout"], c["S"], c["B"]
    cin = sum(cs)
    Sin = S // 2 if c["ups"] else S
    xs = [_rand((B, k, Sin, Sin, Sin), 40 + i) * (1.0 + i) + 0.3 * i for i, k in enumerate(cs)]
    x = torch.cat(xs, 1)
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 50), 0.5 * _rand((cin,), 51)
    w = _rand((cout, cin, 3, 3, 3), 52, 0.05)
    bias = _rand((B, cout), 53)
    res = _rand((B, cout, S, S, S), 54) if c["res"] else None
    parts = [(ops.ncdhw_to_f32b(t.cuda()), k) for t, k in zip(xs, cs)]
    ac = None
    ref_in = x
    if c["gn"]:
        _, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, Sin ** 3, want_ac=True)
        ref_in = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        ref_in = F.silu(ref_in) if c["silu"] else ref_in
    if c["ups"]:
        ref_in = F.interpolate(ref_in, scale_factor=2, mode="nearest")
    ww = ops.WinoWeight(w.cuda(), "cuda")
    t = ops.wino_prep(parts, ac, c["silu"], c["ups"], B, S)
    stats = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda") if c["stats"] else None
    res_f = ops.ncdhw_to_f32b(res.cuda()) if res is not None else None
    out = ops.conv3_wino(ww, t, B, S, bias=bias.cuda(), bias_bstride=cout, residual=res_f,
                         res_bstride=cout * S ** 3 if res is not None else 0, stats=stats)
    y = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()
    ref = F.conv3d(ref_in, w, padding=1) + bias[:, :, None, None, None]
    if res is not None:
        ref = ref + res
    e = rel_l2(y, ref)
    print(f"wino conv ({case}): vs torch fp32 {e:.2e}")
    assert e < TOL_MFMA
    if stats is not None:
        st = stats.cpu()
        yd = y.double()
        assert rel_l2(st[..., 0], yd.sum(dim=(2, 3, 4))) < 1e-6
        assert rel_l2(st[..., 1], (yd * yd).sum(dim=(2, 3, 4))) < 1e-6


def test_conv3_wino_full_tiles_bitwise_repeatable_and_vs_direct(ops):
    """256 -> 128 at 32^3, B = 2 (512 workgroups, two per CU in sequence): against the direct fused kernel on the same
    inputs (both bf16x3: they differ only by the transform rounding) and bit-identical between two launches (no atomics,
    no races between the private LDS-DMA buffers of the four waves)."""
    B, S, cs, cout = 2, 32, [128, 128], 128
    cin = sum(cs)
    xs = [_rand((B, k, S, S, S), 60 + i) for i, k in enumerate(cs)]
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 62), 0.5 * _rand((cin,), 63)
    w = _rand((cout, cin, 3, 3, 3), 64, 0.03)
    bias = _rand((B, cout), 65)
    parts = [(ops.ncdhw_to_f32b(t.cuda()), k) for t, k in zip(xs, cs)]
    _, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, S ** 3, want_ac=True)
    ww = ops.WinoWeight(w.cuda(), "cuda")
    outs = []
    for _ in range(2):
        t = ops.wino_prep(parts, ac, True, False, B, S)
        outs.append(ops.conv3_wino(ww, t, B, S, bias=bias.cuda(), bias_bstride=cout).clone())
    assert torch.equal(outs[0], outs[1])
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_128_FAST, "cuda")
    direct = ops.f32b_empty(B, cout, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_128_FAST, a=pw.data, b=None, out=direct, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S, S, S), bias=bias.cuda(), bias_bstride=cout, b_f32=dict(parts=parts, ac=ac, silu=True))
    e = rel_l2(outs[0].cpu(), direct.cpu())
    print(f"wino vs direct fused kernel: {e:.2e}")
    assert e < 2e-5


@pytest.mark.parametrize("kind", ["conv", "conv_dgrad"])
def test_wino_pack_weights_layout(ops, kind):
    """md_wino_pack_weights: tiles [rows/128][K/16][kd*3+kh][f 4][row tile 4][plane 2][k-group 2][row 32][8 bf16] of
    G g = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) along kw, split into bf16 hi / lo (RNE) -- bit-exact against the same
    arithmetic in torch, for the forward weights and for the data-gradient weights W'[ci][co][t] = W[co][ci][26 - t]."""
    co, ci = (128, 64) if kind == "conv" else (96, 256)
    w = _rand((co, ci, 3, 3, 3), 80, 0.1)
    ww = ops.WinoWeight(w.cuda(), "cuda", kind=kind)
    weff = w if kind == "conv" else w.reshape(co, ci, 27).flip(2).transpose(0, 1).reshape(ci, co, 3, 3, 3)
    rows, K = weff.shape[0], weff.shape[1]
    assert (ww.rows, ww.kdim) == (rows, K)
    g0, g1, g2 = weff[..., 0], weff[..., 1], weff[..., 2]
    G = torch.stack([g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2], -1)          # [rows, K, 3, 3, 4] fp32
    hi, lo = _split_bf16(G)
    planes = torch.stack([hi, lo], 0)                                                     # [plane, rows, K, kd, kh, f]
    t = planes.reshape(2, rows // 128, 4, 32, K // 16, 2, 8, 9, 4)                        # plane, ct, rt, row, chunk, h, e, tap, f
    want = t.permute(1, 4, 7, 8, 2, 0, 5, 3, 6).contiguous()                             # ct, chunk, tap, f, rt, plane, h, row, e
    got = ww.data.cpu().view(rows // 128, K // 16, 9, 4, 4, 2, 2, 32, 8)
    assert torch.equal(got.view(torch.int16), want.view(torch.int16))


def test_conv3_wino_data_gradient_weights(ops):
    """kind "conv_dgrad": the tiles of W'[ci][co][t] = W[co][ci][26 - t] packed in place from W -- the Winograd conv of dy with
    them is the input gradient of F.conv3d (training backward, lib/diffusion/models/backward.py conv3_backward)."""
    B, S, ci, co = 2, 16, 128, 160
    w = _rand((co, ci, 3, 3, 3), 70, 0.05)
    dy = _rand((B, co, S, S, S), 71)
    ww = ops.WinoWeight(w.cuda(), "cuda", kind="conv_dgrad")
    assert (ww.rows, ww.kdim) == (ci, co)
    t = ops.wino_prep([(ops.ncdhw_to_f32b(dy.cuda()), co)], None, False, False, B, S)
    dx = ops.f32b_to_ncdhw(ops.conv3_wino(ww, t, B, S), (S, S, S)).cpu()
    ref = torch.nn.grad.conv3d_input((B, ci, S, S, S), w, dy, padding=1)
    e = rel_l2(dx, ref)
    print(f"wino data gradient: vs torch {e:.2e}")
    assert e < TOL_MFMA


@pytest.mark.parametrize("case", ["raw_two_parts", "gn_silu", "ups", "gn_silu_dropout"])
def test_wino_prep_v2_is_bit_identical(ops, case):
    """The two-phase operand pass (md_wino_prep_v2, csrc/wino_prep2.hip: the default where W divides 256) writes the same T,
    bit for bit, as the one-thread-per-pair md_wino_prep."""
    B, S = 2, 16
    cs = [16, 8] if case == "raw_two_parts" else [32]
    cin = sum(cs)
    ups = case == "ups"
    Sin = S // 2 if ups else S
    xs = [_rand((B, c, Sin, Sin, Sin), 90 + i) for i, c in enumerate(cs)]
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    ac, silu, drop = None, False, None
    if case.startswith("gn_silu"):
        gamma, beta = 1.0 + 0.2 * _rand((cin,), 92), 0.5 * _rand((cin,), 93)
        _, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, Sin ** 3, want_ac=True)
        silu = True
    if case == "gn_silu_dropout":
        drop = (0.1, 12345)
    keep = ops.WINO_PREP_V2
    try:
        ops.WINO_PREP_V2 = False
        t1 = ops.wino_prep(parts, ac, silu, ups, B, S, drop=drop).clone()
        ops.WINO_PREP_V2 = True
        t2 = ops.wino_prep(parts, ac, silu, ups, B, S, drop=drop).clone()
    finally:
        ops.WINO_PREP_V2 = keep
    assert torch.equal(t1.view(torch.int16), t2.view(torch.int16))
    assert float(t1.float().abs().sum()) > 0


def test_conv3_wino_rejects_unsupported_shapes(ops):
    from meshdiffusion_amd import _lib
    lib = _lib.load()
    assert lib.md_wino_weight_bytes(96, 64) < 0          # cout % 128
    assert lib.md_wino_weight_bytes(128, 48) < 0         # cin % 32
    assert lib.md_wino_operand_bytes(1, 32, 8, 8, 7) < 0
    t = torch.zeros(16, device="cuda")
    rc = lib.md_conv3_wino(t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 0, None, 0, None, 1, 32, 128, 6, 8, 8, 0, None)
    assert rc == -2          # MD_ERR_UNSUPPORTED


_ = np


@pytest.mark.parametrize("lift", ["dyn", "const64"])
@pytest.mark.parametrize("mag", [1.0, 1e-3, 3e-6, 1e-9])
def test_conv3_wino_data_gradient_f16f6(ops, mag, lift):
    """Round 6, training backward: the data-gradient conv of a layer on the Winograd path in f16f6 -- md_absmax + md_wino_prep_dual_f6
    (T = the f16f6 operand of 2^k x dy with max |dy| 2^k in [16, 32); U and the channel sums those of md_wino_prep_dual, bit for bit),
    the flipped f16f6 fragments from an MD_PACK_WINO_F6 job of md_pack_batch (fixed pre-scale), md_conv3_wino_f6_scaled.  Against torch
    float64 at gradient magnitudes from 1 down to 1e-9: the dynamic lift is flat in the magnitude; the constant one (2^6, the A/B
    form) loses the fp16 plane below ~1e-6."""
    B, S, ci, co = 2, 16, 128, 160
    w = _rand((co, ci, 3, 3, 3), 70, 0.05)
    dy = _rand((B, co, S, S, S), 71) * mag
    dyb = ops.ncdhw_to_f32b(dy.cuda())
    parts = [(dyb, co)]
    sums0, sums1 = torch.zeros((B, co), device="cuda"), torch.zeros((B, co), device="cuda")
    _, u0 = ops.wino_prep(parts, None, False, False, B, S, dual=True, sums=sums0)
    u0 = u0.clone()
    amax = ops.absmax_word(dyb) if lift == "dyn" else None
    if amax is not None:
        assert float(amax.view(torch.float32).item()) == float(dy.abs().max())
    t, u1 = ops.wino_prep(parts, None, False, False, B, S, dual=True, sums=sums1, f8="f6", tscale=64.0, amax=amax)
    assert torch.equal(u0.view(torch.int16), u1.view(torch.int16))
    assert rel_l2(sums1.cpu(), sums0.cpu()) < 1e-6
    ww = ops.WinoWeightF6Dgrad(w.cuda(), "cuda")
    assert (ww.rows, ww.kdim) == (ci, co)
    dx = ops.f32b_to_ncdhw(ops.conv3_wino(ww, t, B, S, out_scale=1.0 if amax is not None else 1.0 / 64.0, amax=amax), (S, S, S)).cpu()
    ref = torch.nn.grad.conv3d_input((B, ci, S, S, S), w.double(), dy.double(), padding=1)
    e = rel_l2(dx, ref)
    t3 = ops.wino_prep(parts, None, False, False, B, S)
    dx3 = ops.f32b_to_ncdhw(ops.conv3_wino(ops.WinoWeight(w.cuda(), "cuda", kind="conv_dgrad"), t3, B, S), (S, S, S)).cpu()
    print(f"f16f6 data gradient (|dy| ~ {mag:g}, lift {lift}): vs torch fp64 {e:.2e}; bf16x3 {rel_l2(dx3, ref):.2e}")
    assert rel_l2(dx3, ref) < TOL_MFMA
    if lift == "dyn" or mag >= 3e-6:
        assert e < 4e-5
    if mag == 1.0:
        # the packed fragments against the forward-orientation packer on the materialised flipped tensor W'[ci][co][t] = W[co][ci][26 - t]
        wt = torch.flip(w, dims=(2, 3, 4)).permute(1, 0, 2, 3, 4).contiguous()
        t_plain = ops.wino_prep(parts, None, False, False, B, S, f8="f6")
        dx_plain = ops.f32b_to_ncdhw(ops.conv3_wino(ops.WinoWeightF8(wt.cuda(), "cuda", "f6"), t_plain, B, S), (S, S, S)).cpu()
        assert rel_l2(dx_plain, ref) < 4e-5
        # a constant lift far outside its range saturates at the fp16 ends -- clipped elements, never inf / NaN
        big = [(ops.ncdhw_to_f32b((dy * 1e4).cuda()), co)]
        tb, _ = ops.wino_prep(big, None, False, False, B, S, dual=True, sums=None, f8="f6", tscale=64.0)
        assert bool(torch.isfinite(ops.conv3_wino(ww, tb, B, S, out_scale=1.0 / 64.0)).all())


def test_wino_prep_dual_second_output_bit_exact(ops):
    """md_wino_prep_dual: T as md_wino_prep_v2 writes it, and U[b][c/8][f][plane][z][y][pair][8] = (v0, v0 + v1, v0 - v1, v1) of
    every output pair (v0 = x[2i], v1 = x[2i+1]) split into bf16 hi / lo -- bit-exact against the same arithmetic in torch."""
    B, S, cin = 2, 32, 16
    x = _rand((B, cin, S, S, S), 31)
    parts = [(ops.ncdhw_to_f32b(x.cuda()), cin)]
    t_ref = ops.wino_prep(parts, None, False, False, B, S).clone()
    sums = torch.zeros((B, cin), device="cuda")
    t, u = ops.wino_prep(parts, None, False, False, B, S, dual=True, sums=sums)
    assert torch.equal(t.view(torch.int16), t_ref.view(torch.int16))
    ref_s = x.double().sum(dim=(2, 3, 4))                      # the same pass adds up the channels (bias gradients)
    assert float((sums.cpu().double() - ref_s).abs().max()) < 1e-4 * float(x.double().abs().sum(dim=(2, 3, 4)).max())
    u = u.cpu().view(B, cin // 8, 4, 2, S, S, S // 2, 8)
    v0, v1 = x[..., 0::2], x[..., 1::2]
    tr = [v0, v0 + v1, v0 - v1, v1]
    for f in range(4):
        hi, lo = _split_bf16(tr[f])
        for plane, ref in enumerate((hi, lo)):
            want = ref.view(B, cin // 8, 8, S, S, S // 2).permute(0, 1, 3, 4, 5, 2)
            assert torch.equal(u[:, :, f, plane].view(torch.int16), want.contiguous().view(torch.int16)), (f, plane)


@pytest.mark.parametrize("case", ["k128_32", "k256_64_b1", "k128_r256_32_b3"])
def test_wgrad_wino_vs_torch(ops, case):
    """md_wgrad_wino (Winograd-domain weight gradient from the forward's operand T and md_wino_prep_dual's U) against
    torch.nn.grad.conv3d_weight in float64 on sub-sampled channels and against the PB16 kernel md_wgrad on all of them: rows
    of 16 pairs (32^3) and 32 pairs (64^3), several K ranges (planes split unevenly), 1-2 tiles per operand, accumulation
    into a non-zero dW."""
    from meshdiffusion_amd.lib.diffusion.models import backward as bw
    cfg = {"k128_32": dict(B=2, ci=128, co=128, S=32), "k256_64_b1": dict(B=1, ci=256, co=128, S=64),
           "k128_r256_32_b3": dict(B=3, ci=128, co=256, S=32)}[case]
    B, ci, co, S = cfg["B"], cfg["ci"], cfg["co"], cfg["S"]
    P = S ** 3
    a = _rand((B, ci, S, S, S), 70)
    dy = _rand((B, co, S, S, S), 71, 0.1)
    a_f, dy_f = ops.ncdhw_to_f32b(a.cuda()), ops.ncdhw_to_f32b(dy.cuda())
    t_act = ops.wino_prep([(a_f, ci)], None, False, False, B, S, keep=True)
    t_dy, u_dy = ops.wino_prep([(dy_f, co)], None, False, False, B, S, dual=True)
    dw0 = _rand((co, ci, 3, 3, 3), 72, 0.01).cuda()
    dw = dw0.clone()
    ops.wgrad_wino(u_dy, t_act, B, co, ci, S, dw)
    got = (dw - dw0).cpu()
    # (a) the PB16 kernel (direct 27-tap contraction, bf16x3) on the same tensors
    ref_k = torch.zeros((co, ci, 3, 3, 3), device="cuda")
    dy_pb = bw.to_pb16(dy_f, B, co, S, 0, zhalo=False)
    act_pb = bw.to_pb16(bw.split_f32b(a_f, B, ci, P), B, ci, S, 1)
    bw.wgrad(dy_pb, act_pb, B, co, ci, S, 27, ref_k, ci * 27, 27, 1)
    e_k = rel_l2(got, ref_k.cpu())
    # (b) float64 autograd of nn.Conv3d for 4 output x 6 input channels spread over the tiles
    cos, cis = [0, 37, co // 2 + 5, co - 1], [0, 1, 63, ci // 2, ci - 2, ci - 1]
    ref = torch.nn.grad.conv3d_weight(a[:, cis].double(), (len(cos), len(cis), 3, 3, 3), dy[:, cos].double(), padding=1)
    sub = got[cos][:, cis].double()
    e = rel_l2(sub, ref)
    per_tap = [(t, rel_l2(sub.reshape(len(cos), len(cis), 27)[..., t], ref.reshape(len(cos), len(cis), 27)[..., t])) for t in range(27)]
    worst = max(per_tap, key=lambda v: v[1])
    print(f"wgrad_wino ({case}): vs float64 autograd {e:.2e} (worst tap {worst}), vs md_wgrad {e_k:.2e}")
    assert e < TOL_MFMA and worst[1] < 2 * TOL_MFMA and e_k < TOL_MFMA
    dw2 = dw0.clone()
    ops.wgrad_wino(u_dy, t_act, B, co, ci, S, dw2)
    assert torch.equal(dw2, dw)                              # fixed reduction order: run-to-run identical


@pytest.mark.parametrize("case", ["k256_r128_32", "k128_r256_odd_batch"])
def test_wgrad_nin_s16b_vs_torch(ops, case):
    """md_wgrad_nin: NIN weight gradient dW[ci][co] = sum_{b, p} x[ci] dy[co] straight from the S16B tensors (no PB16 operands)
    against float64 einsum and against md_to_pb16 + md_wgrad (taps = 1); accumulates into a non-zero dW; repeatable."""
    from meshdiffusion_amd.lib.diffusion.models import backward as bw
    cfg = {"k256_r128_32": dict(B=2, ci=256, co=128, S=32), "k128_r256_odd_batch": dict(B=3, ci=128, co=256, S=16)}[case]
    B, ci, co, S = cfg["B"], cfg["ci"], cfg["co"], cfg["S"]
    P = S ** 3
    x = _rand((B, ci, S, S, S), 80)
    dy = _rand((B, co, S, S, S), 81, 0.1)
    x_f, dy_f = ops.ncdhw_to_f32b(x.cuda()), ops.ncdhw_to_f32b(dy.cuda())
    xs, dys = bw.split_f32b(x_f, B, ci, P), bw.split_f32b(dy_f, B, co, P)
    dw0 = _rand((ci, co), 82, 0.01).cuda()
    dw = dw0.clone()
    ops.wgrad_nin(dys, xs, B, co, ci, P, dw)
    got = (dw - dw0).cpu().double()
    ref = torch.einsum("bip,bop->io", x.double().reshape(B, ci, P), dy.double().reshape(B, co, P))
    e = rel_l2(got, ref)
    ref_k = torch.zeros((ci, co), device="cuda")
    bw.wgrad_nin(bw.to_pb16(dy_f, B, co, S, 0, zhalo=False), xs, B, co, ci, S, ref_k)
    e_k = rel_l2(got, ref_k.cpu().double())
    print(f"wgrad_nin ({case}): vs float64 {e:.2e}, vs md_to_pb16 + md_wgrad {e_k:.2e}")
    assert e < TOL_MFMA and e_k < TOL_MFMA
    dw2 = dw0.clone()
    ops.wgrad_nin(dys, xs, B, co, ci, P, dw2)
    assert torch.equal(dw2, dw)


# ---------------------------------------------------------------------------------------------------------------------
# "f16f8" arithmetic of the inference Winograd convs (md_wino_prep_f8 + md_wino_pack_weights_f8 + md_conv3_wino_f8):
# product = fp16(a) fp16(b) + [e4m3(a) e4m3(b_lo 2^11) + e4m3(a_lo 2^11) e4m3(b)] 2^-11.  The torch restatement below is the
# arithmetic of tools/f16f8_numerics.py (OCP e4m3 through torch.float8_e4m3fn, RNE, saturating).
# ---------------------------------------------------------------------------------------------------------------------
def _q8(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn)


def _f16f8_images(x):
    """hi (fp16), e4m3(x), e4m3((x - hi) 2^11) of an fp32 tensor."""
    hi = x.clamp(-65504, 65504).half()
    return hi, _q8(x), _q8((x - hi.float()) * 2048.0)


def _wino_T(x):
    S = x.shape[-1]
    xp = F.pad(x, (1, 1))
    d = [xp[..., k:k + S:2] for k in range(4)]
    return [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]


def _wino_G(w):
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    return [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]


def test_wino_prep_f8_layout_and_values(ops):
    """T of md_wino_prep_f8: plane 0 = 8 fp16 (RNE), plane 1 = [e4m3(t) x 8 | e4m3((t - fp16(t)) 2^11) x 8], bit-exact against
    the same arithmetic in torch (two parts, zero padding after the activation)."""
    B, S, cs = 2, 8, [16, 8]
    cin = sum(cs)
    xs = [_rand((B, c, S, S, S), 130 + i) * (3.0 if i else 0.7) for i, c in enumerate(cs)]
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    t = ops.wino_prep(parts, None, False, False, B, S, f8=True).cpu()
    t = t.view(B, cin // 8, 4, 2, S, S, S // 2, 8)            # 8 bf16-sized slots = 16 bytes
    tr = _wino_T(torch.cat(xs, 1))
    for f in range(4):
        hi, q, ql = _f16f8_images(tr[f])
        blk = lambda v: v.view(B, cin // 8, 8, S, S, S // 2).permute(0, 1, 3, 4, 5, 2).contiguous()      # noqa: E731
        assert torch.equal(t[:, :, f, 0].contiguous().view(torch.int16), blk(hi).view(torch.int16)), f
        got = t[:, :, f, 1].contiguous().view(torch.uint8).view(B, cin // 8, S, S, S // 2, 16)
        assert torch.equal(got[..., :8], blk(q).view(torch.uint8)), f
        assert torch.equal(got[..., 8:], blk(ql).view(torch.uint8)), f


def _pack_f8_reference(w):
    """Torch restatement of md_wino_pack_weights_f8: (fragments as uint8 [n_items][16], sw)."""
    cout, cin = w.shape[:2]
    amax = float(w.abs().max())
    sw = 0 if amax == 0 else 7 - int(np.floor(np.log2(1.5 * np.float32(amax))))
    G = torch.stack(_wino_G(w.reshape(cout, cin, 9, 3)), -1) * (2.0 ** sw)        # [co][ci][tap 9][f 4]
    hi, q, ql = _f16f8_images(G)
    nch, npairs = cin // 16, cin // 16 * 9 // 2
    out = torch.zeros((cout // 128, npairs, 4, 4, 4, 64, 16), dtype=torch.uint8)
    step = lambda s: (s // 9, s % 9)                                                   # noqa: E731
    lane = torch.arange(64)
    row, h = lane % 32, lane // 32
    for rtb in range(cout // 128):
        for rt in range(4):
            co = (rtb * 4 + rt) * 32 + row                                             # [64]
            for p in range(npairs):
                for piece in range(4):
                    for f in range(4):
                        if piece < 2:
                            ch, tap = step(2 * p + piece)
                            ci = ch * 16 + h[:, None] * 8 + torch.arange(8)[None]      # [64][8]
                            v = hi[co[:, None], ci, tap, f]                            # [64][8] fp16
                            out[rtb, p, f, rt, piece] = v.contiguous().view(torch.uint8).view(64, 16)
                        else:
                            s_l = 2 * p + h                                            # per lane
                            ch, tap = s_l // 9, s_l % 9
                            ci = ch[:, None] * 16 + (piece - 2) * 8 + torch.arange(8)[None]
                            a = ql[co[:, None], ci, tap[:, None], f].view(torch.uint8)
                            b = q[co[:, None], ci, tap[:, None], f].view(torch.uint8)
                            out[rtb, p, f, rt, piece] = torch.cat([a, b], 1)
    assert nch * 9 == 2 * npairs
    return out.view(-1, 16), sw


def test_wino_pack_weights_f8_layout(ops):
    """md_wino_pack_weights_f8: pre-scale from max |w| (device side), fp16 fragments of both steps of a pair, the K-concatenated
    e4m3 fragment [lo | plain] per 8 channels with lane half h owning step 2p + h -- bit-exact against torch, header included."""
    co, ci = 128, 64
    w = _rand((co, ci, 3, 3, 3), 140, 0.04)
    w[3, 5] *= 1e-4                                            # a nearly dead filter: fp16 subnormals without the pre-scale
    ww = ops.WinoWeightF8(w.cuda(), "cuda")
    raw = ww.data.cpu().view(torch.uint8)
    want, sw = _pack_f8_reference(w)
    n = co * ci * 9
    assert raw.numel() == n * 16 + 256
    hdr = raw[n * 16:n * 16 + 16].view(torch.int32)
    assert hdr[0].view(torch.float32).item() == float(w.abs().max()) and int(hdr[1]) == sw
    assert hdr[2].view(torch.float32).item() == 2.0 ** -sw
    got = raw[:n * 16].view(n, 16)
    bad = (got != want).any(dim=1).nonzero().flatten()
    assert bad.numel() == 0, (bad[:8].tolist(), got[bad[:2]].tolist(), want[bad[:2]].tolist())


def _conv_f16f8_reference(ref_in, w):
    """The f16f8 Winograd conv restated in torch (fp32 accumulation by F.conv3d on exact operand images)."""
    B, _, S = ref_in.shape[:3]
    cout = w.shape[0]
    amax = float(w.abs().max())
    sw = 7 - int(np.floor(np.log2(1.5 * np.float32(amax))))
    T = _wino_T(F.pad(ref_in, (0, 0, 1, 1, 1, 1)))           # z / y zero padding; _wino_T pads w
    G = [g * (2.0 ** sw) for g in _wino_G(w)]
    m = []
    for f in range(4):
        th, tq, tl = _f16f8_images(T[f])
        gh, gq, gl = _f16f8_images(G[f][..., None])
        cross = F.conv3d(tq.float(), gl.float()) + F.conv3d(tl.float(), gq.float())
        m.append(F.conv3d(th.float(), gh.float()) + cross * 2.0 ** -11)
    y0, y1 = (m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]
    return torch.stack([y0, y1], -1).reshape(B, cout, S, S, S) * 2.0 ** -sw


def _q6_e2m3(x):
    """RNE onto OCP e2m3 (steps 1/8 below 2, 1/4 below 4, 1/2 below 8), saturating at 7.5 -- what v_cvt_scalef32_2xpk16_fp6_f32 does
    (tools/probes/f6_probe.hip: 0 of 1984 random values differ)."""
    e = torch.floor(torch.log2(x.abs().clamp_min(1e-30))).clamp(0, 2)
    step = torch.exp2(e - 3)
    return (torch.round(x / step) * step).clamp(-7.5, 7.5)


def _f16f6_images(x, ch_dim):
    """hi (fp16) and the MX e2m3 images of (x, (x - hi) 2^11) with one power-of-two scale per block of 16 along `ch_dim`
    (md_split_f16f6: scale = 2^(floor(log2 amax) - 2) from the block's max over both sequences)."""
    hi = x.clamp(-65504, 65504).half()
    lo = (x - hi.float()) * 2048.0
    xs, ls = x.movedim(ch_dim, -1), lo.movedim(ch_dim, -1)
    shp = xs.shape
    xb, lb = xs.reshape(shp[:-1] + (shp[-1] // 16, 16)), ls.reshape(shp[:-1] + (shp[-1] // 16, 16))
    amax = torch.maximum(xb.abs().amax(-1, keepdim=True), lb.abs().amax(-1, keepdim=True)).float().contiguous()
    eb = ((amax.view(torch.int32) >> 23) & 0xff).clamp_min(20)
    scale = torch.exp2((eb - 129).float())
    q = (_q6_e2m3(xb / scale) * scale).reshape(shp).movedim(-1, ch_dim)
    ql = (_q6_e2m3(lb / scale) * scale).reshape(shp).movedim(-1, ch_dim)
    return hi, q, ql


def _conv_f16f6_reference(ref_in, w):
    """The f16f6 Winograd conv restated in torch (blocks of 16 input channels per position / per weight row and tap)."""
    B, _, S = ref_in.shape[:3]
    cout = w.shape[0]
    amax = float(w.abs().max())
    sw = 7 - int(np.floor(np.log2(1.5 * np.float32(amax))))
    T = _wino_T(F.pad(ref_in, (0, 0, 1, 1, 1, 1)))
    G = [g * (2.0 ** sw) for g in _wino_G(w)]
    m = []
    for f in range(4):
        th, tq, tl = _f16f6_images(T[f], 1)
        gh, gq, gl = _f16f6_images(G[f][..., None], 1)
        cross = F.conv3d(tq.float(), gl.float()) + F.conv3d(tl.float(), gq.float())
        m.append(F.conv3d(th.float(), gh.float()) + cross * 2.0 ** -11)
    y0, y1 = (m[0] + m[1]) + m[2], (m[1] - m[2]) - m[3]
    return torch.stack([y0, y1], -1).reshape(B, cout, S, S, S) * 2.0 ** -sw


@pytest.mark.parametrize("fmt", ["f8", "f6"])
@pytest.mark.parametrize("case", ["plain_128", "two_parts_gn_silu_res_stats", "ups_256rows", "k64_32cube"])
def test_conv3_wino_f8_vs_torch(ops, case, fmt):
    """md_conv3_wino_f8 against torch fp32 (budget: 3e-5, measured ~1.3e-5 = the arithmetic's own error), against the torch
    restatement of the SAME arithmetic (only the fp32 accumulation order differs: < 2e-6) and bit-identical between launches.
    Shapes: one / several chunk pairs (cin 32: a single body, 64: both unrolled bodies, 128: the loop), two row blocks, two
    parts with GroupNorm + SiLU + residual + statistics, the upsampled operand."""
    cfgs = {
        "plain_128": dict(cs=[128], cout=128, S=16, B=2, gn=False, silu=False, ups=False, res=False, stats=False),
        "two_parts_gn_silu_res_stats": dict(cs=[96, 32], cout=128, S=16, B=2, gn=True, silu=True, ups=False, res=True, stats=True),
        "ups_256rows": dict(cs=[32], cout=256, S=16, B=1, gn=False, silu=False, ups=True, res=False, stats=True),
        "k64_32cube": dict(cs=[32, 32], cout=128, S=32, B=1, gn=True, silu=True, ups=False, res=True, stats=False),
    }
    c = cfgs[case]
    cs, cout, S, B = c["cs"], c["cout"], c["S"], c["B"]
    cin = sum(cs)
    Sin = S // 2 if c["ups"] else S
    xs = [_rand((B, k, Sin, Sin, Sin), 150 + i) * (1.0 + i) + 0.3 * i for i, k in enumerate(cs)]
    x = torch.cat(xs, 1)
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 160), 0.5 * _rand((cin,), 161)
    w = _rand((cout, cin, 3, 3, 3), 162, 0.05)
    bias = _rand((B, cout), 163)
    res = _rand((B, cout, S, S, S), 164) if c["res"] else None
    parts = [(ops.ncdhw_to_f32b(t.cuda()), k) for t, k in zip(xs, cs)]
    ac, ref_in = None, x
    if c["gn"]:
        _, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, Sin ** 3, want_ac=True)
        ref_in = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        ref_in = F.silu(ref_in) if c["silu"] else ref_in
    if c["ups"]:
        ref_in = F.interpolate(ref_in, scale_factor=2, mode="nearest")
    ww = ops.WinoWeightF8(w.cuda(), "cuda", fmt)
    res_f = ops.ncdhw_to_f32b(res.cuda()) if res is not None else None
    outs = []
    for _ in range(2):
        t = ops.wino_prep(parts, ac, c["silu"], c["ups"], B, S, f8=fmt)
        stats = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda") if c["stats"] else None
        outs.append(ops.conv3_wino(ww, t, B, S, bias=bias.cuda(), bias_bstride=cout, residual=res_f,
                                   res_bstride=cout * S ** 3 if res is not None else 0, stats=stats).clone())
    assert torch.equal(outs[0], outs[1])
    y = ops.f32b_to_ncdhw(outs[0], (S, S, S)).cpu()
    extra = bias[:, :, None, None, None] + (res if res is not None else 0.0)
    ref = F.conv3d(ref_in.double(), w.double(), padding=1).float() + extra
    e = rel_l2(y, ref)
    # the restatement needs the kernel's own activated operand for a tight comparison only up to the activation's rounding
    # (v_exp / v_rcp in the operand pass): 1e-6-level differences in t, far below the arithmetic's 1e-5
    emu = (_conv_f16f6_reference if fmt == "f6" else _conv_f16f8_reference)(ref_in, w) + extra
    e_emu = rel_l2(y, emu)
    print(f"f16{fmt} wino conv ({case}): vs torch fp64 {e:.2e}, vs the torch restatement of the arithmetic {e_emu:.2e}")
    assert e < (4e-5 if fmt == "f6" else TOL_MFMA)      # f16f6: 1.7e-5 on the CPU model (tools/f16f8_numerics.py)
    # f16f6: an activation that differs in its last bits (v_exp / v_rcp) can move a block maximum across a binade or a value across
    # an e2m3 rounding boundary (steps of 1/16 of the block maximum): rarer, larger differences than with e4m3
    assert e_emu < ((8e-6 if c["gn"] else 2e-6) if fmt == "f6" else (4e-6 if c["gn"] else 2e-6))
    if c["stats"]:
        st, yd = stats.cpu(), y.double()
        assert rel_l2(st[..., 0], yd.sum(dim=(2, 3, 4))) < 1e-6
        assert rel_l2(st[..., 1], (yd * yd).sum(dim=(2, 3, 4))) < 1e-6


def test_conv3_wino_f8_against_bf16x3_on_full_tiles(ops):
    """256 -> 128 at 32^3, B = 2 (512 workgroups): the two arithmetics of the Winograd path on the same operands differ by
    their rounding only (1.5e-5), and the f16f8 path is bit-identical between launches."""
    B, S, cs, cout = 2, 32, [128, 128], 128
    cin = sum(cs)
    xs = [_rand((B, k, S, S, S), 170 + i) for i, k in enumerate(cs)]
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 172), 0.5 * _rand((cin,), 173)
    w = _rand((cout, cin, 3, 3, 3), 174, 0.03)
    bias = _rand((B, cout), 175)
    parts = [(ops.ncdhw_to_f32b(t.cuda()), k) for t, k in zip(xs, cs)]
    _, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, S ** 3, want_ac=True)
    w8, wb = ops.WinoWeightF8(w.cuda(), "cuda"), ops.WinoWeight(w.cuda(), "cuda")
    outs = []
    for _ in range(2):
        t = ops.wino_prep(parts, ac, True, False, B, S, f8=True)
        outs.append(ops.conv3_wino(w8, t, B, S, bias=bias.cuda(), bias_bstride=cout).clone())
    assert torch.equal(outs[0], outs[1])
    t = ops.wino_prep(parts, ac, True, False, B, S)
    yb = ops.conv3_wino(wb, t, B, S, bias=bias.cuda(), bias_bstride=cout)
    e = rel_l2(outs[0].cpu(), yb.cpu())
    print(f"f16f8 vs bf16x3 Winograd conv: {e:.2e}")
    assert e < 3e-5
    w6 = ops.WinoWeightF8(w.cuda(), "cuda", "f6")
    outs6 = []
    for _ in range(2):
        t = ops.wino_prep(parts, ac, True, False, B, S, f8="f6")
        outs6.append(ops.conv3_wino(w6, t, B, S, bias=bias.cuda(), bias_bstride=cout).clone())
    assert torch.equal(outs6[0], outs6[1])
    e6 = rel_l2(outs6[0].cpu(), yb.cpu())
    print(f"f16f6 vs bf16x3 Winograd conv: {e6:.2e}")
    assert e6 < 4e-5


# ------------------------------------------------------------------------------------------------------------------------------
# Round 5: the static per-channel equaliser under the f16f8 / f16f6 arithmetic (md_wino_equaliser, csrc/wino_eq.hip) and the
# adversarial operand gate VERDICT r04 asked for: what a TRAINED GroupNorm affine / weight tensor can look like and i.i.d. weights
# never do.  Everything is compared with torch float64 (the reference convolves in fp32: layers.py:118-124 behind :652-681).
# ------------------------------------------------------------------------------------------------------------------------------
def _equaliser_reference(gamma, beta, w, a2m=None):
    """Restatement of md_wino_equaliser: s_c = 2^round(log2(g_c / a_c) / 2), a_c = rms of silu(gamma_c z + beta_c) over z ~ N(0, 1)
    (64-point midpoint rule on [-6, 6]) -- or the MEASURED a2m (md_wino_equaliser_measured) --, g_c = rms of w[:, c]; the exponent
    clamped to +-14 and to the fp16 headroom floor(14 - log2(8 |gamma_c| + |beta_c|)).  Returns (s, the un-rounded exponents)."""
    g2 = w.double().pow(2).mean(dim=(0, 2, 3, 4))
    if a2m is None:
        z = -6.0 + 12.0 * (torch.arange(64, dtype=torch.float64) + 0.5) / 64.0
        pdf = torch.exp(-0.5 * z * z)
        y = gamma.double()[:, None] * z[None] + beta.double()[:, None]
        a2 = ((y * torch.sigmoid(y)) ** 2 * pdf[None]).sum(1) / pdf.sum()
    else:
        a2 = a2m.double()
    ex = 0.25 * (torch.log2(g2) - torch.log2(a2))
    hi = torch.full_like(ex, 14.0)
    if gamma is not None:
        top = 8.0 * gamma.double().abs() + beta.double().abs()
        hi = torch.minimum(hi, torch.floor(14.0 - torch.log2(top)))
    e = torch.minimum(torch.round(ex).clamp(-14, 14), hi.clamp(min=-14))
    if a2m is not None:      # the measured form's common level shift: the largest equalised channel rms at 2^0
        u = -torch.round(torch.log2((torch.exp2(e) * a2.sqrt()).max()))
        e = torch.minimum(e + u, hi)
    return torch.exp2(e).float(), ex


def _student_t(shape, seed, df=3.0):
    g = torch.Generator().manual_seed(seed)
    z = torch.randn(shape, generator=g)
    chi = torch.randn((int(df),) + tuple(shape), generator=g).pow(2).sum(0) / df
    return z / chi.sqrt() / float(np.sqrt(df / (df - 2.0)))


def test_wino_equaliser_vs_restatement(ops):
    cin, cout = 256, 128
    s = torch.exp2((torch.rand(cin, generator=torch.Generator().manual_seed(7)) * 2 - 1) * 6)
    gamma, beta = (1.0 + 0.2 * _rand((cin,), 200)) * s, 0.5 * _rand((cin,), 201) * s
    w = _rand((cout, cin, 3, 3, 3), 202, 0.05) / s.view(1, -1, 1, 1, 1)
    eq = ops.wino_equaliser(gamma.cuda(), beta.cuda(), w.cuda()).cpu()
    want, ex = _equaliser_reference(gamma, beta, w)
    lg = torch.log2(eq)
    assert torch.equal(lg, torch.round(lg)) and float(lg.abs().max()) <= 14          # exact powers of two
    tie = (ex - torch.floor(ex) - 0.5).abs() < 1e-3                                    # fp32 vs fp64 may round a tie the other way
    assert torch.equal(eq[~tie], want[~tie]) and int(tie.sum()) < 4
    # what it is for: the equalised weights' (and activations') per-channel rms are ~ sqrt(a g) 2^+-0.5 -- within one e2m3 block's
    # useful range of each other -- where the un-equalised ones span 2^12
    g_raw = w.pow(2).mean(dim=(0, 2, 3, 4)).sqrt()
    g_eq = (w / eq.view(1, -1, 1, 1, 1)).pow(2).mean(dim=(0, 2, 3, 4)).sqrt()
    assert float(g_raw.max() / g_raw.min()) > 1000.0 and float(g_eq.max() / g_eq.min()) < 16.0
    # degenerate channels: zero weights / zero affine -> 1
    w0 = w.clone(); w0[:, 5] = 0.0
    g0 = gamma.clone(); g0[9] = 0.0; b0 = beta.clone(); b0[9] = 0.0
    eq0 = ops.wino_equaliser(g0.cuda(), b0.cuda(), w0.cuda()).cpu()
    assert float(eq0[5]) == 1.0 and float(eq0[9]) == 1.0 and bool(torch.isfinite(eq0).all())


@pytest.mark.parametrize("fmt", ["f8", "f6"])
def test_equalised_operand_and_weights_are_exact_rescalings(ops, fmt):
    """eq in the operand pass == the same pass on the pre-multiplied tensor; eq in the weight packing == packing w / eq: bit for bit
    (powers of two), header included.  So the restatements of the plain formats cover the equalised ones."""
    B, S, cin, cout = 2, 16, 64, 128
    eq = torch.exp2(torch.randint(-6, 7, (cin,), generator=torch.Generator().manual_seed(3)).float())
    x = _rand((B, cin, S, S, S), 210)
    w = _rand((cout, cin, 3, 3, 3), 211, 0.05)
    t0 = ops.wino_prep([(ops.ncdhw_to_f32b(x.cuda()), cin)], None, False, False, B, S, f8=fmt, eq=eq.cuda()).clone()
    t1 = ops.wino_prep([(ops.ncdhw_to_f32b((x * eq.view(1, -1, 1, 1, 1)).cuda()), cin)], None, False, False, B, S, f8=fmt).clone()
    assert torch.equal(t0.view(torch.int16), t1.view(torch.int16))
    w0 = ops.WinoWeightF8(w.cuda(), "cuda", fmt, eq=eq.cuda())
    w1 = ops.WinoWeightF8((w / eq.view(1, -1, 1, 1, 1)).cuda(), "cuda", fmt)
    assert torch.equal(w0.data.view(torch.int16), w1.data.view(torch.int16))
    # pairing check: an operand equalised for another layer / in another format is refused, not silently multiplied
    t_ok = ops.wino_prep([(ops.ncdhw_to_f32b(x.cuda()), cin)], None, False, False, B, S, f8=fmt, eq=eq.cuda())
    with pytest.raises(Exception):
        ops.conv3_wino(w1, t_ok, B, S)                       # w1 was packed without the equaliser
    other = "f8" if fmt == "f6" else "f6"
    t_other = ops.wino_prep([(ops.ncdhw_to_f32b(x.cuda()), cin)], None, False, False, B, S, f8=other)
    with pytest.raises(Exception):
        ops.conv3_wino(w1, t_other, B, S)


ADVERSARIAL = ["gamma_span3", "gamma_span6", "outlier100", "student_t", "gamma_span3_student_t_two_parts"]
INSIDE_GROUP = [128, 256, 512]


@pytest.mark.parametrize("fmt", ["f8", "f6"])
@pytest.mark.parametrize("case", ADVERSARIAL)
def test_conv3_wino_f8_adversarial_operands(ops, case, fmt, monkeypatch):
    """GroupNorm -> SiLU -> conv through the PRODUCT dispatch (layers.run_conv3 under hip_ops.precision_scope("f16" + fmt): format
    choice, the layer's cached equaliser, the pairing of operand and weights) on operands i.i.d. test weights never produce:
      gamma_span3 / _span6  per-channel GroupNorm scale 2^U(-3,3) / 2^U(-6,6) on (gamma, beta), compensated in the weights: every
                            channel matters equally, their activations differ by up to 2^6 / 2^12 inside a K block
      outlier100            one channel of every 16-block 100 x larger (weights / 100)
      student_t             heavy-tailed weights (Student-t, 3 degrees of freedom)
      ..._two_parts         both, on a concatenated two-part input
    vs torch float64: the budgets of the friendly-weight tests (3e-5 / 4e-5) must hold; without the equaliser f16f6 is at
    1e-4 .. 3e-4 here (printed; asserted for the span cases)."""
    from meshdiffusion_amd.lib.diffusion.models import layers
    monkeypatch.setattr(ops, "WINO_MIN_WGS", 1)
    B, S, cout = 2, 16, 128
    cs = [96, 32] if case.endswith("two_parts") else [128]
    cin = sum(cs)
    g = torch.Generator().manual_seed(300 + ADVERSARIAL.index(case))
    s = torch.ones(cin)
    if "span" in case:
        s = torch.exp2((torch.rand(cin, generator=g) * 2 - 1) * (6.0 if "span6" in case else 3.0))
    if case == "outlier100":
        s[3::16] = 100.0
    wbase = _student_t((cout, cin, 3, 3, 3), 310, 3.0) * 0.05 if "student_t" in case else _rand((cout, cin, 3, 3, 3), 310, 0.05)
    w = wbase / s.view(1, -1, 1, 1, 1)
    gamma, beta = (1.0 + 0.2 * _rand((cin,), 311)) * s, 0.5 * _rand((cin,), 312) * s
    xs = [_rand((B, k, S, S, S), 320 + i) * (1.5 + i) + 0.3 for i, k in enumerate(cs)]
    x = torch.cat(xs, 1)
    bias = _rand((B, cout), 313)
    ref_in = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), eps=1e-6))
    ref = F.conv3d(ref_in, w.double(), padding=1) + bias.double()[:, :, None, None, None]

    class Pair(layers.HipLayer):
        def __init__(self):
            super().__init__()
            self.gn = torch.nn.GroupNorm(32, cin, eps=1e-6)
            self.conv = torch.nn.Conv3d(cin, cout, 3, padding=1)

    pair = Pair()
    with torch.no_grad():
        pair.gn.weight.copy_(gamma); pair.gn.bias.copy_(beta); pair.conv.weight.copy_(w)
    pair = pair.cuda()
    parts = [(ops.ncdhw_to_f32b(t.cuda()), k) for t, k in zip(xs, cs)]
    _, ac = ops.gn_params(parts, pair.gn.weight, pair.gn.bias, B, S ** 3, want_ac=True)
    pw = layers.conv3_packed(pair, "w", pair.conv, ops.conv_cfg_for(S))

    def run(mode, gn):
        ops.PROFILE = []
        try:
            with ops.precision_scope(mode):
                out = layers.run_conv3(pw, None, B, S, bias=bias.cuda(), bias_bstride=cout,
                                       b_f32=dict(parts=parts, ac=ac, silu=True), wino=layers.conv3_wino_packed(pair, "w", pair.conv, gn=gn))
            tags = [r[5] for r in ops.PROFILE if r[0] == "wino"]
        finally:
            ops.PROFILE = None
        return ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), tags

    y, tags = run("f16" + fmt, pair.gn)
    assert len(tags) == 1 and tags[0].endswith("/" + fmt), tags            # really the reduced-precision kernel
    y_raw, _ = run("f16" + fmt, None)                                       # the round-4 behaviour: no equaliser
    y_b3, tags_b = run("bf16x3", pair.gn)
    assert "/f" not in tags_b[0]
    e, e_raw, e_b3 = rel_l2(y, ref), rel_l2(y_raw, ref), rel_l2(y_b3, ref)
    print(f"f16{fmt} adversarial ({case}): vs torch fp64 with equaliser {e:.2e}, without {e_raw:.2e}; bf16x3 {e_b3:.2e}")
    assert e < (4e-5 if fmt == "f6" else TOL_MFMA) and e_b3 < TOL_MFMA
    if fmt == "f6" and "span" in case:
        assert e_raw > 2.5 * e          # the case is adversarial for the un-equalised format (else it tests nothing)
    # the restatement of the arithmetic on the equalised operands (exact rescalings, see the test above)
    eq = pair._md_cache[f"w/wino_eq"][1].cpu().view(1, -1, 1, 1, 1)
    emu = (_conv_f16f6_reference if fmt == "f6" else _conv_f16f8_reference)((ref_in.float() * eq), w / eq) + bias[:, :, None, None, None]
    assert rel_l2(y, emu) < (8e-6 if fmt == "f6" else 4e-6)


def _pair_layer(layers, cin, cout, gamma, beta, w):
    class Pair(layers.HipLayer):
        def __init__(self):
            super().__init__()
            self.gn = torch.nn.GroupNorm(32, cin, eps=1e-6)
            self.conv = torch.nn.Conv3d(cin, cout, 3, padding=1)

    pair = Pair()
    with torch.no_grad():
        pair.gn.weight.copy_(gamma); pair.gn.bias.copy_(beta); pair.conv.weight.copy_(w)
    return pair.cuda()


@pytest.mark.parametrize("fmt", ["f8", "f6"])
@pytest.mark.parametrize("cin", INSIDE_GROUP)
def test_conv3_wino_f8_inside_group_spread_static_vs_measured_equaliser(ops, cin, fmt, monkeypatch):
    """The static equaliser's blind spot: a per-channel scale 2^U(-3,3) on the INPUT of the GroupNorm.
    GroupNorm normalises groups of cin / 32 channels, so each channel leaves it with its own scale relative to its group's rms --
    which (gamma, beta) do not show.  Compensated in the conv weights (every channel matters equally), vs torch float64:
      static equaliser    (md_wino_equaliser: unit variance per channel assumed)  -- printed; f16f6 reaches 3..4.5e-5 here
      measured equaliser  (the layer calibrated on this operand: hip_ops.CALIBRATE -> md_wino_operand_ms -> md_wino_equaliser_measured)
                          -- asserted: the budgets of the friendly-weight tests hold again (f16f6 <= 4e-5, f16f8 <= TOL_MFMA)."""
    from meshdiffusion_amd.lib.diffusion.models import layers
    monkeypatch.setattr(ops, "WINO_MIN_WGS", 1)
    B, S, cout = 2, 16, 128
    g = torch.Generator().manual_seed(500 + cin)
    s = torch.exp2((torch.rand(cin, generator=g) * 2 - 1) * 3.0)
    x = (_rand((B, cin, S, S, S), 520) * 1.5 + 0.3) * s.view(1, -1, 1, 1, 1)
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 511), 0.5 * _rand((cin,), 512)
    # what the channel looks like behind the GroupNorm: its scale over its group's rms
    grp = (s.view(32, -1) ** 2).mean(1, keepdim=True).sqrt().expand(-1, cin // 32).reshape(-1)
    rel = s / grp
    w = _rand((cout, cin, 3, 3, 3), 510, 0.05) / rel.view(1, -1, 1, 1, 1)
    bias = _rand((B, cout), 513)
    ref_in = F.silu(F.group_norm(x.double(), 32, gamma.double(), beta.double(), eps=1e-6))
    ref = F.conv3d(ref_in, w.double(), padding=1) + bias.double()[:, :, None, None, None]
    pair = _pair_layer(layers, cin, cout, gamma, beta, w)
    parts = [(ops.ncdhw_to_f32b(x.cuda()), cin)]
    _, ac = ops.gn_params(parts, pair.gn.weight, pair.gn.bias, B, S ** 3, want_ac=True)
    pw = layers.conv3_packed(pair, "w", pair.conv, ops.conv_cfg_for(S))

    def run(mode):
        ops.PROFILE = []
        try:
            with ops.precision_scope(mode):
                out = layers.run_conv3(pw, None, B, S, bias=bias.cuda(), bias_bstride=cout, b_f32=dict(parts=parts, ac=ac, silu=True),
                                       wino=layers.conv3_wino_packed(pair, "w", pair.conv, gn=pair.gn))
            tags = [r[5] for r in ops.PROFILE if r[0] == "wino"]
        finally:
            ops.PROFILE = None
        return ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), tags

    y_static, tags = run("f16" + fmt)
    assert len(tags) == 1 and tags[0].endswith("/" + fmt), tags
    ops.CALIBRATE = {}
    try:
        run("f16" + fmt)
        cal = ops.CALIBRATE
    finally:
        ops.CALIBRATE = None
    assert len(cal) == 1
    (owner, site, tot, n), = cal.values()
    assert owner is pair and site == "w" and n == 1
    ms = tot / n
    assert rel_l2(ms.cpu(), ref_in.float().pow(2).mean(dim=(0, 2, 3, 4))) < 1e-3          # md_wino_operand_ms vs torch
    pair._md_act_ms = {"w": ms.contiguous()}
    y_meas, tags = run("f16" + fmt)
    assert len(tags) == 1 and tags[0].endswith("/" + fmt), tags
    eq = pair._md_cache["w/wino_eqm"][1].cpu()
    want, ex = _equaliser_reference(gamma, beta, w, a2m=ms.cpu())
    tie = (ex - torch.floor(ex) - 0.5).abs() < 1e-3
    assert torch.equal(eq[~tie], want[~tie]) and int(tie.sum()) < 6
    y_b3, _ = run("bf16x3")
    e_s, e_m, e_b3 = rel_l2(y_static, ref), rel_l2(y_meas, ref), rel_l2(y_b3, ref)
    print(f"f16{fmt} inside-group spread 2^+-3, cin {cin} ({cin // 32} channels per group): vs torch fp64 static equaliser {e_s:.2e}, "
          f"measured {e_m:.2e}; bf16x3 {e_b3:.2e}")
    assert e_m < (4e-5 if fmt == "f6" else TOL_MFMA) and e_b3 < TOL_MFMA
    assert e_m < e_s * 1.05          # the measurement never makes it worse
    emu = (_conv_f16f6_reference if fmt == "f6" else _conv_f16f8_reference)((ref_in.float() * eq.view(1, -1, 1, 1, 1)), w / eq.view(1, -1, 1, 1, 1)) \
        + bias[:, :, None, None, None]
    assert rel_l2(y_meas, emu) < (8e-6 if fmt == "f6" else 4e-6)


def test_wino_equaliser_fp16_headroom_bound(ops):
    """a near-dead channel (beta strongly negative: silu ~ 0 almost everywhere) must not get 2^14 -- an outlier voxel
    would leave the fp16 hi plane as inf.  The exponent is bounded by floor(14 - log2(8 |gamma| + |beta|))."""
    cin, cout = 64, 128
    gamma, beta = torch.ones(cin), torch.zeros(cin)
    beta[7] = -40.0                                     # silu(z - 40) ~ 1e-16: the static a_c is ~ 0
    gamma[9], beta[9] = 30.0, -200.0
    w = _rand((cout, cin, 3, 3, 3), 530, 0.05)
    eq = ops.wino_equaliser(gamma.cuda(), beta.cuda(), w.cuda()).cpu()
    want, _ = _equaliser_reference(gamma, beta, w)
    assert torch.equal(eq, want)
    top = 8.0 * gamma.abs() + beta.abs()
    assert bool((eq * top * 2.0 < 2.0 ** 15).all()) and float(eq[7]) <= 2.0 ** 8 and float(eq[9]) <= 2.0 ** 5


@pytest.mark.parametrize("fmt", ["f8", "f6"])
def test_unnormalised_tiny_operand_bf16x3_until_calibrated(ops, fmt, monkeypatch):
    """The Upsample conv reads the raw residual stream (no GroupNorm in front: nothing static to equalise, and a 1e-5-magnitude tensor
    is subnormal in fp16 / below e4m3's range: 1.5e-3 in f16f8).  Uncalibrated, the dispatch keeps such convs in bf16x3 (full fp32
    exponent range); once the layer holds the MEASURED mean squares of its operand (DDPMUNet3D.calibrate / hip_ops.CALIBRATE) its
    measured equaliser puts the operand at unit scale and the conv runs in the reduced-precision format: 1e-5-magnitude input with a
    2^+-2 channel spread, nearest-x2 upsampled, vs torch float64 -- <= 2e-5 (f16f8 TOL_MFMA)."""
    from meshdiffusion_amd.lib.diffusion.models import layers
    monkeypatch.setattr(ops, "WINO_MIN_WGS", 1)
    B, S, cin, cout = 1, 16, 128, 128
    spread = torch.exp2((torch.rand(cin, generator=torch.Generator().manual_seed(9)) * 2 - 1) * 2.0)
    x = _rand((B, cin, S // 2, S // 2, S // 2), 330) * 1e-5 * spread.view(1, -1, 1, 1, 1)
    up = layers.Upsample(cin, with_conv=True)
    with torch.no_grad():
        up.Conv_0.weight.copy_(_rand((cout, cin, 3, 3, 3), 331, 0.05) / spread.view(1, -1, 1, 1, 1)); up.Conv_0.bias.zero_()
    up = up.cuda().eval()
    ref = F.conv3d(F.interpolate(x.double(), scale_factor=2, mode="nearest"), up.Conv_0.weight.detach().double().cpu(), padding=1)

    def run():
        ops.PROFILE = []
        try:
            with ops.precision_scope("f16" + fmt), torch.no_grad():
                y = up(x.cuda()).cpu()
            return y, [r[5] for r in ops.PROFILE if r[0] in ("wino", "wino_prep")]
        finally:
            ops.PROFILE = None

    y, tags = run()
    assert tags and all("/f" not in t for t in tags), tags
    e = rel_l2(y, ref)
    print(f"raw 1e-5 operand under a f16{fmt} scope, uncalibrated: {e:.2e} (bf16x3 kernels: {tags})")
    assert e < TOL_MFMA
    ops.CALIBRATE = {}
    try:
        run()
        cal = ops.CALIBRATE
    finally:
        ops.CALIBRATE = None
    assert len(cal) == 1
    for owner, site, tot, n in cal.values():
        owner._md_act_ms = {site: (tot / n).contiguous()}
    y, tags = run()
    assert tags and all(t.endswith("/" + fmt) for t in tags if t.startswith("128->")), tags
    assert any("/" + fmt in t for t in tags)
    e = rel_l2(y, ref)
    print(f"raw 1e-5 operand under a f16{fmt} scope, calibrated: {e:.2e} ({tags})")
    assert e < (2e-5 if fmt == "f6" else TOL_MFMA)


def test_precision_audit_record_and_per_layer_override(ops, monkeypatch):
    """hip_ops.AUDIT (tools/audit_precision.py): a reduced-precision conv launch is repeated in bf16x3 on the same operands and the pair's
    relative difference recorded -- the per-layer check to run on a real checkpoint; `layer.md_bf16x3_sites` takes a conv off the
    reduced-precision path for good."""
    from meshdiffusion_amd.lib.diffusion.models import layers
    monkeypatch.setattr(ops, "WINO_MIN_WGS", 1)
    B, S, cin, cout = 1, 16, 128, 128

    class Pair(layers.HipLayer):
        def __init__(self):
            super().__init__()
            self.gn = torch.nn.GroupNorm(32, cin, eps=1e-6)
            self.conv = torch.nn.Conv3d(cin, cout, 3, padding=1)

    pair = Pair().cuda()
    x = ops.ncdhw_to_f32b(_rand((B, cin, S, S, S), 400).cuda())
    _, ac = ops.gn_params([(x, cin)], pair.gn.weight, pair.gn.bias, B, S ** 3, want_ac=True)
    pw = layers.conv3_packed(pair, "w", pair.conv, ops.conv_cfg_for(S))

    def run():
        ops.PROFILE = []
        try:
            with ops.precision_scope("f16f6"):
                out = layers.run_conv3(pw, None, B, S, bias=pair.conv.bias, b_f32=dict(parts=[(x, cin)], ac=ac, silu=True),
                                       wino=layers.conv3_wino_packed(pair, "w", pair.conv, gn=pair.gn))
            return out.clone(), [r[5] for r in ops.PROFILE if r[0] == "wino"]
        finally:
            ops.PROFILE = None

    y6, tags = run()
    assert tags == [t for t in tags if t.endswith("/f6")] and len(tags) == 1
    ops.AUDIT = []
    try:
        ya, _ = run()
        recs = ops.AUDIT
    finally:
        ops.AUDIT = None
    assert torch.equal(ya, y6)                                      # the audit does not disturb the product launch
    assert len(recs) == 1 and recs[0]["owner"] is pair and recs[0]["site"] == "w" and recs[0]["fmt"] == "f6"
    assert 1e-6 < recs[0]["rel_l2"] < 4e-5
    pair.md_bf16x3_sites = ("w",)
    y3, tags3 = run()
    assert len(tags3) == 1 and "/f" not in tags3[0]
    assert abs(rel_l2(y6.cpu(), y3.cpu()) - recs[0]["rel_l2"]) < 1e-6      # the recorded difference IS f16f6 vs bf16x3 on this layer
