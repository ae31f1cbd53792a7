"""Per-kernel parity on the GPU: each HIP entry point (through the C ABI) against plain PyTorch
fp32 on the CPU (the oracle's building blocks).  Tolerances: bf16x3 GEMM/conv 3e-5 rel-L2
(operand split keeps ~16 mantissa bits; SURVEY 7.1 measured 1.8e-5 for a whole U-Net), fp32
streaming kernels 2e-6, bit-exact where stated."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

TOL_MFMA = 3e-5
TOL_F32 = 2e-6


@pytest.fixture(scope="module")
def ops(hip_lib):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from meshdiffusion_amd import hip_ops
    return hip_ops


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _to_s16(ops, x):
    parts = [(ops.ncdhw_to_f32b(x.cuda()), x.shape[1])]
    B, P = x.shape[0], int(np.prod(x.shape[2:]))
    return ops.gn_apply(parts, None, B, P, norm=False, silu=False), B, P


def test_layout_roundtrip_and_split(ops):
    x = _rand((2, 16, 4, 4, 8), 0)
    f = ops.ncdhw_to_f32b(x.cuda())
    assert torch.equal(ops.f32b_to_ncdhw(f, x.shape[2:]).cpu(), x)
    s, B, P = _to_s16(ops, x)
    back = ops.s16b_to_ncdhw(s, x.shape[2:]).cpu()
    assert rel_l2(back, x) < 2e-5          # hi+lo keeps >= 16 mantissa bits
    hi = s[:, :, 0].float().cpu()           # plane 0 must be exactly bf16(x)
    x_blk = x.reshape(2, 2, 8, -1).permute(0, 1, 3, 2)
    assert torch.equal(hi, x_blk.to(torch.bfloat16).float())
    s2 = ops.ncdhw_to_s16b(x.cuda(), 16)
    assert torch.equal(s2.view(torch.int16).cpu(), s.view(torch.int16).cpu())


def test_gemm_asymmetric_identity(ops):
    """Transpose-detecting check of the MFMA fragment/accumulator mapping: W = shifted identity."""
    Cc, P = 128, 256
    x = torch.arange(Cc * P, dtype=torch.float32).reshape(1, Cc, 1, 1, P) / 7.0
    W = torch.zeros(Cc, Cc)
    for i in range(Cc):
        W[i, (3 * i + 5) % Cc] = 1.0 + i / 64.0      # y[o] = sum_i x[i] W[i,o]: asymmetric permutation
    s, B, Pn = _to_s16(ops, x)
    pw = ops.PackedWeight(W.cuda(), "nin", ops.CFG_G1_128, "cuda")
    out = ops.f32b_empty(1, Cc, P, "cuda")
    ops.gemm_conv(cfg=ops.CFG_G1_128, a=pw.data, b=s, out=out, batch=1, rows=Cc, rows_alloc=Cc, kdim=Cc,
                  dims=(1, 1, P))
    y = ops.f32b_to_ncdhw(out, (1, 1, P)).cpu()
    ref = torch.einsum("bcdhw,co->bodhw", x, W)
    assert rel_l2(y, ref) < TOL_MFMA


# the A/B configuration id CFG_FAST_EC exists in MD_BUILD_ABLATIONS=1 libraries only: it is a case of this test there and nowhere else
_CONV3_CFGS = ["CFG_C3_128", "CFG_C3_128_FAST"] + (["CFG_FAST_EC"] if os.environ.get("MD_BUILD_ABLATIONS") == "1" else [])


@pytest.mark.parametrize("cfg_name", _CONV3_CFGS)
@pytest.mark.parametrize("cin,cout,S,B", [(32, 128, 8, 2), (64, 256, 16, 1), (160, 128, 8, 1)])
def test_conv3_main(ops, cin, cout, S, B, cfg_name):
    cfg = getattr(ops, cfg_name)
    if cfg_name == "CFG_FAST_EC" and cfg not in ops._lib.CFG_NT_KC:
        pytest.skip("timing-only / A/B configuration ids exist in MD_BUILD_ABLATIONS=1 libraries only")
    x = _rand((B, cin, S, S, S), 1)
    w = _rand((cout, cin, 3, 3, 3), 2, 0.05)
    bias = _rand((B, cout), 3)
    res = _rand((B, cout, S, S, S), 4)
    s, _, P = _to_s16(ops, x)
    pw = ops.PackedWeight(w.cuda(), "conv", cfg, "cuda")
    out = ops.f32b_empty(B, cout, P, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=s, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S, S, S), bias=bias.cuda(), bias_bstride=cout, residual=ops.ncdhw_to_f32b(res.cuda()),
                  res_bstride=cout * P)
    y = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()
    ref = F.conv3d(x, w, padding=1) + bias[:, :, None, None, None] + res
    assert rel_l2(y, ref) < TOL_MFMA


@pytest.mark.parametrize("cfg_name,S,ks", [("CFG_C3_128_FAST", 8, 4), ("CFG_C3_LOW", 4, 4), ("CFG_C3_128", 8, 2), ("CFG_C3_LOW", 4, 3)])
def test_conv3_split_k(ops, cfg_name, S, ks):
    """split-K (grid.z partial sums + ordered reduce) == single-pass result up to fp32 summation order."""
    cfg = getattr(ops, cfg_name)
    B, cin, cout = 2, 128, 128
    x = _rand((B, cin, S, S, S), 40); w = _rand((cout, cin, 3, 3, 3), 41, 0.05)
    bias = _rand((B, cout), 42); res = _rand((B, cout, S, S, S), 43)
    s, _, P = _to_s16(ops, x)
    pw = ops.PackedWeight(w.cuda(), "conv", cfg, "cuda")
    outs = []
    for k in (1, ks):
        out = ops.f32b_empty(B, cout, P, "cuda")
        ops.gemm_conv(cfg=cfg, a=pw.data, b=s, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin, dims=(S, S, S),
                      bias=bias.cuda(), bias_bstride=cout, residual=ops.ncdhw_to_f32b(res.cuda()), res_bstride=cout * P,
                      ksplit=k)
        outs.append(ops.f32b_to_ncdhw(out, (S, S, S)).cpu())
    ref = F.conv3d(x, w, padding=1) + bias[:, :, None, None, None] + res
    assert rel_l2(outs[1], ref) < TOL_MFMA and rel_l2(outs[1], outs[0]) < 1e-6


@pytest.mark.parametrize("cin,cout,S,B,ups", [(64, 128, 8, 2, 0), (96, 256, 16, 1, 0), (64, 64, 8, 1, 1)])
def test_conv3_fp16x2_mode(ops, cin, cout, S, B, ups):
    """MD_PREC_FP16X2: weights split fp16 (exact to ~2^-21), activations ONE fp16.  Checked two ways:
    against the conv of the fp16-rounded activations (isolates the kernel: 3e-5) and against the exact
    conv (shows the mode's own error, ~2e-4 for one layer)."""
    Sin = S // 2 if ups else S
    x = _rand((B, cin, Sin, Sin, Sin), 60); w = _rand((cout, cin, 3, 3, 3), 61, 0.05); bias = _rand((B, cout), 62)
    parts = [(ops.ncdhw_to_f32b(x.cuda()), cin)]
    act = ops.gn_apply(parts, None, B, Sin ** 3, norm=False, silu=False, fp16=True)
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_128_FAST, "cuda", ops.PREC_FP16X2)
    out = ops.f32b_empty(B, cout, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_128_FAST, a=pw.data, b=act, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S, S, S), bias=bias.cuda(), bias_bstride=cout, ups=ups, prec=ops.PREC_FP16X2)
    y = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()
    up = (lambda t: F.interpolate(t, scale_factor=2, mode="nearest")) if ups else (lambda t: t)
    ref_q = F.conv3d(up(x.half().float()), w, padding=1) + bias[:, :, None, None, None]
    ref = F.conv3d(up(x), w, padding=1) + bias[:, :, None, None, None]
    assert rel_l2(y, ref_q) < TOL_MFMA
    assert rel_l2(y, ref) < 5e-4


def test_conv3_low_tile(ops):
    x = _rand((2, 64, 4, 4, 4), 5); w = _rand((128, 64, 3, 3, 3), 6, 0.05)
    s, B, P = _to_s16(ops, x)
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_LOW, "cuda")
    out = ops.f32b_empty(2, 128, P, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_LOW, a=pw.data, b=s, out=out, batch=2, rows=128, rows_alloc=128, kdim=64, dims=(4, 4, 4))
    assert rel_l2(ops.f32b_to_ncdhw(out, (4, 4, 4)).cpu(), F.conv3d(x, w, padding=1)) < TOL_MFMA


@pytest.mark.parametrize("S_in", [8, 16])
def test_conv3_stride2(ops, S_in):
    x = _rand((2, 32, S_in, S_in, S_in), 7); w = _rand((64, 32, 3, 3, 3), 8, 0.05); b = _rand((64,), 9)
    s, B, _ = _to_s16(ops, x)
    So = S_in // 2
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_S2, "cuda")
    out = ops.f32b_empty(2, 64, So ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_S2, a=pw.data, b=s, out=out, batch=2, rows=64, rows_alloc=64, kdim=32,
                  dims=(So, So, So), bias=b.cuda())
    ref = F.conv3d(F.pad(x, (0, 1, 0, 1, 0, 1)), w, b, stride=2)
    assert rel_l2(ops.f32b_to_ncdhw(out, (So, So, So)).cpu(), ref) < TOL_MFMA


@pytest.mark.parametrize("B,cin,cout,S_out", [(2, 32, 64, 8), (1, 64, 128, 16), (2, 128, 136, 8), (3, 32, 256, 16)])
def test_conv3_s2_fp32_operand_kernel(ops, B, cin, cout, S_out):
    """md_conv3_s2 (csrc/conv3_s2.hip: Downsample's pad (0, 1) + stride-2 conv on the raw fp32 tensor, (chunk, kd) slabs
    with the x parity de-interleaved in LDS) against torch fp32, against the generic tile it replaces (md_gn_apply split +
    MD_CFG_C3_S2), its GroupNorm sums against float64 sums of what it wrote, bit-identical between launches.
    Cases: one tile / several tiles per axis (far-face padding in every tile position), 2-8 chunks, a partial row tile,
    two row tiles."""
    S_in = 2 * S_out
    x = _rand((B, cin, S_in, S_in, S_in), 70 + cin); w = _rand((cout, cin, 3, 3, 3), 71, 0.05); b = _rand((cout,), 72)
    xf = ops.ncdhw_to_f32b(x.cuda())
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_S2_PACK, "cuda")
    assert ops.conv3_s2_ok(cout, cin, S_out)
    stats = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
    out = ops.conv3_s2(pw, xf, B, S_out, bias=b.cuda(), stats=stats)
    got = ops.f32b_to_ncdhw(out, (S_out,) * 3).cpu()
    ref = F.conv3d(F.pad(x, (0, 1, 0, 1, 0, 1)), w, b, stride=2)
    e = rel_l2(got, ref)
    print(f"conv3_s2 {cin}->{cout} @ {S_in}^3 -> {S_out}^3: vs torch fp32 {e:.2e}")
    assert e < TOL_MFMA
    g64 = got.double()
    ref_sums = torch.stack([g64.sum(dim=(2, 3, 4)), (g64 * g64).sum(dim=(2, 3, 4))], dim=-1)
    assert torch.allclose(stats.cpu(), ref_sums, rtol=1e-5, atol=1e-3)
    # the path it replaces
    s16, _, _ = _to_s16(ops, x)
    pw_old = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_S2, "cuda")
    old = ops.f32b_empty(B, cout, S_out ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_S2, a=pw_old.data, b=s16, out=old, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S_out,) * 3, bias=b.cuda())
    assert rel_l2(got, ops.f32b_to_ncdhw(old, (S_out,) * 3).cpu()) < 1e-5
    out2 = ops.conv3_s2(pw, xf, B, S_out, bias=b.cuda())
    assert torch.equal(out, out2)


@pytest.mark.parametrize("S_in,cfg_name", [(4, "CFG_C3_128"), (8, "CFG_C3_128"), (4, "CFG_C3_128_FAST"), (8, "CFG_C3_128_FAST")])
def test_conv3_upsample_fold(ops, S_in, cfg_name):
    cfg = getattr(ops, cfg_name)
    x = _rand((1, 64, S_in, S_in, S_in), 10); w = _rand((64, 64, 3, 3, 3), 11, 0.05)
    s, B, _ = _to_s16(ops, x)
    So = 2 * S_in
    pw = ops.PackedWeight(w.cuda(), "conv", cfg, "cuda")
    out = ops.f32b_empty(1, 64, So ** 3, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=s, out=out, batch=1, rows=64, rows_alloc=64, kdim=64, dims=(So, So, So), ups=1)
    ref = F.conv3d(F.interpolate(x, scale_factor=2, mode="nearest"), w, padding=1)
    assert rel_l2(ops.f32b_to_ncdhw(out, (So, So, So)).cpu(), ref) < TOL_MFMA


def test_conv3_stem_and_head(ops):
    S = 8
    x = _rand((2, 4, S, S, S), 12); w = _rand((128, 4, 3, 3, 3), 13, 0.1)
    x16 = ops.ncdhw_to_s16b(x.cuda(), 16)
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_128_K16, "cuda")
    out = ops.f32b_empty(2, 128, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_128_K16, a=pw.data, b=x16, out=out, batch=2, rows=128, rows_alloc=128,
                  kdim=16, dims=(S, S, S))
    assert rel_l2(ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), F.conv3d(x, w, padding=1)) < TOL_MFMA
    # head: Cout=4 written straight to NCDHW
    xh = _rand((2, 64, S, S, S), 14); wh = _rand((4, 64, 3, 3, 3), 15, 0.05); bh = _rand((4,), 16)
    s, _, _ = _to_s16(ops, xh)
    pwh = ops.PackedWeight(wh.cuda(), "conv", ops.CFG_C3_32, "cuda")
    o = torch.empty((2, 4, S, S, S), device="cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_32, a=pwh.data, b=s, out=o, batch=2, rows=4, rows_alloc=8, kdim=64, dims=(S, S, S),
                  bias=bh.cuda(), out_mode=ops.OUT_NCDHW)
    assert rel_l2(o.cpu(), F.conv3d(xh, wh, bh, padding=1)) < TOL_MFMA


def test_conv5_stem_and_head(ops):
    """5x5x5 / pad 2 convolutions of ddpm_res128 (stem: Cin 4 -> 128; head: 64 -> 4 straight to NCDHW)."""
    S = 8
    x = _rand((2, 4, S, S, S), 50); w = _rand((128, 4, 5, 5, 5), 51, 0.05); b = _rand((128,), 52)
    x16 = ops.ncdhw_to_s16b(x.cuda(), 16)
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C5_128_K16, "cuda")
    out = ops.f32b_empty(2, 128, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C5_128_K16, a=pw.data, b=x16, out=out, batch=2, rows=128, rows_alloc=128,
                  kdim=16, dims=(S, S, S), bias=b.cuda())
    assert rel_l2(ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), F.conv3d(x, w, b, padding=2)) < TOL_MFMA
    xh = _rand((2, 64, S, S, S), 53); wh = _rand((4, 64, 5, 5, 5), 54, 0.03); bh = _rand((4,), 55)
    s, _, _ = _to_s16(ops, xh)
    pwh = ops.PackedWeight(wh.cuda(), "conv", ops.CFG_C5_32_K16, "cuda")
    o = torch.empty((2, 4, S, S, S), device="cuda")
    ops.gemm_conv(cfg=ops.CFG_C5_32_K16, a=pwh.data, b=s, out=o, batch=2, rows=4, rows_alloc=8, kdim=64, dims=(S, S, S),
                  bias=bh.cuda(), out_mode=ops.OUT_NCDHW)
    assert rel_l2(o.cpu(), F.conv3d(xh, wh, bh, padding=2)) < TOL_MFMA


@pytest.mark.parametrize("P,cin,cout,cfg_name", [(512, 64, 128, "CFG_G1_128"), (64, 96, 128, "CFG_G1_128_LOW"),
                                                 (64, 64, 64, "CFG_G1_64_LOW")])
def test_nin_gemm_and_s16b_out(ops, P, cin, cout, cfg_name):
    cfg = getattr(ops, cfg_name)
    x = _rand((2, cin, 1, 1, P), 17); W = _rand((cin, cout), 18, 0.1); b = _rand((cout,), 19)
    s, _, _ = _to_s16(ops, x)
    pw = ops.PackedWeight(W.cuda(), "nin", cfg, "cuda")
    ref = torch.einsum("bcdhw,co->bodhw", x, W) + b[None, :, None, None, None]
    out = ops.f32b_empty(2, cout, P, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=s, out=out, batch=2, rows=cout, rows_alloc=cout, kdim=pw.kdim, dims=(1, 1, P), bias=b.cuda())
    assert rel_l2(ops.f32b_to_ncdhw(out, (1, 1, P)).cpu(), ref) < TOL_MFMA
    o16 = ops.s16b_empty(2, cout, P, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=s, out=o16, batch=2, rows=cout, rows_alloc=cout, kdim=pw.kdim, dims=(1, 1, P),
                  bias=b.cuda(), out_mode=ops.OUT_S16B)
    assert rel_l2(ops.s16b_to_ncdhw(o16, (1, 1, P)).cpu(), ref) < TOL_MFMA


@pytest.mark.parametrize("Cc,groups_ch", [(32, 1), (128, 4), (96, 3)])
def test_groupnorm_silu_split(ops, Cc, groups_ch):
    B, S = 2, 8
    xa = _rand((B, Cc // 2 if Cc % 16 == 0 else Cc, S, S, S), 20) * 3 + 1.5
    parts_t = [xa] if xa.shape[1] == Cc else [xa, _rand((B, Cc - xa.shape[1], S, S, S), 21) - 0.7]
    gamma = 1 + 0.1 * _rand((Cc,), 22); beta = 0.1 * _rand((Cc,), 23)
    P = S ** 3
    parts = [(ops.ncdhw_to_f32b(t.cuda()), t.shape[1]) for t in parts_t]
    prm = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, P)
    xcat = torch.cat(parts_t, dim=1)
    for silu in (True, False):
        y = ops.s16b_to_ncdhw(ops.gn_apply(parts, prm, B, P, norm=True, silu=silu), (S, S, S)).cpu()
        ref = F.group_norm(xcat, 32, gamma, beta, eps=1e-6)
        ref = F.silu(ref) if silu else ref
        assert rel_l2(y, ref) < 1e-5, (Cc, silu)


def test_softmax_keys(ops):
    B, N = 2, 256
    s = _rand((B, N, N), 24) * 4     # s[b][key][query]
    s_blk = s.reshape(B, N // 8, 8, N).permute(0, 1, 3, 2).contiguous().cuda()
    p = ops.softmax_keys(s_blk, B, N, N)          # [B][N/8][2][N][8]
    pf = (p[:, :, 0].float() + p[:, :, 1].float()).cpu()  # [B][N/8][Nq][8]
    got = pf.permute(0, 1, 3, 2).reshape(B, N, N)
    ref = F.softmax(s, dim=1)
    assert rel_l2(got, ref) < 2e-5


def test_temb_and_linear(ops):
    from oracle import unet_oracle as uo
    t = torch.tensor([999.0, 998.001, 500.3, 0.999])
    emb = ops.timestep_embedding(t.cuda(), 128).cpu()
    ref = uo.timestep_embedding(t, 128)
    assert float((emb - ref).abs().max()) < 2e-4   # sin/cos of arguments up to ~1e3: 1 ulp of the argument
    x = _rand((4, 128), 25); w = _rand((512, 128), 26, 0.1); b = _rand((512,), 27)
    y = ops.linear(x.cuda(), w.cuda(), b.cuda(), silu_in=True).cpu()
    assert rel_l2(y, F.linear(F.silu(x), w, b)) < TOL_F32


def test_ancestral_step_bit_exact(ops):
    from oracle import unet_oracle as uo
    B, Cc, S = 2, 4, 8
    x, e, z = _rand((B, Cc, S, S, S), 28), _rand((B, Cc, S, S, S), 29), _rand((B, Cc, S, S, S), 30)
    mask = (torch.rand(S, S, S, generator=torch.Generator().manual_seed(31)) < 0.3).float()
    t = torch.tensor(0.731)
    betas, _, sq1m = uo.vpsde_tables()
    k = (t * 999).long()
    coef = torch.stack([betas[k], sq1m[k], torch.sqrt(1.0 - betas[k]), torch.sqrt(betas[k])]).expand(B, 4).contiguous()
    xn, xm = ops.ancestral_step(x.cuda(), e.cuda(), z.cuda(), mask.reshape(-1).cuda(), coef.cuda())
    rn, rm = uo.ancestral_step(x, e, z, t, mask)
    assert torch.equal(xm.cpu(), rm) and torch.equal(xn.cpu(), rn)


def test_inpaint_blend_and_renoise(ops):
    B, Cc, S = 2, 4, 8
    x = _rand((B, Cc, S, S, S), 32); src = _rand((1, S, S, S), 33); z = _rand((B, S, S, S), 34)
    pm = (torch.rand(S, S, S, generator=torch.Generator().manual_seed(35)) < 0.5).float()
    gm = (torch.rand(S, S, S, generator=torch.Generator().manual_seed(36)) < 0.5).float()
    xg = x.cuda().clone()
    ops.inpaint_blend_(xg, src.cuda(), pm.reshape(-1).cuda(), gm.reshape(-1).cuda(), 0)
    ref = x.clone(); ref[:, 0] = (x[:, 0] * (1 - pm) + src * pm) * gm
    assert torch.equal(xg.cpu(), ref)
    coef = torch.tensor([[0.9, 0.4], [0.8, 0.6]])
    xm = torch.zeros_like(xg)
    ops.inpaint_renoise_(xg, xm, z.cuda(), pm.reshape(-1).cuda(), gm.reshape(-1).cuda(), coef.cuda(), 0)
    upd = coef[:, 0, None, None, None] * ref[:, 0] + coef[:, 1, None, None, None] * z
    ref2 = ref.clone(); ref2[:, 0] = (ref[:, 0] * (1 - pm) + upd * pm) * gm
    assert rel_l2(xg.cpu(), ref2) < 1e-6 and torch.equal(xm[:, 0].cpu(), xg[:, 0].cpu())


def test_conv_epilogue_groupnorm_statistics(hip_lib):
    """MdGemmConvArgs.stats: the dedicated 3x3x3 kernel adds per-(sample, channel) sum / sum of squares of what it
    writes (after bias + residual) -- must equal md_gn_stats over the written tensor, and a GroupNorm fed from the
    attached sums must equal one fed from a fresh statistics pass (hip_ops.FUSE_GN_STATS A/B)."""
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    B, Ci, Co, S = 8, 64, 256, 16          # 16 tiles x 8 samples x 2 row tiles = 256 workgroups: no split-K
    g = torch.Generator().manual_seed(0)
    conv = layers.ddpm_conv3x3(Ci, Co).cuda()
    conv.weight.data = (torch.rand(conv.weight.shape, generator=g) - 0.5).cuda() * 0.1
    conv.bias.data = torch.randn(Co, generator=g).cuda()
    x = torch.randn((B, Ci, S, S, S), generator=g).cuda()
    res = ops.ncdhw_to_f32b(torch.randn((B, Co, S, S, S), generator=g).cuda())
    pw = ops.PackedWeight(conv.weight, "conv", ops.CFG_C3_128_FAST, x.device)
    a = ops.ncdhw_to_s16b(x, Ci)
    y = layers.run_conv3(pw, a, B, S, bias=conv.bias, residual=res, want_stats=True)
    sums = y._md_sums
    yn = ops.f32b_to_ncdhw(y, (S, S, S)).double()
    ref = torch.stack([yn.sum(dim=(2, 3, 4)), (yn * yn).sum(dim=(2, 3, 4))], dim=-1)
    assert torch.allclose(sums, ref, rtol=1e-5, atol=1e-3)
    gn = torch.nn.GroupNorm(32, Co, eps=1e-6).cuda()
    gn.weight.data, gn.bias.data = torch.randn(Co, generator=g).cuda(), torch.randn(Co, generator=g).cuda()
    P = S ** 3
    p_fused = ops.gn_params([(y, Co)], gn.weight, gn.bias, B, P)
    ops.FUSE_GN_STATS = False
    try:
        p_plain = ops.gn_params([(y, Co)], gn.weight, gn.bias, B, P)
    finally:
        ops.FUSE_GN_STATS = True
    assert rel_l2(p_fused.cpu(), p_plain.cpu()) < 1e-6
    # concatenated GroupNorm: one part with attached sums, one without
    z = ops.ncdhw_to_f32b(torch.randn((B, 64, S, S, S), generator=g).cuda())
    gn2 = torch.nn.GroupNorm(32, Co + 64, eps=1e-6).cuda()
    p_cat = ops.gn_params([(y, Co), (z, 64)], gn2.weight, gn2.bias, B, P)
    ops.FUSE_GN_STATS = False
    try:
        p_cat_plain = ops.gn_params([(y, Co), (z, 64)], gn2.weight, gn2.bias, B, P)
    finally:
        ops.FUSE_GN_STATS = True
    assert rel_l2(p_cat.cpu(), p_cat_plain.cpu()) < 1e-6


@pytest.mark.parametrize("cfg_name,S,ks", [("CFG_C3_128_FAST", 8, 4), ("CFG_C3_LOW", 4, 2)])
def test_splitk_finish_groupnorm_statistics(ops, cfg_name, S, ks):
    """With split-K the GroupNorm sums come from the finish kernel (md_splitk_reduce_stats_kernel): same output bits as the
    plain finish, sums equal to float64 sums of what was written (bias + residual included)."""
    cfg = getattr(ops, cfg_name)
    B, cin, cout = 3, 128, 136
    x = _rand((B, cin, S, S, S), 50); w = _rand((cout, cin, 3, 3, 3), 51, 0.05); bias = _rand((cout,), 52)
    res = ops.ncdhw_to_f32b(_rand((B, cout, S, S, S), 53).cuda())
    s16, _, _ = _to_s16(ops, x)
    pw = ops.PackedWeight(w.cuda(), "conv", cfg, "cuda")
    kw = dict(cfg=cfg, a=pw.data, b=s16, batch=B, rows=cout, rows_alloc=cout, kdim=cin, dims=(S, S, S), bias=bias.cuda(),
              residual=res, res_bstride=cout * S ** 3, ksplit=ks)
    plain = ops.gemm_conv(out=ops.f32b_empty(B, cout, S ** 3, "cuda"), **kw)
    stats = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
    out = ops.gemm_conv(out=ops.f32b_empty(B, cout, S ** 3, "cuda"), stats=stats, **kw)
    assert torch.equal(out, plain)
    y = ops.f32b_to_ncdhw(out, (S, S, S)).double()
    ref = torch.stack([y.sum(dim=(2, 3, 4)), (y * y).sum(dim=(2, 3, 4))], dim=-1)
    assert torch.allclose(stats, ref, rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("case", ["one_part_gn_silu", "two_parts_gn_silu", "ups_split_only", "gn_no_silu"])
def test_conv3_fused_groupnorm_silu_operand(ops, case):
    """MD_B_F32B_GN: the dedicated conv reads fp32 F32B parts (one tensor or a channel concat of two) and applies the
    GroupNorm affine + SiLU + bf16 split in its halo loader -- against F.conv3d(F.silu(F.group_norm(cat(parts)))) in
    fp32 on the CPU, incl. zero padding of the ACTIVATED tensor at the grid boundary (beta != 0 makes silu(gn(0)) != 0)
    and against the two-pass path (md_gn_apply + conv on the S16B tensor)."""
    B, S, cout = 2, 8, 128
    ups = case == "ups_split_only"
    cs = {"one_part_gn_silu": [64], "two_parts_gn_silu": [96, 32], "ups_split_only": [32], "gn_no_silu": [64]}[case]
    cin = sum(cs)
    Sin = S // 2 if ups else S
    Pin = Sin ** 3
    xs = [_rand((B, c, Sin, Sin, Sin), 10 + i) * (1.0 + i) + 0.3 * i for i, c in enumerate(cs)]
    x = torch.cat(xs, 1)
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 20), 0.5 * _rand((cin,), 21)
    w = _rand((cout, cin, 3, 3, 3), 22, 0.05)
    bias = _rand((B, cout), 23)
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    pw = ops.PackedWeight(w.cuda(), "conv", ops.CFG_C3_128_FAST, "cuda")
    silu = case not in ("gn_no_silu", "ups_split_only")   # (SiLU without the affine is rejected by the library: MD_ERR_BAD_ARG)
    ac = None
    if not ups:
        prm, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, Pin, want_ac=True)
        ref_in = F.group_norm(x, 32, gamma, beta, eps=1e-6)
        ref_in = F.silu(ref_in) if silu else ref_in
    else:
        ref_in = F.interpolate(x, scale_factor=2, mode="nearest")
    out = ops.f32b_empty(B, cout, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_128_FAST, a=pw.data, b=None, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S, S, S), bias=bias.cuda(), bias_bstride=cout, ups=1 if ups else 0,
                  b_f32=dict(parts=parts, ac=ac, silu=silu))
    y = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()
    ref = F.conv3d(ref_in, w, padding=1) + bias[:, :, None, None, None]
    e = rel_l2(y, ref)
    # two-pass path on the same inputs
    a16 = ops.gn_apply(parts, prm if not ups else None, B, Pin, norm=not ups, silu=silu and not ups)
    out2 = ops.f32b_empty(B, cout, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3_128_FAST, a=pw.data, b=a16, out=out2, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(S, S, S), bias=bias.cuda(), bias_bstride=cout, ups=1 if ups else 0)
    e2 = rel_l2(y, ops.f32b_to_ncdhw(out2, (S, S, S)).cpu())
    print(f"fused operand ({case}): vs torch fp32 {e:.2e}, vs two-pass HIP path {e2:.2e}")
    assert e < TOL_MFMA and e2 < TOL_MFMA
    if ups:
        assert e2 == 0.0          # no transcendental involved: the in-loader split is the same arithmetic


@pytest.mark.parametrize("case", ["low_4cubed_two_parts_splitk", "low_4cubed_gn_no_silu"])
def test_generic_conv_fused_groupnorm_silu_operand(ops, case):
    """MD_B_F32B_GN on the generic conv configuration built with BF (csrc/gemm_conv.hip: the 4^3 level MD_CFG_C3_LOW):
    GroupNorm affine + SiLU + zero padding + split in the halo loader, against torch fp32 and against the two-pass path
    (md_gn_apply + the same configuration on S16B)."""
    B, cs, rows, S, cfg, kshape, ks = 3, ([64, 32] if "two_parts" in case else [64]), 128, 4, ops.CFG_C3_LOW, (3, 3, 3), (2 if "splitk" in case else 1)
    cin = sum(cs)
    P = S ** 3
    xs = [_rand((B, c, S, S, S), 40 + i) * (1.0 + i) + 0.3 * i for i, c in enumerate(cs)]
    x = torch.cat(xs, 1)
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 41), 0.5 * _rand((cin,), 42)
    w = _rand((rows, cin) + kshape, 43, 0.05)
    bias = _rand((rows,), 44)
    silu = "no_silu" not in case
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    prm, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, P, want_ac=True)
    pw = ops.PackedWeight(w.cuda(), "conv", cfg, "cuda")
    rows_alloc = ((rows + 7) // 8) * 8
    kw = dict(cfg=cfg, a=pw.data, batch=B, rows=rows, rows_alloc=rows_alloc, kdim=cin, dims=(S, S, S), bias=bias.cuda(), ksplit=ks)
    out = ops.gemm_conv(b=None, out=ops.f32b_empty(B, rows_alloc, P, "cuda"), b_f32=dict(parts=parts, ac=ac, silu=silu), **kw)
    y = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()[:, :rows]
    ref_in = F.group_norm(x, 32, gamma, beta, eps=1e-6)
    ref_in = F.silu(ref_in) if silu else ref_in
    ref = F.conv3d(ref_in, w, bias, padding=(1, 1, 0) if kshape[2] == 1 else 1)
    e = rel_l2(y, ref)
    a16 = ops.gn_apply(parts, prm, B, P, norm=True, silu=silu)
    out2 = ops.gemm_conv(b=a16, out=ops.f32b_empty(B, rows_alloc, P, "cuda"), **kw)
    e2 = rel_l2(y, ops.f32b_to_ncdhw(out2, (S, S, S)).cpu()[:, :rows])
    print(f"generic fused operand ({case}): vs torch fp32 {e:.2e}, vs two-pass HIP path {e2:.2e}")
    assert e < TOL_MFMA and e2 < TOL_MFMA


@pytest.mark.parametrize("B,cout,S", [(2, 128, 8), (3, 136, 16)])
def test_conv3_stem_kernel(ops, B, cout, S):
    """md_conv3_stem (csrc/conv3_stem.hip: the dx-folded 3x3x3 conv from 4 channels with a batch-shared residual and the
    GroupNorm sums of its output) against torch fp32, against the generic tile it replaces (bit-identical: same products,
    same order) and float64 sums of what it wrote.  Second case: several tiles per axis, a partial second row tile."""
    x = _rand((B, 4, S, S, S), 60); w = _rand((cout, 4, 3, 3, 3), 61, 0.1); bias = _rand((cout,), 62)
    res = _rand((1, cout, S, S, S), 63)
    res_f = ops.ncdhw_to_f32b(res.cuda())
    w2 = w.permute(0, 1, 4, 2, 3).reshape(cout, 12, 3, 3, 1).contiguous().cuda()
    pw = ops.PackedWeight(w2, "conv", ops.CFG_C3X_128_K16, "cuda")
    xf = ops.ncdhw_to_s16b_xfold(x.cuda(), 3, 16)
    assert ops.conv3_stem_ok(pw.rows, pw.kdim, S)
    stats = torch.zeros((B, cout, 2), dtype=torch.float64, device="cuda")
    out = ops.conv3_stem(pw, xf, B, S, bias=bias.cuda(), residual=res_f, stats=stats)
    got = ops.f32b_to_ncdhw(out, (S, S, S)).cpu()
    ref = F.conv3d(x, w, bias, padding=1) + res
    e = rel_l2(got, ref)
    print(f"conv3_stem 4->{cout} @ {S}^3: vs torch fp32 {e:.2e}")
    assert e < TOL_MFMA
    g64 = got.double()
    assert torch.allclose(stats.cpu(), torch.stack([g64.sum(dim=(2, 3, 4)), (g64 * g64).sum(dim=(2, 3, 4))], dim=-1), rtol=1e-5, atol=1e-3)
    old = ops.f32b_empty(B, cout, S ** 3, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3X_128_K16, a=pw.data, b=xf, out=old, batch=B, rows=cout, rows_alloc=cout, kdim=16, dims=(S, S, S),
                  bias=bias.cuda(), residual=res_f, res_bstride=0)
    assert torch.equal(out, old)


@pytest.mark.parametrize("B,cin,S", [(2, 64, 8), (1, 128, 16)])
def test_conv3_head_fused_kernel(ops, B, cin, S):
    """md_conv3_head (csrc/conv3_head.hip: GroupNorm affine + SiLU + split in the loader of the dx-folded 3x3x3 head, then
    md_fold_dx) against torch fp32 GroupNorm -> SiLU -> conv3d, and against the two-pass path it replaces (md_gn_apply +
    MD_CFG_C3X_32 + md_fold_dx); bit-identical between launches."""
    co = 4
    x = _rand((B, cin, S, S, S), 90) * 1.5 + 0.2
    gamma, beta = 1.0 + 0.2 * _rand((cin,), 91), 0.5 * _rand((cin,), 92)
    w = _rand((co, cin, 3, 3, 3), 93, 0.05); bias = _rand((co,), 94)
    P = S ** 3
    parts = [(ops.ncdhw_to_f32b(x.cuda()), cin)]
    prm, ac = ops.gn_params(parts, gamma.cuda(), beta.cuda(), B, P, want_ac=True)
    w2 = w.permute(0, 4, 1, 2, 3).reshape(co * 3, cin, 3, 3, 1).contiguous().cuda()
    assert ops.conv3_head_ok(co * 3, cin, S)
    pw = ops.PackedWeight(w2, "conv", ops.CFG_HEAD_PACK, "cuda")
    y = ops.conv3_head(pw, parts[0][0], ac, B, S, 16)
    out = ops.fold_dx(y, bias.cuda(), B, co, 3, 16, S)
    ref = F.conv3d(F.silu(F.group_norm(x, 32, gamma, beta, eps=1e-6)), w, bias, padding=1)
    e = rel_l2(out.cpu(), ref)
    a16 = ops.gn_apply(parts, prm, B, P, norm=True, silu=True)
    pw_old = ops.PackedWeight(w2, "conv", ops.CFG_C3X_32, "cuda")
    y_old = ops.f32b_empty(B, 16, P, "cuda")
    ops.gemm_conv(cfg=ops.CFG_C3X_32, a=pw_old.data, b=a16, out=y_old, batch=B, rows=12, rows_alloc=16, kdim=cin, dims=(S, S, S))
    e2 = rel_l2(out.cpu(), ops.fold_dx(y_old, bias.cuda(), B, co, 3, 16, S).cpu())
    print(f"conv3_head {cin}->4 @ {S}^3: vs torch fp32 {e:.2e}, vs the two-pass path {e2:.2e}")
    assert e < TOL_MFMA and e2 < TOL_MFMA
    assert torch.equal(y, ops.conv3_head(pw, parts[0][0], ac, B, S, 16))


def test_pack_batch_bit_identical_to_single_launches(ops):
    """md_pack_batch (csrc/pack_batch.hip: all queued weight packs of a training step in one launch) writes the same tiles,
    bit for bit, as md_pack_weights / md_wino_pack_weights one weight at a time: conv / data-gradient / NIN / rows kinds,
    several tile geometries, Winograd fragments forward and flipped, a partial row tile and a padded K."""
    w1, w2, w3 = _rand((128, 64, 3, 3, 3), 80, 0.05).cuda(), _rand((136, 96, 3, 3, 3), 81, 0.05).cuda(), _rand((256, 128, 3, 3, 3), 82, 0.05).cuda()
    m1, m2 = _rand((96, 128), 83).cuda(), _rand((40, 72), 84).cuda()

    def make():
        return [ops.PackedWeight(w1, "conv", ops.CFG_C3_128_FAST, "cuda"), ops.PackedWeight(w2, "conv_dgrad", ops.CFG_C3_LOW, "cuda"),
                ops.PackedWeight(w2, "conv", ops.CFG_S2_PACK, "cuda"), ops.PackedWeight(m1, "nin", ops.CFG_G1_128, "cuda"),
                ops.PackedWeight(m2, "rows", ops.CFG_G1_64_LOW, "cuda"), ops.WinoWeight(w3, "cuda"),
                ops.WinoWeight(w3, "cuda", kind="conv_dgrad"), ops.WinoWeight(w1, "cuda")]

    assert ops.PACK_BATCH
    batch = make()
    for o in batch:
        o.request()
    ops.flush_packs()
    assert all(o._data is not None for o in batch)
    ops.PACK_BATCH = False
    try:
        single = make()
        for o in single:
            _ = o.data
    finally:
        ops.PACK_BATCH = True
    for a, b in zip(batch, single):
        assert a.data.numel() == b.data.numel() and torch.equal(a.data.view(torch.int16), b.data.view(torch.int16))


def test_gn_finalize_folded_affine(ops):
    B, Cc, S = 2, 64, 4
    x = _rand((B, Cc, S, S, S), 30) * 2.0 + 1.5
    gamma, beta = 1.0 + 0.2 * _rand((Cc,), 31), 0.5 * _rand((Cc,), 32)
    prm, ac = ops.gn_params([(ops.ncdhw_to_f32b(x.cuda()), Cc)], gamma.cuda(), beta.cuda(), B, S ** 3, want_ac=True)
    prm, ac = prm.cpu(), ac.cpu()
    assert rel_l2(ac[..., 0], prm[..., 1]) == 0.0
    assert rel_l2(ac[..., 1], prm[..., 2] - prm[..., 0] * prm[..., 1]) < 1e-6
    y = x * ac[..., 0][:, :, None, None, None] + ac[..., 1][:, :, None, None, None]
    assert rel_l2(y, F.group_norm(x, 32, gamma, beta, eps=1e-6)) < 2e-6


@pytest.mark.parametrize("cfg_name", ["CFG_G1_128", "CFG_G1_128_N128"])
def test_nin_gemm_fp32_parts_operand(ops, cfg_name):
    """MD_B_F32B_GN on the 1x1x1 configuration: the shortcut GEMM reads the fp32 block input (a channel concat of two
    tensors) and splits it in its loader -- bit-identical to the GEMM on the pre-split S16B tensor."""
    B, S, cs, cout = 2, 8, [96, 32], 128
    P, cin = S ** 3, sum(cs)
    xs = [_rand((B, c, S, S, S), 40 + i) for i, c in enumerate(cs)]
    W, bias = _rand((cin, cout), 42, 0.1), _rand((cout,), 43)
    parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
    cfg = getattr(ops, cfg_name)
    pw = ops.PackedWeight(W.cuda(), "nin", cfg, "cuda")
    out = ops.f32b_empty(B, cout, P, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=None, out=out, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(1, 1, P), bias=bias.cuda(), b_f32=dict(parts=parts, ac=None, silu=False))
    a16 = ops.gn_apply(parts, None, B, P, norm=False, silu=False)
    out2 = ops.f32b_empty(B, cout, P, "cuda")
    ops.gemm_conv(cfg=cfg, a=pw.data, b=a16, out=out2, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                  dims=(1, 1, P), bias=bias.cuda())
    assert torch.equal(out, out2)
    ref = torch.einsum("bcdhw,co->bodhw", torch.cat(xs, 1), W) + bias[None, :, None, None, None]
    assert rel_l2(ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), ref) < TOL_MFMA


@pytest.mark.parametrize("cs", [[128], [128, 128], [96, 32], [208, 48]])
def test_nin_stream_kernel(ops, cs):
    """md_nin_f32 (weights resident in LDS, fp32 input streamed HBM -> registers): bit-identical to md_gemm_conv on the
    pre-split operand (same bf16x3 products, same accumulation order over K), on one- and two-part inputs, more tiles than
    workgroups (B * P / 256 = 512 tiles on 256 workgroups) and a ragged tile count."""
    cin, cout = sum(cs), 128
    for B, S in ((4, 32), (3, 16)):       # 512 tiles; 48 tiles (fewer than workgroups)
        P = S ** 3
        xs = [_rand((B, c, S, S, S), 50 + i) for i, c in enumerate(cs)]
        W, bias = _rand((cin, cout), 52, 0.1), _rand((cout,), 53)
        parts = [(ops.ncdhw_to_f32b(t.cuda()), c) for t, c in zip(xs, cs)]
        pw = ops.PackedWeight(W.cuda(), "nin", ops.CFG_G1_128, "cuda")
        out = ops.nin_f32(parts, pw, bias.cuda(), B, P)
        a16 = ops.gn_apply(parts, None, B, P, norm=False, silu=False)
        out2 = ops.f32b_empty(B, cout, P, "cuda")
        ops.gemm_conv(cfg=ops.CFG_G1_128, a=pw.data, b=a16, out=out2, batch=B, rows=cout, rows_alloc=cout, kdim=cin,
                      dims=(1, 1, P), bias=bias.cuda())
        assert torch.equal(out, out2), (cs, B, S)
        ref = torch.einsum("bcdhw,co->bodhw", torch.cat(xs, 1), W) + bias[None, :, None, None, None]
        assert rel_l2(ops.f32b_to_ncdhw(out, (S, S, S)).cpu(), ref) < TOL_MFMA
