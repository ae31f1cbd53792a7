"""Backward pass of the layers on the HIP path vs torch autograd through the oracle's functions (fp32 CPU).
Tolerance 2e-4 rel-L2: the gradients go through bf16x3 contractions twice (dgrad/wgrad)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _randn(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def _load(layer, seed):
    from meshdiffusion_amd import synth
    sd = synth.sensitised_state_dict(layer.state_dict(), seed=seed)
    layer.load_state_dict(sd)
    return sd


def _f32b(ops, t):
    return ops.ncdhw_to_f32b(t.cuda())


@pytest.mark.parametrize("cin_parts,cout,S,B", [((64,), 64, 8, 8), ((64, 64), 64, 8, 8), ((128,), 64, 4, 8),
                                                ((64, 64), 64, 8, 6), ((64,), 64, 8, 11)])
def test_resnet_block_backward(hip_lib, cin_parts, cout, S, B):
    """B = 6 / 11: per-GPU batches that are not a multiple of the 8-sample wgrad block (reference res64 config:
    training.batch_size 48 over 8 GPUs = 6 per GPU)."""
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    from oracle import unet_oracle as uo
    cin = sum(cin_parts)
    blk = layers.ResnetBlockDDPM(act=torch.nn.SiLU(), in_ch=cin, out_ch=cout, temb_dim=128, dropout=0.0)
    sd = _load(blk, 3)
    blk = blk.cuda().train()
    xs = [_randn((B, c, S, S, S), 10 + i) for i, c in enumerate(cin_parts)]
    temb, dy = _randn((B, 128), 2), _randn((B, cout, S, S, S), 4)
    # reference gradients: torch autograd through the oracle restatement
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = [x.clone().requires_grad_(True) for x in xs]
    tr = temb.clone().requires_grad_(True)
    y_ref = uo.resnet_block(sdr, torch.cat(xr, 1), tr)
    y_ref.backward(dy)
    # HIP path
    P = S ** 3
    parts = [(_f32b(ops, x), c) for x, c in zip(xs, cin_parts)]
    tape = []
    with torch.no_grad():
        y = blk.forward_blocked(parts, B, P, temb.cuda(), tape=tape)
        assert rel_l2(ops.f32b_to_ncdhw(y, (S, S, S)).cpu(), y_ref.detach()) < 1e-4
        dparts, dbias0 = blk.backward_blocked(tape[0], _f32b(ops, dy))
    for g, x in zip(dparts, xr):
        assert rel_l2(ops.f32b_to_ncdhw(g, (S, S, S)).cpu(), x.grad) < TOL
    names = ["Conv_0.weight", "Conv_1.weight", "Conv_1.bias", "GroupNorm_0.weight", "GroupNorm_0.bias",
             "GroupNorm_1.weight", "GroupNorm_1.bias"] + (["NIN_0.W", "NIN_0.b"] if cin != cout else [])
    params = dict(blk.named_parameters())
    for n in names:
        assert rel_l2(params[n].grad.cpu(), sdr[n].grad) < TOL, n
    # FiLM: d(bias0)[b, co] == d(Dense_0 output); Conv_0.bias grad == its batch sum
    d_film_ref = torch.autograd.grad(uo.resnet_block(sdr, torch.cat(xr, 1), tr).mul(dy).sum(), sdr["Dense_0.bias"])[0]
    assert rel_l2(dbias0.sum(0).cpu(), d_film_ref) < TOL


def test_resnet_block_dropout_forward_backward(hip_lib):
    """Training-mode dropout (layers.py:682): the HIP path regenerates its mask from (p, seed); the oracle gets
    the same mask explicitly (md_dropout_scale), so outputs and every gradient must agree as without dropout.
    Also: keep rate ~ 1-p, masks differ between seeds / blocks and repeat for the same torch seed."""
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    from oracle import unet_oracle as uo
    B, cin, cout, S, p = 8, 128, 64, 8, 0.3
    P = S ** 3
    blk = layers.ResnetBlockDDPM(act=torch.nn.SiLU(), in_ch=cin, out_ch=cout, temb_dim=128, dropout=p)
    sd = _load(blk, 3)
    blk = blk.cuda().train()
    x, temb, dy = _randn((B, cin, S, S, S), 10), _randn((B, 128), 2), _randn((B, cout, S, S, S), 4)
    parts = [(_f32b(ops, x), cin)]
    tape = []
    torch.manual_seed(77)
    with torch.no_grad():
        y = blk.forward_blocked(parts, B, P, temb.cuda(), tape=tape)
        dparts, _ = blk.backward_blocked(tape[0], _f32b(ops, dy))
    pd, seed = tape[0]["drop"]
    assert pd == p
    scale = ops.f32b_to_ncdhw(ops.dropout_scale(B, cout, P, p, seed, x.cuda().device), (S, S, S)).cpu()
    vals = torch.unique(scale)
    assert vals.numel() == 2 and vals[0] == 0 and abs(float(vals[1]) - 1 / (1 - p)) < 1e-6
    keep = float((scale > 0).double().mean())
    assert abs(keep - (1 - p)) < 4 * (p * (1 - p) / scale.numel()) ** 0.5 + 1e-4
    per_channel = (scale > 0).double().mean(dim=(0, 2, 3, 4))
    assert float((per_channel - (1 - p)).abs().max()) < 0.05           # no dead / always-on channel lanes
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr, tr = x.clone().requires_grad_(True), temb.clone().requires_grad_(True)
    y_ref = uo.resnet_block(sdr, xr, tr, drop=scale)
    y_ref.backward(dy)
    assert rel_l2(ops.f32b_to_ncdhw(y, (S, S, S)).cpu(), y_ref.detach()) < 1e-4
    assert rel_l2(ops.f32b_to_ncdhw(dparts[0], (S, S, S)).cpu(), xr.grad) < TOL
    params = dict(blk.named_parameters())
    for n in ["Conv_0.weight", "Conv_1.weight", "GroupNorm_0.weight", "GroupNorm_1.weight", "GroupNorm_1.bias", "NIN_0.W"]:
        assert rel_l2(params[n].grad.cpu(), sdr[n].grad) < TOL, n
    # eval mode: identity; same torch seed: same mask; next call: another mask
    tape2, tape3 = [], []
    torch.manual_seed(77)
    with torch.no_grad():
        y2 = blk.forward_blocked(parts, B, P, temb.cuda(), tape=tape2)
        y3 = blk.forward_blocked(parts, B, P, temb.cuda(), tape=tape3)
        y_eval = blk.eval().forward_blocked(parts, B, P, temb.cuda())
    assert torch.equal(y2, y) and tape2[0]["drop"] == tape[0]["drop"] and tape3[0]["drop"][1] != seed
    assert not torch.equal(y3, y)
    assert rel_l2(ops.f32b_to_ncdhw(y_eval, (S, S, S)).cpu(), uo.resnet_block(sd, x, temb)) < 1e-4


def test_resnet_block_training_through_winograd_path(hip_lib):
    """A block big enough for the Winograd path (hip_ops.wino_ok: 256 workgroups) in TRAINING mode: two-part input
    (torch.cat of a skip connection), FiLM, dropout 0.1 -- forward convs (md_wino_prep with the dropout mask of md_gn_apply +
    md_conv3_wino) and data-gradient convs (conv_dgrad tiles) against torch autograd through the oracle with the same
    explicit mask; the launches are checked to be the Winograd kernels, and the direct path must give the same numbers."""
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    from oracle import unet_oracle as uo
    B, cin_parts, cout, S, p = 16, (128, 128), 128, 16, 0.1
    cin, P = sum(cin_parts), S ** 3
    assert ops.wino_ok(cout, cin, S, B) and ops.WINO and ops.WINO_TRAIN_FWD
    blk = layers.ResnetBlockDDPM(act=torch.nn.SiLU(), in_ch=cin, out_ch=cout, temb_dim=128, dropout=p)
    sd = _load(blk, 5)
    blk = blk.cuda().train()
    xs = [_randn((B, c, S, S, S), 20 + i) for i, c in enumerate(cin_parts)]
    temb, dy = _randn((B, 128), 22), _randn((B, cout, S, S, S), 23)
    parts = [(_f32b(ops, x), c) for x, c in zip(xs, cin_parts)]

    def run():
        for q in blk.parameters():
            q.grad = None
        tape = []
        torch.manual_seed(99)
        ops.PROFILE = []
        try:
            with torch.no_grad():
                y = blk.forward_blocked(parts, B, P, temb.cuda(), tape=tape)
                dparts, _ = blk.backward_blocked(tape[0], _f32b(ops, dy))
            tags = [e[0] for e in ops.PROFILE]
        finally:
            ops.PROFILE = None
        grads = {n: q.grad.clone() for n, q in blk.named_parameters() if q.grad is not None}
        return y, dparts, grads, tape[0]["drop"], tags

    y, dparts, grads, (pd, seed), tags = run()
    assert tags.count("wino") == 4 and tags.count("wino_prep") == 4        # Conv_0, Conv_1 forward + their two data gradients
    scale = ops.f32b_to_ncdhw(ops.dropout_scale(B, cout, P, p, seed, torch.device("cuda", 0)), (S, S, S)).cpu()
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = [x.clone().requires_grad_(True) for x in xs]
    tr = temb.clone().requires_grad_(True)
    y_ref = uo.resnet_block(sdr, torch.cat(xr, 1), tr, drop=scale)
    y_ref.backward(dy)
    assert rel_l2(ops.f32b_to_ncdhw(y, (S, S, S)).cpu(), y_ref.detach()) < 1e-4
    for g, x in zip(dparts, xr):
        assert rel_l2(ops.f32b_to_ncdhw(g, (S, S, S)).cpu(), x.grad) < TOL
    for n in ["Conv_0.weight", "Conv_1.weight", "GroupNorm_0.weight", "GroupNorm_1.weight", "GroupNorm_1.bias", "NIN_0.W"]:
        assert rel_l2(grads[n].cpu(), sdr[n].grad) < TOL, n
    # the direct kernels on the same inputs and mask
    ops.WINO = False
    try:
        y2, dparts2, grads2, drop2, tags2 = run()
    finally:
        ops.WINO = True
    assert "wino" not in tags2 and drop2 == (pd, seed)
    assert rel_l2(y.cpu(), y2.cpu()) < 2e-5
    for g, g2 in zip(dparts, dparts2):
        assert rel_l2(g.cpu(), g2.cpu()) < 5e-5
    for n in grads:
        assert rel_l2(grads[n].cpu(), grads2[n].cpu()) < 5e-5, n


def test_up_down_nin_backward(hip_lib):
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import backward as bw, layers
    from oracle import unet_oracle as uo
    B, Cc, S = 8, 64, 8
    x = _randn((B, Cc, S, S, S), 6)
    for kind in ("up", "down"):
        lay = layers.Upsample(Cc, with_conv=True) if kind == "up" else layers.Downsample(Cc, with_conv=True)
        sd = _load(lay, 7)
        lay = lay.cuda().train()
        sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        y_ref = uo.upsample(sdr, xr) if kind == "up" else uo.downsample(sdr, xr)
        dy = _randn(tuple(y_ref.shape), 8)
        y_ref.backward(dy)
        tape = []
        with torch.no_grad():
            lay.forward_blocked(_f32b(ops, x), Cc, B, S ** 3, tape=tape)
            dx = lay.backward_blocked(tape[0], _f32b(ops, dy))
        assert rel_l2(ops.f32b_to_ncdhw(dx, (S, S, S)).cpu(), xr.grad) < TOL, kind
        assert rel_l2(lay.Conv_0.weight.grad.cpu(), sdr["Conv_0.weight"].grad) < TOL, kind
        assert rel_l2(lay.Conv_0.bias.grad.cpu(), sdr["Conv_0.bias"].grad) < TOL, kind
    nin = layers.NIN(Cc, 128)
    sd = _load(nin, 9)
    nin = nin.cuda()
    Wr, br, xr = sd["W"].clone().requires_grad_(True), sd["b"].clone().requires_grad_(True), x.clone().requires_grad_(True)
    y_ref = uo.nin(xr, Wr, br)
    dy = _randn(tuple(y_ref.shape), 11)
    y_ref.backward(dy)
    P = S ** 3
    with torch.no_grad():
        xs16 = bw.split_f32b(_f32b(ops, x), B, Cc, P)
        dx = bw.nin_backward(nin, _f32b(ops, dy), xs16, B, P, S)
    assert rel_l2(ops.f32b_to_ncdhw(dx, (S, S, S)).cpu(), xr.grad) < TOL
    assert rel_l2(nin.W.grad.cpu(), Wr.grad) < TOL and rel_l2(nin.b.grad.cpu(), br.grad) < TOL


@pytest.mark.parametrize("Cc,S,B", [(64, 8, 8), (64, 4, 8), (64, 8, 3)])
def test_attn_block_backward(hip_lib, Cc, S, B):
    from meshdiffusion_amd import hip_ops as ops
    from meshdiffusion_amd.lib.diffusion.models import layers
    from oracle import unet_oracle as uo
    blk = layers.AttnBlock(channels=Cc)
    sd = _load(blk, 4)
    blk = blk.cuda().train()
    x, dy = _randn((B, Cc, S, S, S), 5), _randn((B, Cc, S, S, S), 6)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    uo.attn_block(sdr, xr).backward(dy)
    tape = []
    with torch.no_grad():
        blk.forward_blocked(_f32b(ops, x), B, S ** 3, tape=tape)
        dx = blk.backward_blocked(tape[0], _f32b(ops, dy))
    assert rel_l2(ops.f32b_to_ncdhw(dx, (S, S, S)).cpu(), xr.grad) < TOL
    assert rel_l2(ops.f32b_to_ncdhw(dx, (S, S, S)).cpu() - dy, xr.grad - dy) < 5e-4    # the attention branch itself
    params = dict(blk.named_parameters())
    for n in ["NIN_0.W", "NIN_0.b", "NIN_1.W", "NIN_2.W", "NIN_2.b", "NIN_3.W", "NIN_3.b",
              "GroupNorm_0.weight", "GroupNorm_0.bias"]:
        assert rel_l2(params[n].grad.cpu(), sdr[n].grad) < 5e-4, n
    # the key bias cannot change a softmax over keys: its exact gradient is 0 and both sides hold only round-off
    scale = float(sdr["NIN_0.b"].grad.abs().max())
    assert float(params["NIN_1.b"].grad.abs().max()) < 1e-2 * scale + 1e-4
