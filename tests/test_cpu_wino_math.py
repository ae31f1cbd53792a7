"""CPU check of the algebra behind the Winograd path (no GPU, no library): the operand transform md_wino_prep specifies
(d0-d2, d1+d2, d2-d1, d1-d3 of the zero-padded input along w), the weight transform md_wino_pack_weights specifies
(g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2 along kw) and the output transform of md_conv3_wino's epilogue (y0 = m0+m1+m2,
y1 = m1-m2-m3) reproduce nn.Conv3d 3x3x3 pad 1 (lib/diffusion/models/layers.py:118-124); and the data-gradient convolution
uses W'[ci][co][t] = W[co][ci][26 - t] (what `flip = 1` packs in place)."""
import torch
import torch.nn.functional as F


def _rand(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=torch.float64)


def wino_conv_reference(x, w):
    """x [B,Ci,D,H,W] (W even), w [Co,Ci,3,3,3] -> conv3d pad 1 through the F(2,3)-along-w decomposition, float64."""
    B, Ci, D, H, W = x.shape
    xp = F.pad(x, (1, 1, 1, 1, 1, 1))
    d = [xp[..., k:k + W:2] for k in range(4)]                               # d_k = x[2i - 1 + k], each [B,Ci,D+2,H+2,W/2]
    t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]                 # md_wino_prep
    g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]
    g = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]                 # md_wino_pack_weights
    m = [F.conv3d(t[f], g[f][..., None]) for f in range(4)]                  # 9 (kd, kh) taps per frequency: md_conv3_wino
    y0, y1 = m[0] + m[1] + m[2], m[1] - m[2] - m[3]                          # its epilogue
    return torch.stack([y0, y1], -1).reshape(B, w.shape[0], D, H, W)


def test_f23_along_w_equals_conv3d():
    x, w = _rand((2, 5, 4, 6, 8), 0), _rand((7, 5, 3, 3, 3), 1)
    ref = F.conv3d(x, w, padding=1)
    got = wino_conv_reference(x, w)
    assert float((got - ref).abs().max()) < 1e-12


def test_data_gradient_weights_are_the_flipped_transpose():
    ci, co = 5, 7
    w, dy = _rand((co, ci, 3, 3, 3), 2), _rand((2, co, 4, 6, 8), 3)
    ref = torch.nn.grad.conv3d_input((2, ci, 4, 6, 8), w, dy, padding=1)
    wd = w.reshape(co, ci, 27).flip(2).transpose(0, 1).reshape(ci, co, 3, 3, 3)    # W'[ci][co][t] = W[co][ci][26 - t]
    assert float((F.conv3d(dy, wd, padding=1) - ref).abs().max()) < 1e-12
    assert float((wino_conv_reference(dy, wd) - ref).abs().max()) < 1e-12


def test_weight_gradient_in_the_winograd_domain_equals_autograd():
    """The transposed minimal algorithm md_wgrad_wino specifies: with t = B^T d (the forward operand T) and
    u = (dy0, dy0 + dy1, dy0 - dy1, dy1) (md_wino_prep_dual's second output), n_f[kd][kh] = sum over samples, rows and pairs of
    u_f * t_f shifted by (kd - 1, kh - 1) rows, and dg0 = n0 + (n1 + n2)/2, dg1 = (n1 - n2)/2, dg2 = (n1 + n2)/2 - n3
    (md_wgrad_wino_reduce) -- against autograd of nn.Conv3d in float64."""
    B, ci, co, D, H, W = 2, 3, 4, 4, 5, 8
    a, dy = _rand((B, ci, D, H, W), 5), _rand((B, co, D, H, W), 6)
    ref = torch.nn.grad.conv3d_weight(a, (co, ci, 3, 3, 3), dy, padding=1)
    ap = F.pad(a, (1, 1, 1, 1, 1, 1))
    d = [ap[..., k:k + W:2] for k in range(4)]
    t = [d[0] - d[2], d[1] + d[2], d[2] - d[1], d[1] - d[3]]                 # [B,ci,D+2,H+2,W/2] (rows -1 / D, H are zeros)
    dy0, dy1 = dy[..., 0::2], dy[..., 1::2]
    u = [dy0, dy0 + dy1, dy0 - dy1, dy1]
    n = torch.zeros((4, co, ci, 3, 3), dtype=torch.float64)
    for f in range(4):
        for kd in range(3):
            for kh in range(3):
                tt = t[f][:, :, kd:kd + D, kh:kh + H]                          # T rows (z + kd - 1, y + kh - 1)
                n[f, :, :, kd, kh] = torch.einsum("bozyp,bizyp->oi", u[f], tt)
    h12 = 0.5 * (n[1] + n[2])
    got = torch.stack([n[0] + h12, 0.5 * (n[1] - n[2]), h12 - n[3]], -1)      # [co,ci,kd,kh,kw]
    assert float((got - ref).abs().max()) < 1e-11
