"""Numerics of the Winograd F(2,3)-along-w convolution in bf16x3 arithmetic against fp64, on the CPU (PyTorch): the error of
one 128 -> 128 3x3x3 conv on SiLU(GroupNorm)-like inputs for the direct bf16x3 form, the Winograd bf16x3 form, and the
Winograd form in fp32 (the transform's own error).  DESIGN.md section 3 quotes its output.

    python tools/wino_numerics.py
"""
import torch, torch.nn.functional as F
torch.manual_seed(0)
torch.set_num_threads(32)
def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi, lo
def conv3x(x, w, pad):
    xh, xl = split(x); wh, wl = split(w)
    return F.conv3d(xh, wh, padding=pad) + F.conv3d(xh, wl, padding=pad) + F.conv3d(xl, wh, padding=pad)
B, Ci, Co, S = 1, 128, 128, 16
x = torch.randn(B, Ci, S, S, S)
x = F.silu(x * 1.5 + 0.3)     # post GN+SiLU like
w = torch.randn(Co, Ci, 3, 3, 3) * (1.0 / (27 * Ci) ** 0.5)
ref = F.conv3d(x.double(), w.double(), padding=1)
d = conv3x(x, w, 1)
print("direct bf16x3 rel", float((d.double() - ref).norm() / ref.norm()))
# winograd F(2,3) along w
xp = F.pad(x, (1, 1, 1, 1, 1, 1))            # [B,C,S+2,S+2,S+2]
Wd = S + 2
# pairs: outputs (2i, 2i+1) use inputs d0..d3 = xp[..., 2i : 2i+4]
d0 = xp[..., 0:Wd - 3:2]; d1 = xp[..., 1:Wd - 2:2]; d2 = xp[..., 2:Wd - 1:2]; d3 = xp[..., 3:Wd:2]
T = [d0 - d2, d1 + d2, d2 - d1, d1 - d3]     # each [B,C,S+2,S+2,S/2]
g0, g1, g2 = w[..., 0], w[..., 1], w[..., 2]  # [Co,Ci,3,3]
G = [g0, (g0 + g1 + g2) * 0.5, (g0 - g1 + g2) * 0.5, g2]
m = []
for f in range(4):
    th, tl = split(T[f]); gh, gl = split(G[f])
    k = lambda a, b: F.conv3d(a, b[..., None], padding=0)
    m.append(k(th, gh) + k(th, gl) + k(tl, gh))
y0 = m[0] + m[1] + m[2]; y1 = m[1] - m[2] - m[3]
y = torch.stack([y0, y1], -1).reshape(B, Co, S, S, S)
print("wino-w bf16x3 rel", float((y.double() - ref).norm() / ref.norm()))
# fp32 winograd (no split) for the algorithm's own error
m = [F.conv3d(T[f], G[f][..., None]) for f in range(4)]
y = torch.stack([m[0] + m[1] + m[2], m[1] - m[2] - m[3]], -1).reshape(B, Co, S, S, S)
print("wino-w fp32 rel", float((y.double() - ref).norm() / ref.norm()))
d32 = F.conv3d(x, w, padding=1)
print("direct fp32 rel", float((d32.double() - ref).norm() / ref.norm()))

# ---- F(4,3) along w (6 products per 4 outputs): the same measurement, for the next-round note in DESIGN.md section 8 ----
BT = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float32)
Gm = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                   [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float32)
nq = S // 4
dq = torch.stack([xp[..., k:k + 4 * (nq - 1) + 1:4] for k in range(6)], -1)          # [B,C,S+2,S+2,S/4,6]: inputs 4i-1 .. 4i+4
T6 = torch.einsum("fk,...k->...f", BT, dq)                                            # transformed inputs
G6 = torch.einsum("fk,oihwk->oihwf", Gm, w)                                           # transformed weights [Co,Ci,3,3,6]
for name, sp in (("bf16x3", True), ("fp32", False)):
    ms = []
    for f in range(6):
        t, g = T6[..., f], G6[..., f][..., None]
        if sp:
            th, tl = split(t); gh, gl = split(g)
            ms.append(F.conv3d(th, gh) + F.conv3d(th, gl) + F.conv3d(tl, gh))
        else:
            ms.append(F.conv3d(t, g))
    M = torch.stack(ms, -1)                                                            # [B,Co,S,S,S/4,6]
    y4 = torch.einsum("jf,...f->...j", AT, M).reshape(B, Co, S, S, S)
    print(f"wino-w F(4,3) {name} rel", float((y4.double() - ref).norm() / ref.norm()))
