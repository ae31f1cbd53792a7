import sys, os, torch, time
sys.path.insert(0, os.getcwd())
from meshdiffusion_amd import hip_ops as ops
B, C, N = 8, 256, 4096
g = torch.Generator().manual_seed(1)
qk = ops.s16b_empty(B, 2 * C, N, torch.device("cuda")); vT = ops.s16b_empty(B, N, C, torch.device("cuda"))
qk.view(torch.int16).random_(-3000, 3000); vT.view(torch.int16).random_(-3000, 3000)
bias = torch.zeros(C, device="cuda")
for _ in range(3): o = ops.attn_fwd(qk, vT, bias, B, C, N, ops.attn_scale(C))
torch.cuda.synchronize()
ts = []
for _ in range(7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); o = ops.attn_fwd(qk, vT, bias, B, C, N, ops.attn_scale(C)); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
print("md_attn_fwd ms", sorted(ts)[3], "checksum", float(o.float().abs().sum()))
