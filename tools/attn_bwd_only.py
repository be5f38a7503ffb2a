"""Attention forward + backward alone (for PMC passes): B=8, S=2048, H=32, KV=8, hd=64."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
B, S, H, KV, hd = 8, 2048, 32, 8, 64
T = B * S
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
for _ in range(4):
    ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd)
torch.cuda.synchronize()
print("done")
