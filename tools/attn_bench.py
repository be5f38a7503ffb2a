"""In-run A/B of the attention kernels (set SSI_HIP_LIB to pick the library): B=8, S=2048, H=32, KV=8, hd=64 as in the step."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
B, S, H, KV, hd = 8, 2048, 32, 8, 64
T = B * S
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
def t(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for rep in range(2):
    f = t(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd))
    b = t(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
    print(os.environ.get('TAG', ''), f"fwd {f:.0f} us  bwd {b:.0f} us", flush=True)
