"""Yardstick only (never in the product path): PyTorch's own scaled_dot_product_attention on this GPU (the flash kernels torch ships for gfx950)
at the step's attention shape, beside this repo's kernels.  B=8, S=2048, 32 query / 8 kv heads, head_dim 64, causal, bf16.
python tools/attn_yardstick.py [S]"""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops

B, H, KV, hd = 8, 32, 8, 64
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
if S > 2048: B = max(1, B * 2048 // S)
T = B * S
dev = 'cuda'


def t(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


qkv = torch.randn(T, (H + 2 * KV) * hd, device=dev).bfloat16()
out = torch.empty(T, H * hd, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device=dev, dtype=torch.float32)
dout = torch.randn(T, H * hd, device=dev).bfloat16()
dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
f_own = t(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd))
b_own = t(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd))
flop_f = 4.0 * B * H * S * S * hd / 2
flop_b = 10.0 * B * H * S * S * hd / 2
print(f"B={B} S={S}: this repo   fwd {f_own:7.0f} us ({flop_f / f_own / 1e6:5.0f} TF/s)   bwd {b_own:7.0f} us ({flop_b / b_own / 1e6:5.0f} TF/s on the 5-product count)")

q = torch.randn(B, H, S, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, KV, S, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, KV, S, hd, device=dev, dtype=torch.bfloat16, requires_grad=True)
do = torch.randn(B, H, S, hd, device=dev, dtype=torch.bfloat16)
for name, kwargs, expand in (("torch SDPA enable_gqa", dict(enable_gqa=True), False), ("torch SDPA, kv heads expanded", {}, True)):
    try:
        kk, vv = (k.repeat_interleave(H // KV, 1), v.repeat_interleave(H // KV, 1)) if expand else (k, v)
        with torch.no_grad():
            f_t = t(lambda: F.scaled_dot_product_attention(q, kk, vv, is_causal=True, **kwargs))
        o = F.scaled_dot_product_attention(q, kk, vv, is_causal=True, **kwargs)

        def bwd():
            q.grad = k.grad = v.grad = None
            o.backward(do, retain_graph=True)
        b_t = t(bwd)
        print(f"B={B} S={S}: {name:30s} fwd {f_t:7.0f} us ({flop_f / f_t / 1e6:5.0f} TF/s)   bwd {b_t:7.0f} us ({flop_b / b_t / 1e6:5.0f} TF/s)")
    except Exception as e:
        print(f"{name}: not available here ({type(e).__name__}: {str(e)[:120]})")
