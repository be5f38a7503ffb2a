"""Does an HBM-bound kernel (AdamW over a slice of the flat buffers) run UNDER the persistent MFMA GEMMs / the attention backward when it is launched
on a second stream?  Times each alone and both together.  python tools/overlap_probe.py"""
import sys, time, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops

dev, bf = 'cuda', torch.bfloat16
T, D, I = 16384, 2048, 8192
gu = (torch.randn(T, 2 * I, device=dev) * 0.1).to(bf)
w13 = (torch.randn(2 * I, D, device=dev) * 0.02).to(bf)
dx = torch.empty(T, D, device=dev, dtype=bf)
xn = (torch.randn(T, D, device=dev) * 0.5).to(bf)
dw = torch.empty(2 * I, D, device=dev, dtype=bf)
n = 480_000_000  # eight layers' parameters: ~1.2 ms of AdamW against several compute kernels per repetition
p, g, m, v = [(torch.randn(n, device=dev) * 0.01).to(bf) for _ in range(4)]
v.abs_()
B, S, H, KV, hd = 8, 2048, 32, 8, 64
qkv = (torch.randn(B * S, (H + 2 * KV) * hd, device=dev) * 0.5).to(bf)
att = torch.empty(B * S, H * hd, device=dev, dtype=bf)
lse = torch.empty(B * H * S, device=dev, dtype=torch.float32)
datt = (torch.randn(B * S, H * hd, device=dev) * 0.1).to(bf)
dqkv = torch.empty_like(qkv)
delta = torch.empty_like(lse)
ops.attn_fwd(qkv, att, lse, B, S, H, KV, hd, None, None)

work = {
    "NN dgrad 16384x2048x16384": lambda: ops.gemm(ops.GEMM_NN, gu, w13, dx),
    "TN wgrad 16384x2048x16384": lambda: ops.gemm(ops.GEMM_TN, gu, xn, dw),
    "attention backward": lambda: ops.attn_bwd(qkv, att, datt, lse, dqkv, delta, B, S, H, KV, hd, None, None),
}
adam = lambda: ops.adamw_step(p, g, m, v, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, grad_scale_dev=None, zero_grad=False)
side = torch.cuda.Stream()


def wall(fa, fb, reps=10, inner=4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if fb:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fb()
        if fa:
            for _ in range(inner): fa()
        if fb:
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for _ in range(3): adam()
tb = wall(None, adam)
print(f"AdamW {n / 1e6:.0f} M elements alone: {tb:.3f} ms ({7 * 2 * n / tb / 1e9:.2f} TB/s)")
for name, fa in work.items():
    for _ in range(3): fa()
    ta = wall(fa, None)
    tab = wall(fa, adam)
    print(f"{name:28s} alone {ta:.3f} ms | + AdamW on a second stream {tab:.3f} ms | serial sum {ta + tb:.3f} ms | hidden {100 * (ta + tb - tab) / tb:.0f} % of AdamW", flush=True)
