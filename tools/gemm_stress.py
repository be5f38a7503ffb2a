"""Race screen of the persistent GEMM's LDS-DMA loop: many random shapes and all three operand layouts on small-integer operands (exact in
fp32), each launch checked bit for bit against torch's fp32 matmul, with a second stream streaming HBM traffic to perturb the timing of the
DMA pieces.  python tools/gemm_stress.py [iterations]"""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import _lib, ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device='cuda').manual_seed(1234)
noise_a = torch.empty(256 << 20, dtype=torch.uint8, device='cuda'); noise_b = torch.empty_like(noise_a)
side = torch.cuda.Stream()
ops.set_impl(_lib.IMPL_MFMA)
bad = 0
for it in range(iters):
    layout = it % 3
    M = 256 * int(torch.randint(1, 24, (1,), generator=g, device='cuda'))
    N = 256 * int(torch.randint(1, 24, (1,), generator=g, device='cuda'))
    K = 128 * int(torch.randint(2, 40, (1,), generator=g, device='cuda'))
    p = min(0.5, (2048.0 / K) ** 0.5)
    def ints(*shape):
        sign = torch.randint(0, 2, shape, generator=g, device='cuda', dtype=torch.int8) * 2 - 1
        return (sign * (torch.rand(shape, generator=g, device='cuda') < p)).to(torch.bfloat16)
    a = ints(M, K) if layout < 2 else ints(K, M)
    b = ints(N, K) if layout == 0 else ints(K, N)
    ref = (a.float() @ b.float().t()) if layout == 0 else ((a.float() @ b.float()) if layout == 1 else (a.float().t() @ b.float()))
    if it % 2:
        with torch.cuda.stream(side):
            noise_b.copy_(noise_a)
    mode = (it // 3) % 3
    c0 = torch.randint(-2, 3, (M, N), generator=g, device='cuda').to(torch.bfloat16)
    c = c0.clone() if mode else torch.full((M, N), float('nan'), dtype=torch.bfloat16, device='cuda')
    kw = {"accumulate": True} if mode == 1 else ({"residual": c0} if mode == 2 else {})
    ops.gemm(layout, a, b, c, **kw)
    want = ref.bfloat16().float() + (c0.float() if mode else 0)
    ok = (ref.abs() <= 256) & (want == want.bfloat16().float())
    good = bool(((c.float() == want) | ~ok).all())
    if not good:
        bad += 1
        print("MISMATCH", it, layout, M, N, K, mode, int(((c.float() != want) & ok).sum()), flush=True)
    if it % 50 == 49:
        print(f"{it + 1} launches, {bad} mismatches", flush=True)
torch.cuda.synchronize()
print("stress done:", iters, "launches,", bad, "mismatches")
sys.exit(1 if bad else 0)
