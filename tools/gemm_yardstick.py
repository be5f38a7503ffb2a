"""Yardstick only (never used by the product): torch.matmul (hipBLASLt/rocBLAS) vs ssi_gemm on the step's GEMM shapes."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev = 'cuda'
T = 16384
SHAPES = [  # (name, layout, M, N, K)
    ("qkv NT", 0, T, 3072, 2048), ("wo NT", 0, T, 2048, 2048), ("gateup NT", 0, T, 16384, 2048), ("down NT", 0, T, 2048, 8192),
    ("dgu->dx NT K16384", 0, T, 2048, 16384), ("dact NT", 0, T, 8192, 2048), ("head NT", 0, 2048, 133258, 2048),
    ("dW13 TN", 2, 16384, 2048, T), ("dW2 TN", 2, 2048, 8192, T), ("dWqkv TN", 2, 3072, 2048, T), ("dWhead TN", 2, 133258, 2048, 2048),
]
def timeit(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for name, layout, M, N, K in SHAPES:
    a = torch.randn((M, K) if layout < 2 else (K, M), device=dev).bfloat16()
    b = torch.randn((N, K) if layout == 0 else (K, N), device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    if layout == 0: tfn = lambda: torch.matmul(a, b.t(), out=c)
    else: tfn = lambda: torch.matmul(a.t(), b, out=c)
    t_lib = timeit(tfn)
    t_own = timeit(lambda: ops.gemm(layout, a, b, c))
    f = 2 * M * N * K / 1e9
    print(f"{name:22s} M={M:6d} N={N:6d} K={K:5d}  hipBLASLt {f / t_lib:7.0f} TF/s ({t_lib * 1e3:7.0f} us)   ssi_gemm {f / t_own:7.0f} TF/s ({t_own * 1e3:7.0f} us)", flush=True)
