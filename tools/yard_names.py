"""Yardstick only: which hipBLASLt kernels torch.matmul picks for the step's GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
T = 16384
for (M, N, K, tn) in [(T, 16384, 2048, False), (T, 2048, 8192, False), (T, 3072, 2048, False), (T, 2048, 16384, False), (16384, 2048, T, True), (2048, 8192, T, True)]:
    a = torch.randn((M, K) if not tn else (K, M), device='cuda').bfloat16()
    b = torch.randn((N, K) if not tn else (K, N), device='cuda').bfloat16()
    for _ in range(3):
        c = torch.matmul(a, b.t()) if not tn else torch.matmul(a.t(), b)
    torch.cuda.synchronize()
