"""Launch one GEMM shape repeatedly (for rocprofv3 PMC / kernel-trace runs): python tools/gemm_loop.py LAYOUT M N K [iters]."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
layout, M, N, K = (int(x) for x in sys.argv[1:5]); iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
a = torch.randn((M, K) if layout < 2 else (K, M), device='cuda').bfloat16()
b = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16()
c = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
for _ in range(iters): ops.gemm(layout, a, b, c)
torch.cuda.synchronize()
