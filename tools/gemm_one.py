import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev='cuda'
M,N,K = 16384, 2048, 8192
for layout in (0,1,2):
    a = torch.randn((M,K) if layout<2 else (K,M), device=dev).bfloat16()
    b = torch.randn((N,K) if layout==0 else (K,N), device=dev).bfloat16()
    c = torch.empty(M,N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(layout,a,b,c)
torch.cuda.synchronize()
