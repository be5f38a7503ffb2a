import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev='cuda'
def bench(M,N,K, padc=0, pada=0, padb=0, iters=10):
    a = torch.randn(M, K + pada, device=dev).bfloat16()[:, :K]
    b = torch.randn(N, K + padb, device=dev).bfloat16()[:, :K]
    c = torch.zeros(M, N + padc, device=dev, dtype=torch.bfloat16)[:, :N]
    for _ in range(3): ops.gemm(0,a,b,c)
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm(0,a,b,c)
    e.record(); torch.cuda.synchronize()
    return 2*M*N*K/(s.elapsed_time(e)/iters)/1e9
T=16384
for rep in range(2):
    print('gateup N=16384: plain', f"{bench(T,16384,2048):.0f}", 'ldc+64', f"{bench(T,16384,2048,padc=64):.0f}", 'ldc+256', f"{bench(T,16384,2048,padc=256):.0f}", 'lda/b+64', f"{bench(T,16384,2048,pada=64,padb=64):.0f}", 'all+64', f"{bench(T,16384,2048,64,64,64):.0f}", flush=True)
    print('dact N=8192: plain', f"{bench(T,8192,2048):.0f}", 'ldc+64', f"{bench(T,8192,2048,padc=64):.0f}", 'all+64', f"{bench(T,8192,2048,64,64,64):.0f}", '| qkv plain', f"{bench(T,3072,2048):.0f}", 'all+64', f"{bench(T,3072,2048,64,64,64):.0f}", '| wo N=2048 plain', f"{bench(T,2048,2048):.0f}", 'all+64', f"{bench(T,2048,2048,64,64,64):.0f}", flush=True)
