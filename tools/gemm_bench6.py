import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev='cuda'
def bench(layout, M,N,K, iters=20, acc=False, res=False):
    a = torch.randn((M,K) if layout<2 else (K,M), device=dev).bfloat16()
    b = torch.randn((N,K) if layout==0 else (K,N), device=dev).bfloat16()
    c = torch.zeros(M,N, device=dev, dtype=torch.bfloat16)
    r = torch.randn(M,N, device=dev).bfloat16() if res else None
    for _ in range(10): ops.gemm(layout,a,b,c,accumulate=acc,residual=r)
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm(layout,a,b,c,accumulate=acc,residual=r)
    e.record(); torch.cuda.synchronize()
    return 2*M*N*K/(s.elapsed_time(e)/iters)/1e9
T=16384
bench(0,T,2048,2048)
print(os.environ.get('TAG','base'), 'qkv', f"{bench(0,T,3072,2048):.0f}", 'wo+res', f"{bench(0,T,2048,2048,res=True):.0f}", 'dact', f"{bench(0,T,8192,2048):.0f}", 'gateup', f"{bench(0,T,16384,2048):.0f}", 'down', f"{bench(0,T,2048,8192,res=True):.0f}", 'NN dx', f"{bench(1,T,2048,16384):.0f}", 'TN dW2', f"{bench(2,2048,8192,T,acc=True):.0f}", flush=True)
