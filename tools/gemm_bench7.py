import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev='cuda'
def bench(layout, M,N,K, iters=10, acc=False):
    a = torch.randn((M,K) if layout<2 else (K,M), device=dev).bfloat16()
    b = torch.randn((N,K) if layout==0 else (K,N), device=dev).bfloat16()
    c = torch.zeros(M,N, device=dev, dtype=torch.bfloat16)
    for _ in range(6): ops.gemm(layout,a,b,c,accumulate=acc)
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm(layout,a,b,c,accumulate=acc)
    e.record(); torch.cuda.synchronize()
    return 2*M*N*K/(s.elapsed_time(e)/iters)/1e9
T=16384
bench(0,T,2048,2048); bench(0,T,2048,2048)
print(os.environ.get('TAG','base'), 'gateup', f"{bench(0,T,16384,2048):.0f}", 'head', f"{bench(0,T,133376,2048,iters=4):.0f}", 'dact', f"{bench(0,T,8192,2048):.0f}", 'qkv', f"{bench(0,T,3072,2048):.0f}", 'TN dW13', f"{bench(2,16384,2048,T,acc=True):.0f}", flush=True)
