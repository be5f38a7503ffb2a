import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops, _lib
dev='cuda'
def bench(layout, M,N,K, iters=20):
    a = torch.randn((M,K) if layout<2 else (K,M), device=dev).bfloat16()
    b = torch.randn((N,K) if layout==0 else (K,N), device=dev).bfloat16()
    c = torch.empty(M,N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): ops.gemm(layout,a,b,c)
    s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): ops.gemm(layout,a,b,c)
    e.record(); torch.cuda.synchronize()
    return 2*M*N*K/(s.elapsed_time(e)/iters)/1e9
for impl,name in ((2,'glds'),(3,'regstage')):
    ops.set_impl(impl)
    for (M,N,K) in [(16384,2048,8192),(16384,3072,2048)]:
        print('v1', name, (M,N,K), ' '.join(f"{['NT','NN','TN'][l]}={bench(l,M,N,K):.0f}" for l in range(3)), flush=True)
