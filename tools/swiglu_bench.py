import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev='cuda'; T, D, I = 16384, 2048, 8192
x = torch.randn(T, D, device=dev).bfloat16(); w13 = (torch.randn(2*I, D, device=dev)*0.02).bfloat16()
gu = torch.empty(T, 2*I, device=dev, dtype=torch.bfloat16); act = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
w2 = (torch.randn(D, I, device=dev)*0.02).bfloat16(); h = torch.empty(T, D, device=dev, dtype=torch.bfloat16)
def step():
    ops.gemm_swiglu_fwd(x, w13, gu, act)
    ops.gemm(0, act, w2, h, residual=x)
for _ in range(6): step()
s,e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): step()
e.record(); torch.cuda.synchronize()
print(os.environ.get('TAG','base'), f"swiglu-fwd GEMM + down GEMM: {s.elapsed_time(e)/10*1e3:.0f} us", flush=True)
