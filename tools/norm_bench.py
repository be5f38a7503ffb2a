"""RMSNorm forward / backward alone at the step's shape (T = 16384 rows, D = 2048, bf16).  Set SSI_HIP_LIB to compare libraries."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
T, D = 16384, 2048
x = torch.randn(T, D, device='cuda').bfloat16(); dy = torch.randn(T, D, device='cuda').bfloat16(); dres = torch.randn(T, D, device='cuda').bfloat16()
w = torch.ones(D, device='cuda').bfloat16(); y = torch.empty_like(x); dx = torch.empty_like(x)
rstd = torch.empty(T, device='cuda'); dscale = torch.zeros(D, device='cuda').bfloat16()
ws = torch.empty(ops.rmsnorm_bwd_workspace_bytes(T, D), dtype=torch.uint8, device='cuda')
def t(fn, iters=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for rep in range(2):
    f = t(lambda: ops.rmsnorm_fwd(x, w, y, rstd, 1e-5))
    b = t(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, dres, dx, dscale, ws))
    print(os.environ.get('TAG', ''), f"fwd {f:.1f} us ({2 * T * D * 2 / f / 1e6:.2f} TB/s)  bwd+colsum {b:.1f} us ({4 * T * D * 2 / b / 1e6:.2f} TB/s)", flush=True)
print("checksum", float(dx.float().sum()), float(dscale.float().sum()))
