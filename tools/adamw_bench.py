"""AdamW kernel alone on the model's 1.246 G parameters (bf16 p / g / m / v: 17.4 GB per call).  python tools/adamw_bench.py"""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
n = 1_246_058_496
p, g, m, v = [(torch.randn(n, device='cuda') * 0.01).to(torch.bfloat16) for _ in range(4)]
v.abs_()
fn = lambda: ops.adamw_step(p, g, m, v, lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=3, grad_scale_dev=None, zero_grad=False)
for _ in range(3): fn()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): fn()
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / 10
print(f"AdamW {n / 1e9:.3f} G elements: {ms:.3f} ms  {7 * 2 * n / ms / 1e9:.2f} TB/s", flush=True)
