"""Time the step's GEMM shapes through ops.gemm (set SSI_HIP_LIB to compare builds).  TAG=name python tools/gemm_ab.py"""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
shapes = [(0, 16384, 3072, 2048, "qkv fwd NT"), (0, 16384, 2048, 8192, "down fwd NT"), (1, 16384, 2048, 16384, "dgrad w13 NN"),
          (2, 16384, 2048, 16384, "wgrad w13 TN"), (2, 2048, 8192, 16384, "wgrad w2 TN"), (0, 16384, 16384, 2048, "gate-up NT plain")]
def t(fn, iters=20):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for layout, M, N, K, name in shapes:
    a = torch.randn((M, K) if layout < 2 else (K, M), device='cuda').bfloat16()
    b = torch.randn((N, K) if layout == 0 else (K, N), device='cuda').bfloat16()
    c = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16)
    ms = t(lambda: ops.gemm(layout, a, b, c))
    print(os.environ.get("TAG", ""), f"{name:18s} {M}x{N}x{K}: {ms * 1e3:8.1f} us  {2 * M * N * K / ms / 1e9:7.1f} TF/s", flush=True)
