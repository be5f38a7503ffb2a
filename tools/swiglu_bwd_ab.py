"""Fused down-projection data gradient + SwiGLU backward (SURVEY.md §2.3 K7 backward) at the step's shape, in one process: the persistent
4-wave kernel (one wave per SIMD, 256 accumulator registers: the 17 us epilogue runs with nothing beside it) against the round-1 8-wave
kernel (two waves per SIMD, 128 accumulators each: slower main loop, but the epilogue's vector work is split over twice the waves and two
waves interleave their issue), both on the k-contiguous form (transposed W2 copy), and the persistent kernel's NN form the step uses."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import _lib, ops
T, D, I = 16384, 2048, 8192
torch.manual_seed(0)
dy = torch.randn(T, D, device='cuda').bfloat16()
w2 = (torch.randn(D, I, device='cuda') * 0.02).bfloat16()     # [K, I]
w2t = w2.t().contiguous()                                       # [I, K]
gu = torch.randn(T, 2 * I, device='cuda').bfloat16()
dgu = torch.empty_like(gu)
dact = torch.empty(T, I, device='cuda', dtype=torch.bfloat16)
def t(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
fl = 2.0 * T * D * I
res = {}
for rep in range(2):
    for name, impl, layout, w in (("4-wave NN (step)", _lib.IMPL_MFMA, ops.GEMM_NN, w2), ("4-wave NT", _lib.IMPL_MFMA, ops.GEMM_NT, w2t),
                                  ("8-wave NT", _lib.IMPL_MFMA_WG8, ops.GEMM_NT, w2t)):
        prev = ops.set_impl(impl)
        try:
            us = t(lambda: ops.gemm_swiglu_bwd(layout, dy, w, gu, dgu, dact))
            res.setdefault(name, dgu.clone())
        finally:
            ops.set_impl(prev)
        print(f"{name}: {us:.0f} us = {fl / us / 1e6:.0f} TFLOP/s", flush=True)
print("8-wave == 4-wave NT:", torch.equal(res["8-wave NT"], res["4-wave NT"]), " NN == NT:", torch.equal(res["4-wave NN (step)"], res["4-wave NT"]))
