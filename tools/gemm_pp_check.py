"""Prototype of the persistent NT GEMM with 256 x 128 tiles (SSI_GEMM_PP=1: epilogue behind the K-loop; =2: two accumulator sets, the
finished one stored under the next tile's K-steps) against gemm_nt4dma_kernel in ONE process: bit-identical results, time."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
shapes = [(16384, 16384, 2048, "gate-up plain"), (16384, 2048, 8192, "down"), (16384, 3072, 2048, "qkv"), (4096, 16384, 2048, "gate-up T=4096"),
          (16384, 133376, 2048, "LM head")]
modes = [m for m in sys.argv[1:] if m in ('1', '2')] or ['1']
def t(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for M, N, K, name in shapes:
    torch.manual_seed(1)
    a = torch.randn(M, K, device='cuda').bfloat16()
    b = (torch.randn(N, K, device='cuda') * 0.05).bfloat16()
    out, tm = {}, {}
    for mode in ['0'] + modes + ['0'] + modes:
        if mode == '0': os.environ.pop('SSI_GEMM_PP', None)
        else: os.environ['SSI_GEMM_PP'] = mode
        c = torch.full((M, N), float('nan'), device='cuda', dtype=torch.bfloat16)
        ops.gemm(ops.GEMM_NT, a, b, c)
        torch.cuda.synchronize()
        out[mode] = c
        tm.setdefault(mode, []).append(t(lambda: ops.gemm(ops.GEMM_NT, a, b, c)))
    line = f'{name:16s} {M}x{N}x{K}: nt4dma ' + '/'.join(f'{x:.0f}' for x in tm['0'])
    for mode in modes:
        same = torch.equal(out['0'], out[mode])
        line += f'   PP={mode} ' + '/'.join(f'{x:.0f}' for x in tm[mode]) + f' us identical {same}'
        if not same:
            d = (out['0'].float() - out[mode].float())
            line += f' (nan {int(torch.isnan(out[mode]).sum())}, differing {int((d != 0).sum())})'
    print(line, flush=True)
    del a, b, out
