"""CU contention rehearsal: N single-workgroup spin kernels (torch.cuda._sleep on N streams) hold N CUs, as RCCL's channels do
during the gradient exchange, while a persistent GEMM runs; static vs dynamic tile order."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
dev = 'cuda'
T = 16384
def timed(fn, hogs, iters=5):
    streams = [torch.cuda.Stream() for _ in range(hogs)]
    for _ in range(2): fn()
    torch.cuda.synchronize()
    for st in streams:
        with torch.cuda.stream(st):
            torch.cuda._sleep(int(2.0e9 * 0.02))  # ~20 ms at 2 GHz
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for name, (M, N, K) in {'gateup 16 rounds': (T, 16384, 2048), 'dact 8 rounds': (T, 8192, 2048), 'wo 2 rounds': (T, 2048, 2048)}.items():
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); c = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: ops.gemm(0, a, b, c)
    row = []
    for dyn in (False, True):
        ops.set_gemm_tile_order(dyn)
        row.append((timed(fn, 0), timed(fn, 32)))
    ops.set_gemm_tile_order(False)
    print(f"{name:18s} static: alone {row[0][0]:6.0f} us, 32 CUs taken {row[0][1]:6.0f} us | dynamic: alone {row[1][0]:6.0f} us, 32 CUs taken {row[1][1]:6.0f} us", flush=True)
