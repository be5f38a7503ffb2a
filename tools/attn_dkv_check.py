"""dK / dV kernels side by side in ONE process (ssi_set_attn_impl switches the kernel between calls): the round-4 pipelined kernel against the round-1..3 kernel —
agreement, run-to-run reproducibility, time.  B=8, S=2048, H=32, KV=8, hd=64 as in the step; `packed` adds documents of 440-1100 tokens."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
B, S, H, KV, hd = (int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else (8, 2048, 32, 8, 64)))
packed = 'packed' in sys.argv
T = B * S
torch.manual_seed(0)
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
delta = torch.empty_like(lse)
kw = {}
if packed:
    g = torch.Generator().manual_seed(1)
    pos = []
    for _ in range(B):
        left, row = S, []
        while left > 0:
            n = min(left, int(torch.randint(440, 1101, (1,), generator=g)))
            row.append(torch.arange(n)); left -= n
        pos.append(torch.cat(row))
    input_pos = torch.stack(pos).to('cuda', torch.int64)
    _, ds, de = ops.doc_ranges(input_pos, 1 << 20)
    kw = dict(doc_start=ds, doc_end=de)
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd, **kw)
def bwd(sel):
    ops.set_attn_impl(1, int(sel))
    d = torch.zeros_like(qkv)
    ops.attn_bwd(qkv, out, dout, lse, d, delta, B, S, H, KV, hd, **kw)
    torch.cuda.synchronize()
    return d
old, new, new2 = bwd('1'), bwd('2'), bwd('2')
kv = slice(H * hd, None)
print('reproducible:', bool((new == new2).all()), ' q block identical:', bool((old[:, :H * hd] == new[:, :H * hd]).all()))
a, b_ = old[:, kv].float(), new[:, kv].float()
print(f'dK/dV new vs old: max abs {float((a - b_).abs().max()):.3e}  rel fro {float((a - b_).norm() / a.norm()):.3e}  '
      f'nan {int(torch.isnan(b_).sum())}  |old| {float(a.norm()):.3e}')
def t(sel, iters=20):
    ops.set_attn_impl(1, int(sel))
    d = torch.empty_like(qkv)
    f = lambda: ops.attn_bwd(qkv, out, dout, lse, d, delta, B, S, H, KV, hd, **kw)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for rep in range(3):
    print(f'bwd (dQ + dK/dV) us: old {t("1"):.0f}  new {t("2"):.0f}  default {t("0"):.0f}', flush=True)
if 'stamps' in sys.argv:  # SSI_HIP_LIB = a -DDKV2_STAMP build (tools/variant.sh stamp2 attention_mfma.hip -DDKV2_STAMP)
    d = bwd('2')
    names = ['A.dkv (sync tile)', 'A.sp+barrier+tr', 'B.dkv+rows', 'B.sp+dma', 'A.dkv (plain tile)', 'A.sp+tr', 'B.dkv+rows', 'B.sp']
    for g in range(S // 256):
        v = d[g * 256].view(torch.float32)[:11].tolist()  # batch 0, kv head 0
        tiles = v[9]
        print(f'kgrp {int(v[10])}: {int(tiles)} tiles, total {v[8]:.0f} cyc = {v[8] / tiles:.0f} per tile; per half-period: ' +
              ', '.join(f'{n} {x / (tiles / 2):.0f}' for n, x in zip(names, v[:8])))
if 'gaps' in sys.argv:  # SSI_HIP_LIB = a -DDKV2_STAMP_GAPS build: cycles per gap of the tile without barrier / requests
    d = bwd('0')
    v = d[0].view(torch.float32)[:33].tolist()
    n = v[32] / 2
    print('period A gaps:', ' '.join(f'{x / n:.0f}' for x in v[:16]))
    print('period B gaps:', ' '.join(f'{x / n:.0f}' for x in v[16:32]))
