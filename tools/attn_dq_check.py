"""dQ kernels side by side in ONE process (ssi_set_attn_impl switches the kernel between calls): the round-4 pipelined kernel against the round-1..3 kernel —
agreement (and both against an fp32 torch reference on a slice), run-to-run reproducibility, time.  B=8, S=2048, H=32, KV=8, hd=64 as in
the step.  The dK / dV kernel is the same in both runs, so the difference of the two totals is the dQ kernels' difference."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
B, S, H, KV, hd = (int(x) for x in (sys.argv[1:6] if len(sys.argv) >= 6 else (8, 2048, 32, 8, 64)))
T = B * S
torch.manual_seed(0)
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
def bwd(sel):
    ops.set_attn_impl(0, int(sel))
    d = torch.zeros_like(qkv)
    delta = torch.zeros_like(lse)
    ops.attn_bwd(qkv, out, dout, lse, d, delta, B, S, H, KV, hd)
    torch.cuda.synchronize()
    return d, delta
(old, dl_old), (new, dl_new), (new2, _) = bwd('1'), bwd('0'), bwd('0')
q = slice(0, H * hd)
a, b_ = old[:, q].float(), new[:, q].float()
print('reproducible:', bool((new == new2).all()), ' delta identical:', bool((dl_old == dl_new).all()),
      ' k/v blocks identical:', bool((old[:, H * hd:] == new[:, H * hd:]).all()))
print(f'dQ new vs old: max abs {float((a - b_).abs().max()):.3e}  rel fro {float((a - b_).norm() / a.norm()):.3e}  '
      f'nan {int(torch.isnan(b_).sum())}  |old| {float(a.norm()):.3e}  differing elements {float((a != b_).float().mean()):.4f}')
# fp32 reference of dQ for batch 0, head 5 (kv head 1)
hh, kvh = 5, 5 // (H // KV)
Q = qkv[:S, hh * hd:(hh + 1) * hd].float(); K = qkv[:S, (H + kvh) * hd:(H + kvh + 1) * hd].float()
V = qkv[:S, (H + KV + kvh) * hd:(H + KV + kvh + 1) * hd].float(); dO = dout[:S, hh * hd:(hh + 1) * hd].float()
Q.requires_grad_(True)
s = (Q @ K.T) / 8.0
s = s.masked_fill(~torch.ones(S, S, dtype=torch.bool, device='cuda').tril(), float('-inf'))
((torch.softmax(s, -1) @ V) * dO).sum().backward()
for name, d in (('old', old), ('new', new)):
    x = d[:S, hh * hd:(hh + 1) * hd].float()
    print(f'dQ {name} vs fp32 torch (b 0, head {hh}): rel fro {float((x - Q.grad).norm() / Q.grad.norm()):.3e}')
def t(sel, iters=20):
    ops.set_attn_impl(0, int(sel))
    d = torch.empty_like(qkv); delta = torch.empty_like(lse)
    f = lambda: ops.attn_bwd(qkv, out, dout, lse, d, delta, B, S, H, KV, hd)
    for _ in range(3): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for rep in range(3):
    print(f'bwd (dQ + dK/dV) us: old dQ {t("1"):.0f}  new dQ {t("0"):.0f}', flush=True)
if 'stamps' in sys.argv:  # SSI_HIP_LIB = a -DDQ2_STAMP build (tools/variant.sh dq2stamp attention_mfma.hip -DDQ2_STAMP)
    d, _ = bwd('0')
    W = S // 512
    print('batch 0, kv head 0, wave 0 — cycles per phase, summed over the 8 items of a workgroup: wait for rows + tiles | operands | first tile, '
          'first S/dP | unmasked tiles | diagonal tile + drain | wait, barriers, table rows | store || total, 100 MHz ticks -> MHz')
    for g in range(W):
        v = d[g * 64].view(torch.float32)[:10].tolist()   # the group's last item is query block g
        tiles = sum(range(S // 64)) // W                       # unmasked tiles of a workgroup: (sum of j over its 8 blocks)
        print(f'group {int(v[9])}: ' + ' | '.join(f'{x:7.0f}' for x in v[:7]) + f' || {v[7]:8.0f}, {v[8]:6.0f} -> {v[7] / v[8] * 100:5.0f} MHz;'
              f'  per unmasked tile {v[3] / tiles:5.0f}')
if 'where' in sys.argv:  # where do the two kernels differ: by 64-query block, head, column block
    e = (a - b_).abs().view(B, S // 64, 64, H, hd)
    print('by q block:', [f'{float(x):.2g}' for x in e.amax(dim=(0, 2, 3, 4))])
    print('by head   :', [f'{float(x):.2g}' for x in e.amax(dim=(0, 1, 2, 4))])
    print('by row%64 :', [f'{float(x):.2g}' for x in e.amax(dim=(0, 1, 3, 4))])
    print('by column :', [f'{float(x):.2g}' for x in e.amax(dim=(0, 1, 2, 3))])
