"""Reads the cycle totals a -DDKV_STAMP build of attn_bwd_dkv_kernel leaves in dqkv (debug build: SSI_HIP_LIB=variants/libssi_stamp.so)."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
B, S, H, KV, hd = 8, 2048, 32, 8, 64
T = B * S
qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
dout = torch.randn(T, H * hd, device='cuda').bfloat16()
dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd)
for _ in range(3): ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "dq":   # -DDQ_STAMP build: wave 0 (head kvh*4) of each q-block workgroup, row q0
    qn = ["wait+barrier+DMA issue", "row frag reads landed", "S/dP MFMAs issued", "tr reads issued", "exponentials (+wait S/dP)", "cvt + dQ MFMAs issued"]
    for b in (0, 3):
        for qb in (63, 32, 8):
            row = dqkv[b * S + qb * 32].view(torch.uint8)[:36].view(torch.float32).tolist()
            tiles, nt, total = row[7], row[8], row[6]
            if tiles <= 0: continue
            print(f"batch {b} q-block {qb:2d}: {int(nt)} tiles ({int(tiles)} active), {total / nt:7.0f} cycles per tile; per ACTIVE tile: " +
                  ", ".join(f"{n} {row[i] / tiles:5.0f}" for i, n in enumerate(qn)))
    sys.exit(0)
names = ["wait+barrier+DMA issue", "frag reads landed", "S/dP MFMAs issued", "tr reads issued", "exponentials (+wait S/dP)", "cvt + dV/dK MFMAs issued"]
for b in (0, 3):
    for g in (0, 4, 8, 15):
        row = dqkv[b * S + g * 128].view(torch.uint8)[:36].view(torch.float32).tolist()
        steps, nst, total = row[7], row[8], row[6]
        if steps <= 0: continue
        print(f"batch {b} key group {g:2d}: {int(nst)} steps ({int(steps)} active), {total / nst:7.0f} cycles per step; per ACTIVE step: " +
              ", ".join(f"{n} {row[i] / steps:5.0f}" for i, n in enumerate(names)))
