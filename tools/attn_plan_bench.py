"""Attention forward / backward alone on packed rows, with and without the work plan, in ONE process (round 5): 32 / 8 heads as in the model.
usage: python tools/attn_plan_bench.py [padded|packed|plain|b2] — padded: the rows of `bench.py --padded` after the unpadding (4 batches);
packed: BASELINE config E's rows (B = 2, S = 8192, documents of 440-1100 tokens); plain: the headline batch as 8 documents (plan) against plain rows;
b2: the reference's default micro-batch 2 x 2048.  Per-kernel times: run it under `rocprofv3 --kernel-trace --stats`."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
sys.path.insert(0, 'tests')
from ssi import ops, attn_plan, _lib
case = sys.argv[1] if len(sys.argv) > 1 else "padded"
H, KV, hd = 32, 8, 64
g = torch.Generator().manual_seed(5)
def docs_for(case, i):
    if case == "padded":   # lengths ~U(0.4 S, S) of 8 rows, end to end, tile tail as a document of its own
        lens = [int(torch.randint(820, 2049, (1,), generator=g)) for _ in range(8)]
        t = -(-sum(lens) // 256) * 256
        return [lens + ([t - sum(lens)] if t > sum(lens) else [])]
    if case == "packed":
        rows = []
        for _ in range(2):
            lens, left = [], 8192
            while left > 0:
                n = min(left, int(torch.randint(440, 1101, (1,), generator=g)))
                lens.append(n); left -= n
            rows.append(lens)
        return rows
    if case == "b2":
        return [[2048], [2048]]
    return [[2048]] * 8
def t(fn, iters=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for i in range(3):
    rows = docs_for(case, i)
    B, S = len(rows), sum(rows[0])
    T = B * S
    qkv = torch.randn(T, (H + 2 * KV) * hd, device='cuda').bfloat16()
    out = torch.empty(T, H * hd, device='cuda', dtype=torch.bfloat16)
    lse = torch.empty(B * H * S, device='cuda', dtype=torch.float32)
    dout = torch.randn(T, H * hd, device='cuda').bfloat16()
    dqkv = torch.empty_like(qkv); delta = torch.empty_like(lse)
    ds = torch.tensor([o for lens in rows for o, n in zip([sum(lens[:j]) for j in range(len(lens))], lens) for _ in range(n)], dtype=torch.int32).cuda()
    de = torch.tensor([o + n for lens in rows for o, n in zip([sum(lens[:j]) for j in range(len(lens))], lens) for _ in range(n)], dtype=torch.int32).cuda()
    pos = torch.cat([torch.arange(n) for lens in rows for n in lens]).to(torch.int32).cuda()
    table = torch.randn(max(max(r) for r in rows) + 8, hd // 2, 2, device='cuda')
    plan = attn_plan.plan_from_seq_lens(rows, H, KV)
    ws_bytes = ops.attn_bwd_workspace_bytes(B, S, H, KV, hd, torch.bfloat16)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device='cuda') if ws_bytes else None
    one_doc = all(len(r) == 1 for r in rows)
    f = t(lambda: ops.attn_fwd(qkv, out, lse, B, S, H, KV, hd, None if one_doc else ds, None if one_doc else de))
    line = f"{case} #{i}: B x S = {B} x {S}, {sum(len(r) for r in rows)} documents; fwd {f:.0f} us"
    if one_doc:
        b0 = t(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd, rope_table=table, workspace=ws))
        line += f"; bwd plain rows {b0:.0f} us ({hex(ops.attn_last_dispatch())})"
    b1 = t(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd, ds, de, rope_table=table, positions=pos, workspace=ws))
    line += f"; bwd packed, no plan {b1:.0f} us ({hex(ops.attn_last_dispatch())})"
    if plan is not None:
        pd = plan.to_device('cuda')
        b2 = t(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, S, H, KV, hd, ds, de, rope_table=table, positions=pos, workspace=ws, plan=pd))
        line += f"; with plan {b2:.0f} us ({hex(ops.attn_last_dispatch())}; {plan.n_dkv_items} dK/dV items, {plan.n_dq_groups} dQ groups)"
    else:
        line += "; the library builds no plan for these documents"
    print(line, flush=True)
