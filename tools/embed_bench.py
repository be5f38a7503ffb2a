"""Embedding backward (deterministic scatter-add) alone: T = 16384, D = 2048, bf16; the bench's synthetic tokens and contrast cases."""
import os, sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
from ssi.data import synthetic_batch
T, D, V = 16384, 2048, 133376
dout = torch.randn(T, D, device='cuda').bfloat16()
table = torch.zeros(V, D, device='cuda', dtype=torch.bfloat16)
def bench(name, tok):
    tok = tok.cuda().contiguous()
    def run(): ops.embed_bwd(tok, dout, table, V)
    for _ in range(3): run()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run()
    e.record(); torch.cuda.synchronize()
    cnt = torch.bincount(tok)
    print(os.environ.get("TAG", ""), f"{name:28s} {s.elapsed_time(e) / 20 * 1e3:7.1f} us; max count {int(cnt.max())}, {int((cnt > 0).sum())} distinct, "
          f"{int((cnt > 1).sum())} repeated", flush=True)
g = torch.Generator().manual_seed(0)
bench("synthetic batch", synthetic_batch(8, 2048, 5000, rank=0, index=0)["tokens"].reshape(-1))
bench("all distinct", torch.arange(T))
one = torch.arange(T); one[torch.randperm(T, generator=g)[:396]] = 7
bench("one token x396, rest distinct", one)
pairs = torch.arange(T) % (T // 2)
bench("every token twice (far apart)", pairs)
bench("uniform over 5000 ids", torch.randint(0, 5000, (T,), generator=g))
