"""Cross-entropy kernel alone at the step's shape: T = 16384 rows, V = 133258 (ld 133376), bf16, gradient in place."""
import sys, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi import ops
T, V, LD = 16384, 133258, 133376
logits = (torch.randn(T, LD, device='cuda') * 2).bfloat16()
labels = torch.randint(0, V, (T,), device='cuda')
labels[::7] = -100
row_loss = torch.empty(T, device='cuda'); row_lse = torch.empty(T, device='cuda')
work = logits.clone()
def run():
    ops.ce_fwd(work, labels, V, -100, row_loss, row_lse, write_grad=True)
for _ in range(3):
    work.copy_(logits); run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for _ in range(10):
    work.copy_(logits)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    tot += s.elapsed_time(e)
print(f"ce_fwd {tot / 10:.3f} ms  ({3 * T * LD * 2 / (tot / 10) / 1e9:.2f} TB/s at 3 passes, {2 * T * LD * 2 / (tot / 10) / 1e9:.2f} at 2)  loss {float(row_loss.sum()):.4f}")
