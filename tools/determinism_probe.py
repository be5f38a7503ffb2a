"""Is the training step bit-reproducible?  The same accumulation window (ga micro-batches, fixed batches, fixed weights) run again and again on the
full-size model; after every window the flat gradient buffer is compared with the first window's, parameter by parameter.  Reports which
parameters ever differed, where (rows / columns of the first difference) and how often.  usage: python tools/determinism_probe.py [windows] [ga] [B] [S]"""
import sys, copy, torch
sys.path.insert(0, 'speech-integration_amd')
from ssi.data import synthetic_batch
from ssi.llama_configs import configllama3_2_1b
from ssi.loss import CEWithChunkedOutputLoss, compute_loss
from ssi.model import HipLlamaDecoder

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ga = int(sys.argv[2]) if len(sys.argv) > 2 else 4
B = int(sys.argv[3]) if len(sys.argv) > 3 else 8
S = int(sys.argv[4]) if len(sys.argv) > 4 else 2048
dev = torch.device('cuda', 0)
cfg = copy.deepcopy(configllama3_2_1b)
cfg.n_dsus, cfg.modality_tokens = 5000, True
torch.manual_seed(1)
model = HipLlamaDecoder(**cfg.parameters, dtype=torch.bfloat16, device=dev, rope_cache_len=max(S, 2048))
with torch.no_grad():
    model._flat.normal_(0.0, 0.02)
    model._view("emb")[cfg.vocab_size:].zero_()
    for p, name, _ in model._param_src:
        if name.endswith("norm"):
            p.fill_(1.0)
model.train()
loss_fn = CEWithChunkedOutputLoss()
model.set_num_output_chunks(8)
batches = [{k: v.to(dev) for k, v in synthetic_batch(B, S, 5000, index=i).items()} for i in range(ga)]
ref, ref_losses, bad = None, None, {}
for w in range(windows):
    model.zero_grad(set_to_none=True)
    losses = []
    for j in range(ga):
        loss = compute_loss(batches[j], model, loss_fn)
        (loss * 1000.0).backward()
        losses.append(loss.item())
    g = model._flat_grad
    if ref is None:
        ref, ref_losses = g.clone(), losses
        continue
    if losses != ref_losses:
        print(f"window {w}: losses differ {losses} vs {ref_losses}", flush=True)
    if not torch.equal(g, ref):
        for p, name, rows in model._param_src:
            a, b = model._view(name, rows, g), model._view(name, rows, ref)
            if not torch.equal(a, b):
                d = (a != b)
                idx = d.nonzero()
                first = idx[0].tolist()
                n = int(d.sum())
                bad.setdefault(name + (str(rows) if rows else ""), []).append((w, n, first, float((a.float() - b.float()).abs().max())))
        print(f"window {w}: gradients differ in {sorted(k for k, v in bad.items() if v and v[-1][0] == w)}", flush=True)
# ---- second part: whole optimizer steps (the bench.py loop) from one and the same state, again and again --------------------------------
from ssi.optimizer import HipAdamW, scale_grads
from ssi.train_utils import count_token_types_async, get_token_type_ranges
opt = HipAdamW(model.parameters(), model=model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
ranges = get_token_type_ranges(cfg)
pad_id = cfg._base_vocab_size_txt + cfg.n_dsus + 2 + 4
p0 = model._flat.clone()
trials, steps = max(4, windows // 4), 3
ref_p, ref_l, bad_p = None, None, {}
import os
overlap = os.environ.get("SSI_ADAMW_OVERLAP", "0") == "1"
model.zero_grad(set_to_none=True)   # (the first part left a window open)
for t in range(trials):
    with torch.no_grad():
        model._flat.copy_(p0)
        opt._exp_avg.zero_(); opt._exp_avg_sq.zero_()
    opt._step_count = 0
    model._hip_epoch += 1
    ls = []
    for st in range(steps):
        rows, n_dev = [], None
        for j in range(ga):
            b = batches[(st * ga + j) % len(batches)]
            counts = count_token_types_async(b["tokens"], ranges, pad_id, b["labels"], -100)
            n_dev = counts[-1] if n_dev is None else n_dev + counts[-1]
            if j == ga - 1 and overlap:
                opt.overlap_with_backward(1.0 / n_dev.to(torch.float32))
            lb = compute_loss(b, model, loss_fn) * counts[-1]
            lb.backward()
            rows.append(torch.cat((counts.double(), lb.detach().double().reshape(1))))
        host = torch.stack(rows).sum(0).tolist()
        n_tok = int(host[-2])
        scale_grads(model, torch.tensor(1.0 / n_tok))
        opt.step()
        opt.zero_grad(set_to_none=True)
        ls.append(host[-1] / n_tok)
    torch.cuda.synchronize()
    if ref_p is None:
        ref_p, ref_l = model._flat.clone(), ls
        continue
    if ls != ref_l or not torch.equal(model._flat, ref_p):
        names = []
        for p, name, rows_ in model._param_src:
            a, b_ = model._view(name, rows_), model._view(name, rows_, ref_p)
            if not torch.equal(a, b_):
                names.append(name)
                bad_p.setdefault(name, 0)
                bad_p[name] += 1
        print(f"trial {t}: losses {ls} vs {ref_l}; {len(names)} parameters differ: {names[:6]}{'...' if len(names) > 6 else ''}", flush=True)
print(f"{trials} trials of {steps} optimizer steps x {ga} micro-batches (AdamW {'under' if overlap else 'behind'} the backward): "
      f"{'bit-identical every time' if not bad_p else f'DIFFERENCES in {len(bad_p)} parameters'}")
print(f"{windows} windows of {ga} micro-batches ({B} x {S}): {'bit-identical every time' if not bad else 'DIFFERENCES'}")
for k, v in sorted(bad.items()):
    print(f"  {k}: {len(v)} windows; e.g. window {v[0][0]}: {v[0][1]} elements, first at {v[0][2]}, max |d| {v[0][3]:.3e}")
