"""Is the step launch-bound?  Captures forward + backward of one micro-batch (the ctypes launches go to torch's capture stream, the arena keeps every
pointer stable) into a hipGraph and replays it against the eager step.  python tools/graph_probe.py <batch> <seq>
Measured: B=2, S=512 (config P on the GPU) 24.1 ms eager vs 23.1 ms replayed - the small step is bound by under-filled 256 x 256 tile grids and the
weight traffic, not by launches; at the headline shape the kernels already account for the whole step (profiles/)."""
import sys, time, torch, copy
sys.path.insert(0, 'speech-integration_amd'); sys.path.insert(0, '.')
from ssi.data import synthetic_batch
from ssi.llama_configs import configllama3_2_1b
from ssi.loss import CEWithChunkedOutputLoss, compute_loss
from ssi.model import HipLlamaDecoder
from ssi.optimizer import HipAdamW, scale_grads
from ssi.train_utils import count_token_types_async, get_token_type_ranges
B, S = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device('cuda', 0)
lcfg = copy.deepcopy(configllama3_2_1b); lcfg.n_dsus, lcfg.modality_tokens = 5000, True
model = HipLlamaDecoder(**lcfg.parameters, dtype=torch.bfloat16, device=dev, rope_cache_len=2048)
with torch.no_grad():
    model._flat.normal_(0.0, 0.02)
    for p, name, _ in model._param_src:
        if name.endswith("norm"): p.fill_(1.0)
model.train(); loss_fn = CEWithChunkedOutputLoss(); model.set_num_output_chunks(8)
opt = HipAdamW(model.parameters(), model=model, lr=2e-4)
ranges = get_token_type_ranges(lcfg); pad_id = lcfg._base_vocab_size_txt + lcfg.n_dsus + 2 + 4
batches = [{k: v.to(dev) for k, v in synthetic_batch(B, S, 5000, index=i).items()} for i in range(4)]
static = {k: v.clone() for k, v in batches[0].items()}

def fwd_bwd(b):
    counts = count_token_types_async(b["tokens"], ranges, pad_id, b["labels"], -100)
    lb = compute_loss(b, model, loss_fn) * counts[-1]
    lb.backward()
    return counts, lb.detach()

def finish(counts, lb):
    host = torch.cat((counts.double(), lb.double().reshape(1))).tolist()
    n_tok = int(host[-2])
    scale_grads(model, torch.tensor(1.0 / n_tok)); opt.step(); opt.zero_grad(set_to_none=True)
    return host[-1] / n_tok

# eager reference losses
losses = []
for i in range(3): losses.append(finish(*fwd_bwd(batches[i % 4])))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): finish(*fwd_bwd(batches[i % 4]))
torch.cuda.synchronize(); print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3, losses)
# capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    out = fwd_bwd(static)
torch.cuda.current_stream().wait_stream(s)
model._grads_dirty = False
with torch.cuda.graph(g):
    out = fwd_bwd(static)
print("captured")
def graphed(b):
    for k in static: static[k].copy_(b[k])
    model._grads_dirty = False
    g.replay()
    return out
l2 = []
for i in range(3): l2.append(finish(*graphed(batches[i % 4])))
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): finish(*graphed(batches[i % 4]))
torch.cuda.synchronize(); print("graph ms/step", (time.perf_counter() - t0) / 20 * 1e3, l2)
