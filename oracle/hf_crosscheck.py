"""Pin the CPU oracle against an independent implementation and write golden fixtures.  TEST INFRASTRUCTURE ONLY.

Run in the authoring container:  ``python -m oracle.hf_crosscheck``  ->  ``tests/golden/*.npz``

Why HF: the reference's arithmetic lives in ``torchtune==0.5.0`` (pip, absent here and un-vendored:
``/root/reference/.gitmodules:1-3`` points at an empty directory), and the reference's tests pin no loss/logit value
(SURVEY.md §4).  ``transformers.LlamaForCausalLM`` built from a local ``LlamaConfig`` (random init, no hub access) is an
independent implementation of the same architecture; torchtune's own ``convert_weights.tune_to_hf`` maps between the two
with a q/k row permutation (interleaved RoPE pairs <-> half-split), restated in ``tune_to_hf_qk`` below.

No reference source is imported, executed or copied by this script.
"""

from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

from .llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama, compute_loss
from . import step_oracle

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

# name -> (model params, batch, seq, weight seed)
CASES = {
    # ragged vocab, S not divisible by 8, GQA ratio 4, head_dim 16
    "tiny": (dict(vocab_size=515, num_layers=2, num_heads=8, num_kv_heads=2, embed_dim=128, max_seq_len=256,
                  intermediate_dim=256), 2, 45, 11),
    # head_dim 64 (the production head size), T = B*S = 256, dims that the MFMA tiles accept
    "small": (dict(vocab_size=700, num_layers=2, num_heads=4, num_kv_heads=1, embed_dim=256, max_seq_len=512,
                   intermediate_dim=512), 2, 128, 12),
}


def seeded_state_dict(params: dict, seed: int) -> dict:
    """Weights from the frozen ``numpy.random.RandomState`` stream so fixtures need only the seed."""
    rs = np.random.RandomState(seed)
    m = OracleLlama(**params, rope_cache_len=8)
    sd = {}
    for name, p in m.state_dict().items():
        shape = tuple(p.shape)
        if name.endswith("scale"):
            arr = 1.0 + 0.1 * rs.standard_normal(shape)
        else:
            arr = 0.05 * rs.standard_normal(shape)
        sd[name] = torch.from_numpy(arr.astype(np.float32))
    return sd


def seeded_batch(vocab: int, b: int, s: int, seed: int) -> dict:
    rs = np.random.RandomState(seed + 1000)
    tokens = rs.randint(0, vocab, size=(b, s)).astype(np.int64)
    labels = tokens.copy()
    labels[0, : max(1, s // 6)] = -100          # masked prompt span (leading ignore)
    labels[1, s - max(1, s // 5):] = -100       # right padding (trailing ignore)
    labels[0, 0] = -100
    return {"tokens": torch.from_numpy(tokens), "labels": torch.from_numpy(labels)}


def weights_digest(sd: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].numpy().tobytes())
    return h.hexdigest()


def tune_to_hf_qk(w: torch.Tensor, n_heads: int) -> torch.Tensor:
    """Inverse of HF->tune ``w.view(n_heads, 2, hd/2, dim).transpose(1, 2)``."""
    dim = w.shape[1]
    hd = w.shape[0] // n_heads
    return w.view(n_heads, hd // 2, 2, dim).transpose(1, 2).reshape(n_heads * hd, dim)


def build_hf(params: dict, sd: dict):
    from transformers import LlamaConfig, LlamaForCausalLM

    rope = {"rope_type": "llama3", "factor": float(params.get("scale_factor", 32)), "low_freq_factor": 1.0,
            "high_freq_factor": 4.0, "original_max_position_embeddings": 8192,
            "rope_theta": float(params.get("rope_base", 500_000))}
    kw = dict(vocab_size=params["vocab_size"], hidden_size=params["embed_dim"],
              intermediate_size=params["intermediate_dim"], num_hidden_layers=params["num_layers"],
              num_attention_heads=params["num_heads"], num_key_value_heads=params["num_kv_heads"],
              max_position_embeddings=params["max_seq_len"], rms_norm_eps=params.get("norm_eps", 1e-5),
              tie_word_embeddings=True, attention_bias=False, mlp_bias=False, attention_dropout=0.0)
    try:
        cfg = LlamaConfig(**kw, rope_theta=rope["rope_theta"], rope_scaling={k: v for k, v in rope.items() if k != "rope_theta"})
    except Exception:  # newer transformers: rope_parameters
        cfg = LlamaConfig(**kw, rope_parameters=rope)
    cfg._attn_implementation = "eager"
    hf = LlamaForCausalLM(cfg).to(torch.float32).eval()
    H, KV = params["num_heads"], params["num_kv_heads"]
    new = {"model.embed_tokens.weight": sd["tok_embeddings.weight"], "model.norm.weight": sd["norm.scale"],
           "lm_head.weight": sd["tok_embeddings.weight"]}
    for i in range(params["num_layers"]):
        t, h = f"layers.{i}.", f"model.layers.{i}."
        new[h + "self_attn.q_proj.weight"] = tune_to_hf_qk(sd[t + "attn.q_proj.weight"], H)
        new[h + "self_attn.k_proj.weight"] = tune_to_hf_qk(sd[t + "attn.k_proj.weight"], KV)
        new[h + "self_attn.v_proj.weight"] = sd[t + "attn.v_proj.weight"]
        new[h + "self_attn.o_proj.weight"] = sd[t + "attn.output_proj.weight"]
        new[h + "mlp.gate_proj.weight"] = sd[t + "mlp.w1.weight"]
        new[h + "mlp.down_proj.weight"] = sd[t + "mlp.w2.weight"]
        new[h + "mlp.up_proj.weight"] = sd[t + "mlp.w3.weight"]
        new[h + "input_layernorm.weight"] = sd[t + "sa_norm.scale"]
        new[h + "post_attention_layernorm.weight"] = sd[t + "mlp_norm.scale"]
    missing, unexpected = hf.load_state_dict(new, strict=False)
    assert not unexpected and all("rotary" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return hf


def oracle_model(params: dict, sd: dict, chunks: int = 8) -> OracleLlama:
    m = OracleLlama(**params, rope_cache_len=params["max_seq_len"])
    m.load_state_dict(sd)
    m.set_num_output_chunks(chunks)
    return m


def run_case(name: str) -> dict:
    params, b, s, seed = CASES[name]
    sd = seeded_state_dict(params, seed)
    batch = seeded_batch(params["vocab_size"], b, s, seed)
    model = oracle_model(params, sd)
    loss_fn = OracleCEWithChunkedOutputLoss()

    # ---- forward: logits + loss, oracle vs HF --------------------------------------------------------------------
    with torch.no_grad():
        chunks = model(batch["tokens"])
        logits = torch.cat(chunks, dim=1)                                      # [b, s, V]
    loss = compute_loss(batch, model, loss_fn)
    hf = build_hf(params, sd)
    hf_logits = hf(input_ids=batch["tokens"]).logits.float()
    shifted = torch.hstack((batch["labels"][..., 1:], torch.full_like(batch["labels"][..., -1:], -100)))
    hf_loss = torch.nn.functional.cross_entropy(hf_logits.reshape(-1, hf_logits.size(-1)), shifted.reshape(-1),
                                                ignore_index=-100, reduction="sum") / (shifted != -100).sum()
    d_logit = float((logits - hf_logits.detach()).abs().max())
    rel_logit = d_logit / float(hf_logits.detach().abs().max())
    rel_loss = abs(float(loss) - float(hf_loss)) / abs(float(hf_loss))
    print(f"[{name}] oracle-vs-HF  max|dlogit|={d_logit:.3e} rel={rel_logit:.3e}  loss {float(loss):.7f} vs "
          f"{float(hf_loss):.7f} rel={rel_loss:.3e}")
    assert rel_logit < 2e-6 and rel_loss < 2e-6, "oracle disagrees with HF-Llama"

    # ---- backward through the trainer algebra (one window of one micro-batch), grads oracle vs HF ------------------
    n_unshift = int((batch["labels"] != -100).sum())
    n_shift = int((shifted != -100).sum())
    (loss * n_unshift).backward()
    (hf_loss * n_unshift).backward()
    H, KV = params["num_heads"], params["num_kv_heads"]
    g_q = model.layers[0].attn.q_proj.weight.grad
    g_q_hf = hf.model.layers[0].self_attn.q_proj.weight.grad
    gq_rel = float((tune_to_hf_qk(g_q, H) - g_q_hf).abs().max() / g_q_hf.abs().max())
    ge_rel = float((model.tok_embeddings.weight.grad - hf.model.embed_tokens.weight.grad).abs().max()
                   / hf.model.embed_tokens.weight.grad.abs().max())
    print(f"[{name}] grads oracle-vs-HF  q_proj rel={gq_rel:.3e}  tok_embeddings rel={ge_rel:.3e}")
    assert gq_rel < 1e-4 and ge_rel < 1e-4

    # ---- one optimizer step (reference defaults conf/training.yaml:2-10, fp32) -----------------------------------
    grads = {k: p.grad.clone() / n_unshift for k, p in model.named_parameters()}
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    step_oracle.optimizer_step(model, opt, n_unshift)
    after = {k: p.detach().clone() for k, p in model.named_parameters()}

    rows = [0, s // 2, s - 1]
    out = {
        "weights_sha256": np.array(weights_digest(sd)),
        "seed": np.array(seed), "batch": np.array(b), "seq": np.array(s),
        "tokens": batch["tokens"].numpy(), "labels": batch["labels"].numpy(),
        "loss": np.array(float(loss), dtype=np.float64), "hf_loss": np.array(float(hf_loss), dtype=np.float64),
        "n_unshifted": np.array(n_unshift), "n_shifted": np.array(n_shift),
        "logit_rows": np.array(rows), "logits_at_rows": logits[:, rows, :].numpy(),
        "logits_absmax": np.array(float(logits.abs().max())),
        "lse_row0": torch.logsumexp(logits[0], dim=-1).numpy(),
    }
    for k in ("tok_embeddings.weight", "layers.0.attn.q_proj.weight", "layers.0.attn.k_proj.weight",
              "layers.0.attn.v_proj.weight", "layers.0.attn.output_proj.weight", "layers.1.mlp.w1.weight",
              "layers.1.mlp.w2.weight", "layers.1.mlp.w3.weight", "layers.0.sa_norm.scale", "layers.1.mlp_norm.scale",
              "norm.scale"):
        g = grads[k]
        out["gradnorm/" + k] = np.array(float(g.norm()), dtype=np.float64)
        out["grad/" + k] = g.reshape(-1, g.shape[-1])[:4, :16].numpy() if g.dim() > 1 else g[:16].numpy()
        a = after[k]
        out["after/" + k] = a.reshape(-1, a.shape[-1])[:4, :16].numpy() if a.dim() > 1 else a[:16].numpy()
    out["gradnorm_total"] = np.array(float(torch.sqrt(sum(g.double().pow(2).sum() for g in grads.values()))))
    return out


def main() -> int:
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    for name in CASES:
        out = run_case(name)
        path = os.path.join(GOLDEN_DIR, f"llama_{name}.npz")
        np.savez_compressed(path, **out)
        print(f"[{name}] wrote {path} ({os.path.getsize(path)} bytes)")
    # RoPE theta / table known answers at the production head size
    from .llama_oracle import llama3_scaled_theta, rope_cache
    theta = llama3_scaled_theta(64)
    pos = np.array([0, 1, 2047, 8191, 8192, 131071])
    table = rope_cache(theta, 131072)[pos].numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "rope_hd64.npz"), theta=theta.numpy(), pos=pos, table=table)
    print("wrote rope_hd64.npz")
    return 0


if __name__ == "__main__":
    sys.exit(main())
