"""CPU oracle for the speech-integration training hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product package (``speech-integration_amd/ssi``) never does.

What it restates (pure PyTorch, CPU): the arithmetic the reference reaches through ``torchtune==0.5.0``
(``/root/reference/uv.lock:2799``; the package is NOT in the reference tree nor in this image), anchored on the
reference's own call sites:

* model graph ............ ``ssi/model.py:18-39`` -> ``llama3_2(**ConfigLlama3_2.parameters)`` (``ssi/llama_configs.py:118-140``)
* forward + label shift .. ``ssi/loss.py:7-22``
* chunked CE ............. ``ssi/trainer.py:299-304`` (``CEWithChunkedOutputLoss``, 8 chunks, ignore_index -100)
* step algebra ........... ``ssi/trainer.py:385-424`` (see ``oracle/step_oracle.py``)

Parity status: the reference holds NO numeric golden vector for this path, and torchtune cannot be
imported here, so this restatement is pinned by an independent implementation instead: Hugging Face
``transformers.LlamaForCausalLM`` built from a local ``LlamaConfig`` (no download) and fed the same weights through
the HF<->torchtune q/k row permutation — see ``oracle/hf_crosscheck.py`` and ``tests/golden/``.  Against the
*reference's own fixtures* parity is therefore "unpinned"; against HF-Llama it is pinned to <= 1e-6 relative.
"""

from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F
from torch import Tensor, nn

IGNORE_INDEX = -100


# --------------------------------------------------------------------------------------------------------------------
# RoPE table — torchtune.models.llama3_1._position_embeddings.Llama3ScaledRoPE (0.5.0); used via llama3_2()
# --------------------------------------------------------------------------------------------------------------------
def llama3_scaled_theta(
    head_dim: int,
    base: float = 500_000,
    scale_factor: float = 32,
    low_freq_factor: float = 1,
    high_freq_factor: float = 4,
    old_context_len: int = 8192,
) -> Tensor:
    """theta_i (fp32, [head_dim/2]) with the Llama-3 frequency scaling.  Element-wise fp32 tensor arithmetic, in the
    same operation order as torchtune's ``rope_init``/``apply_scaling``."""
    freqs = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: (head_dim // 2)].float() / head_dim))
    low_freq_wavelen = old_context_len / low_freq_factor
    high_freq_wavelen = old_context_len / high_freq_factor
    new_freqs = []
    for freq in freqs:
        wavelen = 2 * math.pi / freq
        if wavelen < high_freq_wavelen:
            new_freqs.append(freq)
        elif wavelen > low_freq_wavelen:
            new_freqs.append(freq / scale_factor)
        else:
            smooth = (old_context_len / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
            new_freqs.append((1 - smooth) * freq / scale_factor + smooth * freq)
    return torch.stack([f.to(torch.float32) for f in new_freqs])


def rope_cache(theta: Tensor, max_seq_len: int) -> Tensor:
    """[max_seq_len, head_dim/2, 2] fp32 (cos, sin) — ``build_rope_cache``."""
    seq_idx = torch.arange(max_seq_len, dtype=theta.dtype)
    idx_theta = torch.einsum("i, j -> ij", seq_idx, theta).float()
    return torch.stack([torch.cos(idx_theta), torch.sin(idx_theta)], dim=-1)


def apply_rope(x: Tensor, cache: Tensor, input_pos: Optional[Tensor] = None) -> Tensor:
    """x [b, s, n_h, h_d]; adjacent-pair rotation in fp32, cast back (``Llama3ScaledRoPE.forward``)."""
    seq_len = x.size(1)
    rc = cache[:seq_len] if input_pos is None else cache[input_pos]
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    rc = rc.view(-1, xs.size(1), 1, xs.size(3), 2)
    out = torch.stack(
        [xs[..., 0] * rc[..., 0] - xs[..., 1] * rc[..., 1], xs[..., 1] * rc[..., 0] + xs[..., 0] * rc[..., 1]], -1
    )
    return out.flatten(3).type_as(x)


# --------------------------------------------------------------------------------------------------------------------
# Modules (state-dict keys = torchtune format, consumed by ssi/checkpoint.py:325-331,352-358)
# --------------------------------------------------------------------------------------------------------------------
class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float):
        super().__init__()
        self.eps = eps
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x: Tensor) -> Tensor:
        x32 = x.float()
        xn = (x32 * torch.rsqrt(x32.pow(2).mean(-1, keepdim=True) + self.eps)).type_as(x)
        return xn * self.scale


class Attention(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_kv: int):
        super().__init__()
        self.n_heads, self.n_kv, self.hd = n_heads, n_kv, dim // n_heads
        self.q_proj = nn.Linear(dim, n_heads * self.hd, bias=False)
        self.k_proj = nn.Linear(dim, n_kv * self.hd, bias=False)
        self.v_proj = nn.Linear(dim, n_kv * self.hd, bias=False)
        self.output_proj = nn.Linear(dim, dim, bias=False)

    def forward(self, x: Tensor, cache: Tensor, mask: Optional[Tensor], input_pos: Optional[Tensor]) -> Tensor:
        b, s, _ = x.shape
        q = apply_rope(self.q_proj(x).view(b, s, self.n_heads, self.hd), cache, input_pos).transpose(1, 2)
        k = apply_rope(self.k_proj(x).view(b, s, self.n_kv, self.hd), cache, input_pos)
        v = self.v_proj(x).view(b, s, self.n_kv, self.hd)
        rep = self.n_heads // self.n_kv
        if rep > 1:  # kv head j serves q heads j*rep .. j*rep+rep-1
            k = k.unsqueeze(3).expand(b, s, self.n_kv, rep, self.hd).flatten(2, 3)
            v = v.unsqueeze(3).expand(b, s, self.n_kv, rep, self.hd).flatten(2, 3)
        k, v = k.transpose(1, 2), v.transpose(1, 2)
        if mask is not None and mask.dim() == 3:
            mask = mask[:, None, :, :]  # [b, s, s] bool (True = attend), broadcast over heads as torchtune's MultiHeadAttention does
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, is_causal=mask is None)
        return self.output_proj(o.transpose(1, 2).contiguous().view(b, s, -1))


class FeedForward(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.w1 = nn.Linear(dim, hidden, bias=False)  # gate
        self.w2 = nn.Linear(hidden, dim, bias=False)  # down
        self.w3 = nn.Linear(dim, hidden, bias=False)  # up

    def forward(self, x: Tensor) -> Tensor:
        return self.w2(F.silu(self.w1(x)) * self.w3(x))


class Layer(nn.Module):
    def __init__(self, dim: int, n_heads: int, n_kv: int, hidden: int, eps: float):
        super().__init__()
        self.attn = Attention(dim, n_heads, n_kv)
        self.mlp = FeedForward(dim, hidden)
        self.sa_norm = RMSNorm(dim, eps)
        self.mlp_norm = RMSNorm(dim, eps)

    def forward(self, x, cache, mask, input_pos):
        h = self.attn(self.sa_norm(x), cache, mask, input_pos) + x
        return h + self.mlp(self.mlp_norm(h))


class OracleLlama(nn.Module):
    """Decoder-only Llama-3.2 with tied LM head in ``num_output_chunks`` sequence chunks
    (``TransformerDecoder`` + ``TiedLinear`` of torchtune 0.5.0)."""

    def __init__(
        self,
        vocab_size: int,
        num_layers: int,
        num_heads: int,
        num_kv_heads: int,
        embed_dim: int,
        max_seq_len: int,
        intermediate_dim: int,
        attn_dropout: float = 0.0,
        norm_eps: float = 1e-5,
        rope_base: int = 500_000,
        scale_factor: int = 32,
        rope_cache_len: Optional[int] = None,
    ):
        super().__init__()
        assert attn_dropout == 0.0
        self.tok_embeddings = nn.Embedding(vocab_size, embed_dim)
        self.layers = nn.ModuleList(
            [Layer(embed_dim, num_heads, num_kv_heads, intermediate_dim, norm_eps) for _ in range(num_layers)]
        )
        self.norm = RMSNorm(embed_dim, norm_eps)
        self.num_output_chunks = 0
        self.max_seq_len = max_seq_len
        with torch.device("cpu"):  # host arithmetic even when the module is built under a meta-device context
            theta = llama3_scaled_theta(embed_dim // num_heads, rope_base, scale_factor)
            # the real model caches max_seq_len (131072) positions; tests may cap the table, values are identical
            table = rope_cache(theta, rope_cache_len or max_seq_len)
        self.register_buffer("rope", table, persistent=False)

    def set_num_output_chunks(self, n: int) -> None:
        self.num_output_chunks = n

    def forward_hidden(self, tokens: Tensor, mask=None, input_pos=None) -> Tensor:
        h = self.tok_embeddings(tokens)
        for layer in self.layers:
            h = layer(h, self.rope, mask, input_pos)
        return self.norm(h)

    def forward(self, tokens: Tensor, mask=None, encoder_input=None, encoder_mask=None, input_pos=None):
        h = self.forward_hidden(tokens, mask, input_pos)
        w = self.tok_embeddings.weight
        if self.num_output_chunks > 0:
            return [F.linear(c, w) for c in h.chunk(self.num_output_chunks, dim=1)]
        return F.linear(h, w).float()


class OracleCEWithChunkedOutputLoss(nn.Module):
    """``torchtune.modules.loss.CEWithChunkedOutputLoss`` (0.5.0): sum-CE per chunk on fp32-upcast logits, divided by
    the count of non-ignored labels."""

    def __init__(self, num_output_chunks: int = 8, ignore_index: int = IGNORE_INDEX):
        super().__init__()
        self.num_output_chunks = num_output_chunks
        self.ignore_index = ignore_index

    def forward(self, logits, labels: Tensor) -> Tensor:
        if not isinstance(logits, list):  # single [N, V] tensor + flat labels (ssi/loss.py:17-19)
            total = (labels != self.ignore_index).sum()
            return F.cross_entropy(logits.float(), labels, ignore_index=self.ignore_index, reduction="sum") / total
        total = (labels != self.ignore_index).sum()
        lab = [c.reshape(-1) for c in labels.chunk(self.num_output_chunks, dim=1)]
        log = [c.reshape(-1, c.size(-1)) for c in logits]
        loss = 0.0
        for lg, lb in zip(log, lab):
            loss = loss + F.cross_entropy(lg.float(), lb, ignore_index=self.ignore_index, reduction="sum")
        return loss / total


def compute_loss(batch: dict, model, loss_fn) -> Tensor:
    """``ssi/loss.py:7-22`` verbatim in behaviour: forward, shift labels left by one, chunked CE.  Does not mutate ``batch``."""
    logits = model(
        tokens=batch["tokens"],
        mask=batch.get("mask"),
        encoder_input=batch.get("encoder_input"),
        encoder_mask=batch.get("encoder_mask"),
        input_pos=batch.get("input_pos"),
    )
    labels = batch["labels"]
    labels = torch.hstack((labels[..., 1:], torch.full_like(labels[..., -1:], loss_fn.ignore_index)))
    if not isinstance(logits, list):
        labels = labels.reshape(-1)
        logits = logits.reshape(-1, logits.size(-1))
    return loss_fn(logits, labels)


def build_oracle(params: dict, dtype: torch.dtype = torch.float32, seed: Optional[int] = None,
                 init_std: float = 0.02, rope_cache_len: Optional[int] = None) -> OracleLlama:
    """Random-init oracle model.  ``params`` = ``ConfigLlama3_2.parameters`` keys."""
    m = OracleLlama(**params, rope_cache_len=rope_cache_len)
    if seed is not None:
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in m.named_parameters():
                if name.endswith("scale"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
                else:
                    p.copy_(init_std * torch.randn(p.shape, generator=g))
    for p in m.parameters():  # parameters only: the RoPE table stays fp32 as in torchtune (explicit .float())
        p.data = p.data.to(dtype)
    return m
