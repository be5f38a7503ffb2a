"""Llama-3.2 decoder with a DSU-extended, tied vocabulary on hand-written HIP kernels (gfx950).

Replaces what the reference builds at ``/root/reference/ssi/model.py:18-39`` (torchtune ``llama3_2(**params)``) and runs at
``/root/reference/ssi/loss.py:8-14``.  Public surface kept: ``setup_llama3_2_1b(cfg, llama_config, model_state_dict,
dtype_default, device_default)``; the returned module is callable as ``model(tokens=, mask=, encoder_input=,
encoder_mask=, input_pos=)``, has ``set_num_output_chunks``, exposes torchtune-format ``state_dict`` keys
 and populates ``p.grad`` on ``loss.backward()``.

MI355X-first design (not a module-per-op port):
* all parameters live in ONE flat HBM buffer (q/k/v fused to one [3072, D] weight, gate/up fused to one [2I, D] weight;
  the torchtune-named parameters are row-slice views), gradients and AdamW moments in matching flat buffers, so the
  optimizer is a single streaming kernel and data-parallel all-reduce works on contiguous per-layer buckets;
* the embedding table is stored with its row count padded to a multiple of 256 so that the LM-head GEMM, its dgrad and
  its wgrad run on full MFMA tiles (pad rows are zero and stay zero);
* forward/backward are two hand-scheduled kernel sequences over a persistent activation arena (stable pointers, no
  allocator traffic, no autograd graph of small ops) wrapped in three ``torch.autograd.Function`` seams:
  decoder stack, tied head -> logits, tied head + cross-entropy (logits are written once in the model dtype and turned
  into their gradient in place);
* there is no CPU path: constructing the model off-GPU raises.
"""

from __future__ import annotations

import os

import logging
import math
from typing import Any, Optional

import torch
from torch import Tensor, nn

from . import _lib, ops
from .constants import CROSS_ENTROPY_IGNORE_IDX, PRECISION_STR_TO_DTYPE
from .llama_configs import ConfigLlama3_2
from .ops import GEMM_NN, GEMM_NT, GEMM_TN

LOGGER = logging.getLogger(__name__)

VOCAB_ALIGN = 256  # MFMA tile edge


def _align(n: int, a: int) -> int:
    return (n + a - 1) // a * a


# --------------------------------------------------------------------------------------------------------------------
# RoPE table (torchtune Llama3ScaledRoPE: rope_init / apply_scaling / build_rope_cache).
# Host-side fp32 tensor arithmetic, done once; the rotation itself is the HIP kernel ``ssi_rope_inplace``.
# --------------------------------------------------------------------------------------------------------------------
def llama3_rope_table(head_dim: int, max_seq_len: int, base: float = 500_000, scale_factor: float = 32,
                      low_freq_factor: float = 1, high_freq_factor: float = 4, old_context_len: int = 8192) -> Tensor:
    freqs = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    lo_wl, hi_wl = old_context_len / low_freq_factor, old_context_len / high_freq_factor
    out = []
    for f in freqs:
        wavelen = 2 * math.pi / f
        if wavelen < hi_wl:
            out.append(f)
        elif wavelen > lo_wl:
            out.append(f / scale_factor)
        else:
            smooth = (old_context_len / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
            out.append((1 - smooth) * f / scale_factor + smooth * f)
    theta = torch.stack([t.to(torch.float32) for t in out])
    idx = torch.einsum("i, j -> ij", torch.arange(max_seq_len, dtype=theta.dtype), theta).float()
    return torch.stack([torch.cos(idx), torch.sin(idx)], dim=-1).contiguous()  # [max_seq_len, head_dim/2, 2]


class _Holder(nn.Module):
    """Parameter container so that ``state_dict`` / ``named_parameters`` carry torchtune's dotted names."""


class _Arena:
    """Persistent activation / workspace buffers keyed by name (stable device pointers across steps)."""

    def __init__(self, device: torch.device):
        self.device = device
        self.buf: dict[str, Tensor] = {}

    def get(self, name: str, shape: tuple, dtype: torch.dtype) -> Tensor:
        n = 1
        for s in shape:
            n *= s
        t = self.buf.get(name)
        if t is None or t.dtype != dtype or t.numel() < n:
            # a buffer that has to GROW (ragged batches: the packed length differs from batch to batch) takes a quarter more than asked, so
            # that a run reaches its largest batch in a few allocations instead of one per new length; fixed shapes never come here twice
            grow = t is not None and t.dtype == dtype
            if grow:
                self.buf[name] = t = None  # (freed BEFORE the larger one is requested: the cached block may serve a smaller sibling)
            t = torch.empty(max(n + n // 4 if grow else n, 1), dtype=dtype, device=self.device)
            self.buf[name] = t
        return t[:n].view(shape)

    def bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.buf.values())


class HipLlamaDecoder(nn.Module):
    def __init__(self, vocab_size: int, num_layers: int, num_heads: int, num_kv_heads: int, embed_dim: int,
                 max_seq_len: int, intermediate_dim: int, attn_dropout: float = 0.0, norm_eps: float = 1e-5,
                 rope_base: int = 500_000, scale_factor: int = 32, *, dtype: torch.dtype = torch.bfloat16,
                 device: torch.device | str = "cuda", rope_cache_len: Optional[int] = None) -> None:
        super().__init__()
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if device.type != "cuda":
            raise _lib.HipLibraryError(
                f"HipLlamaDecoder runs only on an MI355X GPU through libssi_hip.so (got device={device}); "
                "there is no CPU fallback in the product path")
        _lib.load()
        if attn_dropout != 0.0:
            raise NotImplementedError("attn_dropout must be 0.0 (the reference config, llama_configs.py:136)")
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"unsupported dtype {dtype}")
        if embed_dim % num_heads or num_heads % num_kv_heads or embed_dim % 8 or intermediate_dim % 8:
            raise ValueError("embed_dim must divide into heads, heads into kv heads, and dims must be multiples of 8")
        self.vocab_size, self.num_layers = vocab_size, num_layers
        self.num_heads, self.num_kv_heads = num_heads, num_kv_heads
        self.embed_dim, self.intermediate_dim = embed_dim, intermediate_dim
        self.head_dim = embed_dim // num_heads
        self.max_seq_len, self.norm_eps = max_seq_len, float(norm_eps)
        self.dtype, self.device = dtype, device
        self.vocab_pad = _align(vocab_size, VOCAB_ALIGN)
        self.qkv_dim = (num_heads + 2 * num_kv_heads) * self.head_dim
        self.num_output_chunks = 0
        self.ignore_index = CROSS_ENTROPY_IGNORE_IDX

        D, I, Q, KVD = embed_dim, intermediate_dim, num_heads * self.head_dim, num_kv_heads * self.head_dim
        # ---- flat parameter layout ------------------------------------------------------------------------------
        off = 0
        self._slices: dict[str, tuple[int, tuple]] = {}

        def take(name: str, shape: tuple) -> None:
            nonlocal off
            n = 1
            for s in shape:
                n *= s
            self._slices[name] = (off, shape)
            off += _align(n, 8)

        # [E_pad | attention weights of every layer: L0.wqkv, L0.wo, L1.wqkv, ... | L0: w13, w2, sa_norm, mlp_norm | L1: ... | norm].
        # The attention projections sit together, one layer after the other at a fixed stride: their weight gradients (64 and 96 output
        # tiles, too few for 256 CUs) are computed for a GROUP of layers in one batched launch (_backward_hidden), which then finishes a
        # contiguous range of the gradient buffer = one data-parallel bucket per group.
        take("emb", (self.vocab_pad, D))
        for l in range(num_layers):
            take(f"L{l}.wqkv", (self.qkv_dim, D))
            take(f"L{l}.wo", (D, Q))
        for l in range(num_layers):
            take(f"L{l}.w13", (2 * I, D))
            take(f"L{l}.w2", (D, I))
            take(f"L{l}.sa_norm", (D,))
            take(f"L{l}.mlp_norm", (D,))
        take("norm", (D,))
        self._flat_numel = off
        self._flat = torch.zeros(off, dtype=dtype, device=device)
        self._flat_grad = torch.zeros(off, dtype=dtype, device=device)
        # [in, out] copies of the weights named by `dgrad_transposed`: their data-gradient GEMMs dX = dY W then run in the
        # k-contiguous operand form; refreshed lazily after the weights change (one transpose launch per weight per optimizer step)
        self._wt: dict[str, Tensor] = {}
        self._wt_key: Optional[tuple] = None
        self._hip_epoch = 0  # bumped by kernels that modify the weights in place (fused AdamW)
        self._grad_views: dict[str, Tensor] = {}
        # Layers whose attention-projection weight gradients share one batched launch (0 / 1 = every layer on its own, split-K).  8 at the
        # 1B shape: 8 x 64 and 8 x 96 output tiles = 2 and 3 full rounds of the 256 CUs.
        self.wgrad_group = max(1, min(int(os.environ.get("SSI_WGRAD_GROUP", "8")), max(num_layers, 1)))
        # DP buckets in the order backward finishes them: final norm; per layer L-1..0 its MLP block (w13, w2, both norm scales), and
        # behind the lowest layer of every group the attention weights of that group; embedding (tied: finished last)
        self.buckets: list[tuple[str, int, int]] = []
        lo, _ = self._slices["norm"]
        self.buckets.append(("norm", lo, off))
        mlp_lo = lambda l: self._slices[f"L{l}.w13"][0] if l < num_layers else self._slices["norm"][0]  # noqa: E731
        attn_lo = lambda l: self._slices[f"L{l}.wqkv"][0] if l < num_layers else mlp_lo(0)  # noqa: E731
        for l in reversed(range(num_layers)):
            self.buckets.append((f"L{l}.mlp", mlp_lo(l), mlp_lo(l + 1)))
            if l % self.wgrad_group == 0:
                self.buckets.append((f"attn.{l}", attn_lo(l), attn_lo(min(l + self.wgrad_group, num_layers))))
        self.buckets.append(("emb", 0, attn_lo(0) if num_layers else self._slices["norm"][0]))
        self._bucket_by_name = {b[0]: b for b in self.buckets}
        self._bucket_group = self.wgrad_group

        # ---- torchtune-named parameters as views ------------------------------------------------------------------
        def view(name: str, rows: Optional[tuple[int, int]] = None, buf: Optional[Tensor] = None) -> Tensor:
            o, shape = self._slices[name]
            n = 1
            for s in shape:
                n *= s
            t = (self._flat if buf is None else buf)[o:o + n].view(shape)
            return t if rows is None else t[rows[0]:rows[1]]

        self._view = view
        self._param_src: list[tuple[nn.Parameter, str, Optional[tuple[int, int]]]] = []

        def param(holder: nn.Module, attr: str, name: str, rows: Optional[tuple[int, int]] = None) -> None:
            p = nn.Parameter(view(name, rows))
            setattr(holder, attr, p)
            self._param_src.append((p, name, rows))

        self.tok_embeddings = _Holder()
        param(self.tok_embeddings, "weight", "emb", (0, vocab_size))
        self.layers = nn.ModuleList()
        for l in range(num_layers):
            layer = _Holder()
            layer.attn = _Holder()
            for nm, rows in (("q_proj", (0, Q)), ("k_proj", (Q, Q + KVD)), ("v_proj", (Q + KVD, Q + 2 * KVD))):
                h = _Holder()
                param(h, "weight", f"L{l}.wqkv", rows)
                setattr(layer.attn, nm, h)
            h = _Holder()
            param(h, "weight", f"L{l}.wo")
            layer.attn.output_proj = h
            layer.mlp = _Holder()
            for nm, src, rows in (("w1", "w13", (0, I)), ("w2", "w2", None), ("w3", "w13", (I, 2 * I))):
                h = _Holder()
                param(h, "weight", f"L{l}.{src}", rows)
                setattr(layer.mlp, nm, h)
            layer.sa_norm = _Holder()
            param(layer.sa_norm, "scale", f"L{l}.sa_norm")
            layer.mlp_norm = _Holder()
            param(layer.mlp_norm, "scale", f"L{l}.mlp_norm")
            self.layers.append(layer)
        self.norm = _Holder()
        param(self.norm, "scale", "norm")
        with torch.no_grad():
            for p, name, _ in self._param_src:
                if name.endswith("norm"):
                    p.fill_(1.0)

        cache_len = min(rope_cache_len or max_seq_len, max_seq_len)
        self._rope = llama3_rope_table(self.head_dim, cache_len, rope_base, scale_factor).to(device)
        self._arena = _Arena(device)
        self._anchor = torch.zeros(1, dtype=torch.float32, device=device, requires_grad=True)
        self._saved: Optional[dict] = None       # activations of the forward awaiting its backward
        self._fwd_generation = 0
        self.pending_grad_scale: Optional[Tensor] = None  # lazy scale_grads (device fp32 scalar), consumed by the optimizer
        # Gradient buffer protocol: nothing zeroes the 2.5 GB buffer between optimizer steps.  `_grads_dirty` = the buffer holds the
        # gradients of this accumulation window, the next backward ADDS; not dirty = the next backward WRITES every element (weight
        # gradients by the GEMM's plain epilogue, norm scales by assignment, the tied embedding by the head's weight gradient before the
        # scatter-add), so whatever is in the buffer (`_grads_stale`: left-overs of the last window) never has to be cleared.
        self._grads_dirty = False
        self._grads_stale = False
        self._emb_grad_written = False
        self.label_errors: Optional[Tensor] = None        # device count of out-of-range labels seen by the last fused loss
        self.bucket_listener = None                       # callable(name, lo, hi) for the NEXT gradient-exchanging backward (see _backward_hidden)
        self.position_errors: Optional[Tensor] = None     # device count of input_pos entries outside the RoPE table in the last forward
        self.grad_sync = None                             # optional ssi.distributed.GradSync
        self.sync_this_backward = False

    # ---- nn.Module plumbing ------------------------------------------------------------------------------------------
    def _apply(self, fn, recurse=True):
        probe = fn(torch.empty(0, dtype=self.dtype, device=self.device))
        if probe.device != self.device or probe.dtype != self.dtype:
            raise NotImplementedError("HipLlamaDecoder is created on its target device/dtype; .to() cannot move it")
        return self

    def set_num_output_chunks(self, num_output_chunks: int) -> None:
        """Kept for API parity (``trainer.py:304``).  The fused loss path never materialises per-chunk logits lists;
        ``forward`` still returns ``num_output_chunks`` sequence chunks when asked for logits."""
        self.num_output_chunks = int(num_output_chunks)

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        own = dict(self.named_parameters())
        missing = [k for k in own if k not in state_dict]
        unexpected = [k for k in state_dict if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing}, unexpected keys {unexpected}")
        with torch.no_grad():
            for k, p in own.items():
                if k in state_dict:
                    src = state_dict[k]
                    if tuple(src.shape) != tuple(p.shape):
                        raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(src.shape)} vs model {tuple(p.shape)}")
                    p.copy_(src.to(device=self.device, dtype=self.dtype))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # Data gradients dX = dY W run on the untransposed weights (NN form of the persistent GEMM).  `dgrad_transposed` names weights whose
    # data gradient should instead run in the k-contiguous (NT) form on an [in, out] copy refreshed after every optimizer step: none
    # by default.  Measured inside the step (profiles/r02_b, bench.py per-class GEMM timing): W13's data gradient takes 0.701 ms as NN
    # and 0.696 ms as NT, while the 16 transposes cost 0.75 ms — alone on the GPU the NN form looks 17 % slower (913 vs 779 us), but in the
    # step its A operand has just been written and is served from the Infinity Cache.  SSI_DGRAD_NT=1: copies of every 2-D weight.
    dgrad_transposed: tuple = {"1": ("emb", "wqkv", "wo", "w13", "w2"), "w13": ("w13",)}.get(os.environ.get("SSI_DGRAD_NT", ""), ())

    @property
    def transposed_weight_copies(self) -> bool:
        return len(self.dgrad_transposed) > 0

    @transposed_weight_copies.setter
    def transposed_weight_copies(self, on: bool) -> None:  # tests: all weights or none
        self.dgrad_transposed = ("emb", "wqkv", "wo", "w13", "w2") if on else ()
        self._wt, self._wt_key = {}, None

    def _has_t(self, name: str) -> bool:
        return name.rsplit(".", 1)[-1] in self.dgrad_transposed and self._mfma_shapes()

    def _view_t(self, name: str) -> Tensor:
        return self._wt[name]

    def _ensure_transposed(self) -> None:
        """Refresh the [in, out] copies if any weight changed since the last refresh."""
        if not self.dgrad_transposed or not self._mfma_shapes():
            return
        key = (self._flat._version, self._hip_epoch)
        if key == self._wt_key:
            return
        for name, (o, shape) in self._slices.items():
            if len(shape) == 2 and self._has_t(name):
                if name not in self._wt:
                    self._wt[name] = torch.empty((shape[1], shape[0]), dtype=self.dtype, device=self.device)
                ops.transpose(self._view(name), self._wt[name])
        self._wt_key = key

    def attach_grads(self) -> None:
        """Point every ``p.grad`` at its slice of the flat gradient buffer (idempotent)."""
        for p, name, rows in self._param_src:
            if p.grad is None:
                p.grad = self._view(name, rows, self._flat_grad)

    # SSI_ZERO_GRADS=1 (A/B runs): the round-1 protocol — AdamW zeroes the buffer in its pass and every backward accumulates
    always_accumulate = os.environ.get("SSI_ZERO_GRADS", "0") == "1"

    def zero_grad(self, set_to_none: bool = True) -> None:
        """With ``set_to_none`` (torch's default) no memory is touched: the next backward overwrites the buffer.  Only a caller who
        keeps the ``p.grad`` views and wants to read zeros pays for a memset."""
        if set_to_none and not self.always_accumulate:
            self._grads_stale = self._grads_stale or self._grads_dirty
            for p, _, _ in self._param_src:
                p.grad = None
        elif self._grads_dirty or self._grads_stale:
            self._flat_grad.zero_()
            self._grads_stale = False
        if set_to_none and self.always_accumulate:
            for p, _, _ in self._param_src:
                p.grad = None
        self._grads_dirty = False
        self._emb_grad_written = False  # a head backward whose decoder backward never ran (exception in between) leaves it set
        self.pending_grad_scale = None

    def _window_accumulates(self) -> bool:
        """Does the backward about to run ADD to the gradient buffer (True) or WRITE it (False: first backward of a window)?

        ``_grads_dirty`` alone is not enough: an optimizer the model does not know (``torch.optim.AdamW`` on the parameter views,
        INTEGRATION.md level 2) ends a window with ``optimizer.zero_grad(set_to_none=True)``, which drops every ``p.grad`` without telling
        the model.  Every ``p.grad is None`` therefore opens a new window whatever the flag says — ``torch``'s own meaning of "no
        gradient yet".  Some ``None`` and some not (a caller cleared a subset): those parameters' slices are zeroed and the backward adds,
        which is what autograd's accumulation would give."""
        if self.always_accumulate:
            return True
        if not self._grads_dirty:
            return False
        dropped = [(p, name, rows) for p, name, rows in self._param_src if p.grad is None]
        if not dropped:
            return True
        if len(dropped) == len(self._param_src):
            self._grads_dirty, self._grads_stale, self._emb_grad_written = False, True, False
            self.pending_grad_scale = None
            return False
        for p, name, rows in dropped:  # re-attached at once: the decoder backward behind a head backward asks again
            p.grad = self._view(name, rows, self._flat_grad).zero_()
        return True

    def _finish_pending_exchange(self) -> None:
        """A data-parallel bucket left in flight (``GradSync.finish(defer_last=True)``) must have landed before a backward writes the
        gradient buffer again; ``HipAdamW.step`` normally does this, a caller that skipped the step must not race the reduction."""
        sync = self.grad_sync
        if sync is not None and hasattr(sync, "finish_deferred"):
            sync.finish_deferred()

    def activation_bytes(self) -> int:
        return self._arena.bytes()

    # ---- batch geometry ----------------------------------------------------------------------------------------------
    def _mfma_shapes(self) -> bool:
        D, I = self.embed_dim, self.intermediate_dim
        return (self.dtype == torch.bfloat16 and D % 256 == 0 and self.qkv_dim % 256 == 0 and (2 * I) % 256 == 0
                and I % 64 == 0 and (self.num_heads * self.head_dim) % 64 == 0)

    def padded_seq_len(self, batch: int, seq: int) -> int:
        """Sequence length after right-padding so that batch*seq fills whole 256-row MFMA tiles.  Right padding is the
        reference's own batch format (pad_id / -100, ``ssi/data/__init__.py:174-199``): causal attention and the ignored
        labels make it numerically inert for the real positions."""
        if not self._mfma_shapes():
            return seq
        g = 256 // math.gcd(batch, 256)
        g = max(g, 128)  # 128: the MFMA attention kernels walk 128-key groups (a ragged batch maximum would otherwise drop to the
        return _align(seq, g)  # generic attention kernels, 5x slower end to end)

    # ---- forward: decoder stack --------------------------------------------------------------------------------------
    @staticmethod
    def _document_ranges(input_pos: Tensor, max_pos: Optional[int] = None) -> tuple[Tensor, Tensor, Tensor]:
        """Torch restatement of ``ops.doc_ranges`` (kept as the checker of that kernel's test; the forward uses the kernel).
        Packed rows (torchtune ``PackedDataset``: ``input_pos`` restarts at 0 with every document of a row) -> int32 [B*S]
        (positions, doc_start, doc_end): a query attends to the keys doc_start <= key <= query of its own row, which is the
        block-causal mask ``padded_collate_packed`` builds from ``seq_lens``.  The padding tail of a pack continues the last
        document's positions and so joins it: no real query sees those keys (causal) and their labels are ignored."""
        B, S = input_pos.shape
        idx = torch.arange(S, device=input_pos.device, dtype=torch.int64).expand(B, S)
        is_start = input_pos == 0
        is_start[:, 0] = True
        doc_start = torch.where(is_start, idx, torch.zeros_like(idx)).cummax(dim=1).values
        nxt = torch.where(is_start, idx, torch.full_like(idx, S))
        nxt = torch.cat([nxt[:, 1:], torch.full_like(nxt[:, :1], S)], dim=1)  # first document start strictly after s
        doc_end = nxt.flip(1).cummin(dim=1).values.flip(1)
        as32 = lambda t: t.to(torch.int32).reshape(-1).contiguous()  # noqa: E731
        return as32(input_pos if max_pos is None else input_pos.clamp(max=max_pos)), as32(doc_start), as32(doc_end)

    # SSI_SPLITK_NT=0 (A/B runs): keep the k-contiguous / data-gradient GEMMs unsplit whatever their output grid
    split_small_grids = os.environ.get("SSI_SPLITK_NT", "1") != "0"

    # SSI_TAIL_SPLIT=0 (A/B runs): a ragged grid is K-split as a whole (the round-4 form) instead of in its last partial round only
    split_tail_only = os.environ.get("SSI_TAIL_SPLIT", "1") != "0"

    def _gemm(self, layout: int, a: Tensor, b: Tensor, c: Tensor, residual: Optional[Tensor] = None) -> None:
        """``ops.gemm`` for the forward projections and the data gradients, with K split where the 256 x 256 output grid leaves CUs idle:
        at the reference's default micro-batches (``conf/data/_sft_base.yaml:21``: 2 x 2048 = 4096 rows) W_o, W2 and the data gradients of
        the N = 2048 projections have 128 tiles for 256 CUs, and a ragged packed length fills its last round badly.  ``ops.splitk_choice``
        weighs the halved rounds against the fp32 partial tiles' traffic (K = 2048: a draw; K = 8192 / 16 384: 1.6-1.8 x).

        Round 5: a grid of MORE than one round (a right-padded batch after the unpadding: T' = 11 520 rows give the N = 2048 projections 360
        tiles = 1.4 rounds) is split in its last, partial round only: the m-tile rows that fill whole rounds run unsplit, the rows behind
        them as a second launch with its own K split — the same time on the matrix cores as the split of the whole grid (1.5 tile times
        either way at 1.4 rounds), but fp32 partial tiles and their reduction pass for 104 tiles instead of 360."""
        M, N = c.shape
        K = a.shape[1]
        mfma = self.split_small_grids and self.dtype == torch.bfloat16 and self._mfma_shapes() and M % 256 == 0 and N % 256 == 0
        if mfma and self.split_tail_only and layout in (GEMM_NT, GEMM_NN):
            tm, tn = M // 256, N // 256
            rounds = (tm * tn) // 256
            m_main = (rounds * 256) // tn  # m-tile rows that fill whole rounds of the 256 CUs
            if rounds >= 1 and 0 < m_main < tm and (tm - m_main) * tn <= 192:
                rows = m_main * 256
                ops.gemm(layout, a[:rows], b, c[:rows], residual=None if residual is None else residual[:rows])
                self._gemm(layout, a[rows:], b, c[rows:], None if residual is None else residual[rows:])  # (fewer than 256 tiles: split as a whole)
                return
        splits = ops.splitk_choice(M, N, K) if mfma else 1
        if splits > 1:
            wsk = self._arena.get("ws.splitk.nt", (splits * M * N,), torch.float32)
            ops.gemm_splitk(layout, a, b, c, splits, wsk, residual=residual)
        else:
            ops.gemm(layout, a, b, c, residual=residual)

    def build_attn_plan(self, input_pos: Tensor, force: bool = False):
        """Work plan of the attention backward for a packed batch (``ssi/attn_plan.py``) from a HOST ``input_pos`` [B, S] — what the data layer
        calls in its prefetch thread; ``None`` where the pipelined kernels do not apply (device tensor, fp32 model, other head ratios, a length
        the model would still have to pad, positions that are not document-relative, batches the library leaves to the round-1..3 kernels)."""
        from . import attn_plan
        if input_pos is None or input_pos.is_cuda or input_pos.dim() != 2 or not self._mfma_shapes() or self.head_dim != 64:
            return None
        B, S = input_pos.shape
        if self.num_heads != 4 * self.num_kv_heads or self.padded_seq_len(B, S) != S or S % 128:
            return None
        return attn_plan.plan_from_input_pos(input_pos, self.num_heads, self.num_kv_heads, force=force)

    # SSI_PLAIN_PLAN=1 (A/B runs): plain causal rows with a work plan too instead of the dispatcher's fixed patterns.  Measured and left off:
    # 0.2-0.3 % slower in the step at 8 x 2048, 2 x 2048, 16 x 768 and 8 x 4096 (profiles/LAB_NOTES.md, round 5) — equal rows need no balancing
    plan_plain_rows = os.environ.get("SSI_PLAIN_PLAN", "0") == "1"

    def _plain_rows_plan(self, B: int, S: int, device):
        """Work plan for plain causal rows — every row one document — built once per batch shape and kept on the device: the attention
        backward's persistent dQ workgroups then take their query blocks by load (the host's longest-processing-time assignment) and dK / dV
        chunks heavier than the chip's share per compute unit are split over the query heads, whatever B and S are."""
        cache = self.__dict__.setdefault("_plain_plans", {})
        key = (B, S, str(device))
        if key not in cache:
            from . import attn_plan
            plan = None
            if self.head_dim == 64 and self.num_heads == 4 * self.num_kv_heads and S % 128 == 0:
                plan = attn_plan.plan_from_seq_lens([[S]] * B, self.num_heads, self.num_kv_heads)
            cache[key] = plan.to_device(device, non_blocking=False) if plan is not None else None
        return cache[key]

    def _forward_hidden(self, tokens: Tensor, save: bool, input_pos: Optional[Tensor] = None, attn_plan=None) -> Tensor:
        B, S = tokens.shape
        pos = ds = de = None
        self.position_errors = None
        plan = None
        if input_pos is None and save and self.plan_plain_rows and self._mfma_shapes():
            plan = self._plain_rows_plan(B, S, tokens.device)
        if input_pos is not None and save:
            if attn_plan is None and not input_pos.is_cuda:  # host positions: the plan costs no device sync (the trainer's prefetch thread
                attn_plan = self.build_attn_plan(input_pos)   # brings one along with the batch instead)
            if attn_plan is not None and attn_plan.matches(B, S, self.num_heads, self.num_kv_heads) and self._mfma_shapes():
                plan = attn_plan.to_device(tokens.device)
        if input_pos is not None:
            if input_pos.shape != tokens.shape:
                raise ValueError("input_pos must have the shape of tokens")
            if not input_pos.is_cuda and int(input_pos.max()) >= self._rope.shape[0]:  # host tensor: checking costs no device sync
                raise ValueError("input_pos exceeds the RoPE cache")
            # device tensor: no blocking read-back in the step; positions are clamped to the cache instead (the data layer bounds
            # them by tokenizer.max_seq_len <= rope cache, ssi/data/packed.py), so the kernels never index past the table
            # and the clamped positions are COUNTED (position_errors): the trainer raises on them from its one read-back per micro-batch
            self.position_errors = torch.zeros(1, dtype=torch.int32, device=tokens.device)
            pos, ds, de = ops.doc_ranges(input_pos.to(tokens.device), self._rope.shape[0] - 1, self.position_errors)  # one launch
        T, D, I = B * S, self.embed_dim, self.intermediate_dim
        H, KV, hd, dt, A = self.num_heads, self.num_kv_heads, self.head_dim, self.dtype, self._arena
        if pos is None and S > self._rope.shape[0]:  # implicit positions 0 .. S-1; with input_pos the table is indexed by (clamped) positions
            raise ValueError(f"sequence length {S} exceeds the RoPE cache ({self._rope.shape[0]})")
        tok = tokens.reshape(-1).contiguous()
        L = self.num_layers
        # a forward that saves nothing (eval, no_grad) writes ".x" buffers only: it may run between a training forward and its backward
        h = A.get("h0" if save else "h0.x", (T, D), dt)
        ops.embed_fwd(tok, self._view("emb"), h, self.vocab_size)
        for l in range(L):
            sfx = f"{l}" if save else "x"
            xn1 = A.get("xn1.all", (L, T, D), dt)[l] if save else A.get("xn1.x", (T, D), dt)  # saved per layer in ONE buffer: batched wgrads
            rstd1 = A.get(f"rstd1.{sfx}", (T,), torch.float32)
            ops.rmsnorm_fwd(h, self._view(f"L{l}.sa_norm"), xn1, rstd1, self.norm_eps)
            qkv = A.get(f"qkv.{sfx}", (T, self.qkv_dim), dt)
            ops.gemm_rope(xn1, self._view(f"L{l}.wqkv"), qkv, S, H + KV, hd, self._rope, positions=pos)  # RoPE in the GEMM epilogue
            att = A.get("att.all", (L, T, H * hd), dt)[l] if save else A.get("att.x", (T, H * hd), dt)
            lse = A.get(f"lse.{sfx}", (B * H * S,), torch.float32)
            ops.attn_fwd(qkv, att, lse, B, S, H, KV, hd, ds, de)
            hmid = A.get(f"hmid.{sfx}", (T, D), dt)
            self._gemm(GEMM_NT, att, self._view(f"L{l}.wo"), hmid, residual=h)
            xn2 = A.get(f"xn2.{sfx}", (T, D), dt)
            rstd2 = A.get(f"rstd2.{sfx}", (T,), torch.float32)
            ops.rmsnorm_fwd(hmid, self._view(f"L{l}.mlp_norm"), xn2, rstd2, self.norm_eps)
            gu = A.get(f"gu.{sfx}", (T, 2 * I), dt)
            act = A.get(f"act.{sfx}", (T, I), dt)
            ops.gemm_swiglu_fwd(xn2, self._view(f"L{l}.w13"), gu, act)  # SwiGLU rides in the GEMM epilogue
            hn_name = f"h{l + 1}" if save else f"hx{l & 1}"
            hnext = A.get(hn_name, (T, D), dt)
            self._gemm(GEMM_NT, act, self._view(f"L{l}.w2"), hnext, residual=hmid)
            h = hnext
        hn = A.get("hn" if save else "hn.x", (T, D), dt)
        rstdf = A.get("rstdf" if save else "rstdf.x", (T,), torch.float32)
        ops.rmsnorm_fwd(h, self.norm.scale, hn, rstdf, self.norm_eps)
        if save:
            self._fwd_generation += 1
            self._saved = {"tok": tok, "B": B, "S": S, "gen": self._fwd_generation, "pos": pos, "ds": ds, "de": de, "plan": plan}
        return hn

    # ---- backward: decoder stack -------------------------------------------------------------------------------------
    def _backward_hidden(self, d_hn: Tensor, gen: int) -> None:
        sv = self._saved
        if sv is None or sv["gen"] != gen:
            raise RuntimeError("HipLlamaDecoder: backward called for a forward whose activations were overwritten; "
                               "run backward before the next training forward")
        B, S, tok = sv["B"], sv["S"], sv["tok"]
        pos, ds, de, plan = sv["pos"], sv["ds"], sv["de"], sv.get("plan")
        T, D, I = B * S, self.embed_dim, self.intermediate_dim
        H, KV, hd, dt, A = self.num_heads, self.num_kv_heads, self.head_dim, self.dtype, self._arena
        L = self.num_layers
        G = self._flat_grad
        self._finish_pending_exchange()   # first: _window_accumulates() may zero slices of the buffer a deferred bucket is still reduced into
        acc = self._window_accumulates()  # False: first backward of the window, every gradient is written, not added
        gv = lambda name: self._view(name, None, G)  # noqa: E731
        ws = A.get("ws.rms", (max(ops.rmsnorm_bwd_workspace_bytes(T, D), 16),), torch.uint8)

        def wgrad(dy: Tensor, x: Tensor, name: str) -> None:
            """dW += dy^T x; split over K = tokens when the [out, in] grid cannot fill the chip (square projections)."""
            g = gv(name)
            splits = ops.splitk_choice(g.shape[0], g.shape[1], T) if (dt == torch.bfloat16 and T % 64 == 0) else 1
            if splits > 1:
                wsk = A.get("ws.splitk", (splits * g.shape[0] * g.shape[1],), torch.float32)
                ops.gemm_splitk(GEMM_TN, dy, x, g, splits, wsk, accumulate=acc)
            else:
                ops.gemm(GEMM_TN, dy, x, g, accumulate=acc)

        sync = self.grad_sync if (self.grad_sync is not None and self.sync_this_backward) else None
        self._ensure_transposed()
        # who hears that a bucket's gradients are final: the data-parallel exchange, and / or an optimizer that updates the bucket's parameters
        # under the rest of this backward (HipAdamW.overlap_with_backward, one shot: set for the window's last backward)
        listener, self.bucket_listener = (self.bucket_listener if self.sync_this_backward else None), None

        def announce(name: str) -> None:
            if sync:
                sync.bucket_ready(*self._bucket_by_name[name])
            if listener is not None:
                listener(*self._bucket_by_name[name])

        # Weight gradients of the attention projections (dW_o = d hmid^T att: 64 output tiles; dW_qkv = d qkv^T xn1: 96): deferred until the
        # lowest layer of a group has produced its d qkv, then ONE batched launch per weight covers the group at full K (8 layers: 512 and
        # 768 tiles = 2 and 3 rounds of the 256 CUs) instead of a split-K launch + reduction per layer and weight.  Costs G x T x (D + qkv)
        # elements of HBM for the group's output gradients (1.3 GB at the 1B shape, B x S = 16384) — memory this GPU has.
        Gp = self.wgrad_group
        if sync and Gp != self._bucket_group:
            raise RuntimeError("wgrad_group was changed after construction: the data-parallel buckets were laid out for the old grouping")
        defer = (Gp > 1 and dt == torch.bfloat16 and self._mfma_shapes() and T % 128 == 0 and T >= 256
                 and ops.splitk_choice(D, H * hd, T) > 1 and ops.splitk_choice(self.qkv_dim, D, T) > 1)
        xn1_all, att_all = A.get("xn1.all", (L, T, D), dt), A.get("att.all", (L, T, H * hd), dt)
        if defer:
            dhmid_all, dqkv_all = A.get("dh.b.all", (Gp, T, D), dt), A.get("dqkv.all", (Gp, T, self.qkv_dim), dt)
            lstride = (self._slices["L1.wqkv"][0] - self._slices["L0.wqkv"][0]) if L > 1 else 0

        def flush_attention_wgrads(l_lo: int) -> None:
            n = min(Gp, L - l_lo)
            g_o = torch.as_strided(G, (n, D, H * hd), (lstride, H * hd, 1), self._slices[f"L{l_lo}.wo"][0])
            g_qkv = torch.as_strided(G, (n, self.qkv_dim, D), (lstride, D, 1), self._slices[f"L{l_lo}.wqkv"][0])
            ops.gemm_batched(GEMM_TN, dhmid_all[:n], att_all[l_lo:l_lo + n], g_o, accumulate=acc)
            ops.gemm_batched(GEMM_TN, dqkv_all[:n], xn1_all[l_lo:l_lo + n], g_qkv, accumulate=acc)

        def dgrad(dy: Tensor, name: str, dx: Tensor) -> None:
            """dx = dy @ W  (W = [out, in]); NT form on the [in, out] copy where one is kept."""
            if self._has_t(name):
                self._gemm(GEMM_NT, dy, self._view_t(name), dx)
            else:
                self._gemm(GEMM_NN, dy, self._view(name), dx)

        # small micro-batches: dK / dV per query head + a reduction, in a workspace of the arena (0 bytes = the launch fills the chip as it is)
        ws_bytes = ops.attn_bwd_workspace_bytes(B, S, H, KV, hd, dt) if T > 0 else 0
        if plan is not None:  # (partial rows of the dK/dV chunks a plan splits over the query heads)
            ws_bytes = max(ws_bytes, plan.workspace_bytes)
        attn_ws = A.get("ws.attn", (ws_bytes,), torch.uint8) if ws_bytes else None
        dh = A.get("dh.a", (T, D), dt)
        ops.rmsnorm_bwd(d_hn, A.get(f"h{L}", (T, D), dt), self.norm.scale, A.get("rstdf", (T,), torch.float32), None, dh,
                        gv("norm"), ws, accumulate=acc)
        announce("norm")
        for l in reversed(range(L)):
            xn1, xn2 = xn1_all[l], A.get(f"xn2.{l}", (T, D), dt)
            qkv, att = A.get(f"qkv.{l}", (T, self.qkv_dim), dt), att_all[l]
            hmid, gu, act = A.get(f"hmid.{l}", (T, D), dt), A.get(f"gu.{l}", (T, 2 * I), dt), A.get(f"act.{l}", (T, I), dt)
            h_in = A.get(f"h{l}", (T, D), dt)
            # MLP: h_out = hmid + act @ w2^T
            wgrad(dh, act, f"L{l}.w2")
            dgu = A.get("dgu", (T, 2 * I), dt)
            if self._has_t(f"L{l}.w2"):  # d act = dh W2 never reaches memory: the SwiGLU backward rides in the GEMM epilogue
                ops.gemm_swiglu_bwd(GEMM_NT, dh, self._view_t(f"L{l}.w2"), gu, dgu, None)
            else:
                ops.gemm_swiglu_bwd(GEMM_NN, dh, self._view(f"L{l}.w2"), gu, dgu, A.get("dact", (T, I), dt))
            dxn = A.get("dxn", (T, D), dt)
            dgrad(dgu, f"L{l}.w13", dxn)
            wgrad(dgu, xn2, f"L{l}.w13")
            dhmid = dhmid_all[l % Gp] if defer else A.get("dh.b", (T, D), dt)
            ops.rmsnorm_bwd(dxn, hmid, self._view(f"L{l}.mlp_norm"), A.get(f"rstd2.{l}", (T,), torch.float32), dh, dhmid,
                            gv(f"L{l}.mlp_norm"), ws, accumulate=acc)
            # attention: hmid = h_in + att @ wo^T
            datt = A.get("datt", (T, H * hd), dt)
            dgrad(dhmid, f"L{l}.wo", datt)
            if not defer:
                wgrad(dhmid, att, f"L{l}.wo")
            dqkv = dqkv_all[l % Gp] if defer else A.get("dqkv", (T, self.qkv_dim), dt)
            delta = A.get("delta", (B * H * S,), torch.float32)
            # attention backward with the backward of the RoPE rotation fused into its epilogues: dqkv arrives in pre-RoPE space
            ops.attn_bwd(qkv, att, datt, A.get(f"lse.{l}", (B * H * S,), torch.float32), dqkv, delta, B, S, H, KV, hd, ds, de,
                         rope_table=self._rope, positions=pos, workspace=attn_ws, plan=plan)
            dgrad(dqkv, f"L{l}.wqkv", dxn)
            if not defer:
                wgrad(dqkv, xn1, f"L{l}.wqkv")
            ops.rmsnorm_bwd(dxn, h_in, self._view(f"L{l}.sa_norm"), A.get(f"rstd1.{l}", (T,), torch.float32), dhmid, dh,
                            gv(f"L{l}.sa_norm"), ws, accumulate=acc)
            announce(f"L{l}.mlp")
            if defer and l % Gp == 0:
                flush_attention_wgrads(l)
            if l % self._bucket_group == 0:
                announce(f"attn.{l}")
        ws_e = A.get("ws.emb", (max(_lib.load().ssi_embed_bwd_workspace_bytes(self.vocab_size), 16),), torch.uint8)
        if not acc and not self._emb_grad_written:  # backward through the hidden states only (no tied head in front): the scatter-add
            gv("emb").zero_()                       # below touches only the rows of this batch's tokens
        ops.embed_bwd(tok, dh, gv("emb"), self.vocab_size, ws_e)
        self._grads_dirty, self._grads_stale, self._emb_grad_written = True, False, False
        announce("emb")
        self._saved = None
        self.attach_grads()

    # ---- tied LM head ------------------------------------------------------------------------------------------------
    def _head_logits(self, hn: Tensor, arena_name: Optional[str] = None) -> Tensor:
        """logits [T, vocab_pad] = hn E^T.  Callers of the public ``forward`` get a FRESH tensor (as torchtune returns): logits kept
        from one batch must survive the next forward.  Only the fused loss passes an arena name (its buffer never leaves the model)."""
        T = hn.shape[0]
        if arena_name is None:
            logits = torch.empty((T, self.vocab_pad), dtype=self.dtype, device=self.device)
        else:
            logits = self._arena.get(arena_name, (T, self.vocab_pad), self.dtype)
        ops.gemm(GEMM_NT, hn, self._view("emb"), logits)
        return logits

    def _head_backward(self, dlogits: Tensor, hn: Tensor, alpha_dev: Optional[Tensor]) -> Tensor:
        """d_hn = alpha * dlogits @ E ;  dE += alpha * dlogits^T @ hn   (dlogits: [T, vocab_pad], pad columns zero)."""
        T, D = hn.shape
        self._finish_pending_exchange()  # before anything may touch the gradient buffer (see _backward_hidden)
        acc = self._window_accumulates()
        d_hn = self._arena.get("d_hn", (T, D), self.dtype)
        self._ensure_transposed()
        if self._has_t("emb"):
            ops.gemm(GEMM_NT, dlogits, self._view_t("emb"), d_hn, alpha_dev=alpha_dev)
        else:
            ops.gemm(GEMM_NN, dlogits, self._view("emb"), d_hn, alpha_dev=alpha_dev)
        g_emb = self._view("emb", None, self._flat_grad)
        # dE = dlogits^T hn has vocab_pad / 256 x D / 256 output tiles (521 x 8 = 4168 at V = 133 258): 16 full rounds of the 256 CUs and a
        # 17th with 72 tiles, each a full K = T contraction (~360 us with 184 CUs idle).  The rows of the last partial round go to a second
        # launch that also splits K, so that the tail occupies the chip for a third of a tile's time.
        rows_main = self._head_wgrad_main_rows(T)
        if rows_main:
            Vp = self.vocab_pad
            ops.gemm(GEMM_TN, dlogits[:, :rows_main], hn, g_emb[:rows_main], alpha_dev=alpha_dev, accumulate=acc)
            tail, splits = Vp - rows_main, self._head_wgrad_tail_splits
            wsk = self._arena.get("ws.splitk.head", (splits * tail * D,), torch.float32)
            ops.gemm_splitk(GEMM_TN, dlogits[:, rows_main:], hn, g_emb[rows_main:], splits, wsk, alpha_dev=alpha_dev, accumulate=acc)
        else:
            ops.gemm(GEMM_TN, dlogits, hn, g_emb, alpha_dev=alpha_dev, accumulate=acc)
        self._emb_grad_written = True  # the decoder backward that follows adds the token rows on top and closes the window
        return d_hn

    _head_wgrad_tail_splits = 3

    def _head_wgrad_main_rows(self, T: int, n_cu: int = 256) -> int:
        """Rows of the embedding gradient computed by the unsplit launch (whole rounds of the CUs); 0 = one launch for everything (not the
        MFMA path, a last round at least half full, or a tail whose split units would not fit one round)."""
        if os.environ.get("SSI_HEAD_WGRAD_TAIL", "1") == "0" or self.dtype != torch.bfloat16 or not self._mfma_shapes() or T % 128 or T < 64 * 6 * self._head_wgrad_tail_splits:
            return 0
        tm, tn = self.vocab_pad // 256, self.embed_dim // 256
        full_rounds = (tm * tn) // n_cu
        if full_rounds == 0 or (full_rounds * n_cu) % tn:
            return 0
        tail_tiles = tm * tn - full_rounds * n_cu
        if tail_tiles == 0 or 2 * tail_tiles >= n_cu or tail_tiles * self._head_wgrad_tail_splits > n_cu:
            return 0
        return full_rounds * n_cu // tn * 256

    # ---- public API --------------------------------------------------------------------------------------------------
    def _check_inputs(self, tokens: Tensor, mask, encoder_input, encoder_mask, input_pos) -> Tensor:
        if encoder_input is not None or encoder_mask is not None:
            raise NotImplementedError("no encoder inputs: the reference's model is decoder-only (ssi/model.py:34)")
        if mask is not None and input_pos is None:
            raise NotImplementedError("a dense attention mask is never materialised: packed batches pass input_pos, from which "
                                      "the block-causal mask follows (ssi/data/packed.py)")
        # mask together with input_pos (torchtune's padded_collate_packed emits both): the block-causal structure is re-derived
        # from input_pos; the dense [B, S, S] tensor itself is not read
        if tokens.dim() != 2 or tokens.dtype != torch.int64:
            raise ValueError("tokens must be an int64 tensor of shape [batch, seq]")
        if not tokens.is_cuda:
            raise _lib.HipLibraryError("tokens must live on the GPU (no CPU fallback)")
        return tokens

    def forward_hidden(self, tokens: Tensor, input_pos: Optional[Tensor] = None, attn_plan=None) -> Tensor:
        """Final-normed hidden states [B, S, D] (autograd-aware)."""
        B, S = tokens.shape
        if torch.is_grad_enabled() and self.training:
            hn = _DecoderFn.apply(self, tokens, self._anchor, input_pos, attn_plan)
        else:
            hn = self._forward_hidden(tokens, save=False, input_pos=input_pos)
        return hn.view(B, S, self.embed_dim)

    def forward(self, tokens: Tensor, mask=None, encoder_input=None, encoder_mask=None, input_pos=None):
        """Logits.  ``num_output_chunks > 0``: list of [B, ceil(S/n), V] chunks in the model dtype (torch.chunk rule,
        SURVEY.md Appendix A.4); else one fp32 [B, S, V] tensor — as torchtune's ``TransformerDecoder.forward``."""
        tokens = self._check_inputs(tokens, mask, encoder_input, encoder_mask, input_pos)
        B, S0 = tokens.shape
        S = self.padded_seq_len(B, S0)
        if S != S0:  # right-pad to whole MFMA tiles (inert under causal attention), slice the logits back below
            tokens = torch.cat([tokens, torch.zeros(B, S - S0, dtype=tokens.dtype, device=tokens.device)], dim=1)
            if input_pos is not None:
                cont = input_pos[:, -1:].to(tokens.device) + torch.arange(1, S - S0 + 1, device=tokens.device)
                input_pos = torch.cat([input_pos.to(tokens.device), cont.clamp_(max=self._rope.shape[0] - 1)], dim=1)
        hn = self.forward_hidden(tokens, input_pos).view(B * S, self.embed_dim)
        if torch.is_grad_enabled() and self.training:
            logits = _HeadLogitsFn.apply(self, hn, self._anchor)
        else:
            logits = self._head_logits(hn)
        logits = logits.view(B, S, self.vocab_pad)[:, :S0, : self.vocab_size]
        if self.num_output_chunks > 0:
            return list(logits.chunk(self.num_output_chunks, dim=1))
        return logits.float()

    def fused_loss(self, tokens: Tensor, shifted_labels: Tensor, ignore_index: int = CROSS_ENTROPY_IGNORE_IDX,
                   input_pos: Optional[Tensor] = None, attn_plan=None, loss_weights: Optional[Tensor] = None) -> Tensor:
        """Mean NLL over non-ignored (already shifted) labels with the LM head + CE fused: equals
        ``CEWithChunkedOutputLoss()(model(tokens, input_pos=...), shifted_labels)`` of the reference for any chunk count.
        ``input_pos`` ([B, S], restarting at 0 with every document): packed rows, block-causal attention.  ``attn_plan``
        (``build_attn_plan(input_pos)`` made on the host beside the batch): the attention backward then runs its pipelined kernels on the
        packed rows; without one (and with ``input_pos`` on the device) the round-1..3 kernels run — same results to rounding.
        ``loss_weights`` (fp32 ``[B, S]``, >= 0, aligned with ``shifted_labels``): the result is ``sum_i w_i nll_i / n_valid`` — how an
        accumulation window that runs as one batch keeps the reference's per-micro-batch normalisation (``ssi/data/window.py``)."""
        tokens = self._check_inputs(tokens, None, None, None, input_pos)
        B, S = tokens.shape
        if loss_weights is not None:
            if loss_weights.shape != tokens.shape:
                raise ValueError(f"loss_weights {tuple(loss_weights.shape)} vs tokens {tuple(tokens.shape)}")
            loss_weights = loss_weights.to(device=tokens.device, dtype=torch.float32)
        Sp = self.padded_seq_len(B, S)
        if Sp != S:
            pad_t = torch.zeros(B, Sp - S, dtype=tokens.dtype, device=tokens.device)
            tokens = torch.cat([tokens, pad_t], dim=1)
            shifted_labels = torch.cat([shifted_labels, torch.full_like(pad_t, ignore_index)], dim=1)
            if loss_weights is not None:
                loss_weights = torch.cat([loss_weights, torch.ones(B, Sp - S, dtype=torch.float32, device=tokens.device)], dim=1)
            if input_pos is not None:  # the tail continues the last document (as PackedDataset pads a pack)
                cont = input_pos[:, -1:].to(tokens.device) + torch.arange(1, Sp - S + 1, device=tokens.device)
                input_pos = torch.cat([input_pos.to(tokens.device), cont.clamp_(max=self._rope.shape[0] - 1)], dim=1)
        labels = shifted_labels.reshape(-1).contiguous()
        weights = None if loss_weights is None else loss_weights.reshape(-1).contiguous()
        if torch.is_grad_enabled() and self.training:
            return _FusedLossFn.apply(self, tokens, labels, ignore_index, self._anchor, input_pos, attn_plan, weights)
        hn = self._forward_hidden(tokens, save=False, input_pos=input_pos)
        return self._ce_forward(hn, labels, ignore_index, write_grad=False, weights=weights)[0]

    def _ce_forward(self, hn: Tensor, labels: Tensor, ignore_index: int, write_grad: bool,
                    weights: Optional[Tensor] = None) -> tuple[Tensor, Tensor, Tensor]:
        """Tied head + cross-entropy: (mean loss, stats, logits buffer — which holds softmax - onehot when ``write_grad``).  The same
        three launches as the one-call ABI entry ``ssi_lmhead_ce_fwd`` (``ops.lmhead_ce_fwd``), issued one by one here so that ``bench.py``
        can time the head GEMM on its own."""
        T = hn.shape[0]
        logits = self._head_logits(hn, "logits" if write_grad else "logits.x")
        row_loss = self._arena.get("row_loss" if write_grad else "row_loss.x", (T,), torch.float32)
        ops.ce_fwd(logits, labels, self.vocab_size, ignore_index, row_loss, None, write_grad, row_weight=weights)
        out = torch.empty(4, dtype=torch.float32, device=self.device)
        ops.ce_reduce(row_loss, labels, self.vocab_size, ignore_index, out)
        self.label_errors = out[3]  # device scalar: labels outside [0, vocab); the trainer folds it into its one read-back and raises
        return out[0], out, logits


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: HipLlamaDecoder, tokens: Tensor, anchor: Tensor, input_pos: Optional[Tensor] = None, attn_plan=None) -> Tensor:
        hn = model._forward_hidden(tokens, save=True, input_pos=input_pos, attn_plan=attn_plan)
        ctx.model, ctx.gen = model, model._fwd_generation
        return hn

    @staticmethod
    def backward(ctx, d_hn: Tensor):
        ctx.model._backward_hidden(d_hn.contiguous(), ctx.gen)
        return None, None, torch.zeros_like(ctx.model._anchor), None, None


class _HeadLogitsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model: HipLlamaDecoder, hn: Tensor, anchor: Tensor) -> Tensor:
        ctx.model = model
        ctx.save_for_backward(hn)
        return model._head_logits(hn)

    @staticmethod
    def backward(ctx, dlogits: Tensor):
        (hn,) = ctx.saved_tensors
        m = ctx.model
        dl = dlogits
        if not dl.is_contiguous() or dl.dtype != m.dtype:
            dl = dl.to(m.dtype).contiguous()
        return None, m._head_backward(dl, hn, None), torch.zeros_like(m._anchor)


class _FusedLossFn(torch.autograd.Function):
    """tokens, shifted labels -> scalar loss; backward runs head + decoder backward and accumulates into the flat
    gradient buffer (so ``p.grad`` is populated exactly as after ``loss.backward()`` in the reference)."""

    @staticmethod
    def forward(ctx, model: HipLlamaDecoder, tokens: Tensor, labels: Tensor, ignore_index: int, anchor: Tensor,
                input_pos: Optional[Tensor] = None, attn_plan=None, weights: Optional[Tensor] = None) -> Tensor:
        hn = model._forward_hidden(tokens, save=True, input_pos=input_pos, attn_plan=attn_plan)
        loss, stats, dlogits = model._ce_forward(hn, labels, ignore_index, write_grad=True, weights=weights)
        ctx.model, ctx.gen = model, model._fwd_generation
        ctx.save_for_backward(hn, stats, dlogits)
        return loss.clone()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        hn, stats, dlogits = ctx.saved_tensors
        m = ctx.model
        if m._saved is None or m._saved["gen"] != ctx.gen:  # checked BEFORE the head backward touches the gradient buffer
            raise RuntimeError("HipLlamaDecoder: backward called for a forward whose activations were overwritten; "
                               "run backward before the next training forward")
        # d loss / d logits = (softmax - onehot) / n_valid ; the 1/n_valid and the upstream scalar ride in alpha_dev
        alpha = (grad_out.to(torch.float32).reshape(1) / stats[2:3]).contiguous()
        d_hn = m._head_backward(dlogits, hn, alpha)
        m._backward_hidden(d_hn, ctx.gen)
        return None, None, None, None, torch.zeros_like(m._anchor), None, None, None


# --------------------------------------------------------------------------------------------------------------------
# Reference-named entry point
# --------------------------------------------------------------------------------------------------------------------
def get_dtype(dtype: Any) -> torch.dtype:
    if isinstance(dtype, torch.dtype):
        return dtype
    if dtype is None:
        return torch.float32
    if dtype not in PRECISION_STR_TO_DTYPE:
        raise ValueError(f"Dtype {dtype} must be one of {', '.join(PRECISION_STR_TO_DTYPE)}")
    return PRECISION_STR_TO_DTYPE[dtype]


def get_device(device: Any = None) -> torch.device:
    if device is None:
        device = "cuda" if torch.cuda.is_available() else "cpu"
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        import os
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
    return dev


def validate_expected_param_dtype(named_params, dtype: torch.dtype) -> None:
    for name, param in named_params:
        if param.dtype != dtype:
            raise ValueError(f"Parameter {name} has dtype {param.dtype}, but expected {dtype}")


def setup_llama3_2_1b(cfg, llama_config: ConfigLlama3_2, model_state_dict: Optional[dict[str, Any]],
                      dtype_default: torch.dtype | str | None = None,
                      device_default: torch.device | str | None = None) -> HipLlamaDecoder:
    """Same signature and contract as ``/root/reference/ssi/model.py:18-39``: build the decoder from
    ``llama_config.parameters`` in ``dtype_default`` on ``device_default``, load the (torchtune-key) state dict strictly,
    check every parameter's dtype.  ``cfg.compile`` is accepted and ignored: there is no tracing compiler in this stack."""
    if dtype_default is None:
        dtype_default = torch.get_default_dtype()
    elif isinstance(dtype_default, str):
        dtype_default = get_dtype(cfg.dtype)
    if device_default is None:
        device_default = get_device(None)
    elif isinstance(device_default, str):
        device_default = get_device(cfg.device)
    if cfg is not None and cfg.get("compile", False):
        LOGGER.info("cfg.compile=true ignored: kernels are hand-written HIP, there is nothing to trace-compile")
    model = HipLlamaDecoder(**llama_config.parameters, dtype=dtype_default, device=device_default)
    if model_state_dict is not None:
        model.load_state_dict(model_state_dict)
    validate_expected_param_dtype(model.named_parameters(), dtype=dtype_default)
    return model
