"""Loss of the training step (reference: ``/root/reference/ssi/loss.py:7-22`` + torchtune ``CEWithChunkedOutputLoss``).

``compute_loss(batch, model, loss_fn)`` keeps the reference signature and semantics: forward, shift labels left by one
(last position ignored), mean NLL over the SHIFTED non-ignored labels, ``batch`` not mutated.  When ``model`` is the HIP
decoder and ``loss_fn`` is :class:`CEWithChunkedOutputLoss`, the LM head and the cross-entropy run fused on the GPU
(K8+K9); any other (model, loss_fn) pair takes the literal reference route ``loss_fn(model(...), labels)``."""

from __future__ import annotations

from collections.abc import Callable

import torch
from torch import Tensor

from . import ops
from .constants import CROSS_ENTROPY_IGNORE_IDX


class CEWithChunkedOutputLoss(torch.nn.Module):
    """Same constructor, attributes and call contract as torchtune 0.5.0's class of that name (``trainer.py:300``):
    ``loss_fn(list_of_logit_chunks, labels[B, S]) -> sum NLL / count(labels != ignore_index)``; also accepts one
    ``[N, V]`` tensor with flat labels (``loss.py:17-19``).  The arithmetic is the HIP cross-entropy kernel (fp32
    log-sum-exp per row, deterministic row reduction); logits must be GPU tensors."""

    def __init__(self, num_output_chunks: int = 8, ignore_index: int = CROSS_ENTROPY_IGNORE_IDX):
        super().__init__()
        self.num_output_chunks = num_output_chunks
        self.ignore_index = ignore_index

    def forward(self, logits, labels: Tensor) -> Tensor:
        if isinstance(logits, (list, tuple)):
            label_chunks = [c.reshape(-1) for c in labels.chunk(self.num_output_chunks, dim=1)]
            logit_chunks = [c.reshape(-1, c.size(-1)) for c in logits]
            if len(label_chunks) != len(logit_chunks):
                raise ValueError(f"{len(logit_chunks)} logit chunks vs {len(label_chunks)} label chunks")
            logits2d, labels1d = torch.cat(logit_chunks, dim=0), torch.cat(label_chunks, dim=0)
        else:
            logits2d, labels1d = logits.reshape(-1, logits.size(-1)), labels.reshape(-1)
        return _CrossEntropyFn.apply(logits2d, labels1d.contiguous(), self.ignore_index)


class _CrossEntropyFn(torch.autograd.Function):
    """Stand-alone CE over materialised logits (used when logits come from ``model(...)`` rather than the fused path)."""

    @staticmethod
    def forward(ctx, logits: Tensor, labels: Tensor, ignore_index: int) -> Tensor:
        rows, vocab = logits.shape
        ld = (vocab + 7) // 8 * 8
        work = torch.empty(rows, ld, dtype=logits.dtype, device=logits.device)
        work[:, :vocab].copy_(logits)
        row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        ops.ce_fwd(work, labels, vocab, ignore_index, row_loss, None, ctx.needs_input_grad[0])
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
        ops.ce_reduce(row_loss, labels, vocab, ignore_index, out)
        ctx.vocab = vocab
        ctx.save_for_backward(work, out)
        return out[0].clone()

    @staticmethod
    def backward(ctx, grad_out: Tensor):
        work, out = ctx.saved_tensors
        ops.scale_(work, 1.0, (grad_out.to(torch.float32).reshape(1) / out[2:3]).contiguous())
        return work[:, : ctx.vocab], None, None


def compute_loss(batch: dict[str, Tensor], model, loss_fn: Callable) -> Tensor:
    labels = batch["labels"]
    ignore_index = loss_fn.ignore_index
    labels = torch.hstack((labels[..., 1:], torch.full_like(labels[..., -1:], ignore_index)))  # new tensor: batch untouched
    if (hasattr(model, "fused_loss") and isinstance(loss_fn, CEWithChunkedOutputLoss) and batch.get("encoder_input") is None
            and (batch.get("mask") is None or batch.get("input_pos") is not None)):
        # packed batches (ssi/data/packed.py) carry input_pos: block-causal attention, per-document RoPE positions
        extra = {}
        if batch.get("attn_plan") is not None:  # made on the host beside a packed batch (ssi/attn_plan.py): pipelined attention backward
            extra["attn_plan"] = batch["attn_plan"]
        if batch.get("loss_weights") is not None:  # an accumulation window run as one batch (ssi/data/window.py): weights per SHIFTED label
            extra["loss_weights"] = batch["loss_weights"]
        return model.fused_loss(batch["tokens"], labels, ignore_index, input_pos=batch.get("input_pos"), **extra)
    if batch.get("loss_weights") is not None:
        raise ValueError("loss_weights need the fused LM head + cross-entropy of the HIP decoder (model.fused_loss)")
    logits = model(
        tokens=batch["tokens"],
        mask=batch.get("mask"),
        encoder_input=batch.get("encoder_input"),
        encoder_mask=batch.get("encoder_mask"),
        input_pos=batch.get("input_pos"),
    )
    if not isinstance(logits, list):
        labels = labels.reshape(-1)
        logits = logits.reshape(-1, logits.size(-1))
    loss = loss_fn(logits, labels)
    del logits
    return loss
