"""CPU oracle for the trainer's step algebra.  TEST INFRASTRUCTURE ONLY (same import policy as llama_oracle.py).

Restates, on CPU tensors and with plain ``torch.optim.AdamW``:

* ``Trainer._train_step``      /root/reference/ssi/trainer.py:385-395  (UNSHIFTED valid count multiplies the loss)
* ``Trainer._optimizer_step``  /root/reference/ssi/trainer.py:397-424  (grads / sum of unshifted counts, optional clip,
  AdamW, LR schedule)
* ``count_token_types``        /root/reference/ssi/train_utils.py:150-165
* cosine schedule with warmup  /root/reference/ssi/lr_schedule.py:26-31 (torchtune ``get_cosine_schedule_with_warmup``)

SURVEY.md Appendix A.5/A.6 give the algebra.  Parity status: unpinned by the reference (no numeric fixtures); pinned to
``torch.optim.AdamW`` and to HF-Llama through ``llama_oracle``.
"""

from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from .llama_oracle import IGNORE_INDEX, compute_loss


def count_token_types(tokens: Tensor, ranges: dict, pad_idx: int) -> dict:
    counts = {}
    for token_type, (start, end) in ranges.items():
        counts[token_type] = int(((tokens >= start) & (tokens <= end)).sum().item())
    counts["total"] = int((tokens != pad_idx).sum().item())
    return counts


def token_type_ranges(base_vocab_txt: int, n_dsus: int, modality_tokens: bool, n_special_txt: int) -> dict:
    ranges = {"text": (0, base_vocab_txt - 1), "dsu": (base_vocab_txt, base_vocab_txt + n_dsus - 1)}
    off = base_vocab_txt + n_dsus
    if modality_tokens:
        ranges["modality"] = (off, off + 1)
        off += 2
    ranges["special_text"] = (off, off + n_special_txt - 1)
    return ranges


def cosine_with_warmup(step: int, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5) -> float:
    if step < num_warmup_steps:
        return step / max(1, num_warmup_steps)
    progress = (step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress)))


def train_step(model, loss_fn, batch: dict) -> tuple[float, int]:
    """One micro-batch: returns (loss_batch = loss * N_unshifted, N_unshifted); grads accumulate in ``p.grad``."""
    n = int((batch["labels"] != loss_fn.ignore_index).sum().item())
    loss_batch = compute_loss(batch, model, loss_fn) * n
    loss_batch.backward()
    return float(loss_batch.item()), n


def optimizer_step(model, optimizer, num_tokens_step: int, clip_grad_norm: Optional[float] = None,
                   lr_scheduler=None) -> Optional[float]:
    """``scale_grads(1/num_tokens_step)`` -> optional clip -> AdamW -> zero_grad -> LR step.  Returns grad norm if clipped."""
    scaler = torch.tensor(1 / num_tokens_step)
    for p in model.parameters():
        if p.grad is not None:
            p.grad *= scaler
    gn = None
    if clip_grad_norm is not None:
        gn = float(torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float(clip_grad_norm)))
    optimizer.step()
    optimizer.zero_grad(set_to_none=True)
    if lr_scheduler is not None:
        lr_scheduler.step()
    return gn


def run_steps(model, loss_fn, batches: list[list[dict]], optimizer, lr_scheduler=None,
              clip_grad_norm: Optional[float] = None) -> list[float]:
    """``batches`` = list of accumulation windows, each a list of micro-batches.  Returns the logged losses
    (``loss_running / num_tokens_step``, trainer.py:415)."""
    losses = []
    for window in batches:
        loss_running, ntok = 0.0, 0
        for mb in window:
            lb, n = train_step(model, loss_fn, mb)
            loss_running += lb
            ntok += n
        optimizer_step(model, optimizer, ntok, clip_grad_norm, lr_scheduler)
        losses.append(loss_running / ntok)
    return losses


__all__ = ["IGNORE_INDEX", "count_token_types", "token_type_ranges", "cosine_with_warmup", "train_step",
           "optimizer_step", "run_steps"]
