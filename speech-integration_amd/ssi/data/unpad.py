"""Padding-free execution of the reference's own batch format (exact).

The reference collates right-padded batches (``/root/reference/ssi/data/__init__.py:139-199``: ``tokens`` padded with the pad id, ``labels``
with -100) and its packing is a stub (``:66-69``).  On such a batch the decoder spends the same time on a pad position as on a real one: with
lengths ~U(0.4 S, S) that is 30 % of the step.  A row's trailing positions influence nothing, though: ``compute_loss`` shifts the labels left by
one (``ssi/loss.py:16``), so position ``i`` of a row carries a loss term iff ``labels[i + 1] != -100``, and under causal attention it
influences only positions ``>= i`` of its own row.  With ``j`` = the last index of the row whose label is not ignored, the positions ``> j``
therefore reach no loss term and no gradient — dropping them changes neither.

``unpad_batch`` (host side: it runs in the prefetch thread, where the lengths are known without a device sync) keeps positions ``0 .. j`` of
every row, lays the rows end to end as ONE sequence ``[1, T']`` (``T'`` a whole number of 256-row tiles) and gives every row its own positions
``0, 1, ...`` — the packed form the model already runs (``ssi/data/packed.py``): ``input_pos`` restarting at 0 makes each row a document of the
block-causal attention and of RoPE.  The first label of every row is set to -100 in the packed copy: after the global shift it would otherwise
become the target of the PREVIOUS row's last position (the reference never uses column 0 as a target either).  The (position, target) pairs of
the loss are exactly those of the padded batch, so is the count of shifted valid labels the cross-entropy mean divides by.

The original ``tokens`` / ``labels`` stay in the batch: the trainer's token-type counts, the unshifted label count the loss is scaled by
(``ssi/trainer.py:388-393``) and ``max_seq_len_step`` are taken from them."""

from __future__ import annotations

from typing import Any, Callable, Optional

import torch

from ..constants import CROSS_ENTROPY_IGNORE_IDX

PACKED_KEYS = ("packed_tokens", "packed_labels", "packed_input_pos")
PLAN_KEY = "packed_attn_plan"  # ssi.attn_plan.AttnPlan for the packed copy (or for a batch that arrived packed: "attn_plan")


def unpad_batch(batch: dict[str, Any], *, pad_id: int = 0, ignore_index: int = CROSS_ENTROPY_IGNORE_IDX, multiple: int = 256,
                padded_len: Optional[Callable[[int, int], int]] = None, min_saving: float = 0.03,
                plan_fn: Optional[Callable[[torch.Tensor], Any]] = None) -> dict[str, Any]:
    """Add ``packed_tokens`` / ``packed_labels`` / ``packed_input_pos`` (int64 ``[1, T']``) to a right-padded ``{"tokens", "labels"}`` batch of
    host tensors, or return the batch unchanged when there is nothing to gain: already packed (``input_pos`` present), device tensors, no label
    that survives the shift (the reference's loss is 0/0 there and stays so), or fewer than ``min_saving`` of the rows the model would run
    (``padded_len(B, S)`` rows per sequence: the model pads to whole tiles itself) saved.  ``plan_fn`` (the model's ``build_attn_plan``): called
    with the HOST ``input_pos`` of the packed copy — or of a batch that arrived packed — its result (the work plan of the attention backward,
    or ``None``) travels with the batch."""
    tokens, labels = batch.get("tokens"), batch.get("labels")
    if plan_fn is not None and torch.is_tensor(batch.get("input_pos")) and not batch["input_pos"].is_cuda and batch.get("attn_plan") is None:
        plan = plan_fn(batch["input_pos"])
        if plan is not None:
            batch = dict(batch)
            batch["attn_plan"] = plan
        return batch
    if (not torch.is_tensor(tokens) or not torch.is_tensor(labels) or tokens.is_cuda or labels.is_cuda or tokens.dim() != 2
            or tokens.shape != labels.shape or batch.get("input_pos") is not None or batch.get("mask") is not None or PACKED_KEYS[0] in batch):
        return batch
    B, S = tokens.shape
    idx = torch.arange(S, dtype=torch.int64)
    last = torch.where(labels != ignore_index, idx, torch.full_like(idx, -1)).max(dim=1).values  # j per row, -1: nothing to learn from the row
    keep = torch.where(last >= 1, last + 1, torch.zeros_like(last))  # j = 0: column 0 is never a target
    total = int(keep.sum())
    if total == 0:
        return batch
    t_packed = -(-total // multiple) * multiple
    rows_now = B * (padded_len(B, S) if padded_len is not None else S)
    if t_packed > (1.0 - min_saving) * rows_now:
        return batch
    p_tokens = torch.full((1, t_packed), int(pad_id), dtype=tokens.dtype)
    p_labels = torch.full((1, t_packed), int(ignore_index), dtype=labels.dtype)
    p_pos = torch.empty((1, t_packed), dtype=torch.int64)
    o = 0
    for r, n in enumerate(keep.tolist()):
        if n == 0:
            continue
        p_tokens[0, o:o + n] = tokens[r, :n]
        p_labels[0, o:o + n] = labels[r, :n]
        p_labels[0, o] = ignore_index
        p_pos[0, o:o + n] = idx[:n]
        o += n
    p_pos[0, o:] = idx[: t_packed - o] if t_packed - o <= S else torch.arange(t_packed - o)  # the tile tail: a document of its own, all ignored
    out = dict(batch)
    out["packed_tokens"], out["packed_labels"], out["packed_input_pos"] = p_tokens, p_labels, p_pos
    if plan_fn is not None:
        plan = plan_fn(p_pos)
        if plan is not None:
            out[PLAN_KEY] = plan
    return out


def loss_inputs(batch: dict[str, Any]) -> dict[str, Any]:
    """What ``compute_loss`` should see: the packed copy when the prefetcher made one, the batch itself otherwise."""
    if PACKED_KEYS[0] not in batch:
        return batch
    out = {"tokens": batch["packed_tokens"], "labels": batch["packed_labels"]}
    if batch.get("packed_input_pos") is not None:  # (absent: a window's full rows stacked as plain rows, ssi/data/window.py)
        out["input_pos"] = batch["packed_input_pos"]
    if batch.get(PLAN_KEY) is not None:
        out["attn_plan"] = batch[PLAN_KEY]
    if batch.get("packed_loss_weights") is not None:  # a window's micro-batches joined (ssi/data/window.py)
        out["loss_weights"] = batch["packed_loss_weights"]
    return out
