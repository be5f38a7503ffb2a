"""An accumulation window run as ONE batch (exact up to floating-point summation order).

The reference accumulates gradients over ``gradient_accumulation_steps`` micro-batches (default 4, ``/root/reference/conf/training.yaml:11``; the
loop: ``/root/reference/ssi/trainer.py:385-424``) of 2 rows (SFT, ``conf/data/_sft_base.yaml:21``) or 16 (CPT): micro-batches exist to fit the
activations of a step into the memory of the GPU it was written for.  An MI355X holds the activations of the whole window (288 GB), and a
2 x 2048 micro-batch fills its 256 CUs badly: 64-tile GEMM grids, four weight-gradient passes with a quarter of the K each.  So the micro-batches
of a window are collated as usual, and then — on the host, in the prefetch thread — joined: ragged rows laid end to end as one packed sequence,
exactly as ``ssi.data.unpad`` does with the rows of one batch (every row a document of the block-causal attention with its own positions, no
padding); full rows of one width stacked as plain rows; packs of one length stacked row by row.  One forward, one backward, one optimizer step:
the window's gradient is the same sum.

What has to be kept is the reference's normalisation.  Per micro-batch m it adds ``mean_m x u_m`` to the running loss (``trainer.py:393-395``),
where ``mean_m`` is the cross-entropy over the micro-batch's ``s_m`` SHIFTED valid labels (``ssi/loss.py:16-22``) and ``u_m`` its count of
UNSHIFTED ones, and divides the summed gradients by ``U = sum u_m`` at the boundary (``trainer.py:404``): every token's term carries the weight
``u_m / s_m`` of its micro-batch.  The two counts differ by the rows whose column-0 label is valid (all rows in CPT, none in SFT, where BOS is
masked), so the weights are equal in SFT and between CPT micro-batches of full rows, and differ by a few 1e-4 between ragged CPT
micro-batches.  The joined batch therefore carries one fp32 weight per position,
``w = (u_m / s_m) (S / U)`` with ``S = sum s_m`` (``loss_weights``: ``fused_loss`` returns ``sum w nll / S``, the trainer multiplies by ``U`` as
for any micro-batch, which gives ``sum_m mean_m u_m``), and none at all when the ratios are equal.  The cross-entropy kernel applies the
weight as an additive term of its exponent (``ssi_ce_fwd_weighted``): free.

Not fused (the micro-batches then run one by one as before): windows that would exceed ``max_tokens`` are cut into the fewest runs of
consecutive micro-batches that fit (each run is one forward/backward; the last one closes the window); batches that are neither plain right-padded
host ``tokens`` / ``labels`` pairs nor packs of one length (``ssi/data/packed.py``: those are stacked row by row, documents and left shift of
every row as they were); a micro-batch without any shifted valid label (its mean is 0/0 in the reference and stays so here)."""

from __future__ import annotations

from typing import Any, Callable, Iterable, Iterator, Optional

import torch

from ..constants import CROSS_ENTROPY_IGNORE_IDX
from .unpad import PACKED_KEYS, PLAN_KEY

WEIGHTS_KEY = "packed_loss_weights"


def _plain_padded(batch: Any) -> bool:
    if not isinstance(batch, dict):
        return False
    t, l = batch.get("tokens"), batch.get("labels")
    return (torch.is_tensor(t) and torch.is_tensor(l) and not t.is_cuda and not l.is_cuda and t.dim() == 2 and t.shape == l.shape
            and all(batch.get(k) is None for k in ("input_pos", "mask", "encoder_input", "encoder_mask")) and PACKED_KEYS[0] not in batch)


def _packed_rows(batch: Any) -> bool:
    """A batch that arrives packed (``ssi/data/packed.py``: ``tokens`` / ``labels`` / ``input_pos`` of one shape, host tensors, no mask)."""
    if not isinstance(batch, dict):
        return False
    t, l, p = batch.get("tokens"), batch.get("labels"), batch.get("input_pos")
    return (all(torch.is_tensor(x) and not x.is_cuda and x.dim() == 2 for x in (t, l, p)) and t.shape == l.shape == p.shape
            and all(batch.get(k) is None for k in ("mask", "encoder_input", "encoder_mask", "attn_plan")) and PACKED_KEYS[0] not in batch)


def kept_lengths(labels: torch.Tensor, ignore_index: int) -> torch.Tensor:
    """Positions ``0 .. j`` of every row reach a loss term (``j`` = the row's last valid label, ``ssi/data/unpad.py``); 0 for a row without one
    beyond column 0."""
    S = labels.shape[1]
    idx = torch.arange(S, dtype=torch.int64)
    last = torch.where(labels != ignore_index, idx, torch.full_like(idx, -1)).max(dim=1).values
    return torch.where(last >= 1, last + 1, torch.zeros_like(last))


def fuse_micro_batches(batches: list[dict[str, Any]], *, pad_id: int = 0, ignore_index: int = CROSS_ENTROPY_IGNORE_IDX, multiple: int = 256,
                       plan_fn: Optional[Callable[[torch.Tensor], Any]] = None, padded_len: Optional[Callable[[int, int], int]] = None,
                       min_saving: float = 0.03) -> Optional[dict[str, Any]]:
    """One batch for ``len(batches) >= 2`` consecutive micro-batches of a window, or ``None`` when they cannot be fused (see the module text).
    The result has the keys of an unpadded batch (``ssi.data.unpad``): ``tokens`` / ``labels`` — here the micro-batches' own tensors flattened
    and joined, ``[1, sum B_m S_m]``, which is all the trainer's counts need (they are sums over elements) — the packed copy the model runs,
    ``packed_loss_weights`` when the micro-batches' ratios differ, the attention backward's plan, ``max_seq_len`` (the widest micro-batch)
    and ``micro_batches``.  Micro-batches of one width whose rows are (nearly) full — dropping the padding would save less than ``min_saving`` of
    the rows the model runs, ``padded_len(B, S)`` per sequence — are stacked as plain rows ``[sum B_m, S]`` instead: no ``packed_input_pos``,
    the model's plain causal path (the headline's kernels), labels untouched (a row's column 0 is never a target, its last position's target
    is ignored by the shift)."""
    if len(batches) < 2:
        return None
    packs = all(_packed_rows(b) for b in batches) and len({tuple(b["tokens"].shape[1:]) for b in batches}) == 1
    if not packs and not all(_plain_padded(b) for b in batches):
        return None
    u = [int((b["labels"] != ignore_index).sum()) for b in batches]
    s = [int((b["labels"][:, 1:] != ignore_index).sum()) for b in batches]
    if min(s) == 0:
        return None
    if packs:  # packs of one length: their rows stacked, every row keeps its documents (input_pos) and its own left shift
        U, S = sum(u), sum(s)
        out = {"tokens": torch.cat([b["tokens"].reshape(1, -1) for b in batches], dim=1),
               "labels": torch.cat([b["labels"].reshape(1, -1) for b in batches], dim=1),
               "max_seq_len": int(batches[0]["tokens"].shape[1]), "micro_batches": len(batches),
               "packed_tokens": torch.cat([b["tokens"] for b in batches], dim=0), "packed_labels": torch.cat([b["labels"] for b in batches], dim=0),
               "packed_input_pos": torch.cat([b["input_pos"] for b in batches], dim=0)}
        if not all(u[m] * s[0] == u[0] * s[m] for m in range(len(batches))):
            out[WEIGHTS_KEY] = torch.cat([torch.full(b["tokens"].shape, (u[m] / s[m]) * (S / U), dtype=torch.float32)
                                          for m, b in enumerate(batches)], dim=0)
        if plan_fn is not None:
            plan = plan_fn(out["packed_input_pos"])
            if plan is not None:
                out[PLAN_KEY] = plan
        return out
    keeps = [kept_lengths(b["labels"], ignore_index).tolist() for b in batches]
    total = sum(sum(k) for k in keeps)
    t_packed = -(-total // multiple) * multiple
    U, S = sum(u), sum(s)
    uniform = all(u[m] * s[0] == u[0] * s[m] for m in range(len(batches)))
    tok_dtype, lab_dtype = batches[0]["tokens"].dtype, batches[0]["labels"].dtype
    counted = {"tokens": torch.cat([b["tokens"].reshape(1, -1) for b in batches], dim=1),
               "labels": torch.cat([b["labels"].reshape(1, -1) for b in batches], dim=1),
               "max_seq_len": max(int(b["tokens"].shape[1]) for b in batches), "micro_batches": len(batches)}
    widths = {int(b["tokens"].shape[1]) for b in batches}
    if len(widths) == 1:
        (width,) = widths
        n_rows = sum(int(b["tokens"].shape[0]) for b in batches)
        rows_now = n_rows * (padded_len(n_rows, width) if padded_len is not None else width)
        if t_packed > (1.0 - min_saving) * rows_now:  # nothing to gain from dropping the padding: plain rows
            out = dict(counted)
            out["packed_tokens"] = torch.cat([b["tokens"] for b in batches], dim=0)
            out["packed_labels"] = torch.cat([b["labels"] for b in batches], dim=0)
            if not uniform:
                out[WEIGHTS_KEY] = torch.cat([torch.full(b["tokens"].shape, (u[m] / s[m]) * (S / U), dtype=torch.float32)
                                              for m, b in enumerate(batches)], dim=0)
            return out
    p_tokens = torch.full((1, t_packed), int(pad_id), dtype=tok_dtype)
    p_labels = torch.full((1, t_packed), int(ignore_index), dtype=lab_dtype)
    p_pos = torch.empty((1, t_packed), dtype=torch.int64)
    p_w = None if uniform else torch.ones((1, t_packed), dtype=torch.float32)
    o = 0
    for m, (b, keep) in enumerate(zip(batches, keeps)):
        tokens, labels = b["tokens"], b["labels"]
        w_m = (u[m] / s[m]) * (S / U)
        for r, n in enumerate(keep):
            if n == 0:
                continue
            p_tokens[0, o:o + n] = tokens[r, :n]
            p_labels[0, o:o + n] = labels[r, :n]
            p_labels[0, o] = ignore_index  # after the global shift it would be the target of the previous row's last position
            p_pos[0, o:o + n] = torch.arange(n, dtype=torch.int64)
            if p_w is not None:
                p_w[0, o:o + n] = w_m
            o += n
    p_pos[0, o:] = torch.arange(t_packed - o, dtype=torch.int64)  # the tile tail: a document of its own, every label ignored
    out: dict[str, Any] = {**counted, "packed_tokens": p_tokens, "packed_labels": p_labels, "packed_input_pos": p_pos}
    if p_w is not None:
        out[WEIGHTS_KEY] = p_w
    if plan_fn is not None:
        plan = plan_fn(p_pos)
        if plan is not None:
            out[PLAN_KEY] = plan
    return out


def _runs_that_fit(sizes: list[int], max_tokens: int) -> list[tuple[int, int]]:
    """Consecutive micro-batches [a, b) of a window, greedily as long as their kept tokens fit ``max_tokens``."""
    runs, a, acc = [], 0, 0
    for i, n in enumerate(sizes):
        if i > a and acc + n > max_tokens:
            runs.append((a, i))
            a, acc = i, 0
        acc += n
    runs.append((a, len(sizes)))
    return runs


def fused_windows(indexed_batches: Iterable[tuple[int, dict[str, Any]]], window: int, *, max_tokens: int,
                  single: Optional[Callable[[dict[str, Any]], dict[str, Any]]] = None, partial_windows: bool = False,
                  **fuse_kwargs: Any) -> Iterator[tuple[int, dict[str, Any]]]:
    """``(index of the LAST micro-batch it holds, batch)`` pairs from ``(index, micro-batch)`` pairs: the micro-batches of every accumulation
    window (indices ``k window .. (k + 1) window - 1``) joined into as few batches as ``max_tokens`` allows.  A micro-batch that stays alone goes
    through ``single`` (the trainer: ``unpad_batch``).  A window the stream enters in its middle (it cannot, after ``resume_position``) or leaves
    early is passed through unfused — unless ``partial_windows`` (the dev-set loss, ``ssi/eval.py``: there a "window" is just a group of
    batches whose ``loss_b x n_b`` terms are summed, and the last group may be short)."""
    ignore_index = fuse_kwargs.get("ignore_index", CROSS_ENTROPY_IGNORE_IDX)
    single = single or (lambda b: b)
    held: list[tuple[int, dict[str, Any]]] = []

    def flush() -> Iterator[tuple[int, dict[str, Any]]]:
        group, held[:] = list(held), []
        packs = all(_packed_rows(b) for _, b in group)
        whole = (partial_windows or (len(group) == window and group[0][0] % window == 0)) and (packs or all(_plain_padded(b) for _, b in group))
        if not whole:
            for i, b in group:
                yield i, single(b)
            return
        sizes = [int(b["tokens"].numel()) if packs else int(kept_lengths(b["labels"], ignore_index).sum()) for _, b in group]
        for a, z in _runs_that_fit(sizes, max_tokens):
            fused = fuse_micro_batches([b for _, b in group[a:z]], **fuse_kwargs) if z - a > 1 else None
            if fused is not None:
                yield group[z - 1][0], fused
            else:
                for i, b in group[a:z]:
                    yield i, single(b)

    for i, batch in indexed_batches:
        if held and (i // window != held[0][0] // window or i != held[-1][0] + 1):
            yield from flush()
        held.append((i, batch))
        if (i + 1) % window == 0:
            yield from flush()
    if held:
        yield from flush()
