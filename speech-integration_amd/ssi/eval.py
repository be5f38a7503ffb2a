"""Dev-set loss (reference: ``/root/reference/ssi/eval.py:15-41``): forward-only use of the hot path.

Same result as the reference (sum over dev batches of ``loss_b * n_b`` divided by ``sum n_b``, with ``n_b`` the UNSHIFTED
count of non-ignored labels) but accumulated on the device with a single host sync at the end, and all-reduced across
data-parallel ranks when a process group is initialised (the reference omits that reduction,
``plans/Training Cleanup Tasks.md:83-87``).

Round 5: the dev loader's batches (2 rows each by default, ``conf/data/_sft_base.yaml:45``) reach the model ``join_batches`` at a time as one
batch (``ssi/data/window.py``: rows end to end without their padding; the sum of ``loss_b x n_b`` is kept by per-token weights exactly as for a
training window), collated and copied ahead by the prefetch thread — a forward over 2 x 2048 positions fills the GPU as badly as a training
micro-batch of that size does."""

from __future__ import annotations

import logging
from collections.abc import Callable

import torch
import torch.distributed as dist

from .loss import compute_loss

LOGGER = logging.getLogger(__name__)


def batch_to_device(batch: dict, device: torch.device) -> None:
    """In-place move of tensor values (torchtune ``utils.batch_to_device``); non-tensor values are left alone."""
    for k, v in batch.items():
        if isinstance(v, dict):
            batch_to_device(v, device)
        elif isinstance(v, torch.Tensor):
            batch[k] = v.to(device, non_blocking=True)
        elif getattr(v, "is_attn_plan", False) and torch.device(device).type == "cuda":  # ssi.attn_plan.AttnPlan: host copy kept
            batch[k] = v.to_device(device)


def _joined(data_dev, model, loss_fn, device: torch.device, join_batches: int, max_tokens: int, pad_id: int, prefetch: int):
    """The dev batches, ``join_batches`` at a time as one batch where they can be joined (``fused_windows``: plain right-padded batches or
    packs of one length; anything else passes through as it came), prepared ``prefetch`` batches ahead on a background thread."""
    from .data.window import fused_windows
    tiles = getattr(model, "_mfma_shapes", lambda: False)()
    stream = (b for _, b in fused_windows(enumerate(data_dev), join_batches, max_tokens=max_tokens, partial_windows=True, pad_id=pad_id,
                                          ignore_index=loss_fn.ignore_index, multiple=256 if tiles else 1, padded_len=model.padded_seq_len))
    if torch.device(device).type == "cuda" and prefetch > 0:
        from .data.prefetch import DevicePrefetcher
        return DevicePrefetcher(stream, device, depth=prefetch)
    return stream


def compute_dataset_loss(model, data_dev, loss_fn: Callable, epoch: int, global_step: int, steps_per_epoch: int,
                         device: torch.device, *, join_batches: int = 0, max_tokens: int = 32768, pad_id: int = 0, prefetch: int = 2) -> float:
    """The reference's signature; the keyword arguments are this build's (``join_batches`` <= 1: every dev batch on its own, as the reference)."""
    from .data.unpad import loss_inputs
    dev_loss_running = torch.zeros((), dtype=torch.float64, device=device)
    num_tokens_dev = torch.zeros((), dtype=torch.float64, device=device)
    joinable = join_batches > 1 and hasattr(model, "fused_loss") and hasattr(model, "padded_seq_len")
    batches = _joined(data_dev, model, loss_fn, device, join_batches, max_tokens, pad_id, prefetch) if joinable else data_dev
    model.eval()
    with torch.inference_mode():
        for i_dev, dev_batch in enumerate(batches):
            batch_to_device(dev_batch, device)
            n_b = (dev_batch["labels"] != loss_fn.ignore_index).sum()  # (a joined batch: the sum of its batches' counts)
            dev_loss_running += compute_loss(loss_inputs(dev_batch), model, loss_fn).double() * n_b
            num_tokens_dev += n_b
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        both = torch.stack([dev_loss_running, num_tokens_dev])
        dist.all_reduce(both)
        dev_loss_running, num_tokens_dev = both[0], both[1]
    model.train()
    value = float((dev_loss_running / num_tokens_dev).item())
    LOGGER.info(f"Epoch {epoch + 1:03d} | Global Step {global_step} | Dev Loss: {value:.4f}")
    return value
