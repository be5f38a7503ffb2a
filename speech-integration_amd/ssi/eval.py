"""Dev-set loss (reference: ``/root/reference/ssi/eval.py:15-41``): forward-only use of the hot path.

Same result as the reference (sum over dev batches of ``loss_b * n_b`` divided by ``sum n_b``, with ``n_b`` the UNSHIFTED
count of non-ignored labels) but accumulated on the device with a single host sync at the end, and all-reduced across
data-parallel ranks when a process group is initialised (the reference omits that reduction,
``plans/Training Cleanup Tasks.md:83-87``)."""

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


def compute_dataset_loss(model, data_dev, loss_fn: Callable, epoch: int, global_step: int, steps_per_epoch: int,
                         device: torch.device) -> float:
    dev_loss_running = torch.zeros((), dtype=torch.float64, device=device)
    num_tokens_dev = torch.zeros((), dtype=torch.float64, device=device)
    model.eval()
    with torch.inference_mode():
        for i_dev, dev_batch in enumerate(data_dev):
            batch_to_device(dev_batch, device)
            n_b = (dev_batch["labels"] != loss_fn.ignore_index).sum()
            dev_loss_running += compute_loss(dev_batch, model, loss_fn).double() * n_b
            num_tokens_dev += n_b
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        both = torch.stack([dev_loss_running, num_tokens_dev])
        dist.all_reduce(both)
        dev_loss_running, num_tokens_dev = both[0], both[1]
    model.train()
    value = float((dev_loss_running / num_tokens_dev).item())
    LOGGER.info(f"Epoch {epoch + 1:03d} | Global Step {global_step} | Dev Loss: {value:.4f}")
    return value
