"""Cosine-with-warmup LR schedule (reference: ``/root/reference/ssi/lr_schedule.py:12-33`` wrapping torchtune's
``get_cosine_schedule_with_warmup``).  Host-side scalar arithmetic only."""

from __future__ import annotations

import logging
import math

from torch.optim import Optimizer
from torch.optim.lr_scheduler import LambdaLR

LOGGER = logging.getLogger(__name__)


def get_cosine_schedule_with_warmup(optimizer: Optimizer, num_warmup_steps: int, num_training_steps: int,
                                    num_cycles: float = 0.5, last_epoch: int = -1) -> LambdaLR:
    def lr_lambda(current_step: int) -> float:
        if current_step < num_warmup_steps:
            return current_step / max(1, num_warmup_steps)
        progress = (current_step - num_warmup_steps) / max(1, num_training_steps - num_warmup_steps)
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * num_cycles * 2.0 * progress)))

    return LambdaLR(optimizer, lr_lambda, last_epoch)


def setup_lr_scheduler(cfg, optimizer: Optimizer, global_step: int, num_training_steps: int) -> LambdaLR | None:
    if cfg.get("lr_scheduler") is None:
        LOGGER.info("No learning rate scheduler configured. Using constant learning rate.")
        return None
    # LambdaLR steps once in __init__, so last_epoch = global_step - 1 (passed in by the Trainer) applies
    # lr_lambda(global_step) to the first batch on fresh starts and resumes alike.
    if global_step >= 0:  # resuming: LambdaLR requires initial_lr in the param groups when last_epoch != -1
        for group in optimizer.param_groups:
            group.setdefault("initial_lr", group["lr"])
    kwargs = {k: cfg.lr_scheduler[k] for k in cfg.lr_scheduler}
    return get_cosine_schedule_with_warmup(optimizer, num_training_steps=num_training_steps, last_epoch=global_step, **kwargs)


def get_lr(optimizer: Optimizer) -> float:
    """Single learning rate of the optimizer (torchtune ``training.lr_schedulers.get_lr``)."""
    lrs = {g["lr"] for g in optimizer.param_groups}
    if len(lrs) != 1:
        raise RuntimeError(f"expected one learning rate across param groups, found {sorted(lrs)}")
    return lrs.pop()
