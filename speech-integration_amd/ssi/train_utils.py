"""Stateless helpers of the trainer (reference: ``/root/reference/ssi/train_utils.py``).

``count_token_types`` keeps the reference's signature and result (``:150-165``) but, on a GPU batch, runs ONE fused HIP
kernel and one device->host copy instead of 5-6 ``.sum().item()`` round trips; ``count_token_types_async`` exposes the
device-side result so the trainer can fold it into its single per-micro-batch sync."""

from __future__ import annotations

import logging
from typing import Any

import torch
from torch import Tensor

from .constants import (CHECKPOINT_VERSION, CHECKPOINT_VERSION_KEY, CONSUMED_SAMPLES_KEY, CUMULATIVE_METRICS_KEY,
                        GLOBAL_STEP_KEY, LR_SCHEDULER_KEY, OPTIMIZER_KEY, PRECISION_STR_TO_DTYPE, RNG_KEY, SEED, SEED_KEY,
                        SUPPORTED_DTYPES, TRAINING_HPARAMS_KEY)
from .llama_configs import ConfigLlama3_2

LOGGER = logging.getLogger(__name__)


def _missing_keys(cfg) -> set:
    try:
        from .config import DictConfig, OmegaConf
        if isinstance(cfg, DictConfig):
            return OmegaConf.missing_keys(cfg)
    except Exception:  # pragma: no cover
        pass
    try:
        from omegaconf import OmegaConf as _OC  # type: ignore
        return _OC.missing_keys(cfg)
    except Exception:
        return set()


def resolve_n_dsus(cfg) -> None:
    """``speech.n_dsus`` <- ``data.n_dsus`` unless set explicitly (``train_utils.py:37-59``)."""
    if cfg.speech.n_dsus is not None:
        return
    data_n_dsus = cfg.data.get("n_dsus") if cfg.get("data") is not None else None
    if data_n_dsus is not None:
        cfg.speech.n_dsus = data_n_dsus
        LOGGER.info(f"Auto-resolved speech.n_dsus={data_n_dsus} from data config")
    else:
        raise ValueError("speech.n_dsus must be set either via CLI (speech.n_dsus=5000) or "
                         "by using a data config that specifies n_dsus.")


def validate_train_cfg(cfg) -> None:
    if cfg.speech.n_dsus is None:
        raise ValueError("speech.n_dsus is still null at validation time. Call resolve_n_dsus(cfg) before validate_train_cfg().")
    if PRECISION_STR_TO_DTYPE.get(cfg.dtype) not in SUPPORTED_DTYPES:
        raise ValueError(f"Unsupported dtype: {cfg.dtype}. Supported dtypes: {SUPPORTED_DTYPES}")
    missing_keys = _missing_keys(cfg)
    if missing_keys:
        raise ValueError(f"Missing keys in config: {missing_keys}")
    for field in ("gradient_accumulation_steps", "max_steps", "log_interval", "eval_steps", "save_steps"):
        if cfg.get(field, 0) <= 0:
            raise ValueError(f"Config field '{field}' must be a positive integer, got: {cfg.get(field)}")
    if cfg.save_steps % cfg.eval_steps != 0:
        raise ValueError(f"save_steps ({cfg.save_steps}) must be a multiple of eval_steps ({cfg.eval_steps})")


def resume_training_state(ckpt_dict: dict[str, Any]) -> dict[str, Any]:
    """Extract and validate resume state from a schema-v1 checkpoint dict (``train_utils.py:84-107``)."""
    if CHECKPOINT_VERSION_KEY not in ckpt_dict:
        raise ValueError("Checkpoint predates the versioned schema (no 'checkpoint_version' key). "
                         "Legacy checkpoints are not supported. Start a fresh training run.")
    if ckpt_dict[CHECKPOINT_VERSION_KEY] != CHECKPOINT_VERSION:
        raise ValueError(f"Checkpoint version mismatch: checkpoint has version {ckpt_dict[CHECKPOINT_VERSION_KEY]}, "
                         f"but this code expects version {CHECKPOINT_VERSION}.")
    if ckpt_dict[SEED_KEY] != SEED:
        raise ValueError(f"Seed mismatch: config={SEED}, checkpoint={ckpt_dict[SEED_KEY]}")
    return {
        "global_step": ckpt_dict[GLOBAL_STEP_KEY],
        "optimizer_state": ckpt_dict[OPTIMIZER_KEY],
        "lr_scheduler_state": ckpt_dict[LR_SCHEDULER_KEY],
        "rng_state": ckpt_dict[RNG_KEY],
        "training_hparams": ckpt_dict[TRAINING_HPARAMS_KEY],
        "consumed_samples": ckpt_dict[CONSUMED_SAMPLES_KEY],
        "cumulative_metrics": ckpt_dict[CUMULATIVE_METRICS_KEY],
    }


def validate_resume_hparams(ckpt_hparams: dict[str, Any], current_hparams: dict[str, Any], force_resume: bool = False) -> None:
    for key in ("batch_size", "gradient_accumulation_steps", "world_size", "steps_per_epoch"):
        if key in ckpt_hparams and ckpt_hparams[key] != current_hparams[key]:
            msg = (f"Training hparam mismatch on resume for '{key}': checkpoint={ckpt_hparams[key]}, "
                   f"current={current_hparams[key]}. This breaks the step-to-data-position mapping.")
            if force_resume:
                LOGGER.warning(msg)
            else:
                raise ValueError(msg)


def get_token_type_ranges(llama_config: ConfigLlama3_2) -> dict[str, tuple[int, int]]:
    """Inclusive id ranges per token type; layout ``[text | dsu | modality(2) | special_text]``
    (``train_utils.py:129-147``, ``ssi/extend_llama3_2/__init__.py:100``)."""
    base = llama_config._base_vocab_size_txt
    ranges: dict[str, tuple[int, int]] = {"text": (0, base - 1), "dsu": (base, base + llama_config.n_dsus - 1)}
    offset = base + llama_config.n_dsus
    if llama_config.modality_tokens:
        ranges["modality"] = (offset, offset + 1)
        offset += 2
    ranges["special_text"] = (offset, offset + llama_config._n_special_txt - 1)
    offset += llama_config._n_special_txt
    if offset != llama_config.vocab_size:
        raise ValueError(f"Vocab vs token ranges mismatch: {offset} != {llama_config.vocab_size}")
    if "total" in ranges:
        raise AssertionError('"total" key reserved')
    return ranges


_RANGE_CACHE: dict = {}


def count_token_types_async(tokens: Tensor, ranges: dict[str, tuple[int, int]], pad_idx: int,
                            labels: Tensor | None = None, ignore_index: int = -100) -> Tensor:
    """Device-side counts, no host sync: int64 ``[len(ranges) + 2]`` = per-range counts, ``total`` (tokens != pad),
    number of labels != ignore_index.  GPU tensors only (HIP kernel K14)."""
    from . import ops
    key = (tuple(ranges.items()), tokens.device)
    rt = _RANGE_CACHE.get(key)
    if rt is None:
        rt = torch.tensor([v for lohi in ranges.values() for v in lohi], dtype=torch.int64, device=tokens.device)
        _RANGE_CACHE[key] = rt
    out = torch.empty(len(ranges) + 2, dtype=torch.int64, device=tokens.device)
    ops.count_tokens(tokens.contiguous(), None if labels is None else labels.contiguous(), rt, pad_idx, ignore_index, out)
    return out


def count_token_types(tokens: Tensor, ranges: dict[str, tuple[int, int]], pad_idx: int) -> dict[str, int]:
    """Number of tokens of each type (+ ``"total"`` = non-pad tokens); same result as ``train_utils.py:150-165``."""
    if tokens.is_cuda:
        host = count_token_types_async(tokens, ranges, pad_idx).tolist()  # one D2H copy
        counts = {tt: int(host[i]) for i, tt in enumerate(ranges)}
        counts["total"] = int(host[len(ranges)])
        return counts
    # host tensors (data-pipeline side, never the GPU hot path): plain integer comparisons
    counts = {tt: int(((tokens >= lo) & (tokens <= hi)).sum().item()) for tt, (lo, hi) in ranges.items()}
    counts["total"] = int((tokens != pad_idx).sum().item())
    return counts
