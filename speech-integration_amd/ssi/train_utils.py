"""Stateless helpers of the trainer (reference: ``/root/reference/ssi/train_utils.py``).

``count_token_types`` keeps the reference's signature and result (``:150-165``) but, on a GPU batch, runs ONE fused HIP
kernel and one device->host copy instead of 5-6 ``.sum().item()`` round trips; ``count_token_types_async`` exposes the
device-side result so the trainer can fold it into its single per-micro-batch sync."""

from __future__ import annotations

import logging
import os
from typing import Any

import torch
from torch import Tensor

from .constants import (CHECKPOINT_VERSION, CHECKPOINT_VERSION_KEY, CONSUMED_SAMPLES_KEY, CUMULATIVE_METRICS_KEY,
                        GLOBAL_STEP_KEY, LR_SCHEDULER_KEY, OPTIMIZER_KEY, PRECISION_STR_TO_DTYPE, RNG_KEY, SEED, SEED_KEY,
                        SUPPORTED_DTYPES, TRAINING_HPARAMS_KEY)
from .llama_configs import ConfigLlama3_2

LOGGER = logging.getLogger(__name__)


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask and the cgroup's quota, not the host's count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def limit_host_threads(world_size: int = 1) -> int:
    """Keep torch's intra-op thread pool within the process's CPU share.  torch sizes the pool by the HOST's cores (128 on a 256-thread box);
    under a cgroup quota (16 CPUs on the GPU boxes here) the pool's threads, spinning behind every small CPU op of the data path (collate,
    unpadding, the window's counts), burn the quota of a scheduling period in a few ms and the kernel freezes the WHOLE process — launch
    thread included — for the rest of it: the GPU sat idle a third of the time in the trainer's loop at 2 x 2048 (stalls of 40-95 ms at random
    places of the kernel trace, ``profiles/LAB_NOTES.md`` round 5).  The data path's ops are tiny: a few threads lose nothing.  An explicit
    ``OMP_NUM_THREADS`` is respected.  Returns the thread count in force."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    share = max(1, usable_cpus() // max(1, int(world_size)))
    want = max(1, min(torch.get_num_threads(), share // 4, 8))
    if want < torch.get_num_threads():
        torch.set_num_threads(want)
    return torch.get_num_threads()


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
    """Settle ``speech.n_dsus`` before anything reads it: a value given on the command line stands, otherwise the data config's
    ``n_dsus`` (every ``conf/data/*`` child names its tokenizer's codebook size) is copied in.  Behaviour of the reference's
    helper of the same name (``train_utils.py:37-59``)."""
    if cfg.speech.n_dsus is not None:
        return
    data_node = cfg.get("data")
    from_data = data_node.get("n_dsus") if data_node is not None else None
    if from_data is None:
        raise ValueError("speech.n_dsus is unset and the data config names no n_dsus: pass speech.n_dsus=<codebook size> or pick a "
                         "data config that sets n_dsus")
    cfg.speech.n_dsus = from_data
    LOGGER.info(f"speech.n_dsus={from_data} (taken from the data config)")


_POSITIVE_INT_FIELDS = ("gradient_accumulation_steps", "max_steps", "log_interval", "eval_steps", "save_steps")


def validate_train_cfg(cfg) -> None:
    """Refuse a run config the trainer cannot execute; every problem is a ``ValueError`` naming the field."""
    if cfg.speech.n_dsus is None:
        raise ValueError("speech.n_dsus is null: call resolve_n_dsus(cfg) before validate_train_cfg(cfg)")
    if PRECISION_STR_TO_DTYPE.get(cfg.dtype) not in SUPPORTED_DTYPES:
        raise ValueError(f"dtype {cfg.dtype!r} is not supported (supported: {SUPPORTED_DTYPES})")
    unset = _missing_keys(cfg)
    if unset:
        raise ValueError(f"mandatory config values are missing (???): {sorted(unset)}")
    for name in _POSITIVE_INT_FIELDS:
        value = cfg.get(name, 0)
        if value <= 0:
            raise ValueError(f"config field '{name}' must be a positive integer, got {value!r}")
    if cfg.save_steps % cfg.eval_steps:
        raise ValueError(f"save_steps ({cfg.save_steps}) must be a multiple of eval_steps ({cfg.eval_steps}): a checkpoint is "
                         "written only at steps that also evaluate")


# training_state.pt (schema v1): key in the file -> name under which the trainer consumes it
_RESUME_FIELDS = {GLOBAL_STEP_KEY: "global_step", OPTIMIZER_KEY: "optimizer_state", LR_SCHEDULER_KEY: "lr_scheduler_state",
                  RNG_KEY: "rng_state", TRAINING_HPARAMS_KEY: "training_hparams", CONSUMED_SAMPLES_KEY: "consumed_samples",
                  CUMULATIVE_METRICS_KEY: "cumulative_metrics"}


def resume_training_state(ckpt_dict: dict[str, Any]) -> dict[str, Any]:
    """Check a loaded ``training_state.pt`` (schema version, seed) and hand its parts to the trainer under the names it uses."""
    version = ckpt_dict.get(CHECKPOINT_VERSION_KEY)
    if version is None:
        raise ValueError(f"training state has no '{CHECKPOINT_VERSION_KEY}': it predates the versioned schema and cannot be resumed")
    if version != CHECKPOINT_VERSION:
        raise ValueError(f"training state is schema version {version}, this code reads version {CHECKPOINT_VERSION}")
    if ckpt_dict[SEED_KEY] != SEED:
        raise ValueError(f"training state was written with seed {ckpt_dict[SEED_KEY]}, this run uses seed {SEED}")
    return {name: ckpt_dict[key] for key, name in _RESUME_FIELDS.items()}


_DATA_POSITION_HPARAMS = ("batch_size", "gradient_accumulation_steps", "world_size", "steps_per_epoch")


def validate_resume_hparams(ckpt_hparams: dict[str, Any], current_hparams: dict[str, Any], force_resume: bool = False) -> None:
    """``global_step`` maps to a position in the data only while these four values stay what they were when the checkpoint was
    written; any change is an error unless ``force_resume`` (then a warning)."""
    changed = [f"'{k}': checkpoint={ckpt_hparams[k]}, current={current_hparams[k]}" for k in _DATA_POSITION_HPARAMS
               if k in ckpt_hparams and ckpt_hparams[k] != current_hparams[k]]
    if not changed:
        return
    msg = "training hparams changed since the checkpoint (" + "; ".join(changed) + "): the step-to-data-position mapping no longer holds"
    if not force_resume:
        raise ValueError(msg)
    LOGGER.warning(msg)


def get_token_type_ranges(llama_config: ConfigLlama3_2) -> dict[str, tuple[int, int]]:
    """Inclusive ``(first, last)`` id range of every non-empty block of the vocabulary layout, in id order — what
    ``count_token_types`` counts (reference ``train_utils.py:129-147``; layout ``ssi/extend_llama3_2/__init__.py:100``)."""
    ranges = {b.name: (b.first, b.last) for b in llama_config.vocab_layout.blocks() if b.count > 0 or b.name != "modality"}
    assert "total" not in ranges  # reserved by count_token_types for the non-pad count
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
