"""Data side of the training step: what produces the ``{"tokens", "labels"}`` int64 ``[B, S]`` batches
the hot path consumes (``/root/reference/ssi/data/__init__.py``).

* ``setup_sft_data`` / ``setup_text_completion_data`` (reference ``:59-131``): dataset + ``DistributedSampler`` (seed ``SEED``) +
  ``DataLoader`` with the right-padding collate; same config keys (``conf/data/_sft_base.yaml``, ``_cpt_base.yaml``).  ``packed: true``
  — a ``NotImplementedError`` in the reference (``:66-69,105-106``) — packs the samples into rows of ``tokenizer.max_seq_len`` tokens
  for the block-causal attention kernels (``packed.py``).
* ``SFTDataset`` / ``TextCompletionDataset``: samples from speech units and text through ``ssi.tokenizer``.
* ``SyntheticDSUDataset``: MLS-shaped synthetic sequences for benchmarks and tests (no tokenizer file or dataset is on the image).
* ``DevicePrefetcher``: collation, pinning and the host-to-device copy run ahead of the step on a background thread."""

from __future__ import annotations

import logging
from functools import partial
from typing import Any

from torch.utils.data import DataLoader, DistributedSampler

from ..constants import CROSS_ENTROPY_IGNORE_IDX, SEED
from .cpt import CompletionSequenceType, TextCompletionDataset, concatenate_speech_text, get_span_idxs_binomial, interleave
from .packed import PackedDataset, pack_dataset, packed_block_causal_mask, padded_collate_packed
from .prefetch import DevicePrefetcher
from .sft import InputOutputToMessages, SFTDataset
from .sources import load_dataset_subset
from .unpad import loss_inputs, unpad_batch
from .synthetic import SyntheticDSUDataset, padded_collate_sft, setup_synthetic_data, synthetic_batch, synthetic_packed_batch

LOGGER = logging.getLogger(__name__)

__all__ = ["unpad_batch", "loss_inputs", "SyntheticDSUDataset", "padded_collate_sft", "setup_synthetic_data", "synthetic_batch", "synthetic_packed_batch", "DevicePrefetcher",
           "PackedDataset", "pack_dataset", "packed_block_causal_mask", "padded_collate_packed", "SFTDataset", "InputOutputToMessages",
           "TextCompletionDataset", "CompletionSequenceType", "interleave", "concatenate_speech_text", "get_span_idxs_binomial",
           "load_dataset_subset", "setup_sft_data", "setup_text_completion_data"]


def _plain(node: Any, drop: tuple[str, ...] = ("fixed_len",)) -> dict[str, Any]:
    """Config node -> plain dict (OmegaConf-style nodes resolve interpolations on access); ``fixed_len`` belongs to the synthetic source."""
    out = {}
    for k in node:
        if k in drop:
            continue
        v = node[k]
        out[k] = _plain(v) if hasattr(v, "keys") else (list(v) if isinstance(v, (list, tuple)) else v)
    return out


def _loader(dataset, cfg_dataset: Any, model_tokenizer: Any, loss_fn: Any, split_across_pack: bool) -> tuple[DataLoader, DistributedSampler]:
    from ..distributed import get_world_size_and_rank
    if isinstance(cfg_dataset, (list, tuple)):
        raise NotImplementedError("Support for list of datasets not implemented")
    if cfg_dataset.get("packed", False):
        dataset = pack_dataset(dataset, model_tokenizer, split_across_pack=split_across_pack)
        collate_fn = padded_collate_packed
    else:
        ignore_idx = CROSS_ENTROPY_IGNORE_IDX if loss_fn is None else loss_fn.ignore_index
        collate_fn = partial(padded_collate_sft, padding_idx=model_tokenizer.pad_id, ignore_idx=ignore_idx,
                             additional_keys=list(cfg_dataset.dataset.get("additional_keys", None) or []))
    world_size, rank = get_world_size_and_rank()
    sampler = DistributedSampler(dataset, num_replicas=world_size, rank=rank, shuffle=bool(cfg_dataset["shuffle"]), seed=SEED)
    dl = cfg_dataset.dataloader
    workers = int(dl.get("num_workers", 0) or 0)
    loader = DataLoader(dataset=dataset, batch_size=dl.batch_size, sampler=sampler, drop_last=bool(dl.get("drop_last", False)),
                        collate_fn=collate_fn, num_workers=workers, persistent_workers=bool(dl.get("persistent_workers", False)) and workers > 0)
    LOGGER.info(f"Dataset and Sampler initialized from {cfg_dataset.dataset.get('source')!r} ({len(dataset)} samples).")
    return loader, sampler


def setup_sft_data(cfg_dataset: Any, model_tokenizer: Any, loss_fn: Any = None) -> tuple[DataLoader, DistributedSampler]:
    dataset = SFTDataset(model_tokenizer=model_tokenizer, **_plain(cfg_dataset.dataset))
    return _loader(dataset, cfg_dataset, model_tokenizer, loss_fn, split_across_pack=False)


def setup_text_completion_data(cfg_dataset: Any, model_tokenizer: Any, loss_fn: Any = None) -> tuple[DataLoader, DistributedSampler]:
    dataset = TextCompletionDataset(tokenizer=model_tokenizer, **_plain(cfg_dataset.dataset))
    return _loader(dataset, cfg_dataset, model_tokenizer, loss_fn, split_across_pack=bool(cfg_dataset.get("split_across_pack", False)))
