"""Batch format of the hot path and a synthetic MLS-shaped generator.

The reference's data pipeline (HF ``datasets`` download, tiktoken, ``sardalign``) is outside the hot path and cannot run
offline; only its OUTPUT format is part of the boundary: ``{"tokens", "labels"}`` int64 ``[B, S]``,
right-padded with ``pad_id`` / ``-100`` (``/root/reference/ssi/data/__init__.py:139-199``).  ``padded_collate_sft`` keeps
that contract; ``SyntheticDSUDataset`` produces sequences with the vocabulary layout and span statistics of MLS HuBERT
DSU data for benchmarks and tests."""

from .prefetch import DevicePrefetcher
from .packed import PackedDataset, pack_dataset, packed_block_causal_mask, padded_collate_packed
from .synthetic import SyntheticDSUDataset, padded_collate_sft, setup_synthetic_data, synthetic_batch, synthetic_packed_batch

__all__ = ["SyntheticDSUDataset", "padded_collate_sft", "setup_synthetic_data", "synthetic_batch", "synthetic_packed_batch", "DevicePrefetcher", "PackedDataset", "pack_dataset",
           "packed_block_causal_mask", "padded_collate_packed"]
