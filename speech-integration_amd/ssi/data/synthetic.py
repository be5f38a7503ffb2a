from __future__ import annotations

from functools import partial
from typing import Any

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor
from torch.nn.utils.rnn import pad_sequence
from torch.utils.data import DataLoader, Dataset, DistributedSampler

from ..constants import CROSS_ENTROPY_IGNORE_IDX, SEED

BASE_VOCAB_TXT = 128_000
N_SPECIAL_TXT = 256
PAD_SPECIAL_OFFSET = 4  # "<|finetune_right_pad_id|>" is special token #4 of the Llama-3 block (id 128004 before extension)


def padded_collate_sft(batch: list[dict[str, Any]], padding_idx: int = 0, ignore_idx: int = CROSS_ENTROPY_IGNORE_IDX,
                       additional_keys: list[str] | None = None) -> dict[str, Any]:
    """Right-pad to the longest sample; tokens with ``padding_idx``, labels with ``ignore_idx``; int64 tensors."""
    additional_keys = additional_keys or []
    input_ids = pad_sequence([torch.as_tensor(x["tokens"]) for x in batch], batch_first=True, padding_value=padding_idx)
    labels = pad_sequence([torch.as_tensor(x["labels"]) for x in batch], batch_first=True, padding_value=ignore_idx)
    if input_ids.shape[-1] > labels.shape[-1]:
        labels = F.pad(labels, (0, input_ids.shape[-1] - labels.shape[-1]), value=ignore_idx)
    elif labels.shape[-1] > input_ids.shape[-1]:
        input_ids = F.pad(input_ids, (0, labels.shape[-1] - input_ids.shape[-1]), value=padding_idx)
    return {"tokens": input_ids.long(), "labels": labels.long()} | {k: [x[k] for x in batch] for k in additional_keys}


def _one_sequence(rng: np.random.Generator, seq_len: int, n_dsus: int, fixed_len: bool, kind: str) -> tuple[np.ndarray, np.ndarray]:
    """[BOS] [system-prompt text ~24] [MODALITY_SPEECH] dsu span [MODALITY_TEXT] text span [EOT] ... repeated."""
    dsu_lo = BASE_VOCAB_TXT
    mod_speech, mod_text = BASE_VOCAB_TXT + n_dsus, BASE_VOCAB_TXT + n_dsus + 1
    special_lo = BASE_VOCAB_TXT + n_dsus + 2
    bos, eot = special_lo + 0, special_lo + 9
    target = seq_len if fixed_len else int(rng.integers(int(0.4 * seq_len), seq_len + 1))
    toks: list[int] = [bos]
    masked = 1
    if kind == "sft":
        prompt = _zipf_text(rng, 24)
        toks += prompt.tolist()
        masked += len(prompt)
    while len(toks) < target:
        n_dsu = int(rng.integers(350, 701))
        dsu = rng.integers(0, n_dsus, size=n_dsu)
        same = np.flatnonzero(dsu[1:] == dsu[:-1]) + 1  # deduplicated units: no two equal neighbours
        dsu[same] = (dsu[same] + 1 + rng.integers(0, max(1, n_dsus - 1), size=same.size)) % n_dsus
        same = np.flatnonzero(dsu[1:] == dsu[:-1]) + 1
        dsu[same] = (dsu[same] + 1) % n_dsus
        text = _zipf_text(rng, int(rng.integers(30, 75)))
        toks += [mod_speech] + (dsu_lo + dsu).tolist() + [mod_text] + text.tolist() + [eot]
    tokens = np.asarray(toks[:target], dtype=np.int64)
    labels = tokens.copy()
    labels[:masked] = CROSS_ENTROPY_IGNORE_IDX  # system prompt masked; train_on_input=true for the rest (sft.py:343-344)
    return tokens, labels


def _zipf_text(rng: np.random.Generator, n: int) -> np.ndarray:
    return np.minimum(rng.zipf(1.1, size=n) - 1, BASE_VOCAB_TXT - 1).astype(np.int64)


class SyntheticDSUDataset(Dataset):
    """Deterministic per-index samples: ``np.random.default_rng((seed, epoch, idx))`` like the reference's CPT sampler
    (``/root/reference/ssi/data/cpt.py``) so any rank can regenerate any sample."""

    def __init__(self, n_samples: int, seq_len: int, n_dsus: int = 5000, fixed_len: bool = True, kind: str = "sft",
                 seed: int = SEED):
        self.n_samples, self.seq_len, self.n_dsus, self.fixed_len, self.kind, self.seed = n_samples, seq_len, n_dsus, fixed_len, kind, seed
        self.epoch = 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.n_samples

    def __getitem__(self, idx: int) -> dict[str, Any]:
        rng = np.random.default_rng((self.seed, self.epoch, idx))
        tokens, labels = _one_sequence(rng, self.seq_len, self.n_dsus, self.fixed_len, self.kind)
        return {"tokens": tokens, "labels": labels}

    @property
    def pad_id(self) -> int:
        return BASE_VOCAB_TXT + self.n_dsus + 2 + PAD_SPECIAL_OFFSET


def synthetic_batch(batch_size: int, seq_len: int, n_dsus: int = 5000, seed: int = SEED, rank: int = 0, fixed_len: bool = True,
                    kind: str = "sft", index: int = 0) -> dict[str, Tensor]:
    ds = SyntheticDSUDataset(batch_size * (index + 1), seq_len, n_dsus, fixed_len, kind, seed + rank)
    items = [ds[index * batch_size + i] for i in range(batch_size)]
    return padded_collate_sft(items, padding_idx=ds.pad_id)


def setup_synthetic_data(n_samples: int, seq_len: int, batch_size: int, n_dsus: int, world_size: int = 1, rank: int = 0,
                         shuffle: bool = True, drop_last: bool = True, fixed_len: bool = True,
                         kind: str = "sft") -> tuple[DataLoader, DistributedSampler]:
    ds = SyntheticDSUDataset(n_samples, seq_len, n_dsus, fixed_len, kind)
    sampler = DistributedSampler(ds, num_replicas=world_size, rank=rank, shuffle=shuffle, seed=SEED)
    loader = DataLoader(ds, batch_size=batch_size, sampler=sampler, drop_last=drop_last,
                        collate_fn=partial(padded_collate_sft, padding_idx=ds.pad_id, ignore_idx=CROSS_ENTROPY_IGNORE_IDX))
    return loader, sampler


def synthetic_packed_batch(batch_size: int, seq_len: int, n_dsus: int = 5000, seed: int = SEED, rank: int = 0,
                           doc_len: int = 1100, kind: str = "sft") -> dict[str, Any]:
    """``batch_size`` packs of ``seq_len`` tokens filled with MLS-shaped synthetic documents of 0.4-1.0 x ``doc_len`` tokens
    (BASELINE config E: documents of ~600-1100 tokens in rows of 8192), in the format of ``padded_collate_packed``."""
    from .packed import PackedDataset, padded_collate_packed
    n_docs = batch_size * (seq_len // max(1, int(0.4 * doc_len)) + 2)
    ds = SyntheticDSUDataset(n_docs, doc_len, n_dsus, fixed_len=False, kind=kind)
    ds.set_epoch(seed + rank)
    docs = ({"tokens": d["tokens"], "labels": d["labels"]} for d in (ds[i] for i in range(n_docs)))
    packs = PackedDataset(docs, max_seq_len=seq_len, padding_idx=ds.pad_id, max_packs=batch_size)
    if len(packs) < batch_size:
        raise RuntimeError("not enough synthetic documents to fill the packs")
    return padded_collate_packed([packs[i] for i in range(batch_size)])
