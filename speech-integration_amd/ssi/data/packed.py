"""Packed batches: several samples per fixed-length row, block-causal attention, per-document RoPE.

The reference stubs this (``/root/reference/ssi/data/__init__.py:66-73,105-110,202-205``: ``pack_dataset`` ->
torchtune ``PackedDataset`` + ``padded_collate_packed``, both behind a ``NotImplementedError``; plan in
``plans/Feature - Packed Dataset Support.md``).  Semantics restated from torchtune 0.5.0:

* ``PackedDataset``: samples are appended greedily to the current pack until the next one would overflow ``max_seq_len``
  (``split_across_pack=False``: the sample opens the next pack; ``True``: it is cut at the boundary); a pack carries
  ``tokens``, ``labels``, ``input_pos`` (``arange(len)`` per sample, restarting at 0) and ``seq_lens``; the tail is padded with
  ``padding_idx`` / ``-100``, ``input_pos`` continuing the last sample's range (clamped to ``max_seq_len - 1``), and the padding
  counted as one more entry of ``seq_lens``.
* ``padded_collate_packed``: stacks packs; torchtune also expands ``seq_lens`` into a dense block-causal ``[B, S, S]`` mask.
  Here the mask is never materialised: the batch keeps ``input_pos`` (and ``seq_lens``) and the attention kernels take the
  document ranges derived from it (``HipLlamaDecoder._document_ranges``), skipping key tiles outside a query's document."""

from __future__ import annotations

from typing import Any, Iterable, Iterator

import torch
from torch import Tensor

CROSS_ENTROPY_IGNORE_IDX = -100


class PackedDataset(torch.utils.data.Dataset):
    """Greedy packing of a map-style dataset of ``{"tokens": list[int], "labels": list[int]}`` samples."""

    def __init__(self, ds: Iterable[dict[str, Any]], *, max_seq_len: int, padding_idx: int = 0, max_packs: int | None = None,
                 split_across_pack: bool = False) -> None:
        self.max_seq_len, self.padding_idx, self.split_across_pack = int(max_seq_len), int(padding_idx), bool(split_across_pack)
        self.packs: list[dict[str, Tensor]] = []
        cur = {"tokens": [], "labels": [], "input_pos": [], "seq_lens": []}
        for sample in ds:
            tokens, labels = list(sample["tokens"]), list(sample["labels"])
            if len(tokens) != len(labels):
                raise ValueError("tokens and labels of a sample must have the same length")
            if len(tokens) > self.max_seq_len and not self.split_across_pack:
                raise ValueError(f"Dataset sample is too long ({len(tokens)} > {self.max_seq_len}). Please set `split_across_pack=True` "
                                 f"or increase `max_seq_len`.")
            while tokens:
                room = self.max_seq_len - len(cur["tokens"])
                if len(tokens) > room and not self.split_across_pack and cur["tokens"]:
                    self._close(cur)
                    cur = {"tokens": [], "labels": [], "input_pos": [], "seq_lens": []}
                    room = self.max_seq_len
                take = min(len(tokens), room)
                cur["tokens"] += tokens[:take]
                cur["labels"] += labels[:take]
                cur["input_pos"] += list(range(take))
                cur["seq_lens"].append(take)
                tokens, labels = tokens[take:], labels[take:]
                if len(cur["tokens"]) == self.max_seq_len:
                    self._close(cur)
                    cur = {"tokens": [], "labels": [], "input_pos": [], "seq_lens": []}
            if max_packs is not None and len(self.packs) >= max_packs:
                break
        if cur["tokens"] and (max_packs is None or len(self.packs) < max_packs):
            self._close(cur)
        if max_packs is not None:
            del self.packs[max_packs:]

    def _close(self, cur: dict[str, list[int]]) -> None:
        n_pad = self.max_seq_len - len(cur["tokens"])
        tokens = torch.tensor(cur["tokens"] + [self.padding_idx] * n_pad, dtype=torch.long)
        labels = torch.tensor(cur["labels"] + [CROSS_ENTROPY_IGNORE_IDX] * n_pad, dtype=torch.long)
        last = cur["input_pos"][-1]
        tail = torch.arange(last + 1, last + 1 + n_pad).clamp_(0, self.max_seq_len - 1)
        input_pos = torch.cat([torch.tensor(cur["input_pos"], dtype=torch.long), tail])
        seq_lens = torch.tensor(cur["seq_lens"] + ([n_pad] if n_pad > 0 else []), dtype=torch.long)
        self.packs.append({"tokens": tokens, "labels": labels, "input_pos": input_pos, "seq_lens": seq_lens})

    def __len__(self) -> int:
        return len(self.packs)

    def __getitem__(self, i: int) -> dict[str, Tensor]:
        return self.packs[i]

    def __iter__(self) -> Iterator[dict[str, Tensor]]:
        return iter(self.packs)


def padded_collate_packed(batch: list[dict[str, Tensor]]) -> dict[str, Any]:
    """Stack packs of equal length.  No dense mask: ``input_pos`` (and ``seq_lens``, a list of 1-D tensors) carry the structure."""
    return {"tokens": torch.stack([x["tokens"] for x in batch]), "labels": torch.stack([x["labels"] for x in batch]),
            "input_pos": torch.stack([x["input_pos"] for x in batch]), "seq_lens": [x["seq_lens"] for x in batch]}


def packed_block_causal_mask(seq_lens: list[Tensor]) -> Tensor:
    """Dense ``[B, S, S]`` bool mask torchtune builds from ``seq_lens`` (block diagonal of lower-triangular blocks); used by the
    tests and by callers that hand this model's batches to a dense-mask implementation."""
    rows = []
    for lens in seq_lens:
        blocks = [torch.tril(torch.ones(int(n), int(n), dtype=torch.bool)) for n in lens.tolist()]
        rows.append(torch.block_diag(*blocks))
    return torch.stack(rows)


def pack_dataset(dataset, tokenizer, split_across_pack: bool = False) -> PackedDataset:
    """Same name and arguments as the reference's helper (``ssi/data/__init__.py:202-205``)."""
    if getattr(tokenizer, "max_seq_len", None) is None:
        raise ValueError("PackedDataset requires a max_seq_len to be set on the tokenizer.")
    return PackedDataset(dataset, max_seq_len=tokenizer.max_seq_len, padding_idx=getattr(tokenizer, "pad_id", 0),
                         split_across_pack=split_across_pack)
