"""Work plan of the attention backward for packed rows (``include/ssi_hip.h``: ``ssi_attn_plan_build``, ABI v7).

The reference collates right-padded batches (``/root/reference/ssi/data/__init__.py:139-199``) and stubs packing (``:66-73``,
``plans/Feature - Packed Dataset Support.md``); here every such batch runs as packed rows (``ssi/data/unpad.py``, ``ssi/data/packed.py``), whose
attention is block-causal over documents.  The pipelined backward kernels take their work for such rows from a PLAN: the documents cut into
items, sorted by work, dealt to workgroups of equal load.  The plan needs the document bounds on the HOST — they are there wherever batches are
collated (``input_pos`` / ``seq_lens`` are host tensors in the data layer) — and costs no device synchronisation: it is built in the prefetch
thread and copied to the device with the batch.

Only host-side integer bookkeeping lives here; the layout of a plan and the way the items are cut belong to the kernels and are therefore
computed by the library (``ssi_attn_plan_build``), not in Python."""

from __future__ import annotations

from typing import Optional

import torch
from torch import Tensor

from . import _lib

HEADER_WORDS = 16


class AttnPlan:
    """A plan on the host (int32 CPU tensor) and — once ``to_device`` has run — on the device.  ``header`` is what a launch reads on the host."""

    is_attn_plan = True

    def __init__(self, host: Tensor, dev: Optional[Tensor] = None) -> None:
        assert host.dtype == torch.int32 and not host.is_cuda and host.is_contiguous() and host.numel() >= HEADER_WORDS
        self.host, self.dev = host, dev

    @property
    def batch(self) -> int:
        return int(self.host[6])

    @property
    def seq(self) -> int:
        return int(self.host[7])

    @property
    def n_dkv_items(self) -> int:
        return int(self.host[1])

    @property
    def n_dq_groups(self) -> int:
        return int(self.host[3])

    @property
    def workspace_bytes(self) -> int:
        """What the backward needs in its workspace for this plan: fp32 partial rows of the dK/dV chunks split over the query heads."""
        return int(self.host[15]) * int(self.host[9]) * 256 * 128 * 4

    def matches(self, batch: int, seq: int, n_heads: int, n_kv: int) -> bool:
        h = self.host
        return int(h[6]) == batch and int(h[7]) == seq and int(h[8]) == n_heads and int(h[9]) == n_kv

    def to_device(self, device, non_blocking: bool = True) -> "AttnPlan":
        if self.dev is not None and self.dev.device == torch.device(device):
            return self
        src = self.host
        if non_blocking and torch.device(device).type == "cuda" and not src.is_pinned():
            src = src.pin_memory()
        return AttnPlan(self.host, src.to(device, non_blocking=non_blocking))

    def dq_groups(self) -> list[list[tuple[int, int, int, int]]]:
        """(row, query block start, document start, document end) per item, per persistent workgroup — for tests and reports."""
        h = self.host.tolist()
        off, stride = h[4], h[5]
        out = []
        for g in range(h[3]):
            base = off + g * stride
            n = h[base]
            out.append([tuple(h[base + 4 + 4 * i: base + 8 + 4 * i]) for i in range(n)])
        return out

    def dkv_items(self, with_heads: bool = False) -> list[tuple]:
        """(row, first key, document start, document end) per dK/dV workgroup; ``with_heads``: + (first query head, query heads, partial slot)."""
        h = self.host.tolist()
        return [tuple(h[h[2] + 8 * i: h[2] + 8 * i + (7 if with_heads else 4)]) for i in range(h[1])]


def documents_from_input_pos(input_pos: Tensor) -> Optional[tuple[Tensor, Tensor, Tensor]]:
    """(row, start, end) int32 tensors of the documents of a HOST ``input_pos`` [B, S], by the rule of ``ssi_doc_ranges`` (a document starts at
    position 0 of a row and wherever ``input_pos == 0``); ``None`` when the positions are not ``position - document start`` everywhere (the
    plan's kernels derive the RoPE position that way) or the tensor lives on the device (reading it would synchronise)."""
    if input_pos.is_cuda or input_pos.dim() != 2:
        return None
    B, S = input_pos.shape
    ip = input_pos.to(torch.int64)
    is_start = ip == 0
    is_start[:, 0] = True
    idx = torch.arange(S, dtype=torch.int64).expand(B, S)
    doc_start = torch.where(is_start, idx, torch.zeros_like(idx)).cummax(dim=1).values
    if not torch.equal(ip, idx - doc_start):
        return None
    rows, starts = torch.nonzero(is_start, as_tuple=True)  # row-major: documents in order
    ends = torch.cat([starts[1:], starts.new_tensor([S])])
    ends = torch.where(torch.cat([rows[1:] != rows[:-1], rows.new_tensor([True], dtype=torch.bool)]), torch.full_like(ends, S), ends)
    return rows.to(torch.int32).contiguous(), starts.to(torch.int32).contiguous(), ends.to(torch.int32).contiguous()


def build_plan(rows: Tensor, starts: Tensor, ends: Tensor, batch: int, seq: int, n_heads: int, n_kv: int, force: bool = False,
               split_all: bool = False) -> Optional[AttnPlan]:
    """``None`` when the library says the round-1..3 kernels should keep this batch (``ssi_attn_plan_build`` returns 0)."""
    lib = _lib.load()
    n_docs = int(rows.numel())
    for t in (rows, starts, ends):
        assert t.dtype == torch.int32 and not t.is_cuda and t.is_contiguous() and t.numel() == n_docs
    if n_docs == 0:
        return None
    words = int(lib.ssi_attn_plan_words(batch, seq, n_docs))
    host = torch.empty(words, dtype=torch.int32)
    used = int(lib.ssi_attn_plan_build(rows.data_ptr(), starts.data_ptr(), ends.data_ptr(), n_docs, batch, seq, n_heads, n_kv,
                                       (_lib.ATTN_PLAN_FORCE if force else 0) | (_lib.ATTN_PLAN_SPLIT_ALL if split_all else 0), host.data_ptr(), words))
    if used < 0:
        raise ValueError(f"ssi_attn_plan_build refused the documents of a [{batch}, {seq}] batch ({n_docs} documents): they must tile every row")
    if used == 0:
        return None
    return AttnPlan(host[:used].clone())


def plan_from_input_pos(input_pos: Tensor, n_heads: int, n_kv: int, force: bool = False) -> Optional[AttnPlan]:
    docs = documents_from_input_pos(input_pos)
    if docs is None:
        return None
    B, S = input_pos.shape
    return build_plan(*docs, B, S, n_heads, n_kv, force=force)


def plan_from_seq_lens(seq_lens_rows: list[list[int]], n_heads: int, n_kv: int, force: bool = False, split_all: bool = False) -> Optional[AttnPlan]:
    """Documents given as per-row lists of lengths (every row sums to the same S)."""
    rows, starts, ends = [], [], []
    S = sum(seq_lens_rows[0])
    for b, lens in enumerate(seq_lens_rows):
        assert sum(lens) == S
        o = 0
        for n in lens:
            rows.append(b), starts.append(o), ends.append(o + n)
            o += n
    t = lambda x: torch.tensor(x, dtype=torch.int32)  # noqa: E731
    return build_plan(t(rows), t(starts), t(ends), len(seq_lens_rows), S, n_heads, n_kv, force=force, split_all=split_all)
