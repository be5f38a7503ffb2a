"""Packed batches, host side, CPU only: PackedDataset / padded_collate_packed semantics (restated from
torchtune 0.5.0, which the reference stubs at ssi/data/__init__.py:66-73,202-205), the document ranges the attention kernels
take, and the oracle's packed forward pinned against HF-Llama (position_ids + 4-D block-causal mask)."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from ssi.data import PackedDataset, packed_block_causal_mask, padded_collate_packed, synthetic_packed_batch  # noqa: E402


def _samples(lengths):
    out, base = [], 1
    for n in lengths:
        toks = list(range(base, base + n))
        out.append({"tokens": toks, "labels": [t + 1000 for t in toks]})
        base += n
    return out


def test_greedy_packing_padding_positions_and_seq_lens():
    packs = PackedDataset(_samples([5, 4, 6, 3, 10]), max_seq_len=10, padding_idx=99)
    assert len(packs) == 3
    p0, p1, p2 = (packs[i] for i in range(3))
    assert p0["tokens"].tolist() == list(range(1, 10)) + [99]                        # 5 + 4, one pad: the 6-sample opens pack 1
    assert p0["labels"].tolist() == [t + 1000 for t in range(1, 10)] + [-100]
    assert p0["input_pos"].tolist() == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4]                # pad continues the last document's range
    assert p0["seq_lens"].tolist() == [5, 4, 1]                                     # padding counted as one more entry
    assert p1["seq_lens"].tolist() == [6, 3, 1] and p1["input_pos"].tolist() == [0, 1, 2, 3, 4, 5, 0, 1, 2, 3]
    assert p2["seq_lens"].tolist() == [10] and p2["input_pos"].tolist() == list(range(10))   # exact fit: no padding entry
    with pytest.raises(ValueError, match="too long"):
        PackedDataset(_samples([11]), max_seq_len=10)
    split = PackedDataset(_samples([7, 7]), max_seq_len=10, split_across_pack=True)
    assert [p["seq_lens"].tolist() for p in split] == [[7, 3], [4, 6]]
    assert split[1]["input_pos"].tolist()[:4] == [0, 1, 2, 3]                        # the cut remainder restarts at 0 (torchtune)
    batch = padded_collate_packed([p0, p1])
    assert batch["tokens"].shape == (2, 10) and batch["input_pos"].shape == (2, 10) and "mask" not in batch
    assert len(PackedDataset(_samples([5, 4, 6, 3, 10]), max_seq_len=10, max_packs=2)) == 2


def test_document_ranges_equal_the_block_causal_mask_for_real_tokens():
    from ssi.model import HipLlamaDecoder
    batch = synthetic_packed_batch(3, 512, doc_len=150, seed=7)
    pos, ds, de = HipLlamaDecoder._document_ranges(batch["input_pos"])
    B, S = batch["tokens"].shape
    ds, de = ds.view(B, S).long(), de.view(B, S).long()
    dense = packed_block_causal_mask(batch["seq_lens"])
    idx = torch.arange(S)
    mine = (idx[None, None, :] >= ds[:, :, None]) & (idx[None, None, :] <= idx[None, :, None])     # [b, q, k]
    for b in range(B):
        lens = batch["seq_lens"][b]
        padded = bool(batch["labels"][b, -1] == -100) and len(lens) > 1 and int(batch["input_pos"][b, -int(lens[-1])]) != 0
        n = int(lens[:-1].sum()) if padded else S   # real tokens: the padding tail joins the last document here, is a block there
        assert torch.equal(mine[b, :n, :], dense[b, :n, :])          # every real query sees exactly torchtune's keys
        assert not mine[b, :n, n:].any()                             # and never a padding key
    vis = (idx[None, :, None] >= idx[None, None, :]) & (idx[None, :, None] < de[:, None, :])       # key k seen by k <= q < doc_end[k]
    assert torch.equal(vis, mine)
    assert torch.equal(pos.view(B, S).long(), batch["input_pos"])


def test_oracle_packed_forward_matches_hf_llama():
    """position_ids + block-causal mask in HF-Llama == input_pos + mask in the oracle (torchtune semantics)."""
    from oracle import hf_crosscheck as hx
    params = dict(vocab_size=515, num_layers=2, num_heads=8, num_kv_heads=2, embed_dim=128, max_seq_len=64, intermediate_dim=256)
    sd = hx.seeded_state_dict(params, 41)
    packs = PackedDataset(_samples([11, 7, 20, 9, 13]), max_seq_len=32, padding_idx=0)
    batch = padded_collate_packed([packs[0], packs[1]])
    tokens = batch["tokens"] % 515
    mask = packed_block_causal_mask(batch["seq_lens"])
    oracle = hx.oracle_model(params, sd, chunks=0)
    hf = hx.build_hf(params, sd)
    with torch.no_grad():
        got = oracle(tokens, mask=mask, input_pos=batch["input_pos"]).float()
        add = torch.zeros(mask.shape, dtype=torch.float32).masked_fill(~mask, torch.finfo(torch.float32).min)[:, None]
        ref = hf(input_ids=tokens, position_ids=batch["input_pos"], attention_mask=add).logits
    real = batch["labels"] != -100
    assert float((got - ref)[real].abs().max()) <= 2e-5 * float(ref.abs().max())


def test_device_prefetcher_order_passthrough_and_errors():
    """ssi/data/prefetch.py on a CPU device: loader order kept, non-tensor values pass through, loader exceptions re-raise in the
    consumer, early exit stops the worker, attributes of the wrapped loader stay reachable."""
    from ssi.data import DevicePrefetcher

    class Loader:
        sampler = "S"

        def __init__(self, n, fail_at=None):
            self.n, self.fail_at, self.produced = n, fail_at, 0

        def __len__(self):
            return self.n

        def __iter__(self):
            for i in range(self.n):
                if i == self.fail_at:
                    raise RuntimeError("boom")
                self.produced += 1
                yield {"tokens": torch.full((2, 3), i), "ids": [i, i + 1]}

    pf = DevicePrefetcher(Loader(7), "cpu", depth=2)
    got = list(pf)
    assert [int(b["tokens"][0, 0]) for b in got] == list(range(7)) and got[3]["ids"] == [3, 4]
    assert len(pf) == 7 and pf.sampler == "S"
    with pytest.raises(RuntimeError, match="boom"):
        list(DevicePrefetcher(Loader(5, fail_at=2), "cpu"))
    ld = Loader(1000)
    for i, b in enumerate(DevicePrefetcher(ld, "cpu", depth=2)):
        if i == 3:
            break
    assert ld.produced <= 3 + 1 + 2 + 1  # consumed + in hand + queue depth + one being produced
    with pytest.raises(ValueError):
        DevicePrefetcher(Loader(1), "cpu", depth=0)


def test_attention_work_plans_cover_every_position_exactly_once_on_random_document_layouts():
    """``ssi_attn_plan_build`` (host code of the library, no GPU needed) on 60 random layouts — 1-token documents, documents on and off the
    32 / 64 / 256-row grids, one to four rows, long rows of short documents and short rows of long ones: every (key, query head) belongs to
    exactly one dK/dV workgroup and every query to exactly one dQ item; items start on their grids and lie inside their documents' reach;
    dK/dV workgroups come heaviest first, a dQ group's items heaviest first, the groups' loads within one heaviest item of each other
    (longest-processing-time); split chunks have their slots counted in the header; building twice gives the same plan."""
    import random
    import torch
    from ssi import attn_plan
    rnd = random.Random(7)
    for case in range(60):
        B, S = rnd.choice([1, 1, 2, 4]), 128 * rnd.choice([1, 2, 3, 8, 16, 45, 64])
        hi = rnd.choice([3, 40, 300, 1100, 4000])
        rows = []
        for _ in range(B):
            lens, left = [], S
            while left:
                n = min(left, rnd.randint(1, hi))
                lens.append(n)
                left -= n
            rows.append(lens)
        plan = attn_plan.plan_from_seq_lens(rows, 32, 8, force=True, split_all=case % 5 == 0)
        again = attn_plan.plan_from_seq_lens(rows, 32, 8, force=True, split_all=case % 5 == 0)
        assert plan is not None and torch.equal(plan.host, again.host)
        cover_k, cover_q = torch.zeros(B, S, dtype=torch.int32), torch.zeros(B, S, dtype=torch.int32)
        works, slots = [], set()
        for b, k0, d0, d1, h0, heads, slot in plan.dkv_items(with_heads=True):
            assert k0 % 32 == 0 and 0 <= d0 < d1 <= S and k0 < d1
            assert k0 >= (d0 & ~31) and (k0 - (d0 & ~31)) % 256 == 0 and heads in (1, 2, 4) and h0 % heads == 0 and h0 + heads <= 4
            assert (slot >= 0) == (heads < 4)
            if slot >= 0:
                assert slot not in slots
                slots.add(slot)
            cover_k[b, max(k0, d0):min(k0 + 256, d1)] += heads
            works.append((-(-d1 // 32) - k0 // 32) * heads)
        assert bool((cover_k == 4).all()), (case, rows)
        assert works == sorted(works, reverse=True)
        assert plan.workspace_bytes == len(slots) * 8 * 256 * 128 * 4 and (not slots or slots == set(range(len(slots))))
        loads, heaviest = [], 0
        for grp in plan.dq_groups():
            w = [q0 // 64 - d0 // 64 + 1 for _, q0, d0, _ in grp]
            assert grp and w == sorted(w, reverse=True)
            loads.append(sum(x + 6 for x in w))
            heaviest = max(heaviest, max(w) + 6)
            for b, q0, d0, d1 in grp:
                assert q0 % 64 == 0 and q0 >= (d0 & ~63) and q0 < d1
                cover_q[b, max(q0, d0):min(q0 + 64, d1)] += 1
        assert bool((cover_q == 1).all()), (case, rows)
        assert max(loads) - min(loads) <= heaviest, (case, loads)
