"""Padding-free execution of right-padded batches (``ssi/data/unpad.py``): the host-side transform, and — on the CPU oracle, in fp32 — that
the packed copy it builds carries exactly the loss and the gradients of the padded batch (reference batch format:
``/root/reference/ssi/data/__init__.py:139-199``; the loss that makes trailing positions inert: ``/root/reference/ssi/loss.py:16``)."""
import torch

from oracle import hf_crosscheck as hx
from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
from oracle.llama_oracle import compute_loss as oracle_loss
from ssi.data import DevicePrefetcher, loss_inputs, unpad_batch

PAD = 7


def _ragged(B, S, lens, vocab=500, seed=0, prompt=5):
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(8, vocab, (B, S), generator=g)
    labels = tokens.clone()
    for r, n in enumerate(lens):
        tokens[r, n:] = PAD
        labels[r, n:] = -100
        labels[r, :prompt] = -100  # masked prompt span, as the SFT data has it
    return {"tokens": tokens, "labels": labels}


def _doc_mask(input_pos):
    """Dense block-causal mask of a packed row: a document starts wherever input_pos is 0."""
    pos = input_pos[0]
    doc = torch.cumsum((pos == 0).long(), 0)
    same = doc[:, None] == doc[None, :]
    return (same & torch.ones(len(pos), len(pos), dtype=torch.bool).tril())[None]


def test_rows_are_laid_end_to_end_with_their_own_positions():
    b = _ragged(4, 40, [40, 17, 6, 29])
    b["labels"][2] = -100                                  # a row with nothing to learn from: dropped altogether
    b["labels"][3, 20:] = -100                             # real tokens whose labels are ignored behind the last target: dropped too
    out = unpad_batch(b, pad_id=PAD, multiple=16)
    assert out["tokens"] is b["tokens"] and out["labels"] is b["labels"]          # originals kept for the counts
    t, l, p = out["packed_tokens"][0], out["packed_labels"][0], out["packed_input_pos"][0]
    keep = [40, 17, 0, 20]
    assert t.numel() == 80 and t.numel() % 16 == 0 and sum(keep) == 77
    o = 0
    for r, n in enumerate(keep):
        assert torch.equal(t[o:o + n], b["tokens"][r, :n])
        assert torch.equal(p[o:o + n], torch.arange(n))
        if n:
            assert l[o] == -100 and torch.equal(l[o + 1:o + n], b["labels"][r, 1:n])
        o += n
    assert (t[o:] == PAD).all() and (l[o:] == -100).all() and torch.equal(p[o:], torch.arange(3))   # tile tail: its own document
    # the loss sees the same targets: shifted valid labels of the padded batch == those of the packed row
    n_padded = int((b["labels"][:, 1:] != -100).sum())
    assert int((l[1:] != -100).sum()) == n_padded
    li = loss_inputs(out)
    assert set(li) == {"tokens", "labels", "input_pos"} and li["tokens"] is out["packed_tokens"]
    assert loss_inputs(b) is b


def test_batches_with_nothing_to_gain_pass_through_unchanged():
    full = _ragged(2, 32, [32, 32])
    assert unpad_batch(full, pad_id=PAD, multiple=16) is full                      # no padding
    none = _ragged(2, 32, [20, 9])
    none["labels"][:] = -100
    assert unpad_batch(none, pad_id=PAD, multiple=16) is none                      # 0 / 0 stays the reference's 0 / 0
    only_col0 = _ragged(1, 32, [1], prompt=0)
    assert unpad_batch(only_col0, pad_id=PAD, multiple=16) is only_col0            # column 0 is never a target
    packed = {**_ragged(2, 32, [20, 9]), "input_pos": torch.arange(32).expand(2, 32)}
    assert unpad_batch(packed, pad_id=PAD, multiple=16) is packed                  # already packed
    little = _ragged(2, 64, [64, 63])
    assert unpad_batch(little, pad_id=PAD, multiple=16) is little                  # one position saved of 128: below min_saving
    assert "packed_tokens" in unpad_batch(little, pad_id=PAD, multiple=1, min_saving=0.0)
    # the model pads rows to whole tiles itself: the saving is counted against what it would run
    short = _ragged(2, 20, [20, 19])
    assert unpad_batch(short, pad_id=PAD, multiple=16) is short
    assert "packed_tokens" in unpad_batch(short, pad_id=PAD, multiple=16, padded_len=lambda B, S: 128)


def test_the_packed_copy_has_the_loss_and_the_gradients_of_the_padded_batch_on_the_cpu_oracle():
    params, _, _, seed = hx.CASES["tiny"]
    sd = hx.seeded_state_dict(params, seed)
    b = _ragged(4, 45, [45, 23, 31, 12], vocab=params["vocab_size"], seed=3)
    b["labels"][2] = -100
    out = unpad_batch(b, pad_id=PAD, multiple=8)
    assert out["packed_tokens"].shape[1] < 4 * 45
    packed = {**loss_inputs(out), "mask": _doc_mask(out["packed_input_pos"])}
    n = int((b["labels"] != -100).sum())                   # the trainer's UNSHIFTED count, from the original batch both times
    grads = []
    losses = []
    for batch in (b, packed):
        model = hx.oracle_model(params, sd)
        loss = oracle_loss(batch, model, OracleCEWithChunkedOutputLoss())
        (loss * n).backward()
        losses.append(float(loss.detach()))
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    assert abs(losses[0] - losses[1]) <= 1e-6 * abs(losses[0]), losses
    for k in grads[0]:
        err = float((grads[0][k] - grads[1][k]).norm() / grads[0][k].norm())
        assert err <= 2e-5, (k, err)


def test_prefetcher_applies_the_transform_in_its_thread():
    batches = [_ragged(2, 32, [20, 9], seed=i) for i in range(3)]
    seen = list(DevicePrefetcher(batches, "cpu", depth=2, transform=lambda x: unpad_batch(x, pad_id=PAD, multiple=16)))
    assert len(seen) == 3 and all("packed_tokens" in x and x["packed_tokens"].shape == (1, 32) for x in seen)
    assert all(torch.equal(x["tokens"], y["tokens"]) for x, y in zip(seen, batches))


def test_the_work_plan_of_the_attention_backward_travels_with_the_packed_copy():
    """``plan_fn`` (the model's ``build_attn_plan``) is called with the HOST input_pos of the packed copy; its plan rides under
    ``packed_attn_plan`` and reaches ``compute_loss`` as ``attn_plan``; a batch that arrives packed gains its plan too; nothing is attached
    where the function declines.  The plan itself is built by the library on the host (no GPU needed)."""
    from ssi import attn_plan
    seen = []

    def plan_fn(input_pos):
        seen.append(input_pos.clone())
        return attn_plan.plan_from_input_pos(input_pos, 8, 2, force=True)

    b = _ragged(4, 300, [300, 170, 60, 290], prompt=3)
    out = unpad_batch(b, pad_id=PAD, multiple=256, plan_fn=plan_fn)
    plan = out["packed_attn_plan"]
    assert torch.equal(seen[0], out["packed_input_pos"]) and plan.matches(1, 1024, 8, 2) and plan.dev is None
    li = loss_inputs(out)
    assert li["attn_plan"] is plan and set(li) == {"tokens", "labels", "input_pos", "attn_plan"}
    # the documents of the plan are the rows (and the tile tail): every position belongs to exactly one item of either kind
    docs = attn_plan.documents_from_input_pos(out["packed_input_pos"])
    assert docs[1].tolist() == [0, 300, 470, 530, 820] and docs[2].tolist() == [300, 470, 530, 820, 1024]
    cover_k, cover_q = torch.zeros(1024, dtype=torch.int32), torch.zeros(1024, dtype=torch.int32)
    for _, k0, d0, d1, _h0, heads, _slot in plan.dkv_items(with_heads=True):   # (a heavy chunk may be split over the query heads: 4 heads in all)
        assert k0 % 32 == 0
        cover_k[max(k0, d0):min(k0 + 256, d1)] += heads
    cover_k //= 4
    loads = []
    for grp in plan.dq_groups():
        assert grp == sorted(grp, key=lambda it: -(it[1] // 64 - it[2] // 64)), "a group's items come heaviest first"
        loads.append(sum(it[1] // 64 - it[2] // 64 + 1 for it in grp))
        for _, q0, d0, d1 in grp:
            assert q0 % 64 == 0
            cover_q[max(q0, d0):min(q0 + 64, d1)] += 1
    assert bool((cover_k == 1).all()) and bool((cover_q == 1).all())
    works = [(d1 // 32 + (d1 % 32 > 0) - k0 // 32) * heads for _, k0, d0, d1, _h0, heads, _slot in plan.dkv_items(with_heads=True)]
    assert works == sorted(works, reverse=True), "dK/dV items come heaviest first"
    # a batch that arrives packed
    packed = {**_ragged(1, 256, [256]), "input_pos": torch.cat([torch.arange(100), torch.arange(156)])[None]}
    got = unpad_batch(packed, pad_id=PAD, multiple=256, plan_fn=plan_fn)
    assert got["attn_plan"].matches(1, 256, 8, 2) and "packed_tokens" not in got
    # positions that are not document-relative: no plan (the kernels derive the RoPE position from the document start)
    odd = {**_ragged(1, 256, [256]), "input_pos": (torch.arange(256) + 5)[None]}
    assert attn_plan.documents_from_input_pos(odd["input_pos"]) is None and "attn_plan" not in unpad_batch(odd, pad_id=PAD, plan_fn=plan_fn)
    assert "packed_attn_plan" not in unpad_batch(b, pad_id=PAD, multiple=256, plan_fn=lambda ip: None)


def test_prefetcher_hands_the_plan_through():
    from ssi import attn_plan
    batches = [_ragged(2, 300, [200, 90], seed=i) for i in range(2)]
    tf = lambda x: unpad_batch(x, pad_id=PAD, multiple=256, plan_fn=lambda ip: attn_plan.plan_from_input_pos(ip, 4, 1, force=True))  # noqa: E731
    seen = list(DevicePrefetcher(batches, "cpu", depth=2, transform=tf))
    assert all(x["packed_attn_plan"].matches(1, 512, 4, 1) for x in seen)
