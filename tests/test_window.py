"""An accumulation window run as one batch (``ssi/data/window.py``): the host-side joining, the grouping of a batch stream into windows, and —
on the CPU oracle, in fp32 — that the joined batch with its per-token weights carries exactly the running loss and the summed gradients of the
reference's micro-batch loop (``/root/reference/ssi/trainer.py:385-424``: ``loss x n_unshifted`` per micro-batch, gradients divided by the
window's count at the boundary; ``/root/reference/ssi/loss.py:16-22``: mean over the SHIFTED labels)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import hf_crosscheck as hx
from oracle import step_oracle
from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
from ssi.data import loss_inputs, unpad_batch
from ssi.data.window import WEIGHTS_KEY, _runs_that_fit, fuse_micro_batches, fused_windows

PAD = 7


def _ragged(B, S, lens, vocab=500, seed=0, prompt=0):
    """CPT-like rows (every real token a label) by default; ``prompt`` > 0 masks the first labels as the SFT data does."""
    g = torch.Generator().manual_seed(seed)
    tokens = torch.randint(8, vocab, (B, S), generator=g)
    labels = tokens.clone()
    for r, n in enumerate(lens):
        tokens[r, n:] = PAD
        labels[r, n:] = -100
        labels[r, :prompt] = -100
    return {"tokens": tokens, "labels": labels}


def _doc_mask(input_pos):
    pos = input_pos[0]
    doc = torch.cumsum((pos == 0).long(), 0)
    return ((doc[:, None] == doc[None, :]) & torch.ones(len(pos), len(pos), dtype=torch.bool).tril())[None]


def test_rows_of_all_micro_batches_end_to_end_and_the_counts_of_the_originals():
    mbs = [_ragged(2, 40, [40, 17], seed=1), _ragged(2, 33, [9, 33], seed=2), _ragged(2, 25, [25, 20], seed=3)]
    mbs[1]["labels"][0] = -100                               # a row with nothing to learn from
    out = fuse_micro_batches(mbs, pad_id=PAD, multiple=16)
    assert out["micro_batches"] == 3 and out["max_seq_len"] == 40
    # what the trainer counts (elementwise sums) comes from the micro-batches' own tensors, pads and all
    assert out["tokens"].shape == (1, 2 * 40 + 2 * 33 + 2 * 25)
    for f in (lambda b: int((b["tokens"] != PAD).sum()), lambda b: int((b["labels"] != -100).sum()), lambda b: int((b["tokens"] == PAD).sum())):
        assert f(out) == sum(f(b) for b in mbs)
    t, l, p = out["packed_tokens"][0], out["packed_labels"][0], out["packed_input_pos"][0]
    rows = [(0, 0, 40), (0, 1, 17), (1, 1, 33), (2, 0, 25), (2, 1, 20)]
    assert t.numel() == 144 and sum(n for _, _, n in rows) == 135
    o = 0
    for m, r, n in rows:
        assert torch.equal(t[o:o + n], mbs[m]["tokens"][r, :n]) and torch.equal(p[o:o + n], torch.arange(n))
        assert l[o] == -100 and torch.equal(l[o + 1:o + n], mbs[m]["labels"][r, 1:n])
        o += n
    assert (l[o:] == -100).all() and torch.equal(p[o:], torch.arange(9))
    # the shifted valid labels of the joined row are those of the micro-batches
    assert int((l[1:] != -100).sum()) == sum(int((b["labels"][:, 1:] != -100).sum()) for b in mbs)
    # ragged CPT micro-batches: every micro-batch its own ratio unshifted / shifted -> weights, constant over a micro-batch's rows
    u = [int((b["labels"] != -100).sum()) for b in mbs]
    s = [int((b["labels"][:, 1:] != -100).sum()) for b in mbs]
    w = out[WEIGHTS_KEY][0]
    assert w.dtype == torch.float32 and len({u[m] * s[0] == u[0] * s[m] for m in range(3)}) == 2
    expect = [(u[m] / s[m]) * (sum(s) / sum(u)) for m in range(3)]
    assert torch.allclose(w[:57], torch.full((57,), expect[0])) and torch.allclose(w[57:90], torch.full((33,), expect[1]))
    assert torch.allclose(w[90:135], torch.full((45,), expect[2])) and bool((w[135:] == 1).all())
    li = loss_inputs(out)
    assert li["loss_weights"] is out[WEIGHTS_KEY] and li["tokens"] is out["packed_tokens"]


def test_equal_ratios_need_no_weights_and_what_cannot_be_joined_is_declined():
    sft = [_ragged(2, 40, [40, 17], seed=1, prompt=3), _ragged(2, 33, [9, 33], seed=2, prompt=1)]   # column 0 masked: unshifted == shifted
    assert WEIGHTS_KEY not in fuse_micro_batches(sft, pad_id=PAD, multiple=16)
    cpt_fixed = [_ragged(4, 24, [24] * 4, seed=i) for i in range(4)]                               # equal counts: equal ratios
    out = fuse_micro_batches(cpt_fixed, pad_id=PAD, multiple=16, min_saving=-1.0)                  # (forced to pack)
    assert WEIGHTS_KEY not in out and out["packed_tokens"].shape == (1, 384)
    # full rows of one width: nothing to gain from packing — stacked as plain rows, the model's plain causal path
    out = fuse_micro_batches(cpt_fixed, pad_id=PAD, multiple=16)
    assert out["packed_tokens"].shape == (16, 24) and "packed_input_pos" not in out and WEIGHTS_KEY not in out and out["micro_batches"] == 4
    assert torch.equal(out["packed_labels"], torch.cat([b["labels"] for b in cpt_fixed])) and out["tokens"].shape == (1, 384)
    assert set(loss_inputs(out)) == {"tokens", "labels"}
    assert "packed_input_pos" in fuse_micro_batches(cpt_fixed, pad_id=PAD, multiple=16, padded_len=lambda B, S: 32)   # the model would pad each row to 32
    mixed = [cpt_fixed[0], _ragged(4, 24, [24] * 4, seed=9, prompt=2)]                             # ratios 24/23 and 1: weights, per row
    out = fuse_micro_batches(mixed, pad_id=PAD, multiple=16)
    w = out[WEIGHTS_KEY]
    assert w.shape == (8, 24) and len(set(w[:4].flatten().tolist())) == 1 and len(set(w[4:].flatten().tolist())) == 1 and float(w[0, 0]) > float(w[4, 0])
    assert fuse_micro_batches(sft[:1], pad_id=PAD) is None                                         # one micro-batch: nothing to join
    empty = _ragged(2, 20, [20, 5], seed=5)
    empty["labels"][:, 1:] = -100
    assert fuse_micro_batches([sft[0], empty], pad_id=PAD) is None                                 # its mean is 0 / 0 in the reference
    packed = {**_ragged(1, 32, [32]), "input_pos": torch.arange(32)[None]}
    assert fuse_micro_batches([sft[0], packed], pad_id=PAD) is None
    assert fuse_micro_batches([sft[0], {"tokens": sft[1]["tokens"]}], pad_id=PAD) is None
    seen = []
    out = fuse_micro_batches(sft, pad_id=PAD, multiple=16, plan_fn=lambda ip: seen.append(ip) or "plan")
    assert out["packed_attn_plan"] == "plan" and seen[0] is out["packed_input_pos"] and loss_inputs(out)["attn_plan"] == "plan"


def test_packs_of_one_length_are_stacked_row_by_row():
    def pack(seed, lens):
        b = _ragged(2, 48, [48, 48], seed=seed)
        b["input_pos"] = torch.cat([torch.arange(n) for n in lens])[None].expand(2, 48).clone()
        b["seq_lens"] = [torch.tensor(lens)] * 2
        return b
    mbs = [pack(1, [20, 28]), pack(2, [48]), pack(3, [5, 40, 3])]
    mbs[1]["labels"][:, :7] = -100                                           # another ratio unshifted / shifted
    seen = []
    out = fuse_micro_batches(mbs, pad_id=PAD, multiple=16, plan_fn=lambda ip: seen.append(ip) or "plan")
    assert out["packed_tokens"].shape == (6, 48) and torch.equal(out["packed_input_pos"], torch.cat([b["input_pos"] for b in mbs]))
    assert torch.equal(out["packed_labels"], torch.cat([b["labels"] for b in mbs])) and out["micro_batches"] == 3 and out["max_seq_len"] == 48
    assert out["packed_attn_plan"] == "plan" and seen[0] is out["packed_input_pos"] and out["tokens"].shape == (1, 288)
    w = out[WEIGHTS_KEY]
    assert w.shape == (6, 48) and float(w[2, 0]) < float(w[0, 0]) == float(w[5, 47])
    assert set(loss_inputs(out)) == {"tokens", "labels", "input_pos", "attn_plan", "loss_weights"}
    assert fuse_micro_batches([mbs[0], _ragged(2, 48, [48, 30], seed=4)], pad_id=PAD) is None            # a pack and a padded batch: not joined
    other = pack(5, [48])
    other = {k: (v[:, :32] if torch.is_tensor(v) else v) for k, v in other.items()}
    assert fuse_micro_batches([mbs[0], other], pad_id=PAD) is None                                       # packs of two lengths
    got = list(fused_windows(((i, pack(i, [48])) for i in range(4)), 2, max_tokens=10_000, single=lambda b: b, pad_id=PAD, multiple=16))
    assert [(i, b["micro_batches"], tuple(b["packed_tokens"].shape)) for i, b in got] == [(1, 2, (4, 48)), (3, 2, (4, 48))]
    got = list(fused_windows(((i, pack(i, [48])) for i in range(4)), 4, max_tokens=200, single=lambda b: b, pad_id=PAD, multiple=16))
    assert [(i, b["micro_batches"]) for i, b in got] == [(1, 2), (3, 2)]                                 # 96 positions per pack: two fit into 200


def test_a_stream_of_micro_batches_becomes_one_batch_per_window():
    mk = lambda i, lens=(30, 12): _ragged(2, 32, list(lens), seed=i)  # noqa: E731
    kw = dict(pad_id=PAD, multiple=16)
    single = lambda b: {**b, "alone": True}  # noqa: E731
    got = list(fused_windows(((i, mk(i)) for i in range(8)), 4, max_tokens=10_000, single=single, **kw))
    assert [i for i, _ in got] == [3, 7] and all(b["micro_batches"] == 4 for _, b in got)
    assert torch.equal(got[1][1]["tokens"], torch.cat([mk(i)["tokens"].reshape(1, -1) for i in range(4, 8)], dim=1))
    # a window whose tokens do not fit is cut into runs of consecutive micro-batches (42 kept tokens each: 2 fit into 100)
    got = list(fused_windows(((i, mk(i)) for i in range(4)), 4, max_tokens=100, single=single, **kw))
    assert [(i, b.get("micro_batches")) for i, b in got] == [(1, 2), (3, 2)]
    got = list(fused_windows(((i, mk(i)) for i in range(4)), 4, max_tokens=90, single=single, **kw))   # 2 + 2 again: 84 fit, 126 do not
    assert [(i, b.get("micro_batches")) for i, b in got] == [(1, 2), (3, 2)]
    got = list(fused_windows(((i, mk(i)) for i in range(4)), 4, max_tokens=50, single=single, **kw))   # nothing fits with its neighbour
    assert [i for i, _ in got] == [0, 1, 2, 3] and all(b.get("alone") for _, b in got)
    assert _runs_that_fit([5, 5, 5, 50, 5], 12) == [(0, 2), (2, 3), (3, 4), (4, 5)]
    # a stream that enters a window in its middle (a resume cannot: resume_position skips whole windows) or ends inside one: unfused
    got = list(fused_windows(((i, mk(i)) for i in range(2, 10)), 4, max_tokens=10_000, single=single, **kw))
    assert [(i, b.get("micro_batches", 1)) for i, b in got] == [(2, 1), (3, 1), (7, 4), (8, 1), (9, 1)]
    # a window with a batch that is not a plain right-padded pair runs as it came
    odd = {**mk(5), "input_pos": torch.arange(32).expand(2, 32)}
    got = list(fused_windows(((i, odd if i == 5 else mk(i)) for i in range(8)), 4, max_tokens=10_000, single=single, **kw))
    assert [(i, b.get("micro_batches", 1)) for i, b in got] == [(3, 4), (4, 1), (5, 1), (6, 1), (7, 1)]
    # the dev-set loss joins whatever comes, a short last group included (ssi/eval.py)
    got = list(fused_windows(((i, mk(i)) for i in range(7)), 3, max_tokens=10_000, partial_windows=True, **kw))
    assert [(i, b.get("micro_batches", 1)) for i, b in got] == [(2, 3), (5, 3), (6, 1)]
    got = list(fused_windows(((i, mk(i)) for i in range(1, 6)), 3, max_tokens=10_000, partial_windows=True, **kw))
    assert [(i, b.get("micro_batches", 1)) for i, b in got] == [(2, 2), (5, 3)]
    # the single-batch path is the trainer's unpad_batch
    got = list(fused_windows(((i, mk(i)) for i in range(2, 4)), 4, max_tokens=10_000,
                             single=lambda b: unpad_batch(b, pad_id=PAD, multiple=16, min_saving=0.0), **kw))
    assert all("packed_tokens" in b and "micro_batches" not in b for _, b in got)


def _pair_values(tokens, labels, input_pos=None):
    """A stand-in for the per-token loss that depends on exactly what a decoder's loss term depends on at the level the joining can get wrong:
    the (token, target) pair and the token's position in its document."""
    shifted = torch.hstack((labels[..., 1:], torch.full_like(labels[..., -1:], -100)))
    pos = input_pos if input_pos is not None else torch.arange(tokens.shape[1]).expand_as(tokens)
    v = ((tokens * 7919 + shifted * 104729 + pos * 31) % 1009).double() / 1009.0 + 0.5
    return torch.where(shifted != -100, v, torch.zeros_like(v)), shifted


@pytest.mark.parametrize("seed", range(40))
def test_random_windows_keep_the_loops_sum_of_mean_times_count(seed):
    """Random windows (2-5 micro-batches, 1-4 rows each, ragged or full, masked prompts or none, rows without labels, one width or several,
    padded batches or packs): ``sum_m mean_m x u_m`` of the loop == ``(sum w v / S) x U`` of the joined batch, in float64, whatever form the
    joining takes — with a stand-in for the per-token loss that sees the (token, target) pair and the position in the document."""
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    n_mb, one_width, full_rows, packs = r(2, 5), r(0, 1), r(0, 3) == 0, r(0, 4) == 0
    width = r(8, 40)
    mbs = []
    for m in range(n_mb):
        B, S = r(1, 4), (width if (one_width or packs) else r(8, 40))
        lens = [S if full_rows else r(2, S) for _ in range(B)]
        b = _ragged(B, S, lens, seed=seed * 10 + m, prompt=r(0, 3))
        if r(0, 5) == 0 and B > 1:
            b["labels"][0] = -100
        if packs:   # rows as packs: documents inside, input_pos restarting
            cuts = sorted({0, *(r(1, S - 1) for _ in range(r(0, 2)))})
            b["input_pos"] = torch.cat([torch.arange(z - a) for a, z in zip(cuts, cuts[1:] + [S])])[None].expand(B, S).clone()
        mbs.append(b)
    if any(int((b["labels"][:, 1:] != -100).sum()) == 0 for b in mbs):
        assert fuse_micro_batches(mbs, pad_id=PAD, multiple=8) is None
        return
    running, U = 0.0, 0
    for b in mbs:
        v, shifted = _pair_values(b["tokens"], b["labels"], b.get("input_pos"))
        u = int((b["labels"] != -100).sum())
        running, U = running + float(v.sum() / int((shifted != -100).sum())) * u, U + u
    out = fuse_micro_batches(mbs, pad_id=PAD, multiple=8)
    li = loss_inputs(out)
    v, shifted = _pair_values(li["tokens"], li["labels"], li.get("input_pos"))
    w = li["loss_weights"].double() if "loss_weights" in li else torch.ones_like(v)
    assert int((out["labels"] != -100).sum()) == U
    got = float((w * v).sum() / int((shifted != -100).sum())) * U
    assert abs(got - running) <= 2e-7 * abs(running), (got, running, sorted(out))   # (the weights are fp32)


@pytest.mark.parametrize("kind", ["cpt_ragged", "sft", "stacked"])
def test_the_joined_window_has_the_running_loss_and_the_gradients_of_the_micro_batch_loop_on_the_cpu_oracle(kind):
    params, _, _, seed = hx.CASES["tiny"]
    sd = hx.seeded_state_dict(params, seed)
    V = params["vocab_size"]
    if kind == "cpt_ragged":   # every real token a label: unshifted - shifted = rows, the ratio differs from micro-batch to micro-batch
        mbs = [_ragged(2, 45, [45, 9], vocab=V, seed=3), _ragged(2, 30, [30, 28], vocab=V, seed=4), _ragged(2, 38, [12, 38], vocab=V, seed=5)]
    elif kind == "stacked":    # full rows of one width stay plain rows; one micro-batch with masked prompts, one without: two ratios
        mbs = [_ragged(2, 40, [40, 40], vocab=V, seed=3), _ragged(3, 40, [40, 40, 39], vocab=V, seed=4, prompt=3)]
    else:
        mbs = [_ragged(2, 45, [45, 9], vocab=V, seed=3, prompt=4), _ragged(2, 30, [30, 28], vocab=V, seed=4, prompt=2)]
    loss_fn = OracleCEWithChunkedOutputLoss()
    # the reference's loop
    model = hx.oracle_model(params, sd)
    running, n_window = 0.0, 0
    for b in mbs:
        lb, n = step_oracle.train_step(model, loss_fn, b)
        running, n_window = running + lb, n_window + n
    ref = {k: p.grad.clone() for k, p in model.named_parameters()}
    # one batch
    out = fuse_micro_batches(mbs, pad_id=PAD, multiple=8)
    assert (WEIGHTS_KEY in out) == (kind != "sft") and ("packed_input_pos" in out) == (kind != "stacked")
    li = loss_inputs(out)
    model = hx.oracle_model(params, sd)
    if kind == "stacked":
        logits = model(tokens=li["tokens"])
    else:
        logits = model(tokens=li["tokens"], mask=_doc_mask(li["input_pos"]), input_pos=li["input_pos"])
    logits = torch.cat(logits, dim=1) if isinstance(logits, list) else logits
    shifted = torch.hstack((li["labels"][..., 1:], torch.full_like(li["labels"][..., -1:], -100))).reshape(-1)
    nll = F.cross_entropy(logits.reshape(-1, logits.size(-1)).float(), shifted, ignore_index=-100, reduction="none")
    w = li["loss_weights"].reshape(-1) if "loss_weights" in li else torch.ones_like(nll)
    n = int((out["labels"] != -100).sum())              # the trainer's count for the joined batch: the window's unshifted labels
    assert n == n_window
    loss = (w * nll).sum() / int((shifted != -100).sum())   # what fused_loss returns: weighted sum over the count of shifted valid labels
    (loss * n).backward()
    assert abs(float(loss.detach()) * n - running) <= 2e-6 * abs(running), (float(loss.detach()) * n, running)
    for k, p in model.named_parameters():
        err = float((p.grad - ref[k]).norm() / ref[k].norm())
        assert err <= 2e-5, (k, err)
    if kind != "sft":          # and the weights matter: without them the result is NOT the reference's
        plain = float(nll.detach().sum() / int((shifted != -100).sum())) * n
        assert abs(plain - running) > 1e-5 * abs(running)   # (small at a random init, where every token's loss is about log V)
