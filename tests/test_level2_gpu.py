"""INTEGRATION.md level 2: the reference's own step loop on ``HipLlamaDecoder`` with an optimizer the model knows nothing about.

The reference trainer (``/root/reference/ssi/trainer.py:385-412``) does, per accumulation window:
``(compute_loss(...) * n).backward()`` per micro-batch, ``training.scale_grads(model, 1 / num_tokens_step)`` (``p.grad *= s``),
optional ``torch.nn.utils.clip_grad_norm_``, ``optimizer.step()``, ``optimizer.zero_grad(set_to_none=True)`` with
``optimizer = torch.optim.AdamW(model.parameters(), ...)`` (``ssi/optimizer.py:8-17``).  Nothing there tells the model that a window
ended, so the model's write-first gradient protocol has to notice by itself that every ``p.grad`` was dropped.
Checked against ``oracle/step_oracle.run_steps`` (the CPU restatement of that loop) on the same weights and batches.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

MFMA_PARAMS = dict(vocab_size=700, num_layers=2, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=512, intermediate_dim=512)


def _windows(vocab, n_windows, ga, b, s, seed):
    from oracle import hf_crosscheck as hx
    return [[hx.seeded_batch(vocab, b, s, seed + 10 * w + m) for m in range(ga)] for w in range(n_windows)]


def _build(params, sd, dtype):
    from ssi.model import HipLlamaDecoder
    model = HipLlamaDecoder(**params, dtype=dtype, device=DEV)
    model.load_state_dict(sd)
    model.set_num_output_chunks(8)
    model.train()
    return model


@pytest.mark.parametrize("scale_route", ["eager_inplace", "ssi_scale_grads"])
@pytest.mark.parametrize("clip", [None, 0.5])
@pytest.mark.parametrize("dtype,tol_loss", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2)])
def test_reference_loop_with_torch_adamw_matches_step_oracle(dtype, tol_loss, clip, scale_route):
    from oracle import hf_crosscheck as hx
    from oracle import step_oracle
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.optimizer import scale_grads
    params, seed, lr = MFMA_PARAMS, 31, 1e-3
    sd = hx.seeded_state_dict(params, seed)
    if dtype == torch.bfloat16:  # both sides start from the same bf16-representable weights
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    windows = _windows(params["vocab_size"], 3, 2, 2, 128, seed)

    ref = hx.oracle_model(params, sd)
    ropt = torch.optim.AdamW(ref.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    want = step_oracle.run_steps(ref, OracleCEWithChunkedOutputLoss(), windows, ropt, None, clip_grad_norm=clip)

    model = _build(params, sd, dtype)
    assert model._mfma_shapes() or dtype == torch.float32
    loss_fn = CEWithChunkedOutputLoss()
    opt = torch.optim.AdamW(model.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)  # NOT HipAdamW
    got = []
    for window in windows:
        loss_running, ntok = 0.0, 0
        for mb in window:
            batch = {k: v.to(DEV) for k, v in mb.items()}
            n = (batch["labels"] != -100).sum()
            lb = compute_loss(batch, model, loss_fn) * n
            lb.backward()
            loss_running, ntok = loss_running + float(lb), ntok + int(n)
        if scale_route == "eager_inplace":   # torchtune.training.scale_grads
            for p in model.parameters():
                if p.grad is not None:
                    p.grad *= torch.tensor(1 / ntok).to(p.grad.device)
        else:                                 # this package's scale_grads: must not park the factor where nobody consumes it
            scale_grads(model, torch.tensor(1 / ntok))
            assert model.pending_grad_scale is None
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=clip)
        opt.step()
        opt.zero_grad(set_to_none=True)
        assert all(p.grad is None for p in model.parameters())
        got.append(loss_running / ntok)
    print(f"level-2 {dtype} clip={clip} {scale_route}: gpu {got} cpu {want}")
    for a, b in zip(got, want):
        assert abs(a - b) <= tol_loss * abs(b), (got, want)
    if dtype == torch.float32:
        for (k, p), (_, p2) in zip(model.named_parameters(), ref.named_parameters()):
            diff = (p.detach().cpu() - p2.detach()).abs()
            # AdamW moves a weight by ~lr * sign(g): a near-zero gradient may flip with the summation order
            assert float(diff.max()) <= 2 * 3 * lr + 1e-6 and float((diff > 2e-5).float().mean()) < 5e-3, k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_foreign_zero_grad_starts_a_new_window_bit_for_bit(dtype):
    """After ``torch``'s ``zero_grad(set_to_none=True)`` the next backward must WRITE the buffer: its gradients equal, bit for bit,
    those of a freshly built model holding the same weights (the failure was: added onto the last window's scaled gradients)."""
    from oracle import hf_crosscheck as hx
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    params = MFMA_PARAMS
    sd = hx.seeded_state_dict(params, 5)
    model = _build(params, sd, dtype)
    loss_fn = CEWithChunkedOutputLoss()
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    b1, b2 = ({k: v.to(DEV) for k, v in hx.seeded_batch(700, 2, 128, s).items()} for s in (1, 2))
    compute_loss(b1, model, loss_fn).backward()
    compute_loss(b2, model, loss_fn).backward()          # second micro-batch adds
    opt.step()
    opt.zero_grad(set_to_none=True)                       # the model is not told
    compute_loss(b2, model, loss_fn).backward()
    fresh = _build(params, {k: v.detach().clone() for k, v in model.state_dict().items()}, dtype)
    compute_loss(b2, fresh, loss_fn).backward()
    assert torch.equal(model._flat_grad, fresh._flat_grad)
    # a subset dropped: those start from zero, the others keep accumulating (autograd's own semantics)
    model.tok_embeddings.weight.grad = None
    model.layers[1].mlp.w2.weight.grad = None
    keep = model.layers[0].mlp.w2.weight.grad.clone()
    compute_loss(b2, model, loss_fn).backward()
    assert torch.equal(model.tok_embeddings.weight.grad, fresh.tok_embeddings.weight.grad)
    assert torch.equal(model.layers[1].mlp.w2.weight.grad, fresh.layers[1].mlp.w2.weight.grad)
    want = (keep.float() * 2).to(dtype) if dtype == torch.float32 else None
    if want is not None:
        torch.testing.assert_close(model.layers[0].mlp.w2.weight.grad, want, rtol=1e-6, atol=0)
    # set_to_none=False through the foreign optimizer zeroes the views = the flat buffer; the next backward adds onto zeros
    opt.zero_grad(set_to_none=False)
    assert float(model._flat_grad.abs().max()) == 0.0
    compute_loss(b2, model, loss_fn).backward()
    assert torch.equal(model._flat_grad, fresh._flat_grad)


def test_hip_adamw_step_without_backward_changes_nothing():
    """a window without a backward (skipped or failed micro-batches) must not re-apply the last window's gradients, which the
    never-zeroed buffer still holds; torch.optim.AdamW skips parameters whose grad is None."""
    from oracle import hf_crosscheck as hx
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.optimizer import HipAdamW, scale_grads
    params = MFMA_PARAMS
    model = _build(params, hx.seeded_state_dict(params, 9), torch.bfloat16)
    opt = HipAdamW(model.parameters(), model=model, lr=1e-2)
    batch = {k: v.to(DEV) for k, v in hx.seeded_batch(700, 2, 128, 3).items()}
    compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
    w = model._flat.clone()
    scale_grads(model, 0.5)
    opt.step()                                             # no backward in between
    assert torch.equal(model._flat, w) and opt._step_count == 1 and model.pending_grad_scale is None
    compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()
    opt.step()
    assert not torch.equal(model._flat, w) and opt._step_count == 2


def test_failed_backward_behind_the_head_does_not_leak_into_the_next_window():
    """on the unfused route the head's backward writes the embedding gradient before the decoder's generation check can raise;
    ``zero_grad`` must forget that, or a later hidden-states-only backward would scatter-add onto stale embedding gradients."""
    from oracle import hf_crosscheck as hx
    params = MFMA_PARAMS
    model = _build(params, hx.seeded_state_dict(params, 9), torch.float32)
    tok = hx.seeded_batch(700, 2, 128, 4)["tokens"].to(DEV)
    model.set_num_output_chunks(0)
    logits = model(tokens=tok)
    model.forward_hidden(tok)                              # a second training forward overwrites the saved activations
    with pytest.raises(RuntimeError, match="overwritten"):
        logits.sum().backward()
    assert model._emb_grad_written
    model.zero_grad(set_to_none=True)
    assert not model._emb_grad_written
    hid = model.forward_hidden(tok)
    hid.float().pow(2).sum().backward()
    got = model.tok_embeddings.weight.grad.clone()
    fresh = _build(params, {k: v.detach().clone() for k, v in model.state_dict().items()}, torch.float32)
    fresh.forward_hidden(tok).float().pow(2).sum().backward()
    assert torch.equal(got, fresh.tok_embeddings.weight.grad)


def test_backward_after_a_step_without_zero_grad_accumulates_like_torch():
    """torch.optim semantics: ``step()`` does not end the accumulation window, ``zero_grad`` does."""
    from oracle import hf_crosscheck as hx
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.optimizer import HipAdamW
    params = MFMA_PARAMS
    sd = hx.seeded_state_dict(params, 9)
    batch = {k: v.to(DEV) for k, v in hx.seeded_batch(700, 2, 128, 3).items()}
    model = _build(params, sd, torch.float32)
    opt = HipAdamW(model.parameters(), model=model, lr=0.0, weight_decay=0.0)   # a step that moves nothing: same weights before and after
    compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()
    once = model._flat_grad.clone()
    opt.step()
    compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()           # no zero_grad in between
    fresh = _build(params, sd, torch.float32)
    compute_loss(batch, fresh, CEWithChunkedOutputLoss()).backward()
    compute_loss(batch, fresh, CEWithChunkedOutputLoss()).backward()
    assert torch.equal(model._flat_grad, fresh._flat_grad) and not torch.equal(model._flat_grad, once)
    opt.zero_grad(set_to_none=True)
    compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()           # a new window: written, not added
    assert torch.equal(model._flat_grad, once)
