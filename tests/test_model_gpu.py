"""GPU parity of the whole hot path through the drop-in API (ssi.model / ssi.loss / ssi.optimizer / ssi.trainer) against
(a) the committed golden vectors and (b) the CPU oracle run live on the same inputs.

Tolerances (BASELINE.json north_star: loss/logits within 1e-3 relative, fp32):
  fp32 model : loss rel <= 1e-5, logits max-abs <= 1e-4 * max|logit|, gradients rel <= 2e-3 (sum-order differences)
  bf16 model : loss rel <= 1e-2 vs the fp32 oracle (reported), gradients checked by norm to 5e-2
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _build(name, dtype):
    from oracle import hf_crosscheck as hx
    from ssi.model import HipLlamaDecoder
    params, b, s, seed = hx.CASES[name]
    sd = hx.seeded_state_dict(params, seed)
    batch = hx.seeded_batch(params["vocab_size"], b, s, seed)
    model = HipLlamaDecoder(**params, dtype=dtype, device=DEV)
    model.load_state_dict(sd)
    model.set_num_output_chunks(8)
    model.train()
    return model, batch, params, sd


def _to_dev(batch):
    return {k: v.to(DEV) for k, v in batch.items()}


@pytest.mark.parametrize("name", ["tiny", "small"])
def test_fp32_model_matches_golden_forward_backward_step(name, golden_dir):
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.optimizer import HipAdamW, scale_grads
    g = np.load(os.path.join(golden_dir, f"llama_{name}.npz"))
    model, batch, params, sd = _build(name, torch.float32)
    assert set(model.state_dict().keys()) == set(sd.keys())  # torchtune-format keys, no output.weight (tied)
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), sd[k]), k
    dbatch = _to_dev(batch)
    # logits through the public forward: list of 8 chunks with torch.chunk sizes
    with torch.no_grad():
        chunks = model(tokens=dbatch["tokens"])
    s = batch["tokens"].shape[1]
    assert isinstance(chunks, list) and [c.shape[1] for c in chunks] == [c.shape[1] for c in torch.zeros(1, s, 1).chunk(8, dim=1)]
    logits = torch.cat(chunks, dim=1).cpu()
    rows = g["logit_rows"].tolist()
    err = np.abs(logits[:, rows, :].numpy() - g["logits_at_rows"]).max()
    assert err <= 1e-4 * float(g["logits_absmax"]), f"logits max-abs error {err}"
    # fused loss
    loss_fn = CEWithChunkedOutputLoss()
    before = {k: v.clone() for k, v in dbatch.items()}
    loss = compute_loss(dbatch, model, loss_fn)
    assert all(torch.equal(before[k], dbatch[k]) for k in before), "compute_loss must not mutate the batch"
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    # generic route (materialised logits + stand-alone CE) gives the same loss
    loss2 = loss_fn(model(tokens=dbatch["tokens"]), torch.hstack((dbatch["labels"][..., 1:], torch.full_like(dbatch["labels"][..., -1:], -100))))
    assert abs(loss2.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    # backward with the trainer's algebra: (loss * N_unshifted).backward(); grads / N_unshifted
    model.zero_grad()
    n = int(g["n_unshifted"])
    (compute_loss(dbatch, model, loss_fn) * n).backward()
    named = dict(model.named_parameters())
    assert all(p.grad is not None for p in named.values())
    for key in [k[len("gradnorm/"):] for k in g.files if k.startswith("gradnorm/")]:
        grad = named[key].grad.cpu() / n
        assert abs(float(grad.norm()) - float(g["gradnorm/" + key])) <= 2e-3 * float(g["gradnorm/" + key]) + 1e-9, key
        got = grad.reshape(-1, grad.shape[-1])[:4, :16].numpy() if grad.dim() > 1 else grad[:16].numpy()
        np.testing.assert_allclose(got, g["grad/" + key], rtol=5e-3, atol=2e-7, err_msg=key)
    # one optimizer step, reference defaults
    opt = HipAdamW(model.parameters(), model=model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, amsgrad=False, fused=True)
    scale_grads(model, torch.tensor(1 / n))
    opt.step()
    opt.zero_grad(set_to_none=True)
    # nothing zeroes the gradient buffer: the next backward overwrites it (and must reproduce the gradients of a fresh model)
    assert all(p.grad is None for p in model.parameters()) and model._grads_stale and not model._grads_dirty
    for key, p in model.named_parameters():
        if "after/" + key in g.files:
            got = p.detach().cpu().reshape(-1, p.shape[-1])[:4, :16].numpy() if p.dim() > 1 else p.detach().cpu()[:16].numpy()
            np.testing.assert_allclose(got, g["after/" + key], rtol=0, atol=5e-6, err_msg=key)
    stale = model._flat_grad.clone()
    assert float(stale.abs().max()) > 0
    (compute_loss(dbatch, model, loss_fn) * n).backward()          # first backward of the new window: writes, does not add
    g_new = model._flat_grad.clone()
    model.zero_grad(set_to_none=False)                              # the explicit form does clear the buffer
    assert float(model._flat_grad.abs().max()) == 0.0
    (compute_loss(dbatch, model, loss_fn) * n).backward()
    assert torch.equal(model._flat_grad, g_new), "a backward over left-over gradients must equal one over zeros"
    model.zero_grad()
    # optimizer state dict is torch-AdamW shaped and round-trips
    sd_opt = opt.state_dict()
    assert set(sd_opt["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd_opt["state"][0]["step"]) == 1.0
    opt.load_state_dict(sd_opt)
    assert opt._step_count == 1


def test_fp32_model_matches_live_oracle_on_ragged_batch_with_ignored_row():
    """Edge cases: an all-ignored sequence, leading/trailing ignore spans, S not a multiple of anything."""
    from oracle import hf_crosscheck as hx
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    model, _, params, sd = _build("tiny", torch.float32)
    g = torch.Generator().manual_seed(5)
    tok = torch.randint(0, params["vocab_size"], (3, 29), generator=g)
    lab = tok.clone()
    lab[0] = -100
    lab[1, :7] = -100
    lab[2, -9:] = -100
    batch = {"tokens": tok, "labels": lab}
    ref_model = hx.oracle_model(params, sd)
    ref = oracle_loss(batch, ref_model, OracleCEWithChunkedOutputLoss())
    ref.backward()
    loss = compute_loss(_to_dev(batch), model, CEWithChunkedOutputLoss())
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    for (k, p), (k2, p2) in zip(model.named_parameters(), ref_model.named_parameters()):
        assert k == k2
        torch.testing.assert_close(p.grad.cpu(), p2.grad, rtol=5e-3, atol=1e-6, msg=lambda m: f"{k}: {m}")
    # all labels ignored -> NaN like the reference (0/0)
    allign = {"tokens": tok.to(DEV), "labels": torch.full_like(tok, -100).to(DEV)}
    with torch.no_grad():
        assert torch.isnan(compute_loss(allign, model, CEWithChunkedOutputLoss()))


def test_bf16_model_close_to_fp32_oracle(golden_dir):
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    g = np.load(os.path.join(golden_dir, "llama_small.npz"))
    model, batch, params, sd = _build("small", torch.bfloat16)
    dbatch = _to_dev(batch)
    n = int(g["n_unshifted"])
    loss = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    rel = abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))
    print(f"bf16 loss {loss.item():.6f} vs fp32 oracle {float(g['loss']):.6f} rel {rel:.2e}")
    assert rel <= 1e-2
    (loss * n).backward()
    named = dict(model.named_parameters())
    for key in [k[len("gradnorm/"):] for k in g.files if k.startswith("gradnorm/")]:
        gn = float((named[key].grad.float().cpu() / n).norm())
        assert abs(gn - float(g["gradnorm/" + key])) <= 5e-2 * float(g["gradnorm/" + key]) + 1e-6, key
    # eval path under inference_mode gives the same loss and saves nothing
    model.eval()
    with torch.inference_mode():
        l2 = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    assert abs(l2.item() - loss.item()) <= 1e-6 * abs(loss.item())


def test_backward_after_overwritten_forward_fails_loudly():
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    model, batch, _, _ = _build("tiny", torch.float32)
    dbatch = _to_dev(batch)
    l1 = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    _ = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    with pytest.raises(RuntimeError):
        l1.backward()


def test_returned_logits_are_not_overwritten_and_eval_between_forward_and_backward_is_harmless():
    """torchtune returns fresh logits: tensors kept from one batch survive the next forward.  An eval / no_grad forward (dev loss,
    logit inspection) between a training forward and its backward must leave that backward's gradients untouched."""
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    model, batch, params, _ = _build("small", torch.float32)
    dbatch = _to_dev(batch)
    other = {"tokens": (dbatch["tokens"] + 3) % params["vocab_size"], "labels": dbatch["labels"]}
    model.eval()
    with torch.no_grad():
        first = model(tokens=dbatch["tokens"])
        keep = [c.clone() for c in first]
        second = model(tokens=other["tokens"])
        model.set_num_output_chunks(0)
        full = model(tokens=dbatch["tokens"])
        model.set_num_output_chunks(8)
    assert all(torch.equal(a, b) for a, b in zip(first, keep)), "logits of the first batch were overwritten by the second forward"
    assert not torch.equal(first[0], second[0])
    assert torch.equal(torch.cat(first, dim=1).float(), full)
    loss_fn = CEWithChunkedOutputLoss()
    model.train()
    model.zero_grad()
    compute_loss(dbatch, model, loss_fn).backward()
    want = model._flat_grad.clone()
    model.zero_grad()
    loss = compute_loss(dbatch, model, loss_fn)
    model.eval()
    with torch.inference_mode():                       # a dev-loss pass and a logits call squeezed in before backward
        compute_loss(other, model, loss_fn)
        model(tokens=other["tokens"])
    model.train()
    loss.backward()
    assert torch.equal(model._flat_grad, want), "an eval forward between forward and backward changed the gradients"


@pytest.mark.parametrize("ragged", [False, True], ids=["fixed-length", "ragged"])
@pytest.mark.parametrize("dtype_name,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_trainer_three_steps_match_cpu_step_oracle(tmp_path, dtype_name, tol, ragged):
    """End to end through Trainer.setup()/train() with synthetic MLS-shaped data, grad-accum 2, vs the CPU step oracle.  ``ragged``: rows of
    unequal length, right-padded by the collate function as the reference does — the prefetch thread drops the padding (ssi/data/unpad.py)
    and every micro-batch runs as one packed sequence, while the oracle steps through the PADDED batches."""
    import copy
    from oracle import step_oracle
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from ssi.config import compose
    from ssi.train_utils import resolve_n_dsus
    from ssi.trainer import Trainer
    from conftest import PKG
    cfg = compose(os.path.join(PKG, "conf"), "sft", [
        "data=sft/mls-speechtokenizer-rvq_0", f"dtype={dtype_name}", "max_steps=3", "gradient_accumulation_steps=2",
        "tokenizer.max_seq_len=96", "data.train.dataset.n_samples=16", "data.dev.dataset.n_samples=4", "eval_steps=3", "save_steps=3",
        "lr_scheduler.num_warmup_steps=2", "optimizer.lr=1e-3", f"output_dir={tmp_path}", f"checkpointer.output_dir={tmp_path}/ckpt",
        f"checkpointer.checkpoint_dir={tmp_path}/none", "checkpointer.allow_random_init=true", "data.train.shuffle=false",
        f"data.train.dataset.fixed_len={'false' if ragged else 'true'}",
    ])
    cfg.model_overrides = {"num_layers": 2, "num_heads": 4, "num_kv_heads": 2, "embed_dim": 64, "intermediate_dim": 128,
                           "max_seq_len": 256, "_base_vocab_size_txt": 300, "_n_special_txt": 16}
    cfg.speech.n_dsus = 50
    cfg.data.n_dsus = 50
    resolve_n_dsus(cfg)
    # the synthetic generator draws ids from the production layout; remap into the shrunken vocabulary for this test
    t = Trainer(cfg)
    t.setup()
    V = t._llama_config.vocab_size
    assert V == 300 + 16 + 50 + 2

    class Remap:
        def __init__(self, loader):
            self.loader, self.dataset = loader, loader.dataset

        def __len__(self):
            return len(self.loader)

        def __iter__(self):
            for b in self.loader:
                tok = b["tokens"] % V
                lab = torch.where(b["labels"] == -100, b["labels"], tok)
                yield {"tokens": tok, "labels": lab}

    t.data_train, t.data_dev = Remap(t.data_train), Remap(t.data_dev)
    t._loss_log = []
    sd0 = {k: v.detach().float().cpu().clone() for k, v in t.model.state_dict().items()}
    batches = [{k: v.clone() for k, v in b.items()} for b in itertools_islice(t.data_train, 6)]
    t.train()
    assert t.global_step == 3 and len(t._loss_log) == 3 and t.consumed_samples == 3 * 2 * 2
    if ragged:
        assert any(int((b["labels"][r] != -100).sum()) < b["labels"].shape[1] - 8 for b in batches for r in range(2))
        assert t.unpadded_micro_batches >= 4, t.unpadded_micro_batches
    else:
        assert t.unpadded_micro_batches == 0
    assert t.fused_micro_batches == 6   # (round 5: each window's two micro-batches reach the model as one batch, ssi/data/window.py)
    assert os.path.exists(tmp_path / "ckpt" / "step_3" / "model.safetensors") and os.path.exists(tmp_path / "ckpt" / "training_state.pt")
    rec = t.wandb_logger.records
    assert [r["step"] for r in rec] == [1, 2, 3] and "dev_loss" in rec[-1] and np.isfinite(rec[-1]["dev_loss"])
    # logged after lr_scheduler.step(): the lr of the NEXT step; the first step itself runs at lr_lambda(0) = 0
    assert rec[0]["lr"] == pytest.approx(1e-3 * 1 / 2) and rec[1]["lr"] == pytest.approx(1e-3)
    # CPU oracle, same weights, same batches, fp32
    ref = OracleLlama(**t._llama_config.parameters, rope_cache_len=256)
    ref.load_state_dict(sd0)
    ref.set_num_output_chunks(8)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    from ssi.lr_schedule import get_cosine_schedule_with_warmup
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=3, num_cycles=0.5)
    losses = step_oracle.run_steps(ref, OracleCEWithChunkedOutputLoss(), [batches[0:2], batches[2:4], batches[4:6]], opt, sched)
    print("gpu", t._loss_log, "cpu", losses)
    for a, b in zip(t._loss_log, losses):
        assert abs(a - b) <= tol * abs(b)
    tok_total = sum(int((b["labels"] != -100).sum()) for b in batches)
    assert t.tokens_train_total == tok_total
    counts = {}
    for b in batches:
        for k, v in step_oracle.count_token_types(b["tokens"], t.token_type_ranges, t.tokenizer.pad_id).items():
            counts[k] = counts.get(k, 0) + v
    assert dict(t.token_type_counts_total) == counts
    if dtype_name == "fp32":
        for (k, p), (_, p2) in zip(t.model.named_parameters(), ref.named_parameters()):
            diff = (p.detach().cpu() - p2.detach()).abs()
            # Adam's update is ~lr*sign(g) on the first real step: a near-zero gradient may flip sign with summation order
            assert float(diff.max()) <= 2 * 2 * 1e-3 + 1e-6 and float((diff > 2e-5).float().mean()) < 5e-3, k
    t.cleanup()


def itertools_islice(loader, n):
    import itertools
    return list(itertools.islice(iter(loader), n))


@pytest.mark.parametrize("transposed_copies", [False, True])
def test_bf16_mfma_shaped_model_matches_oracle(transposed_copies):
    """A model whose dims satisfy the MFMA tile rules (so the MFMA GEMMs, split-K weight gradients and MFMA attention all
    run), against the fp32 CPU oracle, with the data gradients either on the untransposed weights (NN form, the default)
    or on [in, out] weight copies (NT form, SSI_DGRAD_NT=1), which an optimizer step must refresh."""
    from oracle import hf_crosscheck as hx
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    from ssi.optimizer import HipAdamW, scale_grads
    params = dict(vocab_size=700, num_layers=2, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=512, intermediate_dim=512)
    sd = hx.seeded_state_dict(params, 21)
    batch = hx.seeded_batch(700, 2, 128, 21)
    model = HipLlamaDecoder(**params, dtype=torch.bfloat16, device=DEV)
    model.transposed_weight_copies = transposed_copies
    model.load_state_dict(sd)
    assert model._mfma_shapes()
    ref_model = hx.oracle_model(params, sd)
    ref = oracle_loss(batch, ref_model, OracleCEWithChunkedOutputLoss())
    ref.backward()
    dbatch = _to_dev(batch)
    loss = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    loss.backward()
    assert abs(loss.item() - ref.item()) <= 1e-2 * abs(ref.item())
    assert ("emb" in model._wt) == transposed_copies and (len(model._wt) == (0 if not transposed_copies else 1 + 4 * 2))
    if transposed_copies:
        for name in ("emb", "L0.wqkv", "L1.w2"):
            assert torch.equal(model._view_t(name), model._view(name).t())
    for (k, p), (_, p2) in zip(model.named_parameters(), ref_model.named_parameters()):
        g, g2 = p.grad.float().cpu(), p2.grad
        rel = float((g - g2).norm() / g2.norm())
        assert rel <= 6e-2, f"{k}: relative gradient error {rel}"
    key_before = model._wt_key
    opt = HipAdamW(model.parameters(), model=model, lr=1e-2)
    scale_grads(model, torch.tensor(1.0))
    opt.step()
    opt.zero_grad(set_to_none=True)
    l2 = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    l2.backward()
    if transposed_copies:
        assert model._wt_key != key_before
        for name in ("emb", "L0.wqkv", "L1.w2"):
            assert torch.equal(model._view_t(name), model._view(name).t())
    assert l2.item() < loss.item()  # the step reduced the loss on the same batch


def test_deferred_batched_attention_weight_gradients_equal_the_per_layer_ones():
    """The weight gradients of the attention projections are computed for a group of layers in one batched launch after the group's
    lowest layer (model.wgrad_group; 3 layers in groups of 2 = one full and one partial group).  Against the same model computing
    them per layer (split-K): every other gradient bit-equal, these two within a bf16 rounding of the fp32 sums; also when a second
    micro-batch accumulates on top."""
    from oracle import hf_crosscheck as hx
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    params = dict(vocab_size=700, num_layers=3, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=512, intermediate_dim=512)
    sd = hx.seeded_state_dict(params, 31)
    batches = [_to_dev(hx.seeded_batch(700, 2, 128, 31 + i)) for i in range(2)]
    grads = {}
    for group in (2, 1):
        model = HipLlamaDecoder(**params, dtype=torch.bfloat16, device=DEV)
        model.wgrad_group = group
        model.load_state_dict(sd)
        for b in batches:
            compute_loss(b, model, CEWithChunkedOutputLoss()).backward()
        grads[group] = {k: p.grad.float().clone() for k, p in model.named_parameters()}
        used = set(model._arena.buf)
        assert ("dqkv.all" in used) == (group > 1)
    for k, g in grads[2].items():
        if any(t in k for t in ("q_proj", "k_proj", "v_proj", "output_proj")):
            diff = (g - grads[1][k]).abs()
            assert float(diff.max()) <= 2 ** -6 * float(grads[1][k].abs().max()), k
        else:
            assert torch.equal(g, grads[1][k]), k


def test_head_weight_gradient_tail_round_is_split_over_k(monkeypatch):
    """The tied head's weight gradient: output tiles beyond the last whole round of the 256 CUs go to a second, K-split launch
    (HipLlamaDecoder._head_wgrad_main_rows).  Same gradients as the single launch up to the order of the fp32 partial sums; the rows of the
    whole rounds bit-equal.  V = 67 500 -> 264 row tiles x 1 column tile = one round + 8 tiles."""
    from oracle import hf_crosscheck as hx
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    params = dict(vocab_size=67_500, num_layers=1, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=1024, intermediate_dim=512)
    g = torch.Generator().manual_seed(5)
    batch = _to_dev(hx.seeded_batch(67_500, 2, 640, 41))
    grads = {}
    for on in ("1", "0"):
        monkeypatch.setenv("SSI_HEAD_WGRAD_TAIL", on)
        torch.manual_seed(3)
        model = HipLlamaDecoder(**params, dtype=torch.bfloat16, device=DEV)
        with torch.no_grad():
            model._flat.copy_((torch.randn(model._flat.numel(), generator=g.manual_seed(5)) * 0.05).to(torch.bfloat16))
        assert model._head_wgrad_main_rows(1280) == (256 * 256 if on == "1" else 0)
        compute_loss(batch, model, CEWithChunkedOutputLoss()).backward()
        grads[on] = model.tok_embeddings.weight.grad.float().clone()
        assert ("ws.splitk.head" in model._arena.buf) == (on == "1")
    assert torch.equal(grads["1"][:65_536], grads["0"][:65_536])
    diff = (grads["1"][65_536:] - grads["0"][65_536:]).abs()
    assert float(grads["0"][65_536:].abs().max()) > 0 and float(diff.max()) <= 2 ** -6 * float(grads["0"].abs().max())


def test_hf_format_checkpoint_loads_into_the_hip_model_and_matches_hf_llama(tmp_path):
    """SURVEY.md §8f rank 3, end to end: a randomly initialised HF ``LlamaForCausalLM`` (built from a local config, no hub) is
    written as an HF model directory (sharded safetensors + config.json), read back through ``FullModelHFCheckpointer`` (key map
    and q/k row permutation) into the HIP model, whose fp32 logits must equal HF's own forward; the checkpoint the HIP model
    then writes loads back into HF-Llama unchanged."""
    import json
    from safetensors.torch import load_file, save_file
    from oracle import hf_crosscheck as hx
    from ssi.checkpoint import FullModelHFCheckpointer
    from ssi.constants import MODEL_KEY
    from ssi.model import HipLlamaDecoder
    params = dict(vocab_size=515, num_layers=2, num_heads=8, num_kv_heads=2, embed_dim=128, max_seq_len=64, intermediate_dim=256)
    hf = hx.build_hf(params, hx.seeded_state_dict(params, 31))  # HF module holding HF-ordered weights
    sd_hf = {k: v.detach().clone() for k, v in hf.state_dict().items() if k != "lm_head.weight" and "rotary" not in k}  # tied: one copy on disk
    src = tmp_path / "hf"
    src.mkdir()
    keys = sorted(sd_hf)
    save_file({k: sd_hf[k] for k in keys[: len(keys) // 2]}, str(src / "model-00001-of-00002.safetensors"), metadata={"format": "pt"})
    save_file({k: sd_hf[k] for k in keys[len(keys) // 2:]}, str(src / "model-00002-of-00002.safetensors"), metadata={"format": "pt"})
    (src / "config.json").write_text(json.dumps({"num_attention_heads": 8, "num_key_value_heads": 2, "hidden_size": 128,
                                                 "num_hidden_layers": 2, "vocab_size": 515, "tie_word_embeddings": True}))
    ck = FullModelHFCheckpointer(src, None, output_dir=tmp_path / "out")
    model = HipLlamaDecoder(**params, dtype=torch.float32, device=DEV)
    model.load_state_dict(ck.load_checkpoint()[MODEL_KEY])
    model.set_num_output_chunks(0)
    tokens = torch.randint(0, 515, (2, 40), generator=torch.Generator().manual_seed(32))
    with torch.no_grad():
        ref = hf(input_ids=tokens).logits
        got = model(tokens=tokens.to(DEV)).float().cpu()
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # save from the HIP model, reload into a fresh HF-Llama: bit-identical weights, same logits
    step_dir = ck.save_model_checkpoint(model.state_dict(), 1)
    written = {}
    for f in sorted(step_dir.glob("*.safetensors")):
        written.update(load_file(str(f)))
    assert written.keys() == sd_hf.keys() and all(torch.equal(written[k], sd_hf[k]) for k in sd_hf)


@pytest.mark.parametrize("dtype_name", ["fp32", "bf16"])
def test_packed_batch_matches_oracle_with_block_causal_mask(dtype_name):
    """a packed batch (PackedDataset -> padded_collate_packed: tokens, labels, input_pos) through
    ``compute_loss`` on the HIP model (block-causal attention + per-document RoPE positions derived from input_pos) against the
    oracle fed torchtune's dense block-causal mask and the same input_pos: loss and every gradient.  fp32 = generic kernels,
    bf16 = MFMA-shaped model (MFMA attention with the document masks, S padded to the MFMA tile)."""
    from oracle import hf_crosscheck as hx
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.data import PackedDataset, packed_block_causal_mask, padded_collate_packed
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    bf16 = dtype_name == "bf16"
    params = (dict(vocab_size=700, num_layers=2, num_heads=4, num_kv_heads=2, embed_dim=256, max_seq_len=512, intermediate_dim=512) if bf16
              else dict(vocab_size=515, num_layers=2, num_heads=8, num_kv_heads=2, embed_dim=128, max_seq_len=256, intermediate_dim=256))
    S = 320 if bf16 else 100   # bf16: padded to 384 inside fused_loss
    g = torch.Generator().manual_seed(51)
    lengths = [57, 130, 9, 77, 160, 33, 101, 64, 12] if bf16 else [31, 7, 44, 13, 50, 26]
    samples = []
    for n in lengths:
        toks = torch.randint(0, params["vocab_size"], (n,), generator=g).tolist()
        labs = [(-100 if i < 3 else t) for i, t in enumerate(toks)]   # a masked prompt span at the head of every document
        samples.append({"tokens": toks, "labels": labs})
    packs = PackedDataset(samples, max_seq_len=S, padding_idx=0)
    batch = padded_collate_packed([packs[0], packs[1]])
    sd = hx.seeded_state_dict(params, 52)
    ref_model = hx.oracle_model(params, sd)
    ref_batch = {"tokens": batch["tokens"], "labels": batch["labels"], "input_pos": batch["input_pos"],
                 "mask": packed_block_causal_mask(batch["seq_lens"])}
    ref = oracle_loss(ref_batch, ref_model, OracleCEWithChunkedOutputLoss())
    ref.backward()
    model = HipLlamaDecoder(**params, dtype=torch.bfloat16 if bf16 else torch.float32, device=DEV)
    model.load_state_dict(sd)
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    loss = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    loss.backward()
    tol_l, tol_g = (1e-2, 6e-2) if bf16 else (1e-5, 5e-3)
    assert abs(loss.item() - ref.item()) <= tol_l * abs(ref.item())
    for (k, p), (_, p2) in zip(model.named_parameters(), ref_model.named_parameters()):
        rel = float((p.grad.float().cpu() - p2.grad).norm() / p2.grad.norm())
        assert rel <= tol_g, f"{k}: relative gradient error {rel}"
    # the same rows WITHOUT input_pos are plain causal rows: a different loss (documents then see each other)
    plain = compute_loss({"tokens": dbatch["tokens"], "labels": dbatch["labels"]}, model, CEWithChunkedOutputLoss())
    assert abs(plain.item() - loss.item()) > 1e-4 * abs(loss.item())
    # eval mode and the unfused route (model(tokens, input_pos=...) -> chunks -> CE) agree with the fused training loss
    model.eval()
    with torch.no_grad():
        l_eval = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    assert abs(l_eval.item() - loss.item()) <= 1e-6 * abs(loss.item()) + (1e-3 if bf16 else 0)


def test_device_prefetcher_delivers_device_batches_in_order():
    from ssi.data import DevicePrefetcher, setup_synthetic_data
    loader, _ = setup_synthetic_data(n_samples=24, seq_len=96, batch_size=4, n_dsus=100, shuffle=False, fixed_len=False)
    ref = [b for b in loader]
    got = [b for b in DevicePrefetcher(loader, DEV, depth=2)]
    assert len(got) == len(ref) == 6
    for a, b in zip(ref, got):
        assert b["tokens"].is_cuda and b["labels"].is_cuda
        assert torch.equal(b["tokens"].cpu(), a["tokens"]) and torch.equal(b["labels"].cpu(), a["labels"])


def test_full_size_step_properties():
    """BASELINE config A at full size (Llama-3.2-1B + 5000 DSUs, B=8, S=2048, bf16): size-independent properties.
    (1) random-init loss ~ ln V; (2) bitwise reproducibility of loss and gradients; (3) gradient accumulation is additive;
    (4) eval-mode loss == training loss; (5) token-type counts partition the batch."""
    import copy
    import math
    from ssi.data import synthetic_batch
    from ssi.llama_configs import configllama3_2_1b
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    from ssi.train_utils import count_token_types, get_token_type_ranges
    cfg = copy.deepcopy(configllama3_2_1b)
    cfg.n_dsus, cfg.modality_tokens = 5000, True
    model = HipLlamaDecoder(**cfg.parameters, dtype=torch.bfloat16, device=DEV, rope_cache_len=2048)
    torch.manual_seed(0)
    with torch.no_grad():
        model._flat.normal_(0.0, 0.02)
        model._view("emb")[cfg.vocab_size:].zero_()
        for p, name, _ in model._param_src:
            if name.endswith("norm"):
                p.fill_(1.0)
    model.train()
    loss_fn = CEWithChunkedOutputLoss()
    b1 = _to_dev(synthetic_batch(8, 2048, 5000, index=0))
    b2 = _to_dev(synthetic_batch(8, 2048, 5000, index=1))
    ranges = get_token_type_ranges(cfg)
    counts = count_token_types(b1["tokens"], ranges, 133_006)
    assert sum(counts[k] for k in ranges) == 8 * 2048 and counts["total"] <= 8 * 2048 and counts["dsu"] > counts["text"]

    def run(batch):
        model.zero_grad()
        loss = compute_loss(batch, model, loss_fn)
        loss.backward()
        return loss.item(), model._flat_grad.clone()

    l1, g1 = run(b1)
    assert abs(l1 - math.log(cfg.vocab_size)) < 1.0 and math.isfinite(l1)           # (1)
    l1b, g1b = run(b1)
    assert l1 == l1b and torch.equal(g1, g1b)                                        # (2)
    l2, g2 = run(b2)
    model.zero_grad()
    compute_loss(b1, model, loss_fn).backward()
    compute_loss(b2, model, loss_fn).backward()                                      # accumulates into the same buffer
    acc = model._flat_grad.float()
    ref = g1.float() + g2.float()
    err = float((acc - ref).norm() / ref.norm())
    assert err < 5e-3, err                                                           # (3) up to bf16 accumulation rounding
    assert float(model._view("emb", None, model._flat_grad)[cfg.vocab_size:].abs().max()) == 0.0  # pad rows stay zero
    model.eval()
    with torch.inference_mode():
        le = compute_loss(b1, model, loss_fn).item()
    assert abs(le - l1) <= 1e-6 * abs(l1)                                            # (4)


@pytest.mark.parametrize("ga", [1, 2])
@pytest.mark.parametrize("packed", [False, True])
def test_trainer_on_sft_samples_from_a_json_file_and_a_tokenizer_file(tmp_path, monkeypatch, packed, ga):
    """The data front end feeding the step: speech units + transcripts in a local json file, a (toy) extended tokenizer.model,
    ``setup_sft_data`` -> padded or packed batches -> Trainer.train(); per-step losses against the CPU step oracle on the same batches.
    ``ga = 2``: the two batches are one accumulation window, which reaches the model as ONE batch (``ssi/data/window.py``: padded rows end to
    end without their padding, packs stacked row by row) — against the oracle's loop over the two micro-batches."""
    import json
    from oracle import step_oracle
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from ssi.config import compose
    from ssi.lr_schedule import get_cosine_schedule_with_warmup
    from ssi.train_utils import resolve_n_dsus
    from ssi.trainer import Trainer
    from conftest import PKG
    from test_data_pipeline import N_TXT, N_UNITS, ROWS, toy_ranks
    from ssi.tokenizer import dump_tiktoken_bpe
    monkeypatch.setenv("HF_DATASETS_OFFLINE", "1")
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    dump_tiktoken_bpe(toy_ranks(), tmp_path / "tokenizer.model")
    (tmp_path / "train.jsonl").write_text("\n".join(json.dumps(r) for r in ROWS))
    cfg = compose(os.path.join(PKG, "conf"), "sft", [
        "data=sft/mls-speechtokenizer-rvq_0", "dtype=fp32", f"max_steps={2 // ga}", f"gradient_accumulation_steps={ga}", "tokenizer.max_seq_len=128",
        f"eval_steps={2 // ga}", "save_steps=100", "lr_scheduler.num_warmup_steps=1", "optimizer.lr=1e-3", f"output_dir={tmp_path}",
        f"checkpointer.output_dir={tmp_path}/ckpt", f"checkpointer.checkpoint_dir={tmp_path}/none", "checkpointer.allow_random_init=true", "data.train.shuffle=false",
        f"tokenizer.path={tmp_path}/tokenizer.model", "tokenizer.verbose=false",
        "data.train.dataset.source=json", "data.dev.dataset.source=json", "data.train.dataset.n_samples=null", "data.dev.dataset.n_samples=4",
        "data.dev.dataset.split=train", f"data.train.packed={'true' if packed else 'false'}", "data.train.dataloader.batch_size=2",
    ])
    cfg.data.train.dataset.data_files = str(tmp_path / "train.jsonl")
    cfg.data.dev.dataset.data_files = str(tmp_path / "train.jsonl")
    cfg.model_overrides = {"num_layers": 2, "num_heads": 4, "num_kv_heads": 2, "embed_dim": 64, "intermediate_dim": 128,
                           "max_seq_len": 256, "_base_vocab_size_txt": N_TXT, "_n_special_txt": 256}
    cfg.speech.n_dsus = N_UNITS
    cfg.data.n_dsus = N_UNITS
    resolve_n_dsus(cfg)
    t = Trainer(cfg)
    t.setup()
    assert t.tokenizer.vocab_size == t._llama_config.vocab_size == N_TXT + N_UNITS + 2 + 256
    assert type(t.data_train.dataset).__name__ == ("PackedDataset" if packed else "SFTDataset")
    t._loss_log = []
    sd0 = {k: v.detach().float().cpu().clone() for k, v in t.model.state_dict().items()}
    batches = [{k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()} for b in itertools_islice(t.data_train, 2)]
    assert ("input_pos" in batches[0]) == packed and batches[0]["tokens"].shape[0] == 2
    t.train()
    assert t.global_step == 2 // ga and len(t._loss_log) == 2 // ga and t.fused_micro_batches == (2 if ga == 2 else 0)
    ref = OracleLlama(**t._llama_config.parameters, rope_cache_len=256)
    ref.load_state_dict(sd0)
    ref.set_num_output_chunks(8)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=1, num_training_steps=2 // ga, num_cycles=0.5)
    if packed:
        from ssi.data import packed_block_causal_mask
        batches = [{**b, "mask": packed_block_causal_mask(b["seq_lens"])} for b in batches]
    windows = [batches[0:1], batches[1:2]] if ga == 1 else [batches[0:2]]
    losses = step_oracle.run_steps(ref, OracleCEWithChunkedOutputLoss(), windows, opt, sched)
    print("gpu", t._loss_log, "cpu", losses)
    for a, b in zip(t._loss_log, losses):
        assert abs(a - b) <= 2e-5 * abs(b)
    # the unit and modality counters see the real ids (deduplicated unit runs between the two modality tokens)
    counts = {}
    for b in batches:
        for k, v in step_oracle.count_token_types(b["tokens"], t.token_type_ranges, t.tokenizer.pad_id).items():
            counts[k] = counts.get(k, 0) + v
    assert dict(t.token_type_counts_total) == counts and counts["dsu"] > 0 and counts["modality"] >= 2 * 4
    t.cleanup()
