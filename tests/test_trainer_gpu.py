"""GPU: the trainer's behavioural contracts the reference pins in its own tests, on the HIP model.

* resume equivalence (reference tests/test_resume_equivalence.py:226-297): 8 steps uninterrupted == 4 steps, save, rebuild from
  ``step_4/`` + ``training_state.pt``, 4 more — losses bit-equal, in fp32 and in bf16 (every kernel on the path is deterministic);
* dev loss (reference ssi/eval.py:24-41): ``Trainer._evaluate()`` against the oracle's sum(loss_b * n_b) / sum(n_b) on the same batches;
* the CPT route (``config_name=cpt``: labels = tokens, TextCompletion-shaped batches) with global-norm clipping through
  ``Trainer.train()`` against the CPU step oracle — also the only place the clip path runs end to end on the GPU."""
import itertools
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

SMALL = {"num_layers": 2, "num_heads": 4, "num_kv_heads": 2, "embed_dim": 64, "intermediate_dim": 128, "max_seq_len": 256,
         "_base_vocab_size_txt": 300, "_n_special_txt": 16}
MFMA_SMALL = {"num_layers": 2, "num_heads": 4, "num_kv_heads": 2, "embed_dim": 256, "intermediate_dim": 512, "max_seq_len": 512,
              "_base_vocab_size_txt": 300, "_n_special_txt": 16}


class _Remap:
    """The synthetic generator draws ids from the production vocabulary layout; fold them into the shrunken test vocabulary."""

    def __init__(self, loader, vocab):
        self.loader, self.dataset, self.vocab, self.label_first = loader, loader.dataset, vocab, False

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for b in self.loader:
            tok = b["tokens"] % self.vocab
            labels = torch.where(b["labels"] == -100, b["labels"], tok)
            if self.label_first:  # the reference's CPT rows: labels = tokens, column 0 included (/root/reference/ssi/data/cpt.py:153)
                labels[:, 0] = tok[:, 0]
            yield {"tokens": tok, "labels": labels}


def _trainer(tmp, name, config_name="sft", dtype="fp32", overrides=(), model=SMALL, seq=96):
    from conftest import PKG
    from ssi.config import compose
    from ssi.train_utils import resolve_n_dsus
    from ssi.trainer import Trainer, set_seed
    from ssi.constants import SEED
    out = tmp / name
    data = "sft/mls-speechtokenizer-rvq_0" if config_name == "sft" else "cpt/mls-speechtokenizer-rvq_0"
    cfg = compose(os.path.join(PKG, "conf"), config_name, [
        f"data={data}", f"dtype={dtype}", f"tokenizer.max_seq_len={seq}", "data.train.dataloader.batch_size=2", "data.dev.dataloader.batch_size=2",
        "data.train.dataset.n_samples=24", "data.dev.dataset.n_samples=6", "gradient_accumulation_steps=2", "eval_steps=1000", "save_steps=1000",
        "lr_scheduler.num_warmup_steps=100", "optimizer.lr=2e-2", f"output_dir={out}", f"checkpointer.output_dir={out}/checkpoints",
        f"checkpointer.checkpoint_dir={out}/none", "checkpointer.allow_random_init=true", *overrides])
    cfg.model_overrides = dict(model)
    cfg.speech.n_dsus = 50
    cfg.data.n_dsus = 50
    resolve_n_dsus(cfg)
    set_seed(SEED)
    t = Trainer(cfg)
    t.setup()
    V = t._llama_config.vocab_size
    t.data_train, t.data_dev = _Remap(t.data_train, V), _Remap(t.data_dev, V)
    t._loss_log = []
    return t


@pytest.mark.parametrize("dtype,model", [("fp32", SMALL), ("bf16", MFMA_SMALL)])
def test_resumed_losses_match_uninterrupted(tmp_path, dtype, model):
    total, save_at = 8, 4   # 24 samples / batch 2 / grad-accum 2 = 6 steps per epoch: the resume lands mid-epoch, the run crosses an epoch
    seq = 96 if dtype == "fp32" else 128
    full = _trainer(tmp_path, "full", dtype=dtype, model=model, seq=seq, overrides=[f"max_steps={total}"])
    assert full.geometry.steps_per_epoch == 6
    full.train()
    losses_full = list(full._loss_log)
    final_full = {k: v.detach().clone() for k, v in full.model.state_dict().items()}
    full.cleanup()
    assert len(losses_full) == total and len(set(losses_full)) == total
    del full

    b1 = _trainer(tmp_path, "b1", dtype=dtype, model=model, seq=seq, overrides=[f"max_steps={save_at}", f"save_steps={save_at}", f"eval_steps={save_at}"])
    b1.train()
    assert b1._loss_log == losses_full[:save_at], "pre-resume losses differ"
    b1.cleanup()
    ckpt = tmp_path / "b1" / "checkpoints"
    assert (ckpt / "training_state.pt").exists() and (ckpt / f"step_{save_at}" / "model.safetensors").exists()
    del b1

    b2 = _trainer(tmp_path, "b2", dtype=dtype, model=model, seq=seq, overrides=[
        f"max_steps={total}", f"checkpointer.checkpoint_dir={ckpt}/step_{save_at}", "checkpointer.allow_random_init=false",
        f"checkpointer.training_state_checkpoint={ckpt}/training_state.pt"])
    assert b2.global_step == save_at and b2.consumed_samples == save_at * 2 * 2 and b2.optimizer._step_count == save_at
    assert b2.tokens_train_total > 0 and dict(b2.token_type_counts_total)
    assert float(b2.optimizer._exp_avg.abs().max()) > 0   # the moments arrived in the flat buffers
    b2.train()
    print("full   ", losses_full, "\nresumed", b2._loss_log)
    assert b2._loss_log == losses_full[save_at:], "losses after the resume differ from the uninterrupted run"
    assert b2.global_step == total
    for k, v in b2.model.state_dict().items():
        assert torch.equal(v, final_full[k]), k       # same weights, bit for bit, after 8 steps either way
    b2.cleanup()
    # a changed batch size breaks the step-to-data mapping: refused unless force_resume (reference train_utils.py:110-126)
    with pytest.raises(ValueError, match="batch_size"):
        _trainer(tmp_path, "b3", dtype=dtype, model=model, seq=seq, overrides=[
            f"max_steps={total}", f"checkpointer.checkpoint_dir={ckpt}/step_{save_at}", "data.train.dataloader.batch_size=3",
            f"checkpointer.training_state_checkpoint={ckpt}/training_state.pt"])


def test_dev_loss_matches_the_oracle(tmp_path):
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from oracle.llama_oracle import compute_loss as oracle_loss
    t = _trainer(tmp_path, "dev", overrides=["max_steps=1", "data.dev.dataset.n_samples=7", "data.dev.dataset.fixed_len=false"])
    dev_batches = [{k: v.clone() for k, v in b.items()} for b in t.data_dev]
    assert len(dev_batches) == 4 and dev_batches[-1]["tokens"].shape[0] == 1        # ragged last batch (drop_last: false)
    ref = OracleLlama(**t._llama_config.parameters, rope_cache_len=256)
    ref.load_state_dict({k: v.detach().float().cpu() for k, v in t.model.state_dict().items()})
    ref.set_num_output_chunks(8)
    num, den = 0.0, 0
    with torch.no_grad():
        for b in dev_batches:
            n_b = int((b["labels"] != -100).sum())                                    # UNSHIFTED count (ssi/eval.py:33-38)
            num += float(oracle_loss(b, ref, OracleCEWithChunkedOutputLoss())) * n_b
            den += n_b
    # Round 5: by default the dev batches reach the model several at a time as one batch (ssi/eval.py, eval_join_batches: 16 — here all four,
    # the one-row last batch included; with 3: a group of three and a lone batch); the sum of loss_b x n_b is kept by per-token weights
    from ssi.data import window
    joined_sizes = []
    real = window.fuse_micro_batches
    window.fuse_micro_batches = lambda bs, **k: (joined_sizes.append(len(bs)), real(bs, **k))[1]
    try:
        got = t._evaluate()
        t.cfg.eval_join_batches = 3
        got3 = t._evaluate()
        t.cfg.eval_join_batches = 0
        got0 = t._evaluate()
    finally:
        window.fuse_micro_batches = real
    assert joined_sizes == [4, 3]
    print(f"dev loss {got:.7f} (joined) {got3:.7f} (in threes) {got0:.7f} (batch by batch) vs oracle {num / den:.7f}")
    for value in (got, got3, got0):
        assert abs(value - num / den) <= 1e-5 * abs(num / den)
    assert t.model.training                                                           # back in training mode afterwards
    # the value logged by the training loop at an eval step is that same number
    t2 = _trainer(tmp_path, "dev2", overrides=["max_steps=1", "eval_steps=1", "save_steps=1", "optimizer.lr=0.0",
                                               "data.dev.dataset.n_samples=7", "data.dev.dataset.fixed_len=false"])
    t2.train()
    assert t2.wandb_logger.records[-1]["dev_loss"] == pytest.approx(num / den, rel=1e-5)
    t.cleanup()


@pytest.mark.parametrize("dtype,tol", [("fp32", 2e-5), ("bf16", 2e-2)])
def test_cpt_trainer_with_clipping_matches_cpu_step_oracle(tmp_path, dtype, tol):
    """BASELINE config C's route at test size: ``config_name=cpt`` (labels = tokens, TextCompletion batches of 16 rows), grad-accum 2,
    global-norm clipping, three steps — per-step losses and the logged gradient norms against the CPU step oracle."""
    from oracle import step_oracle
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from ssi.lr_schedule import get_cosine_schedule_with_warmup
    model = SMALL if dtype == "fp32" else MFMA_SMALL
    t = _trainer(tmp_path, "cpt", config_name="cpt", dtype=dtype, model=model, seq=128, overrides=[
        "max_steps=3", "clip_grad_norm=0.05", "data.train.dataloader.batch_size=4", "data.train.shuffle=false", "lr_scheduler.num_warmup_steps=2",
        "optimizer.lr=1e-3"])
    assert t.cfg.config_name == "cpt" and t.cfg.clip_grad_norm == 0.05
    sd0 = {k: v.detach().float().cpu().clone() for k, v in t.model.state_dict().items()}
    batches = [{k: v.clone() for k, v in b.items()} for b in itertools.islice(iter(t.data_train), 6)]
    assert all(bool(((b["labels"] == b["tokens"]) | (b["labels"] == -100)).all()) for b in batches)
    assert int((batches[0]["labels"] != -100).sum()) > 0.9 * batches[0]["tokens"].numel()   # CPT: (nearly) every position is a target
    t.train()
    assert t.global_step == 3
    ref = OracleLlama(**t._llama_config.parameters, rope_cache_len=512)
    ref.load_state_dict(sd0)
    ref.set_num_output_chunks(8)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    sched = get_cosine_schedule_with_warmup(opt, num_warmup_steps=2, num_training_steps=3, num_cycles=0.5)
    losses, norms = [], []
    for w in (batches[0:2], batches[2:4], batches[4:6]):
        lr, nt = 0.0, 0
        for mb in w:
            lb, n = step_oracle.train_step(ref, OracleCEWithChunkedOutputLoss(), mb)
            lr, nt = lr + lb, nt + n
        norms.append(step_oracle.optimizer_step(ref, opt, nt, clip_grad_norm=0.05, lr_scheduler=sched))
        losses.append(lr / nt)
    got_norms = [r["grad_norm"] for r in t.wandb_logger.records]
    print("gpu", t._loss_log, got_norms, "\ncpu", losses, norms)
    for a, b in zip(t._loss_log, losses):
        assert abs(a - b) <= tol * abs(b)
    for a, b in zip(got_norms, norms):
        assert b > 0.05, "the test must actually clip"
        assert abs(a - b) <= (1e-3 if dtype == "fp32" else 5e-2) * b
    t.cleanup()


@pytest.mark.parametrize("what", ["label", "token", "position"])
def test_ids_the_kernels_must_refuse_are_raised_by_the_trainer(tmp_path, what):
    """torch's embedding / cross_entropy device-assert on ids outside the vocabulary; the HIP kernels write zeros, COUNT, and the trainer raises
    from the one read-back it does per accumulation window anyway — labels and tokens outside [0, V), and packed positions beyond the RoPE table (the
    kernel clamps them so that the table is never over-read; ADVICE r2: a clamp alone would hide the data bug)."""
    t = _trainer(tmp_path, f"bad_{what}", overrides=["max_steps=1"])
    V = t._llama_config.vocab_size
    batch = {k: v.clone() for k, v in next(iter(t.data_train)).items()}
    t._train_step({k: v.clone() for k, v in batch.items()})          # a clean micro-batch passes
    if what == "label":
        batch["labels"][0, 5] = V + 3
        match = "labels outside"
    elif what == "token":
        batch["tokens"][1, 7] = V
        batch["labels"][1, 7] = -100
        match = "token ids outside"
    else:
        B, S = batch["tokens"].shape
        batch["input_pos"] = torch.arange(S).expand(B, S).clone()
        batch["input_pos"][0, S - 1] = t.model._rope.shape[0] + 10    # beyond the RoPE cache
        match = "input_pos entries outside"
    t._train_step(batch)                                              # queued: nothing is read back inside a micro-batch ...
    with pytest.raises(IndexError, match=match):
        t.num_tokens_step                                             # ... the window's one read-back raises (here, or in _optimizer_step)
    t.cleanup()


@pytest.mark.parametrize("dtype,model", [("fp32", SMALL), ("bf16", MFMA_SMALL)])
def test_adamw_under_the_backward_changes_no_bit(tmp_path, dtype, model):
    """Round 5 (``adamw_under_backward``, default on): on one GPU without clipping the trainer arms the optimizer before a window's last
    backward, and every bucket of gradients is applied on a side stream the moment the backward has finished it.  Same kernel, same factor
    (1 / the window's token count, taken on the device), same step number: 6 optimizer steps of 2 micro-batches through ``Trainer.train()``
    give the same logged losses and the same final weights, bit for bit, as the run with the switch off; with clipping configured the
    optimizer is never armed."""
    seq = 96 if dtype == "fp32" else 128
    runs = {}
    for name, extra in (("on", []), ("off", ["adamw_under_backward=false"])):
        t = _trainer(tmp_path, name, dtype=dtype, model=model, seq=seq, overrides=["max_steps=6", *extra])
        armed = []
        real = t.optimizer.overlap_with_backward
        t.optimizer.overlap_with_backward = lambda s, real=real, armed=armed: (armed.append(1), real(s))[1]
        t.train()
        runs[name] = (list(t._loss_log), {k: v.detach().clone() for k, v in t.model.state_dict().items()}, t.optimizer._step_count, len(armed))
        t.cleanup()
        del t
    (l_on, w_on, n_on, armed_on), (l_off, w_off, n_off, armed_off) = runs["on"], runs["off"]
    assert armed_on == 6 and armed_off == 0 and n_on == n_off == 6
    assert l_on == l_off and len(set(l_on)) == 6
    assert all(torch.equal(w_on[k], w_off[k]) for k in w_on)
    clip = _trainer(tmp_path, "clip", dtype=dtype, model=model, seq=seq, overrides=["max_steps=2", "clip_grad_norm=1.0"])
    clip.optimizer.overlap_with_backward = lambda s: (_ for _ in ()).throw(AssertionError("armed although gradients are clipped"))
    clip.train()
    clip.cleanup()


@pytest.mark.parametrize("config_name,dtype,model", [("sft", "fp32", SMALL), ("cpt", "fp32", SMALL), ("cpt", "bf16", MFMA_SMALL)])
def test_a_window_run_as_one_batch_is_the_micro_batch_loop(tmp_path, config_name, dtype, model):
    """Round 5 (``fuse_accumulation_window``, default on; ``ssi/data/window.py``): the micro-batches of an accumulation window reach the model
    as ONE packed batch.  The reference's loop (``trainer.py:385-424``) normalises each micro-batch by its own count of shifted labels and
    weights it by its own count of unshifted ones; the joined batch keeps that per token (``ssi_ce_fwd_weighted``) — the CPT rows, ragged
    with every token a label, are the case where the ratios differ, the SFT rows (BOS masked) the one where no weights are needed.  6
    optimizer steps of 3 micro-batches through ``Trainer.train()`` with the switch on and off: same counters, losses and final weights equal
    up to the summation order (fp32) / the rounding of the gradient accumulator (bf16: the joined window rounds once where the loop rounds
    after every micro-batch)."""
    seq = 96 if dtype == "fp32" else 128
    runs = {}
    for name, extra in (("joined", []), ("loop", ["fuse_accumulation_window=false"])):
        t = _trainer(tmp_path, name, config_name=config_name, dtype=dtype, model=model, seq=seq,
                     overrides=["max_steps=6", "gradient_accumulation_steps=3", "data.train.dataset.n_samples=36",
                                "data.train.dataset.fixed_len=false", *extra])
        t.data_train.label_first = config_name == "cpt"
        weighted, real = [], t.model.fused_loss
        t.model.fused_loss = lambda *a, real=real, weighted=weighted, **k: (weighted.append(k.get("loss_weights") is not None), real(*a, **k))[1]
        t.train()
        runs[name] = dict(losses=list(t._loss_log), w={k: v.detach().float().clone() for k, v in t.model.state_dict().items()}, calls=list(weighted),
                          joined=t.fused_micro_batches, tokens=t.tokens_train_total, counts=dict(t.token_type_counts_total),
                          consumed=t.consumed_samples, step=t.global_step, max_seq=t.max_seq_len_step)
        t.cleanup()
        del t
    a, b = runs["joined"], runs["loop"]
    assert a["joined"] == 18 and b["joined"] == 0 and len(a["calls"]) == 6 and len(b["calls"]) == 18 and not any(b["calls"])
    assert all(a["calls"]) if config_name == "cpt" else not any(a["calls"])
    for k in ("tokens", "counts", "consumed", "step"):
        assert a[k] == b[k], k
    tol = 2e-6 if dtype == "fp32" else 3e-3
    assert len(set(a["losses"])) == 6 and all(abs(x - y) <= tol * abs(y) for x, y in zip(a["losses"], b["losses"])), (a["losses"], b["losses"])
    worst = max(float((a["w"][k] - b["w"][k]).abs().max()) for k in a["w"])
    assert worst <= (2e-3 if dtype == "fp32" else 6e-2), worst   # (lr 2e-2 x 6 steps: AdamW moves every weight by up to 0.12)


class _Edit:
    """Wraps a loader: ``edit(i, batch)`` may change batch ``i`` in place."""

    def __init__(self, loader, edit):
        self.loader, self.dataset, self.edit = loader, loader.dataset, edit

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for i, b in enumerate(self.loader):
            self.edit(i, b)
            yield b


@pytest.mark.parametrize("dtype,model", [("fp32", SMALL), ("bf16", MFMA_SMALL)])
def test_the_boundary_that_does_not_wait_for_the_device_logs_what_the_waiting_one_logs(tmp_path, dtype, model):
    """Round 5 (``lagged_readback``, default on; one GPU): at the accumulation boundary the host takes the window's label count from the host
    side of the batches and goes on; the window's loss and counters come back a step later.  Against the boundary that waits
    (``lagged_readback=false``): the same records (step numbers, losses bit for bit, learning rates, token totals and per-type counts),
    the same weights; a window without any label is skipped the same way; a label outside the vocabulary still ends the run with the
    kernels' error count."""
    seq = 96 if dtype == "fp32" else 128
    runs = {}
    for name, extra in (("lagged", []), ("waiting", ["lagged_readback=false"])):
        t = _trainer(tmp_path, name, dtype=dtype, model=model, seq=seq, overrides=["max_steps=5", "data.train.dataset.fixed_len=false", *extra])

        def no_labels_in_window_two(i, b):   # batches 4 and 5 = the third window (grad-accum 2)
            if i in (4, 5):
                b["labels"][:] = -100
        t.data_train = _Edit(t.data_train, no_labels_in_window_two)
        lagged_calls, real = [], t._optimizer_step_lagged
        t._optimizer_step_lagged = lambda *a, real=real, lagged_calls=lagged_calls: (lagged_calls.append(1), real(*a))[1]
        t.train()
        rec = t.wandb_logger.records
        runs[name] = dict(losses=list(t._loss_log), w={k: v.detach().clone() for k, v in t.model.state_dict().items()}, lagged=len(lagged_calls),
                          rec=[{k: r[k] for k in r if k not in ("duration_step", "tokens_per_second_per_gpu", "train_clock_time")} for r in rec],
                          step=t.global_step, tokens=t.tokens_train_total, counts=dict(t.token_type_counts_total), consumed=t.consumed_samples)
        assert all(r["duration_step"] > 0 for r in rec)
        t.cleanup()
        del t
    a, b = runs["lagged"], runs["waiting"]
    assert a["lagged"] == 6 and b["lagged"] == 0            # 5 optimizer steps + the skipped window
    assert a["step"] == b["step"] == 5 and len(a["losses"]) == 5 and a["losses"] == b["losses"]
    assert a["rec"] == b["rec"] and [r["step"] for r in a["rec"]] == [1, 2, 3, 4, 5]
    assert (a["tokens"], a["counts"], a["consumed"]) == (b["tokens"], b["counts"], b["consumed"])
    assert all(torch.equal(a["w"][k], b["w"][k]) for k in a["w"])
    # a label outside the vocabulary: the kernels write zeros and count; the count ends the run when the window's results arrive
    t = _trainer(tmp_path, "bad", dtype=dtype, model=model, seq=seq, overrides=["max_steps=5"])

    def bad_label(i, b):
        if i == 3:
            b["labels"][0, 7] = 10_000_000
    t.data_train = _Edit(t.data_train, bad_label)
    with pytest.raises(IndexError, match="labels outside"):
        t.train()
    t.cleanup()


def test_adamw_under_the_backward_skips_a_window_without_labels():
    """The factor 1 / 0 (a window whose labels are all ignored) reaches the kernel before the host has read the count back: the launches issued
    under the backward must change nothing by themselves — parameters, both moments and the step count stay as they were, the window after
    it equals the same window on an optimizer that never saw the empty one — and a backward that never comes leaves the optimizer usable."""
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    from ssi.optimizer import HipAdamW, scale_grads
    params = dict(vocab_size=300, num_layers=2, num_heads=4, num_kv_heads=2, embed_dim=64, max_seq_len=128, intermediate_dim=128)
    g = torch.Generator().manual_seed(3)
    batch = {"tokens": torch.randint(0, 300, (2, 64), generator=g).to(DEV)}
    batch["labels"] = batch["tokens"].clone()
    empty = {"tokens": batch["tokens"], "labels": torch.full_like(batch["tokens"], -100)}
    states = []
    for with_empty_window in (True, False):
        torch.manual_seed(5)
        model = HipLlamaDecoder(**params, dtype=torch.float32, device=DEV)
        with torch.no_grad():
            model._flat.normal_(0.0, 0.05, generator=torch.Generator(device=DEV).manual_seed(9))
        model.train()
        opt = HipAdamW(model.parameters(), model=model, lr=1e-2)
        loss_fn = CEWithChunkedOutputLoss()
        if with_empty_window:
            before = model._flat.clone()
            n = (empty["labels"] != -100).sum()
            assert opt.overlap_with_backward(1.0 / n.to(torch.float32))          # 1 / 0 = inf on the device
            (compute_loss(empty, model, loss_fn) * n).backward()                  # (loss is 0 / 0 = NaN, as the reference's)
            opt.cancel_overlap()
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            assert torch.equal(model._flat, before) and opt._step_count == 0
            assert float(opt._exp_avg.abs().max()) == 0.0 and float(opt._exp_avg_sq.abs().max()) == 0.0
            assert opt.overlap_with_backward(torch.ones(1, device=DEV))           # armed, but no backward follows: step() does the plain update
        n = (batch["labels"] != -100).sum()
        if not with_empty_window:
            assert opt.overlap_with_backward(1.0 / n.to(torch.float32))
        (compute_loss(batch, model, loss_fn) * n).backward()
        scale_grads(model, 1.0 / int(n))
        opt.step()
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        assert opt._step_count == 1
        states.append((model._flat.clone(), opt._exp_avg.clone(), opt._exp_avg_sq.clone()))
    for a, b in zip(*states):
        assert torch.equal(a, b)
