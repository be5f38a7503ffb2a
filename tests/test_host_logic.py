"""CPU: host-side logic of the drop-in — config layer, model-shape config, token ranges, trainer state machine (mock
components, as /root/reference/tests/test_trainer.py does), schedule, synthetic data, collate."""
import math
import time
from unittest.mock import MagicMock, patch

import numpy as np
import pytest
import torch

from ssi.config import DictConfig, MissingMandatoryValue, OmegaConf, compose
from ssi.llama_configs import ConfigLlama3_2, configllama3_2_1b
from ssi.trainer import Trainer, TrainingGeometry
from conftest import PKG
import copy
import os


# ---- config ---------------------------------------------------------------------------------------------------------
def test_config_interpolation_missing_and_overrides():
    cfg = OmegaConf.create({"a": {"b": 3}, "c": "${a.b}", "d": "x-${a.b}-y", "e": "???", "lr": "2e-4"})
    assert cfg.c == 3 and cfg.d == "x-3-y" and cfg.lr == pytest.approx(2e-4)
    assert OmegaConf.missing_keys(cfg) == {"e"}
    with pytest.raises(MissingMandatoryValue):
        _ = cfg.e
    assert cfg.get("e", 7) == 7 and cfg.get("nope") is None
    cfg.a.b = 5
    assert cfg.c == 5
    assert OmegaConf.to_container(OmegaConf.create({"x": "${y}", "y": 1}), resolve=True) == {"x": 1, "y": 1}


def test_hydra_style_composition_matches_reference_layout():
    conf = os.path.join(PKG, "conf")
    cfg = compose(conf, "sft", ["data=sft/mls-hubert_large_ll60k-layer_22", "max_steps=7", "optimizer.lr=1e-3"])
    assert cfg.config_name == "sft" and cfg.tokenizer.max_seq_len == 2048 and cfg.max_steps == 7
    assert cfg.data.n_dsus == 5000 and cfg.data.train.dataloader.batch_size == 2 and cfg.gradient_accumulation_steps == 4
    assert cfg.optimizer.lr == pytest.approx(1e-3) and cfg.optimizer.betas == [0.9, 0.999] and cfg.optimizer.fused is True
    from ssi.train_utils import resolve_n_dsus, validate_train_cfg
    assert cfg.speech.n_dsus is None
    resolve_n_dsus(cfg)
    assert cfg.speech.n_dsus == 5000 and cfg.extended_model_name == "Llama-3.2-1B-5000-dsus"
    validate_train_cfg(cfg)
    cpt = compose(conf, "cpt", ["data=cpt/mls-mimi-srvq_0"])
    assert cpt.tokenizer.max_seq_len == 768 and cpt.data.n_dsus == 2048 and cpt.data.train.dataloader.batch_size == 16
    with pytest.raises(MissingMandatoryValue):
        compose(conf, "sft", [])  # data: ??? must be chosen


def test_validate_train_cfg_rejects_bad_values():
    from ssi.train_utils import validate_train_cfg
    base = {"speech": {"n_dsus": 5000}, "dtype": "bf16", "gradient_accumulation_steps": 1, "max_steps": 1,
            "log_interval": 1, "eval_steps": 2, "save_steps": 4}
    validate_train_cfg(OmegaConf.create(base))
    for bad in ({"dtype": "fp16"}, {"max_steps": 0}, {"save_steps": 3}, {"speech": {"n_dsus": None}}):
        with pytest.raises(ValueError):
            validate_train_cfg(OmegaConf.create({**base, **bad}))


# ---- model shape ------------------------------------------------------------------------------------------------------
def test_vocab_size_and_parameters():
    c = copy.deepcopy(configllama3_2_1b)
    c.update_from_speech_cfg(OmegaConf.create({"n_dsus": 5000, "use_modality_tokens": True}))
    assert c.vocab_size == 133_258
    assert c.parameters == dict(vocab_size=133_258, num_layers=16, num_heads=32, num_kv_heads=8, embed_dim=2048,
                                max_seq_len=131072, intermediate_dim=8192, attn_dropout=0.0, norm_eps=1e-5,
                                rope_base=500_000, scale_factor=32)
    c.n_dsus = 8192
    assert c.vocab_size == 136_450
    with pytest.raises(ValueError):
        c.n_dsus = -1
    with pytest.raises(TypeError):
        c.update_from_speech_cfg({"n_dsus": 1, "use_modality_tokens": True})
    assert configllama3_2_1b.vocab_size == 128_256  # singleton untouched


def test_token_type_ranges_and_cpu_counts():
    from ssi.train_utils import count_token_types, get_token_type_ranges
    c = copy.deepcopy(configllama3_2_1b)
    c.n_dsus, c.modality_tokens = 5000, True
    r = get_token_type_ranges(c)
    assert r == {"text": (0, 127999), "dsu": (128000, 132999), "modality": (133000, 133001), "special_text": (133002, 133257)}
    t = torch.tensor([[0, 127999, 128000, 132999, 133000, 133001, 133002, 133257, 133006]])
    assert count_token_types(t, r, 133006) == {"text": 2, "dsu": 2, "modality": 2, "special_text": 3, "total": 8}


# ---- trainer state machine (mirrors reference tests T-U1..T-U10) --------------------------------------------------------
def test_trainer_construction_and_initial_state():
    cfg = OmegaConf.create({"dummy": True})
    t = Trainer(cfg)
    assert t.cfg is cfg
    for attr in ("model", "tokenizer", "optimizer", "lr_scheduler", "loss_fn", "checkpointer", "wandb_logger", "data_train",
                 "sampler_train", "data_dev", "token_type_ranges", "geometry", "device", "dtype", "world_size"):
        assert getattr(t, attr) is None
    assert (t.global_step, t.consumed_samples, t.tokens_train_total, t.num_tokens_step, t.max_seq_len_step) == (0, 0, 0, 0, 0)
    assert t.wall_clock_offset == 0.0 and t.loss_running == 0.0 and dict(t.token_type_counts_total) == {}
    assert t._loss_log is None


def _dl(n):
    dl = MagicMock(spec=["__len__"])
    dl.__len__ = MagicMock(return_value=n)
    return dl


def test_geometry():
    cfg = OmegaConf.create({"data": {"train": {"dataloader": {"batch_size": 16}}}, "gradient_accumulation_steps": 4, "max_steps": 100})
    g = TrainingGeometry.from_config(cfg, _dl(100), world_size=1)
    assert (g.batch_size, g.batches_per_epoch, g.steps_per_epoch, g.usable_batches, g.n_epochs, g.gradient_accumulation_steps,
            g.world_size) == (16, 100, 25, 100, 4, 4, 1)
    cfg2 = OmegaConf.create({"data": {"train": {"dataloader": {"batch_size": 8}}}, "gradient_accumulation_steps": 3, "max_steps": 30})
    g2 = TrainingGeometry.from_config(cfg2, _dl(50), world_size=2)
    assert g2.steps_per_epoch == 16 and g2.usable_batches == 48 and g2.n_epochs == 2
    with pytest.raises(ValueError):
        TrainingGeometry.from_config(cfg2, _dl(2), world_size=1)
    with pytest.raises(Exception):
        g.batch_size = 3  # frozen


def _trainer_for_optimizer_step(clip=None):
    cfg = OmegaConf.create({"gradient_accumulation_steps": 2, "clip_grad_norm": clip, "eval_steps": 100, "log_interval": 1, "save_steps": 100})
    t = Trainer(cfg)
    t.world_size, t.device = 1, torch.device("cpu")
    param = torch.nn.Parameter(torch.randn(4, 4))
    t.model = MagicMock(spec=["parameters", "named_parameters"])
    t.model.parameters.return_value = [param]
    t.model.named_parameters.return_value = [("weight", param)]
    t.optimizer = MagicMock()
    t.optimizer.param_groups = [{"lr": 2e-4}]
    t.lr_scheduler, t.wandb_logger, t.checkpointer = MagicMock(), MagicMock(), MagicMock()
    t.geometry = TrainingGeometry(2, 20, 10, 20, 1, 2, 1)
    t.loss_running, t.num_tokens_step, t.max_seq_len_step = 5.0, 100, 256
    t.t_train_start = t.t_step_start = time.perf_counter()
    t.token_type_counts_total = {"text": 80, "dsu": 20}
    return t, param


def test_optimizer_step_counters_loss_log_and_scaling():
    t, param = _trainer_for_optimizer_step()
    param.grad = torch.ones(4, 4) * 100
    t._loss_log = []
    t._optimizer_step(epoch=0, iter_idx=1)
    assert torch.allclose(param.grad, torch.ones(4, 4))  # grads / num_tokens_step (trainer.py:404)
    assert (t.global_step, t.consumed_samples, t.tokens_train_total) == (1, 2 * 2 * 1, 100)
    t.optimizer.step.assert_called_once()
    t.optimizer.zero_grad.assert_called_once_with(set_to_none=True)
    t.lr_scheduler.step.assert_called_once()
    assert t._loss_log == [pytest.approx(0.05)]  # 5.0 / 100
    assert (t.loss_running, t.num_tokens_step, t.max_seq_len_step) == (0.0, 0, 0)
    logged = t.wandb_logger.log_dict.call_args
    assert logged.kwargs["step"] == 1 and logged.args[0]["loss"] == pytest.approx(0.05) and logged.args[0]["lr"] == 2e-4
    assert {"duration_step", "tokens_per_second_per_gpu", "tokens_total", "train_clock_time", "max_seq_len_step",
            "n_tokens.text", "n_tokens.dsu"} <= set(logged.args[0])


def test_optimizer_step_skips_zero_token_window():
    t, _ = _trainer_for_optimizer_step()
    t.num_tokens_step = 0
    t._optimizer_step(epoch=0, iter_idx=1)
    t.optimizer.step.assert_not_called()
    t.optimizer.zero_grad.assert_called_once_with(set_to_none=True)
    assert t.global_step == 0


def test_clip_grad_norm_is_applied_and_logged():
    t, param = _trainer_for_optimizer_step(clip=1.0)
    param.grad = torch.ones(4, 4) * 100 * 3  # after /100 -> norm 12
    t._optimizer_step(epoch=0, iter_idx=1)
    assert float(param.grad.norm()) == pytest.approx(1.0, rel=1e-4)
    assert t.wandb_logger.log_dict.call_args.args[0]["grad_norm"] == pytest.approx(12.0, rel=1e-5)


def test_checkpoint_cadence_and_payload():
    t, _ = _trainer_for_optimizer_step()
    t.cfg.save_steps = 2
    t.model.state_dict = MagicMock(return_value={"w": torch.zeros(1)})
    for step, expect in ((0, 0), (1, 0), (2, 1), (3, 1), (4, 2)):
        t.global_step = step
        t._maybe_save_checkpoint()
        assert t.checkpointer.save_model_checkpoint.call_count == expect
    kw = t.checkpointer.save_training_state.call_args.kwargs
    assert kw["global_step"] == 4 and kw["seed"] == 42_831
    assert kw["training_hparams"] == {"batch_size": 2, "gradient_accumulation_steps": 2, "world_size": 1, "steps_per_epoch": 10}
    assert set(kw["cumulative_metrics"]) == {"tokens_train_total", "token_type_counts", "wall_clock_seconds"}


def test_resume_state_parsing_and_hparam_validation():
    from ssi.train_utils import resume_training_state, validate_resume_hparams
    ck = {"checkpoint_version": 1, "seed": 42_831, "global_step": 7, "optimizer": {}, "lr_scheduler": {}, "rng_state": {},
          "training_hparams": {"batch_size": 2}, "consumed_samples": 56, "cumulative_metrics": {}}
    st = resume_training_state(ck)
    assert st["global_step"] == 7 and st["consumed_samples"] == 56
    for bad in ({"checkpoint_version": 2}, {"seed": 1}):
        with pytest.raises(ValueError):
            resume_training_state({**ck, **bad})
    with pytest.raises(ValueError):
        resume_training_state({k: v for k, v in ck.items() if k != "checkpoint_version"})
    cur = {"batch_size": 2, "gradient_accumulation_steps": 4, "world_size": 1, "steps_per_epoch": 10}
    validate_resume_hparams(cur, cur)
    with pytest.raises(ValueError):
        validate_resume_hparams({**cur, "world_size": 8}, cur)
    validate_resume_hparams({**cur, "world_size": 8}, cur, force_resume=True)


def test_lr_schedule_matches_formula_and_resume_trick():
    from ssi.lr_schedule import setup_lr_scheduler
    cfg = OmegaConf.create({"lr_scheduler": {"num_warmup_steps": 10, "num_cycles": 0.5}})
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = setup_lr_scheduler(cfg, opt, global_step=-1, num_training_steps=110)
    lrs = []
    for _ in range(30):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    ref = [n / 10 if n < 10 else max(0.0, 0.5 * (1 + math.cos(math.pi * (n - 10) / 100))) for n in range(30)]
    assert lrs == pytest.approx(ref)
    opt2 = torch.optim.SGD([p], lr=1.0)
    sch2 = setup_lr_scheduler(cfg, opt2, global_step=14, num_training_steps=110)  # resumed at global_step 15
    assert opt2.param_groups[0]["lr"] == pytest.approx(ref[15])
    assert setup_lr_scheduler(OmegaConf.create({"lr_scheduler": None}), opt, -1, 10) is None


# ---- data -----------------------------------------------------------------------------------------------------------
def test_synthetic_sequences_follow_the_vocab_layout():
    from ssi.data import SyntheticDSUDataset, padded_collate_sft, synthetic_batch
    ds = SyntheticDSUDataset(4, 2048, n_dsus=5000)
    a, b = ds[1], ds[1]
    assert np.array_equal(a["tokens"], b["tokens"]) and len(a["tokens"]) == 2048
    tok, lab = a["tokens"], a["labels"]
    assert tok.min() >= 0 and tok.max() < 133_258
    dsu = (tok >= 128_000) & (tok < 133_000)
    assert 0.6 < dsu.mean() < 0.95
    runs = dsu[1:] & dsu[:-1]
    assert not np.any(tok[1:][runs] == tok[:-1][runs])  # deduplicated units
    assert (lab[:25] == -100).all() and (lab[25:] == tok[25:]).all()
    ds.set_epoch(1)
    assert not np.array_equal(ds[1]["tokens"], a["tokens"])
    ragged = SyntheticDSUDataset(4, 512, fixed_len=False)
    batch = padded_collate_sft([ragged[0], ragged[1]], padding_idx=ragged.pad_id)
    assert batch["tokens"].dtype == torch.int64 and batch["tokens"].shape == batch["labels"].shape
    n0 = len(ragged[0]["tokens"])
    if n0 < batch["tokens"].shape[1]:
        assert (batch["tokens"][0, n0:] == ragged.pad_id).all() and (batch["labels"][0, n0:] == -100).all()
    bb = synthetic_batch(2, 128, 5000)
    assert bb["tokens"].shape == (2, 128)


def test_compute_loss_does_not_mutate_batch_and_routes_generic():
    from ssi.loss import compute_loss
    labels = torch.tensor([[1, 2, 3, -100]])
    batch = {"tokens": torch.tensor([[5, 6, 7, 8]]), "labels": labels.clone()}
    model = MagicMock(spec=[])
    model.side_effect = lambda **kw: torch.zeros(1, 4, 10)
    seen = {}

    class LossFn:
        ignore_index = -100

        def __call__(self, logits, lab):
            seen["logits"], seen["labels"] = logits, lab
            return torch.tensor(1.5)

    out = compute_loss(batch, model, LossFn())
    assert float(out) == 1.5 and torch.equal(batch["labels"], labels)
    assert seen["labels"].tolist() == [2, 3, -100, -100] and seen["logits"].shape == (4, 10)


@pytest.mark.parametrize("global_step,steps_per_epoch,ga,want", [
    (150, 500, 4, (0, 600)),      # mid-epoch (reference tests/test_checkpoint.py:203-210)
    (500, 500, 4, (1, 0)),        # exact epoch boundary (:213-220)
    (0, 500, 4, (0, 0)),          # fresh start (:223-230)
    (1249, 500, 2, (2, 498)),
])
def test_resume_position_arithmetic(global_step, steps_per_epoch, ga, want):
    from ssi.trainer import resume_position
    assert resume_position(global_step, steps_per_epoch, ga) == want


@pytest.mark.parametrize("global_step", [0, 1, 249, 499, 500, 501, 999])
def test_resume_position_skips_less_than_an_epoch(global_step):
    """reference tests/test_checkpoint.py:238-243"""
    from ssi.trainer import resume_position
    _, skip = resume_position(global_step, 500, 4)
    assert 0 <= skip < 500 * 4 and skip % 4 == 0


def test_host_threads_are_kept_within_the_cpu_share(monkeypatch):
    """``limit_host_threads`` (``Trainer.setup()``): torch's intra-op pool is sized by the host's cores; under a cgroup quota its spinning threads
    get the whole process frozen (round 5: a third of the GPU's time idle in the trainer's loop at 2 x 2048).  Lowered to a few threads unless
    ``OMP_NUM_THREADS`` says otherwise; never raised; shared between the ranks of a node."""
    import torch
    from ssi import train_utils
    before = torch.get_num_threads()
    try:
        monkeypatch.setenv("OMP_NUM_THREADS", "7")
        torch.set_num_threads(6)
        assert train_utils.limit_host_threads() == 6                      # explicit setting: left alone
        monkeypatch.delenv("OMP_NUM_THREADS")
        monkeypatch.setattr(train_utils, "usable_cpus", lambda: 16)
        assert train_utils.limit_host_threads() == 4                      # a quarter of a 16-CPU share
        torch.set_num_threads(2)
        assert train_utils.limit_host_threads() == 2                      # never raised
        torch.set_num_threads(6)
        assert train_utils.limit_host_threads(world_size=8) == 1          # 8 ranks share the node's 16 CPUs
        monkeypatch.setattr(train_utils, "usable_cpus", lambda: 256)
        torch.set_num_threads(6)
        assert train_utils.limit_host_threads() == 6 and train_utils.usable_cpus() == 256
    finally:
        torch.set_num_threads(before)
