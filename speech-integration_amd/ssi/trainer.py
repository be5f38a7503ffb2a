"""Stateful trainer for the speech-integration hot path on MI355X.

Mirrors the public surface of ``/root/reference/ssi/trainer.py`` (``Trainer``, ``TrainingGeometry``, the method names and
the attributes its tests assert) and reproduces its step algebra exactly (``:385-424``; SURVEY.md
Appendix A.5): each micro-batch's mean loss over SHIFTED valid labels is multiplied by the UNSHIFTED valid-label count
before ``backward``; at the accumulation boundary gradients are divided by the sum of unshifted counts.

What differs underneath (MI355X-first):
* one fused HIP kernel counts token types and valid labels (K14); counts, loss and the kernels' error counts of a micro-batch stay on the
  device, and the whole accumulation window is fetched with ONE device->host copy at its boundary (the reference blocks 7-8 times per
  micro-batch, before and after the forward): between micro-batches the host never waits for the GPU;
* ``scale_grads`` and the clip coefficient are folded into the single-kernel AdamW step; nothing zeroes the gradient buffer (the first
  backward of the next window overwrites it);
* data parallelism (absent from the reference): per-layer gradient buckets all-reduced over RCCL/xGMI on a side stream
  during the last micro-batch's backward; token counts and running loss reduced with one small collective; every rank
  divides by the GLOBAL token count.
"""

from __future__ import annotations

import copy
import itertools
import logging
import math
import os
import random
import time
from collections import defaultdict
from dataclasses import dataclass
from typing import Any

import numpy as np
import torch
from torch import Tensor

from . import __version__
from .constants import DEBUGGING_TAG, MODEL_KEY, SEED
from .distributed import GradSync, all_reduce_scalars, get_world_size_and_rank, init_distributed
from .data.unpad import loss_inputs
from .eval import batch_to_device, compute_dataset_loss
from .llama_configs import configllama3_2_1b
from .loss import CEWithChunkedOutputLoss, compute_loss
from .lr_schedule import get_lr, setup_lr_scheduler
from .metric_logging import WandBLoggerPatched as WandBLogger
from .model import get_device, get_dtype, setup_llama3_2_1b
from .optimizer import clip_grad_norm_, scale_grads, setup_optimizer
from .train_utils import (count_token_types, count_token_types_async, get_token_type_ranges, limit_host_threads, resume_training_state,
                          validate_resume_hparams, validate_train_cfg)

__all__ = ["Trainer", "TrainingGeometry", "resume_position"]

LOGGER = logging.getLogger(__name__)


def _to_yaml(cfg) -> str:
    try:
        from .config import DictConfig, OmegaConf
        if isinstance(cfg, DictConfig):
            return OmegaConf.to_yaml(cfg, resolve=True, sort_keys=False)
    except Exception:
        pass
    try:
        from omegaconf import OmegaConf as _OC  # type: ignore
        return _OC.to_yaml(cfg, resolve=True, sort_keys=False)
    except Exception:
        return str(cfg)


def set_seed(seed: int, debug_mode: Any = None) -> None:
    """torchtune ``training.set_seed``: seed python / numpy / torch (+ per-rank offset is NOT applied: the reference seeds
    every process identically and shards data with ``DistributedSampler``)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if debug_mode is not None:
        mode = {"default": 0, "warn": 1, "error": 2}.get(debug_mode, debug_mode)
        torch.set_deterministic_debug_mode(int(mode))


@dataclass(frozen=True)
class TrainingGeometry:
    """Constants derived from dataset size and gradient accumulation (``trainer.py:64-125``)."""

    batch_size: int
    batches_per_epoch: int
    steps_per_epoch: int
    usable_batches: int
    n_epochs: int
    gradient_accumulation_steps: int
    world_size: int

    @classmethod
    def from_config(cls, cfg, dataloader, world_size: int) -> "TrainingGeometry":
        """An epoch is the whole number of accumulation windows the loader's batches fill; batches beyond the last full window are
        dropped at the epoch boundary (warned about once, here); fewer batches than one window is an error."""
        ga = cfg.gradient_accumulation_steps
        n_batches = len(dataloader)
        steps, leftover = divmod(n_batches, ga)
        if steps == 0:
            raise ValueError(f"batches_per_epoch ({n_batches}) < gradient_accumulation_steps ({ga}): an epoch would hold no optimizer step")
        if leftover:
            LOGGER.warning(f"{leftover} of the {n_batches} batches of an epoch do not fill an accumulation window of {ga} and are skipped "
                           "at every epoch boundary")
        return cls(batch_size=cfg.data.train.dataloader.batch_size, batches_per_epoch=n_batches, steps_per_epoch=steps,
                   usable_batches=steps * ga, n_epochs=-(-cfg.max_steps // steps), gradient_accumulation_steps=ga, world_size=world_size)


def resume_position(global_step: int, steps_per_epoch: int, gradient_accumulation_steps: int) -> tuple[int, int]:
    """Where a run that has completed ``global_step`` optimizer steps re-enters the data: ``(epochs already run, micro-batches of
    the current epoch to skip)`` — the arithmetic the reference does inline in ``train()`` (``trainer.py:330-336``) and pins in
    ``tests/test_checkpoint.py:203-240``."""
    epochs_run, steps_into_epoch = divmod(int(global_step), int(steps_per_epoch))
    return epochs_run, steps_into_epoch * int(gradient_accumulation_steps)


class Trainer:
    """Usage (as the reference): ``t = Trainer(cfg); t.setup(); t.train(); t.cleanup()``."""

    # Attribute surface (``tests/test_trainer.py:32-74`` of the reference asserts these names): what ``setup()`` fills in starts as None,
    # counters start at zero.  Kept as tables so that adding state means adding a row, not another assignment block.
    _FILLED_BY_SETUP = ("model", "tokenizer", "optimizer", "lr_scheduler", "loss_fn", "checkpointer", "wandb_logger",
                        "data_train", "sampler_train", "data_dev", "token_type_ranges", "geometry", "device", "dtype", "world_size",
                        "grad_sync", "_grad_norm", "_loss_log", "_resume_state", "_resume_rng_state")
    _COUNTERS = {"rank": 0, "global_step": 0, "consumed_samples": 0, "tokens_train_total": 0, "wall_clock_offset": 0.0,
                 "loss_running": 0.0, "num_tokens_step": 0, "max_seq_len_step": 0, "t_train_start": 0.0, "t_step_start": 0.0,
                 "unpadded_micro_batches": 0, "fused_micro_batches": 0}

    def __init__(self, cfg) -> None:
        self.cfg = cfg
        self._pending_readbacks: list[tuple[Tensor, int]] = []  # (device row of a micro-batch's counts / loss / error counts, tokens in it)
        self._window_valid_dev: Tensor | None = None           # valid labels of the open window's micro-batches, summed on the device
        self._window_valid_host: int | None = 0                # the same count from the host side of the batches (None: a batch came without one)
        self._lagged: list[dict[str, Any]] = []                # windows whose read-back is on its way (``_optimizer_step_lagged``)
        self._lag_buffers: list[Tensor] = []                   # page-locked buffers of finished read-backs, reused
        self._t_last_arrival = 0.0                             # when the previous window's results were seen on the host
        for name in self._FILLED_BY_SETUP:
            setattr(self, name, None)
        for name, zero in self._COUNTERS.items():
            setattr(self, name, zero)
        self.token_type_counts_total = defaultdict(int)                          # since step 0 (restored on resume), global under DP
        self._type_counts_window: defaultdict[str, int] = defaultdict(int)      # this rank's counts since the last optimizer step
        self._bad_inputs_window: defaultdict[str, int] = defaultdict(int)       # ids / positions the kernels had to refuse, same window

    # The window's accumulators are host numbers (the reference's attribute names), but on the GPU path a micro-batch leaves its counts and
    # loss ON THE DEVICE: nothing in the step loop needs them before the accumulation boundary, so the host never waits for the GPU between
    # micro-batches (the reference blocks 7-8 times per micro-batch; rounds 1-2 of this trainer once).  Reading any of the three attributes
    # below fetches what is pending first — ONE device-to-host copy per window.
    def _read_back_window(self) -> None:
        self._flush_lagged()  # earlier windows first: counters and log lines stay in order
        if not self._pending_readbacks:
            return
        pending, self._pending_readbacks = self._pending_readbacks, []
        rows = torch.stack([row for row, _ in pending]).tolist()  # the window's one host sync
        kinds = list(self.token_type_ranges)
        for host, (_, n_positions) in zip(rows, pending):
            counts = {tt: int(c) for tt, c in zip(kinds + ["total"], host)}
            for tt, c in counts.items():
                self._token_type_counts_total[tt] += c
                self._type_counts_window[tt] += c
            self._num_tokens_step += int(host[len(kinds) + 1])
            self._loss_running += float(host[-3])
            # ids outside the vocabulary: torch's embedding / cross_entropy would device-assert (the HIP kernels write zeros and count);
            # the token-type ranges partition [0, V), so a token outside them shows up as a short sum; positions beyond the RoPE table are
            # clamped by the kernel and counted
            self._bad_inputs_window["labels outside [0, vocab_size)"] += int(host[-2])
            self._bad_inputs_window["token ids outside [0, vocab_size)"] += n_positions - sum(counts[tt] for tt in kinds)
            self._bad_inputs_window["input_pos entries outside the RoPE table"] += int(host[-1])
        if self.grad_sync is None:  # alone: fail here; data parallel: at the window's scalar all-reduce, on EVERY rank (a rank that raised
            self._raise_on_bad_inputs(self._bad_inputs_window)  # alone would leave the others blocked in their collectives)

    @property
    def num_tokens_step(self) -> int:
        self._read_back_window()
        return self._num_tokens_step

    @num_tokens_step.setter
    def num_tokens_step(self, value: int) -> None:
        self._num_tokens_step = value

    @property
    def loss_running(self) -> float:
        self._read_back_window()
        return self._loss_running

    @loss_running.setter
    def loss_running(self, value: float) -> None:
        self._loss_running = value

    @property
    def token_type_counts_total(self):
        self._read_back_window()
        return self._token_type_counts_total

    @token_type_counts_total.setter
    def token_type_counts_total(self, value) -> None:
        self._token_type_counts_total = value

    # === Setup ===========================================================================================================
    def setup(self) -> None:
        validate_train_cfg(self.cfg)
        set_seed(seed=SEED, debug_mode=self.cfg.get("debug_mode"))
        self.device = get_device(self.cfg.device)
        self.dtype = get_dtype(self.cfg.dtype)
        self.world_size, self.rank = init_distributed(self.device)
        limit_host_threads(self.world_size)  # (not in the reference: torch's CPU thread pool within this process's CPU share)
        self._setup_logging()
        self._setup_model()
        self._setup_tokenizer()
        self._extract_resume_state()
        self._setup_optimizer()
        self._setup_loss()
        self._setup_data()
        self.geometry = TrainingGeometry.from_config(self.cfg, self.data_train, self.world_size)
        self._finalize_resume()
        self._setup_data_parallel()
        self._ckpt_dict = None
        self._resume_rng_state = self._resume_state.pop("rng_state", None) if self._resume_state else None
        self._resume_state = None

    def _setup_logging(self) -> None:
        tags = [__version__, self.cfg.config_name]
        if os.getenv("SLURM_JOB_QOS") == "gpu-debug":
            tags += [DEBUGGING_TAG]
        wandb_cfg = {k: self.cfg.wandb[k] for k in self.cfg.wandb} if self.cfg.get("wandb") is not None else {}
        self.wandb_logger = WandBLogger(**wandb_cfg, tags=tags)
        if self.cfg.checkpointer.get("output_dir") is None:
            from .checkpoint import resolve_checkpointer_output_dir
            self.cfg.checkpointer.output_dir = resolve_checkpointer_output_dir(self.cfg, self.wandb_logger)
            LOGGER.info(f"No checkpointer output dir provided. Resolved to: {self.cfg.checkpointer.output_dir!s}")

    def _setup_model(self) -> None:
        from .checkpoint import make_checkpointer
        self._llama_config = copy.deepcopy(configllama3_2_1b)
        self._llama_config.update_from_speech_cfg(self.cfg.speech)
        overrides = self.cfg.get("model_overrides")  # test/bench hook: shrink the architecture, never used by conf/
        if overrides is not None:
            for k in overrides:
                setattr(self._llama_config, k, overrides[k])
        ck = {k: self.cfg.checkpointer[k] for k in self.cfg.checkpointer}
        self.checkpointer = make_checkpointer(**ck, model_expectations=self._llama_config.checkpoint_expectations)
        self._ckpt_dict = self.checkpointer.load_checkpoint()
        self.model = setup_llama3_2_1b(cfg=self.cfg, llama_config=self._llama_config,
                                       model_state_dict=self._ckpt_dict.get(MODEL_KEY), dtype_default=self.dtype,
                                       device_default=self.device)
        if self._ckpt_dict.get(MODEL_KEY) is None:
            from .checkpoint import random_init_
            random_init_(self.model, seed=SEED)
        self.model.to(device=self.device)
        self.model.train()

    def _setup_tokenizer(self) -> None:
        from .tokenizer import setup_llama3_tokenizer
        tk = {k: self.cfg.tokenizer[k] for k in self.cfg.tokenizer}
        self.tokenizer, _special = setup_llama3_tokenizer(**tk, llama_config=self._llama_config)
        self.token_type_ranges = get_token_type_ranges(llama_config=self._llama_config)

    def _setup_data(self) -> None:
        """``trainer.py:261-271``: SFT or CPT datasets by ``config_name``; ``dataset.source: synthetic`` (not in the reference)
        selects the MLS-shaped generator that needs neither a tokenizer file nor a dataset."""
        from .data import setup_sft_data, setup_synthetic_data, setup_text_completion_data
        for split in ("train", "dev"):
            node = self.cfg.data[split]
            if node.dataset.get("source") == "synthetic":
                loader, sampler = setup_synthetic_data(
                    n_samples=int(node.dataset.get("n_samples") or 1024), seq_len=int(self.cfg.tokenizer.max_seq_len),
                    batch_size=int(node.dataloader.batch_size), n_dsus=int(self.cfg.speech.n_dsus), world_size=self.world_size,
                    rank=self.rank, shuffle=bool(node.get("shuffle", False)), drop_last=bool(node.dataloader.get("drop_last", False)),
                    fixed_len=bool(node.dataset.get("fixed_len", True)), kind=str(self.cfg.config_name))
            elif self.cfg.config_name == "sft":
                loader, sampler = setup_sft_data(cfg_dataset=node, model_tokenizer=self.tokenizer)
            elif self.cfg.config_name == "cpt":
                loader, sampler = setup_text_completion_data(node, self.tokenizer)
            else:
                raise NotImplementedError(f"Unsupported config_name: {self.cfg.config_name}")
            if split == "train":
                self.data_train, self.sampler_train = loader, sampler
            else:
                self.data_dev = loader

    def _extract_resume_state(self) -> None:
        self._resume_state = None
        if getattr(self.checkpointer, "training_state_checkpoint", None) is not None:
            self._resume_state = resume_training_state(self._ckpt_dict)
            self.global_step = self._resume_state["global_step"]
            self.consumed_samples = self._resume_state["consumed_samples"]

    def _setup_optimizer(self) -> None:
        self.optimizer = setup_optimizer(self.cfg, self.model, self._resume_state["optimizer_state"] if self._resume_state else None)
        self.lr_scheduler = setup_lr_scheduler(cfg=self.cfg, optimizer=self.optimizer, global_step=self.global_step - 1,
                                               num_training_steps=self.cfg.max_steps)
        if self._resume_state and self.lr_scheduler is not None:
            self.lr_scheduler.load_state_dict(self._resume_state["lr_scheduler_state"])

    def _setup_loss(self) -> None:
        self.loss_fn = CEWithChunkedOutputLoss()
        if isinstance(self.loss_fn, CEWithChunkedOutputLoss):
            self.model.set_num_output_chunks(self.loss_fn.num_output_chunks)

    def _data_position_hparams(self) -> dict[str, int]:
        """The values that tie ``global_step`` to a position in the data; written into every training state and compared on resume."""
        return {"batch_size": self.geometry.batch_size, "gradient_accumulation_steps": self.cfg.gradient_accumulation_steps,
                "world_size": self.world_size, "steps_per_epoch": self.geometry.steps_per_epoch}

    def _finalize_resume(self) -> None:
        """Second half of a resume, once the data geometry is known: cumulative counters back in place, data-position check."""
        state = self._resume_state
        if state is None:
            return
        totals = state["cumulative_metrics"]
        self.tokens_train_total = totals["tokens_train_total"]
        self.token_type_counts_total.update(totals["token_type_counts"])
        self.wall_clock_offset = totals["wall_clock_seconds"]
        validate_resume_hparams(ckpt_hparams=state["training_hparams"], current_hparams=self._data_position_hparams(),
                                force_resume=self.cfg.get("force_resume", False))

    def _setup_data_parallel(self) -> None:
        """One process per GPU; gradients exchanged per bucket during backward (``ssi.distributed``)."""
        from .distributed import _single_rank_exercise
        if self.model is None or not self.world_size or (self.world_size <= 1 and not _single_rank_exercise()):
            return
        if hasattr(self.model, "_flat_grad"):
            self.grad_sync = GradSync(self.model._flat_grad, self.model.buckets)
            self.model.grad_sync = self.grad_sync
        else:
            self.grad_sync = GradSync.for_module(self.model)

    # === Training ========================================================================================================
    def train(self) -> None:
        """Run until ``cfg.max_steps`` optimizer steps have been taken, starting where ``global_step`` says (0, or a resume)."""
        self.optimizer.zero_grad()
        self.t_train_start = self.t_step_start = self._t_last_arrival = time.perf_counter()
        self._reset_step_accumulators()
        first_epoch, skip = resume_position(self.global_step, self.geometry.steps_per_epoch, self.cfg.gradient_accumulation_steps)
        if self._resume_rng_state is not None:
            from .checkpoint import restore_rng_states
            restore_rng_states(self._resume_rng_state)
            self._resume_rng_state = None
            LOGGER.info("python / numpy / torch generator states restored from the training state")
        LOGGER.info(_to_yaml(self.cfg))
        self.wandb_logger.log_config(self.cfg)
        for epoch in range(first_epoch, self.geometry.n_epochs):
            self._train_epoch(epoch, skip if epoch == first_epoch else 0)
            if self.global_step >= self.cfg.max_steps:
                LOGGER.info(f"max_steps={self.cfg.max_steps} reached")
                break
        self._flush_lagged()  # (not in a finally: an exception on its way out should not be followed by a wait on the device)

    def _epoch_batches(self, epoch: int, batches_to_skip: int):
        """``(index, batch)`` pairs of one epoch: the first ``usable_batches`` of the loader (whole accumulation windows only), minus
        the ones a resumed run has already consumed.  Sampler and dataset are told the epoch (shuffling and the per-sample generators
        of the CPT data key on it); batches are collated, stripped of their padding, pinned and copied to the device ahead of the step by a
        background thread.  With ``gradient_accumulation_steps > 1`` the micro-batches of a window arrive joined into one batch where they can be
        (``ssi.data.window``): the index is then that of the window's LAST micro-batch."""
        for obj in (self.sampler_train, getattr(self.data_train, "dataset", None)):
            if obj is not None and hasattr(obj, "set_epoch"):
                obj.set_epoch(epoch)
        if batches_to_skip:
            LOGGER.info(f"resume: epoch {epoch} starts at batch {batches_to_skip}")
        indexed = itertools.islice(enumerate(self.data_train), batches_to_skip, self.geometry.usable_batches)
        transform = self._host_batch_transform()
        window = int(self.cfg.gradient_accumulation_steps)
        if transform is not None and window > 1 and self.cfg.get("fuse_accumulation_window", True):
            from .data.window import fused_windows
            kw = transform.keywords
            indexed = fused_windows(indexed, window, max_tokens=int(self.cfg.get("fused_window_max_tokens", 32768)), single=transform,
                                    pad_id=kw["pad_id"], ignore_index=kw["ignore_index"], multiple=kw["multiple"], plan_fn=kw["plan_fn"],
                                    padded_len=kw["padded_len"])
        elif transform is not None:
            indexed = ((i, transform(b)) for i, b in indexed)
        if self.device.type == "cuda" and self.grad_sync is None and self.cfg.get("lagged_readback", True):
            indexed = self._with_host_label_counts(indexed)
        depth = int(self.cfg.get("prefetch_batches", 2) or 0)
        if self.device.type != "cuda" or depth <= 0:
            return indexed
        from .data.prefetch import DevicePrefetcher

        def tagged():  # the prefetcher moves dictionaries; the index rides along as a plain value
            for i, b in indexed:
                b = dict(b)
                b["_index"] = i
                yield b

        return ((b.pop("_index"), b) for b in DevicePrefetcher(tagged(), self.device, depth=depth))

    def _with_host_label_counts(self, indexed):
        """``n_valid_host`` beside every batch whose labels are still on the host (they are, in the prefetch thread): the count of non-ignored
        labels — all the accumulation boundary needs to know on the HOST (is the window empty?  what divides the gradients?) — without
        waiting for the device (``_optimizer_step_lagged``)."""
        ignore = self.loss_fn.ignore_index
        for i, b in indexed:
            labels = b.get("labels") if isinstance(b, dict) else None
            if torch.is_tensor(labels) and not labels.is_cuda:
                b = dict(b)
                b["n_valid_host"] = int((labels != ignore).sum())
            yield i, b

    def _host_batch_transform(self):
        """Right-padded batches lose their padding on the host, in the prefetch thread (``ssi.data.unpad``: exact, and the step's time then follows
        the real tokens, not the padded rows).  ``padding_free: false`` in the config keeps the padded form; a
        model without the packed path (stand-ins of the CPU tests) and callers that hand ``_train_step`` device tensors are not affected."""
        if not self.cfg.get("padding_free", True) or not hasattr(self.model, "fused_loss") or not hasattr(self.model, "padded_seq_len"):
            return None
        from functools import partial

        from .data.unpad import unpad_batch
        tiles = getattr(self.model, "_mfma_shapes", lambda: False)()  # the MFMA kernels walk whole 256-row tiles; the generic ones take any length
        return partial(unpad_batch, pad_id=self.tokenizer.pad_id, ignore_index=self.loss_fn.ignore_index, padded_len=self.model.padded_seq_len,
                       multiple=256 if tiles else 1, plan_fn=getattr(self.model, "build_attn_plan", None))

    def _train_epoch(self, epoch: int, batches_to_skip: int = 0) -> None:
        window = self.cfg.gradient_accumulation_steps
        for i, batch in self._epoch_batches(epoch, batches_to_skip):
            closes_window = (i + 1) % window == 0
            self._train_step(batch, sync_gradients=closes_window)  # gradients are exchanged only by the window's last backward
            del batch
            if closes_window:
                self._optimizer_step(epoch, i)
                if self.global_step >= self.cfg.max_steps:
                    return

    def _train_step(self, batch: dict[str, Tensor], sync_gradients: bool = True) -> None:
        """Single micro-batch forward + backward (``trainer.py:385-395``).  On the GPU nothing is read back here: the counts, the loss and the
        kernels' error counts of the micro-batch stay on the device until the window closes (``_read_back_window``)."""
        batch_to_device(batch, self.device)
        tokens, labels = batch["tokens"], batch["labels"]
        ignore = self.loss_fn.ignore_index
        self.max_seq_len_step = max(self.max_seq_len_step, int(batch.get("max_seq_len", tokens.size(1))))
        on_gpu = tokens.is_cuda
        if on_gpu:  # K14: ranges + non-pad + valid labels in one launch, result stays on the device
            counts_dev = count_token_types_async(tokens, self.token_type_ranges, self.tokenizer.pad_id, labels, ignore)
            n_valid = counts_dev[-1]
        else:       # host tensors (stand-in models in CPU tests)
            counts_host = count_token_types(tokens, self.token_type_ranges, self.tokenizer.pad_id)
            n_valid = (labels != ignore).sum()
        if hasattr(self.model, "sync_this_backward"):
            self.model.sync_this_backward = bool(sync_gradients)
        if on_gpu:
            n_host = batch.get("n_valid_host")
            if self._window_valid_host is not None:
                self._window_valid_host = None if n_host is None else self._window_valid_host + int(n_host)
            self._arm_optimizer(n_valid, bool(sync_gradients))
        if batch.get("packed_input_pos") is not None:  # the prefetch thread dropped the padding (ssi/data/unpad.py, ssi/data/window.py)
            self.unpadded_micro_batches += int(batch.get("micro_batches", 1))
        self.fused_micro_batches += int(batch.get("micro_batches", 0))  # micro-batches that arrived joined into one batch (ssi/data/window.py)
        loss_batch = compute_loss(loss_inputs(batch), self.model, self.loss_fn) * n_valid  # mean over SHIFTED x UNSHIFTED count
        loss_batch.backward()
        if on_gpu:
            zero = torch.zeros(1, dtype=torch.float64, device=tokens.device)
            errs = [getattr(self.model, "label_errors", None), getattr(self.model, "position_errors", None)]
            errs = [zero if e is None else e.detach().to(torch.float64).reshape(1) for e in errs]
            # row = [count per token type ..., total (non-pad), valid labels, loss x valid labels, bad labels, bad positions]
            row = torch.cat((counts_dev.to(torch.float64), loss_batch.detach().to(torch.float64).reshape(1), *errs))
            self._pending_readbacks.append((row, tokens.numel()))
            return
        for tt, c in counts_host.items():
            self._token_type_counts_total[tt] += c
            self._type_counts_window[tt] += c
        self._num_tokens_step += int(n_valid.item())
        self._loss_running += float(loss_batch.item())

    def _arm_optimizer(self, n_valid: Tensor, last_of_window: bool) -> None:
        """Round 5: AdamW under the window's last backward (``HipAdamW.overlap_with_backward``).  The window's token count — the divisor of
        ``scale_grads`` (``trainer.py:404``) — is known on the DEVICE before that backward starts (the sum of the micro-batches' valid-label
        counts), so every bucket of gradients can be applied the moment it is final, on a side stream, while the backward runs on: the same
        kernel with the same factor and step number, i.e. the same parameters bit for bit, 1.2 ms of the headline's 105.8.  Only where nothing
        needs all gradients first: one GPU (``overlap_with_backward`` declines under data parallelism), no clipping, ``adamw_under_backward`` not
        switched off.  A window without a label leaves a non-finite factor: the kernel then changes nothing and ``_optimizer_step`` skips as
        the reference does."""
        self._window_valid_dev = n_valid if self._window_valid_dev is None else self._window_valid_dev + n_valid
        if not last_of_window:
            return
        total, self._window_valid_dev = self._window_valid_dev, None
        if (self.cfg.get("adamw_under_backward", True) and self.cfg.clip_grad_norm is None and self.world_size == 1
                and hasattr(self.optimizer, "overlap_with_backward")):
            self.optimizer.overlap_with_backward(1.0 / total.to(torch.float32))

    @staticmethod
    def _raise_on_bad_inputs(bad: dict[str, int], anywhere: int | None = None) -> None:
        mine = {what: n for what, n in bad.items() if n}
        if mine or anywhere:
            here = "; ".join(f"{n} {what}" for what, n in mine.items()) or "none on this rank"
            raise IndexError(f"invalid ids in this accumulation window: {here}" + (f" ({anywhere} over all ranks)" if anywhere else ""))

    def _optimizer_step(self, epoch: int, iter_idx: int) -> None:
        """Accumulation boundary (``trainer.py:397-424``): [all-reduce] -> scale -> clip -> AdamW -> LR -> counters."""
        if (self._window_valid_host is not None and self._pending_readbacks and self.grad_sync is None
                and self.cfg.get("lagged_readback", True)):
            return self._optimizer_step_lagged(epoch, iter_idx)
        self._read_back_window()  # the window's one device-to-host copy: counts, losses and error counts of its micro-batches
        if self.grad_sync is not None:
            # one small collective: token count, running loss and the window's token-type counts (tokens_total is global, so the
            # per-type totals must be too: every rank adds what the OTHER ranks saw in this window)
            kinds = sorted(self._type_counts_window)
            summed = all_reduce_scalars([self.num_tokens_step, self.loss_running, float(sum(self._bad_inputs_window.values())),
                                         *(self._type_counts_window[k] for k in kinds)], self.device, group=self.grad_sync.scalar_group)
            self.num_tokens_step, self.loss_running = int(round(summed[0])), float(summed[1])
            for k, v in zip(kinds, summed[3:]):
                self.token_type_counts_total[k] += int(round(v)) - self._type_counts_window[k]
            self.grad_sync.finish(defer_last=self.cfg.clip_grad_norm is None)  # the embedding bucket lands under the AdamW of the rest
            if summed[2] > 0:  # every rank sees the same sum, so every rank raises (after its reductions have drained)
                self.grad_sync.finish_deferred()
                self._raise_on_bad_inputs(self._bad_inputs_window, anywhere=int(round(summed[2])))
        self._type_counts_window.clear()
        self._bad_inputs_window.clear()
        window_tokens = self.num_tokens_step
        if window_tokens > 0:
            self._apply_window(window_tokens)
            self._close_window(epoch, iter_idx, window_tokens)
        else:  # every label of the window ignored: nothing to learn from, nothing to count (reference: warn, drop the gradients, go on)
            LOGGER.warning("No non-ignored tokens in accumulation window; skipping optimizer step.")
            if hasattr(self.optimizer, "cancel_overlap"):
                self.optimizer.cancel_overlap()  # (updates issued under the backward were no-ops: the kernel's guard on the factor 1 / 0)
            self.optimizer.zero_grad(set_to_none=True)
        self._reset_step_accumulators()
        if window_tokens > 0:
            self._maybe_save_checkpoint()

    def _optimizer_step_lagged(self, epoch: int, iter_idx: int) -> None:
        """The boundary without waiting for the device (one GPU; batches that came through ``_epoch_batches``).  What the host must know to go
        on — is the window empty, what divides the gradients — is the count of valid labels, which the prefetch thread took from the host
        tensors (``n_valid_host``).  Everything else the window left on the device (token-type counts, the loss for the log line, the
        kernels' error counts) is copied back asynchronously and read one boundary later, so the host keeps launching: with a blocking
        read-back the GPU idles 1.6 ms per optimizer step while the host does the boundary's bookkeeping and starts the next forward
        (kernel trace at 8 x 2048, ``profiles/LAB_NOTES.md`` round 5).  The log line and the metric record of step k therefore appear when
        step k + 1 closes — with the values of step k — except on steps that evaluate, save or end the run, which read back at once.  The
        device's own count of valid labels is checked against the host's when it arrives."""
        window_tokens = int(self._window_valid_host)
        pending, self._pending_readbacks = self._pending_readbacks, []
        rows = torch.stack([row for row, _ in pending])
        host = next((b for b in self._lag_buffers if b.numel() >= rows.numel()), None)
        if host is None:
            host = torch.empty(max(rows.numel(), 256), dtype=rows.dtype, pin_memory=True)
        else:
            self._lag_buffers.remove(host)
        host[:rows.numel()].view(rows.shape).copy_(rows, non_blocking=True)
        arrived = torch.cuda.Event()
        arrived.record()
        entry: dict[str, Any] = {"arrived": arrived, "host": host, "shape": tuple(rows.shape), "positions": [n for _, n in pending],
                                 "window_tokens": window_tokens, "epoch": epoch, "iter_idx": iter_idx, "applied": window_tokens > 0}
        self._flush_lagged()  # the window before this one: its copy finished a whole step ago
        now_due = False
        if window_tokens > 0:
            self._apply_window(window_tokens)
            self.global_step += 1
            self.consumed_samples += self.geometry.batch_size * self.cfg.gradient_accumulation_steps * self.world_size
            self.tokens_train_total += window_tokens
            entry.update(global_step=self.global_step, lr=get_lr(self.optimizer), tokens_total=self.tokens_train_total,
                         max_seq_len_step=self.max_seq_len_step, grad_norm=self._grad_norm)  # (its clock is read when its copy arrives)
            now_due = (self.global_step % self.cfg.eval_steps == 0 or self.global_step % self.cfg.save_steps == 0
                       or self.global_step >= self.cfg.max_steps)
        else:
            LOGGER.warning("No non-ignored tokens in accumulation window; skipping optimizer step.")
            if hasattr(self.optimizer, "cancel_overlap"):
                self.optimizer.cancel_overlap()
            self.optimizer.zero_grad(set_to_none=True)
        self._lagged.append(entry)
        self._reset_step_accumulators()
        if now_due:
            self._flush_lagged()  # (evaluation happens inside the log call, on the weights of exactly this step)
        if window_tokens > 0:
            self._maybe_save_checkpoint()

    def _flush_lagged(self) -> None:
        """Finish the boundaries whose read-back was left in flight, oldest first: counts into the totals, the kernels' error counts raised,
        the device's label count checked against the host's, loss logged."""
        while self._lagged:
            e = self._lagged.pop(0)
            e["arrived"].synchronize()
            # the step's duration = from the arrival of the window before it to its own: the device's pace, not the host's (which runs ahead)
            e["now"] = time.perf_counter()
            e["step_seconds"], self._t_last_arrival = e["now"] - self._t_last_arrival, e["now"]
            n_rows, width = e["shape"]
            rows = e["host"][:n_rows * width].view(n_rows, width).tolist()
            self._lag_buffers.append(e["host"])
            kinds = list(self.token_type_ranges)
            bad: defaultdict[str, int] = defaultdict(int)
            loss_sum, n_valid_device = 0.0, 0
            for host_row, n_positions in zip(rows, e["positions"]):
                counts = {tt: int(c) for tt, c in zip(kinds + ["total"], host_row)}
                for tt, c in counts.items():
                    self._token_type_counts_total[tt] += c
                n_valid_device += int(host_row[len(kinds) + 1])
                loss_sum += float(host_row[-3])
                bad["labels outside [0, vocab_size)"] += int(host_row[-2])
                bad["token ids outside [0, vocab_size)"] += n_positions - sum(counts[tt] for tt in kinds)
                bad["input_pos entries outside the RoPE table"] += int(host_row[-1])
            self._raise_on_bad_inputs(bad)
            if n_valid_device != e["window_tokens"]:
                raise RuntimeError(f"the device counted {n_valid_device} valid labels in a window the host counted {e['window_tokens']} in")
            if not e["applied"]:
                continue
            mean_loss = loss_sum / e["window_tokens"]
            if self._loss_log is not None:
                self._loss_log.append(mean_loss)
            self._log_metrics(e["epoch"], e["iter_idx"], mean_loss, snapshot=e)

    def _apply_window(self, window_tokens: int) -> None:
        """Gradients of the window -> parameters: mean over the window's (global) unshifted token count, optional global-norm clip, AdamW,
        schedule.  On the HIP model the scale and the clip coefficient ride into the one-kernel AdamW (``ssi.optimizer``)."""
        scale_grads(self.model, torch.tensor(1 / window_tokens))
        max_norm = self.cfg.clip_grad_norm
        if max_norm is not None:
            self._grad_norm = clip_grad_norm_(self.model, max_norm=float(max_norm))
        self.optimizer.step()
        self.optimizer.zero_grad(set_to_none=True)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()

    def _close_window(self, epoch: int, iter_idx: int, window_tokens: int) -> None:
        """Counters and the step's log record (``trainer.py:413-421``): one optimizer step consumed ga x batch x world samples."""
        self.global_step += 1
        self.consumed_samples += self.geometry.batch_size * self.cfg.gradient_accumulation_steps * self.world_size
        self.tokens_train_total += window_tokens
        mean_loss = self.loss_running / window_tokens
        if self._loss_log is not None:
            self._loss_log.append(mean_loss)
        self._log_metrics(epoch, iter_idx, mean_loss)

    def _evaluate(self) -> float:
        return compute_dataset_loss(self.model, self.data_dev, self.loss_fn,
                                    epoch=self.global_step // self.geometry.steps_per_epoch, global_step=self.global_step,
                                    steps_per_epoch=self.geometry.steps_per_epoch, device=self.device,
                                    join_batches=int(self.cfg.get("eval_join_batches", 16) or 0) if self.cfg.get("padding_free", True) else 0,
                                    max_tokens=int(self.cfg.get("fused_window_max_tokens", 32768)), pad_id=int(getattr(self.tokenizer, "pad_id", 0) or 0),
                                    prefetch=int(self.cfg.get("prefetch_batches", 2) or 0))

    def _log_metrics(self, epoch: int, iter_idx: int, loss_to_log: float, snapshot: dict[str, Any] | None = None) -> None:
        """One console line per optimizer step; the metric record (same keys as the reference logs to W&B, ``trainer.py:440-475``) every
        ``log_interval`` steps from rank 0; the dev loss joins it on steps that evaluate.  ``snapshot``: the step's values as they were when
        its window closed (``_optimizer_step_lagged`` logs a step after the next one has been launched); without one, the current state."""
        if snapshot is None:
            now = self._t_last_arrival = time.perf_counter()
            snapshot = {"global_step": self.global_step, "window_tokens": self.num_tokens_step, "lr": get_lr(self.optimizer), "now": now,
                        "step_seconds": now - self.t_step_start, "tokens_total": self.tokens_train_total,
                        "max_seq_len_step": self.max_seq_len_step, "grad_norm": self._grad_norm}
        step, type_counts = snapshot["global_step"], self._token_type_counts_total
        width = len(str(self.geometry.batches_per_epoch))
        per_type = " | ".join(f"Tokens ({kind}): {n}" for kind, n in type_counts.items())
        LOGGER.info(f"Epoch {epoch + 1:03d} | Iteration {iter_idx:0{width}d} / {self.geometry.batches_per_epoch} | Global Step {step} | "
                    f"Loss: {loss_to_log:.4f} | Tokens (num_tokens_step): {snapshot['window_tokens']}" + (f" | {per_type}" if per_type else ""))
        dev_loss = self._evaluate() if step % self.cfg.eval_steps == 0 else None
        if step % self.cfg.log_interval:
            return
        ranks = self.world_size if (self.grad_sync is not None and self.world_size) else 1  # num_tokens_step is global under DP
        record: dict[str, Any] = {
            "loss": loss_to_log,
            "lr": snapshot["lr"],
            "duration_step": snapshot["step_seconds"],
            "tokens_per_second_per_gpu": snapshot["window_tokens"] / snapshot["step_seconds"] / ranks,
            "tokens_total": snapshot["tokens_total"],
            "train_clock_time": (self.wall_clock_offset + snapshot["now"] - self.t_train_start) / 3600.0,
            "max_seq_len_step": snapshot["max_seq_len_step"],
        }
        record.update({f"n_tokens.{kind}": n for kind, n in type_counts.items()})
        if self.cfg.clip_grad_norm is not None:
            record["grad_norm"] = None if snapshot["grad_norm"] is None else float(snapshot["grad_norm"])
        if dev_loss is not None:
            record["dev_loss"] = dev_loss
        if self.rank == 0:
            self.wandb_logger.log_dict(record, step=step)

    def _maybe_save_checkpoint(self) -> None:
        if self.global_step > 0 and self.global_step % self.cfg.save_steps == 0:
            self.save_checkpoint()
            LOGGER.info(f"checkpoint written at step {self.global_step}")

    def _reset_step_accumulators(self) -> None:
        self.loss_running, self.num_tokens_step, self.max_seq_len_step = 0.0, 0, 0
        self._window_valid_dev, self._window_valid_host = None, 0
        self.t_step_start = time.perf_counter()

    # === Checkpointing ===================================================================================================
    def save_checkpoint(self) -> None:
        """Model weights under ``step_N/`` plus ONE ``training_state.pt`` (schema v1, ``constants.py``) at the checkpoint root —
        rank 0 only: every rank holds the same weights and optimizer state."""
        self._flush_lagged()  # (the cumulative token-type counts go into the training state)
        if self.rank != 0:
            return
        self.checkpointer.save_model_checkpoint(self.model.state_dict(), self.global_step)
        elapsed = self.wall_clock_offset + (time.perf_counter() - self.t_train_start)
        self.checkpointer.save_training_state(
            optimizer_state_dict=self.optimizer.state_dict(),
            lr_scheduler_state_dict=None if self.lr_scheduler is None else self.lr_scheduler.state_dict(),
            global_step=self.global_step, seed=SEED, training_hparams=self._data_position_hparams(), consumed_samples=self.consumed_samples,
            cumulative_metrics={"tokens_train_total": self.tokens_train_total, "token_type_counts": dict(self.token_type_counts_total),
                                "wall_clock_seconds": elapsed})

    # === Cleanup =========================================================================================================
    def cleanup(self) -> None:
        if getattr(self, "wandb_logger", None) is not None:
            self.wandb_logger.close()
        from .distributed import shutdown_distributed
        shutdown_distributed()  # whatever init_distributed created, also the one-rank exercise's group (SSI_DP_SINGLE=1)
