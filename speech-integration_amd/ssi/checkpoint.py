"""Minimal checkpoint I/O for the hot path (torchtune-key safetensors + schema-v1 ``training_state.pt``).

The reference's ``FullModelHFCheckpointer`` (``/root/reference/ssi/checkpoint.py:209-468``: HF<->torchtune key conversion,
shard discovery, config.json validation) is disk I/O outside the step path and is NOT rebuilt this round (SURVEY.md §8f
rank 3).  What the trainer needs from a checkpointer is kept, with the same method names:
``load_checkpoint() -> {"model": state_dict | None, ...training-state keys}``,
``save_model_checkpoint(state_dict, step)``, ``save_training_state(...)`` (schema v1 keys, ``constants.py:78-89``),
``training_state_checkpoint``.  Weights are read/written as ONE safetensors file whose keys are the torchtune names the
model exposes (``tok_embeddings.weight``, ``layers.{i}.attn.q_proj.weight``, ...); with no weights on disk the model is
random-initialised (seeded) — there are no Llama weights on the build/GPU image and no network."""

from __future__ import annotations

import logging
import os
import random
from typing import Any

import numpy as np
import torch

from .constants import (CHECKPOINT_VERSION, CHECKPOINT_VERSION_KEY, CONSUMED_SAMPLES_KEY, CUMULATIVE_METRICS_KEY,
                        GLOBAL_STEP_KEY, LR_SCHEDULER_KEY, MODEL_KEY, OPTIMIZER_KEY, RNG_KEY, SEED_KEY, TRAINING_HPARAMS_KEY)

LOGGER = logging.getLogger(__name__)
MODEL_FILENAME = "model.safetensors"
TRAINING_STATE_FILENAME = "training_state.pt"


def resolve_checkpointer_output_dir(cfg, wandb_logger) -> str:
    base = cfg.get("output_dir") or os.path.join(os.getcwd(), "outputs")
    return os.path.join(str(base), f"{getattr(wandb_logger, 'run_name', 'local')}-id_{getattr(wandb_logger, 'run_id', '0')}", "checkpoints")


def save_rng_states() -> dict[str, Any]:
    state = {"python": random.getstate(), "numpy": np.random.get_state(), "torch_cpu": torch.get_rng_state()}
    if torch.cuda.is_available():
        state["torch_cuda"] = torch.cuda.get_rng_state_all()
    return state


def restore_rng_states(state: dict[str, Any]) -> None:
    random.setstate(state["python"])
    np.random.set_state(state["numpy"])
    torch.set_rng_state(state["torch_cpu"])
    if "torch_cuda" in state and torch.cuda.is_available():
        torch.cuda.set_rng_state_all(state["torch_cuda"])


@torch.no_grad()
def random_init_(model, seed: int, std: float = 0.02) -> None:
    """Seeded N(0, std^2) weights, unit norm scales (throughput and parity tests are weight-value independent)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for name, p in model.named_parameters():
        if name.endswith("scale"):
            p.fill_(1.0)
        else:
            rows = max(1, (1 << 22) // max(1, p.shape[-1]))  # stream in ~4M-element slabs to bound host memory
            flat = p.view(-1, p.shape[-1])
            for r0 in range(0, flat.shape[0], rows):
                blk = torch.randn(min(rows, flat.shape[0] - r0), flat.shape[1], generator=g) * std
                flat[r0:r0 + blk.shape[0]].copy_(blk.to(p.dtype))


class TuneCheckpointer:
    def __init__(self, checkpoint_dir: str | None = None, checkpoint_files: Any = None, config_json: Any = None,
                 output_dir: str | None = None, training_state_checkpoint: str | None = None, safe_serialization: bool = True,
                 model_expectations: Any = None, **_: Any) -> None:
        self.checkpoint_dir, self.output_dir = checkpoint_dir, output_dir
        self.training_state_checkpoint = training_state_checkpoint
        self.model_expectations = model_expectations

    def load_checkpoint(self) -> dict[str, Any]:
        out: dict[str, Any] = {MODEL_KEY: None}
        path = os.path.join(str(self.checkpoint_dir), MODEL_FILENAME) if self.checkpoint_dir else None
        if path and os.path.exists(path):
            from safetensors.torch import load_file
            out[MODEL_KEY] = load_file(path)
            LOGGER.info(f"Loaded {len(out[MODEL_KEY])} tensors from {path}")
        else:
            LOGGER.warning(f"No {MODEL_FILENAME} under {self.checkpoint_dir!r}: the model will be random-initialised (seeded).")
        if self.training_state_checkpoint is not None:
            state = torch.load(self.training_state_checkpoint, map_location="cpu", weights_only=False)  # own file, own writer
            out.update(state)
        return out

    def save_model_checkpoint(self, state_dict: dict[str, torch.Tensor], global_step: int) -> str:
        from safetensors.torch import save_file
        d = os.path.join(str(self.output_dir), f"step_{global_step}")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, MODEL_FILENAME)
        save_file({k: v.detach().to("cpu").contiguous() for k, v in state_dict.items()}, path)
        return path

    def save_training_state(self, optimizer_state_dict, lr_scheduler_state_dict, global_step: int, seed: int,
                            training_hparams: dict, consumed_samples: int, cumulative_metrics: dict) -> str:
        os.makedirs(str(self.output_dir), exist_ok=True)
        path = os.path.join(str(self.output_dir), TRAINING_STATE_FILENAME)
        torch.save({
            CHECKPOINT_VERSION_KEY: CHECKPOINT_VERSION, OPTIMIZER_KEY: optimizer_state_dict,
            LR_SCHEDULER_KEY: lr_scheduler_state_dict, GLOBAL_STEP_KEY: global_step, SEED_KEY: seed, RNG_KEY: save_rng_states(),
            TRAINING_HPARAMS_KEY: training_hparams, CONSUMED_SAMPLES_KEY: consumed_samples,
            CUMULATIVE_METRICS_KEY: cumulative_metrics}, path)
        return path
