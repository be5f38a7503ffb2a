"""Constants of the hot path (reference: ``/root/reference/ssi/constants.py``).  Only what the training step reads is
kept: the seed (``:12``), supported dtypes (``:25``), checkpoint-schema keys (``:63-89``) and reserved batch keys (``:97``).
The reference's import-time dependency on ``sardalign``/``torchtune`` is not reproduced."""

import torch

SEED: int = 42_831
SUPPORTED_DTYPES: set = {torch.float32, torch.bfloat16}
PRECISION_STR_TO_DTYPE: dict = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "fp64": torch.float64}
CROSS_ENTROPY_IGNORE_IDX: int = -100

DEBUGGING_TAG: str = "trial-run"

MODEL_KEY: str = "model"
OPTIMIZER_KEY: str = "optimizer"
SEED_KEY: str = "seed"
EPOCHS_KEY: str = "epochs_run"
TOTAL_EPOCHS_KEY: str = "total_epochs"
GLOBAL_STEP_KEY: str = "global_step"
RNG_KEY: str = "rng_state"
TRAINING_HPARAMS_KEY: str = "training_hparams"
LR_SCHEDULER_KEY: str = "lr_scheduler"
CONSUMED_SAMPLES_KEY: str = "consumed_samples"
CUMULATIVE_METRICS_KEY: str = "cumulative_metrics"
CHECKPOINT_VERSION_KEY: str = "checkpoint_version"
CHECKPOINT_VERSION: int = 1

RESERVED_BATCH_KEYS: set = {"tokens", "mask", "labels"}
