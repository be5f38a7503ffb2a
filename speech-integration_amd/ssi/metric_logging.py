"""Metric sink with the duck type the trainer uses (``log_dict`` / ``log_config`` / ``close``).

The reference logs to Weights & Biases (``/root/reference/ssi/metric_logging.py:13-43``), an external service that is
out of scope here.  This logger writes JSON lines to ``<log_dir>/metrics.jsonl`` and keeps the last
records in memory; a real W&B logger with the same three methods can be injected through ``Trainer.wandb_logger``."""

from __future__ import annotations

import json
import os
from typing import Any


class MetricLogger:
    def __init__(self, log_dir: str | None = None, **_: Any) -> None:
        self.log_dir = log_dir
        self.records: list[dict] = []
        self.run_name, self.run_id = "local", "0"
        self._fh = None
        if log_dir:
            try:
                os.makedirs(log_dir, exist_ok=True)
                self._fh = open(os.path.join(log_dir, "metrics.jsonl"), "a")
            except OSError:
                self._fh = None

    def log_dict(self, payload: dict, step: int) -> None:
        rec = {"step": int(step)}
        for k, v in payload.items():
            rec[k] = float(v) if hasattr(v, "__float__") and not isinstance(v, (bool, str)) else v
        self.records.append(rec)
        if self._fh:
            self._fh.write(json.dumps(rec) + "\n")
            self._fh.flush()

    def log_config(self, cfg) -> None:
        if not self.log_dir:
            return
        try:
            from .config import DictConfig, OmegaConf
            text = OmegaConf.to_yaml(cfg, resolve=False) if isinstance(cfg, DictConfig) else str(cfg)
            with open(os.path.join(self.log_dir, "torchtune_config.yaml"), "w") as f:
                f.write(text)
        except OSError:
            pass

    def close(self) -> None:
        if self._fh:
            self._fh.close()
            self._fh = None


WandBLoggerPatched = MetricLogger
