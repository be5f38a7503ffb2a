#!/usr/bin/env python
"""End-to-end sanity of the drop-in path at full size: `compose(conf/sft.yaml)` -> Trainer.setup() -> Trainer.train() on the MLS-shaped synthetic
data, 1B model, bf16, B = 8, S = 2048, grad-accum 1, warm-up 10 steps to lr 2e-4 — the loss has to fall from ln V towards the entropy of the
synthetic token distribution (DSU ids uniform over 5000 values, Zipf text), and a dev loss is taken at the end.  Writes the curve as JSON.
usage: python tools/train_curve.py <steps> <out.json>"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "speech-integration_amd")
sys.path[:0] = [ROOT, PKG]
import torch  # noqa: E402
from ssi.config import compose  # noqa: E402
from ssi.train_utils import resolve_n_dsus  # noqa: E402
from ssi.trainer import Trainer  # noqa: E402

steps, out = int(sys.argv[1]), sys.argv[2]
tmp = tempfile.mkdtemp(prefix="ssi_curve_")
cfg = compose(os.path.join(PKG, "conf"), "sft", [
    "data=sft/mls-hubert_large_ll60k-layer_22", "dtype=bf16", f"max_steps={steps}", "gradient_accumulation_steps=1", "tokenizer.max_seq_len=2048",
    f"data.train.dataset.n_samples={steps * 8}", "data.dev.dataset.n_samples=16", "data.train.dataloader.batch_size=8", "data.dev.dataloader.batch_size=8",
    f"eval_steps={steps}", f"save_steps={steps * 1000}", "lr_scheduler.num_warmup_steps=10", f"output_dir={tmp}", f"checkpointer.output_dir={tmp}/ckpt",
    f"checkpointer.checkpoint_dir={tmp}/none", "checkpointer.allow_random_init=true", "speech.n_dsus=5000"])
resolve_n_dsus(cfg)
t = Trainer(cfg)
t.setup()
t0 = time.perf_counter()
t.train()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
rec = t.wandb_logger.records
curve = [{"step": r["step"], "loss": r["loss"], "lr": r["lr"], "tokens_per_second_per_gpu": r["tokens_per_second_per_gpu"]} for r in rec]
res = {"steps": steps, "train_wall_s": wall, "first_loss": curve[0]["loss"], "last_loss": curve[-1]["loss"], "min_loss": min(c["loss"] for c in curve),
       "dev_loss_at_end": rec[-1].get("dev_loss"), "tokens_total": rec[-1]["tokens_total"], "n_tokens": {k: v for k, v in rec[-1].items() if k.startswith("n_tokens.")},
       "curve": curve}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "curve"}))
print("loss every 10 steps:", [round(c["loss"], 3) for c in curve[::10]])
t.cleanup()
