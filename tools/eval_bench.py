#!/usr/bin/env python
"""Dev-set loss at full size (`Trainer._evaluate()`, reference ssi/eval.py:15-41): 1B model, bf16, N ragged dev samples in the dev loader's
batches of 2 rows — batch by batch as the reference runs them, and `eval_join_batches` at a time as one batch (ssi/eval.py, round 5).
usage: python tools/eval_bench.py [n_samples=512] [out.json]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tmp = tempfile.mkdtemp(prefix="ssi_eval_")
cfg = compose(os.path.join(PKG, "conf"), "sft", [
    "data=sft/mls-hubert_large_ll60k-layer_22", "dtype=bf16", "max_steps=1", "tokenizer.max_seq_len=2048", "data.train.dataset.n_samples=8",
    f"data.dev.dataset.n_samples={n}", "data.dev.dataset.fixed_len=false", "data.dev.dataloader.batch_size=2", f"output_dir={tmp}",
    f"checkpointer.output_dir={tmp}/ckpt", f"checkpointer.checkpoint_dir={tmp}/none", "checkpointer.allow_random_init=true", "speech.n_dsus=5000"])
resolve_n_dsus(cfg)
t = Trainer(cfg)
t.setup()
res = {"dev_samples": n, "dev_batch_size": 2, "runs": {}}
for join in (16, 0, 16, 0):
    t.cfg.eval_join_batches = join
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    value = t._evaluate()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res["runs"].setdefault(f"join_{join}", []).append({"seconds": round(dt, 3), "dev_loss": value})
    print(f"eval_join_batches={join:2d}: {dt:.3f} s, dev loss {value:.6f}", flush=True)
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
t.cleanup()
