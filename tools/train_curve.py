#!/usr/bin/env python
"""End-to-end sanity of the drop-in path at full size: `compose(conf/sft.yaml)` -> Trainer.setup() -> Trainer.train() on the MLS-shaped synthetic
data, 1B model, bf16, B = 8, S = 2048, grad-accum 1, warm-up 10 steps to lr 2e-4 — the loss has to fall from ln V towards the entropy of the
synthetic token distribution (DSU ids uniform over 5000 values, Zipf text), and a dev loss is taken at the end.  Writes the curve as JSON.
usage: python tools/train_curve.py <steps> <out.json> [--batch B] [--grad-accum G] [--ragged] [--both]"""
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

import argparse  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("steps", type=int)
ap.add_argument("out")
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--grad-accum", type=int, default=1)
ap.add_argument("--ragged", action="store_true", help="rows of unequal length, right-padded by the collate function")
ap.add_argument("--both", action="store_true",
                help="round 5: the run twice on the same data — the accumulation window as one batch (default, ssi/data/window.py) and as the "
                     "reference's micro-batch loop (fuse_accumulation_window=false) — and the two loss curves side by side")
args = ap.parse_args()
steps, out = args.steps, args.out


def run(joined: bool) -> dict:
    tmp = tempfile.mkdtemp(prefix="ssi_curve_")
    ga, B = args.grad_accum, args.batch
    cfg = compose(os.path.join(PKG, "conf"), "sft", [
        "data=sft/mls-hubert_large_ll60k-layer_22", "dtype=bf16", f"max_steps={steps}", f"gradient_accumulation_steps={ga}", "tokenizer.max_seq_len=2048",
        f"data.train.dataset.n_samples={steps * B * ga}", "data.dev.dataset.n_samples=16", f"data.train.dataloader.batch_size={B}", "data.dev.dataloader.batch_size=8",
        f"eval_steps={steps}", f"save_steps={steps * 1000}", "lr_scheduler.num_warmup_steps=10", f"output_dir={tmp}", f"checkpointer.output_dir={tmp}/ckpt",
        f"checkpointer.checkpoint_dir={tmp}/none", "checkpointer.allow_random_init=true", "speech.n_dsus=5000",
        f"fuse_accumulation_window={'true' if joined else 'false'}"] + (["data.train.dataset.fixed_len=false"] if args.ragged else []))
    resolve_n_dsus(cfg)
    t = Trainer(cfg)
    t.setup()
    t0 = time.perf_counter()
    t.train()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    rec = t.wandb_logger.records
    curve = [{"step": r["step"], "loss": r["loss"], "lr": r["lr"], "tokens_per_second_per_gpu": r["tokens_per_second_per_gpu"]} for r in rec]
    res = {"steps": steps, "batch": B, "grad_accum": ga, "ragged": args.ragged, "window_as_one_batch": bool(t.fused_micro_batches), "train_wall_s": wall,
           "first_loss": curve[0]["loss"], "last_loss": curve[-1]["loss"], "min_loss": min(c["loss"] for c in curve), "dev_loss_at_end": rec[-1].get("dev_loss"),
           "tokens_total": rec[-1]["tokens_total"], "n_tokens": {k: v for k, v in rec[-1].items() if k.startswith("n_tokens.")},
           "max_memory_reserved_gib": round(torch.cuda.max_memory_reserved() / 2**30, 1),
           "ms_per_step_percentiles_after_20_steps": (lambda d: {"p1": d[len(d) // 100], "p50": d[len(d) // 2], "p99": d[-max(1, len(d) // 100)], "max": d[-1]})(
               sorted(round(1e3 * r["duration_step"], 1) for r in rec[20:])) if len(rec) > 40 else None,
           "curve": curve}
    t.cleanup()
    del t
    torch.cuda.empty_cache()
    return res


if not args.both:
    res = run(True)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))
    print("loss every 10 steps:", [round(c["loss"], 3) for c in res["curve"][::10]])
else:
    a, b = run(True), run(False)
    assert a["window_as_one_batch"] and not b["window_as_one_batch"] and a["tokens_total"] == b["tokens_total"] and a["n_tokens"] == b["n_tokens"]
    rel = [abs(x["loss"] - y["loss"]) / y["loss"] for x, y in zip(a["curve"], b["curve"])]
    both = {"what": "the same data twice through Trainer.train(): the accumulation window as ONE batch against the reference's micro-batch loop",
            "max_rel_loss_difference": max(rel), "at_step": rel.index(max(rel)) + 1, "rel_loss_difference_every_10_steps": [round(r, 6) for r in rel[::10]],
            "window_as_one_batch": a, "micro_batch_loop": b}
    json.dump(both, open(out, "w"), indent=1)
    for name, r in (("one batch", a), ("loop", b)):
        print(name, json.dumps({k: v for k, v in r.items() if k not in ("curve", "n_tokens")}))
        print("  loss every 10 steps:", [round(c["loss"], 3) for c in r["curve"][::10]])
    print("max relative loss difference", max(rel), "at step", rel.index(max(rel)) + 1)
