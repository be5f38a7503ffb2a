"""One rank of ``Trainer.train()`` on a small MFMA-shaped bf16 model: writes the per-step losses, the final weights' checksum and whether a
gradient exchange was set up.  ``tests/test_dp_nccl_gpu.py`` runs it twice on one GPU — plain, and under ``torchrun --nproc-per-node 1`` with
``SSI_DP_SINGLE=1`` (the RCCL exchange with one rank) — and expects the two runs to agree bit for bit."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "speech-integration_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--workdir", required=True)
    ap.add_argument("--clip", default="null")
    args = ap.parse_args()
    from ssi.config import compose
    from ssi.train_utils import resolve_n_dsus
    from ssi.trainer import Trainer
    cfg = compose(os.path.join(PKG, "conf"), "sft", [
        "data=sft/mls-speechtokenizer-rvq_0", "dtype=bf16", "tokenizer.max_seq_len=128", "data.train.dataloader.batch_size=2",
        "data.dev.dataloader.batch_size=2", "data.train.dataset.n_samples=32", "data.dev.dataset.n_samples=4", "gradient_accumulation_steps=2",
        "max_steps=5", "eval_steps=5", "save_steps=5000", "lr_scheduler.num_warmup_steps=2", "optimizer.lr=1e-2", f"clip_grad_norm={args.clip}",
        f"output_dir={args.workdir}", f"checkpointer.output_dir={args.workdir}/ckpt", f"checkpointer.checkpoint_dir={args.workdir}/none",
        "checkpointer.allow_random_init=true", "data.train.shuffle=false"])
    cfg.model_overrides = {"num_layers": 2, "num_heads": 4, "num_kv_heads": 2, "embed_dim": 256, "intermediate_dim": 512, "max_seq_len": 512,
                           "_base_vocab_size_txt": 300, "_n_special_txt": 16}
    cfg.speech.n_dsus = 50
    cfg.data.n_dsus = 50
    resolve_n_dsus(cfg)
    t = Trainer(cfg)
    t.setup()
    V = t._llama_config.vocab_size

    class Remap:  # the synthetic generator draws ids from the production layout; fold them into the shrunken vocabulary
        def __init__(self, loader):
            self.loader, self.dataset = loader, loader.dataset

        def __len__(self):
            return len(self.loader)

        def __iter__(self):
            for b in self.loader:
                tok = b["tokens"] % V
                yield {"tokens": tok, "labels": torch.where(b["labels"] == -100, b["labels"], tok)}

    t.data_train, t.data_dev = Remap(t.data_train), Remap(t.data_dev)
    t._loss_log = []
    t.train()
    torch.cuda.synchronize()
    flat = t.model._flat.float()
    res = {"losses": t._loss_log, "weights_sum": float(flat.double().sum()), "weights_abs_sum": float(flat.double().abs().sum()),
           "exchange": t.grad_sync is not None, "backend": (torch.distributed.get_backend() if torch.distributed.is_initialized() else None),
           "dev_loss": t.wandb_logger.records[-1].get("dev_loss"), "tokens_total": t.tokens_train_total,
           "bytes_reduced": (t.grad_sync.bytes_reduced if t.grad_sync is not None else 0)}
    json.dump(res, open(args.out, "w"))
    print(json.dumps(res))
    t.cleanup()
    return 0


if __name__ == "__main__":
    sys.exit(main())
