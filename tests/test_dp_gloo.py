"""CPU, world_size 2 over gloo: the trainer's data-parallel step (gradient all-reduce + global token-count scaling) equals
the single-process step over the union of the micro-batches — the definition in SURVEY.md §8e.  The model here is the CPU
oracle used as a stand-in (tests may use it); the exchange code under test is ssi.distributed + ssi.trainer."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

PARAMS = dict(vocab_size=96, num_layers=2, num_heads=4, num_kv_heads=2, embed_dim=32, max_seq_len=64, intermediate_dim=64)
RANGES = {"text": (0, 63), "dsu": (64, 89), "modality": (90, 91), "special_text": (92, 95)}
PAD = 95


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batches():
    g = torch.Generator().manual_seed(7)
    out = []
    for i in range(4):  # 4 micro-batches: rank r takes 2r, 2r+1
        tok = torch.randint(0, 95, (2, 12 + i), generator=g)
        lab = tok.clone()
        lab[0, : 2 + i] = -100
        lab[1, -1 - i:] = -100
        out.append({"tokens": tok, "labels": lab})
    return out


def _make_trainer(world_size, rank, ga):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from unittest.mock import MagicMock
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, build_oracle
    from ssi.config import OmegaConf
    from ssi.distributed import GradSync
    from ssi.trainer import Trainer, TrainingGeometry
    cfg = OmegaConf.create({"gradient_accumulation_steps": ga, "clip_grad_norm": None, "eval_steps": 1000, "log_interval": 1,
                            "save_steps": 1000})
    t = Trainer(cfg)
    t.world_size, t.rank, t.device = world_size, rank, torch.device("cpu")
    t.model = build_oracle(PARAMS, seed=3)
    t.model.set_num_output_chunks(8)
    t.loss_fn = OracleCEWithChunkedOutputLoss()
    t.optimizer = torch.optim.SGD(t.model.parameters(), lr=0.5)  # linear in the gradient: isolates the exchange + scaling
    t.lr_scheduler = None
    t.wandb_logger, t.checkpointer = MagicMock(), MagicMock()
    t.tokenizer = MagicMock()
    t.tokenizer.pad_id = PAD
    t.token_type_ranges = RANGES
    t.geometry = TrainingGeometry(2, 100, 100 // ga, 100, 1, ga, world_size)
    t._loss_log = []
    if world_size > 1:
        t.grad_sync = GradSync.for_module(t.model)
    return t


def _worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        t = _make_trainer(world_size, rank, ga=2)
        mine = _batches()[2 * rank: 2 * rank + 2]
        for i, b in enumerate(mine):
            t._train_step({k: v.clone() for k, v in b.items()}, sync_gradients=(i == 1))
        t._optimizer_step(epoch=0, iter_idx=1)
        torch.save({"params": [p.detach().clone() for p in t.model.parameters()], "loss": t._loss_log,
                    "tokens": t.tokens_train_total, "counts": dict(t.token_type_counts_total),
                    "consumed": t.consumed_samples}, os.path.join(out_dir, f"rank{rank}.pt"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_step(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # single process, 4 micro-batches in one accumulation window
    t = _make_trainer(1, 0, ga=4)
    for b in _batches():
        t._train_step({k: v.clone() for k, v in b.items()})
    t._optimizer_step(epoch=0, iter_idx=3)
    ref = [p.detach() for p in t.model.parameters()]
    for a, b, c in zip(r0["params"], r1["params"], ref):
        assert torch.equal(a, b), "ranks diverged"
        assert torch.allclose(a, c, rtol=1e-5, atol=1e-7)
    assert r0["loss"] == pytest.approx(t._loss_log, rel=1e-6) and r1["loss"] == pytest.approx(t._loss_log, rel=1e-6)
    assert r0["tokens"] == r1["tokens"] == t.tokens_train_total  # global token count after the scalar all-reduce
    assert r0["consumed"] == 2 * 2 * 2  # ga * batch * world
    # the per-type totals are global on every rank, like tokens_total (they ride in the same scalar collective)
    assert r0["counts"] == r1["counts"] == dict(t.token_type_counts_total)


def _bucket_worker(rank, world_size, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        from ssi.distributed import GradSync
        flat = torch.arange(12, dtype=torch.float32) * (rank + 1)            # rank 0: 0..11, rank 1: 0..22 step 2
        buckets = [("norm", 10, 12), ("L0", 4, 10), ("emb", 0, 4)]           # the order backward finishes them
        sync = GradSync(flat, buckets)
        log = {}
        sync.bucket_ready(*buckets[0])                                       # announced during backward
        sync.bucket_ready(*buckets[0])                                       # twice: reduced once
        sync.finish(defer_last=True)                                         # issues L0 and emb, leaves emb in flight
        log["deferred"] = sync.deferred_range()
        log["after_finish"] = flat[4:].clone()
        sync.finish_deferred()
        log["after_deferred"] = flat.clone()
        log["bytes"] = sync.bytes_reduced
        assert sync.deferred_range() is None
        # second window, nothing deferred: everything final after finish()
        flat.copy_(torch.ones(12) * (rank + 1))
        sync.finish()
        log["second"] = flat.clone()
        torch.save(log, os.path.join(out_dir, f"b{rank}.pt"))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 8])
def test_bucketed_exchange_with_the_embedding_bucket_deferred(tmp_path, world):
    """GradSync on a flat gradient buffer: per-bucket SUM all-reduce, each bucket once; ``finish(defer_last=True)`` leaves the bucket
    issued last (the tied embedding) to ``finish_deferred()`` so that the optimizer can update the other parameters meanwhile.
    World size 8 is the target node's rank count (one-GPU boxes admit at most 6 processes on the card, so the GPU rehearsals stop at 5
    ranks: profiles/LAB_NOTES.md, round 4); the exchange logic itself — second communicator, bucket bookkeeping, deferral — runs here."""
    port = _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    total = world * (world + 1) // 2
    want = torch.arange(12, dtype=torch.float32) * total
    for r in range(world):
        log = torch.load(tmp_path / f"b{r}.pt", weights_only=False)
        assert log["deferred"] == (0, 4)
        assert torch.equal(log["after_finish"], want[4:])          # norm and L0 are final after finish()
        assert torch.equal(log["after_deferred"], want)            # the embedding bucket after finish_deferred()
        assert log["bytes"] == 12 * 4
        assert torch.equal(log["second"], torch.full((12,), float(total)))
