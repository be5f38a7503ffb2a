"""GPU, >= 2 devices: the data-parallel step over RCCL (torch.distributed backend "nccl") — per-bucket all-reduce on a side stream
during backward, scalar communicator, deferred embedding bucket, dynamic GEMM tile order — equals the single-process step over the
union of the micro-batches.  Skipped on a one-GPU box (the driver's test box has one GPU); the same worker is rehearsed there by hand
with two ranks sharing the GPU over gloo (see tests/workers/dp_step_worker.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_rccl_step_on_every_gpu_equals_single_process_step(tmp_path):
    n_ranks = min(torch.cuda.device_count(), 8)  # every GPU of the node: the first multi-GPU box exercises the 8-rank communicator
    if n_ranks < 2:   # device_count() does not initialise the GPU
        pytest.skip("needs >= 2 GPUs")
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds the GPU: ranks must be started from a process that has not touched it")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "dp.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SSI_DIST_BACKEND", None)
    env.pop("SSI_LOCAL_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "dp_step_worker.py"), "--out", str(out)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-4000:] + proc.stderr[-4000:]
    verdict = json.loads(out.read_text())
    assert verdict["backend"] == "nccl" and verdict["world"] == n_ranks and verdict["ok_all_ranks"], verdict


def test_single_rank_rccl_exchange_equals_the_plain_step(tmp_path):
    """One GPU is enough to EXECUTE the RCCL calls: ``SSI_DP_SINGLE=1`` runs the whole exchange (process group with ``device_id``, second
    communicator for the scalars, per-bucket ``all_reduce(async_op=True)`` on the side stream with its completion events, deferred embedding
    bucket, dynamic GEMM tile order, float64 scalar collective, barrier, teardown) with world size 1, where a sum is the identity: the step must
    equal the plain single-process step bit for bit.  Not a substitute for ranks on different GPUs — it pins the API use, the stream
    ordering and the protocol against the real library."""
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds the GPU: the rank must be started from a process that has not touched it")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = tmp_path / "dp1.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SSI_DP_SINGLE="1")
    env.pop("SSI_DIST_BACKEND", None)
    env.pop("SSI_LOCAL_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "workers", "dp_step_worker.py"), "--out", str(out)]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-4000:] + proc.stderr[-4000:]
    verdict = json.loads(out.read_text())
    assert verdict["backend"] == "nccl" and verdict["world"] == 1 and verdict["ok_all_ranks"], verdict
    assert verdict["weights_max_abs_err"] == 0.0 and verdict["summed_gradient_rel_err"] == 0.0 and verdict["bytes_reduced"] > 0, verdict


@pytest.mark.parametrize("clip", ["null", "0.5"])
def test_single_rank_rccl_trainer_equals_the_plain_trainer(tmp_path, clip):
    """``Trainer.train()`` (5 optimizer steps, grad-accum 2, warm-up schedule, a dev pass; with and without global-norm clipping — the clip path
    does not defer the embedding bucket) once plainly and once with the RCCL exchange switched on for its one rank: same losses, same dev loss,
    same weights, bit for bit."""
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    if torch.cuda.is_initialized():
        pytest.skip("this process already holds the GPU: the rank must be started from a process that has not touched it")
    worker = os.path.join(ROOT, "tests", "workers", "dp_trainer_worker.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("SSI_DIST_BACKEND", "SSI_LOCAL_DEVICE", "SSI_DP_SINGLE", "WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    plain_out = tmp_path / "plain.json"
    proc = subprocess.run([sys.executable, worker, "--out", str(plain_out), "--workdir", str(tmp_path / "plain"), "--clip", clip],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dp_out = tmp_path / "dp.json"
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                           "--master-port", str(port), worker, "--out", str(dp_out), "--workdir", str(tmp_path / "dp"), "--clip", clip],
                          env=dict(env, SSI_DP_SINGLE="1"), capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    plain, dp = json.loads(plain_out.read_text()), json.loads(dp_out.read_text())
    assert not plain["exchange"] and dp["exchange"] and dp["backend"] == "nccl" and dp["bytes_reduced"] > 0
    assert len(plain["losses"]) == 5 and dp["losses"] == plain["losses"], (plain["losses"], dp["losses"])
    assert dp["dev_loss"] == plain["dev_loss"] and dp["tokens_total"] == plain["tokens_total"]
    assert dp["weights_sum"] == plain["weights_sum"] and dp["weights_abs_sum"] == plain["weights_abs_sum"]
