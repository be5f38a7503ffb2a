#!/usr/bin/env python
"""Headline benchmark: train tokens/s of the speech-integration hot path (Llama-3.2-1B + 5000 HuBERT DSUs, SFT, bf16,
seq_len 2048, batch 8 per GPU) on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one optimizer step with gradient_accumulation_steps=1: micro-batch forward + backward (HIP kernels) ->
[N>1: per-layer RCCL all-reduce overlapped with backward + one scalar all-reduce] -> fused scale+AdamW kernel.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task statement) with
`roofline` (MFMA GEMM kernel family, HIP-event timed inside the timed region) and `cpu_baseline` (the CPU oracle timed on
the host cores on a bounded sample; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "speech-integration_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before any HIP call: RCCL needs dmabuf IPC on this driver
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X (MI355X_MICROARCH.md "Peak BF16/FP16 MFMA ~2.5 PF dense")
N_LAYERS_MM = 973_078_528  # matmul params per token in the 16 layers


def flops_per_token(vocab: int, seq: int, dim: int = 2048, layers: int = 16) -> float:
    """F_tok = 3 * [2 (N_layers_mm + D V) + 2 L D (S + 1)]  (fwd + 2x bwd, causal attention triangular)."""
    return 3.0 * (2.0 * (N_LAYERS_MM + dim * vocab) + 2.0 * layers * dim * (seq + 1))


class _HipEvents:
    """HIP events created with hipEventDisableSystemFence, straight from libamdhip64: torch.cuda.Event's default record carries a
    system-scope release (an L2 write-back after every timed launch), which slowed the step by 8 %; these do not."""

    def __init__(self):
        import ctypes
        self.ct = ctypes
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        self.hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pool = []

    def record(self, stream_ptr):
        ev = self.ct.c_void_p()
        rc = self.hip.hipEventCreateWithFlags(self.ct.byref(ev), 0x20000000)  # hipEventDisableSystemFence
        if rc != 0:
            raise RuntimeError(f"hipEventCreateWithFlags failed: {rc}")
        rc = self.hip.hipEventRecord(ev, self.ct.c_void_p(stream_ptr))
        if rc != 0:
            raise RuntimeError(f"hipEventRecord failed: {rc}")
        self.pool.append(ev)
        return ev

    def elapsed_ms(self, a, b) -> float:
        ms = self.ct.c_float()
        rc = self.hip.hipEventElapsedTime(self.ct.byref(ms), a, b)
        if rc != 0:
            raise RuntimeError(f"hipEventElapsedTime failed: {rc}")
        return float(ms.value)

    def close(self):
        for ev in self.pool:
            self.hip.hipEventDestroy(ev)
        self.pool = []


class GemmTimer:
    """HIP-event timing of every ssi_gemm launch inside the timed region (events on the stream the kernels run on)."""

    def __init__(self):
        self.records = []  # ((layout, epilogue class), flops, start, end)
        self.enabled = False
        self.ev = None

    def install(self):
        from ssi import ops
        inner = ops.gemm
        timer = self
        self.ev = _HipEvents()

        def timed_gemm(layout, a, b, c, **kw):
            if not timer.enabled:
                return inner(layout, a, b, c, **kw)
            M, N = c.shape
            K = a.shape[1] if layout in (ops.GEMM_NT, ops.GEMM_NN) else a.shape[0]
            st = torch.cuda.current_stream().cuda_stream
            s = timer.ev.record(st)
            inner(layout, a, b, c, **kw)
            e = timer.ev.record(st)
            prev = 1 if kw.get("accumulate") else (2 if kw.get("residual") is not None else 0)
            timer.records.append(((layout, prev), 2.0 * M * N * K, s, e))

        inner_b = ops.gemm_batched

        def timed_gemm_batched(layout, a, b, c, **kw):  # the deferred attention-projection weight gradients (one launch per group of layers)
            if not timer.enabled:
                return inner_b(layout, a, b, c, **kw)
            n, M, N = c.shape
            K = a.shape[2] if layout in (ops.GEMM_NT, ops.GEMM_NN) else a.shape[1]
            st = torch.cuda.current_stream().cuda_stream
            s = timer.ev.record(st)
            inner_b(layout, a, b, c, **kw)
            e = timer.ev.record(st)
            timer.records.append(((layout, 1 if kw.get("accumulate") else 0, "batched"), 2.0 * n * M * N * K, s, e))

        ops.gemm = timed_gemm
        ops.gemm_batched = timed_gemm_batched
        import ssi.model as m
        m.ops.gemm = timed_gemm
        m.ops.gemm_batched = timed_gemm_batched

    def summary(self):
        tot_ms, tot_fl, per = 0.0, 0.0, {}
        for key, fl, s, e in self.records:
            ms = self.ev.elapsed_ms(s, e)
            tot_ms += ms
            tot_fl += fl
            d = per.setdefault(key, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += ms
            d[2] += fl
        self.ev.close()
        return tot_ms, tot_fl, per


def pmc_traffic(kernel: str):
    """(HBM bytes per launch of `kernel`, the file it was read from) from the committed PMC passes (profiles/*_traffic.json, written by
    tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this same command) — NOT measured by this run;
    (None, None) when no pass covers the kernel."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    best, src = None, None
    for f in sorted(glob.glob(os.path.join(here, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("bench_kernel") == kernel and d.get("traffic_bytes_per_launch") is not None:
            best, src = d.get("traffic_bytes_per_launch"), os.path.join("profiles", os.path.basename(f))
    return best, src


def usable_cores() -> int:
    """CPU cores this process may really use (affinity mask and cgroup quota, not the host's total)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(seed: int, warmup: int = 1, steps: int = 3) -> dict:
    """CPU oracle (pure torch, fp32) timed on this host, as BASELINE.md §4 plans it: config P of BASELINE.json (B=2, S=512,
    V=133 258, full 16-layer 1B model), optimizer step = forward + backward + AdamW, `warmup` untimed + `steps` timed steps on
    different batches, mean reported.  Bounded sample: (warmup + steps) x 1024 tokens, about a minute on 16 cores."""
    from oracle import step_oracle
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from ssi.data import synthetic_batch
    from ssi.llama_configs import configllama3_2_1b
    import copy
    cores = min(usable_cores(), 32)
    torch.set_num_threads(cores)
    cfg = copy.deepcopy(configllama3_2_1b)
    cfg.n_dsus, cfg.modality_tokens = 5000, True
    t0 = time.perf_counter()
    print(f"[cpu_baseline] building the fp32 CPU oracle on {cores} threads ...", file=sys.stderr, flush=True)
    with torch.device("meta"):
        model = OracleLlama(**cfg.parameters, rope_cache_len=512)
    rope = model.rope.clone()  # built on the host inside the constructor
    model = model.to_empty(device="cpu")
    model.rope = rope
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith("scale"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02)
    model.set_num_output_chunks(8)
    loss_fn = OracleCEWithChunkedOutputLoss()
    opt = torch.optim.AdamW(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    t_build = time.perf_counter() - t0
    print(f"[cpu_baseline] built in {t_build:.1f} s; {warmup} warm-up + {steps} timed optimizer steps (B=2, S=512) ...", file=sys.stderr, flush=True)
    times, tokens, loss = [], 0, float("nan")
    for i in range(warmup + steps):
        batch = synthetic_batch(2, 512, 5000, seed=seed, index=i)
        t1 = time.perf_counter()
        lb, n = step_oracle.train_step(model, loss_fn, batch)
        step_oracle.optimizer_step(model, opt, n)
        dt = time.perf_counter() - t1
        print(f"[cpu_baseline] step {i} ({'warm-up' if i < warmup else 'timed'}): {dt:.1f} s", file=sys.stderr, flush=True)
        if i >= warmup:
            times.append(dt)
            tokens = batch["tokens"].numel()
            loss = lb / max(n, 1)
    mean = sum(times) / len(times)
    return {"value": tokens / mean, "unit": "tokens/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_name(),
            "sample": f"{warmup} warm-up + {steps} timed optimizer steps (fwd+bwd+AdamW) of config P: B=2 x S=512 = {tokens} tokens each, fp32, "
                      f"V=133258, 16 layers; mean {mean:.1f} s/step (min {min(times):.1f}, max {max(times):.1f}), {t_build:.1f} s model build; "
                      f"{cores} threads on {cpu_model_name()}", "loss": loss}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (one per GPU) through torch.distributed.run and
    relay rank 0's JSON line and the children's exit code.  Runs before this process has made any HIP call (a process that has
    touched the GPU must never exec or fork GPU work on this pool)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, usable_cores() // n)))
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


def through_trainer(args, result_out) -> int:
    """Secondary measurement (NOT the headline line): the same workload through the drop-in entry point — what ``scripts/train_sft.py``
    does (``/root/reference/scripts/train_sft.py:9-15``): compose ``conf/sft.yaml``, ``Trainer(cfg).setup(); .train()`` — with the device
    prefetcher, per-step logging and the reference's bookkeeping in the loop, at ``gradient_accumulation_steps`` 1 and the reference's
    default 4.  Reported: the trainer's own ``tokens_per_second_per_gpu`` (``/root/reference/ssi/trainer.py:462-467``: non-ignored label
    tokens of the window / wall time between optimizer steps) and the positions/s it corresponds to (B x S x ga / duration_step), both as the
    mean over the steps after the warm-up, next to the step time."""
    import shutil
    import tempfile
    from ssi.config import compose
    from ssi.train_utils import resolve_n_dsus
    from ssi.trainer import Trainer
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    runs = {}
    # grad-accum 4 twice: the window as ONE packed batch (the trainer's default, ssi/data/window.py) and as the reference's micro-batch loop
    for ga, joined in ((1, True), (4, True), (4, False)):
        key = f"grad_accum_{ga}" + ("" if joined else "_micro_batch_loop")
        tmp = tempfile.mkdtemp(prefix="ssi_through_trainer_")
        steps = args.warmup + args.steps
        route = "cpt" if args.cpt else "sft"   # scripts/train_cpt.py composes conf/cpt.yaml the same way
        cfg = compose(os.path.join(PKG, "conf"), route, [
            f"data={route}/mls-hubert_large_ll60k-layer_22", f"dtype={args.dtype}", f"max_steps={steps}", f"gradient_accumulation_steps={ga}",
            f"tokenizer.max_seq_len={args.seq}", f"data.train.dataset.n_samples={steps * ga * args.batch}", "data.dev.dataset.n_samples=8",
            f"data.train.dataloader.batch_size={args.batch}", "eval_steps=1000000000", "save_steps=1000000000", f"output_dir={tmp}",
            f"checkpointer.output_dir={tmp}/ckpt", f"checkpointer.checkpoint_dir={tmp}/none", "checkpointer.allow_random_init=true",
            f"speech.n_dsus={args.n_dsus}", f"fuse_accumulation_window={'true' if joined else 'false'}",
            f"lagged_readback={'false' if os.environ.get('SSI_LAGGED_READBACK') == '0' else 'true'}"]
            + (["data.train.dataset.fixed_len=false"] if args.padded else []))
        resolve_n_dsus(cfg)
        t = Trainer(cfg)
        t.setup()
        assert t.model.num_layers == 16 and t.model.vocab_size == 128_256 + args.n_dsus + 2
        t0 = time.perf_counter()
        t.train()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        rec = t.wandb_logger.records[args.warmup:]
        assert len(rec) == args.steps and t.global_step == steps
        dur = [r["duration_step"] for r in rec]
        runs[key] = {
            "tokens_per_second_per_gpu": sum(r["tokens_per_second_per_gpu"] for r in rec) / len(rec),
            "positions_per_second": sum(ga * args.batch * args.seq / d for d in dur) / len(dur),
            "ms_per_optimizer_step": 1e3 * sum(dur) / len(dur), "ms_per_micro_batch": 1e3 * sum(dur) / len(dur) / ga,
            "median_ms_per_optimizer_step": 1e3 * sorted(dur)[len(dur) // 2], "ms_of_each_optimizer_step": [round(1e3 * d, 2) for d in dur],
            "label_tokens_of_each_optimizer_step": [int(round(r["tokens_per_second_per_gpu"] * r["duration_step"])) for r in rec],
            "steps": len(rec), "warmup": args.warmup, "train_wall_s": wall, "last_loss": rec[-1]["loss"],
            "micro_batches_joined_into_one_batch_per_window": t.fused_micro_batches}
        if args.padded:  # ragged rows: the prefetch thread dropped the padding and built the attention backward's work plan beside each batch
            from ssi import _lib as _l, ops as _o
            used = _o.attn_last_dispatch()
            runs[key].update({
                "micro_batches_run_without_their_padding": t.unpadded_micro_batches,
                "attention_backward_of_the_last_micro_batch": {"dq2": bool(used & _l.ATTN_USED_DQ2), "dkv2": bool(used & _l.ATTN_USED_DKV2),
                                                               "work_plan": bool(used & _l.ATTN_USED_PLAN), "head_split": bool(used & _l.ATTN_USED_HEAD_SPLIT)}})
        t.cleanup()
        del t
        torch.cuda.empty_cache()
        shutil.rmtree(tmp, ignore_errors=True)
    result_out.emit(json.dumps({"metric": "train_tokens_per_sec", "mode": "through_trainer", "unit": "tokens/s", "n_gpus": 1, "dtype": args.dtype,
                      "data": "synthetic", "higher_is_better": True,
                      "config": {"workload": f"scripts/train_{route}.py path: compose(conf/{route}.yaml) -> Trainer.setup() -> Trainer.train(); Llama-3.2-1B "
                                             f"+{args.n_dsus} DSUs, seq_len={args.seq}, batch={args.batch}, prefetcher on, log_interval=1, 16 layers, "
                                             "random-init weights, MLS-shaped synthetic DSU sequences"
                                             + (", rows of unequal length right-padded by the collate function (the reference's batch format)" if args.padded else "")},
                      "value": runs["grad_accum_1"]["positions_per_second"], "runs": runs}))
    return 0


class _JsonOnlyStdout:
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a five-line version banner to stdout when the first
    communicator is created — seen on the first run that reached it, profiles/r03_s): everything this process and its libraries print goes to
    stderr, and the result line alone is written to the original stdout."""

    def __init__(self):
        sys.stdout.flush()
        self.fd = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line: str) -> None:
        sys.stdout.flush()
        os.write(self.fd, (line.rstrip("\n") + "\n").encode())


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=2048)
    ap.add_argument("--n-dsus", type=int, default=5000)
    ap.add_argument("--layers", type=int, default=16, help="debug only; the headline number needs 16")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--packed", action="store_true",
                    help="secondary workload (BASELINE config E at one GPU): rows packed with ~440-1100-token documents, block-causal "
                         "attention; use with --seq 8192 --batch 2.  Not the headline line.")
    ap.add_argument("--padded", action="store_true",
                    help="secondary workload (SURVEY.md §8d): sequence lengths ~U(0.4 S, S), right-padded to the batch maximum; "
                         "tokens/s then counts NON-PAD tokens.  Not the headline line.")
    ap.add_argument("--grad-accum", type=int, default=1,
                    help="micro-batches per optimizer step (the headline line uses 1, the most conservative reading; the reference's default is 4, "
                         "conf/training.yaml:11): a step is then grad-accum forward + backward passes and one AdamW pass.  Not the headline line.")
    ap.add_argument("--no-unpad", action="store_true",
                    help="with --padded: run the right-padded rows as they are (the round-1..3 behaviour) instead of dropping the padding on the "
                         "host as the trainer's prefetch thread does (ssi/data/unpad.py)")
    ap.add_argument("--no-attn-plan", action="store_true",
                    help="with --packed / --padded: no work plan for the attention backward (ssi/attn_plan.py), i.e. the round-1..3 kernels on the "
                         "packed rows — the in-run comparison partner of the plan")
    ap.add_argument("--through-trainer", action="store_true",
                    help="secondary line: the workload through Trainer.setup()/train() (the scripts/train_sft.py path) at grad-accum 1 and 4; "
                         "one GPU.  Not the headline line.")
    ap.add_argument("--cpt", action="store_true", help="with --through-trainer: the scripts/train_cpt.py path (conf/cpt.yaml; use --batch 16 --seq 768)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gemm-timing", action="store_true")
    args = ap.parse_args()

    rank, world, local = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)  # nothing has touched the GPU yet
    if world != args.gpus:
        print(f"bench.py --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}", file=sys.stderr)
        return 2
    result_out = _JsonOnlyStdout()  # from here on fd 1 is stderr; only result_out.emit() reaches the real stdout
    if args.through_trainer:
        return through_trainer(args, result_out)
    from ssi.train_utils import limit_host_threads
    limit_host_threads(world)  # torch's CPU thread pool within this process's CPU share (cpu_baseline sets its own count afterwards)
    local = int(os.environ.get("SSI_LOCAL_DEVICE", local))  # rehearsal hook: several ranks on one GPU (with SSI_DIST_BACKEND=gloo)
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)

    import copy
    from ssi.data import synthetic_batch, synthetic_packed_batch
    from ssi.distributed import GradSync, all_reduce_scalars, init_distributed
    from ssi.llama_configs import configllama3_2_1b
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    from ssi.optimizer import HipAdamW, scale_grads
    from ssi.train_utils import count_token_types_async, get_token_type_ranges

    init_distributed(device)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lcfg = copy.deepcopy(configllama3_2_1b)
    lcfg.n_dsus, lcfg.modality_tokens = args.n_dsus, True
    lcfg.num_layers = args.layers
    params = lcfg.parameters
    torch.manual_seed(42_831)
    model = HipLlamaDecoder(**params, dtype=dtype, device=device, rope_cache_len=max(args.seq, 2048))
    with torch.no_grad():
        model._flat.normal_(0.0, 0.02)
        model._view("emb")[lcfg.vocab_size:].zero_()
        for p, name, _ in model._param_src:
            if name.endswith("norm"):
                p.fill_(1.0)
    model.train()
    loss_fn = CEWithChunkedOutputLoss()
    model.set_num_output_chunks(loss_fn.num_output_chunks)
    opt = HipAdamW(model.parameters(), model=model, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, amsgrad=False, fused=True)
    sync = None
    dp = world > 1 or os.environ.get("SSI_DP_SINGLE") == "1"  # SSI_DP_SINGLE=1 under torchrun --nproc-per-node 1: the RCCL calls with one rank
    if dp:
        sync = GradSync(model._flat_grad, model.buckets)
        model.grad_sync = sync
    ranges = get_token_type_ranges(lcfg)
    pad_id = lcfg._base_vocab_size_txt + lcfg.n_dsus + 2 + 4

    n_total = args.warmup + args.steps
    # host-side preparation of a batch, as the trainer's prefetch thread does it before the batch is copied over (outside the timed region like
    # every preparation of a resident batch): the padding dropped, the work plan of the attention backward built from the HOST input_pos
    plan_fn = None if args.no_attn_plan else model.build_attn_plan
    to_dev = lambda v: v.to(device) if torch.is_tensor(v) else (v.to_device(device) if getattr(v, "is_attn_plan", False) else v)  # noqa: E731
    from ssi.data import loss_inputs, unpad_batch
    if args.packed:
        host = [synthetic_packed_batch(args.batch, args.seq, args.n_dsus, seed=42_831 + i, rank=rank) for i in range(min(n_total, 4))]
        if plan_fn is not None:
            host = [unpad_batch(b, pad_id=pad_id, plan_fn=plan_fn) for b in host]   # (a batch that arrives packed only gains its plan)
    else:
        host = [synthetic_batch(args.batch, args.seq, args.n_dsus, rank=rank, index=i % 4, fixed_len=not args.padded) for i in range(min(n_total, 4))]
        if args.padded and not args.no_unpad:
            host = [unpad_batch(b, pad_id=pad_id, padded_len=model.padded_seq_len, plan_fn=plan_fn) for b in host]
    batches = [{k: to_dev(v) for k, v in b.items()} for b in host]
    n_plans = sum(1 for b in batches if any(getattr(v, "is_attn_plan", False) for v in b.values()))
    if os.environ.get("SSI_BENCH_TILE_ORDER"):  # diagnostic: price of the data-parallel tile order on one GPU ("dynamic" | "static")
        from ssi import ops as _ops
        _ops.set_gemm_tile_order(dynamic=os.environ["SSI_BENCH_TILE_ORDER"] == "dynamic")
    timer = GemmTimer()
    if not args.no_gemm_timing and rank == 0:
        timer.install()

    ga = max(1, args.grad_accum)
    overlap_adamw = os.environ.get("SSI_ADAMW_OVERLAP", "1") != "0" and sync is None   # (A/B switch; what the trainer does when it does not clip)

    def one_step(i: int) -> tuple[float, int]:
        rows = []
        n_dev = None
        for j in range(ga):  # the accumulation window: only its last backward exchanges gradients; counts and losses stay on the device
            b = batches[(i * ga + j) % len(batches)]
            counts = count_token_types_async(b["tokens"], ranges, pad_id, b["labels"], -100)
            model.sync_this_backward = j == ga - 1
            n_dev = counts[-1] if n_dev is None else n_dev + counts[-1]
            if j == ga - 1 and overlap_adamw:  # the window's token count is known on the device before its last backward: AdamW runs under it
                opt.overlap_with_backward(1.0 / n_dev.to(torch.float32))
            loss_batch = compute_loss(loss_inputs(b) if not args.packed else b, model, loss_fn) * counts[-1]
            loss_batch.backward()
            rows.append(torch.cat((counts.double(), loss_batch.detach().double().reshape(1))))
        host = torch.stack(rows).sum(0).tolist()  # the step's one D2H sync
        n_tok, loss_run = int(host[-2]), host[-1]
        if sync is not None:
            n_tok, loss_run = (lambda v: (int(round(v[0])), v[1]))(all_reduce_scalars([n_tok, loss_run], device, group=sync.scalar_group))
            sync.finish(defer_last=not os.environ.get("SSI_BENCH_NO_DEFER"))
        scale_grads(model, torch.tensor(1.0 / n_tok))
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss_run / n_tok, n_tok

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    loss = float("nan")
    for i in range(args.warmup):
        loss, _ = one_step(i)
    barrier()
    timer.enabled = True
    if sync is not None:
        sync.timing, sync.bytes_reduced = True, 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss, _ = one_step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if dp:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    tokens_per_step = args.batch * args.seq * world * ga
    if args.padded:  # non-pad tokens actually processed in the timed steps (all ranks draw the same length distribution)
        tokens_per_step = world * sum(int((batches[((args.warmup + i) * ga + j) % len(batches)]["tokens"] != pad_id).sum())
                                      for i in range(args.steps) for j in range(ga)) / args.steps
    value = tokens_per_step * args.steps / elapsed
    f_tok = flops_per_token(lcfg.vocab_size, args.seq, layers=args.layers) if args.layers == 16 else None
    if args.packed and f_tok:  # attention term over the documents instead of the whole row
        nn, n1 = 0.0, 0.0
        for bt in batches:
            for lens in bt["seq_lens"]:
                nn += float((lens.double() * (lens.double() + 1)).sum())
                n1 += float(lens.sum())
        f_tok = flops_per_token(lcfg.vocab_size, 0, layers=args.layers) - 3 * 2 * args.layers * 2048 + 3 * 2 * args.layers * 2048 * nn / n1
    if rank == 0:
        out = {
            "metric": "train_tokens_per_sec", "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Llama-3.2-1B +{args.n_dsus} DSUs (V={lcfg.vocab_size}), SFT step fwd+bwd+AdamW, seq_len={args.seq}, "
                                   f"batch={args.batch}/GPU, grad_accum={ga}, {args.layers} layers, random-init weights, MLS-shaped synthetic DSU sequences"
                                   + (", rows packed with 440-1100-token documents (block-causal attention)" if args.packed else "")
                                   + (", lengths ~U(0.4 S, S) right-padded, non-pad tokens counted" if args.padded else "")
                                   + (", padding dropped on the host (ssi/data/unpad.py)" if args.padded and not args.no_unpad else ""),
                       "global_batch": args.batch * world * ga, "seq_len": args.seq, "parallelism": f"dp{world}", "last_loss": loss},
        }
        # which attention backward kernels the last micro-batch took (ssi_attn_last_dispatch): a secondary line says what path its number is from
        from ssi import _lib as _l, ops as _o
        used = _o.attn_last_dispatch()
        out["attention_backward"] = {
            "dq": ("attn_bwd_dq2_kernel<plan>" if used & _l.ATTN_USED_PLAN else f"attn_bwd_dq2_kernel<{(used >> 8) & 15}>") if used & _l.ATTN_USED_DQ2 else "attn_bwd_dq_kernel",
            "dkv": ("attn_bwd_dkv2_kernel<plan>" if used & _l.ATTN_USED_PLAN else "attn_bwd_dkv2_kernel") if used & _l.ATTN_USED_DKV2
                   else ("attn_bwd_dkv_kernel<head split>" if used & _l.ATTN_USED_HEAD_SPLIT else "attn_bwd_dkv_kernel"),
            "batches_with_a_work_plan": f"{n_plans} of {len(batches)}"}
        if f_tok:
            out["mfma_roofline_frac_step"] = value * f_tok / (world * MFMA_PEAK_TFLOPS * 1e12)
            out["gflop_per_token"] = f_tok / 1e9
        if timer.records:
            _, _, per = timer.summary()
            # one kernel symbol per (operand layout, epilogue) class: gemm_nt4dma_kernel<A_COL, B_COL, EPI_PLAIN, PREV, SPLITK=false, BATCHED> (the LDS-DMA
            # loop); the roofline object is the class with the largest share of the step
            lay = {0: "false,false", 1: "false,true", 2: "true,true"}
            sym = {(l, pv): f"gemm_nt4dma_kernel<{lay[l]},0,{pv},false,false>" for l in lay for pv in (0, 1, 2)}
            sym.update({(l, pv, "batched"): f"gemm_nt4dma_kernel<{lay[l]},0,{pv},false,true>" for l in lay for pv in (0, 1)})
            dom = max(per, key=lambda k: per[k][1])
            n, ms, fl = per[dom]
            traffic, traffic_src = pmc_traffic(sym.get(dom, ""))
            out["roofline"] = {
                "bound": "mfma", "kernel": sym.get(dom, f"gemm_mfma_kernel layout={dom[0]}"),
                "achieved": fl / (ms * 1e-3) / 1e12, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                "launches": n, "avg_launch_ms": ms / n, "flop_per_launch_avg": fl / n,
                "share_of_step_time": ms / (1e3 * elapsed),
                "other_gemm_kernels": {sym.get(k, str(k)): {"launches": v[0], "avg_ms": v[1] / v[0], "tflops": v[2] / (v[1] * 1e-3) / 1e12}
                                       for k, v in per.items() if k != dom},
            }
        if sync is not None:  # what crossed xGMI and how much of it the step had to wait for (rank 0's view)
            out["comm"] = {"backend": dist.get_backend(), "library": "RCCL" if dist.get_backend() == "nccl" else dist.get_backend(),
                           "ranks": dist.get_world_size(), "allreduce_bytes_per_step": sync.bytes_reduced / args.steps,
                           "buckets": len(sync.buckets), "exposed_ms_per_step": sync.exposed_ms() / args.steps,
                           # per bucket: bytes, issue -> completion of its all-reduce, time the compute stream waited in front of it
                           "per_bucket": sync.bucket_report(args.steps)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(42_831)
            except Exception as e:  # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": usable_cores(), "kind": "port", "sample": f"failed: {e!r}"}
        result_out.emit(json.dumps(out))
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
