"""Data-parallel gradient exchange over RCCL/xGMI (one process per GPU).

The reference has no gradient synchronisation at all (``DistributedSampler`` + ``world_size`` bookkeeping
only, ``/root/reference/ssi/trainer.py:519``).  This module supplies it MI355X-first: the model's gradients already live in
one flat HBM buffer, so the exchange is a handful of large in-place SUM all-reduces over contiguous per-layer buckets
(~122 MB bf16 each at the 1B shape; the tied embedding bucket, 546 MB, last), each issued on a side stream as soon as
backward has finished that bucket and overlapped with the rest of backward.  ``torch.distributed`` backend ``nccl`` is
RCCL on ROCm; ``gloo`` is used by the CPU tests.  After the exchange every rank scales by 1 / (sum over ranks of
``num_tokens_step``), which makes an N-GPU step equal to the reference's single-process step with N x as many
micro-batches.
"""

from __future__ import annotations

import datetime
import os
from typing import Optional

import torch
import torch.distributed as dist
from torch import Tensor


def get_world_size_and_rank() -> tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))


def _single_rank_exercise() -> bool:
    """SSI_DP_SINGLE=1: run the data-parallel machinery (process group, communicators, bucketed all-reduces on the side stream, scalar
    collective) with ONE rank.  A sum over one rank is the identity, so the step must equal the plain one bit for bit — the only way a one-GPU
    box can execute the RCCL calls themselves (``tests/test_dp_nccl_gpu.py``)."""
    return os.environ.get("SSI_DP_SINGLE", "0") == "1"


def _exchange_active(group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _single_rank_exercise())


def init_distributed(device: torch.device, timeout_s: int = 600) -> tuple[int, int]:
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun).  No-op for single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not _single_rank_exercise():
        return 1, 0
    if not dist.is_initialized():
        if world <= 1 and not all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT")):
            raise RuntimeError("SSI_DP_SINGLE=1 runs the data-parallel exchange with one rank and still needs the launcher's environment "
                               "(RANK, WORLD_SIZE, MASTER_PORT): start it under `python -m torch.distributed.run --nproc-per-node 1 ...`")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL's peer mappings fail with the legacy mode on this driver
        backend = os.environ.get("SSI_DIST_BACKEND") or ("nccl" if device.type == "cuda" else "gloo")
        kwargs = {}
        if device.type == "cuda" and backend == "nccl":
            torch.cuda.set_device(device)
            kwargs["device_id"] = device
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(seconds=timeout_s), **kwargs)
        global _created_process_group
        _created_process_group = True
    return dist.get_world_size(), dist.get_rank()


_created_process_group = False  # set by init_distributed when IT called init_process_group (a group made by the caller is the caller's to destroy)


def shutdown_distributed() -> None:
    """Tear down what ``init_distributed`` created — at ANY world size (the one-rank RCCL exercise creates a group and a scalar
    communicator too; left alive they leak into the next Trainer of the process and warn at exit)."""
    global _created_process_group
    if dist.is_available() and dist.is_initialized() and _created_process_group:
        dist.barrier()
        dist.destroy_process_group()
    _created_process_group = False


class GradSync:
    """Bucketed in-place SUM all-reduce of a flat gradient buffer.

    ``bucket_ready(name, lo, hi)`` is called by the model's backward as soon as ``flat_grad[lo:hi]`` is final;
    ``finish()`` reduces whatever was not announced and blocks the compute stream on all outstanding reductions."""

    def __init__(self, flat_grad: Tensor, buckets: list[tuple[str, int, int]], group=None):
        self.flat_grad, self.buckets, self.group = flat_grad, list(buckets), group
        self._pending: list = []
        self._pending_names: list[str] = []
        self._done: set[str] = set()
        self._deferred = None          # ((work, done event), (lo, hi)) of the bucket finish(defer_last=True) left in flight
        self._last_range = (0, 0)
        self._comm_stream = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self.bytes_reduced = 0
        self.timing = False            # bench: time how long the compute stream sits waiting for reductions (exposed communication)
        self._wait_events: list = []
        self._bucket_events: list = []  # (bucket, bytes, issue event, completion event) while `timing`
        self._bucket_waits: list = []   # (bucket, event before / after the compute stream's wait for it) while `timing`
        self._deferred_name: Optional[str] = None  # name of the bucket `finish(defer_last=True)` left in flight
        self._max_timing_records = 4096  # `timing` without anybody calling bucket_report(): the event lists do not grow without bound
        self._stream_ordered_wait = bool(flat_grad.is_cuda and dist.is_available() and dist.is_initialized()
                                         and dist.get_backend(group) == "nccl")
        # token counts and the running loss travel on a communicator of their own: on the gradients' one they would queue behind
        # every bucket already issued (a communicator runs its collectives in issue order), the deferred embedding bucket included
        self.scalar_group = None
        if self.enabled:
            try:
                self.scalar_group = dist.new_group()
            except Exception as e:  # a backend without communicator splitting: the scalars then share the gradients' communicator (slower, same result)
                import logging
                logging.getLogger(__name__).warning("GradSync: no second communicator for the scalar all-reduce (%r); using the default group", e)
        if flat_grad.is_cuda and self.enabled:
            # RCCL's kernels hold CUs for the length of a reduction while the backward GEMMs run: a persistent GEMM with a fixed
            # tile-to-workgroup map would wait a whole round for the workgroups that could not start (include/ssi_hip.h)
            from . import ops
            ops.set_gemm_tile_order(dynamic=True)

    @property
    def enabled(self) -> bool:
        return _exchange_active(self.group)

    def bucket_ready(self, name: str, lo: int, hi: int) -> None:
        if not self.enabled or name in self._done or hi <= lo:
            return
        self._done.add(name)
        self._last_range = (lo, hi)
        buf = self.flat_grad[lo:hi]
        self.bytes_reduced += buf.numel() * buf.element_size()
        t_issue = None
        if self._comm_stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                if self.timing:  # bench: issue -> completion per bucket (which bucket is the exposed one on a multi-GPU node)
                    t_issue = torch.cuda.Event(enable_timing=True)
                    t_issue.record(self._comm_stream)
                work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                if self._stream_ordered_wait:
                    # RCCL runs the collective on a stream of its own; a stream-level wait puts the SIDE stream behind it, so that the event
                    # recorded next marks the collective's completion (gloo's wait would block the host here: its event marks the issue)
                    work.wait()
                done = self._record_done(buf.device, timed=self.timing)
        else:
            work = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            done = None
        if t_issue is not None:
            self._bucket_events.append((name, buf.numel() * buf.element_size(), t_issue, done))
        self._pending.append((work, done))
        self._pending_names.append(name)

    def _record_done(self, device, timed: bool = False):
        """Event on the communication stream behind the collective just issued.  With RCCL (``_stream_ordered_wait``) the side stream has
        been put behind the collective by a stream-level ``work.wait()``, so the event marks its COMPLETION; with gloo it marks the issue
        (gloo's ``wait()`` blocks the host, it is called in ``_wait`` only).  Either way the compute stream's ordering also comes from
        ``work.wait()`` in ``_wait``, which orders the stream that is CURRENT when it is called — ``_wait`` must run with the compute stream
        current, never under the side stream."""
        if self._comm_stream is None:
            return None
        ev = torch.cuda.Event(enable_timing=timed)
        ev.record(self._comm_stream)
        return ev

    def _wait(self, item) -> None:
        work, done = item
        if self._comm_stream is not None:  # work.wait() orders the CURRENT stream: it must be the compute stream, never the side stream
            assert torch.cuda.current_stream(self.flat_grad.device) != self._comm_stream, "GradSync._wait under the communication stream"
        work.wait()
        if done is not None:
            torch.cuda.current_stream(self.flat_grad.device).wait_event(done)

    def finish(self, defer_last: bool = False) -> None:
        """Reduce whatever was not announced and make the compute stream wait for the reductions.  With ``defer_last`` the
        wait for the bucket issued last (the tied embedding: 546 MB that backward finishes at its very end, so nothing is
        left to hide it behind) is left to ``finish_deferred()``: the optimizer updates every other parameter first
        (``HipAdamW.step``) while that reduction is still on the links.  Only for callers that do not read the whole
        gradient in between (no global-norm clipping)."""
        self.finish_deferred()  # a bucket a caller left in flight is never dropped
        if self.enabled:
            for name, lo, hi in self.buckets:
                self.bucket_ready(name, lo, hi)
            if defer_last and len(self._pending) > 1:
                self._deferred = (self._pending.pop(), self._last_range)
                self._deferred_name = self._pending_names.pop()
            mark = first = self._mark()
            for i, item in enumerate(self._pending):  # every bucket but a deferred one: the compute stream waits for its completion event
                self._wait(item)
                if first is not None:  # timing: one more event per bucket = how long the compute stream sat in front of THIS bucket
                    nxt = self._mark()
                    self._bucket_waits.append((self._pending_names[i] if i < len(self._pending_names) else "?", mark, nxt))
                    mark = nxt
            if first is not None:
                self._wait_events.append((first, mark))
        self._pending.clear()
        self._pending_names.clear()
        self._done.clear()

    def deferred_range(self) -> Optional[tuple[int, int]]:
        """``[lo, hi)`` of the flat gradient that is not final until ``finish_deferred()``; None when nothing is pending."""
        return self._deferred[1] if self._deferred is not None else None

    def finish_deferred(self) -> None:
        if self._deferred is not None:
            mark = self._mark()
            self._wait(self._deferred[0])
            end = self._mark(mark)
            if end is not None:
                self._bucket_waits.append(((self._deferred_name or "?") + " (deferred)", mark, end))
            self._deferred = None
            self._deferred_name = None
        for rec in (self._bucket_events, self._bucket_waits, self._wait_events):  # a long timed run that never reads its report
            if len(rec) > self._max_timing_records:
                del rec[: len(rec) - self._max_timing_records]

    def _mark(self, start=None):
        """Timing events on the compute stream around a wait (only when ``timing``): the gap between them is the time the stream
        had nothing to run but the reductions."""
        if not (self.timing and self._comm_stream is not None):
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(self.flat_grad.device))
        if start is not None:
            self._wait_events.append((start, ev))
        return ev

    def exposed_ms(self) -> float:
        """Sum of the timed waits since the last call (synchronise the device first)."""
        total = sum(a.elapsed_time(b) for a, b in self._wait_events)
        self._wait_events.clear()
        return float(total)

    def bucket_report(self, steps: int) -> list[dict]:
        """Per bucket, averaged over ``steps`` timed steps (synchronise the device first): bytes, issue -> completion of its collective on
        the side stream (RCCL; with gloo the completion event marks the issue and the figure is ~0), and the time the compute stream spent
        waiting in front of it — the first multi-GPU run then says WHICH bucket is exposed without a second run."""
        out: dict[str, dict] = {}
        for name, nbytes, t0, t1 in self._bucket_events:
            d = out.setdefault(name, {"bucket": name, "bytes": nbytes, "issue_to_done_ms": 0.0, "exposed_ms": 0.0})
            if t1 is not None:
                d["issue_to_done_ms"] += t0.elapsed_time(t1) / max(steps, 1)
        for name, a, b in self._bucket_waits:
            key = name.replace(" (deferred)", "")
            d = out.setdefault(key, {"bucket": key, "bytes": None, "issue_to_done_ms": 0.0, "exposed_ms": 0.0})
            d["exposed_ms"] += a.elapsed_time(b) / max(steps, 1)
            if name.endswith("(deferred)"):
                d["deferred"] = True
        self._bucket_events.clear()
        self._bucket_waits.clear()
        return list(out.values())

    @classmethod
    def for_module(cls, module: torch.nn.Module, group=None) -> "ModuleGradSync":
        return ModuleGradSync(module, group)


class ModuleGradSync(GradSync):
    """Gradient exchange for a foreign ``nn.Module`` (stand-in models in host tests): gradients are gathered into one
    buffer at ``finish()``, reduced with a single collective and scattered back.  The HIP decoder never takes this route
    (its gradients are born flat and are reduced bucket by bucket during backward)."""

    def __init__(self, module: torch.nn.Module, group=None):
        self.params = [p for p in module.parameters() if p.requires_grad]
        dev = self.params[0].device
        super().__init__(torch.empty(0, device=dev), [], group)

    def bucket_ready(self, name: str, lo: int, hi: int) -> None:  # no early buckets for foreign modules
        return

    def finish(self, defer_last: bool = False) -> None:
        if not self.enabled:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        self.bytes_reduced += flat.numel() * flat.element_size()
        off = 0
        for g in grads:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()


def all_reduce_scalars(values: list[float], device: torch.device, group=None) -> list[float]:
    """SUM-reduce a few host scalars across ranks (token counts, running loss) in one tiny collective."""
    if not _exchange_active(group):
        return list(values)
    t = torch.tensor(values, dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t.tolist()
