"""Host -> device batch prefetch ("pinned-memory prefetch thread and async H2D so the step never waits on
the host"; the reference collates synchronously with ``num_workers: 0``, ``conf/data/_sft_base.yaml:23``, and copies each batch
with a blocking ``.to(device)`` inside the step, ``ssi/trainer.py:386``).

``DevicePrefetcher(loader, device, depth)`` iterates ``loader`` in a background thread, stages every tensor of a batch in reusable page-locked
buffers and copies it to ``device`` with ``non_blocking=True`` on a side stream; the consumer receives device-resident batches in the loader's order
and its compute stream is made to wait on the copy's event only (no host synchronisation).  Non-tensor values (lists of ids,
``seq_lens``) pass through.  Exceptions of the loader re-raise in the consumer.  On a CPU device it degrades to a plain
background-thread prefetch (used by the CPU tests).  ``transform`` (optional) is applied to every HOST batch in the background
thread before it is pinned and copied — the trainer passes ``ssi.data.unpad.unpad_batch`` there, which needs the rows' lengths and
gets them without a device sync."""

from __future__ import annotations

import queue
import threading
from typing import Any, Callable, Iterable, Iterator, Optional

import torch

_END = object()


class _PinnedRing:
    """Reusable page-locked staging buffers for the host side of the copies: ``slots`` sets of one buffer per batch key, handed out round robin;
    a set is reused only after the copies issued from it have finished (their event).  ``Tensor.pin_memory()`` per batch is fine while every
    batch has the same shape (torch's host allocator returns the block of the batch before last); ragged batches — another packed length every
    time — made it page-lock fresh memory for almost every batch, and that call stalls every other thread's launches inside the runtime for
    ~10 ms: a third of the GPU's time idle in the trainer's loop at 2 ragged rows (kernel trace, ``profiles/LAB_NOTES.md`` round 5)."""

    def __init__(self, slots: int) -> None:
        self.buffers: list[dict[str, torch.Tensor]] = [{} for _ in range(slots)]
        self.events: list[Any] = [None] * slots
        self.turn = 0

    def take(self) -> int:
        slot = self.turn % len(self.buffers)
        self.turn += 1
        if self.events[slot] is not None:
            self.events[slot].synchronize()  # (long finished: the set was used depth + 2 batches ago; blocks the prefetch thread only)
        return slot

    def stage(self, slot: int, key: str, src: torch.Tensor) -> torch.Tensor:
        nbytes = src.numel() * src.element_size()
        buf = self.buffers[slot].get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(2 * nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)  # twice what is asked: lengths vary, growth is rare
            self.buffers[slot][key] = buf
        view = buf[:nbytes].view(src.dtype).view(src.shape)
        view.copy_(src)
        return view


class DevicePrefetcher:
    def __init__(self, loader: Iterable[dict[str, Any]], device: torch.device | str, depth: int = 2,
                 transform: Optional[Callable[[dict[str, Any]], dict[str, Any]]] = None) -> None:
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.loader, self.device, self.depth, self.transform = loader, torch.device(device), int(depth), transform
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())

    def __len__(self) -> int:
        return len(self.loader)  # type: ignore[arg-type]

    def __getattr__(self, name: str) -> Any:  # .dataset, .sampler, .batch_size ... of the wrapped loader
        return getattr(self.loader, name)

    def _move(self, batch: dict[str, Any], stream, ring: Optional[_PinnedRing] = None) -> tuple[dict[str, Any], Any]:
        if self.device.type != "cuda":
            return batch, None
        out: dict[str, Any] = {}
        slot = ring.take() if ring is not None else -1

        def staged(key: str, v: torch.Tensor) -> torch.Tensor:
            if v.is_cuda or v.is_pinned() or not v.is_contiguous():
                return v
            return ring.stage(slot, key, v) if ring is not None else v.pin_memory()

        with torch.cuda.stream(stream):
            for k, v in batch.items():
                if torch.is_tensor(v):
                    out[k] = staged(k, v).to(self.device, non_blocking=True)
                elif getattr(v, "is_attn_plan", False):  # ssi.attn_plan.AttnPlan: keeps its host copy, gains a device copy
                    out[k] = type(v)(v.host, staged(k, v.host).to(self.device, non_blocking=True)) if v.dev is None else v
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(stream)
        if ring is not None:
            ring.events[slot] = ev
        return out, ev

    def __iter__(self) -> Iterator[dict[str, Any]]:
        q: queue.Queue = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        ring = _PinnedRing(self.depth + 2) if self.device.type == "cuda" else None  # queue + the batch in use + the one being filled

        def put(item: Any) -> bool:
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def worker() -> None:
            try:
                if self.device.type == "cuda":
                    torch.cuda.set_device(self.device)
                for batch in self.loader:
                    if self.transform is not None:
                        batch = self.transform(batch)
                    if not put(self._move(batch, stream, ring)):
                        return
                put(_END)
            except BaseException as e:  # noqa: BLE001 - handed to the consumer
                put(e)

        t = threading.Thread(target=worker, name="ssi-prefetch", daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is _END:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)  # stream-side wait: the host does not block
                    for v in batch.values():
                        if getattr(v, "is_attn_plan", False):
                            v = v.dev
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(torch.cuda.current_stream(self.device))
                yield batch
        finally:
            stop.set()
            t.join(timeout=5.0)
