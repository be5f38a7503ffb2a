"""Host -> device batch prefetch ("pinned-memory prefetch thread and async H2D so the step never waits on
the host"; the reference collates synchronously with ``num_workers: 0``, ``conf/data/_sft_base.yaml:23``, and copies each batch
with a blocking ``.to(device)`` inside the step, ``ssi/trainer.py:386``).

``DevicePrefetcher(loader, device, depth)`` iterates ``loader`` in a background thread, pins every tensor of a batch and copies it
to ``device`` with ``non_blocking=True`` on a side stream; the consumer receives device-resident batches in the loader's order
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

    def _move(self, batch: dict[str, Any], stream) -> tuple[dict[str, Any], Any]:
        if self.device.type != "cuda":
            return batch, None
        out: dict[str, Any] = {}
        with torch.cuda.stream(stream):
            for k, v in batch.items():
                if torch.is_tensor(v):
                    if not v.is_cuda:
                        v = v.pin_memory() if not v.is_pinned() else v
                    out[k] = v.to(self.device, non_blocking=True)
                elif getattr(v, "is_attn_plan", False):  # ssi.attn_plan.AttnPlan: keeps its host copy, gains a device copy
                    out[k] = v.to_device(self.device, non_blocking=True)
                else:
                    out[k] = v
            ev = torch.cuda.Event()
            ev.record(stream)
        return out, ev

    def __iter__(self) -> Iterator[dict[str, Any]]:
        q: queue.Queue = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None

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
                    if not put(self._move(batch, stream)):
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
