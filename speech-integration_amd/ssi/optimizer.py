"""Optimizer of the training step (reference: ``/root/reference/ssi/optimizer.py:8-17``, ``conf/training.yaml:2-10``).

``setup_optimizer(cfg, model, optimizer_state_dict)`` keeps the reference signature.  :class:`HipAdamW` is a
``torch.optim.Optimizer`` (so ``LambdaLR``, ``param_groups``, ``state_dict`` work unchanged) whose ``step`` is ONE HIP
kernel over the model's flat parameter / gradient / moment buffers: decoupled weight decay, fp32 op-math, a single
rounding on store — the semantics of ``torch.optim.AdamW(fused=True)`` (K13), with ``scale_grads`` (K11) and the
clip coefficient (K12) folded in as a device-side gradient multiplier.  The gradient buffer is not zeroed: the first backward of the
next accumulation window overwrites it (``HipLlamaDecoder`` gradient-buffer protocol)."""

from __future__ import annotations

import weakref
from typing import Any, Iterable

import torch
from torch import Tensor

from . import ops


class HipAdamW(torch.optim.Optimizer):
    def __init__(self, params: Iterable, lr: float = 1e-3, betas: tuple[float, float] = (0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, amsgrad: bool = False, fused: bool | None = None, *, model=None, **unused: Any):
        if amsgrad:
            raise NotImplementedError("amsgrad=True is not supported (reference default: false, conf/training.yaml:8)")
        if unused:
            raise TypeError(f"unsupported AdamW arguments: {sorted(unused)}")
        if model is None or not hasattr(model, "_flat"):
            raise TypeError("HipAdamW needs model=<HipLlamaDecoder> (flat parameter buffers)")
        defaults = dict(lr=float(lr), betas=tuple(float(b) for b in betas), eps=float(eps), weight_decay=float(weight_decay),
                        amsgrad=False, fused=True)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError("one param group expected (the reference passes model.parameters())")
        self.model = model
        model._hip_optimizer = weakref.ref(self)  # scale_grads / clip_grad_norm_ may defer their factor only while THIS optimizer consumes it
        self._exp_avg = torch.zeros_like(model._flat)
        self._exp_avg_sq = torch.zeros_like(model._flat)
        self._step_count = 0
        self._views_ready = False
        self._side = None    # stream of the updates that run under the backward (overlap_with_backward)
        self._armed = None

    def _ensure_state_views(self) -> None:
        if self._views_ready:
            return
        m = self.model
        for p, name, rows in m._param_src:
            self.state[p] = {
                "step": torch.tensor(float(self._step_count)),
                "exp_avg": m._view(name, rows, self._exp_avg),
                "exp_avg_sq": m._view(name, rows, self._exp_avg_sq),
            }
        self._views_ready = True

    # ---- AdamW under the backward (one GPU, no clipping) -----------------------------------------------------------------------------------
    def overlap_with_backward(self, grad_scale_dev: Tensor) -> bool:
        """Arm the optimizer for the window's LAST backward: every bucket of the flat gradient buffer (``HipLlamaDecoder.buckets``: the final
        norm, each layer's MLP block, each group's attention weights, the tied embedding) is updated on a side stream the moment that backward
        has finished its gradients, while the backward goes on — AdamW is a pure HBM pass (17 GB) and hides partly under the MFMA-bound
        GEMMs.  ``grad_scale_dev`` = the factor ``scale_grads`` would apply afterwards (1 / the window's token count; a DEVICE scalar, so no
        read-back is needed before the backward); ``step()`` then only joins the side stream and counts the step.  The arithmetic per element is
        the one of ``step()`` — same kernel, same scale, same step number — so the parameters come out bit for bit the same.  Not armed (returns
        False: the plain ``step()`` does all the work) under data parallelism (the buckets are still being reduced), with the round-1
        zero-and-accumulate protocol, or when gradients are to be clipped (the global norm needs every gradient first): the caller decides
        the latter by not calling this.  A window without a single label leaves a non-finite scale: the kernel then changes nothing, and the
        caller, who learns of the empty window at its read-back, must not call ``step()`` — exactly the reference's skip."""
        m = self.model
        if getattr(m, "grad_sync", None) is not None or m.always_accumulate or not m._flat.is_cuda:
            return False
        if self._side is None:
            self._side = torch.cuda.Stream(device=m._flat.device)
        self._armed = {"scale": grad_scale_dev.to(torch.float32).reshape(1), "done": [], "step": self._step_count + 1}
        m.bucket_listener = self._bucket_final
        return True

    def _bucket_final(self, name: str, lo: int, hi: int) -> None:
        a, g, m = self._armed, self.param_groups[0], self.model
        if a is None or hi <= lo:
            return
        cur = torch.cuda.current_stream(m._flat.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            ops.adamw_step(m._flat[lo:hi], m._flat_grad[lo:hi], self._exp_avg[lo:hi], self._exp_avg_sq[lo:hi], lr=float(g["lr"]),
                           beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"], step=a["step"],
                           grad_scale_dev=a["scale"], zero_grad=False, skip_nonfinite_scale=True)
        a["done"].append((lo, hi))

    def cancel_overlap(self) -> None:
        """Forget an armed overlap (a window that turned out empty: its updates were no-ops by the kernel's own guard)."""
        if self._armed is not None:
            torch.cuda.current_stream(self.model._flat.device).wait_stream(self._side)
        self._armed = None
        self.model.bucket_listener = None

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        m = self.model
        if self._armed is not None and self._armed["done"]:
            # the buckets were updated under the backward: join the side stream, update whatever no bucket covered (nothing, by construction)
            a, self._armed = self._armed, None
            m.bucket_listener = None
            torch.cuda.current_stream(m._flat.device).wait_stream(self._side)
            self._step_count += 1
            assert a["step"] == self._step_count
            covered = sorted(a["done"])
            pos, n = 0, m._flat.numel()
            for lo, hi in covered + [(n, n)]:
                if lo > pos:  # a gap between buckets (none in HipLlamaDecoder's layout): the plain update, same scale
                    ops.adamw_step(m._flat[pos:lo], m._flat_grad[pos:lo], self._exp_avg[pos:lo], self._exp_avg_sq[pos:lo], lr=float(g["lr"]),
                                   beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"], step=self._step_count,
                                   grad_scale_dev=a["scale"], zero_grad=False)
                pos = max(pos, hi)
            m.pending_grad_scale = None  # (scale_grads of the caller: already applied, bucket by bucket)
            m._hip_epoch += 1
            if self._views_ready:
                for st in self.state.values():
                    st["step"].fill_(float(self._step_count))
            return loss
        self._armed = None
        if all(p.grad is None for p, _, _ in m._param_src):
            # no backward since the last zero_grad (skipped or failed micro-batches): torch.optim.AdamW skips parameters without a
            # gradient; the never-zeroed buffer still holds the LAST window's gradients, which must not be applied a second time
            sync = getattr(m, "grad_sync", None)
            if sync is not None and hasattr(sync, "finish_deferred"):
                sync.finish_deferred()
            m.pending_grad_scale = None
            return loss
        self._step_count += 1

        def update(lo: int, hi: int) -> None:
            if hi > lo:
                ops.adamw_step(m._flat[lo:hi], m._flat_grad[lo:hi], self._exp_avg[lo:hi], self._exp_avg_sq[lo:hi], lr=float(g["lr"]),
                               beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                               step=self._step_count, grad_scale_dev=m.pending_grad_scale, zero_grad=m.always_accumulate)

        sync = getattr(m, "grad_sync", None)
        tail = sync.deferred_range() if sync is not None and hasattr(sync, "deferred_range") else None
        n = m._flat.numel()
        if tail is None:
            update(0, n)
        else:  # data parallel: everything but the bucket still being reduced first, that bucket once it has arrived
            lo, hi = tail
            update(0, lo)
            update(hi, n)
            sync.finish_deferred()
            update(lo, hi)
        m.pending_grad_scale = None
        # the gradients are NOT zeroed (2.5 GB of writes per step for nothing) and the window is NOT closed here: as with torch.optim, a
        # backward that follows a step without a zero_grad in between accumulates; zero_grad (this optimizer's, the model's, or a foreign
        # one that drops every p.grad) ends the window, and the first backward after it overwrites the buffer (model protocol)
        if m.always_accumulate:  # SSI_ZERO_GRADS=1 (round-1 protocol): the kernel has just zeroed the buffer, a memset in zero_grad would do it twice
            m._grads_dirty = m._grads_stale = False
        m._hip_epoch += 1  # weights changed behind torch's version counter
        if self._views_ready:
            for st in self.state.values():
                st["step"].fill_(float(self._step_count))
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        # nothing to clear after a step (the next backward overwrites the buffer); the model zeroes only for set_to_none=False
        sync = getattr(self.model, "grad_sync", None)
        if sync is not None and hasattr(sync, "finish_deferred"):
            sync.finish_deferred()  # a step that was skipped must not zero under a reduction in flight
        self.model.zero_grad(set_to_none=set_to_none)

    def state_dict(self):
        self._ensure_state_views()
        return super().state_dict()

    def load_state_dict(self, state_dict) -> None:
        self._ensure_state_views()
        super().load_state_dict(state_dict)
        m = self.model
        step = 0
        for p, name, rows in m._param_src:  # re-home the loaded moments in the flat buffers
            st = self.state[p]
            ea, es = m._view(name, rows, self._exp_avg), m._view(name, rows, self._exp_avg_sq)
            ea.copy_(st["exp_avg"])
            es.copy_(st["exp_avg_sq"])
            st["exp_avg"], st["exp_avg_sq"] = ea, es
            step = int(float(st["step"]))
            st["step"] = torch.tensor(float(step))
        self._step_count = step


def _to_container(node):
    try:
        from .config import DictConfig, OmegaConf
        if isinstance(node, DictConfig):
            return OmegaConf.to_container(node, resolve=True)
    except Exception:  # pragma: no cover
        pass
    try:
        from omegaconf import OmegaConf as _OC  # type: ignore
        return _OC.to_container(node, resolve=True)
    except Exception:
        return dict(node)


def setup_optimizer(cfg, model, optimizer_state_dict: dict[str, Any] | None = None) -> torch.optim.Optimizer:
    optimizer_kwargs = _to_container(cfg.optimizer)
    if hasattr(model, "_flat"):
        optimizer = HipAdamW(model.parameters(), model=model, **optimizer_kwargs)
    else:  # foreign module (tests with stand-in models): the reference's own choice
        optimizer_kwargs.pop("fused", None)
        optimizer = torch.optim.AdamW(model.parameters(), **optimizer_kwargs)
    if optimizer_state_dict is not None:
        optimizer.load_state_dict(optimizer_state_dict)
    return optimizer


def _hip_optimizer_of(model):
    """The live :class:`HipAdamW` built on ``model`` (it registers itself), or None when the caller brought its own optimizer."""
    ref = getattr(model, "_hip_optimizer", None)
    return ref() if ref is not None else None


def scale_grads(model, scaler: Tensor | float) -> None:
    """torchtune ``training.scale_grads`` (``trainer.py:404``): ``p.grad *= scaler`` for every parameter.  On the HIP
    decoder the multiplication is deferred: it is folded into the optimizer kernel (and into the clip norm), saving a
    full read+write pass over the 2.5 GB gradient buffer.  Without a :class:`HipAdamW` on the model (INTEGRATION.md level 2: the
    caller's own ``torch.optim`` optimizer on the parameter views) nothing would ever consume a deferred factor, so the
    multiplication happens at once, in place, by the HIP ``scale`` kernel."""
    if hasattr(model, "_flat_grad"):
        # a host scalar becomes a device scalar by a FILL kernel, not by an asynchronous copy out of a temporary host tensor (whose memory the
        # allocator may hand out again while the copy is still queued: round 5 saw one bench.py run in eight end on a slightly different loss)
        dev = model._flat_grad.device
        if torch.is_tensor(scaler) and scaler.is_cuda:
            s = scaler.to(dev, torch.float32).reshape(1)
        else:
            s = torch.full((1,), float(scaler), dtype=torch.float32, device=dev)
        if _hip_optimizer_of(model) is None:  # a foreign optimizer reads p.grad itself: multiply now (one pass over the flat buffer)
            sync = getattr(model, "grad_sync", None)
            if sync is not None and hasattr(sync, "finish_deferred"):
                sync.finish_deferred()
            ops.scale_(model._flat_grad, scale_dev=s)
            return
        model.pending_grad_scale = s if model.pending_grad_scale is None else model.pending_grad_scale * s
        return
    for p in model.parameters():
        if p.grad is not None:
            p.grad *= scaler.to(p.grad.device) if isinstance(scaler, Tensor) else scaler


def clip_grad_norm_(model, max_norm: float) -> Tensor:
    """``torch.nn.utils.clip_grad_norm_`` (``trainer.py:405-408``) on the flat gradient buffer: L2 norm by the HIP
    two-stage reduction, clip coefficient ``min(1, max_norm / (norm + 1e-6))`` folded into the pending gradient scale.
    Returns the total norm (device tensor, of the scaled gradients)."""
    if not hasattr(model, "_flat_grad"):
        return torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=float(max_norm))
    sync = getattr(model, "grad_sync", None)
    if sync is not None and hasattr(sync, "finish_deferred"):
        sync.finish_deferred()  # the norm reads every gradient
    out = torch.empty(1, dtype=torch.float32, device=model._flat_grad.device)
    ops.sumsq(model._flat_grad, out)
    scale = model.pending_grad_scale if model.pending_grad_scale is not None else torch.ones_like(out)
    total_norm = out.sqrt() * scale.abs()
    coef = torch.clamp(float(max_norm) / (total_norm + 1e-6), max=1.0)
    if _hip_optimizer_of(model) is None:  # nobody will consume a deferred factor (scale is 1 here: scale_grads was eager too)
        ops.scale_(model._flat_grad, scale_dev=(scale * coef).reshape(1).contiguous())
        model.pending_grad_scale = None
    else:
        model.pending_grad_scale = scale * coef
    return total_norm.reshape(())
