"""Tensor-level wrappers over the C ABI (``include/ssi_hip.h``).  Thin by design: shape checks on the host, raw device
pointers and the current HIP stream across the ABI, no torch compute ops.  Every function raises on failure."""

from __future__ import annotations

import torch
from typing import Optional
from torch import Tensor

from . import _lib
from ._lib import GEMM_NN, GEMM_NT, GEMM_TN, check, dtype_code, ptr, stream_ptr

__all__ = ["lmhead_ce_fwd", "lmhead_ce_bwd", "doc_ranges", "embed_fwd", "embed_bwd", "rmsnorm_fwd", "rmsnorm_bwd", "rope_", "attn_fwd", "attn_bwd", "attn_bwd_workspace_bytes", "swiglu_fwd",
           "swiglu_bwd", "gemm", "gemm_splitk", "splitk_choice", "gemm_swiglu_fwd", "gemm_swiglu_bwd", "transpose", "ce_fwd", "ce_reduce", "count_tokens", "scale_", "sumsq", "adamw_step", "set_impl", "set_attn_impl", "attn_last_dispatch",
           "GEMM_NT", "GEMM_NN", "GEMM_TN"]


def set_impl(impl: int) -> int:
    return _lib.load().ssi_set_impl(impl)


def set_attn_impl(which: int, mode: int) -> int:
    """Which attention backward kernels may run (``_lib.ATTN_KERNEL_*`` x ``_lib.ATTN_MODE_*``; process-global); returns the previous mode."""
    return _lib.load().ssi_set_attn_impl(which, mode)


def attn_last_dispatch() -> int:
    """Bit set of ``_lib.ATTN_USED_*``: the kernels the most recent MFMA attention backward of the process launched."""
    return _lib.load().ssi_attn_last_dispatch()


def _byte_ws(nbytes: int, like: Tensor) -> Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


def embed_fwd(tokens: Tensor, table: Tensor, out: Tensor, vocab: int) -> None:
    n, dim = tokens.numel(), table.shape[1]
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and out.is_contiguous() and table.stride(1) == 1
    assert table.stride(0) == dim and out.shape[-1] == dim and out.numel() == n * dim and out.dtype == table.dtype
    check(_lib.load().ssi_embed_fwd(ptr(tokens), ptr(table), ptr(out), n, dim, vocab, dtype_code(table.dtype),
                                    stream_ptr()), "ssi_embed_fwd")


def embed_bwd(tokens: Tensor, dout: Tensor, dtable: Tensor, vocab: int, workspace: Tensor | None = None) -> None:
    lib = _lib.load()
    n, dim = tokens.numel(), dtable.shape[1]
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and dout.is_contiguous() and dtable.stride(0) == dim
    need = lib.ssi_embed_bwd_workspace_bytes(vocab)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = _byte_ws(need, dout)
    check(lib.ssi_embed_bwd(ptr(tokens), ptr(dout), ptr(dtable), n, dim, vocab, dtype_code(dtable.dtype), ptr(workspace),
                            workspace.numel() * workspace.element_size(), stream_ptr()), "ssi_embed_bwd")


def rmsnorm_fwd(x: Tensor, scale: Tensor, y: Tensor, rstd: Tensor | None, eps: float) -> None:
    dim = x.shape[-1]
    rows = x.numel() // dim
    assert x.is_contiguous() and y.is_contiguous() and scale.is_contiguous() and scale.numel() == dim
    assert rstd is None or (rstd.dtype == torch.float32 and rstd.numel() >= rows)
    check(_lib.load().ssi_rmsnorm_fwd(ptr(x), ptr(scale), ptr(y), ptr(rstd), rows, dim, eps, dtype_code(x.dtype),
                                      stream_ptr()), "ssi_rmsnorm_fwd")


def rmsnorm_bwd_workspace_bytes(rows: int, dim: int) -> int:
    return _lib.load().ssi_rmsnorm_bwd_workspace_bytes(rows, dim)


def rmsnorm_bwd(dy: Tensor, x: Tensor, scale: Tensor, rstd: Tensor, dres: Tensor | None, dx: Tensor, dscale: Tensor,
                workspace: Tensor | None = None, accumulate: bool = True) -> None:
    lib = _lib.load()
    dim = x.shape[-1]
    rows = x.numel() // dim
    assert dy.is_contiguous() and x.is_contiguous() and dx.is_contiguous() and dscale.is_contiguous()
    assert dres is None or dres.is_contiguous()
    need = lib.ssi_rmsnorm_bwd_workspace_bytes(rows, dim)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = _byte_ws(need, dy)
    check(lib.ssi_rmsnorm_bwd(ptr(dy), ptr(x), ptr(scale), ptr(rstd), ptr(dres), ptr(dx), ptr(dscale), int(accumulate), rows, dim,
                              dtype_code(x.dtype), ptr(workspace), workspace.numel() * workspace.element_size(),
                              stream_ptr()), "ssi_rmsnorm_bwd")


def rope_(x: Tensor, seq_len: int, n_heads_rot: int, head_dim: int, table: Tensor, inverse: bool = False,
          positions: Tensor | None = None) -> None:
    """In-place rotation of the first ``n_heads_rot`` heads of each row of ``x`` ([rows, ld])."""
    assert x.dim() == 2 and x.stride(1) == 1 and table.dtype == torch.float32 and table.is_contiguous()
    assert positions is None or (positions.dtype == torch.int32 and positions.numel() == x.shape[0])
    check(_lib.load().ssi_rope_inplace(ptr(x), x.stride(0), x.shape[0], seq_len, n_heads_rot, head_dim, ptr(table),
                                       table.shape[0], ptr(positions), int(inverse), dtype_code(x.dtype), stream_ptr()),
          "ssi_rope_inplace")


def gemm_rope(a: Tensor, b: Tensor, c: Tensor, seq_len: int, n_heads_rot: int, head_dim: int, table: Tensor,
              positions: Tensor | None = None) -> None:
    """c = a @ b^T, then ``rope_`` on the first ``n_heads_rot`` heads of every row of c (QKV projection + RoPE, one launch on the
    MFMA path)."""
    assert a.dim() == 2 and b.dim() == 2 and c.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and c.shape == (M, N) and a.dtype == b.dtype == c.dtype
    assert table.dtype == torch.float32 and table.is_contiguous() and table.shape[1] * 2 == head_dim
    assert positions is None or (positions.dtype == torch.int32 and positions.numel() == M)
    check(_lib.load().ssi_gemm_rope(M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c), c.stride(0), seq_len, n_heads_rot, head_dim,
                                    ptr(table), table.shape[0], ptr(positions), dtype_code(a.dtype), stream_ptr()), "ssi_gemm_rope")


def _doc_ptrs(doc_start: Optional[Tensor], doc_end: Optional[Tensor], rows: int):
    if doc_start is None and doc_end is None:
        return None, None
    assert doc_start is not None and doc_end is not None, "doc_start and doc_end go together"
    for t in (doc_start, doc_end):
        assert t.dtype == torch.int32 and t.is_contiguous() and t.numel() == rows
    return ptr(doc_start), ptr(doc_end)


def doc_ranges(input_pos: Tensor, max_pos: int, n_clamped: Tensor | None = None) -> tuple[Tensor, Tensor, Tensor]:
    """Packed rows: int32 [B*S] (positions clamped to max_pos, doc_start, doc_end) from input_pos [B, S] in one launch.  ``n_clamped``
    (int32 [1], zeroed by the caller) receives the number of positions that had to be clamped."""
    assert input_pos.dim() == 2 and input_pos.dtype == torch.int64
    assert n_clamped is None or (n_clamped.dtype == torch.int32 and n_clamped.numel() == 1)
    ip = input_pos.contiguous()
    B, S = ip.shape
    out = torch.empty(3, B * S, dtype=torch.int32, device=ip.device)
    check(_lib.load().ssi_doc_ranges(ptr(ip), B, S, int(max_pos), ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(n_clamped), stream_ptr()),
          "ssi_doc_ranges")
    return out[0], out[1], out[2]


def set_gemm_tile_order(dynamic: bool) -> None:
    """Dynamic tile order for the persistent GEMMs (use when other kernels share the GPU, e.g. RCCL during backward)."""
    check(_lib.load().ssi_set_gemm_tile_order(_lib.TILES_DYNAMIC if dynamic else _lib.TILES_STATIC), "ssi_set_gemm_tile_order")


def attn_fwd(qkv: Tensor, out: Tensor, lse: Tensor, batch: int, seq: int, n_heads: int, n_kv: int, head_dim: int,
             doc_start: Optional[Tensor] = None, doc_end: Optional[Tensor] = None) -> None:
    """Causal GQA attention; with doc_start / doc_end (int32 [batch*seq]) block-causal over the documents packed in a row."""
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and out.is_contiguous() and lse.dtype == torch.float32
    assert qkv.shape[0] == batch * seq and out.numel() == batch * seq * n_heads * head_dim
    assert lse.numel() >= batch * n_heads * seq
    ds, de = _doc_ptrs(doc_start, doc_end, batch * seq)
    check(_lib.load().ssi_attn_varlen_fwd(ptr(qkv), qkv.stride(0), ptr(out), ptr(lse), ds, de, batch, seq, n_heads, n_kv, head_dim,
                                          dtype_code(qkv.dtype), stream_ptr()), "ssi_attn_varlen_fwd")


def attn_bwd_workspace_bytes(batch: int, seq: int, n_heads: int, n_kv: int, head_dim: int, dtype: torch.dtype) -> int:
    """Bytes of workspace ``attn_bwd`` can use for this shape (0: none): small launches then run dK / dV per query head + a reduction."""
    return int(_lib.load().ssi_attn_bwd_workspace_bytes(batch, seq, n_heads, n_kv, head_dim, dtype_code(dtype)))


def attn_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, dqkv: Tensor, delta: Tensor, batch: int, seq: int,
             n_heads: int, n_kv: int, head_dim: int, doc_start: Optional[Tensor] = None, doc_end: Optional[Tensor] = None,
             rope_table: Optional[Tensor] = None, positions: Optional[Tensor] = None, workspace: Optional[Tensor] = None, plan=None) -> None:
    """dqkv of causal (or block-causal) GQA attention.  With ``rope_table`` the q / k parts come back in pre-RoPE space (the
    backward of ``rope_`` fused in), positions as in ``rope_``.  ``workspace`` (bytes, see ``attn_bwd_workspace_bytes``): optional.
    ``plan`` (``ssi.attn_plan.AttnPlan`` on the device, packed rows only): the pipelined kernels take the documents' work from it."""
    assert qkv.stride(1) == 1 and dqkv.stride() == qkv.stride() and out.is_contiguous() and dout.is_contiguous()
    assert delta.dtype == torch.float32 and delta.numel() >= batch * n_heads * seq
    ds, de = _doc_ptrs(doc_start, doc_end, batch * seq)
    if workspace is not None or plan is not None:
        assert workspace is None or (workspace.is_contiguous() and workspace.dtype == torch.uint8)
        assert rope_table is None or (rope_table.dtype == torch.float32 and rope_table.is_contiguous() and rope_table.shape[1] * 2 == head_dim)
        assert positions is None or (positions.dtype == torch.int32 and positions.numel() == batch * seq)
        ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
        table_len = rope_table.shape[0] if rope_table is not None else 0
        if plan is not None:
            assert plan.dev is not None and plan.dev.device == qkv.device and plan.matches(batch, seq, n_heads, n_kv)
            assert ds is not None or (positions is None and int(plan.host[11]) == batch), "a plan for plain rows has one document per row"
            if ws_bytes < plan.workspace_bytes:  # fp32 partial rows of the dK/dV chunks the plan splits over the query heads
                workspace = _byte_ws(plan.workspace_bytes, qkv)
                ws_bytes = workspace.numel()
            check(_lib.load().ssi_attn_varlen_bwd_plan(ptr(qkv), qkv.stride(0), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), ds, de,
                                                       ptr(rope_table), table_len, ptr(positions), batch, seq, n_heads, n_kv, head_dim,
                                                       dtype_code(qkv.dtype), ptr(workspace), ws_bytes, ptr(plan.dev), plan.host.data_ptr(),
                                                       stream_ptr()), "ssi_attn_varlen_bwd_plan")
            return
        check(_lib.load().ssi_attn_varlen_bwd_ws(ptr(qkv), qkv.stride(0), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), ds, de,
                                                 ptr(rope_table), table_len, ptr(positions), batch,
                                                 seq, n_heads, n_kv, head_dim, dtype_code(qkv.dtype), ptr(workspace), ws_bytes, stream_ptr()),
              "ssi_attn_varlen_bwd_ws")
        return
    if rope_table is None:
        check(_lib.load().ssi_attn_varlen_bwd(ptr(qkv), qkv.stride(0), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), ds, de, batch,
                                              seq, n_heads, n_kv, head_dim, dtype_code(qkv.dtype), stream_ptr()), "ssi_attn_varlen_bwd")
        return
    assert rope_table.dtype == torch.float32 and rope_table.is_contiguous() and rope_table.shape[1] * 2 == head_dim
    assert positions is None or (positions.dtype == torch.int32 and positions.numel() == batch * seq)
    check(_lib.load().ssi_attn_varlen_bwd_rope(ptr(qkv), qkv.stride(0), ptr(out), ptr(dout), ptr(lse), ptr(dqkv), ptr(delta), ds, de,
                                               ptr(rope_table), rope_table.shape[0], ptr(positions), batch, seq, n_heads, n_kv, head_dim,
                                               dtype_code(qkv.dtype), stream_ptr()), "ssi_attn_varlen_bwd_rope")


def swiglu_fwd(gu: Tensor, act: Tensor) -> None:
    inter = act.shape[-1]
    rows = act.numel() // inter
    assert gu.is_contiguous() and act.is_contiguous() and gu.shape[-1] == 2 * inter
    check(_lib.load().ssi_swiglu_fwd(ptr(gu), ptr(act), rows, inter, dtype_code(gu.dtype), stream_ptr()), "ssi_swiglu_fwd")


def swiglu_bwd(dact: Tensor, gu: Tensor, dgu: Tensor) -> None:
    inter = dact.shape[-1]
    rows = dact.numel() // inter
    assert gu.is_contiguous() and dact.is_contiguous() and dgu.is_contiguous() and gu.shape[-1] == 2 * inter
    check(_lib.load().ssi_swiglu_bwd(ptr(dact), ptr(gu), ptr(dgu), rows, inter, dtype_code(gu.dtype), stream_ptr()),
          "ssi_swiglu_bwd")


def gemm(layout: int, a: Tensor, b: Tensor, c: Tensor, *, residual: Tensor | None = None, alpha: float = 1.0,
         alpha_dev: Tensor | None = None, accumulate: bool = False) -> None:
    """c = (accumulate ? c : 0) + alpha * [*alpha_dev] * op(a) op(b) (+ residual).  2-D row-major operands."""
    assert a.dim() == 2 and b.dim() == 2 and c.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1
    assert a.dtype == b.dtype == c.dtype
    M, N = c.shape
    if layout == GEMM_NT:
        K = a.shape[1]
        assert a.shape[0] == M and b.shape == (N, K)
    elif layout == GEMM_NN:
        K = a.shape[1]
        assert a.shape[0] == M and b.shape == (K, N)
    else:
        K = a.shape[0]
        assert a.shape[1] == M and b.shape == (K, N)
    if residual is not None:
        assert residual.shape == c.shape and residual.stride() == c.stride() and residual.dtype == c.dtype
    if alpha_dev is not None:
        assert alpha_dev.dtype == torch.float32 and alpha_dev.numel() == 1
    check(_lib.load().ssi_gemm(layout, M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c), c.stride(0),
                               ptr(residual), alpha, ptr(alpha_dev), int(accumulate), dtype_code(c.dtype), stream_ptr()),
          "ssi_gemm")


def splitk_choice(M: int, N: int, K: int, n_cu: int = 256) -> int:
    """Number of K-slices for a 256x256-tiled MFMA GEMM: minimise (waves of workgroups) x (K-tiles per slice) plus the slab
    write + read traffic; 1 when the output grid already fills the chip."""
    if M % 256 or N % 256 or K % 64:
        return 1
    tiles, nk = (M // 256) * (N // 256), K // 64
    if tiles >= 4 * n_cu or tiles % n_cu == 0:
        return 1  # whole rounds, or so many rounds that the partial last one is a small share
    best, best_t = 1, None
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        if s > nk:
            break
        waves = -(-tiles * s // n_cu)
        t = waves * -(-nk // s) * 1.8e-6 + ((s + 1) * M * N * 4 / 4.0e12 if s > 1 else 0.0)
        if best_t is None or t < best_t * 0.97:
            best, best_t = s, t
    return best


def gemm_splitk(layout: int, a: Tensor, b: Tensor, c: Tensor, splits: int, workspace: Tensor, *, alpha: float = 1.0,
                alpha_dev: Tensor | None = None, accumulate: bool = False, residual: Tensor | None = None) -> None:
    if splits <= 1:
        return gemm(layout, a, b, c, alpha=alpha, alpha_dev=alpha_dev, accumulate=accumulate, residual=residual)
    M, N = c.shape
    K = a.shape[1] if layout in (GEMM_NT, GEMM_NN) else a.shape[0]
    assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and a.dtype == b.dtype == c.dtype
    if residual is not None:
        assert residual.shape == c.shape and residual.stride() == c.stride() and residual.dtype == c.dtype and not accumulate
    check(_lib.load().ssi_gemm_splitk(layout, M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(c), c.stride(0), ptr(residual),
                                      alpha, ptr(alpha_dev), int(accumulate), dtype_code(c.dtype), splits, ptr(workspace),
                                      workspace.numel() * workspace.element_size(), stream_ptr()), "ssi_gemm_splitk")


def gemm_batched(layout: int, a: Tensor, b: Tensor, c: Tensor, *, alpha: float = 1.0, alpha_dev: Tensor | None = None,
                 accumulate: bool = False) -> None:
    """c[i] = (accumulate ? c[i] : 0) + alpha * op(a[i]) op(b[i]) for every i of the leading dimension, ONE launch on the MFMA path.
    3-D operands with unit inner stride; the leading (batch) stride is free, so ``c`` may be a strided view of the flat gradient."""
    assert a.dim() == 3 and b.dim() == 3 and c.dim() == 3 and a.shape[0] == b.shape[0] == c.shape[0]
    assert a.stride(2) == 1 and b.stride(2) == 1 and c.stride(2) == 1 and a.dtype == b.dtype == c.dtype
    n, M, N = c.shape
    if layout == GEMM_NT:
        K = a.shape[2]
        assert a.shape[1] == M and b.shape[1:] == (N, K)
    elif layout == GEMM_NN:
        K = a.shape[2]
        assert a.shape[1] == M and b.shape[1:] == (K, N)
    else:
        K = a.shape[1]
        assert a.shape[2] == M and b.shape[1:] == (K, N)
    if alpha_dev is not None:
        assert alpha_dev.dtype == torch.float32 and alpha_dev.numel() == 1
    check(_lib.load().ssi_gemm_batched(layout, n, M, N, K, ptr(a), a.stride(1), a.stride(0), ptr(b), b.stride(1), b.stride(0), ptr(c),
                                       c.stride(1), c.stride(0), alpha, ptr(alpha_dev), int(accumulate), dtype_code(c.dtype), stream_ptr()),
          "ssi_gemm_batched")


def gemm_swiglu_fwd(x: Tensor, w13: Tensor, gu: Tensor, act: Tensor) -> None:
    """gu = x @ w13^T ([gate | up]); act = silu(gate) * up — one launch on the MFMA path."""
    M, K = x.shape
    inter = act.shape[1]
    assert w13.shape == (2 * inter, K) and gu.shape == (M, 2 * inter) and act.shape[0] == M
    assert x.stride(1) == 1 and w13.stride(1) == 1 and gu.is_contiguous() and act.is_contiguous()
    check(_lib.load().ssi_gemm_swiglu_fwd(M, inter, K, ptr(x), x.stride(0), ptr(w13), w13.stride(0), ptr(gu), gu.stride(0), ptr(act),
                                          act.stride(0), dtype_code(x.dtype), stream_ptr()), "ssi_gemm_swiglu_fwd")


def gemm_swiglu_bwd(layout: int, dy: Tensor, w2: Tensor, gu: Tensor, dgu: Tensor, dact_ws: Tensor | None) -> None:
    """dgu = swiglu_backward(dy @ W2, gu); w2 is [I, K] for GEMM_NT (transposed copy) or [K, I] for GEMM_NN."""
    M, K = dy.shape
    inter = gu.shape[1] // 2
    assert w2.shape == ((inter, K) if layout == GEMM_NT else (K, inter)) and dgu.shape == gu.shape and gu.shape[0] == M
    assert dy.stride(1) == 1 and w2.stride(1) == 1 and gu.is_contiguous() and dgu.is_contiguous()
    check(_lib.load().ssi_gemm_swiglu_bwd(layout, M, inter, K, ptr(dy), dy.stride(0), ptr(w2), w2.stride(0), ptr(gu), gu.stride(0),
                                          ptr(dgu), dgu.stride(0), ptr(dact_ws), dtype_code(dy.dtype), stream_ptr()),
          "ssi_gemm_swiglu_bwd")


def transpose(src: Tensor, dst: Tensor) -> None:
    """dst[c, r] = src[r, c] (2-D, unit inner stride, dims multiples of 8)."""
    assert src.dim() == 2 and dst.dim() == 2 and src.stride(1) == 1 and dst.stride(1) == 1 and src.dtype == dst.dtype
    assert dst.shape == (src.shape[1], src.shape[0])
    check(_lib.load().ssi_transpose(ptr(src), src.stride(0), ptr(dst), dst.stride(0), src.shape[0], src.shape[1],
                                    dtype_code(src.dtype), stream_ptr()), "ssi_transpose")


def ce_fwd(logits: Tensor, labels: Tensor, vocab: int, ignore_index: int, row_loss: Tensor, row_lse: Tensor | None,
           write_grad: bool, row_weight: Tensor | None = None) -> None:
    """``row_weight`` (fp32, one per row, >= 0; ``ssi_ce_fwd_weighted``): loss and gradient of row r times ``row_weight[r]``."""
    assert logits.dim() == 2 and logits.stride(1) == 1 and labels.dtype == torch.int64 and labels.is_contiguous()
    rows = logits.shape[0]
    assert labels.numel() == rows and row_loss.dtype == torch.float32 and row_loss.numel() >= rows
    if row_weight is None:
        check(_lib.load().ssi_ce_fwd(ptr(logits), logits.stride(0), ptr(labels), rows, vocab, ignore_index, ptr(row_loss),
                                     ptr(row_lse), int(write_grad), dtype_code(logits.dtype), stream_ptr()), "ssi_ce_fwd")
        return
    assert row_weight.dtype == torch.float32 and row_weight.is_contiguous() and row_weight.numel() == rows and row_weight.device == logits.device
    check(_lib.load().ssi_ce_fwd_weighted(ptr(logits), logits.stride(0), ptr(labels), ptr(row_weight), rows, vocab, ignore_index, ptr(row_loss),
                                          ptr(row_lse), int(write_grad), dtype_code(logits.dtype), stream_ptr()), "ssi_ce_fwd_weighted")


def ce_reduce(row_loss: Tensor, labels: Tensor, vocab: int, ignore_index: int, out: Tensor) -> None:
    """out[0] mean NLL over valid labels, out[1] sum, out[2] valid count, out[3] labels that are neither ignored nor in [0, vocab)."""
    assert out.dtype == torch.float32 and out.numel() >= 4 and labels.is_contiguous()
    check(_lib.load().ssi_ce_reduce(ptr(row_loss), ptr(labels), labels.numel(), vocab, ignore_index, ptr(out), stream_ptr()),
          "ssi_ce_reduce")


def lmhead_ce_fwd(hidden: Tensor, table: Tensor, labels: Tensor, vocab: int, ignore_index: int, logits_ws: Tensor, row_loss: Tensor,
                  stats: Tensor, write_grad: bool) -> None:
    """Tied LM head + cross-entropy, one ABI call: logits_ws = hidden @ table^T, row losses, stats = (mean, sum, n_valid, n_out_of_range);
    with write_grad logits_ws holds softmax - onehot afterwards."""
    rows, dim = hidden.shape
    vocab_pad = table.shape[0]
    assert table.shape[1] == dim and logits_ws.shape == (rows, vocab_pad) and hidden.stride(1) == 1 and table.stride(1) == 1
    assert labels.dtype == torch.int64 and labels.is_contiguous() and labels.numel() == rows and stats.dtype == torch.float32 and stats.numel() >= 4
    assert row_loss.dtype == torch.float32 and row_loss.numel() >= rows and hidden.dtype == table.dtype == logits_ws.dtype
    check(_lib.load().ssi_lmhead_ce_fwd(ptr(hidden), hidden.stride(0), ptr(table), table.stride(0), ptr(labels), rows, dim, vocab, vocab_pad,
                                        ignore_index, ptr(logits_ws), logits_ws.stride(0), ptr(row_loss), ptr(stats), int(write_grad),
                                        dtype_code(hidden.dtype), stream_ptr()), "ssi_lmhead_ce_fwd")


def lmhead_ce_bwd(dlogits: Tensor, hidden: Tensor, table: Tensor, alpha_dev: Tensor | None, d_hidden: Tensor, d_table: Tensor,
                  accumulate_d_table: bool) -> None:
    """d_hidden = alpha * dlogits @ table;  d_table (+)= alpha * dlogits^T @ hidden."""
    rows, dim = hidden.shape
    vocab_pad = table.shape[0]
    assert dlogits.shape == (rows, vocab_pad) and d_hidden.shape == hidden.shape and d_table.shape == table.shape
    assert alpha_dev is None or (alpha_dev.dtype == torch.float32 and alpha_dev.numel() == 1)
    check(_lib.load().ssi_lmhead_ce_bwd(ptr(dlogits), dlogits.stride(0), ptr(hidden), hidden.stride(0), ptr(table), table.stride(0),
                                        ptr(alpha_dev), rows, dim, vocab_pad, ptr(d_hidden), d_hidden.stride(0), ptr(d_table),
                                        d_table.stride(0), int(accumulate_d_table), dtype_code(hidden.dtype), stream_ptr()), "ssi_lmhead_ce_bwd")


def count_tokens(tokens: Tensor, labels: Tensor | None, ranges: Tensor, pad_id: int, ignore_index: int, out: Tensor) -> None:
    """out[0:n_ranges] range counts, out[n_ranges] non-pad tokens, out[n_ranges+1] non-ignored labels (int64, device)."""
    n_ranges = ranges.numel() // 2
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and ranges.dtype == torch.int64 and out.dtype == torch.int64
    assert out.numel() >= n_ranges + 2 and (labels is None or (labels.is_contiguous() and labels.numel() == tokens.numel()))
    check(_lib.load().ssi_count_tokens(ptr(tokens), ptr(labels), tokens.numel(), ptr(ranges), n_ranges, pad_id,
                                       ignore_index, ptr(out), stream_ptr()), "ssi_count_tokens")


def scale_(x: Tensor, scale: float = 1.0, scale_dev: Tensor | None = None) -> None:
    assert x.is_contiguous()
    check(_lib.load().ssi_scale_inplace(ptr(x), x.numel(), scale, ptr(scale_dev), dtype_code(x.dtype), stream_ptr()),
          "ssi_scale_inplace")


def sumsq(x: Tensor, out: Tensor, workspace: Tensor | None = None) -> None:
    lib = _lib.load()
    assert x.is_contiguous() and out.dtype == torch.float32
    need = lib.ssi_sumsq_workspace_bytes(x.numel())
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = _byte_ws(need, x)
    check(lib.ssi_sumsq(ptr(x), x.numel(), dtype_code(x.dtype), ptr(out), ptr(workspace),
                        workspace.numel() * workspace.element_size(), stream_ptr()), "ssi_sumsq")


def adamw_step(param: Tensor, grad: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, *, lr: float, beta1: float, beta2: float,
               eps: float, weight_decay: float, step: int, grad_scale_dev: Tensor | None = None,
               zero_grad: bool = False, skip_nonfinite_scale: bool = False) -> None:
    """``skip_nonfinite_scale``: the launch changes nothing when ``*grad_scale_dev`` is inf or NaN (1 / 0 label tokens: the reference skips such
    a window's optimizer step; an update issued before the host knows the count must do the same by itself)."""
    assert param.is_contiguous() and grad.is_contiguous() and exp_avg.is_contiguous() and exp_avg_sq.is_contiguous()
    assert param.numel() == grad.numel() == exp_avg.numel() == exp_avg_sq.numel()
    assert param.dtype == grad.dtype == exp_avg.dtype == exp_avg_sq.dtype
    check(_lib.load().ssi_adamw_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, beta1, beta2,
                                     eps, weight_decay, step, ptr(grad_scale_dev), int(zero_grad) | (2 if skip_nonfinite_scale else 0), dtype_code(param.dtype),
                                     stream_ptr()), "ssi_adamw_step")
