"""ctypes binding of ``libssi_hip.so`` (C ABI declared in ``include/ssi_hip.h``).

The product path has NO CPU fallback: if the shared library is missing or a GPU entry is called on a non-GPU tensor,
this module raises.  Build the library with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C speech-integration_amd/csrc``.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSI_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libssi_hip.so")  # override: diagnostic builds

SSI_F32, SSI_BF16 = 0, 1
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
IMPL_AUTO, IMPL_GENERIC, IMPL_MFMA, IMPL_MFMA_WG8 = 0, 1, 2, 3
TILES_STATIC, TILES_DYNAMIC = 0, 1
ABI_VERSION = 8
ATTN_KERNEL_DQ, ATTN_KERNEL_DKV = 0, 1
ATTN_MODE_AUTO, ATTN_MODE_OLD, ATTN_MODE_NEW, ATTN_MODE_NO_HEAD_SPLIT = 0, 1, 2, 3
ATTN_USED_DQ2, ATTN_USED_DKV2, ATTN_USED_HEAD_SPLIT, ATTN_USED_PLAN = 1, 2, 4, 8
ATTN_PLAN_HEADER, ATTN_PLAN_FORCE, ATTN_PLAN_SPLIT_ALL = 16, 1, 2

# name -> (restype, argtypes); mirrors include/ssi_hip.h line by line
_P = c_void_p
PROTOTYPES = {
    "ssi_abi_version": (c_int, []),
    "ssi_last_error": (c_char_p, []),
    "ssi_set_impl": (c_int, [c_int]),
    "ssi_set_attn_impl": (c_int, [c_int, c_int]),
    "ssi_attn_last_dispatch": (c_int, []),
    "ssi_embed_fwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, _P]),
    "ssi_embed_bwd_workspace_bytes": (c_int64, [c_int64]),
    "ssi_embed_bwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, _P, c_int64, _P]),
    "ssi_rmsnorm_fwd": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_float, c_int, _P]),
    "ssi_rmsnorm_bwd_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "ssi_rmsnorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int64, c_int64, c_int, _P, c_int64, _P]),
    "ssi_rope_inplace": (c_int, [_P, c_int64, c_int64, c_int64, c_int, c_int, _P, c_int64, _P, c_int, c_int, _P]),
    "ssi_gemm_rope": (c_int, [c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, _P, c_int64, _P,
                              c_int, _P]),
    "ssi_attn_fwd": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "ssi_attn_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "ssi_set_gemm_tile_order": (c_int, [c_int]),
    "ssi_attn_varlen_fwd": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "ssi_attn_varlen_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, c_int, _P]),
    "ssi_attn_varlen_bwd_rope": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, c_int,
                                         c_int, _P]),
    "ssi_attn_bwd_workspace_bytes": (c_int64, [c_int64, c_int64, c_int, c_int, c_int, c_int]),
    "ssi_attn_varlen_bwd_ws": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, c_int,
                                       c_int, _P, c_int64, _P]),
    "ssi_attn_plan_words": (c_int64, [c_int64, c_int64, c_int64]),
    "ssi_attn_plan_workspace_bytes": (c_int64, [_P]),
    "ssi_attn_plan_build": (c_int64, [_P, _P, _P, c_int64, c_int64, c_int64, c_int, c_int, c_int, _P, c_int64]),
    "ssi_attn_varlen_bwd_plan": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int, c_int, c_int,
                                         c_int, _P, c_int64, _P, _P, _P]),
    "ssi_doc_ranges": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P]),
    "ssi_swiglu_fwd": (c_int, [_P, _P, c_int64, c_int64, c_int, _P]),
    "ssi_swiglu_bwd": (c_int, [_P, _P, _P, c_int64, c_int64, c_int, _P]),
    "ssi_gemm": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_float, _P,
                         c_int, c_int, _P]),
    "ssi_gemm_splitk_workspace_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "ssi_gemm_splitk": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_float, _P,
                                c_int, c_int, c_int, _P, c_int64, _P]),
    "ssi_gemm_batched": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int64, c_int64,
                                 c_float, _P, c_int, c_int, _P]),
    "ssi_gemm_swiglu_fwd": (c_int, [c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int, _P]),
    "ssi_gemm_swiglu_bwd": (c_int, [c_int, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, _P, c_int,
                                    _P]),
    "ssi_transpose": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, c_int, _P]),
    "ssi_ce_fwd": (c_int, [_P, c_int64, _P, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, _P]),
    "ssi_ce_fwd_weighted": (c_int, [_P, c_int64, _P, _P, c_int64, c_int64, c_int64, _P, _P, c_int, c_int, _P]),
    "ssi_ce_reduce": (c_int, [_P, _P, c_int64, c_int64, c_int64, _P, _P]),
    "ssi_lmhead_ce_fwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P, _P, c_int, c_int, _P]),
    "ssi_lmhead_ce_bwd": (c_int, [_P, c_int64, _P, c_int64, _P, c_int64, _P, c_int64, c_int64, c_int64, _P, c_int64, _P, c_int64, c_int, c_int, _P]),
    "ssi_count_tokens": (c_int, [_P, _P, c_int64, _P, c_int, c_int64, c_int64, _P, _P]),
    "ssi_scale_inplace": (c_int, [_P, c_int64, c_float, _P, c_int, _P]),
    "ssi_sumsq_workspace_bytes": (c_int64, [c_int64]),
    "ssi_sumsq": (c_int, [_P, c_int64, c_int, _P, _P, c_int64, _P]),
    "ssi_adamw_step": (c_int, [_P, _P, _P, _P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int64, _P, c_int,
                               c_int, _P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load (once) and type the shared library.  Raises ``HipLibraryError`` if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). Build it with "
            f"`make -C {os.path.join(os.path.dirname(_HERE), 'csrc')}` or `__graft_entry__.build()`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype, fn.argtypes = res, args
    v = lib.ssi_abi_version()
    if v != ABI_VERSION:
        raise HipLibraryError(f"libssi_hip ABI version {v} != expected {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ssi_last_error()
        raise RuntimeError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")


def dtype_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return SSI_F32
    if dtype == torch.bfloat16:
        return SSI_BF16
    raise TypeError(f"unsupported dtype {dtype}; supported: torch.float32, torch.bfloat16")


def ptr(t: torch.Tensor | None) -> int | None:
    if t is None:
        return None
    if not t.is_cuda:
        raise HipLibraryError("libssi_hip kernels need GPU tensors (there is no CPU fallback in the product path)")
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_CURRENT_DEVICE = getattr(torch._C, "_cuda_getDevice", None)
_USE_RAW_STREAM = os.environ.get("SSI_RAW_STREAM", "1") != "0"  # (0: torch.cuda.current_stream() per launch, for in-run comparisons)


def stream_ptr() -> int:
    """The current HIP stream of the current device as an integer.  ``torch.cuda.current_stream()`` builds a Python Stream object behind a
    device-index lookup — 20 us per call, once per launch: 12 of the 20 ms a backward of ~600 launches took on the host (cProfile of the trainer's
    loop at 2 x 2048, ``profiles/LAB_NOTES.md`` round 5); the raw getter (what torch's own compiled code calls) is a C call."""
    if _RAW_STREAM is not None and _CURRENT_DEVICE is not None and _USE_RAW_STREAM:
        return _RAW_STREAM(_CURRENT_DEVICE())
    return torch.cuda.current_stream().cuda_stream
