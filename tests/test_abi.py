"""CPU: the C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/ssi_hip.h declares.
No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ssi_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssi_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_kernel_families():
    syms = _declared_symbols()
    for needed in ("ssi_embed_fwd", "ssi_embed_bwd", "ssi_rmsnorm_fwd", "ssi_rmsnorm_bwd", "ssi_rope_inplace", "ssi_attn_fwd",
                   "ssi_attn_bwd", "ssi_swiglu_fwd", "ssi_swiglu_bwd", "ssi_gemm", "ssi_ce_fwd", "ssi_ce_reduce",
                   "ssi_count_tokens", "ssi_adamw_step", "ssi_sumsq", "ssi_scale_inplace"):
        assert needed in syms


def test_library_loads_and_exports_every_declared_symbol():
    from ssi import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail(f"{_lib.LIB_PATH} missing: run __graft_entry__.build()")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"library does not export: {missing}"
    assert set(_lib.PROTOTYPES) == set(_declared_symbols()), "ctypes prototypes and header disagree"
    assert _lib.load().ssi_abi_version() == _lib.ABI_VERSION


def test_product_model_refuses_cpu():
    import torch
    from ssi import _lib
    from ssi.model import HipLlamaDecoder
    with pytest.raises(_lib.HipLibraryError):
        HipLlamaDecoder(vocab_size=64, num_layers=1, num_heads=2, num_kv_heads=1, embed_dim=32, max_seq_len=16,
                        intermediate_dim=64, dtype=torch.float32, device="cpu")


def test_ops_refuse_cpu_tensors():
    import torch
    from ssi import _lib, ops
    x = torch.zeros(4, 8)
    with pytest.raises(_lib.HipLibraryError):
        ops.rmsnorm_fwd(x, torch.ones(8), torch.empty_like(x), None, 1e-5)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "speech-integration_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_the_launch_path_reads_no_environment():
    """Process-global switches are atomics behind setters (``ssi_set_impl``, ``ssi_set_gemm_tile_order``, ``ssi_set_attn_impl``); the only
    ``getenv`` in the native sources is the one-time initialiser of the attention switches (C++ static initialisation), and the header lists
    them under "PROCESS-GLOBAL state"."""
    csrc = os.path.join(ROOT, "speech-integration_amd", "csrc")
    hits = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            for n, line in enumerate(open(os.path.join(csrc, f)), 1):
                if re.search(r"\bgetenv\s*\(", line) and not line.lstrip().startswith("//"):
                    hits.append((f, n, line.strip()))
    assert len(hits) == 1 and hits[0][0] == "attention_mfma.hip" and "const char* s = getenv(name)" in hits[0][2], hits
    src = open(os.path.join(csrc, "attention_mfma.hip")).read()
    assert re.search(r"static std::atomic<int> modes\[2\] = \{\{attn_env_mode\(", src), "the environment is read under static initialisation only"
    header = open(os.path.join(ROOT, "include", "ssi_hip.h")).read()
    assert "ssi_set_attn_impl" in header.split("#ifndef SSI_HIP_H")[0], "the header's PROCESS-GLOBAL list names the attention switches"
    from ssi import _lib
    lib = _lib.load()
    prev = lib.ssi_set_attn_impl(_lib.ATTN_KERNEL_DQ, _lib.ATTN_MODE_OLD)
    assert lib.ssi_set_attn_impl(_lib.ATTN_KERNEL_DQ, prev) == _lib.ATTN_MODE_OLD
    assert lib.ssi_set_attn_impl(_lib.ATTN_KERNEL_DQ, 7) == prev and lib.ssi_set_attn_impl(5, 0) == -1   # invalid mode only reads, invalid kernel -> -1
    assert lib.ssi_attn_last_dispatch() == 0
