"""Recipe shared by the full-size parity tests, the committed config-A fixture and the script that made it (tests/golden/make_config_a.py):
the seeded weights, the batch, and the per-parameter gradient SKETCH that lets a 2.5 MB file stand in for 5 GB of reference gradients.

Why a fixture: the fp32 CPU oracle (oracle/llama_oracle.py) on BASELINE config A's batch (B = 8, S = 2048, full 1B model) takes 219 s of
the GPU box's 16 host cores — more than half of the `pytest -m gpu` limit for one test.  The oracle side of that test now runs ONCE, by
the committed script; the GPU test still runs the HIP model at B = 8, S = 2048 and compares against what the script stored.

Sketch: a count-sketch of the flattened gradient into 4096 buckets, bucket and sign of element i from a multiplicative hash of i.  For
any two tensors, ||sketch(a) - sketch(b)||^2 is an unbiased estimate of ||a - b||^2 with relative standard deviation sqrt(2 / 4096) = 2.2 %,
so `sketch_rel_error` reproduces the tests' per-parameter relative gradient error ||g - g_ref|| / ||g_ref|| to a few percent of itself."""
import copy
import hashlib

import torch

SKETCH_BUCKETS = 4096
WEIGHT_SEED = 2024
BATCH_SEED = 42_831
NAMED = ("tok_embeddings.weight", "layers.0.attn.q_proj.weight", "layers.7.attn.k_proj.weight", "layers.15.mlp.w2.weight", "norm.scale")


def full_config(n_dsus=5000):
    from ssi.llama_configs import configllama3_2_1b
    cfg = copy.deepcopy(configllama3_2_1b)
    cfg.n_dsus, cfg.modality_tokens = n_dsus, True
    return cfg


def seeded_full_state_dict(params, seed, only_first: int = 0):
    """N(0, 0.02^2) weights, norm scales 1 + 0.1 N(0,1), every value rounded to bf16 so that the fp32 oracle, the fp32 HIP model and
    the bf16 HIP model hold bit-identical weights (what differs is then only the arithmetic under test).  ``only_first``: stop after that
    many tensors of the (fixed) state-dict order — the generator stream is sequential, so a prefix is reproducible on its own."""
    from oracle.llama_oracle import OracleLlama
    with torch.device("meta"):
        shapes = {k: tuple(v.shape) for k, v in OracleLlama(**params, rope_cache_len=8).state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in shapes.items():
        t = torch.randn(shape, generator=g)
        t = (1.0 + 0.1 * t) if name.endswith("scale") else 0.02 * t
        sd[name] = t.bfloat16().float()
        if only_first and len(sd) >= only_first:
            break
    return sd


def digest(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def sketch(t: torch.Tensor, buckets: int = SKETCH_BUCKETS, chunk: int = 1 << 24) -> torch.Tensor:
    """Count-sketch of ``t`` (any shape, any device) as float64 [buckets]; same result on CPU and GPU up to the order of the fp64 sums."""
    flat = t.detach().reshape(-1)
    out = torch.zeros(buckets, dtype=torch.float64, device=flat.device)
    shift = 32 - (buckets.bit_length() - 1)
    for lo in range(0, flat.numel(), chunk):
        x = flat[lo:lo + chunk].to(torch.float64)
        i = torch.arange(lo, lo + x.numel(), dtype=torch.int64, device=flat.device)
        h = (i * 2654435761) & 0xFFFFFFFF
        sign = 1.0 - 2.0 * ((h >> (shift - 1)) & 1).to(torch.float64)
        out.index_add_(0, h >> shift, x * sign)
    return out


def sketch_rel_error(got: torch.Tensor, want_sketch: torch.Tensor, want_norm: float) -> float:
    """Estimate of ||got - want|| / ||want|| from want's sketch and norm."""
    d = sketch(got).cpu() - want_sketch.to(torch.float64).cpu()
    return float(d.norm()) / float(want_norm)
