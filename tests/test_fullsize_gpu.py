"""GPU parity at the sizes the headline number is measured at (BASELINE.json configs A, A', C, E, P), for the kernels bench.py times:

* the whole 1B model (16 layers, D = 2048, V = 133 258) against the CPU oracle on config P's batch (B = 2, S = 512): fp32 (generic
  kernels) and bf16 (MFMA GEMMs, split-K weight gradients, MFMA attention, fused LM head + cross-entropy) from the SAME weights;
* the persistent MFMA GEMM on the step's own shapes (LM head NT / NN / TN at N = 133 376 and 136 704, the split-K weight gradients
  of the square projections, the K = 16 384 weight gradients) — exact on small-integer operands;
* MFMA attention forward / backward at S = 4096 and S = 8192 with 32 query / 8 kv heads, plain causal and packed with 440-1100-token
  documents (the `block_to_work` map at 32 and 64 key groups), against torch SDPA in fp32 on the CPU.

Tolerances are written at each assert (north star: loss / logits within 1e-3 relative in fp32)."""
import copy
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"
# Per-parameter relative gradient error of the bf16 model vs the fp32 oracle.  The yardstick test below shows where it comes from: the
# oracle itself run with bf16 parameters (the reference's arithmetic on a bf16 model) is 4.77e-2 from its fp32 self on its worst
# parameter (layers.15.attn.k_proj.weight), the HIP model 4.69e-2 on the same one; worst HIP / oracle-bf16 ratio over the 146 parameters
# 1.04, median 0.99 (config P; at configs C / E / A the worst HIP errors are 4.41e-2 / 4.39e-2 / 4.56e-2).  Round 2 had 6e-2 here.
TOL_GRAD_BF16 = 5.5e-2


# ---------------------------------------------------------------------------------------------------------------------
# 1. Whole model, config P dimensions
# ---------------------------------------------------------------------------------------------------------------------
from fullsize_recipe import NAMED, full_config as _full_config, seeded_full_state_dict as _seeded_full_state_dict  # noqa: E402  (shared with the config-A fixture)


_REFERENCES = {}


def _full_size_reference(n_dsus):
    """Oracle forward + backward of the full model on config P's batch (about 15 s on the GPU box's 16 host cores); once per vocabulary."""
    if n_dsus in _REFERENCES:
        return _REFERENCES[n_dsus]
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.data import synthetic_batch
    cfg = _full_config(n_dsus)
    params = cfg.parameters
    sd = _seeded_full_state_dict(params, 2024)
    batch = synthetic_batch(2, 512, n_dsus, seed=42_831)
    with torch.device("meta"):
        ref = OracleLlama(**params, rope_cache_len=512)
    rope = ref.rope.clone()
    ref = ref.to_empty(device="cpu")
    ref.rope = rope
    ref.load_state_dict(sd)
    ref.set_num_output_chunks(8)
    loss = oracle_loss(batch, ref, OracleCEWithChunkedOutputLoss())
    loss.backward()
    rows = [(0, 0), (0, 17), (0, 255), (0, 511), (1, 1), (1, 300), (1, 448), (1, 510)]
    with torch.no_grad():
        hn = ref.forward_hidden(batch["tokens"])
        logit_rows = torch.stack([F.linear(hn[b, s], ref.tok_embeddings.weight) for b, s in rows])
    grads = {k: p.grad.clone() for k, p in ref.named_parameters()}
    n_shifted = int((torch.hstack((batch["labels"][..., 1:], torch.full_like(batch["labels"][..., -1:], -100))) != -100).sum())
    out = dict(cfg=cfg, params=params, sd=sd, batch=batch, loss=float(loss.detach()), rows=rows, logit_rows=logit_rows, grads=grads, n_shifted=n_shifted)
    del ref
    _REFERENCES.clear()   # one vocabulary at a time: a reference holds 10 GB of host memory
    _REFERENCES[n_dsus] = out
    return out


@pytest.mark.parametrize("dtype_name,n_dsus", [("bf16", 8192), ("fp32", 5000), ("bf16", 5000)],   # V = 133 258 last: the yardstick test below reuses its oracle pass
                         ids=["bf16-V136450-config-A-prime", "fp32-V133258", "bf16-V133258"])
def test_full_size_model_matches_the_cpu_oracle(dtype_name, n_dsus):
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    R = _full_size_reference(n_dsus)
    assert R["params"]["vocab_size"] == {5000: 133_258, 8192: 136_450}[n_dsus]
    dtype = torch.float32 if dtype_name == "fp32" else torch.bfloat16
    model = HipLlamaDecoder(**R["params"], dtype=dtype, device=DEV, rope_cache_len=512)
    model.load_state_dict(R["sd"])
    held = model.state_dict()
    for k in ("tok_embeddings.weight", "layers.7.attn.k_proj.weight", "layers.15.mlp.w3.weight", "norm.scale"):
        assert torch.equal(held[k].float().cpu(), R["sd"][k]), k   # bf16-representable weights: both dtypes hold them exactly
    model.set_num_output_chunks(8)
    assert model._mfma_shapes() == (dtype == torch.bfloat16)
    batch = {k: v.to(DEV) for k, v in R["batch"].items()}
    model.train()
    loss = compute_loss(batch, model, CEWithChunkedOutputLoss())
    loss.backward()
    rel = abs(loss.item() - R["loss"]) / abs(R["loss"])
    tol_loss, tol_logit, tol_grad = (1e-4, 1e-3, 2e-3) if dtype_name == "fp32" else (1e-2, 5e-2, TOL_GRAD_BF16)
    print(f"[full-size {dtype_name}] loss {loss.item():.6f} vs oracle {R['loss']:.6f}: rel {rel:.2e} (tolerance {tol_loss})")
    assert rel <= tol_loss
    worst, worst_key = 0.0, None
    for k, p in model.named_parameters():
        g, g_ref = p.grad.float().cpu(), R["grads"][k]
        err = float((g - g_ref).norm() / g_ref.norm())
        if err > worst:
            worst, worst_key = err, k
        assert err <= tol_grad, f"{k}: relative gradient error {err}"
        nrm = float(g.norm()) / float(g_ref.norm())
        assert abs(nrm - 1.0) <= tol_grad, f"{k}: gradient norm ratio {nrm}"
    print(f"[full-size {dtype_name}] worst relative gradient error {worst:.2e} ({worst_key})")
    # logits through the public forward (8 chunks), at sampled positions
    model.eval()
    with torch.no_grad():
        logits = torch.cat(model(tokens=batch["tokens"]), dim=1)
    got = torch.stack([logits[b, s].float().cpu() for b, s in R["rows"]])
    err = float((got - R["logit_rows"]).abs().max())
    scale = float(R["logit_rows"].abs().max())
    print(f"[full-size {dtype_name}] logits max-abs error {err:.3e} vs max |logit| {scale:.3f} (tolerance {tol_logit} x)")
    assert err <= tol_logit * scale
    assert int(torch.argmax(got, -1).eq(torch.argmax(R["logit_rows"], -1)).sum()) >= (8 if dtype_name == "fp32" else 6)


# ---------------------------------------------------------------------------------------------------------------------
# 1b. The whole 1B model against the CPU oracle at the OTHER BASELINE.json shapes: config C (S = 4096), config E (packed rows of 8192,
#     V = 130 306, the oracle fed torchtune's dense block-causal mask) and the headline batch itself, config A (B = 8, S = 2048; since round 4
#     with the oracle side in a committed fixture, tests/golden/config_a.npz).
#     What config P's 512-token rows cannot reach: RoPE positions > 2047, the 32- and 64-key-group attention work maps inside the model,
#     M = 8192 / 16 384-row tile walks of every GEMM, the 16-round head weight gradient, per-document positions at full width.
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_at(n_dsus, batch, rope_len, dtype=torch.float32):
    """Loss and every gradient of the CPU oracle (``oracle/llama_oracle.py``, restating ssi/loss.py:7-22) on ``batch``; the weights are
    ``_seeded_full_state_dict(params, 2024)`` (bf16-representable), cast to ``dtype`` for the rounding yardstick."""
    import time
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.data import packed_block_causal_mask
    cfg = _full_config(n_dsus)
    params = cfg.parameters
    sd = _seeded_full_state_dict(params, 2024)
    with torch.device("meta"):
        ref = OracleLlama(**params, rope_cache_len=rope_len)
    rope = ref.rope.clone()
    ref = ref.to_empty(device="cpu")
    ref.rope = rope
    ref.load_state_dict(sd)
    if dtype != torch.float32:
        for p in ref.parameters():   # parameters only: the RoPE table stays fp32 (torchtune calls .float() on it)
            p.data = p.data.to(dtype)
    ref.set_num_output_chunks(8)
    ref_batch = {k: v for k, v in batch.items() if k in ("tokens", "labels", "input_pos")}
    if "seq_lens" in batch:
        ref_batch["mask"] = packed_block_causal_mask(batch["seq_lens"])
    t0 = time.time()
    loss = oracle_loss(ref_batch, ref, OracleCEWithChunkedOutputLoss())
    loss.backward()
    grads = {k: p.grad.float().clone() for k, p in ref.named_parameters()}
    print(f"[oracle {dtype}] {batch['tokens'].numel()} tokens: forward + backward {time.time() - t0:.1f} s on {torch.get_num_threads()} threads")
    out = dict(cfg=cfg, params=params, sd=sd, loss=float(loss.detach()), grads=grads)
    del ref
    return out


def _hip_at(R, batch, rope_len, on_device=False, plain_kernels=False):
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    model = HipLlamaDecoder(**R["params"], dtype=torch.bfloat16, device=DEV, rope_cache_len=rope_len)
    model.load_state_dict(R["sd"])
    model.set_num_output_chunks(8)
    assert model._mfma_shapes()
    if plain_kernels:  # the unsplit GEMM forms (the attention kernels are switched by the caller: process-global)
        model.split_small_grids = False
    model.train()
    dbatch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in batch.items()}
    loss = compute_loss(dbatch, model, CEWithChunkedOutputLoss())
    loss.backward()
    grads = {k: (p.grad.float() if on_device else p.grad.float().cpu()) for k, p in model.named_parameters()}
    return float(loss.item()), grads


def _grad_errors(got, want):
    return {k: float((got[k] - want[k]).norm() / want[k].norm()) for k in want}


def test_full_size_model_matches_the_cpu_oracle_at_config_a_through_the_committed_fixture(golden_dir):
    """The headline batch itself (BASELINE config A: B = 8, S = 2048, V = 133 258; M = 16 384-row tile walks of every GEMM, the 16-round head
    weight gradient): the bf16 HIP model runs here, the fp32 CPU oracle ran ONCE — ``tests/golden/make_config_a.py``, 219 s of 16 host cores
    that used to sit inside this suite — and left ``tests/golden/config_a.npz``: loss, counts, and per parameter the gradient norm, a
    4096-bucket count-sketch (``fullsize_recipe.sketch``: relative gradient errors to ~2 % of themselves) and the first 4096 elements of five
    named parameters.  The recipe's digests are re-derived first, so the file and the inputs built here cannot drift apart."""
    import os

    import numpy as np
    import fullsize_recipe as fr
    from ssi.data import synthetic_batch
    g = np.load(os.path.join(golden_dir, "config_a.npz"))
    n_dsus, B, S = int(g["recipe_n_dsus"]), int(g["recipe_B"]), int(g["recipe_S"])
    assert (n_dsus, B, S) == (5000, 8, 2048) and int(g["recipe_weight_seed"]) == fr.WEIGHT_SEED and int(g["sketch_buckets"]) == fr.SKETCH_BUCKETS
    params = _full_config(n_dsus).parameters
    assert params["vocab_size"] == int(g["recipe_vocab"]) == 133_258
    sd = _seeded_full_state_dict(params, fr.WEIGHT_SEED)
    batch = synthetic_batch(B, S, n_dsus, seed=int(g["recipe_batch_seed"]))
    assert fr.digest(sd[str(g["recipe_first_tensor"])][:64]) == str(g["digest_first_tensor_rows"]), "seeded weights differ from the fixture's"
    assert fr.digest(batch["tokens"]) == str(g["digest_tokens"]) and fr.digest(batch["labels"]) == str(g["digest_labels"])
    assert int((batch["labels"] != -100).sum()) == int(g["n_unshifted"])
    loss, grads = _hip_at(dict(params=params, sd=sd), batch, S, on_device=True)
    want_loss = float(g["loss"])
    rel = abs(loss - want_loss) / abs(want_loss)
    names = [str(n) for n in g["names"]]
    assert names == list(grads) and len(names) == 146
    errs, worst = {}, None
    for i, k in enumerate(names):
        want_norm = float(g["norms"][i])
        errs[k] = fr.sketch_rel_error(grads[k], torch.from_numpy(g["sketches"][i]), want_norm)
        nrm = float(grads[k].double().norm()) / want_norm
        assert abs(nrm - 1.0) <= TOL_GRAD_BF16, f"{k}: gradient norm ratio {nrm}"
        assert errs[k] <= TOL_GRAD_BF16, f"{k}: relative gradient error {errs[k]} (count-sketch estimate)"
        if worst is None or errs[k] > errs[worst]:
            worst = k
    heads = {}
    for k in NAMED:   # and element for element where the fixture holds the elements
        want = torch.from_numpy(g["head/" + k])
        got = grads[k].reshape(-1)[:4096].float().cpu()
        heads[k] = float((got - want).norm() / want.norm())
        assert heads[k] <= 1.5 * TOL_GRAD_BF16, f"{k}: first 4096 elements {heads[k]}"
    print(f"[config A: B={B} S={S} V={params['vocab_size']}, oracle from the fixture ({float(g['oracle_seconds']):.0f} s on {int(g['oracle_threads'])} threads when made)] "
          f"loss {loss:.6f} vs oracle {want_loss:.6f}: rel {rel:.2e}; worst gradient error {errs[worst]:.2e} ({worst}); "
          + ", ".join(f"{k} {errs[k]:.2e} (head {heads[k]:.2e})" for k in NAMED))
    assert rel <= 1e-2
    # The fixture's count-sketch sees an error that is spread over a tensor better than one that sits in a few tiles, and this is the only
    # test at M = 16 384.  An exact cross-check that needs no oracle (round-4 advice): the same step on the round-1..3 attention backward
    # kernels and unsplit GEMM grids must give every one of the 146 gradients to 2e-2 (another summation order, 16 layers of bf16
    # re-rounding: measured 1.23e-2 on layers.0.attn.q_proj.weight, the tensor with the longest backward path behind it; a tile gone wrong
    # in one GEMM round would be O(1) on its tensor).
    from ssi import _lib, ops
    prev = [ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, _lib.ATTN_MODE_OLD), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, _lib.ATTN_MODE_OLD)]
    try:
        loss_old, grads_old = _hip_at(dict(params=params, sd=sd), batch, S, on_device=True, plain_kernels=True)
        assert ops.attn_last_dispatch() & (_lib.ATTN_USED_DQ2 | _lib.ATTN_USED_DKV2) == 0
    finally:
        ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, prev[0]), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, prev[1])
    cross = _grad_errors(grads, grads_old)
    worst_x = max(cross, key=cross.get)
    print(f"[config A: default kernels vs round-1..3 attention backward + unsplit GEMM grids] loss {loss:.6f} vs {loss_old:.6f}; worst gradient "
          f"difference {cross[worst_x]:.2e} ({worst_x})")
    assert abs(loss - loss_old) <= 1e-5 * abs(loss_old) and cross[worst_x] <= 2e-2, (worst_x, cross[worst_x])


@pytest.mark.parametrize("config,n_dsus,B,S,packed", [("C", 5000, 1, 4096, False), ("E", 2048, 1, 8192, True)],
                         ids=["config-C-S4096", "config-E-packed-S8192-V130306"])
def test_full_size_model_matches_the_cpu_oracle_at_the_baseline_shapes(config, n_dsus, B, S, packed):
    """bf16 HIP model (every MFMA kernel of the step) vs the fp32 CPU oracle from the same bf16-representable weights: loss within 1e-2
    (north star 1e-3 is the fp32 bar; observed is printed), every one of the 146 gradients within the bf16 tolerance of config P."""
    from ssi.data import synthetic_batch, synthetic_packed_batch
    batch = synthetic_packed_batch(B, S, n_dsus, seed=42_831) if packed else synthetic_batch(B, S, n_dsus, seed=42_831)
    if packed:
        assert int((batch["input_pos"] == 0).sum()) >= 8 and int(batch["input_pos"].max()) < S
    R = _oracle_at(n_dsus, batch, S)
    assert R["params"]["vocab_size"] == {5000: 133_258, 2048: 130_306}[n_dsus]
    loss, grads = _hip_at(R, batch, S)
    rel = abs(loss - R["loss"]) / abs(R["loss"])
    errs = _grad_errors(grads, R["grads"])
    worst = max(errs, key=errs.get)
    named = {k: errs[k] for k in NAMED}
    print(f"[config {config}: B={B} S={S} packed={packed} V={R['params']['vocab_size']}] loss {loss:.6f} vs oracle {R['loss']:.6f}: rel {rel:.2e}; "
          f"worst gradient error {errs[worst]:.2e} ({worst}); " + ", ".join(f"{k} {v:.2e}" for k, v in named.items()))
    assert rel <= 1e-2
    for k, e in errs.items():
        assert e <= TOL_GRAD_BF16, f"{k}: relative gradient error {e}"
        nrm = float(grads[k].norm() / R["grads"][k].norm())
        assert abs(nrm - 1.0) <= TOL_GRAD_BF16, f"{k}: gradient norm ratio {nrm}"


def test_bf16_gradient_error_is_the_reference_arithmetics_own_rounding():
    """Yardstick for the bf16 tolerance: the SAME oracle run once more with bf16 parameters on the CPU (what the reference computes on a
    bf16 model: bf16 GEMMs with fp32 accumulation, fp32 norm / RoPE / softmax / CE islands) against the fp32 oracle, per parameter, next to
    the HIP model's error against the fp32 oracle — config P's batch (B = 2, S = 512), full 1B model.  The HIP path may not be further from
    fp32 than 1.5 x the reference arithmetic's own bf16 rounding."""
    R = _full_size_reference(5000)   # the fp32 oracle pass on config P's batch: built once per session (cached from the tests above)
    batch = R["batch"]
    Rb = _oracle_at(5000, batch, 512, dtype=torch.bfloat16)
    loss, grads = _hip_at(R, batch, 512)
    e_hip, e_ref = _grad_errors(grads, R["grads"]), _grad_errors(Rb["grads"], R["grads"])
    ratio = {k: e_hip[k] / max(e_ref[k], 1e-12) for k in e_hip}
    worst_hip, worst_ref, worst_ratio = max(e_hip, key=e_hip.get), max(e_ref, key=e_ref.get), max(ratio, key=ratio.get)
    print(f"[bf16 yardstick] loss: HIP rel {abs(loss - R['loss']) / R['loss']:.2e}, oracle-bf16 rel {abs(Rb['loss'] - R['loss']) / R['loss']:.2e}")
    print(f"[bf16 yardstick] worst HIP gradient error {e_hip[worst_hip]:.2e} ({worst_hip}; oracle-bf16 there {e_ref[worst_hip]:.2e}); "
          f"worst oracle-bf16 error {e_ref[worst_ref]:.2e} ({worst_ref}; HIP there {e_hip[worst_ref]:.2e}); "
          f"worst ratio HIP / oracle-bf16 {ratio[worst_ratio]:.2f} ({worst_ratio}); median ratio {sorted(ratio.values())[len(ratio) // 2]:.2f}")
    for k in e_hip:
        assert e_hip[k] <= 1.5 * e_ref[k] + 1e-3, f"{k}: HIP {e_hip[k]:.3e} vs the reference arithmetic in bf16 {e_ref[k]:.3e}"


def test_full_width_model_on_ragged_padded_micro_batches_matches_the_cpu_oracle():
    """The reference's batch format at the model's real widths (D = 2048, I = 8192, 32 / 8 heads, V = 133 258; 2 layers to keep the oracle at
    seconds): right-padded rows of unequal length (``padded_collate_sft``: pad id / -100), a row whose labels are all ignored, sequence lengths
    that are multiples of nothing (333, then 777: padded to whole MFMA tiles inside the model), two micro-batches accumulated with the trainer's
    algebra (mean over SHIFTED labels x UNSHIFTED count) — bf16 HIP model against the fp32 oracle: per-micro-batch losses, every gradient."""
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from oracle.llama_oracle import compute_loss as oracle_loss
    from ssi.data import synthetic_batch
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    cfg = _full_config(5000)
    cfg.num_layers = 2
    params = cfg.parameters
    assert params["num_layers"] == 2 and params["embed_dim"] == 2048 and params["vocab_size"] == 133_258
    sd = _seeded_full_state_dict(params, 77)
    pad_id = 133_006
    batches = []
    for (B, S), seed in (((3, 333), 1), ((2, 777), 2)):
        b = synthetic_batch(B, S, 5000, seed=seed)
        lens = [S, S - 100, S // 3][:B]
        for r, n in enumerate(lens):      # right padding as the collate function does it
            b["tokens"][r, n:] = pad_id
            b["labels"][r, n:] = -100
        batches.append(b)
    batches[0]["labels"][2] = -100        # a row that contributes nothing
    batches[1]["labels"][:, 0] = batches[1]["tokens"][:, 0]   # CPT-style rows (labels = tokens, column 0 included): unshifted != shifted count
    ref = OracleLlama(**params, rope_cache_len=1024)
    ref.load_state_dict(sd)
    ref.set_num_output_chunks(8)
    model = HipLlamaDecoder(**params, dtype=torch.bfloat16, device=DEV, rope_cache_len=1024)
    model.load_state_dict(sd)
    model.set_num_output_chunks(8)
    model.train()
    assert model._mfma_shapes() and model.padded_seq_len(3, 333) != 333
    n_total = 0
    want_loss = []
    for b in batches:
        n = int((b["labels"] != -100).sum())
        n_total += n
        want = oracle_loss(b, ref, OracleCEWithChunkedOutputLoss())
        (want * n).backward()
        want_loss.append(want.item())
    # Round 4: the same micro-batches twice through the HIP model — as right-padded rows, and with the padding dropped on the host the way the
    # trainer's prefetch thread does it (ssi/data/unpad.py: rows end to end as one packed sequence, block-causal attention, per-row positions).
    # Both against the oracle on the PADDED batch, and against each other: the transform is exact, the two runs differ by the order of fp32
    # sums inside the attention kernels and the bf16 rounding behind them.
    from ssi.data import loss_inputs, unpad_batch
    grads, losses = {}, {}
    # Round 5: a third time with the work plan of the attention backward the prefetch thread builds beside the packed copy (ssi/attn_plan.py):
    # the packed rows then run the pipelined dQ / dK-dV kernels (asserted), document-aware forms.
    from ssi import _lib, ops
    to_dev = lambda v: v.to(DEV) if torch.is_tensor(v) else (v.to_device(DEV) if getattr(v, "is_attn_plan", False) else v)  # noqa: E731
    for how in ("padded", "unpadded", "unpadded, work plan"):
        model.zero_grad(set_to_none=True)
        losses[how] = []
        for b, want in zip(batches, want_loss):
            n = int((b["labels"] != -100).sum())
            plan_fn = (lambda ip: model.build_attn_plan(ip, force=True)) if how.endswith("plan") else None   # (forced: a handful of rows is not what the plan is for)
            hb = b if how == "padded" else unpad_batch(b, pad_id=pad_id, padded_len=model.padded_seq_len, plan_fn=plan_fn)
            assert ("packed_tokens" in hb) == (how != "padded") and ("packed_attn_plan" in hb) == how.endswith("plan")
            if how != "padded":
                assert hb["packed_tokens"].shape[1] < b["tokens"].shape[0] * model.padded_seq_len(*b["tokens"].shape)
            got = compute_loss(loss_inputs({k: to_dev(v) for k, v in hb.items()}), model, CEWithChunkedOutputLoss())
            (got * n).backward()
            want_bits = (_lib.ATTN_USED_DQ2 | _lib.ATTN_USED_DKV2 | _lib.ATTN_USED_PLAN) if how.endswith("plan") else 0
            assert ops.attn_last_dispatch() & _lib.ATTN_USED_PLAN == want_bits & _lib.ATTN_USED_PLAN and ops.attn_last_dispatch() & want_bits == want_bits
            rel = abs(got.item() - want) / abs(want)
            print(f"[ragged full-width, {how}] B x S = {tuple(b['tokens'].shape)}: loss {got.item():.6f} vs oracle {want:.6f} (rel {rel:.2e})")
            assert rel <= 1e-2
            losses[how].append(got.item())
        worst, worst_key = 0.0, None
        for (k, p), (_, p2) in zip(model.named_parameters(), ref.named_parameters()):
            err = float((p.grad.float().cpu() - p2.grad).norm() / p2.grad.norm())
            if err > worst:
                worst, worst_key = err, k
            assert err <= TOL_GRAD_BF16, f"{how} {k}: relative gradient error {err}"
        print(f"[ragged full-width, {how}] worst relative gradient error over two accumulated micro-batches {worst:.2e} ({worst_key}); {n_total} label tokens")
        pad_rows = model._view("emb", None, model._flat_grad)[params["vocab_size"]:]
        assert float(pad_rows.abs().max()) == 0.0      # the rows that pad the table to whole tiles never receive a gradient
        grads[how] = {k: p.grad.float().cpu().clone() for k, p in model.named_parameters()}
    # Round 5: the window's two micro-batches as ONE batch (ssi/data/window.py, the trainer's default): rows of both end to end, one forward /
    # backward.  The reference gives every micro-batch its own weight (unshifted / shifted label count: 1 for the first micro-batch, 1.0014 for
    # the CPT-style second) — carried per position, applied by the cross-entropy kernel per row.  Against the oracle's LOOP over the micro-batches.
    from ssi.data.window import WEIGHTS_KEY, fuse_micro_batches
    model.zero_grad(set_to_none=True)
    joined = fuse_micro_batches(batches, pad_id=pad_id, padded_len=model.padded_seq_len, plan_fn=lambda ip: model.build_attn_plan(ip, force=True))
    assert joined is not None and WEIGHTS_KEY in joined and "packed_attn_plan" in joined and joined["micro_batches"] == 2
    w = joined[WEIGHTS_KEY]
    assert len(set(w.flatten().tolist())) == 3 and 1e-4 < float(w.max() - w.min()) < 1e-2   # (two micro-batches + the tile tail's 1)
    n = int((joined["labels"] != -100).sum())
    assert n == n_total
    got = compute_loss(loss_inputs({k: to_dev(v) for k, v in joined.items()}), model, CEWithChunkedOutputLoss())
    (got * n).backward()
    want_running = sum(wl * int((b["labels"] != -100).sum()) for wl, b in zip(want_loss, batches))
    rel = abs(got.item() * n - want_running) / want_running
    print(f"[ragged full-width, window as one batch] running loss {got.item() * n:.4f} vs the oracle's loop {want_running:.4f} (rel {rel:.2e})")
    assert rel <= 2e-4
    worst, worst_key = 0.0, None
    for (k, p), (_, p2) in zip(model.named_parameters(), ref.named_parameters()):
        err = float((p.grad.float().cpu() - p2.grad).norm() / p2.grad.norm())
        worst, worst_key = max((worst, worst_key), (err, k))
        assert err <= TOL_GRAD_BF16, f"window as one batch, {k}: relative gradient error {err}"
    print(f"[ragged full-width, window as one batch] worst relative gradient error {worst:.2e} ({worst_key})")
    for a, b_ in zip(losses["padded"], losses["unpadded"]):
        # bf16 model: a row's attention outputs depend on where its keys fall in the 64-key tiles (fp32 sums in another order, then the bf16
        # rounding): the two losses sit 6e-6 .. 9e-5 from the fp32 oracle, on either side of it.  In fp32 the transform is exact to 1e-6
        # (tests/test_unpad.py, on the CPU oracle).
        assert abs(a - b_) <= 1e-4 * abs(a), (a, b_)
    assert losses["unpadded"] == losses["unpadded, work plan"]   # the plan is a matter of the backward
    for other in ("unpadded", "unpadded, work plan"):
        worst, worst_key = 0.0, None
        for k in grads["padded"]:
            err = float((grads["padded"][k] - grads[other][k]).norm() / grads["padded"][k].norm())
            if err > worst:
                worst, worst_key = err, k
        print(f"[ragged full-width] padded vs {other}: losses {losses}, worst relative gradient difference {worst:.2e} ({worst_key})")
        assert worst <= 1e-2, (worst_key, worst)


# ---------------------------------------------------------------------------------------------------------------------
# 2. The step's own GEMM shapes, exact on small integers
# ---------------------------------------------------------------------------------------------------------------------
def _sparse_ints(shape, p_nonzero, seed):
    """Entries in {-1, 0, 1}, nonzero with probability p: products and partial sums are exact in fp32, and with K p^2 small enough
    nearly every result is an integer of magnitude <= 256, which bf16 holds exactly."""
    g = torch.Generator(device=DEV).manual_seed(seed)
    sign = torch.randint(0, 2, shape, generator=g, device=DEV, dtype=torch.int8) * 2 - 1
    keep = torch.rand(shape, generator=g, device=DEV) < p_nonzero
    return (sign * keep).to(torch.bfloat16)


STEP_SHAPES = [
    # (what, layout, M, N, K, splits)                                                  NT=0: A[M,K] B[N,K];  NN=1: B[K,N];  TN=2: A[K,M] B[K,N]
    ("LM head forward, V=133258", 0, 16384, 133_376, 2048, 1),
    ("LM head forward, V=136450 (config A')", 0, 16384, 136_704, 2048, 1),
    ("LM head data gradient", 1, 16384, 2048, 133_376, 1),
    ("LM head weight gradient", 2, 133_376, 2048, 16384, 1),
    ("LM head weight gradient, V=130306 (config E)", 2, 130_560, 2048, 16384, 1),
    ("LM head weight gradient, rows of the last partial round (split-K 3)", 2, 2304, 2048, 16384, 3),
    ("dW_qkv (split-K)", 2, 3072, 2048, 16384, 0),
    ("dW_o (split-K)", 2, 2048, 2048, 16384, 0),
    ("dW13", 2, 16384, 2048, 16384, 1),
    ("dW2", 2, 2048, 8192, 16384, 1),
    ("W2 forward", 0, 16384, 2048, 8192, 1),
    ("gate-up data gradient", 1, 16384, 2048, 16384, 1),
    # round 4: the reference's default micro-batch (conf/data/_sft_base.yaml:21: 2 x 2048 rows) and a ragged packed length — the k-contiguous
    # and data-gradient forms with K split on the persistent kernel (0 = the model's own choice, which must be a split)
    ("W2 forward at T=4096 (split-K)", 0, 4096, 2048, 8192, 0),
    ("W_o forward at T=4096 (split-K)", 0, 4096, 2048, 2048, 0),
    ("gate-up data gradient at T=4096 (split-K)", 1, 4096, 2048, 16384, 0),
    ("QKV data gradient at T=4096 (split-K)", 1, 4096, 2048, 3072, 0),
    ("W2 forward at T=11520, 1.4 rounds (split-K)", 0, 11520, 2048, 8192, 0),
]


@pytest.mark.parametrize("what,layout,M,N,K,splits", STEP_SHAPES, ids=[s[0] for s in STEP_SHAPES])
def test_gemm_at_the_training_step_shapes_is_exact_on_integers(what, layout, M, N, K, splits):
    from ssi import _lib, ops
    p = min(0.5, math.sqrt(4096.0 / K))          # K p^2 = 4096 -> sums ~ N(0, 64^2): |sum| <= 256 for all but ~1e-4 of the entries
    a = _sparse_ints((M, K) if layout in (0, 1) else (K, M), p, 101)
    b = _sparse_ints((N, K) if layout == 0 else (K, N), p, 102)
    ref = (a.float() @ b.float().t()) if layout == 0 else ((a.float() @ b.float()) if layout == 1 else (a.float().t() @ b.float()))
    exact = ref.abs() <= 256
    assert float(exact.float().mean()) > 0.99
    prev = ops.set_impl(_lib.IMPL_MFMA)   # the MFMA path or an error: no silent fallback to the generic kernel
    try:
        c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
        if splits != 1:
            if splits == 0:
                splits = ops.splitk_choice(M, N, K)
                assert splits > 1, "the model splits this weight gradient over K"
            ws = torch.empty(splits * M * N, dtype=torch.float32, device=DEV)
            ops.gemm_splitk(layout, a, b, c, splits, ws)
        else:
            ops.gemm(layout, a, b, c)
        # (boolean-mask indexing breaks beyond 2^31 elements: compare under the mask instead)
        assert bool(((c.float() == ref) | ~exact).all()), f"{what}: not exact"
        assert bool(torch.isfinite(c.float()).all())
        # accumulate + device-side alpha, as the backward pass calls it (dW += alpha * dY^T X)
        c0 = torch.randint(-2, 3, (M, N), device=DEV, generator=torch.Generator(device=DEV).manual_seed(103)).to(torch.bfloat16)
        c1 = c0.clone()
        alpha = torch.tensor([0.5], dtype=torch.float32, device=DEV)
        if splits > 1:
            ops.gemm_splitk(layout, a, b, c1, splits, ws, alpha_dev=alpha, accumulate=True)
        else:
            ops.gemm(layout, a, b, c1, alpha_dev=alpha, accumulate=True)
        want = (0.5 * ref).bfloat16().float() + c0.float()      # the product is rounded to bf16, then added (two roundings, as F.linear then +)
        ok = exact & ((0.5 * ref) == (0.5 * ref).bfloat16().float()) & (want == want.bfloat16().float())  # exact under either rounding order
        assert float(ok.float().mean()) > 0.5
        assert bool(((c1.float() == want) | ~ok).all()), f"{what}: accumulate form not exact"
        if splits > 1 and layout in (0, 1):   # residual form of the split forward projections: rounded product + residual
            c2 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            ops.gemm_splitk(layout, a, b, c2, splits, ws, residual=c0)
            want = ref.bfloat16().float() + c0.float()
            ok = exact & (want == want.bfloat16().float())
            assert float(ok.float().mean()) > 0.5
            assert bool(((c2.float() == want) | ~ok).all()), f"{what}: residual form not exact"
            c3 = torch.empty_like(c2)
            ops.gemm(layout, a, b, c3, residual=c0)
            assert torch.equal(c2, c3), f"{what}: split and unsplit residual forms differ on exact data"
    finally:
        ops.set_impl(prev)
        del a, b, ref, exact, c
        torch.cuda.empty_cache()


TAIL_SHAPES = [
    # (what, layout, M, N, K): ragged row counts of an unpadded batch — more than one round of the 256 CUs, the last one partial
    ("W_o forward at T'=11520 (360 tiles: 256 + 104 split)", 0, 11520, 2048, 2048),
    ("W2 forward at T'=11520", 0, 11520, 2048, 8192),
    ("gate-up data gradient at T'=11520", 1, 11520, 2048, 16384),
    ("QKV data gradient at T'=9984 (312 tiles: 256 + 56 split)", 1, 9984, 2048, 3072),
    ("W_o forward at T'=16640 (520 tiles: 512 + 8 split)", 0, 16640, 2048, 2048),
    ("W_o forward at T'=13824 (432 tiles: 256 + 176 unsplit tail)", 0, 13824, 2048, 2048),
]


@pytest.mark.parametrize("what,layout,M,N,K", TAIL_SHAPES, ids=[s[0] for s in TAIL_SHAPES])
def test_gemm_with_the_last_partial_round_split_is_exact_on_integers(what, layout, M, N, K):
    """Round 5: ``HipLlamaDecoder._gemm`` on a grid of more than one round — whole rounds unsplit, the rows of the last partial round as a second
    launch with its own K split — exact on sparse integers, with and without residual, and equal to the whole-grid forms bit for bit on such
    data; the launches are counted (the split must actually happen where this test says it does)."""
    from ssi import _lib, ops
    from ssi.model import HipLlamaDecoder
    cfg = _full_config(5000)
    cfg.num_layers = 1
    model = HipLlamaDecoder(**cfg.parameters, dtype=torch.bfloat16, device=DEV, rope_cache_len=256)
    assert model._mfma_shapes() and model.split_tail_only and model.split_small_grids
    p = min(0.5, math.sqrt(4096.0 / K))
    a = _sparse_ints((M, K), p, 301)
    b = _sparse_ints((N, K) if layout == 0 else (K, N), p, 302)
    ref = (a.float() @ b.float().t()) if layout == 0 else (a.float() @ b.float())
    exact = ref.abs() <= 256
    assert float(exact.float().mean()) > 0.99
    res = torch.randint(-2, 3, (M, N), device=DEV, generator=torch.Generator(device=DEV).manual_seed(303)).to(torch.bfloat16)
    calls = []
    real_gemm, real_split = ops.gemm, ops.gemm_splitk
    ops.gemm = lambda layout_, a_, b_, c_, **kw: (calls.append(("gemm", c_.shape[0])), real_gemm(layout_, a_, b_, c_, **kw))[1]
    ops.gemm_splitk = lambda layout_, a_, b_, c_, splits, ws, **kw: (calls.append(("splitk", c_.shape[0], splits)), real_split(layout_, a_, b_, c_, splits, ws, **kw))[1]
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        out = {}
        for tail_only in (True, False):
            model.split_tail_only = tail_only
            for r in (None, res):
                c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
                calls.clear()
                model._gemm(layout, a, b, c, residual=r)
                out[tail_only, r is not None] = (c, list(calls))
        tm, tn = M // 256, N // 256
        m_main = ((tm * tn) // 256 * 256) // tn
        got_calls = out[True, False][1]
        assert got_calls[0] == ("gemm", m_main * 256), got_calls
        tail = (tm - m_main) * tn
        assert len(got_calls) == 2 and got_calls[1][1] == M - m_main * 256 and (got_calls[1][0] == "splitk") == (tail <= 128), got_calls
        for r in (False, True):
            want = ref.bfloat16().float() + (res.float() if r else 0.0)
            ok = exact & (want == want.bfloat16().float())
            assert float(ok.float().mean()) > 0.5
            c = out[True, r][0]
            assert bool(torch.isfinite(c.float()).all())
            assert bool(((c.float() == want) | ~ok).all()), f"{what}: not exact (residual {r})"
            assert torch.equal(c, out[False, r][0]), f"{what}: tail-split and whole-grid forms differ on exact data"
    finally:
        ops.set_impl(prev)
        ops.gemm, ops.gemm_splitk = real_gemm, real_split
        del model, a, b, ref, exact
        torch.cuda.empty_cache()


@pytest.mark.parametrize("what,M,N", [("dW_o x 8 layers", 2048, 2048), ("dW_qkv x 8 layers", 3072, 2048)])
def test_batched_weight_gradients_at_the_step_shape_are_exact_on_integers(what, M, N):
    """The deferred attention-projection weight gradients of a group of 8 layers as the model launches them (ssi_gemm_batched, K = 16384
    tokens, outputs strided through the flat gradient buffer): exact on sparse integers, plain and accumulate form."""
    from ssi import _lib, ops
    n, K = 8, 16384
    p = math.sqrt(4096.0 / K)
    a, b = _sparse_ints((n, K, M), p, 201), _sparse_ints((n, K, N), p, 202)
    ref = torch.bmm(a.float().transpose(1, 2), b.float())
    exact = ref.abs() <= 256
    assert float(exact.float().mean()) > 0.99
    stride = M * N + 5120 * 2048           # the other projection's gradient lies between two layers' outputs
    flat = torch.full((n * stride,), float("nan"), dtype=torch.bfloat16, device=DEV)
    c = torch.as_strided(flat, (n, M, N), (stride, N, 1), 0)
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        ops.gemm_batched(2, a, b, c)
        assert bool(((c.float() == ref) | ~exact).all()), f"{what}: not exact"
        assert bool(torch.isnan(torch.as_strided(flat, (n, stride - M * N), (stride, 1), M * N)).all())  # nothing written between the outputs
        c0 = torch.randint(-2, 3, (n, M, N), device=DEV, generator=torch.Generator(device=DEV).manual_seed(203)).to(torch.bfloat16)
        c.copy_(c0)
        alpha = torch.tensor([0.5], dtype=torch.float32, device=DEV)
        ops.gemm_batched(2, a, b, c, alpha_dev=alpha, accumulate=True)
        want = (0.5 * ref).bfloat16().float() + c0.float()
        ok = exact & ((0.5 * ref) == (0.5 * ref).bfloat16().float()) & (want == want.bfloat16().float())
        assert float(ok.float().mean()) > 0.5
        assert bool(((c.float() == want) | ~ok).all()), f"{what}: accumulate form not exact"
    finally:
        ops.set_impl(prev)
        del a, b, ref, exact, flat, c
        torch.cuda.empty_cache()


# ---------------------------------------------------------------------------------------------------------------------
# 3. Attention at S = 4096 / 8192 with the model's head counts
# ---------------------------------------------------------------------------------------------------------------------
def _documents(S, seed, lo=440, hi=1100):
    g = torch.Generator().manual_seed(seed)
    lens, left = [], S
    while left > 0:
        n = min(left, int(torch.randint(lo, hi + 1, (1,), generator=g)))
        lens.append(n)
        left -= n
    return lens


@pytest.mark.parametrize("S", [4096, 8192])
@pytest.mark.parametrize("packed", [False, True])
def test_attention_mfma_long_rows_32_heads(S, packed):
    from ssi import _lib, ops
    from test_kernels_gpu import _doc_arrays, _sdpa_block_ref, _sdpa_ref, rnd
    B, H, KV, hd = 1, 32, 8, 64
    qkv = rnd(B * S, (H + 2 * KV) * hd, dtype=torch.bfloat16, seed=111)
    qkv[S // 2 + 3, H * hd: H * hd + hd] *= 6.0
    do = rnd(B * S, H * hd, dtype=torch.bfloat16, seed=112)
    qr = qkv.float().clone().requires_grad_(True)
    ds = de = None
    if packed:
        rows = [_documents(S, 113)]
        assert len(rows[0]) >= S // 1100
        ds, de = (t.to(DEV) for t in _doc_arrays(rows, S))
        oref = _sdpa_block_ref(qr, B, S, H, KV, hd, rows)
    else:
        oref = _sdpa_ref(qr, B, S, H, KV, hd)
    oref.backward(do.float())
    prev = ops.set_impl(_lib.IMPL_MFMA)
    try:
        x, dout = qkv.to(DEV), do.to(DEV)
        out = torch.full((B * S, H * hd), float("nan"), dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_fwd(x, out, lse, B, S, H, KV, hd, ds, de)
        dqkv = torch.full_like(x, float("nan"))
        delta = torch.empty(B * H * S, dtype=torch.float32, device=DEV)
        ops.attn_bwd(x, out, dout, lse, dqkv, delta, B, S, H, KV, hd, ds, de)
        out2 = torch.empty_like(out)
        ops.attn_fwd(x, out2, lse, B, S, H, KV, hd, ds, de)
        d2 = torch.empty_like(x)
        ops.attn_bwd(x, out2, dout, lse, d2, delta, B, S, H, KV, hd, ds, de)
    finally:
        ops.set_impl(prev)
    assert torch.equal(out, out2) and torch.equal(dqkv, d2), "not bitwise reproducible"
    out, dqkv = out.cpu().float(), dqkv.cpu().float()
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    torch.testing.assert_close(out, oref.detach(), rtol=2e-2, atol=2e-2)
    # the log-sum-exp the backward reads, against fp32 math on sampled heads
    q = qr.detach()[:, : H * hd].view(S, H, hd)
    k = qr.detach()[:, H * hd:(H + KV) * hd].view(S, KV, hd)
    lse = lse.cpu().view(H, S)
    for h in (0, 13, 31):
        s = (q[:, h] @ k[:, h // (H // KV)].t()) / math.sqrt(hd)
        mask = torch.ones(S, S, dtype=torch.bool).tril()
        if packed:
            dsr = ds.cpu().long()
            mask &= torch.arange(S)[None, :] >= dsr[:, None]
        want = torch.logsumexp(s.masked_fill(~mask, float("-inf")), dim=-1)
        torch.testing.assert_close(lse[h], want, rtol=1e-4, atol=2e-3)
    scale = float(qr.grad.abs().max())
    assert float((dqkv - qr.grad).abs().max()) <= 3e-2 * scale
    rel = float((dqkv - qr.grad).norm() / qr.grad.norm())
    print(f"[attention S={S} packed={packed}] dqkv relative error {rel:.2e}")
    assert rel <= 1.5e-2


# ---------------------------------------------------------------------------------------------------------------------
# 4. The other BASELINE.json shapes through the whole model: config C (S = 4096) and config E (packed rows of 8192, V = 130 306)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,n_dsus,B,S,packed", [("config C shape: S=4096", 5000, 2, 4096, False), ("config E shape: packed S=8192", 2048, 1, 8192, True)],
                         ids=["C-S4096", "E-packed-S8192"])
def test_full_size_model_on_long_rows(name, n_dsus, B, S, packed):
    """Size-independent properties at the long-row shapes (no oracle run fits in seconds here): random-init loss ~ ln V, bitwise
    reproducibility of loss and gradients, eval == train loss, pad rows of the embedding gradient stay zero; for packed rows: a pack whose
    input_pos never restarts equals the plain causal row bit for bit, and documents change the loss."""
    from ssi.data import synthetic_batch, synthetic_packed_batch
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    cfg = _full_config(n_dsus)
    assert cfg.vocab_size == {5000: 133_258, 2048: 130_306}[n_dsus]
    model = HipLlamaDecoder(**cfg.parameters, dtype=torch.bfloat16, device=DEV, rope_cache_len=S)
    with torch.no_grad():
        model._flat.normal_(0.0, 0.02, generator=torch.Generator(device=DEV).manual_seed(7))
        model._view("emb")[cfg.vocab_size:].zero_()
        for p, nm, _ in model._param_src:
            if nm.endswith("norm"):
                p.fill_(1.0)
    model.train()
    loss_fn = CEWithChunkedOutputLoss()
    if packed:
        batch = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in synthetic_packed_batch(B, S, n_dsus, seed=42_831).items()}
        assert int((batch["input_pos"] == 0).sum()) >= 8          # several documents per row
    else:
        batch = {k: v.to(DEV) for k, v in synthetic_batch(B, S, n_dsus, seed=42_831).items()}

    def run(b):
        model.zero_grad()
        loss = compute_loss(b, model, loss_fn)
        loss.backward()
        return loss.item(), model._flat_grad.clone()

    l1, g1 = run(batch)
    assert math.isfinite(l1) and abs(l1 - math.log(cfg.vocab_size)) < 1.0
    l2, g2 = run(batch)
    assert l1 == l2 and torch.equal(g1, g2)
    assert bool(torch.isfinite(g1.float()).all()) and float(g1.float().abs().max()) > 0
    assert float(model._view("emb", None, model._flat_grad)[cfg.vocab_size:].abs().max()) == 0.0
    model.eval()
    with torch.inference_mode():
        le = compute_loss(batch, model, loss_fn).item()
    model.train()
    assert abs(le - l1) <= 1e-6 * abs(l1)
    if packed:
        plain = {"tokens": batch["tokens"], "labels": batch["labels"]}
        lp, gp = run(plain)
        assert abs(lp - l1) > 1e-5 * abs(l1)                       # documents no longer isolated -> a different loss
        one_doc = dict(plain, input_pos=torch.arange(S, device=DEV).expand(B, S).contiguous())
        lo, go = run(one_doc)
        # One document per row == plain causal attention.  The forward bit for bit (one kernel, the document bounds change nothing).  The
        # backward: a DEVICE input_pos brings no work plan, so these rows run the round-1..3 dQ / dK / dV kernels where the plain rows run
        # what the dispatcher picks for them — same products, another order of the fp32 sums: 1e-4 relative apart per kernel call, 0.3 % of
        # the bf16 results a step apart, and 16 layers of bf16 re-rounding on top.  Measured at HEAD of round 5: see ONE_DOC_VS_PLAIN (the
        # bound is 1.5 x the measurement; the bf16 model is 4.4e-2 from the fp32 oracle at this shape, a wrong kernel O(1)).  That the growth
        # is re-rounding and not a kernel is shown on ONE layer below (<= 1e-3); with the pipelined kernels on both sides — a HOST input_pos
        # gets its plan, the switches force the pipelined kernels for the plain rows — only the split of the heavy dK / dV chunks over the
        # query heads differs (6.5e-3 over 16 layers).
        assert lo == lp
        rel = float((go.float() - gp.float()).norm() / gp.float().norm())
        print(f"[one document per row vs plain rows, 16 layers, round-1..3 kernels vs dispatcher's] {rel:.3e}")
        assert rel <= ONE_DOC_VS_PLAIN, rel
        from ssi import _lib, ops
        prev = [ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, _lib.ATTN_MODE_NEW), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, _lib.ATTN_MODE_NEW)]
        try:
            lpn, gpn = run(plain)
            used_plain = ops.attn_last_dispatch()
            one_doc_host = dict(plain, input_pos=torch.arange(S).expand(B, S).contiguous(),
                                attn_plan=model.build_attn_plan(torch.arange(S).expand(B, S).contiguous(), force=True))
            assert one_doc_host["attn_plan"] is not None
            loh, goh = run(one_doc_host)
            used_doc = ops.attn_last_dispatch()
        finally:
            ops.set_attn_impl(_lib.ATTN_KERNEL_DQ, prev[0]), ops.set_attn_impl(_lib.ATTN_KERNEL_DKV, prev[1])
        both = _lib.ATTN_USED_DQ2 | _lib.ATTN_USED_DKV2
        assert used_plain & both == both and used_doc & (both | _lib.ATTN_USED_PLAN) == both | _lib.ATTN_USED_PLAN, (hex(used_plain), hex(used_doc))
        # (the plan of ONE 8192-token document splits its heavy dK / dV chunks over the query heads — another order of those sums; without a
        #  split the gradients are equal bit for bit: tests/test_kernels_gpu.py::test_attention_plan_for_plain_causal_rows holds that per kernel)
        relh = float((goh.float() - gpn.float()).norm() / gpn.float().norm())
        print(f"[one document per row, work plan vs plain rows, pipelined kernels on both sides] {relh:.3e} (plan splits {one_doc_host['attn_plan'].workspace_bytes > 0})")
        assert loh == lpn and (torch.equal(goh, gpn) if one_doc_host["attn_plan"].workspace_bytes == 0 else relh <= ONE_DOC_VS_PLAIN), relh


ONE_DOC_VS_PLAIN = 1.4e-2   # 1.5 x 9.07e-3, measured at HEAD of round 5 (gpurun_out/r05_t2.log; the same 9.07e-3 as before round 4's end-of-kernel
                            # drain went into attn_bwd_dq2_kernel: the difference is summation order + re-rounding, not that race); one layer: 3.6e-4


def test_one_document_per_row_equals_plain_rows_on_one_layer():
    """The comparison of test_full_size_model_on_long_rows on a ONE-layer model of the full width, where no re-rounding through further layers
    amplifies the kernels' different summation orders: the gradients of packed rows whose input_pos never restarts (round-1..3 backward
    kernels: a device input_pos brings no plan) and of plain rows (the dispatcher's kernels) agree to 1e-3."""
    from ssi.data import synthetic_batch
    from ssi.loss import CEWithChunkedOutputLoss, compute_loss
    from ssi.model import HipLlamaDecoder
    B, S = 1, 8192
    cfg = _full_config(2048)
    params = dict(cfg.parameters, num_layers=1)
    model = HipLlamaDecoder(**params, dtype=torch.bfloat16, device=DEV, rope_cache_len=S)
    with torch.no_grad():
        model._flat.normal_(0.0, 0.02, generator=torch.Generator(device=DEV).manual_seed(7))
        model._view("emb")[cfg.vocab_size:].zero_()
        for p, nm, _ in model._param_src:
            if nm.endswith("norm"):
                p.fill_(1.0)
    model.train()
    loss_fn = CEWithChunkedOutputLoss()
    plain = {k: v.to(DEV) for k, v in synthetic_batch(B, S, 2048, seed=42_831).items()}

    def run(b):
        model.zero_grad()
        loss = compute_loss(b, model, loss_fn)
        loss.backward()
        return loss.item(), model._flat_grad.clone()

    lp, gp = run(plain)
    lo, go = run(dict(plain, input_pos=torch.arange(S, device=DEV).expand(B, S).contiguous()))
    assert lo == lp
    rel = float((go.float() - gp.float()).norm() / gp.float().norm())
    print(f"[one document per row vs plain rows, 1 layer] {rel:.3e}")
    assert rel <= 1e-3, rel
    del model
    torch.cuda.empty_cache()
