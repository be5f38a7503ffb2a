#!/usr/bin/env python
"""Writes tests/golden/config_a.npz: the fp32 CPU oracle (oracle/llama_oracle.py, restating /root/reference/ssi/loss.py:7-22 over the
torchtune-0.5.0 decoder) on BASELINE config A's batch — B = 8, S = 2048, V = 133 258, the full 16-layer 1B model — reduced to what the
GPU test compares: the loss, both label counts, and per parameter the gradient norm, a 4096-bucket count-sketch of the gradient
(tests/fullsize_recipe.py) and, for five named parameters, the first 4096 elements.  Recipe (all in the file too): weights
seeded_full_state_dict(params, 2024) (bf16-representable), batch synthetic_batch(8, 2048, 5000, seed=42831).

Needs ~60 GB of host memory and ~4 minutes on 16 cores: run it on the GPU box's host (no GPU is touched),

    gpurun -- 'python tests/golden/make_config_a.py --out gpurun_out/config_a.npz'

and copy the result to tests/golden/.  tests/test_oracle_golden.py re-derives the recipe's digests on the CPU and checks the sketch
arithmetic at a reduced size, so the fixture cannot drift from the code that reads it."""
import argparse
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "speech-integration_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import fullsize_recipe as fr  # noqa: E402


def oracle_fixture(params, sd, batch, rope_len):
    """Loss, counts and the reduced gradients of the oracle on ``batch`` (any size: the reduced-size check of the CPU suite calls this too)."""
    from oracle.llama_oracle import OracleCEWithChunkedOutputLoss, OracleLlama
    from oracle.llama_oracle import compute_loss as oracle_loss
    with torch.device("meta"):
        ref = OracleLlama(**params, rope_cache_len=rope_len)
    rope = ref.rope.clone()
    ref = ref.to_empty(device="cpu")
    ref.rope = rope
    ref.load_state_dict(sd)
    ref.set_num_output_chunks(8)
    loss = oracle_loss(batch, ref, OracleCEWithChunkedOutputLoss())
    loss.backward()
    names, norms, sketches, heads = [], [], [], {}
    for k, p in ref.named_parameters():
        names.append(k)
        norms.append(float(p.grad.double().norm()))
        sketches.append(fr.sketch(p.grad).float().numpy())
        if k in fr.NAMED:
            heads[k] = p.grad.reshape(-1)[:4096].clone().numpy()
    labels = batch["labels"]
    shifted = torch.hstack((labels[..., 1:], torch.full_like(labels[..., -1:], -100)))
    return dict(loss=float(loss.detach()), n_unshifted=int((labels != -100).sum()), n_shifted=int((shifted != -100).sum()), names=names,
                norms=np.asarray(norms, dtype=np.float64), sketches=np.stack(sketches), heads=heads)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(HERE, "config_a.npz"))
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    from ssi.data import synthetic_batch
    if args.threads:
        torch.set_num_threads(args.threads)
    n_dsus, B, S = 5000, 8, 2048
    params = fr.full_config(n_dsus).parameters
    t0 = time.time()
    sd = fr.seeded_full_state_dict(params, fr.WEIGHT_SEED)
    batch = synthetic_batch(B, S, n_dsus, seed=fr.BATCH_SEED)
    print(f"weights + batch in {time.time() - t0:.0f} s; oracle forward + backward on {torch.get_num_threads()} threads ...", flush=True)
    t1 = time.time()
    fx = oracle_fixture(params, sd, batch, S)
    secs = time.time() - t1
    print(f"oracle: {secs:.0f} s, loss {fx['loss']:.6f}", flush=True)
    first = next(iter(sd))
    np.savez_compressed(
        args.out, loss=fx["loss"], n_unshifted=fx["n_unshifted"], n_shifted=fx["n_shifted"], names=np.asarray(fx["names"]), norms=fx["norms"],
        sketches=fx["sketches"], **{"head/" + k: v for k, v in fx["heads"].items()},
        recipe_n_dsus=n_dsus, recipe_B=B, recipe_S=S, recipe_weight_seed=fr.WEIGHT_SEED, recipe_batch_seed=fr.BATCH_SEED,
        recipe_vocab=params["vocab_size"], recipe_first_tensor=first, digest_first_tensor_rows=fr.digest(sd[first][:64]),
        digest_tokens=fr.digest(batch["tokens"]), digest_labels=fr.digest(batch["labels"]), sketch_buckets=fr.SKETCH_BUCKETS,
        oracle_seconds=secs, oracle_threads=torch.get_num_threads(), torch_version=torch.__version__)
    print(f"wrote {args.out} ({os.path.getsize(args.out) / 1e6:.1f} MB)")


if __name__ == "__main__":
    main()
