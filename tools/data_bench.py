"""Host-side throughput of the data front end: tokens per second one CPU thread turns from MLS-shaped rows
(about 790 speech units at 50 Hz for 15.8 s of audio, a 40-word transcript) into collated int64 batches — to be held against what
one GPU consumes (bench.py: ~131 k tokens/s).  The merge table is a toy one (no tokenizer.model on the image): unit and modality
tokens cost what they cost with the real table (one dictionary lookup per pre-token), text words fall to the byte-pair loop more
often than with the real 128 k merges (whole-word hits) until the piece cache has seen them.

    python tools/data_bench.py [--rows 2000] [--batch 8]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "speech-integration_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=8)
    args = ap.parse_args()
    from ssi.data import SFTDataset, TextCompletionDataset, padded_collate_sft
    from ssi.tokenizer import Llama3TokenizerPUA
    from test_data_pipeline import toy_ranks
    rng = np.random.default_rng(0)
    n_units = 5000
    tok = Llama3TokenizerPUA(ranks=toy_ranks(n_units), max_seq_len=2048)
    vocab = ["".join(rng.choice(list("etaoinshrdlu"), size=rng.integers(2, 9))) for _ in range(20000)]
    zipf = np.minimum(rng.zipf(1.1, size=(args.rows, 40)), len(vocab)) - 1
    rows = []
    for i in range(args.rows):
        words = [vocab[j] for j in zipf[i]]
        units = np.repeat(rng.integers(0, n_units, size=560), rng.integers(1, 3, size=560))[:790].tolist()
        rows.append({"speech_tokens": units, "transcript": " ".join(words), "tokenized": words,
                     "aligned_start_times": (np.arange(40) * 0.39).tolist(), "aligned_end_times": (np.arange(40) * 0.39 + 0.3).tolist()})
    sft = SFTDataset(source=rows, model_tokenizer=tok, deduplicate=True, use_modality_tokens=True, train_on_input=True,
                     column_map={"input": "speech_tokens", "output": "transcript"}, new_system_prompt="You will act as an ASR system. ")
    cpt = TextCompletionDataset(tok, rows, sequence_type="interleaved", deduplicate=True, use_modality_tokens=True,
                                interleave_kwargs={"sampling_rate": 16000, "downsampling_ratio": 320, "mean_seq_len_tokens": 39.43, "binom_prob": 0.1})
    for name, ds in (("sft", sft), ("cpt interleaved", cpt)):
        for rep in ("cold", "warm"):
            t0, n = time.perf_counter(), 0
            for b in range(0, args.rows - args.batch + 1, args.batch):
                batch = padded_collate_sft([ds[i] for i in range(b, b + args.batch)], padding_idx=tok.pad_id)
                n += int((batch["tokens"] != tok.pad_id).sum())
            dt = time.perf_counter() - t0
            print(f"{name:16s} {rep}: {n / dt / 1e3:8.1f} k tokens/s per thread ({n} tokens, {args.rows} rows, {dt:.2f} s; piece cache {len(tok.bpe._cache)})")


if __name__ == "__main__":
    main()
