"""Continued-pretraining samples: text and speech units of one utterance concatenated or interleaved at word boundaries
(``TextCompletionDataset``, ``interleave``, ``concatenate_speech_text``, ``get_span_idxs_binomial`` of
``/root/reference/ssi/data/cpt.py:41-222``; constructor keys of ``conf/data/_cpt_base.yaml``).

Every sample draws from its own generator ``np.random.default_rng((SEED, epoch, index))`` (``cpt.py:122-125``), so what a sample
looks like depends on neither the order of access nor the number of loader workers or ranks; the reference's own tests of that
property (``tests/test_cpt_deterministic_rng.py``) are mirrored in ``tests/test_data_pipeline.py``.

Not in the reference tree (``sardalign`` is un-vendored) and therefore **parity unpinned**: the default column names below and
``times_to_dsu_idxs``, restated from ``plans/Bugfix - CPT Interleave Config Sampling Parameters.md:221-233``
(``int(seconds * sampling_rate / downsampling_ratio)``).  Unlike the reference, which resolves the ``*_key`` arguments and then
ignores them (``plans/Training Cleanup Tasks.md:90``), the column names given here are the ones read."""

from __future__ import annotations

from enum import Enum
from functools import partial
from itertools import zip_longest
from typing import Any, Callable, Mapping

import numpy as np
from torch.utils.data import Dataset

from ..constants import SEED
from ..tokenizer import MODALITY_TOKEN_SPEECH, MODALITY_TOKEN_TEXT, deduplicate_units, truncate, units_to_text
from .sources import open_source

TOKENIZED_KEY = "tokenized"                          # list of words
ALIGNMENT_START_TIME_KEY = "aligned_start_times"     # seconds, one per word
ALIGNMENT_END_TIME_KEY = "aligned_end_times"
SPEECH_TOKENS_KEY = "speech_tokens"                  # confirmed by conf/data/_sft_base.yaml:12


class CompletionSequenceType(Enum):
    INTERLEAVED = "interleaved"
    CONCATENATED_TXT_DSU = "concatenated_txt_dsu"
    CONCATENATED_DSU_TXT = "concatenated_dsu_txt"
    DSU_ONLY = "dsu_only"          # the last three are named but not implemented by the reference either
    TEXT_ONLY = "text_only"
    ALTERNATING = "alternating"


def times_to_dsu_idxs(times: tuple[float, float], sampling_rate: int, downsampling_ratio: int) -> tuple[int, int]:
    return tuple(int(t * sampling_rate / downsampling_ratio) for t in times)


def get_span_idxs_binomial(n: int, p: float, seq_len: int, rng: np.random.Generator) -> list[int]:
    """Span boundaries ``[0, ..., seq_len]`` with Binomial(n, p) span lengths, each at least 1 (``cpt.py:158-160``)."""
    subspan_idxs = np.maximum(rng.binomial(n, p, size=seq_len), 1).cumsum()
    return [0, *subspan_idxs[subspan_idxs < seq_len].tolist(), seq_len]


def interleave(sample: Mapping[str, Any], deduplicate: bool, use_modality_tokens: bool, *, rng: np.random.Generator, sampling_rate: int,
               downsampling_ratio: int, mean_seq_len_tokens: float, binom_prob: float, keys: tuple[str, str, str, str] | None = None,
               modality_tokens: tuple[str, str] = (MODALITY_TOKEN_TEXT, MODALITY_TOKEN_SPEECH)) -> str:
    k_tok, k_t0, k_t1, k_sp = keys or (TOKENIZED_KEY, ALIGNMENT_START_TIME_KEY, ALIGNMENT_END_TIME_KEY, SPEECH_TOKENS_KEY)
    start_with_text = rng.choice([True, False], p=[0.5, 0.5])      # first draw, then the span lengths: the order is part of the format
    words, t_starts, t_ends, units = sample[k_tok], sample[k_t0], sample[k_t1], sample[k_sp]
    span_idxs = get_span_idxs_binomial(int(mean_seq_len_tokens), binom_prob, len(words), rng=rng)
    even = list(zip(span_idxs[:-1:2], span_idxs[1::2], strict=True))     # spans 0, 2, 4, ... of the word sequence
    odd = list(zip(span_idxs[1:-1:2], span_idxs[2::2], strict=True))     # spans 1, 3, 5, ...
    text_idxs, dsu_idxs = (even, odd) if start_with_text else (odd, even)
    text_spans = [" ".join(words[a:b]) for a, b in text_idxs]
    dsu_spans = []
    for a, b in dsu_idxs:
        lo, hi = times_to_dsu_idxs((t_starts[a], t_ends[b - 1]), sampling_rate, downsampling_ratio)
        span = units[lo:hi]
        if deduplicate:
            span = deduplicate_units(span)
        dsu_spans.append(units_to_text(span))
    if use_modality_tokens:
        text_spans = [" ".join((modality_tokens[0], s)) for s in text_spans]
        dsu_spans = [" ".join((modality_tokens[1], s)) for s in dsu_spans]
    first, second = (text_spans, dsu_spans) if start_with_text else (dsu_spans, text_spans)
    return " ".join(s for pair in zip_longest(first, second) for s in pair if s is not None)


def concatenate_speech_text(sample: Mapping[str, Any], deduplicate: bool, use_modality_tokens: bool, *, rng: np.random.Generator,
                            start_with_text: bool, keys: tuple[str, str, str, str] | None = None,
                            modality_tokens: tuple[str, str] = (MODALITY_TOKEN_TEXT, MODALITY_TOKEN_SPEECH)) -> str:
    k_tok, _, _, k_sp = keys or (TOKENIZED_KEY, ALIGNMENT_START_TIME_KEY, ALIGNMENT_END_TIME_KEY, SPEECH_TOKENS_KEY)
    units = sample[k_sp]
    if deduplicate:
        units = deduplicate_units(units)
    text, dsus = " ".join(sample[k_tok]), units_to_text(units)
    if use_modality_tokens:
        text, dsus = " ".join((modality_tokens[0], text)), " ".join((modality_tokens[1], dsus))
    return " ".join((text, dsus) if start_with_text else (dsus, text))


class TextCompletionDataset(Dataset):
    def __init__(self, tokenizer: Any, source: Any, split: str | None = None, *, sequence_type: str, deduplicate: bool,
                 use_modality_tokens: bool, add_eos: bool = True, n_samples: int | None = None, tokenized_key: str | None = None,
                 alignment_start_time_key: str | None = None, alignment_end_time_key: str | None = None,
                 speech_tokens_key: str | None = None, filter_fn: Callable | None = None,
                 interleave_kwargs: dict[str, Any] | None = None, **load_dataset_kwargs: Any) -> None:
        self._tokenizer = tokenizer
        if split is not None:
            load_dataset_kwargs["split"] = split
        self._data = open_source(source, n_samples, **(load_dataset_kwargs if isinstance(source, str) else {}))
        self.add_eos = add_eos
        keys = (tokenized_key or TOKENIZED_KEY, alignment_start_time_key or ALIGNMENT_START_TIME_KEY,
                alignment_end_time_key or ALIGNMENT_END_TIME_KEY, speech_tokens_key or SPEECH_TOKENS_KEY)
        common: dict[str, Any] = {"keys": keys}
        if getattr(tokenizer, "modality_tokens", None) is not None:
            common["modality_tokens"] = tuple(tokenizer.modality_tokens)
        self.sequence_type = CompletionSequenceType(sequence_type)
        if self.sequence_type is CompletionSequenceType.INTERLEAVED:
            if not interleave_kwargs:
                raise ValueError("interleave_kwargs must be provided for interleaved sequence type")
            self.prompt_fn = partial(interleave, **dict(interleave_kwargs), **common)
        elif self.sequence_type is CompletionSequenceType.CONCATENATED_TXT_DSU:
            self.prompt_fn = partial(concatenate_speech_text, start_with_text=True, **common)
        elif self.sequence_type is CompletionSequenceType.CONCATENATED_DSU_TXT:
            self.prompt_fn = partial(concatenate_speech_text, start_with_text=False, **common)
        else:
            raise ValueError(f"Unsupported sequence type: {self.sequence_type}")
        self.deduplicate, self.use_modality_tokens = deduplicate, use_modality_tokens
        self._seed, self._epoch = SEED, 0
        if filter_fn is not None:
            self._data = self._data.filter(filter_fn)

    def set_epoch(self, epoch: int) -> None:
        self._epoch = epoch

    def __len__(self) -> int:
        return len(self._data)

    def __getitem__(self, index: int) -> dict[str, list[int]]:
        rng = np.random.default_rng((self._seed, self._epoch, index))
        return self._prepare_sample(self._data[index], rng)

    def _prepare_sample(self, sample: Mapping[str, Any], rng: np.random.Generator) -> dict[str, list[int]]:
        prompt = self.prompt_fn(sample=sample, deduplicate=self.deduplicate, use_modality_tokens=self.use_modality_tokens, rng=rng)
        tokens = self._tokenizer.encode(text=prompt, add_bos=True, add_eos=self.add_eos)
        if self._tokenizer.max_seq_len is not None:
            tokens = truncate(tokens, self._tokenizer.max_seq_len - 1)   # the reference's (torchtune's) "- 1"; the last id is not coerced to EOS
        return {"tokens": tokens, "labels": tokens.copy()}   # labels are shifted in the step
