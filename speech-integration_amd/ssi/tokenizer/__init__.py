"""``setup_llama3_tokenizer`` (``/root/reference/ssi/tokenizer/__init__.py:18-45``): load the (extended) tiktoken rank file and
number the 256 Llama-3 special tokens after it.  Without a file (``path`` null or missing — synthetic-data runs) the special ids
follow from the vocabulary layout in ``llama_config`` and the returned object cannot encode text."""

from __future__ import annotations

import hashlib
import os
from pprint import pformat
from typing import Any

from .bpe import BytePairEncoder, dump_tiktoken_bpe, load_tiktoken_bpe
from .layout import VocabLayoutTokenizer, layout_tokenizer
from .llama3_pua import CL100K_PATTERN_PUA, LLAMA3_SPECIAL_TOKENS, Llama3TokenizerPUA, Message, truncate, validate_messages
from .units import MODALITY_TOKEN_SPEECH, MODALITY_TOKEN_TEXT, deduplicate_units, dsu2pua, pua2dsu, units_to_text

Llama3Tokenizer = Llama3TokenizerPUA  # the reference's alias (tokenizer/__init__.py:10)

__all__ = ["setup_llama3_tokenizer", "Llama3Tokenizer", "Llama3TokenizerPUA", "VocabLayoutTokenizer", "Message", "validate_messages",
           "truncate", "LLAMA3_SPECIAL_TOKENS", "CL100K_PATTERN_PUA", "BytePairEncoder", "load_tiktoken_bpe", "dump_tiktoken_bpe",
           "dsu2pua", "pua2dsu", "units_to_text", "deduplicate_units", "MODALITY_TOKEN_TEXT", "MODALITY_TOKEN_SPEECH"]


def setup_llama3_tokenizer(path: Any = None, max_seq_len: int | None = None, prompt_template: Any = None, verbose: bool = False,
                           llama_config: Any = None, modality_tokens: Any = None, **_: Any):
    if path is None or not os.path.isfile(str(path)):
        if llama_config is None:
            raise ValueError(f"tokenizer.path={path!r} is not a file and no llama_config (vocabulary layout) was given")
        return layout_tokenizer(llama_config, max_seq_len)
    with open(path, "rb") as f:
        expected_hash = hashlib.sha256(f.read()).hexdigest()
    ranks = load_tiktoken_bpe(path, expected_hash)
    base_vocab_size = len(ranks)
    special = dict(zip(LLAMA3_SPECIAL_TOKENS, range(base_vocab_size, base_vocab_size + len(LLAMA3_SPECIAL_TOKENS)), strict=True))
    kw = {} if modality_tokens is None else {"modality_tokens": tuple(modality_tokens)}
    tokenizer = Llama3TokenizerPUA(path=str(path), special_tokens=special, max_seq_len=max_seq_len, prompt_template=prompt_template,
                                   ranks=ranks, **kw)
    if llama_config is not None and tokenizer.vocab_size != llama_config.vocab_size:
        raise ValueError(f"tokenizer vocabulary ({tokenizer.vocab_size}) != model vocabulary ({llama_config.vocab_size})")
    if verbose:
        print(f"Llama 3 tokenizer: {path}: {base_vocab_size} ranked tokens ({tokenizer.n_units} speech units) + {len(special)} specials = "
              f"{tokenizer.vocab_size}; specials: {pformat(special, sort_dicts=False, underscore_numbers=True, compact=True)}")
    return tokenizer, special
