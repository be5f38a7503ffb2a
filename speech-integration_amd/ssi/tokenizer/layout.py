"""Tokenizer facts the hot path itself reads: only ``pad_id`` (``/root/reference/ssi/trainer.py:388``).

When no ``tokenizer.model`` is at hand (benchmarks and tests on synthetic token ids) this object derives the special ids from
the vocabulary layout ``[text | dsu | modality(2) | special_text(256)]`` (``/root/reference/ssi/extend_llama3_2/__init__.py:100``).
It cannot encode text and says so; with a rank file ``setup_llama3_tokenizer`` returns ``Llama3TokenizerPUA`` instead."""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any

# positions of the named tokens inside Llama-3's block of 256 reserved specials
_SPECIALS = {"<|begin_of_text|>": 0, "<|end_of_text|>": 1, "<|finetune_right_pad_id|>": 4, "<|step_id|>": 5, "<|start_header_id|>": 6,
             "<|end_header_id|>": 7, "<|eom_id|>": 8, "<|eot_id|>": 9, "<|python_tag|>": 10, "<|image|>": 11, "<|video|>": 12}


@dataclass
class VocabLayoutTokenizer:
    vocab_size: int
    special_offset: int
    max_seq_len: int | None = None
    special_tokens: dict[str, int] = field(default_factory=dict)

    @property
    def pad_id(self) -> int:
        return self.special_tokens["<|finetune_right_pad_id|>"]

    @property
    def bos_id(self) -> int:
        return self.special_tokens["<|begin_of_text|>"]

    @property
    def eos_id(self) -> int:
        return self.special_tokens["<|end_of_text|>"]

    def encode(self, *_: Any, **__: Any):
        raise RuntimeError("this tokenizer only knows the vocabulary layout: give tokenizer.path (the extended original/tokenizer.model) "
                           "to encode text")

    tokenize_messages = __call__ = encode


def layout_tokenizer(llama_config: Any, max_seq_len: int | None = None) -> tuple[VocabLayoutTokenizer, dict[str, int]]:
    off = llama_config._base_vocab_size_txt + llama_config.n_dsus + 2 * int(llama_config.modality_tokens)
    special = {name: off + i for name, i in _SPECIALS.items()}
    return VocabLayoutTokenizer(llama_config.vocab_size, off, max_seq_len, special), special
