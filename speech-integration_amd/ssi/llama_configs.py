"""Model shape and vocabulary layout of the Llama-3.2 decoders with a speech-unit (DSU) extended vocabulary.

Boundary kept from ``/root/reference/ssi/llama_configs.py:29-155``: the names a reference user touches —
``ConfigLlama3_2`` with ``n_dsus`` / ``modality_tokens`` (validated setters), ``update_from_speech_cfg``, ``vocab_size``,
``parameters`` (the keyword set of the model builder), ``checkpoint_expectations``, the two module-level configs.  What is behind
them is organised differently here: the id space is an explicit :class:`VocabLayout` (ordered, contiguous blocks) from which the
vocabulary size, the token-type ranges the trainer counts (``ssi/train_utils.py``) and the pad-to-tile size of the tied
embedding all follow, instead of arithmetic repeated at each use.
"""

from __future__ import annotations

from typing import Iterator, NamedTuple

BUILDER_KEYS = ("num_layers", "num_heads", "num_kv_heads", "embed_dim", "max_seq_len", "intermediate_dim", "attn_dropout", "norm_eps",
                "rope_base", "scale_factor")  # what the decoder constructor takes besides vocab_size, in the reference's order


class ModelCheckpointExpectations(NamedTuple):
    model_name: str
    n_shards: int
    num_layers: int
    hidden_size: int
    vocab_size: int


class VocabBlock(NamedTuple):
    name: str
    first: int   # first id of the block
    count: int

    @property
    def last(self) -> int:
        """Last id (inclusive); ``first - 1`` for an empty block."""
        return self.first + self.count - 1


class VocabLayout:
    """Id space of the extended tokenizer, in id order: ``text`` (the base BPE ranks), ``dsu`` (one id per discrete speech unit),
    ``modality`` (the two modality-switch tokens, present or not), ``special_text`` (Llama's 256 reserved specials, renumbered to
    the end) — the layout ``/root/reference/ssi/extend_llama3_2/__init__.py:88-106`` produces."""

    def __init__(self, n_text: int, n_special_text: int, n_dsus: int = 0, modality_tokens: bool = False) -> None:
        self.n_text, self.n_special_text, self.n_dsus, self.modality_tokens = n_text, n_special_text, n_dsus, modality_tokens

    def blocks(self) -> Iterator[VocabBlock]:
        first = 0
        for name, count in (("text", self.n_text), ("dsu", self.n_dsus), ("modality", 2 if self.modality_tokens else 0),
                            ("special_text", self.n_special_text)):
            yield VocabBlock(name, first, count)
            first += count

    def block(self, name: str) -> VocabBlock:
        for b in self.blocks():
            if b.name == name:
                return b
        raise KeyError(name)

    @property
    def size(self) -> int:
        return sum(b.count for b in self.blocks())

    def padded_size(self, multiple: int) -> int:
        """Rows of the tied embedding / LM-head matrix when stored in whole tiles of ``multiple`` rows."""
        return -(-self.size // multiple) * multiple


class ConfigLlama3_2:
    """Mutable on purpose: the trainer copies a module-level config and then sets the speech fields (and tests shrink the
    architecture through ``model_overrides``)."""

    def __init__(self, _base_vocab_size_txt: int, _n_special_txt: int, num_layers: int, num_heads: int, num_kv_heads: int, embed_dim: int,
                 max_seq_len: int, intermediate_dim: int, attn_dropout: float, norm_eps: float, rope_base: int, scale_factor: int,
                 _n_dsus: int = 0, _modality_tokens: bool = False) -> None:
        self._base_vocab_size_txt, self._n_special_txt = _base_vocab_size_txt, _n_special_txt
        self.num_layers, self.num_heads, self.num_kv_heads = num_layers, num_heads, num_kv_heads
        self.embed_dim, self.max_seq_len, self.intermediate_dim = embed_dim, max_seq_len, intermediate_dim
        self.attn_dropout, self.norm_eps, self.rope_base, self.scale_factor = attn_dropout, norm_eps, rope_base, scale_factor
        self._n_dsus, self._modality_tokens = 0, False
        self.n_dsus, self.modality_tokens = _n_dsus, _modality_tokens

    def __repr__(self) -> str:
        fields = ", ".join(f"{k}={getattr(self, k)!r}" for k in ("vocab_size", *BUILDER_KEYS))
        return f"ConfigLlama3_2({fields}, n_dsus={self.n_dsus}, modality_tokens={self.modality_tokens})"

    # ---- the speech part of the vocabulary ---------------------------------------------------------------------------------------
    @property
    def n_dsus(self) -> int:
        return self._n_dsus

    @n_dsus.setter
    def n_dsus(self, value: int) -> None:
        if isinstance(value, bool) or not isinstance(value, int) or value < 0:
            raise ValueError("n_dsus must be a non-negative integer")
        self._n_dsus = value

    @property
    def modality_tokens(self) -> bool:
        return self._modality_tokens

    @modality_tokens.setter
    def modality_tokens(self, value: bool) -> None:
        if not isinstance(value, bool):
            raise ValueError("modality_tokens must be boolean")
        self._modality_tokens = value

    def update_from_speech_cfg(self, cfg_speech) -> None:
        """Take ``n_dsus`` and ``use_modality_tokens`` from the ``speech`` node of the run config (a DictConfig, not a plain dict)."""
        if isinstance(cfg_speech, dict) or not all(hasattr(cfg_speech, k) for k in ("n_dsus", "use_modality_tokens")):
            raise TypeError("cfg_speech must be a DictConfig object")
        self.n_dsus, self.modality_tokens = cfg_speech.n_dsus, cfg_speech.use_modality_tokens

    # ---- derived ---------------------------------------------------------------------------------------------------------------
    @property
    def vocab_layout(self) -> VocabLayout:
        return VocabLayout(self._base_vocab_size_txt, self._n_special_txt, self._n_dsus, self._modality_tokens)

    @property
    def vocab_size(self) -> int:
        return self.vocab_layout.size

    @property
    def parameters(self) -> dict:
        """Keyword arguments of the decoder constructor (the reference hands the same set to torchtune's ``llama3_2``)."""
        return {"vocab_size": self.vocab_size, **{k: getattr(self, k) for k in BUILDER_KEYS}}

    @property
    def checkpoint_expectations(self) -> ModelCheckpointExpectations:
        label = {2048: "1B", 3072: "3B"}.get(self.embed_dim, f"{self.embed_dim}d")
        return ModelCheckpointExpectations(model_name=f"Llama 3.2 {label}", n_shards=1, num_layers=self.num_layers,
                                           hidden_size=self.embed_dim, vocab_size=self.vocab_size)


_SHARED = dict(_base_vocab_size_txt=128_000, _n_special_txt=256, num_kv_heads=8, max_seq_len=131072, intermediate_dim=8192,
               attn_dropout=0.0, norm_eps=1e-5, rope_base=500_000, scale_factor=32)
configllama3_2_1b = ConfigLlama3_2(num_layers=16, num_heads=32, embed_dim=2048, **_SHARED)
configllama3_2_3b = ConfigLlama3_2(num_layers=28, num_heads=24, embed_dim=3072, **_SHARED)
