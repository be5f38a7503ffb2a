"""Model-shape source of truth (reference: ``/root/reference/ssi/llama_configs.py:29-155``).

``vocab_size = base text (128 000) + special text (256) + n_dsus + 2 * modality_tokens`` (``:97-99``); ``parameters``
returns exactly the keyword set the reference passes to torchtune's ``llama3_2`` builder (``:118-122``) and that
``ssi.model.HipLlamaDecoder`` accepts."""

from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import NamedTuple


class ModelCheckpointExpectations(NamedTuple):
    model_name: str
    n_shards: int
    num_layers: int
    hidden_size: int
    vocab_size: int


@dataclass
class ConfigLlama3_2:
    _base_vocab_size_txt: int
    _n_special_txt: int
    num_layers: int
    num_heads: int
    num_kv_heads: int
    embed_dim: int
    max_seq_len: int
    intermediate_dim: int
    attn_dropout: float
    norm_eps: float
    rope_base: int
    scale_factor: int
    _n_dsus: int = 0
    _modality_tokens: bool = False

    @property
    def n_dsus(self) -> int:
        return self._n_dsus

    @n_dsus.setter
    def n_dsus(self, n_dsus: int) -> None:
        if not isinstance(n_dsus, int) or isinstance(n_dsus, bool) or n_dsus < 0:
            raise ValueError("n_dsus must be a non-negative integer")
        self._n_dsus = n_dsus

    @property
    def modality_tokens(self) -> bool:
        return self._modality_tokens

    @modality_tokens.setter
    def modality_tokens(self, enable: bool) -> None:
        if not isinstance(enable, bool):
            raise ValueError("modality_tokens must be boolean")
        self._modality_tokens = enable

    def update_from_speech_cfg(self, cfg_speech) -> None:
        """In-place update from the ``speech`` config node (``n_dsus``, ``use_modality_tokens``)."""
        if not (hasattr(cfg_speech, "n_dsus") and hasattr(cfg_speech, "use_modality_tokens")) or isinstance(cfg_speech, dict):
            raise TypeError("cfg_speech must be a DictConfig object")
        self.n_dsus = cfg_speech.n_dsus
        self.modality_tokens = cfg_speech.use_modality_tokens

    @property
    def vocab_size(self) -> int:
        return self._base_vocab_size_txt + self._n_special_txt + self.n_dsus + (2 * self._modality_tokens)

    @property
    def checkpoint_expectations(self) -> ModelCheckpointExpectations:
        size_label = {2048: "1B", 3072: "3B"}.get(self.embed_dim, f"{self.embed_dim}d")
        return ModelCheckpointExpectations(f"Llama 3.2 {size_label}", 1, self.num_layers, self.embed_dim, self.vocab_size)

    @property
    def parameters(self) -> dict:
        return {"vocab_size": self.vocab_size} | {k: v for k, v in asdict(self).items() if not k.startswith("_")}


configllama3_2_1b = ConfigLlama3_2(
    _base_vocab_size_txt=128_000, _n_special_txt=256, num_layers=16, num_heads=32, num_kv_heads=8, embed_dim=2048,
    max_seq_len=131072, intermediate_dim=8192, attn_dropout=0.0, norm_eps=1e-5, rope_base=500_000, scale_factor=32,
)

configllama3_2_3b = ConfigLlama3_2(
    _base_vocab_size_txt=128_000, _n_special_txt=256, num_layers=28, num_heads=24, num_kv_heads=8, embed_dim=3072,
    max_seq_len=131072, intermediate_dim=8192, attn_dropout=0.0, norm_eps=1e-5, rope_base=500_000, scale_factor=32,
)
