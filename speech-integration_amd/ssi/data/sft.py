"""Supervised fine-tuning samples (speech units in, transcript out): ``SFTDataset`` and ``InputOutputToMessages`` of the
reference (``/root/reference/ssi/data/sft.py:25-231`` and ``:234-345``), same constructor keys (``conf/data/_sft_base.yaml``).

A sample becomes ``[system?] user(<speech> units <text>) assistant(transcript)`` messages, the tokenizer turns them into ids and a
mask, and ``labels = where(mask, -100, tokens)``.  The label shift happens in the step (``ssi/loss.py``), not here."""

from __future__ import annotations

from pathlib import Path
from typing import Any, Callable, Mapping

import numpy as np
from torch.utils.data import Dataset

from ..constants import CROSS_ENTROPY_IGNORE_IDX, RESERVED_BATCH_KEYS
from ..tokenizer import MODALITY_TOKEN_SPEECH, MODALITY_TOKEN_TEXT, Message, deduplicate_units, units_to_text, validate_messages
from .sources import open_source


class InputOutputToMessages:
    def __init__(self, train_on_input: bool, column_map: dict[str, str] | None = None, new_system_prompt: str | None = None,
                 image_dir: Path | None = None, modality_tokens: tuple[str, str] = (MODALITY_TOKEN_TEXT, MODALITY_TOKEN_SPEECH)):
        self.train_on_input = train_on_input
        self.new_system_prompt = new_system_prompt
        if column_map is not None:
            for key in ("input", "output"):
                if key not in column_map:
                    raise ValueError(f"Expected a key of '{key}' in column_map but found {column_map.keys()}.")
            self.column_map = dict(column_map)
        else:
            self.column_map = {"input": "input", "output": "output", "image": "image"}
        if "image" not in self.column_map and image_dir is not None:
            raise ValueError(f"image_dir is specified as {image_dir} but 'image' is not in column_map. "
                             "Please specify an 'image' key in column_map.")
        self.image_dir = image_dir
        self.modality_token_text, self.modality_token_speech = modality_tokens

    def __call__(self, sample: Mapping[str, Any], *, deduplicate: bool, use_modality_tokens: bool, inference: bool) -> dict[str, Any]:
        if "image" in sample or ("image" in self.column_map and self.column_map["image"] in sample):
            raise NotImplementedError("image inputs are not part of the speech path")   # the reference inherits them from torchtune
        sp_tkns = sample[self.column_map["input"]]
        if deduplicate:
            sp_tkns = deduplicate_units(sp_tkns)
        sp_span = units_to_text(sp_tkns)
        if use_modality_tokens:   # text follows: the next tokens are the assistant header
            sp_span = self.modality_token_speech + sp_span + self.modality_token_text
        output = "" if inference else sample[self.column_map["output"]]   # generation: empty assistant message
        messages = [Message(role="user", content=sp_span, masked=not self.train_on_input, eot=True),
                    Message(role="assistant", content=output, masked=False, eot=True)]
        if self.new_system_prompt is not None:
            messages.insert(0, Message(role="system", content=self.new_system_prompt, masked=True, eot=True))
        return {"messages": messages}


class SFTDataset(Dataset):
    def __init__(self, *, source: Any, model_tokenizer: Any, inference: bool = False, deduplicate: bool, use_modality_tokens: bool,
                 n_samples: int | None = None, filter_fn: Callable | None = None, train_on_input: bool,
                 column_map: dict[str, str] | None = None, new_system_prompt: str | None = None, image_dir: Path | None = None,
                 additional_keys: list[str] | None = None, **load_dataset_kwargs: Any) -> None:
        extra = {}
        if getattr(model_tokenizer, "modality_tokens", None) is not None:
            extra["modality_tokens"] = tuple(model_tokenizer.modality_tokens)
        self._message_transform = InputOutputToMessages(train_on_input=train_on_input, column_map=column_map,
                                                        new_system_prompt=new_system_prompt, image_dir=image_dir, **extra)
        self._model_tokenizer = model_tokenizer
        self._data = open_source(source, n_samples, **load_dataset_kwargs)
        if any(k in self._data.features for k in RESERVED_BATCH_KEYS):
            raise ValueError(f"Dataset contains reserved keys: {RESERVED_BATCH_KEYS}")
        if filter_fn is not None:
            self._data = self._data.filter(filter_fn)
        self.inference, self.deduplicate, self.use_modality_tokens = inference, deduplicate, use_modality_tokens
        self.additional_keys = list(additional_keys or [])

    def _bool_property(name: str):  # noqa: N805 - the three switches only take booleans (sft.py:161-192)
        def get(self) -> bool:
            return getattr(self, "_" + name)

        def set_(self, value: bool) -> None:
            if not isinstance(value, bool):
                raise TypeError(f"{name} must be a boolean.")
            setattr(self, "_" + name, value)
        return property(get, set_)

    inference = _bool_property("inference")
    deduplicate = _bool_property("deduplicate")
    use_modality_tokens = _bool_property("use_modality_tokens")
    del _bool_property

    def __len__(self) -> int:
        return len(self._data)

    def __getitem__(self, index: int) -> dict[str, Any]:
        sample = self._data[index]
        return self._prepare_sample(sample) | {k: sample[k] for k in self.additional_keys}

    def _prepare_sample(self, sample: Mapping[str, Any]) -> dict[str, Any]:
        transformed = self._message_transform(sample, deduplicate=self._deduplicate, use_modality_tokens=self._use_modality_tokens,
                                              inference=self._inference)
        validate_messages(transformed["messages"])
        out = self._model_tokenizer(transformed, inference=self._inference)
        if not ("tokens" in out and "mask" in out):
            raise ValueError(f"model_tokenizer returned the following keys: {', '.join(out)}. Must return 'tokens' and 'mask' as keys.")
        out["labels"] = np.where(out["mask"], CROSS_ENTROPY_IGNORE_IDX, out["tokens"]).tolist()
        assert len(out["tokens"]) == len(out["labels"])
        return out
