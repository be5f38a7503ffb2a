"""Discrete speech units (DSUs) as text: each unit is one Unicode Private-Use-Area character, which the extended
``tokenizer.model`` lists as a token of its own (``/root/reference/ssi/extend_llama3_2/__init__.py:60-72``: one line per
``dsu2pua(i)``, then the two modality tokens, appended to the 128 000 text merges).

``dsu2pua`` and the modality-token literals live in the un-vendored ``sardalign`` package (pinned git rev in the reference's
``pyproject.toml:75``) which is not on this image — **parity unpinned** for the two facts below that the reference tree does
not state:

* unit ``k`` -> ``chr(0xE000 + k)`` is documented ("U+E000 onwards via dsu2pua()",
  ``plans/Tokenizer Architecture - tiktoken vs HF tokenizer.json.md:70``); the BMP area holds 6400 code points, so for the
  8192-unit vocabularies this restatement continues in Supplementary Private Use Area-A (U+F0000...), an assumption;
* the literals of ``MODALITY_TOKEN_TEXT`` / ``MODALITY_TOKEN_SPEECH`` are not in the tree: the defaults below are placeholders,
  and every consumer takes them as arguments (they only need to equal the two lines appended to ``tokenizer.model``)."""

from __future__ import annotations

from itertools import groupby
from typing import Iterable, Sequence

import numpy as np

PUA_BMP_START, PUA_BMP_END = 0xE000, 0xF8FF          # 6400 code points
PUA_A_START, PUA_A_END = 0xF0000, 0xFFFFD            # Supplementary Private Use Area-A
_BMP_SIZE = PUA_BMP_END - PUA_BMP_START + 1

MODALITY_TOKEN_TEXT: str = "<|text|>"
MODALITY_TOKEN_SPEECH: str = "<|speech|>"


def dsu2pua(unit: int) -> str:
    unit = int(unit)
    if unit < 0:
        raise ValueError(f"negative speech unit {unit}")
    if unit < _BMP_SIZE:
        return chr(PUA_BMP_START + unit)
    if unit - _BMP_SIZE <= PUA_A_END - PUA_A_START:
        return chr(PUA_A_START + unit - _BMP_SIZE)
    raise ValueError(f"speech unit {unit} beyond the private-use areas")


def pua2dsu(ch: str) -> int:
    cp = ord(ch)
    if PUA_BMP_START <= cp <= PUA_BMP_END:
        return cp - PUA_BMP_START
    if PUA_A_START <= cp <= PUA_A_END:
        return cp - PUA_A_START + _BMP_SIZE
    raise ValueError(f"U+{cp:04X} is not a private-use code point")


def units_to_text(units: Iterable[int]) -> str:
    return "".join(map(dsu2pua, units))


def deduplicate_units(units: Sequence[int]) -> list[int]:
    """Collapse runs of equal units (``[k for k, g in groupby(units)]``, ``ssi/data/sft.py:305``, ``cpt.py:189,216``)."""
    if isinstance(units, np.ndarray):
        if units.size == 0:
            return []
        keep = np.empty(units.shape[0], dtype=bool)
        keep[0] = True
        np.not_equal(units[1:], units[:-1], out=keep[1:])
        return units[keep].tolist()
    return [k for k, _ in groupby(units)]
