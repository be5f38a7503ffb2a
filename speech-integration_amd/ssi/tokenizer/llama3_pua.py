"""Llama-3 tokenizer with private-use-area support: the text front end of the data pipeline.

Restates, without tiktoken or torchtune (neither is on this image):

* ``Llama3TokenizerPUA`` (``/root/reference/ssi/tokenizer/monkeypatch.py:14-43``): torchtune-0.5.0's ``Llama3Tokenizer`` with the
  pre-tokenisation pattern changed so that every ``\\p{Co}`` character is a pre-token of its own;
* torchtune-0.5.0 ``TikTokenBaseTokenizer.encode`` (specials in the text are ordinary text; long runs are split first) and
  ``Llama3Tokenizer.tokenize_message(s)`` / ``__call__`` (header ``<|start_header_id|>role<|end_header_id|>\\n\\n``, stripped
  body, ``<|eot_id|>`` / ``<|eom_id|>``; BOS and EOS always masked; truncation to ``max_seq_len``).

torchtune's source is not in the container, so these rules come from the published 0.5.0 release as the reference uses it
(``ssi/data/sft.py:205-209``, ``ssi/data/cpt.py:137-144``) — **parity unpinned**: no fixture of the reference holds token ids."""

from __future__ import annotations

import re as _stdlib_re
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Callable, Mapping, Sequence

import numpy as np
import regex

from .bpe import BytePairEncoder, load_tiktoken_bpe
from .units import MODALITY_TOKEN_SPEECH, MODALITY_TOKEN_TEXT, dsu2pua

# tiktoken's public cl100k_base pattern, and the reference's variant (monkeypatch.py:7): private-use characters are kept out of
# the two "anything but letters/digits" classes and matched one at a time by a last alternative
CL100K_PATTERN = (r"""(?i:'s|'t|'re|'ve|'m|'ll|'d)|[^\r\n\p{L}\p{N}]?\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+[\r\n]*|\s*[\r\n]+|"""
                  r"""\s+(?!\S)|\s+""")
CL100K_PATTERN_PUA = CL100K_PATTERN.replace(r"\p{N}]", r"\p{N}\p{Co}]") + r"|\p{Co}"
assert CL100K_PATTERN_PUA.count(r"\p{Co}") == 3

# torchtune 0.5.0 LLAMA3_SPECIAL_TOKENS, in dictionary order (the reference numbers them base_vocab + position,
# ssi/tokenizer/__init__.py:29-31; it asserts there are 256, :14-15)
_NAMED_SPECIALS = ["<|begin_of_text|>", "<|end_of_text|>", "<|reserved_special_token_0|>", "<|reserved_special_token_1|>",
                   "<|finetune_right_pad_id|>", "<|step_id|>", "<|start_header_id|>", "<|end_header_id|>", "<|eom_id|>",
                   "<|eot_id|>", "<|python_tag|>", "<|image|>", "<|video|>"]
LLAMA3_SPECIAL_TOKENS: list[str] = _NAMED_SPECIALS + [f"<|reserved_special_token_{2 + i}|>" for i in range(256 - len(_NAMED_SPECIALS))]
assert len(LLAMA3_SPECIAL_TOKENS) == 256

MAX_ENCODE_CHARS = 400_000          # torchtune TikTokenBaseTokenizer constants
MAX_NO_WHITESPACE_CHARS = 25_000
_SENTINELS = ("\U0010FFFC", "\U0010FFFD")   # stand in for the modality literals during pre-tokenisation (Private Use Area-B)

Role = str  # "system" | "user" | "assistant" | "ipython"


@dataclass
class Message:
    """torchtune ``Message``: a role, a list of ``{"type": "text"|"image", "content": ...}`` items (a bare string is one text
    item), whether the loss ignores it, and whether it ends the turn (``eot``) or only the message (``eom``)."""
    role: Role
    content: Any
    masked: bool = False
    ipython: bool = False
    eot: bool = True

    def __post_init__(self) -> None:
        if self.role not in ("system", "user", "assistant", "ipython"):
            raise ValueError(f"unknown role {self.role!r}")
        if isinstance(self.content, str):
            self.content = [{"type": "text", "content": self.content}]
        if self.ipython and self.contains_media:
            raise ValueError("Media tokens in tool calls are not supported.")

    @property
    def contains_media(self) -> bool:
        return any(item["type"] == "image" for item in self.content)

    @property
    def text_content(self) -> str:
        return "".join(item["content"] for item in self.content if item["type"] == "text")


def validate_messages(messages: Sequence[Message]) -> None:
    """System prompt first if at all, then user and assistant taking turns (torchtune ``validate_messages``)."""
    if len(messages) < 2:
        raise ValueError(f"Messages must be at least length 2, but got {len(messages)} messages")
    last_turn = "assistant"
    for i, message in enumerate(messages):
        if message.role == "assistant" and last_turn != "user":
            raise ValueError(f"Assistant message before expected user message at index {i} in messages")
        if message.role == "user" and last_turn == "user":
            raise ValueError(f"Two consecutive user messages at index {i} and {i - 1} in messages")
        if message.role == "system" and i > 0:
            raise ValueError(f"System message at index {i} in messages, but system messages must come first")
        last_turn = message.role


def truncate(tokens: list, max_seq_len: int, eos_id: Any = None) -> list:
    out = tokens[:max_seq_len]
    if eos_id is not None and out and out[-1] != eos_id:
        out[-1] = eos_id
    return out


def _split_long_repetitions(s: str, max_run: int) -> list[str]:
    """Cut ``s`` wherever a run of only-whitespace or only-non-whitespace characters exceeds ``max_run``."""
    if len(s) <= max_run:
        return [s]
    out, start, run, state = [], 0, 0, None
    for i, ch in enumerate(s):
        sp = ch.isspace()
        if sp != state:
            state, run = sp, 1
        else:
            run += 1
            if run > max_run:
                out.append(s[start:i])
                start, run = i, 1
    out.append(s[start:])
    return out


class Llama3TokenizerPUA:
    def __init__(self, path: str | Path | None = None, special_tokens: Mapping[str, int] | None = None, max_seq_len: int | None = None,
                 prompt_template: Callable[[list[Message]], list[Message]] | None = None, *,
                 ranks: dict[bytes, int] | None = None, modality_tokens: tuple[str, str] = (MODALITY_TOKEN_TEXT, MODALITY_TOKEN_SPEECH)):
        if ranks is None:
            if path is None:
                raise ValueError("Llama3TokenizerPUA needs the tiktoken rank file (original/tokenizer.model) or a rank table")
            ranks = load_tiktoken_bpe(path)
        self.path = None if path is None else str(path)
        self.bpe = BytePairEncoder(ranks)
        self.base_vocab_size = len(ranks)
        if self.base_vocab_size != max(ranks.values()) + 1:
            raise ValueError("Requirement: base vocab to be contiguous and 0-indexed")
        if special_tokens is None:
            special_tokens = {t: self.base_vocab_size + i for i, t in enumerate(LLAMA3_SPECIAL_TOKENS)}
        self.special_tokens = dict(special_tokens)
        for name in ("<|begin_of_text|>", "<|end_of_text|>", "<|finetune_right_pad_id|>", "<|start_header_id|>", "<|end_header_id|>",
                     "<|eom_id|>", "<|eot_id|>", "<|python_tag|>"):
            if name not in self.special_tokens:
                raise ValueError(f"{name} missing from special_tokens")
        st = self.special_tokens
        self.bos_id, self.eos_id, self.pad_id = st["<|begin_of_text|>"], st["<|end_of_text|>"], st["<|finetune_right_pad_id|>"]
        self.start_header_id, self.end_header_id = st["<|start_header_id|>"], st["<|end_header_id|>"]
        self.eom_id, self.eot_id, self.python_tag = st["<|eom_id|>"], st["<|eot_id|>"], st["<|python_tag|>"]
        self.step_id, self.image_id = st.get("<|step_id|>"), st.get("<|image|>")
        self.stop_tokens = [self.eos_id, self.eot_id]
        self.max_seq_len, self.prompt_template = max_seq_len, prompt_template
        self._pat = regex.compile(CL100K_PATTERN_PUA)
        # modality literals that the table lists are kept whole (each stands for one token id)
        self._atoms: dict[str, int] = {}
        self._atom_subs: list[tuple[str, str]] = []
        for literal, sentinel in zip(modality_tokens, _SENTINELS):
            rank = ranks.get(literal.encode("utf-8"))
            if rank is not None:
                self._atoms[sentinel] = rank
                self._atom_subs.append((literal, sentinel))
        self.modality_tokens = modality_tokens
        self._special_by_id = {v: k for k, v in self.special_tokens.items()}
        self._unit_ids: np.ndarray | None = None

    # ---- sizes -------------------------------------------------------------------------------------------------------------
    @property
    def vocab_size(self) -> int:
        return self.base_vocab_size + len(self.special_tokens)

    @property
    def n_units(self) -> int:
        return len(self.unit_ids)

    @property
    def unit_ids(self) -> np.ndarray:
        """Token id of every speech unit the table lists (units 0, 1, ... until the first one that is missing)."""
        if self._unit_ids is None:
            ids, k = [], 0
            while (r := self.bpe.ranks.get(dsu2pua(k).encode("utf-8"))) is not None:
                ids.append(r)
                k += 1
            self._unit_ids = np.asarray(ids, dtype=np.int64)
        return self._unit_ids

    # ---- text -> ids -------------------------------------------------------------------------------------------------------
    def _encode_ordinary(self, text: str, out: list[int]) -> None:
        for literal, sentinel in self._atom_subs:
            if sentinel in text:
                raise ValueError(f"text contains the reserved code point U+{ord(sentinel):X}")
            text = text.replace(literal, sentinel)
        atoms, enc = self._atoms, self.bpe.encode_piece
        for piece in self._pat.findall(text):
            a = atoms.get(piece)
            if a is not None:
                out.append(a)
            else:
                out.extend(enc(piece.encode("utf-8")))

    def encode(self, text: str, add_bos: bool = True, add_eos: bool = True) -> list[int]:
        tokens: list[int] = [self.bos_id] if add_bos else []
        for i in range(0, len(text), MAX_ENCODE_CHARS):
            for sub in _split_long_repetitions(text[i:i + MAX_ENCODE_CHARS], MAX_NO_WHITESPACE_CHARS):
                self._encode_ordinary(sub, tokens)
        if add_eos:
            tokens.append(self.eos_id)
        return tokens

    def encode_units(self, units: Sequence[int]) -> list[int]:
        """Ids of a run of speech units without going through text (equal to ``encode("".join(map(dsu2pua, units)))``)."""
        return self.unit_ids[np.asarray(units, dtype=np.int64)].tolist()

    def decode(self, token_ids: Sequence[int], truncate_at_eos: bool = True, skip_special_tokens: bool = True) -> str:
        ids = list(token_ids)
        if truncate_at_eos and self.eos_id in ids:
            ids = ids[:ids.index(self.eos_id)]
        chunks: list[bytes] = []
        for t in ids:
            name = self._special_by_id.get(int(t))
            if name is None:
                chunks.append(self.bpe.tokens[int(t)])
            elif not skip_special_tokens:
                chunks.append(name.encode())
        return b"".join(chunks).decode("utf-8", errors="replace")

    # ---- messages -> ids ---------------------------------------------------------------------------------------------------
    def _tokenize_header(self, message: Message) -> list[int]:
        return ([self.start_header_id] + self.encode(message.role.strip(), add_bos=False, add_eos=False) + [self.end_header_id]
                + self.encode("\n\n", add_bos=False, add_eos=False))

    def _tokenize_body(self, message: Message) -> list[int]:
        body: list[int] = []
        for item in message.content:
            if item["type"] == "text":
                body += self.encode(item["content"].strip(), add_bos=False, add_eos=False)
            elif item["type"] == "image":
                if self.image_id is None:
                    raise ValueError("<|image|> is not among the special tokens")
                body.append(self.image_id)
            else:
                raise RuntimeError(f"Unsupported message content type: {item['type']}")
        return ([self.python_tag] + body) if message.ipython else body

    def tokenize_message(self, message: Message, tokenize_header: bool = True, tokenize_end: bool = True) -> list[int]:
        header = self._tokenize_header(message) if tokenize_header else []
        end = ([self.eot_id] if message.eot else [self.eom_id]) if tokenize_end else []
        return header + self._tokenize_body(message) + end

    def tokenize_messages(self, messages: list[Message], add_eos: bool = True) -> tuple[list[int], list[bool]]:
        if self.prompt_template is not None:
            messages = self.prompt_template(messages)
        tokens, mask = [self.bos_id], [True]     # BOS and EOS are always masked
        for message in messages:
            t = self.tokenize_message(message)
            tokens += t
            mask += [message.masked] * len(t)
            if self.max_seq_len and len(tokens) >= self.max_seq_len:
                break
        if add_eos:
            tokens.append(self.eos_id)
            mask.append(True)
        if self.max_seq_len:
            tokens = truncate(tokens, self.max_seq_len, self.eos_id if add_eos else None)
            mask = truncate(mask, self.max_seq_len, True if add_eos else None)
        return tokens, mask

    def __call__(self, sample: dict[str, Any], inference: bool = False) -> dict[str, Any]:
        messages = sample.pop("messages")
        sample["tokens"], sample["mask"] = self.tokenize_messages(messages, add_eos=not inference)
        return sample
