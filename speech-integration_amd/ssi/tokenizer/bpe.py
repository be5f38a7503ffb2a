"""Byte-pair encoding over a tiktoken-format rank file (``original/tokenizer.model`` of Llama 3: one ``base64(token) rank``
line per token).  ``tiktoken`` — the Rust library the reference reaches through torchtune's ``TikTokenBaseTokenizer``
(``/root/reference/ssi/tokenizer/__init__.py:5,26``) — is not on this image; its published algorithm is restated here:
a pre-token that is itself in the table maps to its rank, otherwise its bytes are merged greedily, lowest-ranked
adjacent pair first, until no adjacent pair is in the table."""

from __future__ import annotations

import base64
import hashlib
from pathlib import Path


def load_tiktoken_bpe(path: str | Path, expected_hash: str | None = None) -> dict[bytes, int]:
    data = Path(path).read_bytes()
    if expected_hash is not None and hashlib.sha256(data).hexdigest() != expected_hash:
        raise ValueError(f"hash mismatch for {path}")
    ranks: dict[bytes, int] = {}
    for line in data.splitlines():
        if not line:
            continue
        token, rank = line.split()
        ranks[base64.b64decode(token)] = int(rank)
    return ranks


def dump_tiktoken_bpe(ranks: dict[bytes, int], path: str | Path) -> None:
    with open(path, "wb") as f:
        for token, rank in sorted(ranks.items(), key=lambda kv: kv[1]):
            f.write(base64.b64encode(token) + b" " + str(rank).encode() + b"\n")


class BytePairEncoder:
    def __init__(self, ranks: dict[bytes, int], cache_size: int = 1 << 20):
        self.ranks = ranks
        self.tokens: dict[int, bytes] = {r: t for t, r in ranks.items()}
        if len(self.tokens) != len(ranks):
            raise ValueError("duplicate ranks in the merge table")
        self._cache: dict[bytes, tuple[int, ...]] = {}
        self._cache_size = cache_size

    def __len__(self) -> int:
        return len(self.ranks)

    def encode_piece(self, piece: bytes) -> tuple[int, ...]:
        rank = self.ranks.get(piece)
        if rank is not None:
            return (rank,)
        hit = self._cache.get(piece)
        if hit is not None:
            return hit
        ranks = self.ranks
        parts = [piece[i:i + 1] for i in range(len(piece))]
        while len(parts) > 1:
            best, best_rank = -1, None
            for i in range(len(parts) - 1):
                r = ranks.get(parts[i] + parts[i + 1])
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best_rank is None:
                break
            parts[best:best + 2] = [parts[best] + parts[best + 1]]
        try:
            out = tuple(ranks[p] for p in parts)
        except KeyError as e:  # a table without all 256 single bytes
            raise ValueError(f"byte sequence {e.args[0]!r} is not in the merge table") from None
        if len(self._cache) < self._cache_size:
            self._cache[piece] = out
        return out

    def decode_bytes(self, ids) -> bytes:
        return b"".join(self.tokens[int(i)] for i in ids)
